"""In-tree build of libpathtracer_amd.so (hipcc, gfx950) and of the `pathtracer` CLI.

`python -m pathtracer_amd.build` or `__graft_entry__.build()`.  hipcc cross-compiles
without a GPU; the built .so travels to the GPU box with the source tree.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpathtracer_amd.so")
CLI = os.path.join(HERE, "pathtracer")
OBJ_DIR = os.path.join(HERE, "build")  # intermediate objects (git-ignored)

# -ffp-contract=off: the arithmetic contract — no FMA is ever formed, on the
# host (BVH builder, camera) or the device, so CPU and GPU evaluate identical expressions
# to identical bits.  IEEE-correct fp32 divide and sqrt are hipcc's defaults for HIP.
COMMON = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fvisibility=hidden",
          "-Wall", "-Wno-unused-function", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(CSRC, "device")]
DEVICE_SRC = [os.path.join(CSRC, "device", f) for f in ("pt_kernels.hip", "pt_context.hip")]
FAST_SRC = os.path.join(CSRC, "device", "pt_kernels_fast.hip")
RS_SRC = os.path.join(CSRC, "device", "pt_kernels_rs.hip")  # the traversal kernels with (record, side) stack entries
HOST_SRC = [os.path.join(CSRC, "host", f) for f in ("bvh_builder.cpp", "scene_pack.cpp")]


def _newer(target: str, sources: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    deps = list(sources)
    for d in (os.path.join(ROOT, "include"), os.path.join(CSRC, "device"), os.path.join(CSRC, "host")):
        deps += [os.path.join(d, f) for f in os.listdir(d)]
    deps.append(os.path.abspath(__file__))
    return any(os.path.getmtime(s) > t for s in deps)


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def build(force: bool = False, verbose: bool = False) -> str:
    host_extra = [os.path.join(CSRC, "host", f) for f in sorted(os.listdir(os.path.join(CSRC, "host")))
                  if f.endswith(".cpp") and os.path.join(CSRC, "host", f) not in HOST_SRC and f not in ("main.cpp", "renderer.cpp")]
    srcs = DEVICE_SRC + HOST_SRC + host_extra
    if force or _newer(LIB, srcs):
        extra = os.environ.get("PT_EXTRA_HIPCC_FLAGS", "").split()
        # every translation unit to its own object (in parallel), then one link. pt_kernels_fast.hip is the same kernel
        # source with FMA contraction allowed: the fast-arithmetic build of the path kernel.
        os.makedirs(OBJ_DIR, exist_ok=True)
        fast_flags = [f if f != "-ffp-contract=off" else "-ffp-contract=fast" for f in COMMON]
        jobs = [(src, COMMON) for src in srcs] + [(FAST_SRC, fast_flags), (RS_SRC, COMMON)]
        objs = []
        cmds = []
        for src, flags in jobs:
            obj = os.path.join(OBJ_DIR, os.path.basename(src) + ".o")
            objs.append(obj)
            cmds.append([hipcc(), "--offload-arch=gfx950", "-c", *flags, *extra, src, "-o", obj])
        if verbose:
            for cmd in cmds:
                print(" ".join(cmd))
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(8, len(cmds))) as pool:
            for rc in pool.map(subprocess.call, cmds):
                if rc != 0:
                    raise subprocess.CalledProcessError(rc, "hipcc -c")
        link = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", os.environ.get("PT_LIB_OUT", LIB), "-lz", "-pthread"]
        if verbose:
            print(" ".join(link))
        subprocess.check_call(link)
    # the CLI: the reference's main.cc shape — front end + Renderer binding compiled as plain C++,
    # the device path reached only through the C ABI of the shared library
    main_cpp = os.path.join(CSRC, "host", "main.cpp")
    cli_src = [main_cpp] + [os.path.join(CSRC, "host", f) for f in ("renderer.cpp", "gltf_loader.cpp", "image_io.cpp", "jpeg_decode.cpp")]
    if os.path.exists(main_cpp) and (force or _newer(CLI, cli_src + [LIB])):
        cmd = ["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-Wall", "-I", os.path.join(ROOT, "include"), *cli_src,
               "-o", CLI, "-L", HERE, "-lpathtracer_amd", "-lz", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(LIB)
