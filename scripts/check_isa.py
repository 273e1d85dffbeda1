"""Search the gfx950 assembly of the three kernel compilations for the code shape behind round 2's GPU faults (DESIGN.md §5,
"Stack entries that are the child"): hipcc (ROCm 7.2) turned a wave-uniform `bool` tested inside the traversal loops into a LANE
MASK — `v_cndmask_b32 vN, 0, 1, s[a:b]` then `v_cmp_ne_u32_e64 s[c:d], 1, vN` — formed under one loop's partial exec mask and
reused under another's, so lanes disagreed about a branch the source calls uniform. The pair of instructions is harmless
where exec is full (one known instance: the prologue of k_shadow<true>); anywhere else it has to be read before the code runs
on a GPU. Runs on the CPU (hipcc cross-compiles): `python scripts/check_isa.py` prints the instances and exits 1 on an unknown one."""
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pathtracer_amd import build  # noqa: E402

# (mangled-name fragment, instances allowed): the prologue of the all-opaque shadow kernel, read in round 2 (full exec mask)
KNOWN = {"8k_shadowILb1EE": 1}


def assembly(src: str, flags: list[str], out_dir: str) -> str:
    out = os.path.join(out_dir, os.path.basename(src) + ".s")
    cmd = [build.hipcc(), "--offload-arch=gfx950", "--cuda-device-only", "-S", *flags, "-Wno-unused-command-line-argument", src, "-o", out]
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return out


def instances(path: str) -> list[tuple[str, int]]:
    lines = open(path).read().split("\n")
    fn, found = "?", []
    for i, line in enumerate(lines):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            fn = m.group(1)
        m = re.search(r"v_cndmask_b32(?:_e64)?\s+(v\d+), 0, 1, s\[", line)
        if not m:
            continue
        reg = m.group(1)
        for later in lines[i + 1:i + 16]:
            if re.search(r"v_cmp_ne_u32_e64 s\[\d+:\d+\], 1, " + reg + r"\b", later):
                found.append((fn, i + 1))
                break
    return found


def check(verbose: bool = True) -> list[str]:
    fast = [f if f != "-ffp-contract=off" else "-ffp-contract=fast" for f in build.COMMON]
    jobs = [(build.DEVICE_SRC[0], build.COMMON), (build.FAST_SRC, fast), (build.RS_SRC, build.COMMON)]
    unknown = []
    with tempfile.TemporaryDirectory() as tmp, ThreadPoolExecutor(max_workers=3) as pool:
        for (src, _), path in zip(jobs, pool.map(lambda j: assembly(j[0], j[1], tmp), jobs)):
            per_fn: dict[str, int] = {}
            for fn, line in instances(path):
                per_fn[fn] = per_fn.get(fn, 0) + 1
                if verbose:
                    print(f"{os.path.basename(src)}: {fn} (assembly line {line})")
            for fn, n in per_fn.items():
                allowed = max((k for frag, k in KNOWN.items() if frag in fn), default=0)
                if n > allowed:
                    unknown.append(f"{os.path.basename(src)}: {fn}: {n} instance(s), {allowed} known")
    return unknown


if __name__ == "__main__":
    bad = check()
    for b in bad:
        print("UNKNOWN:", b)
    print("ok" if not bad else "read these before running on a GPU")
    sys.exit(1 if bad else 0)
