"""A stand-in for bench.py's HipBackend that renders with the CPU oracle and reduces over gloo, so that the script
the driver launches under torch.distributed.run (its argument handling, step / reduce / resolve sequence, max-over-ranks
timing and JSON line) runs on a machine without a GPU. Test infrastructure: lives under tests/, never imported by bench.py."""
import ctypes as C
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import oracle_lib  # noqa: E402
from pathtracer_amd import lib as ptlib  # noqa: E402
from pathtracer_amd.scene import PackedScene  # noqa: E402


class OracleBackend:
    dist_backend = "gloo"

    def __init__(self, local_rank: int):
        self.lib = ptlib.load()  # host half of the C ABI (BVH builder, packing, push constants): no GPU needed
        self.oracle = oracle_lib.load()
        self.tensor_device = torch.device("cpu")
        self.reduces = 0

    def pack(self, scene):
        return PackedScene(scene, self.lib)

    def upload(self, packed):
        self.packed = packed

    def alloc(self, pixels):
        self.pixels = pixels
        self.accum = torch.zeros(pixels * 4, dtype=torch.float32)
        self.ids = torch.zeros(pixels, dtype=torch.int32)
        self.rgba = torch.zeros(pixels, dtype=torch.int32)

    def render(self, rp, stats, want_ids=False):
        _, accum, ids, st = oracle_lib.render(self.oracle, self.packed, rp, threads=2, flags=oracle_lib.GPU_MIRROR_LDS)  # the test scene is LDS-sized
        self.accum.copy_(torch.from_numpy(accum.reshape(-1)))
        self.ids.copy_(torch.from_numpy(ids.reshape(-1).view(np.int32)))
        if stats is not None:
            for name, _ in stats._fields_:
                v = st[name]
                if not isinstance(v, dict):
                    setattr(stats, name, v)
            stats.render_ms = 1.0

    def reduce(self, dist):
        dist.reduce(self.accum, dst=0, op=dist.ReduceOp.SUM)
        self.reduces += 1

    def reduce_ms(self):
        return 0.0

    def resolve(self):
        a = self.accum.numpy().reshape(-1, 4)
        out = np.zeros((self.pixels, 4), dtype=np.uint8)
        px = (C.c_uint8 * 4)()
        for i in range(self.pixels):
            self.oracle.orc_resolve_pixel(C.c_float(a[i, 0]), C.c_float(a[i, 1]), C.c_float(a[i, 2]), C.c_float(a[i, 3]), px)
            out[i] = px[:]
        self.rgba.copy_(torch.from_numpy(out.view(np.int32).reshape(-1)))

    def sync(self):
        pass

    def host_frame(self):
        return (self.accum.numpy().reshape(-1, 4).copy(), self.ids.numpy().view(np.uint32).copy(),
                self.rgba.numpy().view(np.uint8).reshape(-1, 4).copy())

    def close(self):
        pass


if __name__ == "__main__":  # torch.distributed.run entry: bench.py's main() with the oracle backend injected
    import bench
    bench.main(sys.argv[1:], backend_factory=OracleBackend)
