"""Per-launch shape of a k_paths frame: live slots and device time of every launch (PT_LOG_LAUNCHES + PT_FLAG_TIME_KERNELS).
Usage: python scripts/frame_shape.py [width height sample_count]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("PT_FRAME_SHAPE_CHILD"):
    sys.path.insert(0, ROOT)
    import torch  # noqa: F401
    from pathtracer_amd import abi, scenes
    from pathtracer_amd.renderer import Renderer
    w, h, spp = (int(a) for a in sys.argv[1:4])
    r = Renderer(0)
    p = r.pack(scenes.cornell_scene())
    r.upload(p)
    rp = p.render_params(r.lib, w, h, spp, seed=0x5EED0000)
    r.render_params(rp, want_accum=False, want_ids=False)
    rp.flags = abi.FLAG_TIME_KERNELS
    print("[pt] ---- timed frame", file=sys.stderr, flush=True)
    _, _, _, st = r.render_params(rp, want_accum=False, want_ids=False)
    print(f"RESULT render_ms {st['render_ms']:.3f} rays {st['rays']} paths {st['paths']} shaded {st['shaded_vertices']} launches {st['iterations']}")
    sys.exit(0)

args = sys.argv[1:4] if len(sys.argv) >= 4 else ["1920", "1080", "128"]
env = dict(os.environ, PT_FRAME_SHAPE_CHILD="1", PT_LOG_LAUNCHES="1")
p = subprocess.run([sys.executable, os.path.abspath(__file__), *args], env=env, capture_output=True, text=True)
if p.returncode != 0:
    print(p.stderr[-2000:])
    sys.exit(1)
err = p.stderr.split("[pt] ---- timed frame")[-1]
live = [int(m.group(1)) for m in re.finditer(r"(\d+) slots live after", err)]
ms = [float(m.group(1)) for m in re.finditer(r"launch \d+: kernel \d+, ([0-9.]+) ms", err)]
w, h = int(args[0]), int(args[1])
slots = ((w + 7) // 8) * ((h + 7) // 8) * 64
# the count polled at launch k is the queue AFTER launch k-2, i.e. the input of launch k-1
inputs = [slots] + live[1:] if live else [slots]
print(p.stdout.strip())
print(f"{'launch':>6} {'live slots in':>14} {'ms':>8} {'us per 1000 live':>18}")
for k, t in enumerate(ms):
    n = inputs[k] if k < len(inputs) else None
    print(f"{k:6d} {n if n is not None else '?':>14} {t:8.3f} " + (f"{1e3 * t / (n / 1e3):18.2f}" if n else ""))
print(f"sum of launches {sum(ms):.2f} ms")
