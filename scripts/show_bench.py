#!/usr/bin/env python3
"""Print the headline numbers of bench.py JSON lines: scripts/show_bench.py gpurun_out/a.json ..."""
import json
import sys
for f in sys.argv[1:]:
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "unreadable:", e)
        continue
    print(f"{f}: {d['value']:.0f} Mrays/s  {d['mpaths_per_s']:.0f} Mpaths/s  {d['ms_per_step']:.1f} ms/step  iters {d['config']['iterations']}")
    for k, v in d.get("kernels", {}).items():
        print(f"   {k:7s} launches {v['launches']:5d}  avg {v['avg_launch_us']:8.1f} us  {v.get('algorithmic_gbs', 0.0):7.0f} GB/s  {v.get('algorithmic_tflops', 0.0):.2f} TF")
    c = d.get("counters")
    if c:
        print(f"   rays {c['rays']}  node visits {c['node_visits']}  tri tests {c['tri_tests']}  shortcut {c['rays_shadow_shortcut']}")
    if "cpu_baseline" in d:
        print("   cpu:", d["cpu_baseline"]["value"], d["cpu_baseline"]["unit"], d["cpu_baseline"]["cores"], "threads")
