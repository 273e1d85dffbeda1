#!/bin/bash
# Experiment: frame time vs band size (PT_BAND_SLOTS) for bench.py's workload.
for b in 0 1048576 524288 262144 131072; do
  PT_BAND_SLOTS=$b timeout -k 10 300 python3 -u bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/band_$b.json 2> gpurun_out/band_$b.err || { echo "band $b failed"; tail -3 gpurun_out/band_$b.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/band_$b.json')); print('band', $b, 'Mrays/s %.0f' % d['value'], 'ms/step %.1f' % d['ms_per_step'], {k: round(v,1) for k,v in d['roofline']['kernel_ms'].items()}, 'iters', d['config']['iterations'])"
done
