set -u
export TMPDIR=/tmp
for v in head base; do
  export PT_LIB_PATH=$PWD/variants/$v.so
  out=gpurun_out/pmc_probe_$v; mkdir -p $out
  timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU --output-format csv -d $out/pass1 -- python3 bench.py --steps 1 --warmup 0 --configs none --no-fast-arith --no-cpu-baseline --no-kernel-timing --workload atrium1080 --spp 16 > $out/pass1.log 2>&1 || { echo "pass failed $v"; tail -n 5 $out/pass1.log; exit 1; }
  python3 scripts/pmc_summarize.py $out > /dev/null
  echo "== $v"; grep -A 9 "k_paths_merged" $out/pmc_summary.txt
  rm -rf $out/pass1
done
