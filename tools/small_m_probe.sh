#!/bin/bash
# On the GPU box: per-shape kernel durations of the small-batch GEMMs (rocprofv3 kernel trace, one process per shape).
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_smallm
for cfg in "2048 256 64" "2048 1024 64" "2048 2048 64" "2048 4096 64" "2048 256 32" "2048 1024 32" "2048 2048 32" "2048 4096 32" "256 1024 8" "256 4096 8"; do
  set -- $cfg
  d=gpurun_out/prof_smallm/$1_$2_$3
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 tools/small_m_probe.py $1 $2 $3 > $d.log 2>&1 || { echo "failed $cfg"; tail -3 $d.log; exit 1; }
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  echo "N=$1 K=$2 gs=$3: $(python3 tools/kernel_avg.py $f gemm_)"
done
