#!/bin/bash
# On the GPU box: the one-wave kernel with its full ring (hint 32) against a four-stage ring (hint 324), per shape.
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_ring
for cfg in "2048 4096 64 32" "2048 4096 64 324" "2048 4096 32 32" "2048 4096 32 324" "2048 1024 32 32" "2048 1024 32 324"; do
  set -- $cfg
  d=gpurun_out/prof_ring/$1_$2_$3_$4
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 tools/small_m_probe.py $1 $2 $3 $4 > $d.log 2>&1 || { echo "failed $cfg"; tail -3 $d.log; exit 1; }
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  echo "N=$1 K=$2 gs=$3 hint=$4: $(python3 tools/kernel_avg.py $f gemm_)"
done
