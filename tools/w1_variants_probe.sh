#!/bin/bash
# On the GPU box: the one-wave kernel (hint 32) against the variant with loader waves (hint 325), per shape; N K GROUP_SIZE LIVE_ROWS per line.
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_var
for cfg in "2048 4096 64 32" "2048 1024 64 32" "2048 3072 128 32" "2048 2048 128 32" "2048 4096 32 32" "2048 4096 32 64"; do
  for hint in 32 325; do
    set -- $cfg
    d=gpurun_out/prof_var/$1_$2_$3_$4_$hint
    timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 tools/small_m_probe.py $1 $2 $3 $hint 0 $4 > $d.log 2>&1 || { echo "failed $cfg $hint"; tail -3 $d.log; exit 1; }
    f=$(find $d -name "*kernel_stats.csv" | head -1)
    echo "N=$1 K=$2 gs=$3 live=$4 hint=$hint: $(python3 tools/kernel_avg.py $f gemm_)"
  done
done
