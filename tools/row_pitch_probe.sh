#!/bin/bash
# On the GPU box: does the row pitch of W matter at a small batch?  Power-of-two pitches put the same 128-byte line of every
# row on the same memory channels.  N K GROUP_SIZE HINT PAD per line.
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_pitch
for cfg in "2048 4096 64 0 0" "2048 4096 64 0 64" "2048 4096 64 0 192" "2048 4096 32 0 0" "2048 4096 32 0 64" "2048 2048 64 0 0" "2048 2048 64 0 64" "2048 1024 32 0 0" "2048 1024 32 0 64" "2048 4096 64 128 0" "2048 4096 64 128 64"; do
  set -- $cfg
  d=gpurun_out/prof_pitch/$1_$2_$3_$4_$5
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 tools/small_m_probe.py $1 $2 $3 $4 $5 > $d.log 2>&1 || { echo "failed $cfg"; tail -3 $d.log; exit 1; }
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  echo "N=$1 K=$2 gs=$3 hint=$4 pad=$5: $(python3 tools/kernel_avg.py $f gemm_)"
done
