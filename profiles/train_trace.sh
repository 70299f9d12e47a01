#!/bin/bash
# Runs on the GPU box from the repository root: profiles/train_trace.sh TAG
# Kernel trace of the data-parallel training-step workload (one GPU): per-kernel time of forward + backward + clip + AdamW + re-pack.
set -o pipefail
TAG=${1:-train}
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG} -- python3 bench.py --workload train_dp --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/prof_${TAG}.log 2>&1 || { tail -5 gpurun_out/prof_${TAG}.log; exit 1; }
grep -h '"metric"' gpurun_out/prof_${TAG}.log
f=$(find gpurun_out/prof_${TAG} -name "*kernel_stats.csv")
cut -d, -f1-4,8 $f | cut -c1-200
find gpurun_out/prof_${TAG} -name "*kernel_trace.csv" -size +30M -delete
