#!/bin/bash
# Runs on the GPU box from the repository root: profiles/small_batch.sh TAG
# Wall time of the generator's default call (32 latents) and a kernel trace of the same script.
set -o pipefail
TAG=${1:-small}
export TMPDIR=/tmp
timeout -k 10 200 python3 tools/small_batch_latency.py 32 bf16 > gpurun_out/${TAG}.json 2> gpurun_out/${TAG}.err || { tail -5 gpurun_out/${TAG}.err; exit 1; }
cat gpurun_out/${TAG}.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG} -- python3 tools/small_batch_latency.py 32 bf16 > gpurun_out/prof_${TAG}.log 2>&1 || { tail -5 gpurun_out/prof_${TAG}.log; exit 1; }
f=$(find gpurun_out/prof_${TAG} -name "*kernel_stats.csv")
cut -d, -f1-4 $f | cut -c1-150
find gpurun_out/prof_${TAG} -name "*kernel_trace.csv" -size +30M -delete
