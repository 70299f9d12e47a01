#!/bin/bash
# Runs on the GPU box (via gpurun) from the repository root:  profiles/variants.sh TAG
# Every bench workload on ONE box, one JSON line each, into gpurun_out/<TAG>_bench_variants.jsonl (copied to profiles/ afterwards).
set -o pipefail
TAG=${1:-r03}
OUT=gpurun_out/${TAG}_bench_variants.jsonl
: > $OUT
run() { timeout -k 10 400 python3 bench.py "$@" 2>/dev/null | grep '"metric"' >> $OUT || { echo "failed: $*"; exit 1; }; }
run --steps 3 --warmup 1
run --steps 3 --warmup 1 --condition --no-cpu-baseline --no-fp32
run --steps 5 --warmup 1 --sampler ddim50 --no-cpu-baseline
run --workload fp32_b4096 --steps 2 --warmup 1 --no-cpu-baseline
run --workload train_dp --steps 20 --warmup 3
run --workload bf16_b32 --sampler ddim50 --steps 10 --warmup 2 --no-cpu-baseline
run --workload bf16_b32 --steps 3 --warmup 1 --no-cpu-baseline
run --workload shard_1m --steps 1 --warmup 0 --timesteps 100 --no-cpu-baseline
wc -l $OUT
