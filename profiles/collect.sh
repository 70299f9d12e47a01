#!/bin/bash
# Runs on the GPU box (via gpurun) from the repository root: kernel trace of the benchmark command and the
# two HBM-traffic counter passes, each rocprofv3 run on its own as the pool requires.  Outputs go to
# gpurun_out/prof_<tag>/ and are summarised into profiles/ by profiles/summarize.py afterwards.
set -o pipefail
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH --timesteps 250 > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace.log; exit 1; }
grep -h '"metric"' $OUT/trace.log > $OUT/bench_line.json
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $BENCH --timesteps 12 > $OUT/pmc_fetch.log 2>&1 || { echo "pmc fetch failed"; tail -5 $OUT/pmc_fetch.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $BENCH --timesteps 12 > $OUT/pmc_write.log 2>&1 || { echo "pmc write failed"; tail -5 $OUT/pmc_write.log; exit 1; }
find $OUT -name "*kernel_trace.csv" -size +30M -delete
ls -R $OUT | head -40
