#!/bin/bash
# Runs on the GPU box (via gpurun) from the repository root:  profiles/collect.sh TAG [bench.py arguments ...]
# Kernel trace of the benchmark command, the two HBM-traffic counter passes and one matrix-core counter pass, each rocprofv3
# run on its own as the pool requires (never --pmc together with other trace domains).  Outputs go to gpurun_out/prof_<tag>/
# and are summarised into profiles/ by profiles/summarize.py afterwards.
set -o pipefail
TAG=${1:-r02}
shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline $*"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH --timesteps 250 > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace.log; exit 1; }
grep -h '"metric"' $OUT/trace.log > $OUT/bench_line.json
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $BENCH --timesteps 12 > $OUT/pmc_fetch.log 2>&1 || { echo "pmc fetch failed"; tail -5 $OUT/pmc_fetch.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $BENCH --timesteps 12 > $OUT/pmc_write.log 2>&1 || { echo "pmc write failed"; tail -5 $OUT/pmc_write.log; exit 1; }
# matrix-core utilisation: cycles the MFMA pipes are busy against the cycles the shader engines are busy (seven SQ counters, one pass)
timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_COEXEC_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_mfma -- $BENCH --timesteps 12 > $OUT/pmc_mfma.log 2>&1 || { echo "pmc mfma failed"; tail -5 $OUT/pmc_mfma.log; exit 1; }
find $OUT -name "*kernel_trace.csv" -size +30M -delete
ls -R $OUT | head -40
