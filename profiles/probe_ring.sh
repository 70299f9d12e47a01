set -o pipefail
for rm in 0 256 512 1024 2048 100000; do
  for cfg in "fp32 4096" "bf16 4096" "bf16 16384"; do
    ADCDIFF_GEMM_RING_MAX=$rm timeout -k 10 120 python3 tools/step_breakdown.py $cfg 2>/dev/null | tail -1 | sed "s/^/ring_max=$rm $cfg: /" || exit 1
  done
done
