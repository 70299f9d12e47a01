set -o pipefail
export TMPDIR=/tmp
for no in -1 1 3; do timeout -k 10 60 python3 tools/gemm_probe.py 2048 2048 $no 256 20 2>&1 | grep -v amdgpu.ids || exit 1; done
timeout -k 10 60 python3 tools/gemm_probe.py 2048 1024 1 256 20 2>&1 | grep -v amdgpu.ids || exit 1
