"""Static checks on the gfx950 ISA of the hand-scheduled GEMM kernels (hipcc cross-compiles without a GPU; ~1 minute).

The kernels that issue LDS-DMA from inline assembly rely on facts the compiler does not know about, so they are checked on
the ISA it actually produced:
* no scratch (a spill inside a loop that counts vmcnt by hand would add loads to the count);
* M0 is only written by the LDS-DMA sequences (the kernels leave it modified, which is fine as long as nothing else reads it);
* the loader / compute variant has exactly the five barrier sites whose execution counts are equal by construction (one in the
  loaders' loop, four in the compute wave: first tile, two in the unrolled pair, one in the odd tail), and the one-wave
  kernel has none."""
import glob
import os
import re
import shutil
import subprocess

import pytest

SRC = glob.glob("2025-iflytek*/csrc/gemm.hip")[0]
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def isa(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa") / "gemm.s"
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only", "-o", str(out), SRC],
                   check=True, capture_output=True, timeout=900)
    return out.read_text()


def _kernel(isa, mangled):
    a = isa.index(mangled + ":")
    body = isa[a:isa.index(".end_amdhsa_kernel", a)]
    lines = [ln.strip().split(";")[0].strip() for ln in body.split("\n")]
    lines = [ln for ln in lines if ln and not ln.startswith(".")]
    scratch = int(re.search(r"\.amdhsa_private_segment_fixed_size\s+(\d+)", body).group(1))
    return lines, scratch


def _m0_writes_outside_dma(lines):
    ok = re.compile(r"^(s_mov_b32 m0, s\d+|s_add_u32 m0, m0, 0x1000)$")
    return [ln for ln in lines if "m0" in ln and not ok.match(ln)]


@pytest.mark.parametrize("bm,bn", [(32, 32), (32, 64), (32, 128), (64, 32), (64, 64), (64, 128)])
def test_one_wave_kernel_isa(isa, bm, bn):
    lines, scratch = _kernel(isa, "_ZN7adcdiff19gemm_w1_bf16_kernelILi%dELi%dEEEvNS_10GemmParamsE" % (bm, bn))
    assert scratch == 0
    assert _m0_writes_outside_dma(lines) == []
    assert lines.count("s_barrier") == 0
    pieces = (bm + bn) // 8
    assert sum("offen" in ln and ln.endswith("lds") for ln in lines) % pieces == 0   # whole tiles of LDS-DMA pieces only


@pytest.mark.parametrize("bm,bn", [(32, 32), (32, 64), (32, 128), (64, 32), (64, 64), (64, 128)])
def test_loader_wave_kernel_isa(isa, bm, bn):
    lines, scratch = _kernel(isa, "_ZN7adcdiff19gemm_cl_bf16_kernelILi%dELi%dEEEvNS_10GemmParamsE" % (bm, bn))
    assert scratch == 0
    assert _m0_writes_outside_dma(lines) == []
    assert lines.count("s_barrier") == 5


def test_four_wave_kernel_has_no_scratch(isa):
    # epilogue batches of 1, 2 and 4 (the default) rows per lane; the batch of 8 is a measurement aid that spills in its epilogue
    # (second parameter: what the epilogue knows about its side inputs; 1 and 2 are the variants that look ahead for the residual)
    for u, side in ((1, 0), (2, 0), (4, 0), (4, 1), (4, 2)):
        _, scratch = _kernel(isa, "_ZN7adcdiff20gemm256w_bf16_kernelILi%dELi%dEEEvNS_10GemmParamsE" % (u, side))
        assert scratch == 0
