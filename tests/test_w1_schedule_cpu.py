"""CPU model of the one-wave GEMM kernel's LDS ring (gemm_w1_bf16_kernel in csrc/gemm.hip): the wait constants are read from
the source and replayed against an in-order load counter, the way the hardware's vmcnt behaves.

What the model checks, for every K-tile count and every tile shape:
* never more than 63 loads in flight (the 6-bit counter wraps beyond that and later waits return early);
* every fragment read of a tile comes after a wait that guarantees all of the tile's LDS-DMA pieces have landed;
* a stage is only overwritten after the tile it held has been read into registers;
* the epilogue starts with nothing in flight (it reuses the ring's memory).
The model is pessimistic: a load only counts as landed once a wait forces it."""
import glob
import re

import pytest

SRC = glob.glob("2025-iflytek*/csrc/gemm.hip")[0]


def _kernel_text():
    s = open(SRC).read()
    a = s.index("void gemm_w1_bf16_kernel(const GemmParams p)")
    return s[a:s.index("static int launch_w1", a)]


def _consts(bn, bm=32):
    """P, D and the four wait constants, evaluated from the expressions in the source."""
    k = _kernel_text()
    env = {"BM": bm, "BN": bn}
    env["AP"], env["WP"] = env["BM"] // 8, bn // 8
    env["P"] = env["AP"] + env["WP"]
    m = re.search(r"constexpr int D = ([^;]+);", k)
    expr = m.group(1).replace("/", "//")
    cond, yes, no = re.match(r"(.+?) \? (.+?) : (.+)", expr).groups()
    env["D"] = eval(yes, {}, env) if eval(cond, {}, env) else eval(no, {}, env)
    waits = [eval(w.replace("/", "//"), {}, env) for w in re.findall(r"wait_vmcnt<([^>]+)>\(\)", k)]
    assert len(waits) == 4, waits   # before the last prologue tile, before the first fragment reads, per iteration, before the epilogue
    return env["P"], env["D"], waits


class InOrderCounter:
    def __init__(self):
        self.issued = []      # tags in issue order
        self.landed = 0       # how many of the oldest are known to have landed
        self.peak = 0

    def issue(self, tag):
        self.issued.append(tag)
        self.peak = max(self.peak, len(self.issued) - self.landed)

    def wait(self, n):
        self.landed = max(self.landed, len(self.issued) - n)

    def has_landed(self, tag):
        idx = [i for i, t in enumerate(self.issued) if t == tag]
        return bool(idx) and idx[-1] < self.landed


@pytest.mark.parametrize("bm", [32, 64])
@pytest.mark.parametrize("bn", [32, 64, 128])
@pytest.mark.parametrize("nt", list(range(1, 20)) + [31, 32, 64, 65])
def test_ring_schedule_of_the_one_wave_kernel(bn, nt, bm):
    P, D, (w_room, w_first, w_iter, w_end) = _consts(bn, bm)
    assert D >= 3
    assert (D - 1) * P + 3 <= 63
    c = InOrderCounter()
    stage_tile = {}           # stage -> tile whose pieces were issued into it last
    in_regs = set()           # tiles whose fragments have been read
    param_loads = 3           # the epilogue's vectors: at most this many compiler-tracked loads at a time

    def issue_tile(tile, stage):
        prev = stage_tile.get(stage)
        assert prev is None or prev in in_regs or prev >= nt, (tile, stage, prev)   # overwritten only after it was read
        stage_tile[stage] = tile
        for m in range(P):
            c.issue(("piece", min(tile, nt - 1) if tile >= nt else tile, m, tile))

    def read_tile(tile, stage):
        if tile >= nt:
            return            # reads past the last tile feed nothing
        assert stage_tile[stage] == tile
        for m in range(P):
            assert c.has_landed(("piece", tile, m, tile)), (tile, m)
        in_regs.add(tile)

    # prologue
    for k in range(D):
        if k == D - 1:
            for i in range(param_loads):
                c.issue(("param", i))
            c.wait(0)         # the compiler waits for its own loads, which are the youngest
            c.wait(w_room)
        issue_tile(k, k)
    c.wait(w_first)
    read_tile(0, 0)
    # main loop: iteration t issues tile t + D into the stage of tile t and reads tile t + 1
    slot = 0
    for t in range(nt):
        dst = slot
        slot = 0 if slot == D - 1 else slot + 1
        c.wait(w_iter)
        assert t in in_regs   # the MFMAs of tile t use fragments read one iteration earlier
        issue_tile(t + D, dst)
        read_tile(t + 1, slot)
    c.wait(w_end)
    assert c.landed == len(c.issued)
    assert c.peak <= 63, c.peak
    assert in_regs == set(range(nt))
