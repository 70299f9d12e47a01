#!/usr/bin/env python3
"""Generates w4_mainloop.inc: the hand-allocated gfx950 main loop of the 4-wave 256 x 256 bf16 GEMM tile.

One wave per SIMD computes a 128 x 128 block of the tile: 64 accumulators of 16 x 16 in a[0:255], two sets of
operand fragments in v[0:127].  hipcc cannot allocate such a loop (it spills hundreds of registers), so the
whole loop is one asm statement with every register named here; the HIP side pins its inputs and the 64
accumulator outputs to the same physical registers through {reg} constraints.

LDS: two stages of 64 KiB, each a whole 64-wide K tile: A rows at 128 * row, W rows at 32768 + 128 * row, the
16-byte chunk c of a row at chunk position c ^ ((row >> 1) & 7) (the image of gemm256_bf16_kernel: a ds_read_b128 of a
16-row fragment is conflict-free).  A tile is two k halves of 32 (one v_mfma_f32_16x16x32_bf16 per 16 x 16 block each).
Tile t lives in stage t & 1.
  phase A(t): the 64 MFMAs of k half 0 from register set 0; reads the fragments of k half 1 of tile t into set 1.
              End: lgkmcnt(0) (stage t & 1 is no longer read by this wave), vmcnt(0) (this wave's pieces of tile
              t + 1 have landed), the ONE barrier of the tile.
  phase B(t): the 64 MFMAs of k half 1 from set 1; reads the fragments of k half 0 of tile t + 1 (other stage) into
              set 0; issues this wave's sixteen LDS-DMA pieces of tile t + 2 into stage t & 1.  End: lgkmcnt(0).
Behind the pieces of tile t + 2 a wave issues ONE buffer_load_dword (result never read) whose lanes touch cache lines of
tile t + 2 + PF (a cursor of its own that walks the segments like the loader's): the HIP side points the lanes of one A-loading wave at this workgroup's share of the A rows (the
workgroups of one tile row share the A rows and split them), every other wave at a single line (which costs nothing).  The
LDS-DMA of the A rows then hits in L2.  Without it the loop waits on the A operand, which streams from HBM while the
workgroups that share it ask for the same lines at the same time: with every wave loading L2-resident rows the loop runs
2425 cycles per tile, with the A rows from HBM 3470.  The address unit's time grows with the lines an instruction touches
(a prefetch of all 512 lines of a tile by every workgroup cost more than it gained), hence the split.  This load is younger
than the pieces, so the phase-A wait is vmcnt(1).
A piece is 8 rows x 128 B = eight whole cache lines per instruction: the address unit's time per LDS-DMA instruction
grows with the number of lines it touches (16 rows x 64 B per piece measured 1.4 x the loop time of this layout).
Waves 0, 1 load the A rows (128 each), waves 2, 3 the W rows; the code is the same, the inputs differ.  The loader
walks up to three K segments (the A operand may be [A0 | A1 | A2] along K); a segment switch reloads the buffer
descriptor, the per-lane offsets and the piece stride.

Register map (inputs are read-only):
  a[0:255]      accumulators, block (i, j) at a[4 (8 i + j) : +3]
  v[0:31]       W fragments, set 0      v[32:63]   A fragments, set 0
  v[64:95]      W fragments, set 1      v[96:127]  A fragments, set 1
  v128 / v129   A / W fragment read address of the phase
  v130 / v131   in: per-lane A / W fragment offset inside a stage, k half 0      v132 / v133  in: the same, k half 1
  v134 / v135   per-lane source offset of the current segment, even / odd pieces
  v136..v141    in: per-lane source offsets {even, odd} for segment 0, 1, 2
  v142          prefetch result (never read)   v143  per-lane prefetch offset of the current segment
  v144..v146    in: per-lane prefetch offset for segment 0, 1, 2
  s40..s54      in: per segment {base lo, base hi (16 bits), row pitch in bytes, K tiles, bytes addressable}
  s55           in: K tiles in total (>= 2)      s56  in: LDS address of this wave's first piece in stage 0
  s[60:63]      buffer descriptor   s64 k byte offset in the segment   s65 tiles left in the segment
  s66 piece stride (8 rows)  s67 piece source offset  s68 stage offset of tile t (0 / 0x10000)  s70 loop count
  s71 segment index  s72 scratch  s73 the caller's M0  s74 stage offset of the other stage
  s[80:83], s84, s85, s86  the prefetch cursor: descriptor, k byte offset, tiles left in its segment, segment index
"""
import sys

A_SET = [32, 96]
W_SET = [0, 64]
STAGE = 0x10000
PF = 4           # prefetch distance in K tiles beyond the tile being staged
L = []
ABLATE = set()   # lab only: "nodma", "noread", "nobar", "nowait", "dmafirsthalf", "nopf"; "stamps": the measurement build's loop


def e(s):
    L.append(s)


def mfma(i, j, s):
    acc = 4 * (8 * i + j)
    a = A_SET[s] + 4 * i
    w = W_SET[s] + 4 * j
    e(f"v_mfma_f32_16x16x32_bf16 a[{acc}:{acc + 3}], v[{a}:{a + 3}], v[{w}:{w + 3}], a[{acc}:{acc + 3}]")


def read_frag(r, s):
    """fragment read r (0..7: W fragments, 8..15: A fragments) into set s from the addresses in v128 / v129"""
    if "noread" in ABLATE:
        return
    if r < 8:
        d = W_SET[s] + 4 * r
        e(f"ds_read_b128 v[{d}:{d + 3}], v129 offset:{r * 2048}")
    else:
        d = A_SET[s] + 4 * (r - 8)
        e(f"ds_read_b128 v[{d}:{d + 3}], v128 offset:{(r - 8) * 2048}")


def dma_piece(j):
    if "nodma" not in ABLATE:
        e(f"buffer_load_dwordx4 v{134 + (j & 1)}, s[60:63], s67 offen lds")
    e("s_add_u32 s67, s67, s66")
    e("s_add_u32 m0, m0, 0x400")


def prefetch():
    """one dword of the cache lines the lanes of v143 point at, at the prefetch cursor"""
    if "nopf" in ABLATE:
        return
    e("buffer_load_dword v142, v143, s[80:83], s84 offen")


def begin_issue(stage_reg):
    e(f"s_add_u32 m0, s56, {stage_reg}")
    e("s_mov_b32 s67, s64")


uid = [0]


def advance_cursor(pf=False):
    """next K tile of the loader's (pf: the prefetch) cursor, switching segment when this one is used up"""
    if pf and "nopf" in ABLATE:
        return
    n = uid[0]
    uid[0] += 1
    srd, kcur, left, seg = (80, 84, 85, 86) if pf else (60, 64, 65, 71)
    e(f"s_add_u32 s{kcur}, s{kcur}, 0x80")
    e(f"s_sub_u32 s{left}, s{left}, 1")
    e(f"s_cmp_eq_u32 s{left}, 0")
    e(f"s_cbranch_scc0 .Lw4_noswitch{n}_%=")
    e(f"s_add_u32 s{seg}, s{seg}, 1")
    e(f"s_cmp_eq_u32 s{seg}, 1")
    e(f"s_cselect_b32 s{srd}, s45, s50")
    e(f"s_cselect_b32 s{srd + 1}, s46, s51")
    e(f"s_cselect_b32 s{srd + 2}, s49, s54")
    e("s_cselect_b32 s72, s47, s52")
    e(f"s_cselect_b32 s{left}, s48, s53")
    e(f"s_mov_b32 s{kcur}, 0")
    e(f"s_cbranch_scc1 .Lw4_seg1_{n}_%=")
    if pf:
        e("v_mov_b32 v143, v146")
    else:
        e("v_mov_b32 v134, v140")
        e("v_mov_b32 v135, v141")
    e(f"s_branch .Lw4_segdone{n}_%=")
    e(f".Lw4_seg1_{n}_%=:")
    if pf:
        e("v_mov_b32 v143, v145")
    else:
        e("v_mov_b32 v134, v138")
        e("v_mov_b32 v135, v139")
    e(f".Lw4_segdone{n}_%=:")
    if not pf:
        e("s_lshl_b32 s66, s72, 3")
    e(f".Lw4_noswitch{n}_%=:")


def phase(s, read_stage, read_half, issue_stage, end):
    """64 MFMAs on register set s; read_stage / read_half: where the 16 fragment reads into the other set come from
    (None: no reads); issue_stage: the SGPR with the stage the 16 LDS-DMA pieces go to (None: none); end: lines"""
    if read_stage is not None:
        e(f"v_add_u32 v128, {read_stage}, v{130 + 2 * read_half}")
        e(f"v_add_u32 v129, {read_stage}, v{131 + 2 * read_half}")
    if issue_stage is not None:
        begin_issue(issue_stage)
    first_half = "dmafirsthalf" in ABLATE
    for x in range(64):
        i = x >> 3
        j = 7 - (x & 7) if (i & 1) else (x & 7)
        mfma(i, j, s)
        if read_stage is not None and "readlate" in ABLATE:
            if (x & 1) and x >= 32:
                read_frag((x - 32) >> 1, s ^ 1)
        elif read_stage is not None and (x & 1) and x < 32:
            read_frag(x >> 1, s ^ 1)
        if issue_stage is not None:
            if first_half:
                if x % 2 == 0 and x < 32:
                    dma_piece(x >> 1)
            elif x % 4 == 2:
                dma_piece(x >> 2)
    if issue_stage is not None:
        prefetch()
        advance_cursor(pf=True)
        advance_cursor()
    for line in end:
        if line == "STAMP":
            stamp_keep()
            continue
        if line.startswith("s_waitcnt vmcnt") and "nowait" in ABLATE:
            continue
        if line == "s_barrier" and "nobar" in ABLATE:
            continue
        e(line)


def stamp_issue():
    """measurement build: read the 100 MHz counter (a scalar load: it returns with the next lgkmcnt(0))"""
    if "stamps" in ABLATE:
        e("s_memrealtime s[88:89]")


def stamp_keep():
    """... and, behind that wait, keep its low word in lane s90 of v147 (one lane per stamp, at most 64)"""
    if "stamps" in ABLATE:
        e("v_mov_b32 v148, s88")                 # (v_writelane would need two scalar operands: one constant-bus read per instruction)
        e("v_cmp_eq_u32 vcc, s90, v149")
        e("v_cndmask_b32 v147, v147, v148, vcc")
        e("s_add_u32 s90, s90, 1")
        e("s_min_u32 s90, s90, 63")


def tile(issue, last=False):
    stamp_issue()   # at the start of a tile; kept at the end of its phase A
    keep = ["s_waitcnt lgkmcnt(0)", "STAMP"] if "stamps" in ABLATE else ["s_waitcnt lgkmcnt(0)"]
    phase(0, "s68", 1, None, keep if last else keep + ["s_waitcnt vmcnt(0)" if "nopf" in ABLATE else "s_waitcnt vmcnt(1)", "s_barrier"])
    if last:
        phase(1, None, 0, None, [])
    else:
        phase(1, "s74", 0, "s68" if issue else None, ["s_waitcnt lgkmcnt(0)"])
    e(f"s_xor_b32 s68, s68, {STAGE:#x}")
    e(f"s_xor_b32 s74, s74, {STAGE:#x}")


def generate():
    global PF
    PF = 4
    for f in ABLATE:
        if f.startswith("pf") and f[2:].isdigit():
            PF = int(f[2:])
    # ---- set-up
    if "stamps" in ABLATE:   # lane 0: entry
        e("s_mov_b32 s90, 0")
        e("v_mov_b32 v147, 0")
        e("v_mbcnt_lo_u32_b32 v149, -1, 0")
        e("v_mbcnt_hi_u32_b32 v149, -1, v149")
        stamp_issue()
        e("s_waitcnt lgkmcnt(0)")
        stamp_keep()
    e("s_mov_b32 s73, m0")
    e("s_mov_b32 s60, s40")
    e("s_mov_b32 s61, s41")
    e("s_mov_b32 s62, s44")
    e("s_mov_b32 s63, 0x00020000")
    e("s_mov_b32 s64, 0")
    e("s_mov_b32 s65, s43")
    e("s_lshl_b32 s66, s42, 3")
    e("s_mov_b32 s68, 0")
    e(f"s_mov_b32 s74, {STAGE:#x}")
    e("s_mov_b32 s71, 0")
    e("v_mov_b32 v134, v136")
    e("v_mov_b32 v135, v137")
    # prefetch cursor: starts at tile 2; the tiles the loop's own prefetch will not reach are touched here (the oldest
    # loads, so every later wait covers them)
    if "nopf" not in ABLATE:
        e("s_mov_b32 s80, s40")
        e("s_mov_b32 s81, s41")
        e("s_mov_b32 s82, s44")
        e("s_mov_b32 s83, 0x00020000")
        e("s_mov_b32 s84, 0")
        e("s_mov_b32 s85, s43")
        e("s_mov_b32 s86, 0")
        e("v_mov_b32 v143, v144")
    advance_cursor(pf=True)
    advance_cursor(pf=True)
    for k in range(2, 1 + PF):
        prefetch()
        advance_cursor(pf=True)
    for r in range(256):
        e(f"v_accvgpr_write_b32 a{r}, 0")
    # ---- tiles 0 and 1 under way
    for g in range(2):
        begin_issue("s68" if g == 0 else "s74")
        for j in range(16):
            e("s_nop 0")
            dma_piece(j)
        if g == 1:
            prefetch()        # tile 1 + PF; the youngest load, as in the loop
            advance_cursor(pf=True)
        advance_cursor()
    if "stamps" in ABLATE:   # lane 1: the first two tiles' pieces are issued; lane 2: tile 0 has landed (behind the barrier)
        stamp_issue()
        e("s_waitcnt lgkmcnt(0)")
        stamp_keep()
    e("s_waitcnt vmcnt(16)" if "nopf" in ABLATE else "s_waitcnt vmcnt(17)")
    e("s_barrier")
    if "stamps" in ABLATE:
        stamp_issue()
        e("s_waitcnt lgkmcnt(0)")
        stamp_keep()
    e("v_mov_b32 v128, v130")
    e("v_mov_b32 v129, v131")
    for r in range(16):
        read_frag(r, 0)
    e("s_waitcnt lgkmcnt(0)")
    # ---- main loop: tiles 0 .. tiles - 3
    e("s_sub_u32 s70, s55, 2")
    e("s_cmp_eq_u32 s70, 0")
    e("s_cbranch_scc1 .Lw4_tail_%=")
    e(".Lw4_loop_%=:")
    tile(True)
    e("s_sub_u32 s70, s70, 1")
    e("s_cmp_eq_u32 s70, 0")
    e("s_cbranch_scc0 .Lw4_loop_%=")
    e(".Lw4_tail_%=:")
    # ---- last two tiles: nothing left to issue
    tile(False)
    tile(False, last=True)
    if "stamps" in ABLATE:   # last lane: the loop's end
        stamp_issue()
        e("s_waitcnt lgkmcnt(0)")
        stamp_keep()
    # the last prefetch load may still be in flight: its result register is the compiler's again after this statement
    e("s_waitcnt vmcnt(0)")
    # the compiler does not know that MFMAs are still writing the accumulators it reads next
    e("s_nop 15")
    e("s_nop 15")
    e("s_mov_b32 m0, s73")


def emit_constraints(f):
    """the operand lists that go with the loop: ACC is an array f32x4[8][8]; inputs are named by the caller"""
    outs = ", ".join(f'"={{a[{4 * k}:{4 * k + 3}]}}"(ACC[{k >> 3}][{k & 7}])' for k in range(64))
    f.write(f"#define ADCDIFF_W4_OUTS(ACC) {outs}\n")
    vregs = list(range(130)) + [134, 135, 142, 143]
    sregs = list(range(57, 87))
    clob = ", ".join(['"memory"', '"vcc"', '"scc"'] + [f'"v{r}"' for r in vregs] + [f'"s{r}"' for r in sregs])
    f.write(f"#define ADCDIFF_W4_CLOBBERS {clob}\n")


def emit(f, name):
    f.write(f"#define {name} \\\n")
    for k, line in enumerate(L):
        f.write(f'    "{line}\\n\\t"' + (" \\\n" if k + 1 < len(L) else "\n"))


def main():
    # gen_w4_mainloop.py [out]                      the product's loop
    # gen_w4_mainloop.py --lab out name=flag+flag.. ablated variants for the main-loop laboratory
    if len(sys.argv) > 1 and sys.argv[1] == "--lab":
        with open(sys.argv[2], "w") as f:
            f.write("// Generated by gen_w4_mainloop.py --lab -- do not edit.\n")
            emit_constraints(f)
            for spec in sys.argv[3:]:
                name, _, flags = spec.partition("=")
                ABLATE.clear()
                ABLATE.update(x for x in flags.split("+") if x)
                del L[:]
                uid[0] = 0
                generate()
                emit(f, "W4LAB_" + name)
        return
    if len(sys.argv) > 2 and sys.argv[1] == "--stamps":
        # the measurement build's loop (-DADCDIFF_STAMPS): the product's loop plus a time stamp per tile in v147 (s88..s90, v148, v149 scratch);
        # the operand lists are those of w4_mainloop.inc
        ABLATE.add("stamps")
        generate()
        with open(sys.argv[2], "w") as f:
            f.write("// Generated by gen_w4_mainloop.py --stamps -- do not edit.\n")
            emit(f, "ADCDIFF_W4S_MAINLOOP_ASM")
        return
    import os
    ABLATE.update(x for x in os.environ.get("W4_ABLATE", "").split("+") if x)   # experiments: W4_ABLATE=dmafirsthalf+readlate
    generate()
    out = sys.argv[1] if len(sys.argv) > 1 else "w4_mainloop.inc"
    with open(out, "w") as f:
        f.write("// Generated by gen_w4_mainloop.py -- do not edit.  Register map and protocol: see the generator.\n")
        emit_constraints(f)
        emit(f, "ADCDIFF_W4_MAINLOOP_ASM")


if __name__ == "__main__":
    main()
