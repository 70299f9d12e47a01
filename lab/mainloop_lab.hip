// Main-loop laboratory for the 256 x 256-tile bf16 GEMM (gfx950): stand-alone program, no torch.
//   C[M, N] (bf16) = A[M, K] * W[N, K]^T, M = 65536 by default.
// Structure under test: K is consumed in steps of 32 (one v_mfma_f32_16x16x32_bf16 per 16 x 16 output tile and step).
// A step's operands (256 rows of A and of W, 64 bytes each) form one 32 KiB LDS region; four regions form a ring.
// During step R a wave issues the MFMAs of step R from registers, reads the fragments of step R + 1 into a second
// register set, and issues its LDS-DMA pieces of region R + 4; one barrier per step.
// Template parameters choose 4 waves (2 x 2, 128 x 128 per wave, one wave per SIMD, accumulators in 256 registers) or
// 8 waves (2 x 4, 128 x 64 per wave, two waves per SIMD), and ablations (no reads / no DMA / no barrier) that show
// what each part of a step costs.  Results are checked against a plain kernel on sampled rows.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void glds16_untracked(const void* gsrc, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_addr)
                 : "memory");
}

typedef int i32x4 __attribute__((ext_vector_type(4)));
// The same 1 KiB copy through the buffer path: wave-uniform descriptor + scalar byte offset + per-lane 32-bit offset.
__device__ __forceinline__ void blds16_untracked(i32x4 srd, unsigned voff, unsigned soff, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
                 :
                 : "v"(voff), "s"(srd), "s"(soff), "s"(lds_addr)
                 : "memory");
}

// M0 written once per step; the pieces address their 1 KiB slots through the instruction's offset field, which shifts the global
// address too, so the caller's source pointer is pre-shifted by -OFF
template <int OFF>
__device__ __forceinline__ void glds16_m0set(const void* gsrc) {
    asm volatile("global_load_lds_dwordx4 %0, off offset:%1" : : "v"(gsrc), "n"(OFF) : "memory");
}
__device__ __forceinline__ void set_m0(unsigned v) { asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" : : "s"(v) : "memory"); }

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

enum { ABL_NOREAD = 1, ABL_NODMA = 2, ABL_NOBAR = 4, ABL_NOWAIT = 8, ABL_SAMEK = 16, ABL_SAMETILE = 32, ABL_STAGGER = 64, ABL_STAGGER2 = 128, ABL_TAIL = 256, ABL_ASYM = 512, ABL_BUF = 1024, ABL_REG = 2048, ABL_M0ONCE = 4096 };

struct Params {
    const bf16_t* A; const bf16_t* W; bf16_t* C;
    int M, N, K, ntn;
    long long* stamps;
};

// chunk swizzle of the 64-byte-row image: physical chunk = logical chunk ^ g((row >> 2) & 3), g = {0, 2, 3, 1}: the four
// 16-lane groups of a ds_read_b128 (rows l & 15, chunk l >> 4) then hit sixteen distinct 16-byte slots of a 256-byte bank row
__device__ __forceinline__ int swz(int row) { return (0x78 >> (2 * ((row >> 2) & 3))) & 3; }   // 0b01111000 -> {0,2,3,1}

template <int WAVES_N, int ABL, int WAVES_M = 2>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, WAVES_M * WAVES_N / 4) void ring_kernel(const Params p) {
    constexpr int NW = WAVES_M * WAVES_N, WTM = 256 / WAVES_M, WTN = 256 / WAVES_N, TM = WTM / 16, TN = WTN / 16;
    constexpr bool ASYM = (ABL & ABL_ASYM) != 0;   // only the first four waves issue LDS-DMA (after their MFMAs), 8 pieces each
    constexpr int PPW = ASYM ? 8 : 32 / NW;        // LDS-DMA pieces per (loading) wave and region
    constexpr int REGION = 32768;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    int bid = blockIdx.x;
    const int nwg = gridDim.x;
    if ((nwg & 7) == 0) bid = (bid & 7) * (nwg >> 3) + (bid >> 3);
    const int m0 = (bid / p.ntn) * 256, n0 = (bid % p.ntn) * 256;
    const int lm0 = (ABL & ABL_SAMETILE) ? 0 : m0, ln0 = (ABL & ABL_SAMETILE) ? 0 : n0;   // load-side tile origin
    const long long t0 = __builtin_amdgcn_s_memtime();

    // DMA geometry: piece q = wave * PPW + j; q < 16: A rows 16 q .. 16 q + 15, else W rows 16 (q - 16) ..
    const int prow = lane >> 2;
    const char* src[PPW];
    unsigned dst[PPW];
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int q = (ASYM ? (wave & 3) : wave) * PPW + j;
        const bool isA = q < 16;
        const int row = 16 * (q & 15) + prow;
        const int chunk = (lane & 3) ^ swz(row);
        const char* base = isA ? reinterpret_cast<const char*>(p.A) + (long)(lm0 + row) * p.K * 2
                               : reinterpret_cast<const char*>(p.W) + (long)(ln0 + row) * p.K * 2;
        src[j] = base + chunk * 16;
        dst[j] = (isA ? 0 : 16384) + (q & 15) * 1024;
    }
    const int nR = p.K / 32;
    const int wslot = wave % (TM * TN / PPW);                 // every wave its own issue slot inside an 8-MFMA window
    const int wslot2 = ((wave & 3) * 2 + (wave >> 2)) % (TM * TN / PPW);   // SIMD partners (w, w + 4) in adjacent slots
    const bool loader = !ASYM || wave < 4;
    auto issue = [&](int R) {
        if (loader && R < nR) {
            const unsigned rb = lds0 + (R & 3) * REGION;
#pragma unroll
            for (int j = 0; j < PPW; ++j) glds16_untracked(src[j] + (long)R * 64, rb + dst[j]);
        }
    };
    // buffer path: descriptor over this wave's operand (A or W rows of the tile), lane offset without the piece / region part
    const bool wave_is_a = (ASYM ? (wave & 3) : wave) * PPW < 16;
    const char* bbase = wave_is_a ? reinterpret_cast<const char*>(p.A) + (long)lm0 * p.K * 2 : reinterpret_cast<const char*>(p.W) + (long)ln0 * p.K * 2;
    const unsigned long long bb = (unsigned long long)bbase;
    const i32x4 srd = {__builtin_amdgcn_readfirstlane((int)(unsigned)bb), __builtin_amdgcn_readfirstlane((int)((unsigned)(bb >> 32) & 0xffffu)),
                       __builtin_amdgcn_readfirstlane((int)(256u * (unsigned)p.K * 2u)), 0x00020000};   // base, stride 0, bytes, raw dword format
    const unsigned bvoff = (unsigned)prow * (unsigned)(p.K * 2) + (unsigned)(((lane & 3) ^ swz(prow)) * 16);
    auto issue1 = [&](int R, int j) {
        if (ABL & ABL_BUF) {
            if (loader && R < nR) {
                const int q = (ASYM ? (wave & 3) : wave) * PPW + j;
                blds16_untracked(srd, bvoff, (unsigned)(16 * (q & 15)) * (unsigned)(p.K * 2) + (unsigned)R * 64u, lds0 + (R & 3) * REGION + dst[j]);
            }
            return;
        }
        if (R < nR) glds16_untracked(src[j] + ((ABL & ABL_SAMEK) ? 0L : (long)R * 64), lds0 + (R & 3) * REGION + dst[j]);
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fchunk = lane >> 4;
    const int a_off = (wm * WTM + frow) * 64 + ((fchunk ^ swz(frow)) << 4);
    const int b_off = 16384 + (wn * WTN + frow) * 64 + ((fchunk ^ swz(frow)) << 4);
    uint4 fa[2][TM], fb[2][TN];

#define READ_A(S, RB, I) fa[S][I] = *reinterpret_cast<const uint4*>((RB) + a_off + (I) * 1024)
#define READ_B(S, RB, J) fb[S][J] = *reinterpret_cast<const uint4*>((RB) + b_off + (J) * 1024)
#define MMA(S, I, J)                                                                                               \
    acc[I][J] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[S][I]), __builtin_bit_cast(bf16x8, fb[S][J]), \
                                                         acc[I][J], 0, 0, 0)

    // prologue
    uint4 stg[2][PPW];   // ABL_REG: register staging, two regions in flight
    issue(0); issue(1);
    if (!(ABL & ABL_REG)) { issue(2); issue(3); }
    if (ABL & ABL_REG) wait_vmcnt<0>();
    else {
        const int inflight = nR - 1 < 3 ? nR - 1 : 3;   // regions allowed to stay in flight so that region 0 has landed
        if (inflight >= 3) wait_vmcnt<3 * PPW>(); else if (inflight == 2) wait_vmcnt<2 * PPW>(); else if (inflight == 1) wait_vmcnt<PPW>(); else wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < TM; ++i) READ_A(0, smem, i);
#pragma unroll
    for (int j = 0; j < TN; ++j) READ_B(0, smem, j);
    if (ABL & ABL_NOREAD) {
#pragma unroll
        for (int i = 0; i < TM; ++i) READ_A(1, smem, i);
#pragma unroll
        for (int j = 0; j < TN; ++j) READ_B(1, smem, j);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (ABL & ABL_REG) {
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int j = 0; j < PPW; ++j)
                if (2 + g < nR) stg[g][j] = *reinterpret_cast<const uint4*>(src[j] + (long)(2 + g) * 64);
    } else {
        const int inflight = nR - 2 < 2 ? nR - 2 : 2;
        if (inflight >= 2) wait_vmcnt<2 * PPW>(); else if (inflight == 1) wait_vmcnt<PPW>(); else wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    const long long t1 = __builtin_amdgcn_s_memtime();

    // one step: S = register set holding step R's fragments; the other set receives step R + 1's
#define STEP(R_, S)                                                                                                \
    do {                                                                                                           \
        const int R = (R_);   /* R_ must not be spelled R at the call site */                                     \
        const char* rb = smem + ((R + 1) & 3) * REGION;                                                            \
        const bool more = R + 1 < nR;                                                                              \
        constexpr int NMMA = TM * TN, NRD = TM + TN;                                                               \
        constexpr int RD_EVERY = NMMA / NRD >= 4 ? 4 : 2;       /* a read behind every RD_EVERY-th MFMA */          \
        constexpr int DMA_EVERY = NMMA / PPW;                                                                      \
        _Pragma("unroll") for (int x = 0; x < NMMA; ++x) {                                                         \
            const int i = x / TN, j = (x / TN) & 1 ? TN - 1 - x % TN : x % TN;                                     \
            MMA(S, i, j);                                                                                          \
            if (!(ABL & ABL_NOREAD) && more && x % RD_EVERY == 1 && x / RD_EVERY < NRD) {                          \
                const int r = x / RD_EVERY;                                                                        \
                if (r < TN) READ_B((S) ^ 1, rb, r); else READ_A((S) ^ 1, rb, r - TN);                              \
            }                                                                                                      \
            if (ABL & ABL_M0ONCE) {                                                                               \
                if (x == 0 && R + 4 < nR) set_m0(lds0 + ((R + 4) & 3) * REGION + dst[0]);                          \
                if (R + 4 < nR) {                                                                                  \
                    if (x == 4) glds16_m0set<0>(src[0] + (long)(R + 4) * 64);                                      \
                    if (x == 12) glds16_m0set<1024>(src[1] + (long)(R + 4) * 64 - 1024);                           \
                    if (x == 20) glds16_m0set<2048>(src[2] + (long)(R + 4) * 64 - 2048);                           \
                    if (x == 28) glds16_m0set<3072>(src[3] + (long)(R + 4) * 64 - 3072);                           \
                }                                                                                                  \
            } else                                                                                                 \
            if (ABL & ABL_REG) {                                                                                  \
                if (x % DMA_EVERY == 1 && R + 2 < nR)                                                              \
                    *reinterpret_cast<uint4*>(smem + ((R + 2) & 3) * REGION + dst[x / DMA_EVERY] + lane * 16) = stg[(S)][x / DMA_EVERY]; \
                if (x % DMA_EVERY == 5 && R + 4 < nR)                                                              \
                    stg[(S)][x / DMA_EVERY] = *reinterpret_cast<const uint4*>(src[x / DMA_EVERY] + ((ABL & ABL_SAMEK) ? 0L : (long)(R + 4) * 64)); \
            } else                                                                                                 \
            if (ABL & ABL_STAGGER) { if (x % DMA_EVERY == wslot) issue1(R + 4, x / DMA_EVERY); }                   \
            else if (ABL & ABL_STAGGER2) { if (x % DMA_EVERY == wslot2) issue1(R + 4, x / DMA_EVERY); }             \
            else if (!(ABL & (ABL_NODMA | ABL_TAIL | ABL_ASYM)) && x % DMA_EVERY == DMA_EVERY / 2) issue1(R + 4, x / DMA_EVERY); \
            if (x % 2 == 1) __builtin_amdgcn_sched_barrier(0);                                                     \
        }                                                                                                          \
        if ((ABL & (ABL_TAIL | ABL_ASYM)) && loader) { _Pragma("unroll") for (int j = 0; j < PPW; ++j) issue1(R + 4, j); } \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                         \
        if (loader && !(ABL & (ABL_NODMA | ABL_NOWAIT | ABL_REG))) {                                                                   \
            if (R + 4 < nR) wait_vmcnt<2 * PPW>(); else if (R + 3 < nR) wait_vmcnt<PPW>(); else wait_vmcnt<0>();    \
        }                                                                                                          \
        if (!(ABL & ABL_NOBAR)) __builtin_amdgcn_s_barrier();                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
    } while (0)

    for (int rr = 0; rr < nR; rr += 2) {
        STEP(rr, 0);
        STEP(rr + 1, 1);
    }
    const long long t2 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // plain epilogue: bf16 stores straight from the accumulator layout (col = l & 15, row = 4 * (l >> 4) + r)
    bf16_t* C = p.C;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long row = m0 + wm * WTM + i * 16 + 4 * (lane >> 4) + r;
                const int col = n0 + wn * WTN + j * 16 + (lane & 15);
                C[row * p.N + col] = (bf16_t)acc[i][j][r];
            }
    if (lane == 0 && p.stamps) {
        long long* o = p.stamps + ((long)blockIdx.x * NW + wave) * 4;
        o[0] = t1 - t0; o[1] = t2 - t1; o[2] = __builtin_amdgcn_s_memtime() - t2; o[3] = 0;
    }
#undef STEP
#undef MMA
#undef READ_B
#undef READ_A
}

// ---- 4 waves (2 x 2), 128 x 128 per wave, ONE wave per SIMD, accumulators in AGPRs, operands staged through registers ----
// Two 64 KiB LDS stages (a whole 64-wide K tile of A and W each, 128-byte rows, chunk ^= (row >> 1) & 7).  During K tile t a
// wave (1) stores the registers that hold its sixteen 1 KiB pieces of tile t + 1 into the other stage (ds_write_b128, loaded
// during tile t - 1), barrier in the MIDDLE of the tile, (2) reloads those registers with tile t + 2 (global_load_dwordx4);
// fragment reads of the next 32-wide k step go into a second register set, one read / store / load behind every fourth MFMA.
// NOT MEASURED: hipcc (ROCm 7.2) does not allocate this kernel.  With the MFMA builtin it spills 190-430 VGPRs (even the variant
// with no staging and no reads inside the loop, 256 accumulator + 128 operand registers, spills 307) and routes MFMA operands
// through AGPRs; with the MFMA as inline asm and "+a" accumulators it copies every accumulator around each asm statement (1247
// v_accvgpr moves, 474 spills).  A 512-register main loop needs hand-allocated assembly.  Kept for the record; not in the table.
enum { W4_NOSTAGE = 1, W4_NOREAD = 2, W4_NOBAR = 4 };
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int ABL>
__global__ __launch_bounds__(256, 1) void w4_kernel(const Params p) {
    constexpr int STAGE = 65536, A_BYTES = 32768;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    int bid = blockIdx.x;
    const int nwg = gridDim.x;
    if ((nwg & 7) == 0) bid = (bid & 7) * (nwg >> 3) + (bid >> 3);
    const int m0 = (bid / p.ntn) * 256, n0 = (bid % p.ntn) * 256;
    const long long t0 = __builtin_amdgcn_s_memtime();

    // staging: piece q = wave * 8 + j of each operand = rows 8 q .. 8 q + 7; lane -> row 8 q + (lane >> 3), chunk lane & 7
    const int r8 = lane >> 3, ch = lane & 7;
    const unsigned rowbytes = (unsigned)p.K * 2u;
    // buffer loads: descriptor over the tile's 256 rows, ONE 32-bit lane offset; piece and K tile go into the scalar offset
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.A) + (long)m0 * p.K, 0, (int)(256u * rowbytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.W) + (long)n0 * p.K, 0, (int)(256u * rowbytes), 0x00020000);
    const unsigned voff = (unsigned)(wave * 64 + r8) * rowbytes + (unsigned)ch * 16u;
    const unsigned piece = 8u * rowbytes;
    // LDS store offsets: row = 64 wave + 8 j + r8; (row >> 1) & 7 = (4 j + (r8 >> 1)) & 7 depends on the parity of j only
    const int st_even = (wave * 64 + r8) * 128 + ((ch ^ ((r8 >> 1) & 7)) << 4);
    const int st_odd = (wave * 64 + r8) * 128 + ((ch ^ ((4 + (r8 >> 1)) & 7)) << 4);
    u32x4 stg[16];
    const int nt = p.K / 64;
    auto gload1 = [&](int t, int j) {
        stg[j] = __builtin_amdgcn_raw_buffer_load_b128(j < 8 ? rA : rW, (int)voff, (int)((unsigned)(j & 7) * piece + (unsigned)t * 128u), 0);
    };
    auto gload = [&](int t) {
#pragma unroll
        for (int j = 0; j < 16; ++j) gload1(t, j);
    };
    auto lstore1 = [&](char* stage, int j) {
        char* dst = stage + (j < 8 ? 0 : A_BYTES) + ((j & 1) ? st_odd : st_even) + (j & 7) * 1024;
        *reinterpret_cast<u32x4*>(dst) = stg[j];
    };

    f32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fchunk = lane >> 4;
    int a_off[2], b_off[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int c = 4 * ks + fchunk;
        a_off[ks] = (wm * 128 + frow) * 128 + ((c ^ ((frow >> 1) & 7)) << 4);
        b_off[ks] = A_BYTES + (wn * 128 + frow) * 128 + ((c ^ ((frow >> 1) & 7)) << 4);
    }
    i32x4 fa[2][8], fb[2][8];
#define W4_READ(S, STG, KS, R)                                                                                        \
    do {                                                                                                              \
        if ((R) < 8) fb[S][(R)] = *reinterpret_cast<const i32x4*>((STG) + b_off[KS] + (R) * 2048);                    \
        else fa[S][(R) - 8] = *reinterpret_cast<const i32x4*>((STG) + a_off[KS] + ((R) - 8) * 2048);                  \
    } while (0)
#define W4_MMA(S, I, J)                                                                                               \
    acc[I][J] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[S][I]), __builtin_bit_cast(bf16x8, fb[S][J]), \
                                                         acc[I][J], 0, 0, 0)

    // prologue: tile 0 -> stage 0, tile 1 -> registers
    gload(0);
#pragma unroll
    for (int j = 0; j < 16; ++j) lstore1(smem, j);
    if (nt > 1) gload(1);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) W4_READ(0, smem, 0, r);
    if (ABL & W4_NOREAD) {
#pragma unroll
        for (int r = 0; r < 16; ++r) W4_READ(1, smem, 1, r);
    }
    __builtin_amdgcn_sched_barrier(0);
    const long long t1 = __builtin_amdgcn_s_memtime();

    // one K tile; CUR / NXT = the stage that holds tile t / receives tile t + 1
#define W4_TILE(T_, CUR, NXT)                                                                                         \
    do {                                                                                                              \
        const int tt = (T_);                                                                                          \
        const bool more1 = tt + 1 < nt, more2 = tt + 2 < nt;                                                          \
        _Pragma("unroll") for (int x = 0; x < 64; ++x) {                                                              \
            const int i = x >> 3, j = (i & 1) ? 7 - (x & 7) : (x & 7);                                                \
            W4_MMA(0, i, j);                                                                                          \
            if (x % 4 == 1 && !(ABL & W4_NOREAD)) W4_READ(1, CUR, 1, x / 4);                                          \
            if (x % 4 == 3 && more1 && !(ABL & W4_NOSTAGE)) lstore1(NXT, x / 4);                                      \
            if (x % 2 == 1) __builtin_amdgcn_sched_barrier(0);                                                        \
        }                                                                                                             \
        if (!(ABL & W4_NOBAR)) __syncthreads();                                                                       \
        _Pragma("unroll") for (int x = 0; x < 64; ++x) {                                                              \
            const int i = x >> 3, j = (i & 1) ? 7 - (x & 7) : (x & 7);                                                \
            W4_MMA(1, i, j);                                                                                          \
            if (x % 4 == 1 && more1 && !(ABL & W4_NOREAD)) W4_READ(0, NXT, 0, x / 4);                                 \
            if (x % 4 == 3 && more2 && !(ABL & W4_NOSTAGE)) gload1(tt + 2, x / 4);                                    \
            if (x % 2 == 1) __builtin_amdgcn_sched_barrier(0);                                                        \
        }                                                                                                             \
    } while (0)

    char* const s0 = smem;
    char* const s1 = smem + STAGE;
    for (int t = 0; t < nt; t += 2) {
        W4_TILE(t, s0, s1);
        if (t + 1 < nt) W4_TILE(t + 1, s1, s0);
    }
    const long long t2 = __builtin_amdgcn_s_memtime();

    bf16_t* C = p.C;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long row = m0 + wm * 128 + i * 16 + 4 * (lane >> 4) + r;
                const int col = n0 + wn * 128 + j * 16 + (lane & 15);
                C[row * p.N + col] = (bf16_t)acc[i][j][r];
            }
    if (lane == 0 && p.stamps) {
        long long* o = p.stamps + ((long)blockIdx.x * 4 + wave) * 4;
        o[0] = t1 - t0; o[1] = t2 - t1; o[2] = __builtin_amdgcn_s_memtime() - t2; o[3] = 0;
    }
#undef W4_TILE
#undef W4_MMA
#undef W4_READ
}

template <int ABL>
[[maybe_unused]] static void launch_w4(const Params& p, hipStream_t st) {
    constexpr int smem = 2 * 65536;
    static bool set = false;
    if (!set) { CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&w4_kernel<ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, smem)); set = true; }
    const int grid = (p.M / 256) * (p.N / 256);
    hipLaunchKernelGGL((w4_kernel<ABL>), dim3(grid), dim3(256), smem, st, p);
}


// ---- 4 waves (2 x 2), 128 x 128 per wave, ONE wave per SIMD: the hand-allocated main loop of the product
// (csrc/gen_w4_mainloop.py: register map and protocol there), with the generator's ablations (V) and source address
// patterns (PAT: 0 = the real one, 8 rows x 128 B per piece; 1 = every lane the same 16 bytes; 2 = 1 KiB contiguous) ----
#include "w4_variants.inc"
#define W4_OUTS ADCDIFF_W4_OUTS(acc)
#define W4_INS "{s40}"(s_base_lo), "{s41}"(s_base_hi), "{s42}"(s_ld), "{s43}"(s_steps), "{s44}"(s_nrec), \
               "{s45}"(s_zero), "{s46}"(s_zero), "{s47}"(s_zero), "{s48}"(s_zero), "{s49}"(s_zero), \
               "{s50}"(s_zero), "{s51}"(s_zero), "{s52}"(s_zero), "{s53}"(s_zero), "{s54}"(s_zero), \
               "{s55}"(s_steps), "{s56}"(s_dst), \
               "{v130}"(v_aoff0), "{v131}"(v_boff0), "{v132}"(v_aoff1), "{v133}"(v_boff1), \
               "{v136}"(v_src0), "{v137}"(v_src1), "{v138}"(v_src0), "{v139}"(v_src1), "{v140}"(v_src0), "{v141}"(v_src1), \
               "{v144}"(v_pf), "{v145}"(v_pf), "{v146}"(v_pf)
#define W4_CLOBBERS ADCDIFF_W4_CLOBBERS
template <int V, int PAT>
__global__ __launch_bounds__(256, 1) void w4asm_kernel(const Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    int bid = blockIdx.x;
    const int nwg = gridDim.x;
    if ((nwg & 7) == 0) bid = (bid & 7) * (nwg >> 3) + (bid >> 3);
    const int m0 = (bid / p.ntn) * 256, n0 = (bid % p.ntn) * 256;
    const long long t0 = __builtin_amdgcn_s_memtime();

    const bool loads_a = wave < 2;
    const unsigned ldb = (unsigned)p.K * 2u;
    // PAT 3: the A loaders read W rows too (every load an L2 hit); PAT 4: source chunks not permuted
    const unsigned long long base = loads_a && PAT != 3 ? (unsigned long long)(p.A + (long)(m0 + (wave & 1) * 128) * p.K)
                                                        : (unsigned long long)(p.W + (long)(n0 + (wave & 1) * 128) * p.K);
    const int prow = lane >> 3, frow = lane & 15, fchunk = lane >> 4, fx = (frow >> 1) & 7;
    const int s_base_lo = __builtin_amdgcn_readfirstlane((int)(unsigned)base);
    const int s_base_hi = __builtin_amdgcn_readfirstlane((int)((unsigned)(base >> 32) & 0xffffu));
    const int s_ld = __builtin_amdgcn_readfirstlane((int)ldb);
    const int s_steps = __builtin_amdgcn_readfirstlane(p.K / 64);
    const int s_nrec = __builtin_amdgcn_readfirstlane((int)(127u * ldb + (unsigned)p.K * 2u));
    const int s_zero = __builtin_amdgcn_readfirstlane(0);
    const int s_dst = __builtin_amdgcn_readfirstlane((int)(lds0 + (loads_a ? 0u : 32768u) + (unsigned)(wave & 1) * 16384u));
    const int v_aoff0 = (int)lds0 + (wm * 128 + frow) * 128 + ((fchunk ^ fx) << 4);
    const int v_aoff1 = (int)lds0 + (wm * 128 + frow) * 128 + (((4 + fchunk) ^ fx) << 4);
    const int v_boff0 = (int)lds0 + 32768 + (wn * 128 + frow) * 128 + ((fchunk ^ fx) << 4);
    const int v_boff1 = (int)lds0 + 32768 + (wn * 128 + frow) * 128 + (((4 + fchunk) ^ fx) << 4);
    // piece j of this wave = rows 8 j .. 8 j + 7 of its 128: (row >> 1) & 7 = (4 j + (lane >> 4)) & 7 depends on the parity of j only
    const int v_src0 = PAT == 1 ? 0 : PAT == 2 ? lane * 16 : PAT == 4 ? (int)((unsigned)prow * ldb + (unsigned)((lane & 7) * 16))
                                                                        : (int)((unsigned)prow * ldb + (unsigned)(((lane & 7) ^ (lane >> 4)) * 16));
    const int v_src1 = PAT == 1 ? 0 : PAT == 2 ? lane * 16 : PAT == 4 ? (int)((unsigned)prow * ldb + (unsigned)((lane & 7) * 16))
                                                                        : (int)((unsigned)prow * ldb + (unsigned)(((lane & 7) ^ (4 + (lane >> 4))) * 16));

    // prefetch share: the p.ntn workgroups of a tile row split its 256 A rows; the rows of this workgroup lie in one wave's half
    const int nshare = 256 / p.ntn, nidx = bid % p.ntn, prow0 = nidx * nshare;
    const bool pf_wave = loads_a && (prow0 >> 7) == (wave & 1);
    const int v_pf = pf_wave ? (int)((unsigned)((prow0 & 127) + lane % nshare) * ldb + (unsigned)(lane / nshare) * 128u) : 0;
    f32x4 acc[8][8];
    if constexpr (V == 0) asm volatile(W4LAB_full : W4_OUTS : W4_INS : W4_CLOBBERS);
    else if constexpr (V == 1) asm volatile(W4LAB_nodma : W4_OUTS : W4_INS : W4_CLOBBERS);
    else if constexpr (V == 2) asm volatile(W4LAB_noread : W4_OUTS : W4_INS : W4_CLOBBERS);
    else if constexpr (V == 3) asm volatile(W4LAB_nobar : W4_OUTS : W4_INS : W4_CLOBBERS);
    else if constexpr (V == 4) asm volatile(W4LAB_nowait : W4_OUTS : W4_INS : W4_CLOBBERS);
    else if constexpr (V == 5) asm volatile(W4LAB_mfmaonly : W4_OUTS : W4_INS : W4_CLOBBERS);
    else if constexpr (V == 6) asm volatile(W4LAB_dmafirsthalf : W4_OUTS : W4_INS : W4_CLOBBERS);
    else if constexpr (V == 7) asm volatile(W4LAB_nopf : W4_OUTS : W4_INS : W4_CLOBBERS);
    else if constexpr (V == 8) asm volatile(W4LAB_pf2 : W4_OUTS : W4_INS : W4_CLOBBERS);
    else if constexpr (V == 9) asm volatile(W4LAB_pf3 : W4_OUTS : W4_INS : W4_CLOBBERS);
    else if constexpr (V == 10) asm volatile(W4LAB_pf6 : W4_OUTS : W4_INS : W4_CLOBBERS);
    else if constexpr (V == 11) asm volatile(W4LAB_pf8 : W4_OUTS : W4_INS : W4_CLOBBERS);
    else if constexpr (V == 12) asm volatile(W4LAB_pf12 : W4_OUTS : W4_INS : W4_CLOBBERS);
    const long long t2 = __builtin_amdgcn_s_memtime();

    bf16_t* C = p.C;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long row = m0 + wm * 128 + i * 16 + 4 * (lane >> 4) + r;
                const int col = n0 + wn * 128 + j * 16 + (lane & 15);
                C[row * p.N + col] = (bf16_t)acc[i][j][r];
            }
    if (lane == 0 && p.stamps) {
        long long* o = p.stamps + ((long)blockIdx.x * 4 + wave) * 4;
        o[0] = 0; o[1] = t2 - t0; o[2] = __builtin_amdgcn_s_memtime() - t2; o[3] = 0;
    }
}

template <int V, int PAT>
[[maybe_unused]] static void launch_w4asm(const Params& p, hipStream_t st) {
    constexpr int smem = 4 * 32768;
    static bool set = false;
    if (!set) { CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&w4asm_kernel<V, PAT>), hipFuncAttributeMaxDynamicSharedMemorySize, smem)); set = true; }
    const int grid = (p.M / 256) * (p.N / 256);
    hipLaunchKernelGGL((w4asm_kernel<V, PAT>), dim3(grid), dim3(256), smem, st, p);
}

__global__ void ref_kernel(const bf16_t* A, const bf16_t* W, float* out, const int* rows, int nrows, int N, int K) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nrows * N) return;
    const int r = rows[idx / N], c = idx % N;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += (float)A[(long)r * K + k] * (float)W[(long)c * K + k];
    out[idx] = s;
}

__global__ void fill_kernel(bf16_t* x, long n, uint32_t seed, float scale, int zero) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t h = (uint32_t)i * 2654435761u ^ seed;
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    uint32_t g = h * 1664525u + 1013904223u;
    // sum of two uniforms, centred: roughly bell shaped in (-1, 1), full sign and exponent variety
    float v = ((h >> 8) * (1.0f / 16777216.0f) + (g >> 8) * (1.0f / 16777216.0f) - 1.0f) * scale;
    x[i] = zero ? (bf16_t)0.0f : (bf16_t)v;
}

template <int WAVES_N, int ABL, int WAVES_M = 2>
static void launch(const Params& p, hipStream_t st) {
    constexpr int smem = 4 * 32768;
    static bool set = false;
    if (!set) { CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ring_kernel<WAVES_N, ABL, WAVES_M>), hipFuncAttributeMaxDynamicSharedMemorySize, smem)); set = true; }
    const int grid = (p.M / 256) * (p.N / 256);
    hipLaunchKernelGGL((ring_kernel<WAVES_N, ABL, WAVES_M>), dim3(grid), dim3(64 * WAVES_M * WAVES_N), smem, st, p);
}

typedef void (*launch_fn)(const Params&, hipStream_t);
struct Variant { const char* name; launch_fn fn; int nw; bool exact; };

int main(int argc, char** argv) {
    int M = 65536, N = 2048, K = 2048, iters = 20, zero = 0;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "-M")) M = atoi(argv[++i]);
        else if (!strcmp(argv[i], "-N")) N = atoi(argv[++i]);
        else if (!strcmp(argv[i], "-K")) K = atoi(argv[++i]);
        else if (!strcmp(argv[i], "-i")) iters = atoi(argv[++i]);
        else if (!strcmp(argv[i], "-z")) zero = 1;
    }
    if (M % 256 || N % 256 || K % 64) { fprintf(stderr, "M, N multiples of 256, K of 64\n"); return 1; }
    bf16_t *A, *W, *C;
    CHECK(hipMalloc(&A, (size_t)M * K * 2)); CHECK(hipMalloc(&W, (size_t)N * K * 2)); CHECK(hipMalloc(&C, (size_t)M * N * 2));
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)(((long)M * K + 255) / 256)), dim3(256), 0, 0, A, (long)M * K, 1u, 1.0f, zero);
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)(((long)N * K + 255) / 256)), dim3(256), 0, 0, W, (long)N * K, 2u, 0.05f, zero);
    const int grid = (M / 256) * (N / 256);
    long long* stamps;
    CHECK(hipMalloc(&stamps, (size_t)grid * 16 * 4 * 8));
    // reference on sampled rows
    const int nrows = 64;
    std::vector<int> hrows(nrows);
    for (int i = 0; i < nrows; ++i) hrows[i] = (int)(((long)i * 2654435761u) % (unsigned)M);
    hrows[0] = 0; hrows[1] = M - 1; hrows[2] = 255; hrows[3] = 256;
    int* drows; float* dref;
    CHECK(hipMalloc(&drows, nrows * 4)); CHECK(hipMalloc(&dref, (size_t)nrows * N * 4));
    CHECK(hipMemcpy(drows, hrows.data(), nrows * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(ref_kernel, dim3((nrows * N + 255) / 256), dim3(256), 0, 0, A, W, dref, drows, nrows, N, K);
    std::vector<float> href((size_t)nrows * N);
    CHECK(hipMemcpy(href.data(), dref, href.size() * 4, hipMemcpyDeviceToHost));

    Variant vs[] = {
        {"ring8 (8 waves 128x64)", launch<4, 0>, 8, true},
        {"w4asm (4 waves 128x128, asm loop)", launch_w4asm<0, 0>, 4, true},
        {"w4asm no prefetch", launch_w4asm<7, 0>, 4, true},
        {"w4asm prefetch distance 2", launch_w4asm<8, 0>, 4, true},
        {"w4asm prefetch distance 3", launch_w4asm<9, 0>, 4, true},
        {"w4asm prefetch distance 6", launch_w4asm<10, 0>, 4, true},
        {"w4asm prefetch distance 8", launch_w4asm<11, 0>, 4, true},
        {"w4asm prefetch distance 12", launch_w4asm<12, 0>, 4, true},
        {"w4asm (again)", launch_w4asm<0, 0>, 4, true},
        {"w4asm no DMA", launch_w4asm<1, 0>, 4, false},
        {"w4asm no fragment reads", launch_w4asm<2, 0>, 4, false},
        {"w4asm no barrier", launch_w4asm<3, 0>, 4, false},
        {"w4asm no vmcnt wait", launch_w4asm<4, 0>, 4, false},
        {"w4asm MFMAs only", launch_w4asm<5, 0>, 4, false},
        {"w4asm DMA in the first half of the phase", launch_w4asm<6, 0>, 4, true},
        {"w4asm DMA: all lanes one address", launch_w4asm<0, 1>, 4, false},
        {"w4asm DMA: 1 KiB contiguous", launch_w4asm<0, 2>, 4, false},
        {"w4asm DMA: every wave loads W rows", launch_w4asm<0, 3>, 4, false},
        {"w4asm DMA: chunks not permuted", launch_w4asm<0, 4>, 4, false},
        {"w4asm no vmcnt wait, W rows only", launch_w4asm<4, 3>, 4, false},
    };
    Params p{A, W, C, M, N, K, N / 256, stamps};
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const double fl = 2.0 * M * N * K;
    std::vector<bf16_t> hc((size_t)N);
    for (const Variant& v : vs) {
        CHECK(hipMemset(C, 0, (size_t)M * N * 2));
        for (int i = 0; i < 3; ++i) v.fn(p, 0);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0, 0));
        for (int i = 0; i < iters; ++i) v.fn(p, 0);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / iters;
        std::vector<long long> hs((size_t)grid * v.nw * 4);
        CHECK(hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost));
        double pro = 0, mainl = 0, epi = 0;
        for (size_t i = 0; i < hs.size(); i += 4) { pro += hs[i]; mainl += hs[i + 1]; epi += hs[i + 2]; }
        const double n = (double)grid * v.nw;
        double maxerr = 0;
        if (v.exact) {
            for (int i = 0; i < nrows; ++i) {
                CHECK(hipMemcpy(hc.data(), C + (size_t)hrows[i] * N, (size_t)N * 2, hipMemcpyDeviceToHost));
                for (int c = 0; c < N; ++c) {
                    const double d = fabs((double)(float)hc[c] - href[(size_t)i * N + c]);
                    const double tol = 0.01 * fabs(href[(size_t)i * N + c]) + 0.02;
                    if (!(d / tol <= maxerr)) maxerr = d == d ? d / tol : 1e30;
                }
            }
        }
        printf("%-36s %8.1f us %7.1f TF | cycles/wave: prologue %6.0f main %7.0f (%5.0f per 64-K) epilogue %6.0f | err/tol %.2f %s\n", v.name, us,
               fl / us / 1e6, pro / n, mainl / n, mainl / n / (K / 64), epi / n, maxerr, v.exact ? (maxerr <= 1.0 ? "OK" : "MISMATCH") : "");
        fflush(stdout);
    }
    return 0;
}
