// NT GEMM with split-K operands and fused epilogue for the denoiser's nn.Linear layers.
//
//   out[M, N] = act( [A0 | A1 | A2][M, K] * W[N, K]^T + bias + rowbias + residual )
//
// One workgroup (4 waves, 2 x 2) computes a BM x BN tile with v_mfma_f32_32x32x16_bf16 (bf16
// operands) or v_mfma_f32_32x32x2_f32 (float32 operands, exact f32 fma chain), float32 accumulate.
// Both operand tiles are staged in LDS as rows of 128 bytes (64 bf16 / 32 f32 of K), eight 16-byte
// chunks per row, chunk index XOR-swizzled with (row >> 1) & 7 so that the 16-lane groups of a
// ds_read_b128 hit sixteen different 16-byte slots of the 256-byte bank row.  Tiles go global -> LDS
// directly (global_load_lds_dwordx4, no VGPR round trip, no ds_write); one LDS stage of 32 KiB per
// workgroup, two barriers per K step, latency hidden by 3-4 resident workgroups per CU.
// Workgroup ids are remapped so that the N-tiles of one M-tile run on one XCD (they share the A rows
// in that XCD's L2).
#include "common.h"

namespace adcdiff {

struct GemmParams {
    const void* A[3];
    int lda[3];
    int ktiles[3];
    const void* W;
    int ldw;
    int M, N, nt_total, ntn, m_valid;
    const float* bias;
    const float* rowbias;
    long long ldrb;
    const int64_t* rowbias_idx;
    const float* residual;
    int ldr;
    int act;
    float* out_f32;
    int ldo_f32;
    bf16_t* out_bf16;
    int ldo_bf16;
};

// LDS-DMA copy of 16 bytes per lane: the wave's 64 lanes fill 1 KiB starting at the wave-uniform `lds`.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}

template <typename T, int BM, int BN>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const GemmParams p) {
    constexpr bool BF = sizeof(T) == 2;
    constexpr int EPC = 16 / (int)sizeof(T);   // elements per 16-byte chunk
    constexpr int BK = 8 * EPC;                // K elements per 128-byte LDS row
    constexpr int A_BYTES = BM * 128;
    constexpr int WTM = BM / 2, WTN = BN / 2;  // wave tile (2 x 2 waves)
    constexpr int TS = BF ? 16 : 32;           // MFMA tile edge: 16x16x32 bf16 / 32x32x2 f32
    constexpr int TM = WTM / TS, TN = WTN / TS;
    constexpr int NACC = BF ? 4 : 16;
    constexpr int A_PIECES = BM / 8 / 4, B_PIECES = BN / 8 / 4;   // 1 KiB pieces (8 rows) per wave
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const sa = smem;
    char* const sb = smem + A_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    int bid = blockIdx.x;
    const int nwg = gridDim.x;
    if ((nwg & 7) == 0) bid = (bid & 7) * (nwg >> 3) + (bid >> 3);
    const int m0 = (bid / p.ntn) * BM, n0 = (bid % p.ntn) * BN;

    // staging: lane l of piece q fills LDS row 8q + (l >> 3), chunk position l & 7, which holds the logical
    // chunk (l & 7) ^ ((row >> 1) & 7): the swizzle is applied on the SOURCE address, the LDS image stays
    // lane-linear as the LDS-DMA requires.
    const int prow = lane >> 3;
    long a_off[A_PIECES], b_off[B_PIECES];
    int a_row[A_PIECES];
#pragma unroll
    for (int j = 0; j < A_PIECES; ++j) {
        const int row = 8 * (wave * A_PIECES + j) + prow;
        a_row[j] = m0 + row;
        a_off[j] = (long)(((lane & 7) ^ ((row >> 1) & 7)) * 16);
    }
#pragma unroll
    for (int j = 0; j < B_PIECES; ++j) {
        const int row = 8 * (wave * B_PIECES + j) + prow;
        b_off[j] = (long)(n0 + row) * p.ldw * (long)sizeof(T) + (long)(((lane & 7) ^ ((row >> 1) & 7)) * 16);
    }
    const char* const Wb = reinterpret_cast<const char*>(p.W);

    typedef float acc_t __attribute__((ext_vector_type(NACC)));
    acc_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < NACC; ++r) acc[i][j][r] = 0.0f;

    // fragment addressing: bf16 16x16x32 -- lane holds row (l & 15), k chunk (l >> 4) of each 32-wide k step;
    // f32 32x32x2 -- lane holds row (l & 31) and four consecutive k of chunk 2s + (l >> 5); A and B use the
    // same k permutation, which is all the contraction needs.
    const int frow = BF ? (lane & 15) : (lane & 31);
    const int fchunk = BF ? (lane >> 4) : (lane >> 5);

    int seg = 0, tt = 0;
    const char* Ab = reinterpret_cast<const char*>(p.A[0]);
    long lda_b = (long)p.lda[0] * (long)sizeof(T);
    for (int t = 0; t < p.nt_total; ++t, ++tt) {
        if (tt == p.ktiles[seg]) {   // next K segment (wave-uniform)
            ++seg; tt = 0;
            if (tt == p.ktiles[seg]) { ++seg; }
            Ab = reinterpret_cast<const char*>(seg == 1 ? p.A[1] : p.A[2]);
            lda_b = (long)(seg == 1 ? p.lda[1] : p.lda[2]) * (long)sizeof(T);
        }
        const long ka = (long)tt * 128, kw = (long)t * 128;
#pragma unroll
        for (int j = 0; j < A_PIECES; ++j)
            glds16(Ab + (long)a_row[j] * lda_b + ka + a_off[j], sa + (wave * A_PIECES + j) * 1024);
#pragma unroll
        for (int j = 0; j < B_PIECES; ++j)
            glds16(Wb + b_off[j] + kw, sb + (wave * B_PIECES + j) * 1024);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if constexpr (BF) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int c = 4 * ks + fchunk;
                uint4 af[TM], bfr[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int row = wm * WTM + i * 16 + frow;
                    af[i] = *reinterpret_cast<const uint4*>(sa + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int row = wn * WTN + j * 16 + frow;
                    bfr[j] = *reinterpret_cast<const uint4*>(sb + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[i]),
                                                                            __builtin_bit_cast(bf16x8, bfr[j]), acc[i][j], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int c = 2 * ks + fchunk;
                f32x4 af[TM], bfr[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int row = wm * WTM + i * 32 + frow;
                    af[i] = *reinterpret_cast<const f32x4*>(sa + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int row = wn * WTN + j * 32 + frow;
                    bfr[j] = *reinterpret_cast<const f32x4*>(sb + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bfr[j][e], acc[i][j], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // epilogue.  C/D layouts: 16x16 -- col = l & 15, row = 4 * (l >> 4) + r; 32x32 -- col = l & 31,
    // row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5).  Optional inputs are read through always-valid pointers
    // (W stands in for an absent one, with stride 0) and masked by selects, so the loads of a row are issued
    // together instead of one wave-uniform branch and one wait per element.
    const int ccol = BF ? (lane & 15) : (lane & 31);
    const int crow0 = BF ? 4 * (lane >> 4) : 4 * (lane >> 5);
    const bool has_bias = p.bias != nullptr, has_rb = p.rowbias != nullptr, has_idx = p.rowbias_idx != nullptr;
    const bool has_res = p.residual != nullptr, act = p.act == 1;
    const float* const dummy = reinterpret_cast<const float*>(p.W);
    const float* const biasp = has_bias ? p.bias : dummy;
    const float* const rbp = has_rb ? p.rowbias : dummy;
    const long ldrb = has_rb ? p.ldrb : 0;
    const float* const resp = has_res ? p.residual : dummy;
    const long ldr = has_res ? p.ldr : 0;
    float bcol[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const float b = biasp[n0 + wn * WTN + j * TS + ccol];
        bcol[j] = has_bias ? b : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < NACC; ++r) {
            const int row = m0 + wm * WTM + i * TS + crow0 + (BF ? r : (r & 3) + 8 * (r >> 2));
            if (row < p.m_valid) {
                const long ridx = has_idx ? (long)p.rowbias_idx[row] : (long)row;
                const float* const rb = rbp + ridx * ldrb;
                const float* const rs = resp + (long)row * ldr;
                float add0[TN], add1[TN];
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int col = n0 + wn * WTN + j * TS + ccol;
                    add0[j] = rb[col];
                    add1[j] = rs[col];
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int col = n0 + wn * WTN + j * TS + ccol;
                    float v = acc[i][j][r] + bcol[j];
                    v += has_rb ? add0[j] : 0.0f;
                    v += has_res ? add1[j] : 0.0f;
                    if (act) v = silu_f(v);   // wave-uniform
                    if (p.out_f32) p.out_f32[(long)row * p.ldo_f32 + col] = v;
                    if (p.out_bf16) p.out_bf16[(long)row * p.ldo_bf16 + col] = (bf16_t)v;
                }
            }
        }
    }
}

template <typename T, int BM, int BN>
static int launch(const GemmParams& p, hipStream_t stream) {
    constexpr int smem = (BM + BN) * 128;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_kernel<T, BM, BN>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return fail(ADCDIFF_E_LAUNCH, "gemm: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    GemmParams q = p;
    q.ntn = p.N / BN;
    const int grid = (p.M / BM) * q.ntn;
    hipLaunchKernelGGL((gemm_nt_kernel<T, BM, BN>), dim3(grid), dim3(256), smem, stream, q);
    return check_launch("gemm");
}

int gemm_dispatch(const adcdiff_gemm_args& a, hipStream_t stream) {
    if (a.dtype != ADCDIFF_F32 && a.dtype != ADCDIFF_BF16) return fail(ADCDIFF_E_BADARG, "gemm: unknown dtype %d", a.dtype);
    const int esize = a.dtype == ADCDIFF_BF16 ? 2 : 4;
    const int bk = 128 / esize;
    if (a.M <= 0 || a.N <= 0 || a.M % 128 != 0 || a.N % 64 != 0)
        return fail(ADCDIFF_E_BADARG, "gemm: M=%d must be a positive multiple of 128 and N=%d of 64", a.M, a.N);
    if (a.nseg < 1 || a.nseg > 3) return fail(ADCDIFF_E_BADARG, "gemm: nseg=%d not in 1..3", a.nseg);
    if (!a.W || !aligned16(a.W) || ((long)a.ldw * esize) % 16 != 0) return fail(ADCDIFF_E_BADARG, "gemm: W null or misaligned");
    if (!a.out_f32 && !a.out_bf16) return fail(ADCDIFF_E_BADARG, "gemm: no output");
    GemmParams p{};
    int ktot = 0;
    for (int s = 0; s < 3; ++s) {
        if (s < a.nseg) {
            if (!a.A[s] || !aligned16(a.A[s]) || a.K[s] <= 0 || a.K[s] % bk != 0 || a.lda[s] < a.K[s] ||
                ((long)a.lda[s] * esize) % 16 != 0)
                return fail(ADCDIFF_E_BADARG, "gemm: segment %d: K=%d (multiple of %d required), lda=%d, ptr %p", s, a.K[s], bk,
                            a.lda[s], a.A[s]);
            p.A[s] = a.A[s]; p.lda[s] = a.lda[s]; p.ktiles[s] = a.K[s] / bk;
            ktot += a.K[s];
        } else {
            p.A[s] = a.A[0]; p.lda[s] = a.lda[0]; p.ktiles[s] = 0;
        }
    }
    if (a.ldw < ktot) return fail(ADCDIFF_E_BADARG, "gemm: ldw=%d < K=%d", a.ldw, ktot);
    if (a.rowbias && a.ldrb != 0 && a.ldrb < a.N) return fail(ADCDIFF_E_BADARG, "gemm: ldrb=%lld < N=%d", a.ldrb, a.N);
    if (a.residual && a.ldr < a.N) return fail(ADCDIFF_E_BADARG, "gemm: ldr=%d < N=%d", a.ldr, a.N);
    if ((a.out_f32 && a.ldo_f32 < a.N) || (a.out_bf16 && a.ldo_bf16 < a.N)) return fail(ADCDIFF_E_BADARG, "gemm: ldo < N");
    if (a.m_valid < 0 || a.m_valid > a.M) return fail(ADCDIFF_E_BADARG, "gemm: m_valid=%d outside 0..M", a.m_valid);
    p.W = a.W; p.ldw = a.ldw; p.M = a.M; p.N = a.N; p.nt_total = ktot / bk;
    p.m_valid = a.m_valid == 0 ? a.M : a.m_valid;
    p.bias = a.bias; p.rowbias = a.rowbias; p.ldrb = a.ldrb; p.rowbias_idx = a.rowbias_idx;
    p.residual = a.residual; p.ldr = a.ldr; p.act = a.act;
    p.out_f32 = a.out_f32; p.ldo_f32 = a.ldo_f32;
    p.out_bf16 = reinterpret_cast<bf16_t*>(a.out_bf16); p.ldo_bf16 = a.ldo_bf16;
    const double esz = esize;
    TimingScope ts(a.dtype == ADCDIFF_BF16 ? ADCDIFF_KIND_GEMM_BF16 : ADCDIFF_KIND_GEMM_F32, a.M, a.N, ktot,
                   2.0 * a.M * a.N * ktot, esz * ((double)a.M * ktot + (double)a.N * ktot) + (double)a.M * a.N * (a.out_f32 ? 4.0 : 0.0) +
                       (double)a.M * a.N * (a.out_bf16 ? 2.0 : 0.0) + (a.residual ? 4.0 * a.M * a.N : 0.0),
                   stream);
    const bool wide = (a.N % 128 == 0);
    if (a.dtype == ADCDIFF_BF16) return wide ? launch<bf16_t, 128, 128>(p, stream) : launch<bf16_t, 128, 64>(p, stream);
    return wide ? launch<float, 128, 128>(p, stream) : launch<float, 128, 64>(p, stream);
}

}  // namespace adcdiff

extern "C" int adcdiff_gemm(const adcdiff_gemm_args* args, adcdiff_stream_t stream) {
    if (!args) return adcdiff::fail(ADCDIFF_E_BADARG, "gemm: null args");
    return adcdiff::gemm_dispatch(*args, adcdiff::as_stream(stream));
}
