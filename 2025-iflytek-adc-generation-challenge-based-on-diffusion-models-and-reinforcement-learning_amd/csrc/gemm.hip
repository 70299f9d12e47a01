// NT GEMM with split-K operands and a fused GroupNorm/SiLU epilogue for the denoiser's nn.Linear layers.
//
//   v[M, N]  = [A0 | A1 | A2][M, K] * W[N, K]^T + bias + rowbias + residual           (float32)
//   outputs  = v as float32, v in the operand type, and up to two SiLU(GroupNorm(v)) in the operand type
//
// One workgroup (4 waves, 2 x 2) computes a BM x BN tile with v_mfma_f32_16x16x32_bf16 (bf16 operands) or
// v_mfma_f32_32x32x2_f32 (float32 operands, exact f32 fma chain), float32 accumulate.
// Main loop: both operand tiles are staged in LDS as rows of 128 bytes (64 bf16 / 32 f32 of K), eight
// 16-byte chunks per row, chunk index XOR-swizzled with (row >> 1) & 7 so that the 16-lane groups of a
// ds_read_b128 hit sixteen different 16-byte slots of the 256-byte bank row.  Tiles go global -> LDS
// directly (global_load_lds_dwordx4, no VGPR round trip, no ds_write); one LDS stage of 32 KiB per
// workgroup, two barriers per K step, latency hidden by 3-4 resident workgroups per CU.  Workgroup ids
// are remapped so that the N-tiles of one M-tile run on one XCD (they share the A rows in its L2).
// Epilogue: the accumulators are transposed through LDS (64 rows x BN columns per pass) so that every
// thread owns 32 contiguous columns of one row: bias / row-bias / residual loads and all stores are 16-byte
// vectors over whole 128-byte lines, and GroupNorm statistics over groups of 4..128 channels are in-lane
// sums plus at most two xor-shuffles (groups never cross the BN-wide tile).  The consumer's GroupNorm is
// applied by the producer, so no layer output is ever re-read just to be normalised.
#include <type_traits>

#include "common.h"

namespace adcdiff {

struct GnOut {
    const float* gamma;   // pre-offset to this GEMM's column 0
    const float* beta;
    void* out;            // operand type, [M, ldo]
    int ldo;
    int gs;               // channels per group
};

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
    const void* residual;   // operand type
    int ldr;
    int act;
    float* out_f32;
    int ldo_f32;
    void* out_op;           // operand type
    int ldo_op;
    int n_gn;
    GnOut gn[2];
    float eps;
};

// LDS-DMA copy of 16 bytes per lane: the wave's 64 lanes fill 1 KiB starting at the wave-uniform `lds`.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}

template <typename T>
__device__ __forceinline__ float silu_out(float v) {
    if constexpr (sizeof(T) == 2) return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));   // result is rounded to bf16
    else return silu_f(v);
}

template <typename T>
__device__ __forceinline__ void store8(T* dst, const float (&y)[8]) {
    if constexpr (sizeof(T) == 2) {
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16_t)y[e];
        *reinterpret_cast<bf16x8*>(dst) = o;
    } else {
        f32x4 o0 = {y[0], y[1], y[2], y[3]}, o1 = {y[4], y[5], y[6], y[7]};
        *reinterpret_cast<f32x4*>(dst) = o0;
        *reinterpret_cast<f32x4*>(dst + 4) = o1;
    }
}

template <typename T>
__device__ __forceinline__ void store32(T* dst, const float (&v)[32]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float y[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) y[e] = v[8 * q + e];
        store8<T>(dst + 8 * q, y);
    }
}

__device__ __forceinline__ void load32(const float* src, float (&y)[32]) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(src + 4 * q);
        y[4 * q] = t[0]; y[4 * q + 1] = t[1]; y[4 * q + 2] = t[2]; y[4 * q + 3] = t[3];
    }
}

__device__ __forceinline__ void add32(const float* src, float (&v)[32]) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(src + 4 * q);
        v[4 * q] += t[0]; v[4 * q + 1] += t[1]; v[4 * q + 2] += t[2]; v[4 * q + 3] += t[3];
    }
}

// SiLU(GroupNorm(v)) for the 32 columns of one row held by this lane; GS channels per group.  Groups of
// 64 / 128 span 2 / 4 adjacent lanes (the threads of one row are adjacent lanes).  Statistics first (two
// passes over the registers), then normalise / scale / activate / store in chunks of 8 columns so that only
// 8 gamma and 8 beta values are live at a time.
template <typename T, int GS>
__device__ __forceinline__ void gn_silu_store(const float (&v)[32], const GnOut& g, long row, int col, float eps, bool live) {
    constexpr int NG = GS <= 32 ? 32 / GS : 1;      // groups held by this lane
    constexpr int GL = GS <= 32 ? GS : 32;          // columns of one group held by this lane
    float mean[NG], rstd[NG];
#pragma unroll
    for (int n = 0; n < NG; ++n) {
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < GL; ++k) s += v[n * GL + k];
        if constexpr (GS >= 64) s += __shfl_xor(s, 1);
        if constexpr (GS == 128) s += __shfl_xor(s, 2);
        mean[n] = s * (1.0f / GS);
        float q = 0.0f;
#pragma unroll
        for (int k = 0; k < GL; ++k) { const float d = v[n * GL + k] - mean[n]; q += d * d; }
        if constexpr (GS >= 64) q += __shfl_xor(q, 1);
        if constexpr (GS == 128) q += __shfl_xor(q, 2);
        rstd[n] = 1.0f / __builtin_sqrtf(q * (1.0f / GS) + eps);
    }
    T* const dst = reinterpret_cast<T*>(g.out) + row * g.ldo + col;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(g.gamma + col + 8 * c), g1 = *reinterpret_cast<const f32x4*>(g.gamma + col + 8 * c + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(g.beta + col + 8 * c), b1 = *reinterpret_cast<const f32x4*>(g.beta + col + 8 * c + 4);
        float y[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = 8 * c + e, n = k / GL;
            const float ga = e < 4 ? g0[e] : g1[e - 4], be = e < 4 ? b0[e] : b1[e - 4];
            y[e] = silu_out<T>((v[k] - mean[n]) * rstd[n] * ga + be);
        }
        if (live) store8<T>(dst + 8 * c, y);
    }
}

template <typename T>
__device__ __forceinline__ void gn_dispatch(const float (&v)[32], const GnOut& g, long row, int col, float eps, bool live) {
    switch (g.gs) {   // wave-uniform
        case 4: gn_silu_store<T, 4>(v, g, row, col, eps, live); break;
        case 8: gn_silu_store<T, 8>(v, g, row, col, eps, live); break;
        case 16: gn_silu_store<T, 16>(v, g, row, col, eps, live); break;
        case 32: gn_silu_store<T, 32>(v, g, row, col, eps, live); break;
        case 64: gn_silu_store<T, 64>(v, g, row, col, eps, live); break;
        default: gn_silu_store<T, 128>(v, g, row, col, eps, live); break;
    }
}

template <typename T, int BM, int BN>
__global__ __launch_bounds__(256, 3) void gemm_nt_kernel(const GemmParams p) {
    constexpr bool BF = sizeof(T) == 2;
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

    // ---- epilogue: transpose through LDS, then one thread = 32 contiguous columns of one row ----
    // C/D layouts: 16x16 -- col = l & 15, row = 4 * (l >> 4) + r; 32x32 -- col = l & 31,
    // row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5).
    constexpr int TPR = BN / 32;        // threads per row
    constexpr int RPP = 256 / TPR;      // block rows per pass
    constexpr int PRW = RPP / 2;        // rows of each wave tile per pass
    constexpr int NPASS = WTM / PRW;
    constexpr int LDSW = BN + 4;        // padded row of the staging image, in floats
    float* const stage = reinterpret_cast<float*>(smem);
    const int ccol = BF ? (lane & 15) : (lane & 31);
    const int crow0 = BF ? 4 * (lane >> 4) : 4 * (lane >> 5);
    const int erow = tid / TPR, echunk = tid % TPR;
    const int col = n0 + echunk * 32;
    const T* const resp = reinterpret_cast<const T*>(p.residual);
    auto pass = [&](auto hc) __attribute__((always_inline)) {
        constexpr int h = decltype(hc)::value;
        if (h > 0) __syncthreads();
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if (i / (PRW / TS) != h) continue;   // folds at compile time: h and i are constants after unrolling
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < NACC; ++r) {
                    const int wrow = i * TS + crow0 + (BF ? r : (r & 3) + 8 * (r >> 2)) - h * PRW;
                    stage[(wm * PRW + wrow) * LDSW + wn * WTN + j * TS + ccol] = acc[i][j][r];
                }
        }
        __syncthreads();
        const long row = m0 + (erow / PRW) * WTM + h * PRW + (erow % PRW);
        const bool live = row < p.m_valid;
        const long lrow = live ? row : m0;   // dead rows compute on row m0's side inputs and store nothing
        float v[32];
        load32(stage + erow * LDSW + echunk * 32, v);
        if (p.bias) add32(p.bias + col, v);
        if (p.rowbias) {
            const long ridx = p.rowbias_idx ? (long)p.rowbias_idx[lrow] : lrow;
            add32(p.rowbias + ridx * p.ldrb + col, v);
        }
        if (resp) {
            const T* rp = resp + lrow * p.ldr + col;
            if constexpr (BF) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const bf16x8 t = *reinterpret_cast<const bf16x8*>(rp + 8 * q);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[8 * q + e] += (float)t[e];
                }
            } else {
                add32(reinterpret_cast<const float*>(rp), v);
            }
        }
        if (p.act == 1) {
#pragma unroll
            for (int k = 0; k < 32; ++k) v[k] = silu_f(v[k]);
        }
        if (p.out_f32 && live) store32<float>(p.out_f32 + row * p.ldo_f32 + col, v);
        if (p.out_op && live) store32<T>(reinterpret_cast<T*>(p.out_op) + row * p.ldo_op + col, v);
        if (p.n_gn > 0) gn_dispatch<T>(v, p.gn[0], row, col, p.eps, live);
        if (p.n_gn > 1) gn_dispatch<T>(v, p.gn[1], row, col, p.eps, live);
    };
    pass(std::integral_constant<int, 0>{});
    if constexpr (NPASS > 1) pass(std::integral_constant<int, 1>{});
    static_assert(NPASS <= 2, "at most two epilogue passes");
}

template <typename T, int BM, int BN>
static int launch(const GemmParams& p, hipStream_t stream) {
    constexpr int main_bytes = (BM + BN) * 128;
    constexpr int epi_bytes = (256 / (BN / 32)) * (BN + 4) * 4;
    constexpr int smem = main_bytes > epi_bytes ? main_bytes : epi_bytes;
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
    if (!a.out_f32 && !a.out_op && a.n_gn <= 0) return fail(ADCDIFF_E_BADARG, "gemm: no output");
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
    if (a.bias && !aligned16(a.bias)) return fail(ADCDIFF_E_BADARG, "gemm: bias must be 16-byte aligned");
    if (a.rowbias && (!aligned16(a.rowbias) || a.ldrb % 4 != 0 || (a.ldrb != 0 && a.ldrb < a.N)))
        return fail(ADCDIFF_E_BADARG, "gemm: rowbias misaligned or ldrb=%lld (0 or a multiple of 4 >= N=%d)", a.ldrb, a.N);
    if (a.residual && (a.ldr < a.N || !aligned16(a.residual) || ((long)a.ldr * esize) % 16 != 0))
        return fail(ADCDIFF_E_BADARG, "gemm: residual misaligned or ldr=%d < N=%d", a.ldr, a.N);
    if (a.out_f32 && (a.ldo_f32 < a.N || !aligned16(a.out_f32) || a.ldo_f32 % 4 != 0)) return fail(ADCDIFF_E_BADARG, "gemm: out_f32 misaligned or ldo < N");
    if (a.out_op && (a.ldo_op < a.N || !aligned16(a.out_op) || ((long)a.ldo_op * esize) % 16 != 0))
        return fail(ADCDIFF_E_BADARG, "gemm: out_op misaligned or ldo < N");
    if (a.m_valid < 0 || a.m_valid > a.M) return fail(ADCDIFF_E_BADARG, "gemm: m_valid=%d outside 0..M", a.m_valid);
    if (a.n_gn < 0 || a.n_gn > 2) return fail(ADCDIFF_E_BADARG, "gemm: n_gn=%d not in 0..2", a.n_gn);
    const bool wide = (a.N % 128 == 0);
    for (int g = 0; g < a.n_gn; ++g) {
        const adcdiff_gn_out& o = a.gn[g];
        const int gs = o.group_size;
        if (!o.gamma || !o.beta || !o.out || !aligned16(o.gamma) || !aligned16(o.beta) || !aligned16(o.out) || o.ldo < a.N ||
            ((long)o.ldo * esize) % 16 != 0)
            return fail(ADCDIFF_E_BADARG, "gemm: GroupNorm output %d: null / misaligned pointers or ldo < N", g);
        if (gs < 4 || gs > (wide ? 128 : 64) || (gs & (gs - 1)) != 0 || a.N % gs != 0)
            return fail(ADCDIFF_E_BADARG, "gemm: GroupNorm output %d: group size %d must be a power of two in 4..%d dividing N=%d", g, gs,
                        wide ? 128 : 64, a.N);
        p.gn[g] = GnOut{o.gamma, o.beta, o.out, o.ldo, gs};
    }
    p.n_gn = a.n_gn; p.eps = a.gn_eps;
    p.W = a.W; p.ldw = a.ldw; p.M = a.M; p.N = a.N; p.nt_total = ktot / bk;
    p.m_valid = a.m_valid == 0 ? a.M : a.m_valid;
    p.bias = a.bias; p.rowbias = a.rowbias; p.ldrb = a.ldrb; p.rowbias_idx = a.rowbias_idx;
    p.residual = a.residual; p.ldr = a.ldr; p.act = a.act;
    p.out_f32 = a.out_f32; p.ldo_f32 = a.ldo_f32;
    p.out_op = a.out_op; p.ldo_op = a.ldo_op;
    const double esz = esize;
    const double nout = (a.out_f32 ? 4.0 : 0.0) + esz * ((a.out_op ? 1 : 0) + a.n_gn);
    TimingScope ts(a.dtype == ADCDIFF_BF16 ? ADCDIFF_KIND_GEMM_BF16 : ADCDIFF_KIND_GEMM_F32, a.M, a.N, ktot, 2.0 * a.M * a.N * ktot,
                   esz * ((double)a.M * ktot + (double)a.N * ktot) + (double)a.M * a.N * (nout + (a.residual ? esz : 0.0)), stream);
    if (a.dtype == ADCDIFF_BF16) return wide ? launch<bf16_t, 128, 128>(p, stream) : launch<bf16_t, 128, 64>(p, stream);
    return wide ? launch<float, 128, 128>(p, stream) : launch<float, 128, 64>(p, stream);
}

}  // namespace adcdiff

extern "C" int adcdiff_gemm(const adcdiff_gemm_args* args, adcdiff_stream_t stream) {
    if (!args) return adcdiff::fail(ADCDIFF_E_BADARG, "gemm: null args");
    return adcdiff::gemm_dispatch(*args, adcdiff::as_stream(stream));
}
