// NT GEMM with split-K operands and fused epilogue for the denoiser's nn.Linear layers.
//
//   out[M, N] = act( [A0 | A1 | A2][M, K] * W[N, K]^T + bias + rowbias + residual )
//
// One workgroup (4 waves, 2 x 2) computes a BM x BN tile with v_mfma_f32_32x32x16_bf16 (bf16
// operands) or v_mfma_f32_32x32x2_f32 (float32 operands, exact f32 fma chain), float32 accumulate.
// Both operand tiles are staged in LDS as rows of 128 bytes (64 bf16 / 32 f32 of K), eight 16-byte
// chunks per row, chunk index XOR-swizzled with (row >> 1) & 7 so that the 16-lane groups of a
// ds_read_b128 hit sixteen different 16-byte slots of the 256-byte bank row.  Global loads for tile
// t+1 are issued before the MFMAs of tile t and written to the other LDS stage after them.
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

template <typename T, int BM, int BN>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const GemmParams p) {
    constexpr int EPC = 16 / (int)sizeof(T);   // elements per 16-byte chunk
    constexpr int BK = 8 * EPC;                // K elements per 128-byte LDS row
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    constexpr int TM = BM / 64, TN = BN / 64;  // 32x32 MFMA tiles per wave
    constexpr int A_CH = BM * 8 / 256, B_CH = BN * 8 / 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    int bid = blockIdx.x;
    const int nwg = gridDim.x;
    if ((nwg & 7) == 0) bid = (bid & 7) * (nwg >> 3) + (bid >> 3);
    const int m0 = (bid / p.ntn) * BM, n0 = (bid % p.ntn) * BN;

    uint4 ra[A_CH], rb[B_CH];
    const char* Wb = reinterpret_cast<const char*>(p.W);
    const long ldw_b = (long)p.ldw * (long)sizeof(T);

    auto gload = [&](int t) {
        int s = 0, tt = t;
        if (tt >= p.ktiles[0]) { tt -= p.ktiles[0]; s = 1; if (tt >= p.ktiles[1]) { tt -= p.ktiles[1]; s = 2; } }
        const char* Ab = reinterpret_cast<const char*>(s == 0 ? p.A[0] : (s == 1 ? p.A[1] : p.A[2]));
        const long lda_b = (long)(s == 0 ? p.lda[0] : (s == 1 ? p.lda[1] : p.lda[2])) * (long)sizeof(T);
#pragma unroll
        for (int j = 0; j < A_CH; ++j) {
            const int q = tid + 256 * j, row = q >> 3, c = q & 7;
            ra[j] = *reinterpret_cast<const uint4*>(Ab + (long)(m0 + row) * lda_b + ((long)tt * BK + c * EPC) * (long)sizeof(T));
        }
#pragma unroll
        for (int j = 0; j < B_CH; ++j) {
            const int q = tid + 256 * j, row = q >> 3, c = q & 7;
            rb[j] = *reinterpret_cast<const uint4*>(Wb + (long)(n0 + row) * ldw_b + ((long)t * BK + c * EPC) * (long)sizeof(T));
        }
    };
    auto lstore = [&](int buf) {
        char* sa = smem + buf * STAGE;
        char* sb = sa + A_BYTES;
#pragma unroll
        for (int j = 0; j < A_CH; ++j) {
            const int q = tid + 256 * j, row = q >> 3, c = q & 7;
            *reinterpret_cast<uint4*>(sa + row * 128 + ((c ^ ((row >> 1) & 7)) << 4)) = ra[j];
        }
#pragma unroll
        for (int j = 0; j < B_CH; ++j) {
            const int q = tid + 256 * j, row = q >> 3, c = q & 7;
            *reinterpret_cast<uint4*>(sb + row * 128 + ((c ^ ((row >> 1) & 7)) << 4)) = rb[j];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int lrow = lane & 31, lhalf = lane >> 5;
    auto compute = [&](int buf) {
        const char* sa = smem + buf * STAGE;
        const char* sb = sa + A_BYTES;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int c = 2 * s + lhalf;
            uint4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = wm * (BM / 2) + i * 32 + lrow;
                af[i] = *reinterpret_cast<const uint4*>(sa + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row = wn * (BN / 2) + j * 32 + lrow;
                bf[j] = *reinterpret_cast<const uint4*>(sb + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if constexpr (sizeof(T) == 2) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(bf16x8, af[i]), __builtin_bit_cast(bf16x8, bf[j]), acc[i][j], 0, 0, 0);
                    } else {
                        const f32x4 a4 = __builtin_bit_cast(f32x4, af[i]), b4 = __builtin_bit_cast(f32x4, bf[j]);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], b4[e], acc[i][j], 0, 0, 0);
                    }
                }
        }
    };

    const int nt = p.nt_total;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        if (t + 1 < nt) gload(t + 1);
        compute(t & 1);
        if (t + 1 < nt) lstore((t + 1) & 1);
        __syncthreads();
    }

    // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * (BN / 2) + j * 32 + lrow;
            const float bcol = p.bias ? p.bias[col] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
                if (row >= p.m_valid) continue;
                float v = acc[i][j][r] + bcol;
                if (p.rowbias) {
                    const long rr = p.rowbias_idx ? (long)p.rowbias_idx[row] : (long)row;
                    v += p.rowbias[rr * p.ldrb + col];
                }
                if (p.residual) v += p.residual[(long)row * p.ldr + col];
                if (p.act == 1) v = silu_f(v);
                if (p.out_f32) p.out_f32[(long)row * p.ldo_f32 + col] = v;
                if (p.out_bf16) p.out_bf16[(long)row * p.ldo_bf16 + col] = (bf16_t)v;
            }
        }
    }
}

template <typename T, int BM, int BN>
static int launch(const GemmParams& p, hipStream_t stream) {
    constexpr int smem = 2 * (BM + BN) * 128;
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
    const bool wide = (a.N % 128 == 0);
    if (a.dtype == ADCDIFF_BF16) return wide ? launch<bf16_t, 128, 128>(p, stream) : launch<bf16_t, 128, 64>(p, stream);
    return wide ? launch<float, 128, 128>(p, stream) : launch<float, 128, 64>(p, stream);
}

}  // namespace adcdiff

extern "C" int adcdiff_gemm(const adcdiff_gemm_args* args, adcdiff_stream_t stream) {
    if (!args) return adcdiff::fail(ADCDIFF_E_BADARG, "gemm: null args");
    return adcdiff::gemm_dispatch(*args, adcdiff::as_stream(stream));
}
