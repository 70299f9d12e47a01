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
// Epilogue (struct Epilogue below): each wave transposes its accumulators through a private LDS region so that
// a lane owns 8 contiguous columns of a row; GroupNorm statistics are in-lane / DPP sums.  The consumer's
// GroupNorm is applied by the producer, so no layer output is ever re-read just to be normalised.
#include <cstdlib>
#ifdef ADCDIFF_STAMPS
#include <algorithm>
#include <cstdio>
#include <vector>
#endif
#include <type_traits>

#include <atomic>
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
    int rowbias_rows;
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
    adcdiff_step_update upd;     // schedule update applied to v before the outputs (kind 0 = none)
    int ksplit;                  // grid.y: K slices (1 without split-K)
    long long split_k_bytes;     // byte offset along K of slice blockIdx.y (0 without split-K)
    long long split_out_stride;  // floats between the partial outputs of consecutive slices
    int pf_w;                    // four-wave kernel: prefetch the W rows as well (debug switch gemm_pf_w, on by default)
    int pf_next;                 // four-wave kernel: workgroups resident at a time (one per CU), 0 = nothing is touched for the successor
    int sk_splits;               // few-row split-K kernel: workgroups per output tile (0 = not used)
    float* sk_part;              // its partial accumulators, [tiles][sk_splits][TM * TN][64 lanes][4] floats
    unsigned* sk_cnt;            // its arrival counters, one per output tile; zero before and after every launch
#ifdef ADCDIFF_STAMPS
    long long* stamps;
    int* tile_stamps;   // four-wave kernel: 64 low words of the 100 MHz counter per workgroup (wave 0): entry, pieces issued, tile 0 landed, every tile's start, end
#endif
};

template <typename T>
__device__ __forceinline__ float silu_out(float v) {
    if constexpr (sizeof(T) == 2) return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));   // result is rounded to bf16
    else return silu_f(v);
}

// 1 / sqrt(x): the hardware estimate (1 ulp) where the result is rounded to bf16 anyway, the exact pair otherwise.
template <typename T>
__device__ __forceinline__ float inv_sqrt_out(float x) {
    if constexpr (sizeof(T) == 2) return __builtin_amdgcn_rsqf(x);
    else return 1.0f / __builtin_sqrtf(x);
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

__device__ __forceinline__ void load8(const float* src, float (&y)[8]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
    y[0] = a[0]; y[1] = a[1]; y[2] = a[2]; y[3] = a[3]; y[4] = b[0]; y[5] = b[1]; y[6] = b[2]; y[7] = b[3];
}

// Sum over LANES adjacent lanes (aligned group) with DPP row operations: quad_perm for xor 1 and 2, then
// row_half_mirror (the quads are uniform by then, so mirroring within 8 lanes brings in the other quad).
template <int CTRL>
__device__ __forceinline__ float dpp_f(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
template <int LANES>
__device__ __forceinline__ float group_sum(float x) {
    if constexpr (LANES >= 2) x += dpp_f<0xB1>(x);    // quad_perm [1,0,3,2]
    if constexpr (LANES >= 4) x += dpp_f<0x4E>(x);    // quad_perm [2,3,0,1]
    if constexpr (LANES >= 8) x += dpp_f<0x141>(x);   // row_half_mirror
    return x;
}

// ---- epilogue shared by the GEMM kernels -------------------------------------------------------------------
// Every wave transposes its own WTM x WTN accumulator tile through a private LDS region, 32 rows at a time, so
// that a lane owns 8 contiguous columns of one row (WTN / 8 lanes per row): no workgroup barrier, the two waves
// of a SIMD interleave freely, every global store is a 16-byte piece of a full 128-byte line.  Column-dependent
// inputs (bias + broadcast row bias, GroupNorm weight / bias of both outputs) are fetched into LDS at kernel start
// and sit in registers for the whole tile.  GroupNorm groups of 4 / 8 channels are in-lane, 16..WTN channels are
// DPP sums over 2..8 adjacent lanes; a group of 2 * WTN channels (128 at WTN = 64) spans the two waves wn, wn ^ 1,
// which exchange (mean, centred sum of squares) through LDS once per step and combine them with the pairwise
// (Chan) update -- the only place the epilogue uses a workgroup barrier.
// C/D layouts: 16x16 -- col = l & 15, row = 4 * (l >> 4) + r; 32x32 -- col = l & 31,
// row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5).
template <typename T, int NT, int BN, int WTM, int WTN, int TM, int TN>
struct Epilogue {
    static constexpr bool BF = sizeof(T) == 2;
    static constexpr int TS = BF ? 16 : 32;
    static constexpr int NACC = BF ? 4 : 16;
    static constexpr int NWAVE = NT / 64;
    static constexpr int CH = 32;                   // rows per chunk
    static constexpr int NCHUNK = WTM / CH;
    static constexpr int LPR = WTN / 8;             // lanes per row
    static constexpr int RPS = 64 / LPR;            // rows per step
    static constexpr int NSTEP = CH / RPS;
    static constexpr int LDW = WTN + 4;             // floats per staged row
    static constexpr int STAGE_BYTES = NWAVE * CH * LDW * 4;
    static constexpr int PRM_BYTES = 5 * BN * 4;    // cb, gamma0, beta0, gamma1, beta1
    static constexpr int XB = NWAVE * RPS * 2;      // floats of one exchange buffer: (mean, M2) per wave and row
    static constexpr int XCH_BYTES = 4 * XB * 4;    // [step parity][GroupNorm output]
    static constexpr int EXTRA_BYTES = PRM_BYTES + XCH_BYTES;
    typedef float acc_t __attribute__((ext_vector_type(NACC)));
    static_assert(NCHUNK >= 1 && NCHUNK <= 4 && CH % TS == 0 && (LPR == 16 || LPR == 8 || LPR == 4), "unsupported epilogue geometry");

    // Called by all threads at kernel start: column-dependent vectors of this block's BN columns -> LDS.
    static __device__ __forceinline__ void stage_params(const GemmParams& p, int n0, int tid, float* prm) {
        for (int idx = tid; idx < 5 * BN / 4; idx += NT) {
            const int vec = idx / (BN / 4), c = (idx % (BN / 4)) * 4;
            f32x4 val = {0.0f, 0.0f, 0.0f, 0.0f};
            if (vec == 0) {
                if (p.bias) val = *reinterpret_cast<const f32x4*>(p.bias + n0 + c);
                if (p.rowbias && !p.rowbias_idx && p.ldrb == 0) {
                    const f32x4 rb = *reinterpret_cast<const f32x4*>(p.rowbias + n0 + c);
                    val[0] += rb[0]; val[1] += rb[1]; val[2] += rb[2]; val[3] += rb[3];
                }
            } else {
                const int g = (vec - 1) >> 1;
                if (g < p.n_gn) val = *reinterpret_cast<const f32x4*>(((vec - 1) & 1 ? p.gn[g].beta : p.gn[g].gamma) + n0 + c);
            }
            *reinterpret_cast<f32x4*>(prm + vec * BN + c) = val;
        }
    }

    // The same in two halves, for a kernel whose main loop leaves registers free: the loads go out in front of the main loop and
    // their round trip runs under it; the values reach LDS behind it (followed by the caller's barrier).
    static constexpr int NPV = (5 * BN / 4 + NT - 1) / NT;
    struct ParamRegs { f32x4 v[NPV], rb; };
    static __device__ __forceinline__ void load_params(const GemmParams& p, int n0, int tid, ParamRegs& pr) {
        const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
        pr.rb = zero;
        // the five sources in scalar registers first, chosen by comparisons: an index into p.gn (or a choice between the fields'
        // addresses) would be resolved by a vector load from the argument block, a round trip in front of the load proper
        const float *s0 = p.bias, *s1 = p.gn[0].gamma, *s2 = p.gn[0].beta, *s3 = p.gn[1].gamma, *s4 = p.gn[1].beta;
        asm volatile("" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4));
#pragma unroll
        for (int i = 0; i < NPV; ++i) {
            const int idx = tid + i * NT;
            const int vec = idx / (BN / 4), c = (idx % (BN / 4)) * 4;
            const float* src = vec == 0 ? s0 : vec == 1 ? s1 : vec == 2 ? s2 : vec == 3 ? s3 : s4;
            const bool have = idx < 5 * BN / 4 && (vec == 0 ? p.bias != nullptr : (vec - 1) >> 1 < p.n_gn);
            typedef const __attribute__((address_space(1))) f32x4* gvec_t;   // global, not generic: the empty asm hides where s0..s4 point
            if (have) pr.v[i] = *(gvec_t)(src + n0 + c);
            else pr.v[i] = zero;
            if (i == 0 && vec == 0 && p.rowbias && !p.rowbias_idx && p.ldrb == 0) pr.rb = *reinterpret_cast<const f32x4*>(p.rowbias + n0 + c);
        }
    }
    static __device__ __forceinline__ void store_params(const GemmParams& p, int tid, float* prm, const ParamRegs& pr) {
        // vmcnt(0), unconditionally: the loads of load_params sit behind run-time conditions, and without a wait on every path the
        // compiler keeps their registers as "possibly still being loaded" and waits for all loads wherever it reuses one of them
        __builtin_amdgcn_s_waitcnt(0x0f70);
#pragma unroll
        for (int i = 0; i < NPV; ++i) {
            const int idx = tid + i * NT;
            f32x4 val = pr.v[i];
            if (i == 0 && idx < BN / 4 && p.rowbias && !p.rowbias_idx && p.ldrb == 0) {   // as stage_params: bias, then the broadcast row
                val[0] += pr.rb[0]; val[1] += pr.rb[1]; val[2] += pr.rb[2]; val[3] += pr.rb[3];
            }
            if (idx < 5 * BN / 4) *reinterpret_cast<f32x4*>(prm + (idx / (BN / 4)) * BN + (idx % (BN / 4)) * 4) = val;
        }
    }

    // SiLU(GroupNorm(v)) of the lane's 8 columns in the operand type: the arithmetic only, no store and no lane-dependent
    // branch, so that several rows can be computed in one basic block and interleaved by the scheduler (gn_multi)
    typedef T outv_t __attribute__((ext_vector_type(8)));
    template <int GS>
    static __device__ __forceinline__ void gn_compute(const float (&v)[8], const float (&gam)[8], const float (&bet)[8], outv_t& out, float eps,
                                                      float* xch, int wave, int rr, int cg) {
        float y[8];
        if constexpr (GS == 4) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float mean = ((v[4 * h] + v[4 * h + 1]) + (v[4 * h + 2] + v[4 * h + 3])) * 0.25f;
                float q = 0.0f;
#pragma unroll
                for (int k = 0; k < 4; ++k) { const float d = v[4 * h + k] - mean; q += d * d; }
                const float rstd = inv_sqrt_out<T>(q * 0.25f + eps);
#pragma unroll
                for (int k = 0; k < 4; ++k) y[4 * h + k] = silu_out<T>((v[4 * h + k] - mean) * rstd * gam[4 * h + k] + bet[4 * h + k]);
            }
        } else {
            // A group of 128 channels is always summed as two halves of 64 that are merged with the pairwise update, so that
            // every tiling produces the same bits: the halves sit in two waves (WTN = 64, exchange through LDS) or in the two
            // 8-lane halves of a 16-lane row of one wave (WTN = 128, exchange with a DPP rotate by 8).
            constexpr bool DPP_SPLIT = WTN == 128 && GS == 128;
            constexpr bool LDS_SPLIT = !DPP_SPLIT && GS > WTN;
            constexpr int IN_WAVE = DPP_SPLIT ? 64 : GS <= WTN ? GS : WTN;   // channels summed inside the wave before any merge
            constexpr int LANES = IN_WAVE / 8;
            if constexpr (BF) {
                // bf16 outputs: the same two-pass statistics on pairs of elements (v_pk_add / v_pk_mul / v_pk_fma_f32 do two
                // lanes' worth per issue slot: the epilogue is VALU-issue bound, ~15 -> ~9 instructions per element), SiLU
                // through v_exp_f32 / v_rcp_f32 directly
                // every fused multiply-add is written out and contraction is off: which of two products the compiler fuses into
                // a sum is otherwise its choice per call site, and this code and gn_multi below must give the same bits
#pragma clang fp contract(off)
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                f32x2 d2[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) d2[i] = f32x2{v[2 * i], v[2 * i + 1]};
                const f32x2 t2 = (d2[0] + d2[1]) + (d2[2] + d2[3]);
                float mean = group_sum<LANES>(t2[0] + t2[1]) * (1.0f / IN_WAVE);
#pragma unroll
                for (int i = 0; i < 4; ++i) d2[i] = d2[i] - mean;
                f32x2 q2 = d2[0] * d2[0];
#pragma unroll
                for (int i = 1; i < 4; ++i) q2 = __builtin_elementwise_fma(d2[i], d2[i], q2);
                float q = group_sum<LANES>(q2[0] + q2[1]);
                if constexpr (LDS_SPLIT || DPP_SPLIT) {
                    float om, oq;
                    if constexpr (LDS_SPLIT) {
                        if (cg == 0) { xch[(wave * RPS + rr) * 2] = mean; xch[(wave * RPS + rr) * 2 + 1] = q; }
                        __syncthreads();
                        om = xch[((wave ^ 1) * RPS + rr) * 2]; oq = xch[((wave ^ 1) * RPS + rr) * 2 + 1];
                    } else {
                        om = dpp_f<0x128>(mean); oq = dpp_f<0x128>(q);   // row_ror:8 = the other half of the 16-lane row
                    }
                    const float delta = om - mean;
                    q = __builtin_fmaf(delta * delta, 0.5f * IN_WAVE, q + oq);
                    const float shift = -0.5f * delta;   // old mean - merged mean
#pragma unroll
                    for (int i = 0; i < 4; ++i) d2[i] = d2[i] + shift;
                }
                const float rstd = __builtin_amdgcn_rsqf(__builtin_fmaf(q, 1.0f / GS, eps));
                bf16x8 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x2 a2 = f32x2{gam[2 * i], gam[2 * i + 1]} * rstd;
                    const f32x2 y2 = __builtin_elementwise_fma(d2[i], a2, f32x2{bet[2 * i], bet[2 * i + 1]});
                    f32x2 e2 = y2 * -1.4426950408889634f;
                    e2 = f32x2{__builtin_amdgcn_exp2f(e2[0]), __builtin_amdgcn_exp2f(e2[1])} + 1.0f;
                    const f32x2 o2 = y2 * f32x2{__builtin_amdgcn_rcpf(e2[0]), __builtin_amdgcn_rcpf(e2[1])};
                    o[2 * i] = (bf16_t)o2[0];
                    o[2 * i + 1] = (bf16_t)o2[1];
                }
                out = o;
                return;
            }
            float s = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
            s = group_sum<LANES>(s);
            float mean = s * (1.0f / IN_WAVE);
            float q = 0.0f;
#pragma unroll
            for (int k = 0; k < 8; ++k) { const float d = v[k] - mean; q += d * d; }
            q = group_sum<LANES>(q);
            if constexpr (LDS_SPLIT || DPP_SPLIT) {
                // the other half of the group lives in wave ^ 1 (or in the other 8 lanes of the row): exchange (mean, M2) and merge
                float om, oq;
                if constexpr (LDS_SPLIT) {
                    if (cg == 0) { xch[(wave * RPS + rr) * 2] = mean; xch[(wave * RPS + rr) * 2 + 1] = q; }
                    __syncthreads();
                    om = xch[((wave ^ 1) * RPS + rr) * 2]; oq = xch[((wave ^ 1) * RPS + rr) * 2 + 1];
                } else {
                    om = dpp_f<0x128>(mean); oq = dpp_f<0x128>(q);
                }
                const float delta = om - mean;
                q = q + oq + delta * delta * (0.5f * IN_WAVE);
                mean = 0.5f * (mean + om);
            }
            const float rstd = inv_sqrt_out<T>(q * (1.0f / GS) + eps);
#pragma unroll
            for (int k = 0; k < 8; ++k) y[k] = silu_out<T>((v[k] - mean) * rstd * gam[k] + bet[k]);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) out[k] = (T)y[k];
    }

    template <int GS>
    static __device__ __forceinline__ void gn_store(const float (&v)[8], const float (&gam)[8], const float (&bet)[8], T* dst, float eps,
                                                    bool live, float* xch, int wave, int rr, int cg) {
        outv_t o;
        gn_compute<GS>(v, gam, bet, o, eps, xch, wave, rr, cg);
        if (live) *reinterpret_cast<outv_t*>(dst) = o;
    }

    // R rows at once: R independent dependency chains in one basic block (a wave that is alone on its SIMD has nothing else to
    // fill the latency of the DPP sums and of v_exp / v_rcp with), the stores behind them.  Not for groups that span two waves.
    template <int GS, int R>
    static __device__ __forceinline__ void gn_multi(const float (&v)[R][8], const float (&gam)[8], const float (&bet)[8], T* const (&dst)[R],
                                                    float eps, const bool (&live)[R], int wave, int rr, int cg) {
        static_assert(GS <= WTN, "gn_multi: the group must lie inside the wave");
        outv_t o[R];
        if constexpr (BF && GS >= 8) {
            // gn_compute's bf16 arithmetic, row for row the same operations in the same order, written operation by operation
            // ACROSS the rows: the scheduler keeps source order, so the R dependency chains (DPP sums, v_exp, v_rcp) interleave
#pragma clang fp contract(off)
            constexpr bool DPP_SPLIT = WTN == 128 && GS == 128;
            constexpr int IN_WAVE = DPP_SPLIT ? 64 : GS;
            constexpr int LANES = IN_WAVE / 8;
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            f32x2 d2[R][4], q2[R];
            float mean[R], q[R], rstd[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
#pragma unroll
                for (int i = 0; i < 4; ++i) d2[r][i] = f32x2{v[r][2 * i], v[r][2 * i + 1]};
                const f32x2 t2 = (d2[r][0] + d2[r][1]) + (d2[r][2] + d2[r][3]);
                mean[r] = t2[0] + t2[1];
            }
            auto sum_rows = [&](float (&x)[R]) {
                if constexpr (LANES >= 2) {
#pragma unroll
                    for (int r = 0; r < R; ++r) x[r] += dpp_f<0xB1>(x[r]);
                }
                if constexpr (LANES >= 4) {
#pragma unroll
                    for (int r = 0; r < R; ++r) x[r] += dpp_f<0x4E>(x[r]);
                }
                if constexpr (LANES >= 8) {
#pragma unroll
                    for (int r = 0; r < R; ++r) x[r] += dpp_f<0x141>(x[r]);
                }
            };
            sum_rows(mean);
#pragma unroll
            for (int r = 0; r < R; ++r) mean[r] *= 1.0f / IN_WAVE;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < R; ++r) d2[r][i] = d2[r][i] - mean[r];
#pragma unroll
            for (int r = 0; r < R; ++r) q2[r] = d2[r][0] * d2[r][0];
#pragma unroll
            for (int i = 1; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < R; ++r) q2[r] = __builtin_elementwise_fma(d2[r][i], d2[r][i], q2[r]);
#pragma unroll
            for (int r = 0; r < R; ++r) q[r] = q2[r][0] + q2[r][1];
            sum_rows(q);
            if constexpr (DPP_SPLIT) {
                float om[R], oq[R];
#pragma unroll
                for (int r = 0; r < R; ++r) { om[r] = dpp_f<0x128>(mean[r]); oq[r] = dpp_f<0x128>(q[r]); }
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const float delta = om[r] - mean[r];
                    q[r] = __builtin_fmaf(delta * delta, 0.5f * IN_WAVE, q[r] + oq[r]);
                    const float shift = -0.5f * delta;
#pragma unroll
                    for (int i = 0; i < 4; ++i) d2[r][i] = d2[r][i] + shift;
                }
            }
#pragma unroll
            for (int r = 0; r < R; ++r) rstd[r] = __builtin_amdgcn_rsqf(__builtin_fmaf(q[r], 1.0f / GS, eps));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x2 y2[R], e2[R];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const f32x2 a2 = f32x2{gam[2 * i], gam[2 * i + 1]} * rstd[r];
                    y2[r] = __builtin_elementwise_fma(d2[r][i], a2, f32x2{bet[2 * i], bet[2 * i + 1]});
                    e2[r] = y2[r] * -1.4426950408889634f;
                }
#pragma unroll
                for (int r = 0; r < R; ++r) e2[r] = f32x2{__builtin_amdgcn_exp2f(e2[r][0]), __builtin_amdgcn_exp2f(e2[r][1])} + 1.0f;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const f32x2 o2 = y2[r] * f32x2{__builtin_amdgcn_rcpf(e2[r][0]), __builtin_amdgcn_rcpf(e2[r][1])};
                    o[r][2 * i] = (bf16_t)o2[0];
                    o[r][2 * i + 1] = (bf16_t)o2[1];
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) gn_compute<GS>(v[r], gam, bet, o[r], eps, nullptr, wave, rr, cg);
        }
        // keep the arithmetic in front of the lane-dependent stores (the compiler would sink each row's into its branch)
#pragma unroll
        for (int r = 0; r < R; ++r) asm volatile("" : "+v"(o[r]));
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (live[r]) *reinterpret_cast<outv_t*>(dst[r]) = o[r];
    }

    template <int R>
    static __device__ __forceinline__ void gn_dispatch_multi(int gs, const float (&v)[R][8], const float (&gam)[8], const float (&bet)[8],
                                                             T* const (&dst)[R], float eps, const bool (&live)[R], int wave, int rr, int cg) {
        switch (gs) {   // workgroup-uniform
            case 4: gn_multi<4, R>(v, gam, bet, dst, eps, live, wave, rr, cg); break;
            case 8: gn_multi<8, R>(v, gam, bet, dst, eps, live, wave, rr, cg); break;
            case 16: gn_multi<16, R>(v, gam, bet, dst, eps, live, wave, rr, cg); break;
            case 32: gn_multi<32, R>(v, gam, bet, dst, eps, live, wave, rr, cg); break;
            case 64:
                if constexpr (WTN >= 64) gn_multi<64, R>(v, gam, bet, dst, eps, live, wave, rr, cg);
                break;
            default:
                if constexpr (WTN >= 128) gn_multi<128, R>(v, gam, bet, dst, eps, live, wave, rr, cg);
                break;
        }
    }

    static __device__ __forceinline__ void gn_dispatch(int gs, const float (&v)[8], const float (&gam)[8], const float (&bet)[8], T* dst,
                                                       float eps, bool live, float* xch, int wave, int rr, int cg) {
        switch (gs) {   // workgroup-uniform
            case 4: gn_store<4>(v, gam, bet, dst, eps, live, xch, wave, rr, cg); break;
            case 8: gn_store<8>(v, gam, bet, dst, eps, live, xch, wave, rr, cg); break;
            case 16: gn_store<16>(v, gam, bet, dst, eps, live, xch, wave, rr, cg); break;
            case 32: gn_store<32>(v, gam, bet, dst, eps, live, xch, wave, rr, cg); break;
            case 64: gn_store<64>(v, gam, bet, dst, eps, live, xch, wave, rr, cg); break;
            default:
                if constexpr (WTN >= 64) gn_store<128>(v, gam, bet, dst, eps, live, xch, wave, rr, cg);
                break;
        }
    }

    // Accumulators of row chunk H (32 rows of the wave tile) -> this wave's staging region.
    template <int H>
    static __device__ __forceinline__ void stage_chunk(const acc_t (&acc)[TM][TN], float* stage, int crow0, int ccol) {
        if constexpr (H < NCHUNK) {
#pragma unroll
            for (int i = H * (CH / TS); i < (H + 1) * (CH / TS); ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < NACC; ++r) {
                        const int crow = i * TS + crow0 + (BF ? r : (r & 3) + 8 * (r >> 2)) - H * CH;
                        stage[crow * LDW + j * TS + ccol] = acc[i][j][r];
                    }
        }
    }

    // v = predicted noise of 8 contiguous columns of row `row` -> the latent after the reverse step (see adcdiff_step_update).
    // Same expressions, in the same order and without fma contraction, as ddpm_step_kernel / ddim_step_kernel.
    static __device__ __forceinline__ void apply_update(const adcdiff_step_update& u, float (&v)[8], long row, long col) {
#pragma clang fp contract(off)
        float x[8], z[8];
        load8(u.x + row * u.ldx + col, x);
        const float sigma = u.kind == 1 ? u.c[3] : u.c[4];
        if (sigma != 0.0f) {
            if (u.noise) {
                load8(u.noise + row * u.ldn + col, z);
            } else {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f32x4 q = philox_normal4(u.seed, ADCDIFF_STREAM_STEP, u.step, u.row_offset + (uint64_t)row, (uint32_t)(col >> 2) + h);
                    z[4 * h] = q[0]; z[4 * h + 1] = q[1]; z[4 * h + 2] = q[2]; z[4 * h + 3] = q[3];
                }
            }
        }
        if (u.kind == 1) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float o = u.c[0] * (x[k] - (u.c[1] * v[k]) / u.c[2]);
                if (sigma != 0.0f) o = o + sigma * z[k];
                v[k] = o;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float dir = u.c[3] * v[k];
                if (sigma != 0.0f) dir = dir + sigma * z[k];
                const float pred_x0 = (x[k] - u.c[0] * v[k]) / u.c[1];
                v[k] = u.c[2] * pred_x0 + dir;
            }
        }
    }

    // One step: RPS rows of the staged chunk, a lane owning 8 contiguous columns of one row.  step_load fetches what the step
    // reads (the staged accumulators, the row's bias-table row, the residual); step_finish does the arithmetic and the stores.
    // A wave that is alone on its SIMD runs the loads of several steps before the first finish (run<ROLLED, U>): nothing
    // else on the SIMD would cover their latency.
    struct StepIn {
        float v[8], rb[8];
        bf16x8 res_bf;
        float res_f[8];
    };
    // The step's inputs from global memory (latencies of microseconds under load): the row's bias-table row and the residual.
    // SIDE: what the kernel knows at compile time about these two inputs.  0: nothing (both tested at run time); 1: no per-row bias
    // and no residual; 2: no per-row bias, a residual.  A load that is skipped at run time leaves the compiler with registers that
    // "may" still be the target of a load in flight, and it then waits for ALL loads (vmcnt(0)) wherever one of them is reused:
    // a look-ahead only works where the loads are unconditional.
    template <int SIDE>
    static __device__ __forceinline__ bool has_row_rb(const GemmParams& p) {
        return SIDE == 0 && p.rowbias != nullptr && (p.rowbias_idx != nullptr || p.ldrb != 0);
    }
    template <int SIDE>
    static __device__ __forceinline__ bool has_residual(const GemmParams& p) { return SIDE == 0 ? p.residual != nullptr : SIDE == 2; }

    template <int SIDE = 0>
    static __device__ __forceinline__ void rowbias_load(const GemmParams& p, int h, int st, int m0, int wm, int rr, long col, StepIn& in) {
        const long row = m0 + wm * WTM + h * CH + st * RPS + rr;
        const long lrow = row < p.m_valid ? row : 0;    // dead rows read row 0's side inputs (always valid) and store nothing
        if (has_row_rb<SIDE>(p)) {
            long ridx = p.rowbias_idx ? (long)p.rowbias_idx[lrow] : lrow;
            if (p.rowbias_rows > 0) ridx = ridx < 0 ? 0 : ridx >= p.rowbias_rows ? p.rowbias_rows - 1 : ridx;   // never read outside the table
            load8(p.rowbias + ridx * p.ldrb + col, in.rb);
        }
    }
    template <int SIDE = 0>
    static __device__ __forceinline__ void residual_load(const GemmParams& p, int h, int st, int m0, int wm, int rr, long col, StepIn& in) {
        const T* const resp = reinterpret_cast<const T*>(p.residual);
        const long row = m0 + wm * WTM + h * CH + st * RPS + rr;
        const long lrow = row < p.m_valid ? row : 0;
        if (has_residual<SIDE>(p)) {
            const T* rp = resp + lrow * p.ldr + col;
            if constexpr (BF) in.res_bf = *reinterpret_cast<const bf16x8*>(rp);
            else load8(reinterpret_cast<const float*>(rp), in.res_f);
        }
    }
    static __device__ __forceinline__ void step_load(const GemmParams& p, const float* stage, int h, int st, int m0, int wm, int rr, int cg,
                                                     long col, StepIn& in) {
        load8(stage + (st * RPS + rr) * LDW + cg * 8, in.v);
        rowbias_load(p, h, st, m0, wm, rr, col, in);
        residual_load(p, h, st, m0, wm, rr, col, in);
    }

    template <bool PRM_IN_REGS>
    static __device__ __forceinline__ void step_finish(const GemmParams& p, const StepIn& in, const float* prm, float* xch_base, int h, int st,
                                                       int step_id, int m0, int wm, int wave, int rr, int cg, int bcol, long col,
                                                       const float (&cb)[8], const float (&g0)[8], const float (&b0)[8],
                                                       const float (&g1)[8], const float (&b1)[8]) {
        const bool row_rb = p.rowbias != nullptr && (p.rowbias_idx != nullptr || p.ldrb != 0);
        const int crow = st * RPS + rr;
        const long row = m0 + wm * WTM + h * CH + crow;
        const bool live = row < p.m_valid;
        const long lrow = live ? row : 0;
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = in.v[k] + cb[k];
        if (row_rb) {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] += in.rb[k];
        }
        if (p.residual) {
            if constexpr (BF) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += (float)in.res_bf[e];
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] += in.res_f[k];
            }
        }
        if (p.act == 1) {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = silu_f(v[k]);
        }
        if (p.upd.kind != 0) apply_update(p.upd, v, lrow, col);
        if (p.out_f32 && live) store8<float>(p.out_f32 + (long)blockIdx.y * p.split_out_stride + row * p.ldo_f32 + col, v);
        if (p.out_op && live) store8<T>(reinterpret_cast<T*>(p.out_op) + row * p.ldo_op + col, v);
        float* const xch = xch_base + (step_id & 1) * (2 * XB);
        if (p.n_gn > 0) {
            T* const dst = reinterpret_cast<T*>(p.gn[0].out) + row * p.gn[0].ldo + col;
            if constexpr (PRM_IN_REGS) {
                gn_dispatch(p.gn[0].gs, v, g0, b0, dst, p.eps, live, xch, wave, rr, cg);
            } else {
                float gam[8], bet[8];   // from LDS at each use: sixteen registers less per output across the rolled loop
                load8(prm + BN + bcol, gam);
                load8(prm + 2 * BN + bcol, bet);
                gn_dispatch(p.gn[0].gs, v, gam, bet, dst, p.eps, live, xch, wave, rr, cg);
            }
        }
        if (p.n_gn > 1) {
            T* const dst = reinterpret_cast<T*>(p.gn[1].out) + row * p.gn[1].ldo + col;
            if constexpr (PRM_IN_REGS) {
                gn_dispatch(p.gn[1].gs, v, g1, b1, dst, p.eps, live, xch + XB, wave, rr, cg);
            } else {
                float gam[8], bet[8];
                load8(prm + 3 * BN + bcol, gam);
                load8(prm + 4 * BN + bcol, bet);
                gn_dispatch(p.gn[1].gs, v, gam, bet, dst, p.eps, live, xch + XB, wave, rr, cg);
            }
        }
    }

    // U consecutive steps finished together (U rows per lane): side inputs added row by row in the order of step_finish, then
    // every output of the U rows from one basic block.  For kernels whose GroupNorm groups never span two waves.
    template <int U, int SIDE = 0>
    static __device__ __forceinline__ void finish_multi(const GemmParams& p, const StepIn (&in)[U], const float* prm, int h, int st0,
                                                        int m0, int wm, int wave, int rr, int cg, int bcol, long col, const float (&cb)[8]) {
        const bool row_rb = has_row_rb<SIDE>(p);
        float v[U][8];
        long row[U];
        bool live[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            row[u] = m0 + wm * WTM + h * CH + (st0 + u) * RPS + rr;
            live[u] = row[u] < p.m_valid;
#pragma unroll
            for (int k = 0; k < 8; ++k) v[u][k] = in[u].v[k] + cb[k];
        }
        if (row_rb) {
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int k = 0; k < 8; ++k) v[u][k] += in[u].rb[k];
        }
        if (has_residual<SIDE>(p)) {
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int k = 0; k < 8; ++k) v[u][k] += BF ? (float)in[u].res_bf[k] : in[u].res_f[k];
        }
        // SIDE != 0 also says: no activation, no schedule update, no float32 output (the GEMMs inside the residual blocks)
        if (SIDE == 0 && p.act == 1) {
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int k = 0; k < 8; ++k) v[u][k] = silu_f(v[u][k]);
        }
        if (SIDE == 0 && p.upd.kind != 0) {
#pragma unroll
            for (int u = 0; u < U; ++u) apply_update(p.upd, v[u], live[u] ? row[u] : 0, col);
        }
        if (SIDE == 0 && p.out_f32) {
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (live[u]) store8<float>(p.out_f32 + (long)blockIdx.y * p.split_out_stride + row[u] * p.ldo_f32 + col, v[u]);
        }
        if (p.out_op) {
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (live[u]) store8<T>(reinterpret_cast<T*>(p.out_op) + row[u] * p.ldo_op + col, v[u]);
        }
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            if (g < p.n_gn) {
                float gam[8], bet[8];
                load8(prm + (1 + 2 * g) * BN + bcol, gam);
                load8(prm + (2 + 2 * g) * BN + bcol, bet);
                T* dst[U];
#pragma unroll
                for (int u = 0; u < U; ++u) dst[u] = reinterpret_cast<T*>(p.gn[g].out) + row[u] * p.gn[g].ldo + col;
                gn_dispatch_multi<U>(p.gn[g].gs, v, gam, bet, dst, p.eps, live, wave, rr, cg);
            }
        }
    }

    template <bool PRM_IN_REGS>
    static __device__ __forceinline__ void step(const GemmParams& p, const float* stage, const float* prm, float* xch_base, int h, int st,
                                                int step_id, int m0, int wm, int wave, int rr, int cg, int bcol, long col,
                                                const float (&cb)[8], const float (&g0)[8], const float (&b0)[8],
                                                const float (&g1)[8], const float (&b1)[8]) {
        StepIn in;
        step_load(p, stage, h, st, m0, wm, rr, cg, col, in);
        step_finish<PRM_IN_REGS>(p, in, prm, xch_base, h, st, step_id, m0, wm, wave, rr, cg, bcol, col, cb, g0, b0, g1, b1);
    }

    // smem_epi: NWAVE private staging regions (reusing the main loop's LDS); prm / xch: behind the main loop's LDS.
    // ROLLED: the chunk and step loops stay loops, so the step body exists once per GroupNorm variant and is re-executed
    // from the instruction cache -- for the one-workgroup-per-CU kernel, whose waves otherwise all fetch ~100 KB of
    // straight-line code cold.  Only the accumulator -> LDS copy needs compile-time register indices, hence the switch.
    // Unrolled (the kernels with 3-4 workgroups per CU): steps overlap each other's loads, parameters sit in registers.
    struct NoHook { __device__ __forceinline__ void operator()() const {} };
    // last_chunk: called once, in the last chunk between its two batches (the variants with SIDE != 0 only)
    template <bool ROLLED, int U = 1, int SIDE = 0, typename HOOK = NoHook>
    static __device__ __forceinline__ void run(const acc_t (&acc)[TM][TN], const GemmParams& p, int m0, int n0, int wm, int wn, int wave,
                                               int lane, float* smem_epi, float* prm, float* xch_base, HOOK last_chunk = HOOK()) {
        float* const stage = smem_epi + wave * (CH * LDW);
        const int ccol = BF ? (lane & 15) : (lane & 31);
        const int crow0 = BF ? 4 * (lane >> 4) : 4 * (lane >> 5);
        const int cg = lane % LPR, rr = lane / LPR;
        const int bcol = wn * WTN + cg * 8;      // column inside the block tile
        const long col = n0 + bcol;
        float cb[8], g0[8], b0[8], g1[8], b1[8];
        load8(prm + bcol, cb);
        if constexpr (!ROLLED) {
            load8(prm + BN + bcol, g0);
            load8(prm + 2 * BN + bcol, b0);
            load8(prm + 3 * BN + bcol, g1);
            load8(prm + 4 * BN + bcol, b1);
        }
        if constexpr (ROLLED) {
            int step_id = 0;
            StepIn ahead[U];
            // Chunk 0 is staged in front of the loop and chunk h + 1 at the end of iteration h (a wave's LDS accesses execute in
            // order): the registers of chunk 0's accumulators are then free throughout the loop.
            stage_chunk<0>(acc, stage, crow0, ccol);
#pragma unroll 1
            for (int h = 0; h < NCHUNK; ++h) {
#ifdef ADCDIFF_STAMPS
                if (U > 1 && NT == 256 && p.stamps && lane == 0 && blockIdx.x < 8192)
                    p.stamps[((long)blockIdx.x * 4 + wave) * 16 + 5 + 2 * h] = __builtin_amdgcn_s_memrealtime();
#endif
                if constexpr (U == 1) {
#pragma unroll 1
                    for (int st = 0; st < NSTEP; ++st, ++step_id)
                        step<false>(p, stage, prm, xch_base, h, st, step_id, m0, wm, wave, rr, cg, bcol, col, cb, g0, b0, g1, b1);
                } else if constexpr (NSTEP == 2 * U && SIDE != 0) {
                    // Two batches per chunk, the residual fetched one batch ahead (that of batch 0 of chunk h + 1 during batch 1
                    // of chunk h): a wave alone on its SIMD would otherwise sit through a memory round trip per batch.
                    // Past the last chunk the look-ahead re-reads the last chunk's rows (cached, unused).
                    if (h == 0) {
#pragma unroll
                        for (int u = 0; u < U; ++u) residual_load<SIDE>(p, 0, u, m0, wm, rr, col, ahead[u]);
                    }
                    StepIn in[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) residual_load<SIDE>(p, h, U + u, m0, wm, rr, col, in[u]);
#pragma unroll
                    for (int u = 0; u < U; ++u) load8(stage + (u * RPS + rr) * LDW + cg * 8, ahead[u].v);
                    finish_multi<U, SIDE>(p, ahead, prm, h, 0, m0, wm, wave, rr, cg, bcol, col, cb);
                    const int hn = h + 1 < NCHUNK ? h + 1 : h;
#pragma unroll
                    for (int u = 0; u < U; ++u) residual_load<SIDE>(p, hn, u, m0, wm, rr, col, ahead[u]);
                    if (h == NCHUNK - 1) last_chunk();   // behind the look-ahead: loads it issues are the youngest, no wait of this loop covers them
#pragma unroll
                    for (int u = 0; u < U; ++u) load8(stage + ((U + u) * RPS + rr) * LDW + cg * 8, in[u].v);
                    finish_multi<U, SIDE>(p, in, prm, h, U, m0, wm, wave, rr, cg, bcol, col, cb);
                    step_id += 2 * U;
                } else {
                    static_assert(NSTEP % U == 0, "steps per chunk must be a multiple of the load batch");
#pragma unroll 1
                    for (int st = 0; st < NSTEP; st += U, step_id += U) {
                        StepIn in[U];
#pragma unroll
                        for (int u = 0; u < U; ++u) step_load(p, stage, h, st + u, m0, wm, rr, cg, col, in[u]);
                        finish_multi<U>(p, in, prm, h, st, m0, wm, wave, rr, cg, bcol, col, cb);
                    }
                }
#ifdef ADCDIFF_STAMPS
                if (U > 1 && NT == 256 && p.stamps && lane == 0 && blockIdx.x < 8192)
                    p.stamps[((long)blockIdx.x * 4 + wave) * 16 + 6 + 2 * h] = __builtin_amdgcn_s_memrealtime();
#endif
                switch (h) {   // only the accumulator -> LDS copy needs compile-time register indices
                    case 0: stage_chunk<1>(acc, stage, crow0, ccol); break;
                    case 1: stage_chunk<2>(acc, stage, crow0, ccol); break;
                    case 2: stage_chunk<3>(acc, stage, crow0, ccol); break;
                    default: break;
                }
            }
        } else {
#pragma unroll
            for (int h = 0; h < NCHUNK; ++h) {
                if (h == 0) stage_chunk<0>(acc, stage, crow0, ccol);
                if (h == 1) stage_chunk<1>(acc, stage, crow0, ccol);
                if (h == 2) stage_chunk<2>(acc, stage, crow0, ccol);
                if (h == 3) stage_chunk<3>(acc, stage, crow0, ccol);
#pragma unroll
                for (int st = 0; st < NSTEP; ++st)
                    step<true>(p, stage, prm, xch_base, h, st, h * NSTEP + st, m0, wm, wave, rr, cg, bcol, col, cb, g0, b0, g1, b1);
            }
        }
    }
};

// NSTAGE = 1: the structure described at the top (one stage, latency hidden by the other resident workgroups).
// NSTAGE > 1: a ring of NSTAGE stages with the LDS-DMA of K tile t + NSTAGE - 1 issued right after the barrier of tile t, one
// barrier per tile -- for launches with so few workgroups (small batches, the float32 batch-4096 case) that nothing else on
// the CU hides the load latency of a tile (~1 us per K tile otherwise); one such workgroup per CU, so LDS is plentiful.
template <typename T, int BM, int BN, int NSTAGE>
__global__ __launch_bounds__(256, NSTAGE == 1 ? 3 : 2) void gemm_nt_kernel(const GemmParams p) {
    constexpr bool BF = sizeof(T) == 2;
    constexpr int A_BYTES = BM * 128;
    constexpr int STAGE_BYTES = (BM + BN) * 128;
    constexpr int WTM = BM / 2, WTN = BN / 2;  // wave tile (2 x 2 waves)
    constexpr int TS = BF ? 16 : 32;           // MFMA tile edge: 16x16x32 bf16 / 32x32x2 f32
    constexpr int TM = WTM / TS, TN = WTN / TS;
    constexpr int NACC = BF ? 4 : 16;
    constexpr int A_PIECES = BM / 8 / 4, B_PIECES = BN / 8 / 4;   // 1 KiB pieces (8 rows) per wave
    typedef Epilogue<T, 256, BN, WTM, WTN, TM, TN> Epi;
    constexpr int MAIN_BYTES = NSTAGE * STAGE_BYTES > Epi::STAGE_BYTES ? NSTAGE * STAGE_BYTES : Epi::STAGE_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* const prm = reinterpret_cast<float*>(smem + MAIN_BYTES);
    float* const xch = prm + 5 * BN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    int bid = blockIdx.x;
    const int nwg = gridDim.x;
    if ((nwg & 7) == 0) bid = (bid & 7) * (nwg >> 3) + (bid >> 3);
    const int m0 = (bid / p.ntn) * BM, n0 = (bid % p.ntn) * BN;
    if (m0 >= p.m_valid) return;   // a row tile of padding only (a batch of 32 in 128 padded rows): nothing to store
    Epi::stage_params(p, n0, tid, prm);

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
    const long ksb = (long)blockIdx.y * p.split_k_bytes;   // split-K: this workgroup's slice of the (single) K segment
    const char* const Wb = reinterpret_cast<const char*>(p.W) + ksb;

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

    // the MFMAs of one K tile staged at sa (A rows) / sb (W rows)
    auto compute = [&](const char* sa, const char* sb) {
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
    };

    int seg = 0, tt = 0;     // load cursor over the K segments
    const char* Ab = reinterpret_cast<const char*>(p.A[0]) + ksb;
    long lda_b = (long)p.lda[0] * (long)sizeof(T);
    if constexpr (NSTAGE == 1) {
        char* const sa = smem;
        char* const sb = smem + A_BYTES;
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
            compute(sa, sb);
            __syncthreads();
        }
    } else {
        const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
        const int nt = p.nt_total;
        int it = 0;          // tiles issued so far
        auto issue = [&](int slot) {
            if (tt == p.ktiles[seg]) {
                ++seg; tt = 0;
                if (tt == p.ktiles[seg]) { ++seg; }
                Ab = reinterpret_cast<const char*>(seg == 1 ? p.A[1] : p.A[2]);
                lda_b = (long)(seg == 1 ? p.lda[1] : p.lda[2]) * (long)sizeof(T);
            }
            const long ka = (long)tt * 128, kw = (long)it * 128;
            const unsigned base = lds0 + slot * STAGE_BYTES;
#pragma unroll
            for (int j = 0; j < A_PIECES; ++j) glds16_untracked(Ab + (long)a_row[j] * lda_b + ka + a_off[j], base + (wave * A_PIECES + j) * 1024);
#pragma unroll
            for (int j = 0; j < B_PIECES; ++j) glds16_untracked(Wb + b_off[j] + kw, base + A_BYTES + (wave * B_PIECES + j) * 1024);
            ++tt; ++it;
        };
        constexpr int AHEAD = NSTAGE - 1, P = A_PIECES + B_PIECES;
        static_assert((AHEAD - 1) * P <= 63 && AHEAD <= 8, "prefetch depth exceeds the vmcnt counter");
#pragma unroll
        for (int k = 0; k < AHEAD; ++k)
            if (k < nt) issue(k);
        int slot = 0, free_slot = AHEAD;   // free_slot: the slot of tile t-1 (initially the one the prologue left empty)
        for (int t = 0; t < nt; ++t) {
            // this wave's pieces of tile t have landed (up to AHEAD - 1 later tiles may stay in flight) ...
            const int later = nt - 1 - t;
            wait_tiles_in_flight<P>(later < AHEAD - 1 ? later : AHEAD - 1);
            // ... and after the barrier everyone's have, and nobody still reads tile t-1, whose slot takes tile t + AHEAD
            __builtin_amdgcn_s_barrier();
            if (t + AHEAD < nt) issue(free_slot);
            compute(smem + slot * STAGE_BYTES, smem + slot * STAGE_BYTES + A_BYTES);
            free_slot = slot;
            slot = slot == NSTAGE - 1 ? 0 : slot + 1;
        }
        __syncthreads();
    }

    Epi::template run<false>(acc, p, m0, n0, wm, wn, wave, lane, reinterpret_cast<float*>(smem), prm, xch);
}

// ---- 256 x 256 tile, 8 waves (2 x 4, 128 x 64 per wave), bf16, one workgroup (2 waves per SIMD) per CU ----------
// For the large layers (M, N multiples of 256).  Two 64 KiB LDS stages; LDS-DMA through glds16_untracked, vmcnt counted by hand.

// Phased main loop: two wave groups half a phase apart.
// A K tile (64) is four phases of 16 MFMAs per wave, one 64 x 32 quadrant of the wave's 128 x 64 tile each, in the
// order (0,0) (0,1) (1,1) (1,0).  A phase is: ds_read the fragments it needs that are not in registers yet, issue one
// 16 KiB region (two LDS-DMA pieces per wave) six regions ahead, wait until the region read NEXT phase has landed,
// barrier, 16 MFMAs, barrier.  Waves 4-7 (the lower 128 rows) run one barrier behind waves 0-3, and every SIMD holds
// one wave of each group: while one group issues its MFMAs the other does its reads and DMA issue, so the matrix pipe
// sees back-to-back MFMA clusters instead of idling through each wave's load section.
// Regions of a K tile, in issue order: A rows of quadrant row 0 (both groups), W rows of quadrant column 0 (all four
// wave columns), W rows of quadrant column 1, A rows of quadrant row 1.  Phase P (= 4 * tile + quadrant) reads regions
// P and P+1 (quadrant 0), P+1 (quadrants 1, 2) or nothing (quadrant 3: its W fragments are still in registers), and
// issues region P+6 into the buffer of region P-2, whose last reader ran at phase P-2 or earlier and retired those
// reads (lgkmcnt(0)) before a barrier every wave has passed since.  A region is read one phase after the counted
// vmcnt + barrier that every wave's pieces of it have passed, never in the same phase.
__global__ __launch_bounds__(512, 2) void gemm256_bf16_kernel(const GemmParams p) {
    typedef bf16_t T;
    constexpr int BM = 256, BN = 256, WTM = 128, WTN = 64, TM = 8, TN = 4, STAGE_BYTES = 65536;
    typedef Epilogue<T, 512, BN, WTM, WTN, TM, TN> Epi;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* const prm = reinterpret_cast<float*>(smem + 2 * STAGE_BYTES);
    float* const xch = prm + 5 * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    int bid = blockIdx.x;
    const int nwg = gridDim.x;
#ifdef ADCDIFF_STAMPS
    const long long st0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    if ((nwg & 7) == 0) bid = (bid & 7) * (nwg >> 3) + (bid >> 3);
    const int m0 = (bid / p.ntn) * BM, n0 = (bid % p.ntn) * BN;
    Epi::stage_params(p, n0, tid, prm);

    // LDS-DMA geometry.  This wave's two pieces (8 rows x 128 B) of a region: A regions -- piece k = 2 * wave + j lies
    // in wave-group row k >> 3, row group k & 7 of the 64 quadrant rows; W regions -- wave column k >> 2, row group
    // k & 3 of the 32 quadrant rows.  The swizzle chunk ^ ((row >> 1) & 7) sits in the source address.
    const int prow = lane >> 3;
    int arow[2], wrow[2];        // tile row of this lane's piece at quadrant 0
    long asw[2];                 // swizzled byte offset inside the 128-byte row
    const char* wsrc[2];         // W source at K tile 0, quadrant column 0
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int k = 2 * wave + j;
        arow[j] = (k >> 3) * 128 + (k & 7) * 8 + prow;
        wrow[j] = (k >> 2) * 64 + (k & 3) * 8 + prow;
        asw[j] = (long)(((lane & 7) ^ ((arow[j] >> 1) & 7)) * 16);
        wsrc[j] = reinterpret_cast<const char*>(p.W) + (long)(n0 + wrow[j]) * p.ldw * 2 + (long)(((lane & 7) ^ ((wrow[j] >> 1) & 7)) * 16) +
                  (long)blockIdx.y * p.split_k_bytes;
        asw[j] += (long)blockIdx.y * p.split_k_bytes;   // split-K slice of the (single) K segment
    }
    const int kt0 = p.ktiles[0], kt01 = p.ktiles[0] + p.ktiles[1];
    const int nt = p.nt_total, nreg = 4 * p.nt_total;

    // issue region G (tile G >> 2, kind G & 3: 0 = A quadrant row 0, 1 = W quadrant column 0, 2 = W column 1, 3 = A row 1)
#define ADCDIFF_P_ISSUE(G)                                                                                          \
    do {                                                                                                           \
        const int g_ = (G);                                                                                        \
        if (g_ < nreg) {                                                                                           \
            const int tile_ = g_ >> 2, kind_ = g_ & 3;                                                             \
            const unsigned sbase_ = lds0 + (tile_ & 1) * STAGE_BYTES;                                              \
            if (kind_ == 0 || kind_ == 3) {                                                                        \
                const int seg_ = tile_ < kt0 ? 0 : tile_ < kt01 ? 1 : 2;                                           \
                const int kk_ = tile_ - (seg_ == 0 ? 0 : seg_ == 1 ? kt0 : kt01);                                  \
                const char* ab_ = reinterpret_cast<const char*>(seg_ == 0 ? p.A[0] : seg_ == 1 ? p.A[1] : p.A[2]); \
                const long ld_ = (long)(seg_ == 0 ? p.lda[0] : seg_ == 1 ? p.lda[1] : p.lda[2]) * 2;               \
                const int q_ = kind_ == 3 ? 64 : 0;                                                                \
                _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                      \
                    glds16_untracked(ab_ + (long)(m0 + arow[j] + q_) * ld_ + (long)kk_ * 128 + asw[j],             \
                                     sbase_ + (((2 * wave + j) >> 3) * 128 + q_ + ((2 * wave + j) & 7) * 8) * 128); \
            } else {                                                                                               \
                const int q_ = kind_ == 2 ? 32 : 0;                                                                \
                _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                      \
                    glds16_untracked(wsrc[j] + (long)q_ * p.ldw * 2 + (long)tile_ * 128,                           \
                                     sbase_ + 32768 + (((2 * wave + j) >> 2) * 64 + q_ + ((2 * wave + j) & 3) * 8) * 128); \
            }                                                                                                      \
        }                                                                                                          \
    } while (0)
    // wait until this wave's pieces of region P+2 have landed: at most the regions P+3 .. min(P+6, last) stay in flight
#define ADCDIFF_P_WAIT(P)                                                                                           \
    do {                                                                                                           \
        const int rem_ = nreg - 3 - (P);                                                                           \
        if (rem_ >= 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                       \
        else if (rem_ == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                                       \
        else if (rem_ == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                       \
        else if (rem_ == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");                                       \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                      \
    } while (0)

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    const int frow = lane & 15, fchunk = lane >> 4;
    // fragment byte offsets inside a stage: A row r of the wave tile, k half ks; the XOR depends on frow only
    // ((row >> 1) & 7 with row = 16 * n + frow), so it is the same for every fragment row of the lane
    const int fx = (frow >> 1) & 7;
    const int a_off0 = (wm * WTM + frow) * 128 + ((fchunk ^ fx) << 4), a_off1 = (wm * WTM + frow) * 128 + (((4 + fchunk) ^ fx) << 4);
    const int b_off0 = 32768 + (wn * WTN + frow) * 128 + ((fchunk ^ fx) << 4), b_off1 = 32768 + (wn * WTN + frow) * 128 + (((4 + fchunk) ^ fx) << 4);

    uint4 fa[4][2], fb0[2][2], fb1[2][2];
#define ADCDIFF_P_READ_A(QI, ST)                                                                                    \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                \
        fa[i][0] = *reinterpret_cast<const uint4*>((ST) + a_off0 + ((QI) * 4 + i) * 2048);                         \
        fa[i][1] = *reinterpret_cast<const uint4*>((ST) + a_off1 + ((QI) * 4 + i) * 2048);                         \
    }
#define ADCDIFF_P_READ_B(FB, QJ, ST)                                                                                \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                                \
        FB[j][0] = *reinterpret_cast<const uint4*>((ST) + b_off0 + ((QJ) * 2 + j) * 2048);                         \
        FB[j][1] = *reinterpret_cast<const uint4*>((ST) + b_off1 + ((QJ) * 2 + j) * 2048);                         \
    }
#define ADCDIFF_P_MMA(FB, QI, QJ)                                                                                   \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                               \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                              \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                          \
                acc[(QI) * 4 + i][(QJ) * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                         \
                    __builtin_bit_cast(bf16x8, fa[i][ks]), __builtin_bit_cast(bf16x8, FB[j][ks]), acc[(QI) * 4 + i][(QJ) * 2 + j], 0, 0, 0)
    // the part of a phase after its reads: DMA issue, counted wait, barrier, MFMA cluster, barrier
#define ADCDIFF_P_BODY(P, FB, QI, QJ)                                                                               \
    do {                                                                                                           \
        ADCDIFF_P_ISSUE((P) + 6);                                                                                  \
        ADCDIFF_P_WAIT(P);                                                                                         \
        __builtin_amdgcn_s_barrier();                                                                              \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
        __builtin_amdgcn_s_setprio(1);                                                                             \
        ADCDIFF_P_MMA(FB, QI, QJ);                                                                                 \
        __builtin_amdgcn_s_setprio(0);                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
        __builtin_amdgcn_s_barrier();                                                                              \
    } while (0)

#pragma unroll
    for (int g = 0; g < 6; ++g) ADCDIFF_P_ISSUE(g);
    ADCDIFF_P_WAIT(-1);
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();   // the lower-rows group runs one barrier behind from here on
#ifdef ADCDIFF_STAMPS
    const long long st1 = __builtin_amdgcn_s_memtime();
#endif
    for (int t = 0; t < nt; ++t) {
        const char* st = smem + (t & 1) * STAGE_BYTES;
        const int P = 4 * t;
        ADCDIFF_P_READ_B(fb0, 0, st);
        __builtin_amdgcn_sched_barrier(0);
        ADCDIFF_P_READ_A(0, st);
        ADCDIFF_P_BODY(P, fb0, 0, 0);
        ADCDIFF_P_READ_B(fb1, 1, st);
        ADCDIFF_P_BODY(P + 1, fb1, 0, 1);
        ADCDIFF_P_READ_A(1, st);
        ADCDIFF_P_BODY(P + 2, fb1, 1, 1);
        ADCDIFF_P_BODY(P + 3, fb0, 1, 0);
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();
#undef ADCDIFF_P_BODY
#undef ADCDIFF_P_MMA
#undef ADCDIFF_P_READ_B
#undef ADCDIFF_P_READ_A
#undef ADCDIFF_P_WAIT
#undef ADCDIFF_P_ISSUE
    __syncthreads();
#ifdef ADCDIFF_STAMPS
    const long long st2 = __builtin_amdgcn_s_memtime();
#endif
    Epi::template run<true>(acc, p, m0, n0, wm, wn, wave, lane, reinterpret_cast<float*>(smem), prm, xch);
#ifdef ADCDIFF_STAMPS
    const long long st3 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long st4 = __builtin_amdgcn_s_memtime(), rt4 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0 && blockIdx.x < 8192) {
        long long* o = p.stamps + ((long)blockIdx.x * 8 + wave) * 8;
        o[0] = st0; o[1] = st1; o[2] = st2; o[3] = st3; o[4] = st4; o[5] = rt0; o[6] = rt4; o[7] = 0;
    }
#endif
}

// Ring main loop: K is consumed in steps of 32 (ONE v_mfma_f32_16x16x32_bf16 per 16 x 16 output tile and step, so all 32
// MFMAs of a wave's step are independent).  A step's operands -- 256 rows of A and 256 rows of W, 64 bytes each -- form one
// 32 KiB LDS region (rows of 64 bytes, four 16-byte chunks, physical chunk = logical chunk ^ g((row >> 2) & 3) with
// g = {0, 2, 3, 1}: the four 16-lane groups of a ds_read_b128 then hit sixteen distinct 16-byte slots of a bank row); four
// regions form a ring.  During step R a wave issues the MFMAs of step R from registers, reads the fragments of step R + 1
// (region R + 1) into its second register set behind every second MFMA, and issues its four LDS-DMA pieces of region
// R + 4 into the buffer of region R, whose reads were retired (lgkmcnt(0)) before the barrier that ended step R - 1.  At
// the end of the step it retires its reads, waits until its own pieces of region R + 2 have landed (counted vmcnt: regions
// R + 3 and R + 4 stay in flight) and meets the other waves at the ONE barrier of the step; region R + 2 is first read in
// step R + 1, i.e. after that barrier.  Waves 0-3 load the A rows, waves 4-7 the W rows (four 16-row pieces each per region).
// The K order seen by every accumulator (32-wide steps in order, k = 8 * (lane >> 4) + j inside an MFMA) is the one of
// gemm_nt_kernel, so results are bit-identical between the tilings.
__device__ __forceinline__ int ring_swz(int row) { return (0x78 >> (2 * ((row >> 2) & 3))) & 3; }

__global__ __launch_bounds__(512, 2) void gemm256r_bf16_kernel(const GemmParams p) {
    typedef bf16_t T;
    constexpr int BN = 256, WTM = 128, WTN = 64, TM = 8, TN = 4, REGION = 32768, PPW = 4;
    typedef Epilogue<T, 512, BN, WTM, WTN, TM, TN> Epi;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* const prm = reinterpret_cast<float*>(smem + 4 * REGION);
    float* const xch = prm + 5 * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    int bid = blockIdx.x;
    const int nwg = gridDim.x;
#ifdef ADCDIFF_STAMPS
    const long long st0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    if ((nwg & 7) == 0) bid = (bid & 7) * (nwg >> 3) + (bid >> 3);
    const int m0 = (bid / p.ntn) * 256, n0 = (bid % p.ntn) * 256;
    Epi::stage_params(p, n0, tid, prm);

    // LDS-DMA geometry.  A piece is 16 rows x 64 bytes (1 KiB): lane l fills row l >> 2, chunk position l & 3, which holds
    // the logical chunk (l & 3) ^ g(row) -- the swizzle sits in the SOURCE address, the LDS image stays lane-linear.
    // g depends on bits 2..3 of the row only, i.e. on the lane, not on the piece.
    const bool loads_a = wave < 4;
    const int prow = lane >> 2;
    const int piece0 = (wave & 3) * PPW;      // this wave's pieces are rows 16 * piece0 .. 16 * piece0 + 63 of its operand
    const unsigned chunk_b = (unsigned)(((lane & 3) ^ ring_swz(prow)) * 16);
    const long ksplit_off = (long)blockIdx.y * p.split_k_bytes;   // split-K slice of the (single) K segment
    const int nR = 2 * p.nt_total;
    // K segments in regions: W is one run of K; the A loaders switch operand at rs1 / rs2
    const int rs1 = loads_a ? 2 * p.ktiles[0] : 0x7fffffff, rs2 = loads_a ? 2 * (p.ktiles[0] + p.ktiles[1]) : 0x7fffffff;
    long ld2[3];              // row pitch in bytes
    const char* sbase[3];     // piece 0, lane row 0, region 0 of the segment
    unsigned voff[3];         // per-lane byte offset inside a piece
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        ld2[s] = loads_a ? (long)p.lda[s] * 2 : (long)p.ldw * 2;
        sbase[s] = loads_a ? reinterpret_cast<const char*>(p.A[s]) + (long)(m0 + 16 * piece0) * ld2[s] + (s == 0 ? ksplit_off : 0)
                           : reinterpret_cast<const char*>(p.W) + (long)(n0 + 16 * piece0) * ld2[s] + ksplit_off;
        voff[s] = (unsigned)prow * (unsigned)ld2[s] + chunk_b;
    }
    const unsigned dst0 = lds0 + (loads_a ? 0 : 16384) + piece0 * 1024;
    // piece J of region G
#define ADCDIFF_R_ISSUE1(G, J)                                                                                      \
    do {                                                                                                           \
        const int g_ = (G);                                                                                        \
        if (g_ < nR) {                                                                                             \
            const int seg_ = g_ < rs1 ? 0 : g_ < rs2 ? 1 : 2;                                                      \
            const int kk_ = g_ - (seg_ == 0 ? 0 : seg_ == 1 ? rs1 : rs2);                                          \
            const char* b_ = (seg_ == 0 ? sbase[0] : seg_ == 1 ? sbase[1] : sbase[2]) + (long)kk_ * 64 +           \
                             (long)(16 * (J)) * (seg_ == 0 ? ld2[0] : seg_ == 1 ? ld2[1] : ld2[2]);                \
            glds16_untracked(b_ + (seg_ == 0 ? voff[0] : seg_ == 1 ? voff[1] : voff[2]), dst0 + (g_ & 3) * REGION + (J) * 1024); \
        }                                                                                                          \
    } while (0)
    // at most N of the regions this wave has issued (PPW pieces each) may still be in flight
#define ADCDIFF_R_INFLIGHT(N)                                                                                       \
    do {                                                                                                           \
        const int n_ = (N);                                                                                        \
        if (n_ >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");                                             \
        else if (n_ == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                         \
        else if (n_ == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                         \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                      \
    } while (0)

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    // fragment addressing: lane holds row (l & 15) and k chunk (l >> 4) of a 16-row fragment
    const int frow = lane & 15, fchunk = lane >> 4;
    const int a_off = (wm * WTM + frow) * 64 + ((fchunk ^ ring_swz(frow)) << 4);
    const int b_off = 16384 + (wn * WTN + frow) * 64 + ((fchunk ^ ring_swz(frow)) << 4);
    uint4 fa[2][TM], fb[2][TN];   // two register sets: one feeds the MFMAs of the step, the other receives the next step's reads
#define ADCDIFF_R_READ(S, RB, X)                                                                                    \
    do {                                                                                                           \
        if ((X) < TN) fb[S][(X) < TN ? (X) : 0] = *reinterpret_cast<const uint4*>((RB) + b_off + ((X) < TN ? (X) : 0) * 1024); \
        else fa[S][(X) >= TN ? (X) - TN : 0] = *reinterpret_cast<const uint4*>((RB) + a_off + ((X) >= TN ? (X) - TN : 0) * 1024); \
    } while (0)
    // One step.  S = register set holding step R's fragments.  32 MFMAs; read X of the 12 fragment reads of step R + 1 goes
    // out behind MFMA 2 X + 1, DMA piece J of region R + 4 behind MFMA 8 J + 4.
#define ADCDIFF_R_STEP(R_, S)                                                                                       \
    do {                                                                                                           \
        const int r_ = (R_);                                                                                       \
        const char* rb_ = smem + ((r_ + 1) & 3) * REGION;                                                          \
        const bool more_ = r_ + 1 < nR;                                                                            \
        _Pragma("unroll") for (int x = 0; x < TM * TN; ++x) {                                                      \
            const int i = x / TN, j = (x / TN) & 1 ? TN - 1 - x % TN : x % TN;                                     \
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[S][i]),              \
                                                                __builtin_bit_cast(bf16x8, fb[S][j]), acc[i][j], 0, 0, 0); \
            if (more_ && (x & 1) && x / 2 < TM + TN) ADCDIFF_R_READ((S) ^ 1, rb_, x / 2);                           \
            if (x % 8 == 4) ADCDIFF_R_ISSUE1(r_ + 4, x / 8);                                                       \
            if (x & 1) __builtin_amdgcn_sched_barrier(0);                                                          \
        }                                                                                                          \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                         \
        ADCDIFF_R_INFLIGHT(nR - 3 - r_ < 2 ? nR - 3 - r_ : 2);   /* regions r + 3, r + 4 may stay in flight */      \
        __builtin_amdgcn_s_barrier();                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
    } while (0)

    // prologue: four regions under way; region 0 landed -> its fragments into set 0; region 1 landed
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int j = 0; j < PPW; ++j) ADCDIFF_R_ISSUE1(g, j);
    }
    ADCDIFF_R_INFLIGHT(nR - 1 < 3 ? nR - 1 : 3);
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int x = 0; x < TM + TN; ++x) ADCDIFF_R_READ(0, smem, x);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    ADCDIFF_R_INFLIGHT(nR - 2 < 2 ? nR - 2 : 2);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#ifdef ADCDIFF_STAMPS
    const long long st1 = __builtin_amdgcn_s_memtime();
#endif
    for (int r = 0; r < nR; r += 2) {   // nR is even: a K tile of 64 is two steps
        ADCDIFF_R_STEP(r, 0);
        ADCDIFF_R_STEP(r + 1, 1);
    }
#undef ADCDIFF_R_STEP
#undef ADCDIFF_R_READ
#undef ADCDIFF_R_INFLIGHT
#undef ADCDIFF_R_ISSUE1
    __syncthreads();
#ifdef ADCDIFF_STAMPS
    const long long st2 = __builtin_amdgcn_s_memtime();
#endif
    Epi::template run<true>(acc, p, m0, n0, wm, wn, wave, lane, reinterpret_cast<float*>(smem), prm, xch);
#ifdef ADCDIFF_STAMPS
    const long long st3 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long st4 = __builtin_amdgcn_s_memtime(), rt4 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0 && blockIdx.x < 8192) {
        long long* o = p.stamps + ((long)blockIdx.x * 8 + wave) * 8;
        o[0] = st0; o[1] = st1; o[2] = st2; o[3] = st3; o[4] = st4; o[5] = rt0; o[6] = rt4; o[7] = 0;
    }
#endif
}

// ---- 256 x 256 tile, 4 waves (2 x 2, 128 x 128 per wave), bf16, ONE wave per SIMD, accumulators in a[0:255] ----------
// The main loop is hand-allocated assembly (w4_mainloop.inc, generated by gen_w4_mainloop.py, which documents the register
// map, the LDS image, the phase protocol and the L2 prefetch): hipcc spills hundreds of registers on a 512-register loop.
// This wrapper computes the loop's inputs, pins them and the 64 accumulators to the registers the loop names, and runs the
// shared epilogue on the result.  K order per accumulator = the other kernels' (32-wide steps in order), so results are
// bit-identical between the tilings.  Not for split-K launches.
typedef int i32x4 __attribute__((ext_vector_type(4)));
#include "w4_mainloop.inc"
#ifdef ADCDIFF_STAMPS
#include "w4_mainloop_stamps.inc"   // python3 gen_w4_mainloop.py --stamps w4_mainloop_stamps.inc
#endif

template <int EPI_U, int SIDE>
__global__ __launch_bounds__(256, 1) void gemm256w_bf16_kernel(const GemmParams p) {
    typedef bf16_t T;
    constexpr int BN = 256, WTM = 128, WTN = 128, TM = 8, TN = 8, STAGE_BYTES = 65536;
    typedef Epilogue<T, 256, BN, WTM, WTN, TM, TN> Epi;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* const prm = reinterpret_cast<float*>(smem + 2 * STAGE_BYTES);
    float* const xch = prm + 5 * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    int bid = blockIdx.x;
    const int nwg = gridDim.x;
#ifdef ADCDIFF_STAMPS
    const long long rt0 = __builtin_amdgcn_s_memrealtime(), ct0 = __builtin_amdgcn_s_memtime();
#endif
    if ((nwg & 7) == 0) bid = (bid & 7) * (nwg >> 3) + (bid >> 3);
    const int nidx = bid % p.ntn;
    const int m0 = (bid / p.ntn) * 256, n0 = nidx * BN;
    typename Epi::ParamRegs pv;
    Epi::load_params(p, n0, tid, pv);

    // Loader role: waves 0, 1 stage the A rows 128 (wave & 1) .. + 127 of the tile, waves 2, 3 the W rows.  Per K segment
    // {base, row pitch, K tiles, addressable bytes}; W is one segment.  The descriptor's size stops any stray access.
    const bool loads_a = wave < 2;
    const int half = wave & 1;
    int seg_lo[3], seg_hi[3], seg_ld[3], seg_kt[3], seg_nrec[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const unsigned ld = loads_a ? (unsigned)p.lda[s] * 2u : (unsigned)p.ldw * 2u;
        const int kt = loads_a ? p.ktiles[s] : s == 0 ? p.nt_total : 0;
        const unsigned long long base = loads_a ? (unsigned long long)p.A[s] + (unsigned long long)(m0 + half * 128) * ld
                                                : (unsigned long long)p.W + (unsigned long long)(n0 + half * 128) * ld;
        seg_lo[s] = __builtin_amdgcn_readfirstlane((int)(unsigned)base);
        seg_hi[s] = __builtin_amdgcn_readfirstlane((int)((unsigned)(base >> 32) & 0xffffu));
        seg_ld[s] = __builtin_amdgcn_readfirstlane((int)ld);
        seg_kt[s] = __builtin_amdgcn_readfirstlane(kt);
        seg_nrec[s] = __builtin_amdgcn_readfirstlane(kt > 0 ? (int)(127u * ld + (unsigned)kt * 128u) : 0);
    }
    const int s_nt = __builtin_amdgcn_readfirstlane(p.nt_total);
    const int s_dst = __builtin_amdgcn_readfirstlane((int)(lds0 + (loads_a ? 0u : 32768u) + (unsigned)half * 16384u));
    // fragment read offsets inside a stage: row 16 n + frow of the wave's block, k chunk 4 h + fchunk, chunk ^ ((row >> 1) & 7)
    const int frow = lane & 15, fchunk = lane >> 4, fx = (frow >> 1) & 7;
    const int v_aoff0 = (int)lds0 + (wm * WTM + frow) * 128 + ((fchunk ^ fx) << 4);
    const int v_aoff1 = (int)lds0 + (wm * WTM + frow) * 128 + (((4 + fchunk) ^ fx) << 4);
    const int v_boff0 = (int)lds0 + 32768 + (wn * WTN + frow) * 128 + ((fchunk ^ fx) << 4);
    const int v_boff1 = (int)lds0 + 32768 + (wn * WTN + frow) * 128 + (((4 + fchunk) ^ fx) << 4);
    // LDS-DMA source offsets: piece j of the wave = its rows 8 j .. 8 j + 7, lane -> row 8 j + (lane >> 3), chunk position
    // lane & 7, which holds chunk (lane & 7) ^ ((row >> 1) & 7) = (lane & 7) ^ ((4 j + (lane >> 4)) & 7): two offsets, by the parity of j
    // Prefetch offsets: the ntn workgroups of a tile row split its 256 A rows.  Four or more workgroups: a share of 256 / ntn
    // <= 64 rows lies in ONE loader wave's half; its lanes point at the share's lines, 64 / share consecutive K tiles deep,
    // every other wave at a single line (which costs nothing).  Two workgroups: each A loader takes 64 rows of its own half
    // (workgroup 0 the first 64, workgroup 1 the second).  One workgroup: the first 64 rows of each half (half the lines).
    const int share = p.ntn >= 4 ? 256 / p.ntn : 64;
    const int prow0 = p.ntn >= 4 ? nidx * share : half * 128 + (p.ntn == 2 ? nidx * 64 : 0);
    const bool pf_wave = loads_a && (prow0 >> 7) == half;
    const int pf_row = (prow0 & 127) + lane % share;
    const int pf_k = (lane / share) * 128;
    int v_src0[3], v_src1[3], v_pf[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const unsigned ld = (unsigned)seg_ld[s];
        v_src0[s] = (int)((unsigned)(lane >> 3) * ld + (unsigned)(((lane & 7) ^ (lane >> 4)) * 16));
        v_src1[s] = (int)((unsigned)(lane >> 3) * ld + (unsigned)(((lane & 7) ^ (4 + (lane >> 4))) * 16));
        v_pf[s] = pf_wave ? (int)((unsigned)pf_row * ld + (unsigned)pf_k) : 0;
    }
    // The W rows of a column tile are fetched by the four row tiles that an XCD runs side by side: the W loaders' prefetch load
    // (otherwise a single line) touches this workgroup's quarter of them, 64 rows of one K tile
    if (p.pf_w && !loads_a) {
        const int wrow0 = ((bid / p.ntn) & 3) * 64;
        if ((wrow0 >> 7) == half) v_pf[0] = (int)((unsigned)((wrow0 & 127) + lane) * (unsigned)seg_ld[0]);
    }

#ifdef ADCDIFF_STAMPS
    const long long rta = __builtin_amdgcn_s_memrealtime();
#endif
    f32x4 acc[TM][TN];
#ifdef ADCDIFF_STAMPS
    int tile_stamp;
    asm volatile(ADCDIFF_W4S_MAINLOOP_ASM
                 : ADCDIFF_W4_OUTS(acc), "={v147}"(tile_stamp)
#else
    asm volatile(ADCDIFF_W4_MAINLOOP_ASM
                 : ADCDIFF_W4_OUTS(acc)
#endif
                 : "{s40}"(seg_lo[0]), "{s41}"(seg_hi[0]), "{s42}"(seg_ld[0]), "{s43}"(seg_kt[0]), "{s44}"(seg_nrec[0]),
                   "{s45}"(seg_lo[1]), "{s46}"(seg_hi[1]), "{s47}"(seg_ld[1]), "{s48}"(seg_kt[1]), "{s49}"(seg_nrec[1]),
                   "{s50}"(seg_lo[2]), "{s51}"(seg_hi[2]), "{s52}"(seg_ld[2]), "{s53}"(seg_kt[2]), "{s54}"(seg_nrec[2]),
                   "{s55}"(s_nt), "{s56}"(s_dst),
                   "{v130}"(v_aoff0), "{v131}"(v_boff0), "{v132}"(v_aoff1), "{v133}"(v_boff1),
                   "{v136}"(v_src0[0]), "{v137}"(v_src1[0]), "{v138}"(v_src0[1]), "{v139}"(v_src1[1]), "{v140}"(v_src0[2]), "{v141}"(v_src1[2]),
                   "{v144}"(v_pf[0]), "{v145}"(v_pf[1]), "{v146}"(v_pf[2])
#ifdef ADCDIFF_STAMPS
                 : ADCDIFF_W4_CLOBBERS, "s88", "s89", "s90", "v148", "v149");
    if (wave == 0 && blockIdx.x < 8192) p.tile_stamps[(long)blockIdx.x * 64 + lane] = tile_stamp;
    const long long rt1 = __builtin_amdgcn_s_memrealtime(), ct1 = __builtin_amdgcn_s_memtime();
#else
                 : ADCDIFF_W4_CLOBBERS);
#endif
    Epi::store_params(p, tid, prm, pv);
    __syncthreads();
#ifdef ADCDIFF_STAMPS
    const long long rtb = __builtin_amdgcn_s_memrealtime();
#endif
    // A workgroup's first two K tiles come from beyond L2 (the loop's prefetch starts at tile 2): issuing their LDS-DMA pieces
    // takes 4.6 us at N = K = 2048 (rows 4 KB apart), against 1.3 us per tile in the loop.  So a workgroup touches, near the end of
    // its epilogue, the lines that the first two K tiles of its successor on this XCD will read (workgroup blockIdx + pf_next, i.e.
    // 1/8 of pf_next tiles further on: the same columns and the tile rows behind when the tiles per row divide that).  The ntn
    // workgroups of a tile row split the A rows, the tile rows that share a column of W its rows; one dword per 128-byte line by
    // LDS-DMA into a part of the main loop's LDS that the epilogue does not use (no register waits for the data, and the
    // compiler, which does not see these loads, puts no wait in their way); lanes without a line point past the descriptor's
    // range (no access).  Late in the epilogue, because its stores sweep the L2: touched at the start, the lines were gone again.
    auto touch_successor = [&]() {
        const int step = p.pf_next >> 3;   // tiles between a workgroup and its successor
        if (step > 0 && (nwg & 7) == 0 && (int)blockIdx.x + p.pf_next < nwg && step % p.ntn == 0 && step / p.ntn <= 256 && p.ktiles[0] >= 2) {
            const int R = step / p.ntn;                       // tile rows that share a column of W at a time
            const int a_rows = 256 / p.ntn, w_rows = 256 / R;
            const int mt = bid / p.ntn;
            const unsigned lda_b = (unsigned)p.lda[0] * 2u, ldw_b = (unsigned)p.ldw * 2u;
            auto make_srd = [](unsigned long long base, unsigned bytes) {
                return i32x4{__builtin_amdgcn_readfirstlane((int)(unsigned)base), __builtin_amdgcn_readfirstlane((int)((unsigned)(base >> 32) & 0xffffu)),
                             __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000};
            };
            const i32x4 srd_a = make_srd((unsigned long long)p.A[0] + (unsigned long long)((long)(mt + R) * 256 + nidx * a_rows) * lda_b,
                                         (unsigned)(a_rows - 1) * lda_b + 256u);
            const i32x4 srd_w = make_srd((unsigned long long)p.W + (unsigned long long)((long)n0 + (mt % R) * w_rows) * ldw_b,
                                         (unsigned)(w_rows - 1) * ldw_b + 256u);
            const unsigned sink = lds0 + 98304u + (unsigned)wave * 256u;   // behind the epilogue's staging regions
            int va[2], vw[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int i = tid + 256 * j;
                va[j] = i < 2 * a_rows ? (int)((unsigned)(i >> 1) * lda_b + (unsigned)(i & 1) * 128u) : 0x7ffffff0;
                vw[j] = i < 2 * w_rows ? (int)((unsigned)(i >> 1) * ldw_b + (unsigned)(i & 1) * 128u) : 0x7ffffff0;
            }
            int m0_keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %7\n\ts_nop 0\n\t"
                         "buffer_load_dword %1, %5, 0 offen lds\n\tbuffer_load_dword %2, %5, 0 offen lds\n\t"
                         "buffer_load_dword %3, %6, 0 offen lds\n\tbuffer_load_dword %4, %6, 0 offen lds\n\t"
                         "s_mov_b32 m0, %0"
                         : "=&s"(m0_keep) : "v"(va[0]), "v"(va[1]), "v"(vw[0]), "v"(vw[1]), "s"(srd_a), "s"(srd_w), "s"(sink));
        }
    };
    Epi::template run<true, EPI_U, SIDE>(acc, p, m0, n0, wm, wn, wave, lane, reinterpret_cast<float*>(smem), prm, xch, touch_successor);
    if constexpr (SIDE != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA in flight when the workgroup's LDS is handed on
#ifdef ADCDIFF_STAMPS
    const long long rt2 = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long rt3 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0 && blockIdx.x < 8192) {
        long long* o = p.stamps + ((long)blockIdx.x * 4 + wave) * 16;
        o[0] = rt0; o[1] = rt1; o[2] = rt2; o[3] = rt3; o[4] = rtb; o[13] = ct1 - ct0; o[14] = rta;
    }
#endif
}

// pipelined = the ring main loop (gemm256r_bf16_kernel), else the two-group phased loop
static int launch256(const GemmParams& p, bool pipelined, hipStream_t stream) {
    constexpr int smem = 2 * 65536 + Epilogue<bf16_t, 512, 256, 128, 64, 8, 4>::EXTRA_BYTES;
    // once per process, thread-safe (initialisation of a function-local static)
    static const hipError_t attr_err = [] {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_bf16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256r_bf16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        return e;
    }();
    if (attr_err != hipSuccess) return fail(ADCDIFF_E_LAUNCH, "gemm: hipFuncSetAttribute: %s", hipGetErrorString(attr_err));
    GemmParams q = p;
    q.ntn = p.N / 256;
#ifdef ADCDIFF_STAMPS
    static long long* dbuf = nullptr;
    if (!dbuf) (void)hipMalloc(&dbuf, (size_t)8192 * 64 * 8);
    q.stamps = dbuf;
#endif
    const int grid = (p.M / 256) * q.ntn;
    if (pipelined) hipLaunchKernelGGL(gemm256r_bf16_kernel, dim3(grid, p.ksplit), dim3(512), smem, stream, q);
    else hipLaunchKernelGGL(gemm256_bf16_kernel, dim3(grid, p.ksplit), dim3(512), smem, stream, q);
#ifdef ADCDIFF_STAMPS
    {
        static int calls = 0;
        if (++calls == 10 && grid <= 8192) {
            (void)hipDeviceSynchronize();
            std::vector<long long> h((size_t)grid * 64);
            (void)hipMemcpy(h.data(), q.stamps, h.size() * 8, hipMemcpyDeviceToHost);
            double pro = 0, mainl = 0, epi = 0, drain = 0; long long rmin = 1LL << 62, rmax = 0;
            double pw[8] = {0}, ew[8] = {0};
            for (int b = 0; b < grid; ++b)
                for (int w = 0; w < 8; ++w) {
                    const long long* o = &h[((size_t)b * 8 + w) * 8];
                    pro += o[1] - o[0]; mainl += o[2] - o[1]; epi += o[3] - o[2]; drain += o[4] - o[3];
                    pw[w] += o[1] - o[0]; ew[w] += o[3] - o[2];
                    if (o[5] < rmin) rmin = o[5];
                    if (o[6] > rmax) rmax = o[6];
                }
            const double n = (double)grid * 8;
            fprintf(stderr, "[stamps] grid %d: prologue %.0f  main %.0f  epilogue %.0f  drain %.0f cycles (mean per wave); kernel span %.1f us\n",
                    grid, pro / n, mainl / n, epi / n, drain / n, (rmax - rmin) / 100.0);
        }
    }
#endif
    return check_launch("gemm256");
}

// CUs of the current device = workgroups of the four-wave kernel resident at a time (one per CU); 0 if the query fails
static int compute_units() {
    static std::atomic<int> cache[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    int v = cache[dev].load(std::memory_order_relaxed);
    if (v == 0 && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
        cache[dev].store(v, std::memory_order_relaxed);
    return v > 0 ? v : 0;
}

// SIDE (see Epilogue::rowbias_load): 0 serves every launch; 1 / 2 know at compile time that there is no per-row bias and no /
// a residual (and no activation, schedule update or float32 output), which lets the epilogue fetch the residual one batch ahead
template <int EPI_U, int SIDE = 0>
static int launch256w(const GemmParams& p, hipStream_t stream) {
    constexpr int smem = 2 * 65536 + Epilogue<bf16_t, 256, 256, 128, 128, 8, 8>::EXTRA_BYTES;
    static const hipError_t attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256w_bf16_kernel<EPI_U, SIDE>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (attr_err != hipSuccess) return fail(ADCDIFF_E_LAUNCH, "gemm: hipFuncSetAttribute: %s", hipGetErrorString(attr_err));
    GemmParams q = p;
    q.ntn = p.N / 256;
    q.pf_w = (int)debug_value(DBG_GEMM_PF_W);
    q.pf_next = debug_value(DBG_GEMM_PF_NEXT) ? compute_units() : 0;
    const int grid = (p.M / 256) * q.ntn;
#ifdef ADCDIFF_STAMPS
    static long long* dbuf = nullptr;
    static int* tbuf = nullptr;
    if (!dbuf) (void)hipMalloc(&dbuf, (size_t)8192 * 64 * 8);
    if (!tbuf) (void)hipMalloc(&tbuf, (size_t)8192 * 64 * 4);
    q.stamps = dbuf;
    q.tile_stamps = tbuf;
#endif
    hipLaunchKernelGGL((gemm256w_bf16_kernel<EPI_U, SIDE>), dim3(grid), dim3(256), smem, stream, q);
#ifdef ADCDIFF_STAMPS
    {
        // measurement build: the 10th launch is dissected (100 MHz constant clock, 10 ns units): per round of 256 workgroups (by
        // start order) the mean time from kernel start to the workgroup's start, its main phase (set-up + prologue + K loop), its
        // epilogue, and the drain of its stores
        static int calls = 0;
        if (++calls % 10 == 0 && grid <= 8192) {
            (void)hipDeviceSynchronize();
            std::vector<long long> h((size_t)grid * 64);
            (void)hipMemcpy(h.data(), q.stamps, h.size() * 8, hipMemcpyDeviceToHost);
            std::vector<std::pair<long long, int>> order;
            long long t0 = 1LL << 62, t_end = 0;
            for (int b = 0; b < grid; ++b) { const long long* o = &h[(size_t)b * 64]; order.push_back({o[0], b}); if (o[0] < t0) t0 = o[0]; if (o[3] > t_end) t_end = o[3]; }
            std::sort(order.begin(), order.end());
            fprintf(stderr, "[stamps w4] N %d K tiles %d grid %d: kernel span %.1f us\n", p.N, p.nt_total, grid, (t_end - t0) / 100.0);
            for (int r = 0; r * 256 < grid; ++r) {
                double st = 0, mn = 0, ep = 0, dr = 0; int n = 0;
                for (int k = r * 256; k < (r + 1) * 256 && k < grid; ++k, ++n) {
                    const long long* o = &h[(size_t)order[k].second * 64];
                    st += o[0] - t0; mn += o[1] - o[0]; ep += o[2] - o[1]; dr += o[3] - o[2];
                }
                fprintf(stderr, "[stamps w4]   round %d: start +%.1f us, main %.2f us, epilogue %.2f us, drain %.2f us\n", r, st / n / 100.0, mn / n / 100.0, ep / n / 100.0, dr / n / 100.0);
                if (r == 1) {   // inside the epilogue, wave 0 of the round's workgroups: barrier, then per 32-row chunk (staging, steps)
                    double seg[9] = {0}, clk = 0, setup = 0;
                    for (int k = r * 256; k < (r + 1) * 256 && k < grid; ++k) {
                        const long long* o = &h[(size_t)order[k].second * 64];
                        clk += (double)o[13] / (double)(o[1] - o[0]);   // shader-clock ticks per 10 ns over the main phase
                        setup += o[14] - o[0];
                        seg[0] += o[4] - o[1];
                        long long prev = o[4];
                        for (int i = 0; i < 8; ++i) { seg[1 + i] += o[5 + i] - prev; prev = o[5 + i]; }
                    }
                    fprintf(stderr, "[stamps w4]     set-up in front of the main loop %.2f us; s_memtime rate over the main phase %.0f MHz; barrier %.2f us; chunks (stage, steps):", setup / n / 100.0, clk / n * 100.0, seg[0] / n / 100.0);
                    for (int i = 0; i < 4; ++i) fprintf(stderr, " (%.2f, %.2f)", seg[1 + 2 * i] / n / 100.0, seg[2 + 2 * i] / n / 100.0);
                    fprintf(stderr, " us\n");
                    // inside the main loop (wave 0): lane 0 entry, 1 pieces of tiles 0 and 1 issued, 2 tile 0 landed, 3 .. the start of
                    // every tile, then the end
                    std::vector<int> ts((size_t)grid * 64);
                    (void)hipMemcpy(ts.data(), q.tile_stamps, ts.size() * 4, hipMemcpyDeviceToHost);
                    const int nst = p.nt_total + 4 < 64 ? p.nt_total + 4 : 64;
                    std::vector<double> d(nst, 0.0);
                    for (int k = r * 256; k < (r + 1) * 256 && k < grid; ++k)
                        for (int i = 1; i < nst; ++i) d[i] += (unsigned)(ts[(size_t)order[k].second * 64 + i] - ts[(size_t)order[k].second * 64 + i - 1]);
                    fprintf(stderr, "[stamps w4]     loop: issue %.2f, first tile lands %.2f, then per tile (us):", d[1] / n / 100.0, d[2] / n / 100.0);
                    for (int i = 3; i < nst; ++i) fprintf(stderr, " %.2f", d[i] / n / 100.0);
                    fprintf(stderr, "\n");
                }
            }
        }
    }
#endif
    return check_launch("gemm256w");
}

// ---- BM (32 / 64) rows x BN (32 / 64 / 128 by the widest GroupNorm group) columns, ONE wave per workgroup, bf16: batches <= 64 ----------------------------
// At such a batch a launch has N / 64 workgroups of the tile kernels, one per CU at most, and each K tile is a serial chain of
// barrier, LDS-DMA issue, fragment reads and MFMAs (0.33 us per tile for 64 x 64, 0.59 us for 128 x 128, whatever the ring depth).
// This kernel loads only what such a batch needs -- 32 or 64 rows of A (not 64 of padding for a batch of 32) and 32 to 128 rows
// of W, as the GroupNorm groups allow -- and so spreads a layer over up to four times the CUs.  A single wave owns the whole tile: no
// barrier, the in-order vmcnt counter alone orders the ring (stages of 8 / 12 KiB, 8 / 6 tiles ahead).  A lone wave issues its
// instructions one after another, so the loop is software-pipelined by hand: between the MFMAs of tile t sit the LDS-DMA
// pieces of tile t + D (buffer loads: K offset in a scalar register, M0 written once per four pieces) and
// the fragment reads of tile t + 1 into the other register set.  Every iteration issues the same instructions: past the last
// tile the pieces of the last tile go out again (into stages nobody reads), which keeps the vmcnt arithmetic constant.
// M0 is left at the last LDS-DMA address: the compiler has no use for it in this kernel.
// What bounds a tile now is the rate at which one CU gets lines from beyond its L2, about 48 GB/s: the tile time is proportional
// to the tile's bytes (8 KiB 0.17 us, 12 KiB 0.24 us, 20 KiB 0.40 us); a four-stage ring, a third fewer scalar instructions and a
// padded row pitch of W all left it where it was.
// Fragments, K order and epilogue are those of the tile kernels: bit-identical results.

// Piece m of a tile (1 KiB: 8 rows x 128 bytes) into LDS at the tile's stage + 1024 m.  M0 is written once per four pieces
// (absolutely for m = 0, stepped by 4 KiB for m = 4, 8); the other three reach their slots through the instruction's offset
// field, which shifts the global address as well: their lane offsets `voff` are computed 1024 (m & 3) lower to make up for it.
__device__ __forceinline__ void dma_piece(int m, int voff, i32x4 srd, int soff, unsigned lds) {
    if (m == 0)
        asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" : : "v"(voff), "s"(srd), "s"(soff), "s"(lds) : "memory");
    else if ((m & 3) == 0)
        asm volatile("s_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" : : "v"(voff), "s"(srd), "s"(soff) : "memory");
    else if ((m & 3) == 1)
        asm volatile("buffer_load_dwordx4 %0, %1, %2 offen offset:1024 lds" : : "v"(voff), "s"(srd), "s"(soff) : "memory");
    else if ((m & 3) == 2)
        asm volatile("buffer_load_dwordx4 %0, %1, %2 offen offset:2048 lds" : : "v"(voff), "s"(srd), "s"(soff) : "memory");
    else
        asm volatile("buffer_load_dwordx4 %0, %1, %2 offen offset:3072 lds" : : "v"(voff), "s"(srd), "s"(soff) : "memory");
}

template <int BM, int BN>
__global__ __launch_bounds__(64) void gemm_w1_bf16_kernel(const GemmParams p) {
    typedef bf16_t T;
    constexpr int TM = BM / 16, TN = BN / 16;
    constexpr int AP = BM / 8, WP = BN / 8, P = AP + WP;       // 1 KiB pieces per K tile: A rows, then W rows, contiguous in a stage
    constexpr int STAGE_BYTES = (BM + BN) * 128;
    constexpr int D = 63 / P + 1 < 8 ? 63 / P + 1 : 8;         // tiles ahead = ring stages; (D - 1) * P pieces fit the vmcnt counter
    constexpr int NM = 2 * TM * TN, NR = 2 * (TM + TN);        // MFMAs / fragment reads per K tile
    static_assert(P <= NM && NR <= NM, "one piece and one read at most per MFMA");
    typedef Epilogue<T, 64, BN, BM, BN, TM, TN> Epi;
    static_assert(D * STAGE_BYTES >= Epi::STAGE_BYTES, "the epilogue stages through the ring's memory");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* const prm = reinterpret_cast<float*>(smem + D * STAGE_BYTES);
    float* const xch = prm + 5 * BN;
    const int lane = threadIdx.x;
    // workgroup -> (row tile, column tile), column tiles fastest: the row tiles of one column tile are ntn apart and so (ntn a
    // multiple of 8) on the same XCD, where they find its W rows in L2
    const int m0 = ((int)blockIdx.x / p.ntn) * BM, n0 = ((int)blockIdx.x % p.ntn) * BN;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    // descriptors: the BM rows of A per K segment, the BN rows of W
    auto make_srd = [](unsigned long long base, unsigned bytes) {
        return i32x4{__builtin_amdgcn_readfirstlane((int)(unsigned)base), __builtin_amdgcn_readfirstlane((int)((unsigned)(base >> 32) & 0xffffu)),
                     __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000};
    };
    const unsigned ldw_b = (unsigned)p.ldw * 2u;
    const i32x4 srd_w = make_srd((unsigned long long)p.W + (unsigned long long)n0 * ldw_b, (unsigned)(BN - 1) * ldw_b + (unsigned)p.nt_total * 128u);
    // piece j = rows 8 j .. 8 j + 7: lane -> row 8 j + (lane >> 3), chunk position lane & 7 holding chunk (lane & 7) ^ ((row >> 1) & 7);
    // less 1024 (j & 3) for the instruction offset (rows are at least 128 bytes apart, so the offset stays non-negative)
    int voff[P];
#pragma unroll
    for (int j = 0; j < WP; ++j) {
        const int row = 8 * j + (lane >> 3);
        voff[AP + j] = (int)((unsigned)row * ldw_b + (unsigned)(((lane & 7) ^ ((row >> 1) & 7)) * 16)) - 1024 * ((AP + j) & 3);
    }
    i32x4 srd_a;
    auto enter_segment = [&](int s) {
        const unsigned ld = (unsigned)p.lda[s] * 2u;
        srd_a = make_srd((unsigned long long)p.A[s] + (unsigned long long)m0 * ld, (unsigned)(BM - 1) * ld + (unsigned)p.ktiles[s] * 128u);
#pragma unroll
        for (int j = 0; j < AP; ++j) {
            const int row = 8 * j + (lane >> 3);
            voff[j] = (int)((unsigned)row * ld + (unsigned)(((lane & 7) ^ ((row >> 1) & 7)) * 16)) - 1024 * (j & 3);
        }
    };
    enter_segment(0);

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    const int nt = p.nt_total;
    int seg = 0, tt = 0, it = 0;     // load cursor: K segment, tile inside it, tiles issued (running past nt)
    // descriptors and K offsets of tile `it`; past the last tile they stay at the last tile's
    i32x4 da, dw;
    int soa, sow;
    auto cursor = [&]() {
        if (it < nt) {
            if (tt == p.ktiles[seg]) {   // next K segment that has tiles (wave-uniform)
                ++seg; tt = 0;
                if (tt == p.ktiles[seg]) ++seg;
                enter_segment(seg);
            }
            da = srd_a; dw = srd_w;
            soa = tt * 128; sow = it * 128;
            ++tt;
        }
        ++it;
    };
    // vmcnt counts to 63 and the wave is NOT stalled beyond that: with 72 pieces in flight the counter wrapped and later waits
    // (the compiler's own, for the parameter loads below, included) returned before their data -- stale registers, and a
    // wild pointer once.  So never more than 63 loads in flight: D - 1 tiles, the epilogue's parameter vectors behind them
    // (a load or two at a time; one memory latency for both, not two), a wait that leaves room for a tile, then tile D - 1.
    static_assert((D - 1) * P + 3 <= 63, "LDS-DMA pieces in flight exceed the vmcnt counter");
#pragma unroll
    for (int k = 0; k < D; ++k) {
        if (k == D - 1) {
            Epi::stage_params(p, n0, lane, prm);
            wait_vmcnt<63 - P>();
        }
        cursor();
#pragma unroll
        for (int m = 0; m < P; ++m) {
            dma_piece(m, voff[m], m < AP ? da : dw, m < AP ? soa : sow, lds0 + k * STAGE_BYTES);
        }
    }

    // fragment m of a tile: k half m / (TM + TN), then the TM A fragments and the TN W fragments of that half
    const int frow = lane & 15, fchunk = lane >> 4;
    uint4 fa[2][2][TM], fb[2][2][TN];
    auto read_frag = [&](auto SET, int m, const char* stage) {
        constexpr int S = decltype(SET)::value;
        const int ks = m / (TM + TN), r = m % (TM + TN);
        const int c = 4 * ks + fchunk;
        const int row = (r < TM ? r : r - TM) * 16 + frow;
        const uint4 v = *reinterpret_cast<const uint4*>(stage + (r < TM ? 0 : BM * 128) + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
        if (r < TM) fa[S][ks][r] = v; else fb[S][ks][r - TM] = v;
    };
    wait_vmcnt<(D - 1) * P>();   // tile 0 has landed
#pragma unroll
    for (int m = 0; m < NR; ++m) read_frag(std::integral_constant<int, 0>{}, m, smem);

    int slot = 0;   // stage of tile t: free (its fragments are in registers), takes tile t + D
    auto body = [&](auto SET) {
        constexpr int S = decltype(SET)::value;
        // Every fragment read of tile t (issued during the previous body; the MFMAs below wait for them one by one) must have
        // RETURNED before the first piece of tile t + D is issued into the same stage: with the last reads still queued in LDS
        // and the piece served from a near cache, the piece overtook them now and then, and the rows 32 .. 39 or 48 .. 55 of the
        // tile (the fragments read last, the pieces that land first) were multiplied with the data of tile t + D
        // (1 forward pass in 50 to 1 in 3 at 1024 rows, by machine; tools/train_forward_repro.py).
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        cursor();
        const unsigned dst = lds0 + (unsigned)slot * STAGE_BYTES;
        slot = slot == D - 1 ? 0 : slot + 1;
        const char* const next = smem + slot * STAGE_BYTES;
        wait_vmcnt<(D - 2) * P>();   // tile t + 1 has landed (the D - 2 younger tiles may stay in flight)
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const int ks = m / (TM * TN), i = (m / TN) % TM, j = m % TN;
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[S][ks][i]), __builtin_bit_cast(bf16x8, fb[S][ks][j]),
                                                                acc[i][j], 0, 0, 0);
            if (m < P) dma_piece(m, voff[m], m < AP ? da : dw, m < AP ? soa : sow, dst);
            if (m < NR) read_frag(std::integral_constant<int, S ^ 1>{}, m, next);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    for (int t = 0; t + 1 < nt; t += 2) {
        body(std::integral_constant<int, 0>{});
        body(std::integral_constant<int, 1>{});
    }
    if (nt & 1) body(std::integral_constant<int, 0>{});
    wait_vmcnt<0>();   // the repeated pieces of the last iterations too, before the ring's memory is reused
    Epi::template run<true, (BN >= 64 ? 4 : 2)>(acc, p, m0, n0, 0, 0, 0, lane, reinterpret_cast<float*>(smem), prm, xch);
}

template <int BM, int BN>
static int launch_w1(const GemmParams& p, hipStream_t stream) {
    typedef Epilogue<bf16_t, 64, BN, BM, BN, BM / 16, BN / 16> Epi;
    constexpr int P = (BM + BN) / 8, D = 63 / P + 1 < 8 ? 63 / P + 1 : 8;
    constexpr int smem = D * (BM + BN) * 128 + Epi::EXTRA_BYTES;
    static const hipError_t attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_w1_bf16_kernel<BM, BN>),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (attr_err != hipSuccess) return fail(ADCDIFF_E_LAUNCH, "gemm: hipFuncSetAttribute: %s", hipGetErrorString(attr_err));
    GemmParams q = p;
    q.ntn = p.N / BN;
    const int row_tiles = (p.m_valid + BM - 1) / BM;   // row tiles of padding only are not launched
    hipLaunchKernelGGL((gemm_w1_bf16_kernel<BM, BN>), dim3(row_tiles * q.ntn), dim3(64), smem, stream, q);
    return check_launch("gemm (one wave per tile)");
}

// ---- the one-wave tile with K split over workgroups: launches with at most BM live rows (the generator's batch of 32) ----------------
// A layer at such a batch is its weight matrix streamed once: N / BN one-wave workgroups (8 to 32 of them) each pull BN rows of W
// through ONE CU, and a CU gets about 48 GB/s of lines from beyond its L2 whatever the loop around the loads.  Here every output tile
// has S workgroups, each with a contiguous range of at most D K tiles: a workgroup issues ALL its LDS-DMA pieces at once (at most 63,
// the vmcnt counter), multiplies, and writes its partial accumulators (raw MFMA layout, 1 KiB per store instruction) to a scratch
// buffer; one atomic increment per workgroup on the tile's counter decides who arrived last, and that
// workgroup adds the S partials IN SPLIT ORDER (its own from memory as well, so the sum does not depend on who was last), resets
// the counter and runs the shared epilogue.  The K order differs from the other kernels' (S partial sums instead of one chain), so
// the result is NOT bit-identical to theirs: it is deterministic, and equal to them up to float32 summation order.
template <int BM, int BN>
__global__ __launch_bounds__(64) void gemm_w1s_bf16_kernel(const GemmParams p) {
    typedef bf16_t T;
    constexpr int TM = BM / 16, TN = BN / 16;
    constexpr int AP = BM / 8, WP = BN / 8, P = AP + WP;
    constexpr int STAGE_BYTES = (BM + BN) * 128;
    constexpr int D = 57 / P < 8 ? 57 / P : 8;                 // tiles per round: their pieces and six more loads fit the vmcnt counter
    constexpr int NM = 2 * TM * TN, NR = 2 * (TM + TN);
    typedef Epilogue<T, 64, BN, BM, BN, TM, TN> Epi;
    constexpr int MAIN_BYTES = D * STAGE_BYTES > Epi::STAGE_BYTES ? D * STAGE_BYTES : Epi::STAGE_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* const prm = reinterpret_cast<float*>(smem + MAIN_BYTES);
    float* const xch = prm + 5 * BN;
    const int lane = threadIdx.x;
    const int S = p.sk_splits;
#ifdef ADCDIFF_STAMPS
    long long stamp[8] = {(long long)__builtin_amdgcn_s_memrealtime(), 0, 0, 0, 0, 0, 0, 0};
    auto put_stamps = [&]() {
        if (lane == 0)
            for (int i = 0; i < 8; ++i) p.stamps[(size_t)blockIdx.x * 8 + i] = stamp[i];
    };
#endif
    // workgroup -> (tile, split): the S workgroups of a tile on ONE XCD (consecutive workgroup ids go round the eight XCDs), so that
    // the last one finds the others' partials in its own L2; any other tile count takes the plain order (the fences do not care)
    int tile, sp;
    {
        const int b = (int)blockIdx.x;
        if ((p.ntn & 7) == 0) { const int j = b >> 3; sp = j % S; tile = (j / S) * 8 + (b & 7); }
        else { tile = b / S; sp = b % S; }
    }
    const int m0 = 0, n0 = tile * BN;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const int nt = p.nt_total;
    const int t0 = (int)((long)sp * nt / S), t1 = (int)((long)(sp + 1) * nt / S);   // this workgroup's K tiles

    auto make_srd = [](unsigned long long base, unsigned bytes) {
        return i32x4{__builtin_amdgcn_readfirstlane((int)(unsigned)base), __builtin_amdgcn_readfirstlane((int)((unsigned)(base >> 32) & 0xffffu)),
                     __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000};
    };
    const unsigned ldw_b = (unsigned)p.ldw * 2u;
    const i32x4 srd_w = make_srd((unsigned long long)p.W + (unsigned long long)n0 * ldw_b, (unsigned)(BN - 1) * ldw_b + (unsigned)nt * 128u);
    int voff[P];
#pragma unroll
    for (int j = 0; j < WP; ++j) {
        const int row = 8 * j + (lane >> 3);
        voff[AP + j] = (int)((unsigned)row * ldw_b + (unsigned)(((lane & 7) ^ ((row >> 1) & 7)) * 16)) - 1024 * ((AP + j) & 3);
    }
    i32x4 srd_a;
    auto enter_segment = [&](int sg) {
        const unsigned ld = (unsigned)p.lda[sg] * 2u;
        srd_a = make_srd((unsigned long long)p.A[sg] + (unsigned long long)m0 * ld, (unsigned)(BM - 1) * ld + (unsigned)p.ktiles[sg] * 128u);
#pragma unroll
        for (int j = 0; j < AP; ++j) {
            const int row = 8 * j + (lane >> 3);
            voff[j] = (int)((unsigned)row * ld + (unsigned)(((lane & 7) ^ ((row >> 1) & 7)) * 16)) - 1024 * (j & 3);
        }
    };
    // the K segment that holds tile t0 (t0 < nt = the sum of the segments' tiles, so the walk ends inside the segments)
    int seg = 0, tt = t0;
    while (seg < 2 && tt >= p.ktiles[seg]) { tt -= p.ktiles[seg]; ++seg; }
    enter_segment(seg);

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    const int frow = lane & 15, fchunk = lane >> 4;
    // rounds of at most D tiles: all pieces of a round go out at once (at most 57 loads in flight, plus the epilogue's column
    // vectors in the first round: the vmcnt counter holds 63), then the round's tiles are multiplied.  Nothing overlaps a
    // round's loads with the previous round's MFMAs: a workgroup has a handful of tiles and its time is one or two memory
    // latencies either way.
    int it = t0;
    bool first = true;
#ifdef ADCDIFF_STAMPS
    stamp[1] = __builtin_amdgcn_s_memrealtime();
#endif
    while (it < t1) {
        const int nround = t1 - it < D ? t1 - it : D;
#pragma unroll
        for (int k = 0; k < D; ++k) {
            if (k < nround) {   // wave-uniform
                if (tt == p.ktiles[seg]) {
                    ++seg; tt = 0;
                    if (seg < 2 && tt == p.ktiles[seg]) ++seg;
                    enter_segment(seg);
                }
                const int soa = tt * 128, sow = it * 128;
#pragma unroll
                for (int m = 0; m < P; ++m) dma_piece(m, voff[m], m < AP ? srd_a : srd_w, m < AP ? soa : sow, lds0 + k * STAGE_BYTES);
                ++tt; ++it;
            }
        }
        if (first) Epi::stage_params(p, n0, lane, prm);   // younger than every piece of the round
        wait_vmcnt<0>();
#ifdef ADCDIFF_STAMPS
        if (first) stamp[2] = __builtin_amdgcn_s_memrealtime();
#endif
        first = false;
        for (int t = 0; t < nround; ++t) {
            const char* const stage = smem + t * STAGE_BYTES;
            uint4 fa[2][TM], fb[2][TN];
#pragma unroll
            for (int m = 0; m < NR; ++m) {
                const int ks = m / (TM + TN), r = m % (TM + TN);
                const int c = 4 * ks + fchunk;
                const int row = (r < TM ? r : r - TM) * 16 + frow;
                const uint4 v = *reinterpret_cast<const uint4*>(stage + (r < TM ? 0 : BM * 128) + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
                if (r < TM) fa[ks][r] = v; else fb[ks][r - TM] = v;
            }
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                const int ks = m / (TM * TN), i = (m / TN) % TM, j = m % TN;
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[ks][i]), __builtin_bit_cast(bf16x8, fb[ks][j]), acc[i][j], 0, 0, 0);
            }
        }
        // the round's fragment reads have returned (the MFMAs consumed them) before the next round's pieces overwrite the stages
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }

    // partial accumulators out, arrival count up; the last workgroup of the tile sums all S partials in split order.
    // The exchange goes THROUGH the caches instead of flushing them: the partials are stored and loaded with the system-scope cache
    // bits (sc0 sc1: write-through to / read from the memory side), 16 bytes per lane = 1 KiB per instruction, the stores are waited
    // for (vmcnt) before the lane-0 increment, which is performed at the memory side as well.  A device-scope fence pair here (an L2
    // write-back and invalidate per workgroup) cost 6 us per launch, more than the split gains.  Four-byte accesses (relaxed
    // system-scope atomics, the first form of this exchange) are bound by their instruction count: 480 loads = 4 us for the 15
    // foreign partials of a 32 x 64 tile; with 16-byte accesses the reduction is a memory round trip per batch of loads.
    // Layout: [tile][split][16 x 16 block][lane] float4.
    auto exchange = [&]() {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        constexpr int SC = 17;   // sc0 | sc1
        constexpr int NB = TM * TN;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.sk_part, 0, (int)((unsigned)p.ntn * (unsigned)S * (unsigned)(NB * 1024)), 0x00020000);
        const int voff = lane * 16;
        const int mine = __builtin_amdgcn_readfirstlane((tile * S + sp) * (NB * 1024));
#pragma unroll
        for (int k = 0; k < NB; ++k)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[k / TN][k % TN]), rs, voff, mine + k * 1024, SC);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every lane's partials have reached the coherence point
#ifdef ADCDIFF_STAMPS
        stamp[4] = __builtin_amdgcn_s_memrealtime();
#endif
        unsigned arrived = 0;
        if (lane == 0) arrived = __hip_atomic_fetch_add(p.sk_cnt + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        arrived = (unsigned)__builtin_amdgcn_readfirstlane((int)arrived);
#ifdef ADCDIFF_STAMPS
        stamp[5] = __builtin_amdgcn_s_memrealtime();
#endif
        if (arrived != (unsigned)(S - 1)) return false;
        asm volatile("" ::: "memory");   // the loads below stay behind the increment's return
        const int all = __builtin_amdgcn_readfirstlane(tile * S * (NB * 1024));
#pragma unroll
        for (int k = 0; k < NB; ++k)
            acc[k / TN][k % TN] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, all + k * 1024, SC));
        // batches of U splits' loads in flight (up to 256 registers of a lone wave's 512), added in split order
        constexpr int U = 64 / NB >= 1 ? (64 / NB > 15 ? 15 : 64 / NB) : 1;
        for (int q = 1; q < S; q += U) {
            f32x4 v[U][NB];
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (q + u < S) {
                    const int src = all + (q + u) * (NB * 1024);
#pragma unroll
                    for (int k = 0; k < NB; ++k) v[u][k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, src + k * 1024, SC));
                }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (q + u < S) {
#pragma unroll
                    for (int k = 0; k < NB; ++k) acc[k / TN][k % TN] += v[u][k];
                }
        }
        if (lane == 0) __hip_atomic_store(p.sk_cnt + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // for the next launch
        return true;
    };
#ifdef ADCDIFF_STAMPS
    stamp[3] = __builtin_amdgcn_s_memrealtime();
    if (!exchange()) { put_stamps(); return; }
    stamp[6] = __builtin_amdgcn_s_memrealtime();
#else
    if (!exchange()) return;
#endif
    Epi::template run<true, (BN >= 64 ? 4 : 2)>(acc, p, m0, n0, 0, 0, 0, lane, reinterpret_cast<float*>(smem), prm, xch);
#ifdef ADCDIFF_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp[7] = __builtin_amdgcn_s_memrealtime();
    put_stamps();
#endif
}

constexpr int w1s_tiles(int bm, int bn) { return 57 / ((bm + bn) / 8) < 8 ? 57 / ((bm + bn) / 8) : 8; }

template <int BM, int BN>
static int launch_w1s(const GemmParams& p, int splits, hipStream_t stream) {
    typedef Epilogue<bf16_t, 64, BN, BM, BN, BM / 16, BN / 16> Epi;
    constexpr int D = w1s_tiles(BM, BN);
    constexpr int main_bytes = D * (BM + BN) * 128 > Epi::STAGE_BYTES ? D * (BM + BN) * 128 : Epi::STAGE_BYTES;
    constexpr int smem = main_bytes + Epi::EXTRA_BYTES;
    static_assert(smem <= 160 * 1024, "LDS of a CU");
    static const hipError_t attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_w1s_bf16_kernel<BM, BN>),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (attr_err != hipSuccess) return fail(ADCDIFF_E_LAUNCH, "gemm: hipFuncSetAttribute: %s", hipGetErrorString(attr_err));
    GemmParams q = p;
    q.ntn = p.N / BN;
    q.sk_splits = splits;
    (void)D;
    if (p.m_valid > BM || splits < 2 || splits > p.nt_total)
        return fail(ADCDIFF_E_BADARG, "gemm: split-K geometry (rows %d > %d, %d splits of %d K tiles)", p.m_valid, BM, splits, p.nt_total);
#ifdef ADCDIFF_STAMPS
    static long long* dbuf = nullptr;
    if (!dbuf) (void)hipMalloc(&dbuf, (size_t)8192 * 8 * 8);
    q.stamps = dbuf;
#endif
    hipLaunchKernelGGL((gemm_w1s_bf16_kernel<BM, BN>), dim3(q.ntn * splits), dim3(64), smem, stream, q);
#ifdef ADCDIFF_STAMPS
    {
        // measurement build: every 40th launch of an instantiation is dissected (10 ns units): workgroup start after the first
        // workgroup's, set-up, first round of loads, the rest of the K range, partial stores until acknowledged, the arrival
        // atomic; for the finishing workgroups the reduction and the epilogue
        static int calls = 0;
        const int grid = q.ntn * splits;
        if (++calls % 40 == 0 && grid <= 8192) {
            (void)hipDeviceSynchronize();
            std::vector<long long> h((size_t)grid * 8);
            (void)hipMemcpy(h.data(), q.stamps, h.size() * 8, hipMemcpyDeviceToHost);
            long long t0 = 1LL << 62, t_end = 0, last_start = 0;
            for (int b = 0; b < grid; ++b) {
                const long long* o = &h[(size_t)b * 8];
                if (o[0] < t0) t0 = o[0];
                if (o[0] > last_start) last_start = o[0];
                const long long e = o[7] ? o[7] : o[5];
                if (e > t_end) t_end = e;
            }
            double seg[5] = {0}, fin[2] = {0}, fin_start = 0; int nfin = 0;
            for (int b = 0; b < grid; ++b) {
                const long long* o = &h[(size_t)b * 8];
                seg[0] += o[1] - o[0]; seg[1] += o[2] - o[1]; seg[2] += o[3] - o[2]; seg[3] += o[4] - o[3]; seg[4] += o[5] - o[4];
                if (o[7]) { fin[0] += o[6] - o[5]; fin[1] += o[7] - o[6]; fin_start += o[5] - t0; ++nfin; }
            }
            fprintf(stderr, "[stamps w1s<%d,%d>] N %d K tiles %d splits %d grid %d: first to last workgroup start %.2f us, span %.2f us; mean set-up %.2f, first round %.2f, "
                            "rest of K %.2f, partial stores %.2f, arrival %.2f us; finishing workgroups (%d): arrive at +%.2f us, reduction %.2f, epilogue %.2f us\n",
                    BM, BN, p.N, p.nt_total, splits, grid, (last_start - t0) / 100.0, (t_end - t0) / 100.0, seg[0] / grid / 100.0, seg[1] / grid / 100.0, seg[2] / grid / 100.0,
                    seg[3] / grid / 100.0, seg[4] / grid / 100.0, nfin, fin_start / (nfin ? nfin : 1) / 100.0, fin[0] / (nfin ? nfin : 1) / 100.0, fin[1] / (nfin ? nfin : 1) / 100.0);
        }
    }
#endif
    return check_launch("gemm (one wave per tile, K split over workgroups)");
}

// ---- the same tile with loader waves: wave 0 reads fragments, multiplies and runs the epilogue, waves 1 .. P / 4 issue the LDS-DMA --
// The lone wave above issues all 8 to 24 pieces of a tile between its MFMAs.  Here every group of four pieces (one M0 write) has a
// wave of its own with its
// own ring bookkeeping and vmcnt, and the waves meet at ONE barrier per K tile: a loader arrives when its part of tile t has landed,
// wave 0 when its fragment reads of tile t - 1 have returned, after which the loaders refill the stage of tile t - 1.  Every wave
// runs exactly nt barriers.  Fragments, K order and epilogue as in the other kernels: bit-identical results.
constexpr int cl_loaders(int bm, int bn) { return bm == 64 && bn > 32 ? 3 : (bm + bn) / 32; }

template <int BM, int BN>
__global__ __launch_bounds__(64 * (1 + cl_loaders(BM, BN))) void gemm_cl_bf16_kernel(const GemmParams p) {
    typedef bf16_t T;
    constexpr int TM = BM / 16, TN = BN / 16;
    constexpr int AP = BM / 8, WP = BN / 8, P = AP + WP;
    constexpr int STAGE_BYTES = (BM + BN) * 128;
    constexpr int D = 147456 / STAGE_BYTES < 8 ? 147456 / STAGE_BYTES : 8;   // ring stages; a loader has (D - 1) * 4 pieces in flight
    // groups of four pieces; loader waves: one per group, except for the 64-row tiles wider than 32 columns, whose compute wave needs
    // more than the 256 registers that five or seven waves would leave it: three loaders there, with one or two groups each
    constexpr int G = P / 4, NL = cl_loaders(BM, BN);
    constexpr int NM = 2 * TM * TN, NR = 2 * (TM + TN);
    static_assert(G <= 2 * NL && (D - 1) * 8 <= 63, "loader geometry");
    static_assert(NR <= NM && D >= 3 && D - 2 <= 7 && AP % 4 == 0 && WP % 4 == 0, "ring geometry");
    typedef Epilogue<T, 64, BN, BM, BN, TM, TN> Epi;
    static_assert(D * STAGE_BYTES >= Epi::STAGE_BYTES, "the epilogue stages through the ring's memory");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* const prm = reinterpret_cast<float*>(smem + D * STAGE_BYTES);
    float* const xch = prm + 5 * BN;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m0 = ((int)blockIdx.x / p.ntn) * BM, n0 = ((int)blockIdx.x % p.ntn) * BN;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const int nt = p.nt_total;

    if (wave != 0) {
      if constexpr (NL == G) {
        // ---- loader: pieces 4 (wave - 1) .. + 3 of every tile, all of them A rows or all of them W rows
        const int first = 4 * (wave - 1);
        const bool loads_a = first < AP;
        const int row0 = 8 * (loads_a ? first : first - AP);   // first row of this wave's 32 rows inside the operand's tile
        auto make_srd = [](unsigned long long base, unsigned bytes) {
            return i32x4{__builtin_amdgcn_readfirstlane((int)(unsigned)base), __builtin_amdgcn_readfirstlane((int)((unsigned)(base >> 32) & 0xffffu)),
                         __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000};
        };
        i32x4 srd;
        int voff[4];   // lane offsets of the four pieces, 1024 j lower for the instruction offset
        auto enter_segment = [&](int sg) {   // A: K segment sg; W: its single segment
            const unsigned ld = loads_a ? (unsigned)p.lda[sg] * 2u : (unsigned)p.ldw * 2u;
            const unsigned long long base = loads_a ? (unsigned long long)p.A[sg] + (unsigned long long)m0 * ld
                                                    : (unsigned long long)p.W + (unsigned long long)n0 * ld;
            const int kt = loads_a ? p.ktiles[sg] : nt;
            srd = make_srd(base, (unsigned)((loads_a ? BM : BN) - 1) * ld + (unsigned)kt * 128u);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = row0 + 8 * j + (lane >> 3);
                voff[j] = (int)((unsigned)row * ld + (unsigned)(((lane & 7) ^ ((row >> 1) & 7)) * 16)) - 1024 * j;
            }
        };
        enter_segment(0);
        int seg = 0, tt = 0;
        auto issue = [&](int stage) {   // the next tile into `stage`
            if (loads_a && tt == p.ktiles[seg]) {
                ++seg; tt = 0;
                if (tt == p.ktiles[seg]) ++seg;
                enter_segment(seg);
            }
            const int so = tt * 128;
            const unsigned dst = lds0 + (unsigned)stage * STAGE_BYTES + (unsigned)first * 1024u;
#pragma unroll
            for (int j = 0; j < 4; ++j) dma_piece(j, voff[j], srd, so, dst);
            ++tt;
        };
#pragma unroll
        for (int k = 0; k < D - 1; ++k)
            if (k < nt) issue(k);
        int stage = D - 1;   // of tile t - 1 + D: the never-used last stage at t = 0, then the stage of tile t - 1
        for (int t = 0; t < nt; ++t) {
            const int ahead = (nt < t + D - 1 ? nt : t + D - 1) - (t + 1);   // tiles issued after tile t
            wait_tiles_in_flight<4>(ahead);                                   // this wave's part of tile t has landed
            __builtin_amdgcn_s_barrier();                                     // ... everyone's has, and wave 0 has read tile t - 1
            if (t + D - 1 < nt) issue(stage);
            stage = stage == D - 1 ? 0 : stage + 1;
        }
        return;
      } else {
        // ---- loader: the groups wave - 1, wave - 1 + NL, ... of four pieces each (a group is all A rows or all W rows)
        const int w = wave - 1;
        const int ng = (G - w + NL - 1) / NL;   // 1 or 2 groups
        auto make_srd = [](unsigned long long base, unsigned bytes) {
            return i32x4{__builtin_amdgcn_readfirstlane((int)(unsigned)base), __builtin_amdgcn_readfirstlane((int)((unsigned)(base >> 32) & 0xffffu)),
                         __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000};
        };
        i32x4 srd[2];
        int voff[2][4];   // lane offsets of a group's four pieces, 1024 j lower for the instruction offset
        bool is_a[2];
        int row0[2];      // first row of the group's 32 rows inside its operand's tile
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
            const int first = 4 * (w + gi * NL);
            is_a[gi] = first < AP;
            row0[gi] = 8 * (is_a[gi] ? first : first - AP);
        }
        auto enter_segment = [&](int gi, int sg) {   // A: K segment sg; W: its single segment
            const unsigned ld = is_a[gi] ? (unsigned)p.lda[sg] * 2u : (unsigned)p.ldw * 2u;
            const unsigned long long base = is_a[gi] ? (unsigned long long)p.A[sg] + (unsigned long long)m0 * ld
                                                     : (unsigned long long)p.W + (unsigned long long)n0 * ld;
            const int kt = is_a[gi] ? p.ktiles[sg] : nt;
            srd[gi] = make_srd(base, (unsigned)((is_a[gi] ? BM : BN) - 1) * ld + (unsigned)kt * 128u);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = row0[gi] + 8 * j + (lane >> 3);
                voff[gi][j] = (int)((unsigned)row * ld + (unsigned)(((lane & 7) ^ ((row >> 1) & 7)) * 16)) - 1024 * j;
            }
        };
#pragma unroll
        for (int gi = 0; gi < 2; ++gi)
            if (gi < ng) enter_segment(gi, 0);
        int seg = 0, tt = 0, it = 0;   // K segment of A, tile inside it, tiles issued
        auto issue = [&](int stage) {   // the next tile into `stage`
            if (tt == p.ktiles[seg]) {
                ++seg; tt = 0;
                if (tt == p.ktiles[seg]) ++seg;
#pragma unroll
                for (int gi = 0; gi < 2; ++gi)
                    if (gi < ng && is_a[gi]) enter_segment(gi, seg);
            }
#pragma unroll
            for (int gi = 0; gi < 2; ++gi) {
                if (gi < ng) {
                    const int so = (is_a[gi] ? tt : it) * 128;
                    const unsigned dst = lds0 + (unsigned)stage * STAGE_BYTES + (unsigned)(w + gi * NL) * 4096u;
                    const i32x4 sd = {__builtin_amdgcn_readfirstlane(srd[gi][0]), __builtin_amdgcn_readfirstlane(srd[gi][1]),
                                      __builtin_amdgcn_readfirstlane(srd[gi][2]), 0x00020000};   // wave-uniform: keep it in scalar registers
#pragma unroll
                    for (int j = 0; j < 4; ++j) dma_piece(j, voff[gi][j], sd, __builtin_amdgcn_readfirstlane(so), __builtin_amdgcn_readfirstlane((int)dst));
                }
            }
            ++tt; ++it;
        };
#pragma unroll
        for (int k = 0; k < D - 1; ++k)
            if (k < nt) issue(k);
        int stage = D - 1;
        for (int t = 0; t < nt; ++t) {
            const int ahead = (nt < t + D - 1 ? nt : t + D - 1) - (t + 1);
            if (ng == 2) wait_tiles_in_flight<8>(ahead);
            else wait_tiles_in_flight<4>(ahead);
            __builtin_amdgcn_s_barrier();
            if (t + D - 1 < nt) issue(stage);
            stage = stage == D - 1 ? 0 : stage + 1;
        }
        return;
      }
    }

    // ---- compute wave
    Epi::stage_params(p, n0, lane, prm);
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    const int frow = lane & 15, fchunk = lane >> 4;
    uint4 fa[2][2][TM], fb[2][2][TN];
    auto read_frag = [&](auto SET, int m, const char* stage) {
        constexpr int S = decltype(SET)::value;
        const int ks = m / (TM + TN), r = m % (TM + TN);
        const int c = 4 * ks + fchunk;
        const int row = (r < TM ? r : r - TM) * 16 + frow;
        const uint4 v = *reinterpret_cast<const uint4*>(stage + (r < TM ? 0 : BM * 128) + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
        if (r < TM) fa[S][ks][r] = v; else fb[S][ks][r - TM] = v;
    };
    auto mfma = [&](auto SET, int m) {
        constexpr int S = decltype(SET)::value;
        const int ks = m / (TM * TN), i = (m / TN) % TM, j = m % TN;
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[S][ks][i]), __builtin_bit_cast(bf16x8, fb[S][ks][j]),
                                                            acc[i][j], 0, 0, 0);
    };
    int slot = 0;   // stage of the tile read next
    auto next_stage = [&]() { const char* const st = smem + slot * STAGE_BYTES; slot = slot == D - 1 ? 0 : slot + 1; return st; };
    __builtin_amdgcn_s_barrier();   // tile 0
    {
        const char* const st = next_stage();
#pragma unroll
        for (int m = 0; m < NR; ++m) read_frag(std::integral_constant<int, 0>{}, m, st);
    }
    // tile t into set S while the MFMAs of tile t - 1 run from the other set
    auto body = [&](auto SET) {
        constexpr int S = decltype(SET)::value;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the reads of tile t - 1 have returned: its stage may be refilled
        __builtin_amdgcn_s_barrier();
        const char* const st = next_stage();
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            mfma(std::integral_constant<int, S ^ 1>{}, m);
            if (m < NR) read_frag(SET, m, st);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    int t = 1;
    for (; t + 1 < nt; t += 2) {
        body(std::integral_constant<int, 1>{});
        body(std::integral_constant<int, 0>{});
    }
    if (t < nt) {
        body(std::integral_constant<int, 1>{});
#pragma unroll
        for (int m = 0; m < NM; ++m) mfma(std::integral_constant<int, 1>{}, m);
    } else {
#pragma unroll
        for (int m = 0; m < NM; ++m) mfma(std::integral_constant<int, 0>{}, m);
    }
    // every tile had landed before the last barrier and this wave's reads have returned: the ring's memory is free
    Epi::template run<true, (BN >= 64 ? 4 : 2)>(acc, p, m0, n0, 0, 0, 0, lane, reinterpret_cast<float*>(smem), prm, xch);
}

template <int BM, int BN>
static int launch_cl(const GemmParams& p, hipStream_t stream) {
    typedef Epilogue<bf16_t, 64, BN, BM, BN, BM / 16, BN / 16> Epi;
    constexpr int STAGE = (BM + BN) * 128, D = 147456 / STAGE < 8 ? 147456 / STAGE : 8;
    constexpr int smem = D * STAGE + Epi::EXTRA_BYTES;
    static_assert(smem <= 160 * 1024, "LDS of a CU");
    static const hipError_t attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_cl_bf16_kernel<BM, BN>),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (attr_err != hipSuccess) return fail(ADCDIFF_E_LAUNCH, "gemm: hipFuncSetAttribute: %s", hipGetErrorString(attr_err));
    GemmParams q = p;
    q.ntn = p.N / BN;
    const int row_tiles = (p.m_valid + BM - 1) / BM;
    hipLaunchKernelGGL((gemm_cl_bf16_kernel<BM, BN>), dim3(row_tiles * q.ntn), dim3(64 * (1 + cl_loaders(BM, BN))), smem, stream, q);
    return check_launch("gemm (compute wave + loader waves)");
}

template <typename T, int BM, int BN, int NSTAGE>
static int launch(const GemmParams& p, hipStream_t stream) {
    constexpr int TS = sizeof(T) == 2 ? 16 : 32;
    typedef Epilogue<T, 256, BN, BM / 2, BN / 2, BM / 2 / TS, BN / 2 / TS> Epi;
    constexpr int main_bytes = NSTAGE * (BM + BN) * 128;
    constexpr int smem = (main_bytes > Epi::STAGE_BYTES ? main_bytes : Epi::STAGE_BYTES) + Epi::EXTRA_BYTES;
    static const hipError_t attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_kernel<T, BM, BN, NSTAGE>),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (attr_err != hipSuccess) return fail(ADCDIFF_E_LAUNCH, "gemm: hipFuncSetAttribute: %s", hipGetErrorString(attr_err));
    GemmParams q = p;
    q.ntn = p.N / BN;
    const int grid = (p.M / BM) * q.ntn;
    hipLaunchKernelGGL((gemm_nt_kernel<T, BM, BN, NSTAGE>), dim3(grid, p.ksplit), dim3(256), smem, stream, q);
    return check_launch("gemm");
}

template <typename T, int BM, int BN>
static int launch_any(const GemmParams& p, bool ring, hipStream_t stream) {
    // three stages: deeper rings (six stages of 64 x 64, four of 128 x 128) measured 3-5 % slower -- past a prefetch distance
    // of two tiles the per-tile barrier and LDS-DMA issue are the critical path, not the load latency
    // more workgroups than CUs (fp32 batch 4096: 512 tiles of 128 x 128): two stages of 32 KiB let two workgroups share a CU,
    // and one computes while the other waits at its barrier: +8 % samples/s over three stages with one workgroup per CU
    const bool two_ok = debug_value(DBG_GEMM_RING2) != 0;
    if (ring && two_ok && BM == 128 && BN == 128 && (long)(p.M / BM) * (p.N / BN) * p.ksplit > 256) return launch<T, BM, BN, 2>(p, stream);
    return ring ? launch<T, BM, BN, 3>(p, stream) : launch<T, BM, BN, 1>(p, stream);
}

int gemm_dispatch(const adcdiff_gemm_args& a, hipStream_t stream) {
    if (a.dtype != ADCDIFF_F32 && a.dtype != ADCDIFF_BF16) return fail(ADCDIFF_E_BADARG, "gemm: unknown dtype %d", a.dtype);
    if (a.tile_hint != 0 && debug_value(DBG_TILE_HINTS) == 0)
        return fail(ADCDIFF_E_BADARG, "gemm: tile_hint=%d: kernel selection by hint is a measurement aid, enable it with adcdiff_debug_set(\"tile_hints\", 1)", a.tile_hint);
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
    p.ksplit = 1; p.split_k_bytes = 0; p.split_out_stride = 0;
    int kslice = ktot;
    if (a.split_k > 1) {
        if (a.nseg != 1 || !a.out_f32 || a.out_op || a.n_gn || a.bias || a.rowbias || a.residual || a.act)
            return fail(ADCDIFF_E_BADARG, "gemm: split_k needs one K segment and out_f32 as the only output, no bias / rowbias / residual / act");
        if (ktot % (a.split_k * bk) != 0 || a.split_stride < (long long)(a.M - 1) * a.ldo_f32 + a.N)
            return fail(ADCDIFF_E_BADARG, "gemm: split_k=%d must divide K=%d into multiples of %d, split_stride=%lld must hold one output", a.split_k,
                        ktot, bk, a.split_stride);
        kslice = ktot / a.split_k;
        p.ksplit = a.split_k; p.split_k_bytes = (long long)kslice * esize; p.split_out_stride = a.split_stride;
    }
    p.W = a.W; p.ldw = a.ldw; p.M = a.M; p.N = a.N; p.nt_total = kslice / bk;
    if (a.split_k > 1) p.ktiles[0] = kslice / bk;
    p.m_valid = a.m_valid == 0 ? a.M : a.m_valid;
    p.bias = a.bias; p.rowbias = a.rowbias; p.ldrb = a.ldrb; p.rowbias_idx = a.rowbias_idx;
    p.rowbias_rows = a.rowbias_idx ? a.rowbias_rows : 0;
    p.residual = a.residual; p.ldr = a.ldr; p.act = a.act;
    p.upd = a.update;
    if (a.update.kind != 0) {
        const adcdiff_step_update& u = a.update;
        if (u.kind != 1 && u.kind != 2) return fail(ADCDIFF_E_BADARG, "gemm: update kind %d (1 = DDPM, 2 = DDIM)", u.kind);
        if (a.act || a.split_k > 1) return fail(ADCDIFF_E_BADARG, "gemm: the schedule update is not combined with act or split_k");
        if (!u.x || !aligned16(u.x) || u.ldx < a.N || u.ldx % 4 != 0 || (u.noise && (!aligned16(u.noise) || u.ldn < a.N || u.ldn % 4 != 0)))
            return fail(ADCDIFF_E_BADARG, "gemm: update: x / noise null, misaligned or leading dimension < N");
    }
    p.out_f32 = a.out_f32; p.ldo_f32 = a.ldo_f32;
    p.out_op = a.out_op; p.ldo_op = a.ldo_op;
    const double esz = esize;
    const double nout = (a.out_f32 ? 4.0 : 0.0) + esz * ((a.out_op ? 1 : 0) + a.n_gn);
    // Kernel choice for bf16, from A/B timings of the kernels on the denoiser's layer shapes at M = 65536 (interleaved in one
    // process): the 256 x 256 kernel (one workgroup per CU, prefetch under the MFMAs, rolled epilogue) wins or ties on every
    // shape with at least eight K tiles, by up to 27 % on the 256-wide layers; the 128-row kernel (3-4 workgroups per CU)
    // keeps the K = 256 layers and everything with too few 256-tiles to fill the chip.
    bool big = false, pipelined = false;
    // the two-group phased loop is the default of the eight-wave kernel: same cycles per tile as the ring loop within 1 %, but
    // the chip holds a higher clock on it (debug switch gemm_ring / tile_hint 2561 select the ring loop)
    const bool pipe_default = debug_value(DBG_GEMM_RING) != 0;
    if (a.dtype == ADCDIFF_BF16 && a.M % 256 == 0 && a.N % 256 == 0 && a.tile_hint != 128) {
        const int forced = (int)debug_value(DBG_GEMM_TILE);
        const bool enough_tiles = forced != 128 && (long)(a.M / 256) * (a.N / 256) * p.ksplit >= 192;
        // K = 256 (four K tiles) pays on the four-wave kernel unless the epilogue carries the schedule update: 69 -> 54 us for the
        // N = 512 layer at M = 65536, but 150 -> 167 us for the output head with the update folded in, which the eight-wave kernel
        // (two waves per SIMD to hide the update's latent reads) runs in 137 us
        const bool long_k = kslice >= 512 || (kslice >= (int)debug_value(DBG_GEMM_BIG_MIN_K) && a.N >= (int)debug_value(DBG_GEMM_W4_MIN_N));
        big = forced == 256 || forced == 2561 || a.tile_hint == 256 || a.tile_hint == 2561 || a.tile_hint == 2560 || (a.tile_hint >= 2564 && a.tile_hint <= 2568) || (enough_tiles && long_k);
        pipelined = a.tile_hint == 2561 || (a.tile_hint != 2560 && pipe_default);
    }
    // few rows (the float32 batch-4096 case, small generation batches): 64 x 64 tiles give four times the workgroups
    bool small = (long)(a.M / 128) * (a.N / (wide ? 128 : 64)) * p.ksplit < 256 && a.tile_hint != 128 && a.tile_hint != 1 && a.tile_hint != 3;
    // only where the GroupNorm groups stay inside a 32-column wave tile: every kernel then sums a group in the same
    // order, so a row's result does not depend on the batch size it is computed in (bit for bit)
    const bool few = small;
    for (int g = 0; g < a.n_gn; ++g) small = small && a.gn[g].group_size <= 32;
    // the same launches with groups of 64 channels: 64 rows x 128 columns (a 32 x 64 wave tile holds a whole group), half
    // the MFMA work of a 128-row tile whose upper rows are padding
    bool mid = few && !small && wide;
    for (int g = 0; g < a.n_gn; ++g) mid = mid && a.gn[g].group_size <= 64;
    // big: the 4-wave kernel with the assembly main loop unless the launch splits K, has a single K tile, is one tile wide (the
    // 8-wave kernel's epilogue is the faster one there and the main loop is short), or a hint / the debug switch gemm_w4 = 0
    // asks for one of the 8-wave kernels
    const bool w4_default = debug_value(DBG_GEMM_W4) != 0;
    const bool w4_hint = a.tile_hint >= 2564 && a.tile_hint <= 2568;
    const bool w4 = big && p.ksplit == 1 && p.nt_total >= 2 && !(kslice < 512 && a.update.kind != 0) &&
                    (w4_hint || (w4_default && a.N >= (int)debug_value(DBG_GEMM_W4_MIN_N) && a.tile_hint != 2560 && a.tile_hint != 2561 && a.tile_hint != 256));
    const int kind = a.dtype == ADCDIFF_F32 ? ADCDIFF_KIND_GEMM_F32 : w4 ? ADCDIFF_KIND_GEMM_BF16_W256 : big ? ADCDIFF_KIND_GEMM_BF16_T256 : ADCDIFF_KIND_GEMM_BF16;
    TimingScope ts(kind, a.M, a.N, ktot, 2.0 * a.M * a.N * ktot,
                   esz * ((double)a.M * ktot + (double)a.N * ktot) + (double)a.M * a.N * (nout + (a.residual ? esz : 0.0)), stream);
    if (big) {
        // measurement aids: 2565 / 2566 / 2568 = epilogue batches of 1 / 2 / 8 rows per lane
        if (w4) {
            if (a.tile_hint == 2565) return launch256w<1>(p, stream);
            if (a.tile_hint == 2566) return launch256w<2>(p, stream);
            if (a.tile_hint == 2568) return launch256w<8>(p, stream);
            const bool row_rb = p.rowbias != nullptr && (p.rowbias_idx != nullptr || p.ldrb != 0);
            if (row_rb || a.tile_hint == 2567) return launch256w<4, 0>(p, stream);
            if (p.act != 0 || p.upd.kind != 0 || p.out_f32) return launch256w<4, 0>(p, stream);
            return p.residual ? launch256w<4, 2>(p, stream) : launch256w<4, 1>(p, stream);
        }
        return launch256(p, pipelined, stream);
    }
    // few live rows (the generator's default batch is 32): the one-wave kernel, 32 or 64 rows and 32 columns per workgroup
    // where every GroupNorm group fits in 32, else 64 or 128 (a batch-32 sampling run takes 24 % less time than on the tile kernels); tile_hint 32
    // asks for it, any other hint (or the debug switch gemm_w1 = 0) for the tile kernels
    const bool w1_default = debug_value(DBG_GEMM_W1) != 0;
    // up to 1024 rows (16 row tiles of 64): 12 % less time per sampling step than the 128-row tile kernels at batches of 128 to 1024
    const int w1_rows = (int)debug_value(DBG_GEMM_W1_ROWS);
    if (a.dtype == ADCDIFF_BF16 && p.m_valid >= 1 && p.ksplit == 1 && (a.tile_hint == 0 || a.tile_hint == 32 || a.tile_hint == 325 || a.tile_hint == 327)) {
        int gmax = 0;
        for (int g = 0; g < a.n_gn; ++g) gmax = a.gn[g].group_size > gmax ? a.gn[g].group_size : gmax;
        const int bn = gmax <= 32 ? 32 : gmax <= 64 ? 64 : 128;
        const bool r32 = p.m_valid <= 32;
        // by default not for launches with thousands of such tiles (the [1000, 9984] time-projection table of a training step):
        // there the 128-row tile kernels' reuse of W wins
        // a 64-channel group is summed inside one wave here, as in the tile kernels whose wave tile is 64 columns wide; where N is
        // not a multiple of 128 those are not used at any batch size (the 128 x 64 tile merges two halves of 32, which differs
        // by a last bit in rare elements), so such launches stay on the tile kernels
        const bool same_sums = bn != 64 || a.N % 128 == 0;
        const bool take = same_sums && (a.tile_hint != 0 || (w1_default && p.m_valid <= w1_rows && (long)((p.m_valid + 63) / 64) * (a.N / bn) <= 1024));
        // the variant with loader waves where it is the faster one: 32 rows and tiles of 64 or 128 columns (-10 % / -19 % per launch);
        // with 32 columns it loses.  325 = that variant wherever it exists, 32 = never (measurement aids)
        const bool cl_default = debug_value(DBG_GEMM_CL) != 0;
        if (take && a.N % bn == 0) {
            // at most 64 live rows and a scratch buffer from the caller: K split over workgroups, S per output tile, so that the
            // layer's weights stream through (about) every CU instead of N / bn of them.  S: enough for 256 workgroups, at least
            // two K tiles each, at most what one workgroup can hold in flight, and what the scratch buffer holds.
            // Where it pays (device time per launch at 32 rows, one-wave / loader-wave kernel against this one): the layers that
            // stream at least 4 MB of weights through tiles of 32 or 64 columns -- N = K = 2048 12.4 -> 10.3 us, N = 1024 with
            // K = 4096 / 5120 14.9 -> 9.3 / 21.5 -> 14.6 us; it loses about 1 us on the small layers (the exchange) and changes
            // nothing for the 128-column tiles (22 us either way); tile_hint 327 forces it.
            // (with 32-column tiles the exchange is one batch of 16-byte loads, and the split also wins on the 512- and 256-wide layers with
            // K >= 1280: N = 512 / K = 2560 12.7 -> 9.4 us, N = 512 / K = 2048 10.2 -> 9.4 us, N = 256 / K = 1280 8.2 -> 7.2 us; it loses below 20 K tiles)
            const bool pays = (bn <= 64 && p.nt_total >= 32 && (long long)a.N * ktot >= (1LL << 22)) ||
                              (bn == 32 && p.nt_total >= 20 && (long long)a.N * ktot >= (1LL << 18));
            if (a.splitk_scratch && p.m_valid <= 64 && (a.tile_hint == 327 || (a.tile_hint == 0 && pays)) && debug_value(DBG_GEMM_SPLITK) != 0) {
                const int bm = r32 ? 32 : 64, ntn = a.N / bn;
                int S = 256 / ntn;
                if (S > p.nt_total / 2) S = p.nt_total / 2;
                if (S > 16) S = 16;
                const long long need = 4096 + (long long)S * bm * a.N * 4;
                if (S >= 2 && ntn <= 1024 && need <= a.splitk_scratch_bytes && aligned16(a.splitk_scratch)) {
                    p.sk_cnt = reinterpret_cast<unsigned*>(a.splitk_scratch);
                    p.sk_part = reinterpret_cast<float*>(reinterpret_cast<char*>(a.splitk_scratch) + 4096);
                    if (bn == 32) return r32 ? launch_w1s<32, 32>(p, S, stream) : launch_w1s<64, 32>(p, S, stream);
                    if (bn == 64) return r32 ? launch_w1s<32, 64>(p, S, stream) : launch_w1s<64, 64>(p, S, stream);
                    return r32 ? launch_w1s<32, 128>(p, S, stream) : launch_w1s<64, 128>(p, S, stream);
                }
            }
            if (a.tile_hint == 325 || (a.tile_hint == 0 && cl_default)) {
                if (bn == 32 && a.tile_hint == 325) return r32 ? launch_cl<32, 32>(p, stream) : launch_cl<64, 32>(p, stream);
                // 64-row tiles: while every workgroup gets a CU of its own (their rings fill a CU's LDS; with 16 row tiles at
                // batch 1024 the one-wave kernel, two workgroups per CU, is the faster one: 17.7 against 19.8 ms DDIM-50)
                const bool one_round = a.tile_hint == 325 || (long)((p.m_valid + 63) / 64) * (a.N / bn) <= 256;
                if (bn == 64 && (r32 || one_round)) return r32 ? launch_cl<32, 64>(p, stream) : launch_cl<64, 64>(p, stream);
                // the two halves of a 128-channel group meet inside a 16-lane row
                if (bn == 128 && (r32 || one_round)) return r32 ? launch_cl<32, 128>(p, stream) : launch_cl<64, 128>(p, stream);
            }
            if (bn == 32) return r32 ? launch_w1<32, 32>(p, stream) : launch_w1<64, 32>(p, stream);
            if (bn == 64) return r32 ? launch_w1<32, 64>(p, stream) : launch_w1<64, 64>(p, stream);
            return r32 ? launch_w1<32, 128>(p, stream) : launch_w1<64, 128>(p, stream);
        }
    }
    // few workgroups: nothing else on the CU hides a tile's load latency, so take the three-stage ring
    const int bn = small ? 64 : wide ? 128 : 64, bm = small || mid ? 64 : 128;
    const long blocks = (long)(a.M / bm) * (a.N / bn) * p.ksplit;
    const long ring_max = (long)debug_value(DBG_GEMM_RING_MAX);
    const bool ring = a.tile_hint == 3 || (a.tile_hint != 1 && blocks <= ring_max);
    if (a.dtype == ADCDIFF_BF16) {
        if (small) return launch_any<bf16_t, 64, 64>(p, ring, stream);
        if (mid) return launch_any<bf16_t, 64, 128>(p, ring, stream);
        return wide ? launch_any<bf16_t, 128, 128>(p, ring, stream) : launch_any<bf16_t, 128, 64>(p, ring, stream);
    }
    if (small) return launch_any<float, 64, 64>(p, ring, stream);
    if (mid) return launch_any<float, 64, 128>(p, ring, stream);
    return wide ? launch_any<float, 128, 128>(p, ring, stream) : launch_any<float, 128, 64>(p, ring, stream);
}

}  // namespace adcdiff

extern "C" int adcdiff_gemm(const adcdiff_gemm_args* args, adcdiff_stream_t stream) {
    if (!args) return adcdiff::fail(ADCDIFF_E_BADARG, "gemm: null args");
    return adcdiff::gemm_dispatch(*args, adcdiff::as_stream(stream));
}
