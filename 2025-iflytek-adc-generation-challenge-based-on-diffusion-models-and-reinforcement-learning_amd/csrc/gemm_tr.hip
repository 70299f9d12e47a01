// bf16 GEMMs whose operands are stored CONTRACTION-MAJOR, for the backward pass of y = x W^T:
//
//   d(x)[m, k] = sum_n d(y)[m, n] * W[n, k]         A = d(y) row-major (contraction index contiguous), B = W as it is stored
//   d(W)[n, k] = sum_m d(y)[m, n] * x[m, k]         A = d(y) and B = x as they are stored (contraction = the batch rows)
//
// The NT kernel of gemm.hip needs both operands with the contraction index contiguous, which costs a transposed copy of
// every weight and activation per step.  Here a contraction-major operand tile (64 contraction rows x 128 output
// columns, 256-byte rows) goes global -> LDS by LDS-DMA as it lies in memory and is read back with
// ds_read_b64_tr_b16, which hands each lane four consecutive CONTRACTION elements of its output column: two such reads
// make the 16x16x32 operand that the NT kernel gets from one ds_read_b128, in the same k order, so the MFMA chain and
// the float32 result are those of the NT kernel on transposed copies.
//
// LDS image of a contraction-major tile: 16-byte chunk ch of row r sits at 256 r + 16 (ch ^ f(r)),
// f(r) = ((r & 3) << 2) | ((r >> 2) & 3): the two 4-row blocks that the lanes of a 32-lane half read (8 rows apart, same
// columns) then fall on different banks.  The XOR is applied to the SOURCE address of the LDS-DMA (the image stays
// lane-linear: a 1 KiB piece = 4 rows).  A row-major A tile uses the NT kernel's image (128-byte rows, chunk ^ (row >> 1) & 7).
//
// 128 x 128 tile, 4 waves (2 x 2, 64 x 64 per wave), K step 64, three-stage ring with one barrier per step (the NT
// kernel's ring).  Output: float32, written straight from the accumulator layout; grid.y slices the contraction, every
// slice writing its own partial output (summed by the caller in slice order).
#include <cstdlib>

#include "unet_plan.h"

namespace adcdiff {

struct TrParams {
    const bf16_t* A;
    long lda;
    const bf16_t* B;
    long ldb;
    float* C;
    long ldc;
    int ntn;            // column tiles
    int nt;             // K tiles (64) per slice
    long a_slice;       // element offset of A between consecutive slices
    long b_slice;
    long c_slice;
};

typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int tr_xor(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

// the 16x16x32 operand of output columns c0 .. c0 + 15 (c0 a multiple of 16), contraction rows k0 .. k0 + 31 of a
// contraction-major tile: lane (g = l >> 4, t = l & 15) receives rows k0 + 8 g .. + 7 of column c0 + t
__device__ __forceinline__ bf16x8 tr_fragment(const char* tile, int k0, int c0, int lane) {
    const int g = lane >> 4, t = lane & 15, q = t >> 2, pp = t & 3;
    const int r1 = k0 + 8 * g + q, r2 = r1 + 4;
    const int ch = (c0 >> 3) + (pp >> 1);
    const char* a1 = tile + 256 * r1 + 16 * (ch ^ tr_xor(r1)) + 8 * (pp & 1);
    const char* a2 = tile + 256 * r2 + 16 * (ch ^ tr_xor(r2)) + 8 * (pp & 1);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a1);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a2);
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, both);
}

// NSTAGE = 3: one workgroup per CU with two tiles in flight; NSTAGE = 2 (64 KiB): two workgroups share a CU and one computes
// while the other waits at its barrier -- for launches with more workgroups than CUs.
// one 128 x 128 tile: workgroup `bid` of the `nwg` of its problem, contraction slice `slice`
template <bool TA, int NSTAGE>
__device__ __forceinline__ void tr_tile(const TrParams& p, int bid, int nwg, int slice, char* smem) {
    constexpr int STAGE_BYTES = 32768, A_BYTES = 16384, AHEAD = NSTAGE - 1, P = 8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    if ((nwg & 7) == 0) bid = (bid & 7) * (nwg >> 3) + (bid >> 3);
    const int m0 = (bid / p.ntn) * 128, n0 = (bid % p.ntn) * 128;
    const bf16_t* const A = p.A + (long)slice * p.a_slice;
    const bf16_t* const B = p.B + (long)slice * p.b_slice;

    // staging: this wave's four 1 KiB pieces of each operand tile
    const char* a_src[4];
    const char* b_src[4];
    long a_step, b_step = 64 * p.ldb * 2;   // bytes per K tile
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int q = wave * 4 + j;
        {
            const int kr = 4 * q + (lane >> 4), pc = lane & 15;
            b_src[j] = reinterpret_cast<const char*>(B + (long)kr * p.ldb + n0) + 16 * (pc ^ tr_xor(kr));
            if constexpr (TA) a_src[j] = reinterpret_cast<const char*>(A + (long)kr * p.lda + m0) + 16 * (pc ^ tr_xor(kr));
        }
        if constexpr (!TA) {
            const int row = 8 * q + (lane >> 3);
            a_src[j] = reinterpret_cast<const char*>(A + (long)(m0 + row) * p.lda) + 16 * ((lane & 7) ^ ((row >> 1) & 7));
        }
    }
    a_step = TA ? 64 * p.lda * 2 : 128;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const int nt = p.nt;
    int it = 0;
    auto issue = [&](int slot) {
        const unsigned base = lds0 + slot * STAGE_BYTES + wave * 4096;
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16_untracked(a_src[j] + (long)it * a_step, base + j * 1024);
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16_untracked(b_src[j] + (long)it * b_step, base + A_BYTES + j * 1024);
        ++it;
    };
    const int frow = lane & 15, fchunk = lane >> 4;
    auto compute = [&](const char* sa, const char* sb) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[4], bfr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if constexpr (TA) {
                    af[i] = tr_fragment(sa, 32 * ks, wm * 64 + i * 16, lane);
                } else {
                    const int row = wm * 64 + i * 16 + frow, c = 4 * ks + fchunk;
                    af[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(sa + row * 128 + ((c ^ ((row >> 1) & 7)) << 4)));
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) bfr[j] = tr_fragment(sb, 32 * ks, wn * 64 + j * 16, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
    };

#pragma unroll
    for (int k = 0; k < AHEAD; ++k)
        if (k < nt) issue(k);
    int slot = 0, free_slot = AHEAD;
    for (int t = 0; t < nt; ++t) {
        const int later = nt - 1 - t;
        wait_tiles_in_flight<P>(later < AHEAD - 1 ? later : AHEAD - 1);
        __builtin_amdgcn_s_barrier();          // tile t has landed for every wave; nobody still reads tile t - 1
        if (t + AHEAD < nt) issue(free_slot);
        compute(smem + slot * STAGE_BYTES, smem + slot * STAGE_BYTES + A_BYTES);
        free_slot = slot;
        slot = slot == NSTAGE - 1 ? 0 : slot + 1;
    }

    // C/D layout of the 16x16 tile: col = l & 15, row = 4 * (l >> 4) + r
    float* const C = p.C + (long)slice * p.c_slice;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long row = m0 + wm * 64 + i * 16 + 4 * (lane >> 4) + r;
                const long col = n0 + wn * 64 + j * 16 + (lane & 15);
                C[row * p.ldc + col] = acc[i][j][r];
            }
}

template <bool TA, int NSTAGE>
__global__ __launch_bounds__(256, NSTAGE == 3 ? 1 : 2) void gemm_tr_kernel(const TrParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    tr_tile<TA, NSTAGE>(p, (int)blockIdx.x, (int)gridDim.x, (int)blockIdx.y, smem);
}

// Several d(W) = d(y)^T x products in ONE launch: the weight gradients of a backward pass feed nothing but the optimizer, so
// they can all wait for its end, where one launch with thousands of tiles fills the chip (a product on its own has 4 to 256
// tiles and a fixed cost per launch).  Workgroup -> problem by binary search in the prefix table of tile counts.  The table
// travels as the kernel's argument (48 bytes per problem, 3.4 KB): no copy to the device, nothing to keep alive.
struct TrItem {
    const bf16_t* A;
    const bf16_t* B;
    float* C;
    int lda, ldb, ldc;
    int ntn, nt, first_tile;
};
struct TrGroup {
    TrItem items[TR_MAX_GROUP];
    int nprob;
};
static_assert(sizeof(TrGroup) <= 3800, "the grouped launch's table must fit the kernel argument segment");
__global__ __launch_bounds__(256, 2) void gemm_tr_grouped_kernel(const TrGroup g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = (int)blockIdx.x;
    int lo = 0, hi = g.nprob - 1;
    while (lo < hi) {   // last problem whose first tile <= b (workgroup-uniform)
        const int mid = (lo + hi + 1) >> 1;
        if (g.items[mid].first_tile <= b) lo = mid; else hi = mid - 1;
    }
    const TrItem it = g.items[lo];
    TrParams p;
    p.A = it.A; p.lda = it.lda; p.B = it.B; p.ldb = it.ldb; p.C = it.C; p.ldc = it.ldc;
    p.ntn = it.ntn; p.nt = it.nt; p.a_slice = 0; p.b_slice = 0; p.c_slice = 0;
    tr_tile<true, 2>(p, b - it.first_tile, 1, 0, smem);
}

template <bool TA, int NSTAGE>
static int launch_tr(const TrParams& p, dim3 grid, hipStream_t stream) {
    constexpr int smem = NSTAGE * 32768;
    // once per process, thread-safe (initialisation of a function-local static)
    static const hipError_t attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tr_kernel<TA, NSTAGE>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (attr_err != hipSuccess) return fail(ADCDIFF_E_LAUNCH, "gemm_tr: hipFuncSetAttribute: %s", hipGetErrorString(attr_err));
    hipLaunchKernelGGL((gemm_tr_kernel<TA, NSTAGE>), grid, dim3(256), smem, stream, p);
    return ADCDIFF_OK;
}

static int tr_params(bool ta, const void* A, long lda, const void* B, long ldb, float* C, long ldc, int M, int N, int K, int slices, long c_slice,
                     TrParams* out) {
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || M % 128 || N % 128 || slices < 1 || K % (64 * slices) != 0)
        return fail(ADCDIFF_E_BADARG, "gemm_tr: M=%d N=%d must be multiples of 128, K=%d of 64 x slices (%d)", M, N, K, slices);
    if (!aligned16(A) || !aligned16(B) || !aligned16(C) || lda % 8 || ldb % 8 || ldb < N || lda < (ta ? M : K) || ldc < N)
        return fail(ADCDIFF_E_BADARG, "gemm_tr: operands must be 16-byte aligned with leading dimensions that are multiples of 8 and cover the rows");
    if (slices > 1 && c_slice < (long)(M - 1) * ldc + N) return fail(ADCDIFF_E_BADARG, "gemm_tr: c_slice=%ld does not hold one output", c_slice);
    TrParams& p = *out;
    p.A = reinterpret_cast<const bf16_t*>(A); p.lda = lda;
    p.B = reinterpret_cast<const bf16_t*>(B); p.ldb = ldb;
    p.C = C; p.ldc = ldc;
    p.ntn = N / 128;
    const int kslice = K / slices;
    p.nt = kslice / 64;
    p.a_slice = ta ? (long)kslice * lda : (long)kslice;
    p.b_slice = (long)kslice * ldb;
    p.c_slice = slices > 1 ? c_slice : 0;
    return ADCDIFF_OK;
}

// The queued products C_i[M_i, N_i] = A_i^T B_i (A_i [K, M_i], B_i [K, N_i] row-major, K_i contraction rows) in one launch.
int gemm_tr_grouped(const TrProblem* probs, int nprob, hipStream_t stream) {
    if (nprob <= 0) return ADCDIFF_OK;
    if (nprob > TR_MAX_GROUP) return fail(ADCDIFF_E_BADARG, "gemm_tr_grouped: %d problems (at most %d)", nprob, TR_MAX_GROUP);
    // the largest products first: the tail of the launch is then made of small tiles
    int order[TR_MAX_GROUP];
    for (int i = 0; i < nprob; ++i) order[i] = i;
    for (int i = 1; i < nprob; ++i)
        for (int j = i; j > 0 && (double)probs[order[j]].M * probs[order[j]].N * probs[order[j]].K > (double)probs[order[j - 1]].M * probs[order[j - 1]].N * probs[order[j - 1]].K; --j) {
            const int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t;
        }
    TrGroup g;
    int tiles = 0;
    double flops = 0.0, bytes = 0.0;
    for (int i = 0; i < nprob; ++i) {
        const TrProblem& q = probs[order[i]];
        TrParams p;
        const int rc = tr_params(true, q.A, q.lda, q.B, q.ldb, q.C, q.ldc, q.M, q.N, q.K, 1, 0, &p);
        if (rc != ADCDIFF_OK) return rc;
        if (q.lda > 0x7fffffffL || q.ldb > 0x7fffffffL || q.ldc > 0x7fffffffL) return fail(ADCDIFF_E_BADARG, "gemm_tr_grouped: leading dimension too large");
        g.items[i] = TrItem{p.A, p.B, p.C, (int)p.lda, (int)p.ldb, (int)p.ldc, p.ntn, p.nt, tiles};
        tiles += (q.M / 128) * (q.N / 128);
        flops += 2.0 * q.M * q.N * q.K;
        bytes += 2.0 * ((double)q.M * q.K + (double)q.N * q.K) + 4.0 * q.M * q.N;
    }
    g.nprob = nprob;
    constexpr int smem = 2 * 32768;
    static const hipError_t attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tr_grouped_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (attr_err != hipSuccess) return fail(ADCDIFF_E_LAUNCH, "gemm_tr: hipFuncSetAttribute: %s", hipGetErrorString(attr_err));
    TimingScope ts(ADCDIFF_KIND_GEMM_BF16, tiles * 128, 128, 0, flops, bytes, stream);
    hipLaunchKernelGGL(gemm_tr_grouped_kernel, dim3((unsigned)tiles), dim3(256), smem, stream, g);
    return check_launch("gemm_tr_grouped");
}

// C[M, N] (+ slices) = A * B over K.  ta: A is [K, M] row-major (lda >= M), else [M, K] row-major (lda >= K); B is [K, N]
// row-major.  slices > 1: slice s takes K / slices contraction rows and writes C + s * c_slice.
int gemm_tr(bool ta, const void* A, long lda, const void* B, long ldb, float* C, long ldc, int M, int N, int K, int slices, long c_slice,
            hipStream_t stream) {
    TrParams p;
    {
        const int rc0 = tr_params(ta, A, lda, B, ldb, C, ldc, M, N, K, slices, c_slice, &p);
        if (rc0 != ADCDIFF_OK) return rc0;
    }
    const dim3 grid((unsigned)((M / 128) * p.ntn), (unsigned)slices);
    TimingScope ts(ADCDIFF_KIND_GEMM_BF16, M, N, K, 2.0 * M * N * K, 2.0 * ((double)M * K + (double)N * K) + 4.0 * M * N * slices, stream);
    const bool two_ok = debug_value(DBG_GEMM_RING2) != 0;
    const bool two = two_ok && (long)grid.x * grid.y > 256;
    int rc;
    if (two) rc = ta ? launch_tr<true, 2>(p, grid, stream) : launch_tr<false, 2>(p, grid, stream);
    else rc = ta ? launch_tr<true, 3>(p, grid, stream) : launch_tr<false, 3>(p, grid, stream);
    if (rc != ADCDIFF_OK) return rc;
    return check_launch("gemm_tr");
}

}  // namespace adcdiff

// Test hook for the contraction-major GEMM (bf16 operands, float32 output); see gemm_tr above.
extern "C" int adcdiff_gemm_tr(int ta, const void* A, long long lda, const void* B, long long ldb, float* C, long long ldc, int M, int N,
                               int K, int slices, long long c_slice, adcdiff_stream_t stream) {
    return adcdiff::gemm_tr(ta != 0, A, (long)lda, B, (long)ldb, C, (long)ldc, M, N, K, slices, (long)c_slice, adcdiff::as_stream(stream));
}
