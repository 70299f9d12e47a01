// Training step of the denoiser: forward with saved activations, noise-prediction loss, and the backward
// pass (diffusion_model.py:165-184 p_losses, :401-471 train_step) on the same MFMA GEMM as inference.
//
// Forward: the unfused path of unet.hip with a tape (float32 layer outputs stay in the workspace) and
// training-mode dropout in the norm2 activation (counter RNG keyed by layer / row / column).
// Backward of y = x W^T (+ b):   dX = dY W   is the NT GEMM  [dY] x [W^T stored (K, N)],
//                                 dW = dY^T X is the NT GEMM  [dY^T stored (N, M)] x [X^T stored (K, M)],
// so every Linear costs two GEMM launches plus transposes (cast_transpose_kernel, which also yields the bias
// gradient as its column sums).  GroupNorm+SiLU(+dropout) backward recomputes the statistics from the saved
// pre-norm tensor (gn_silu_bwd_kernel).  The time branch is differentiated through the per-timestep tables:
// d(linear1 output) rows are scatter-added into d_tb_table[t_row], and the table-building GEMMs (M = T) are
// back-propagated once per step.
#include <cstdlib>

#include "unet_plan.h"

namespace adcdiff {

// ------------------------------------------------------------------ kernels
// dst_rm[r, c] = (OutT) src[r, c] (r < rows), dst_t[c, r] = (OutT) src[r, c] with zeros for rows <= r < rows_pad,
// colsum[c] += sum_r src[r, c].  64 x 64 tile per workgroup through LDS.
template <typename InT, typename OutT>
__global__ __launch_bounds__(256) void cast_transpose_kernel(const InT* __restrict__ src, long lds_, int rows, int rows_pad, int cols,
                                                             OutT* __restrict__ dst_rm, long ldrm, OutT* __restrict__ dst_t, long ldt,
                                                             float* __restrict__ colsum) {
    __shared__ float tile[64][65];
    const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 64 x 4
    float csum = 0.0f;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i;
        float v = 0.0f;
        if (r < rows) {
            v = (float)src[(long)r * lds_ + c0 + tx];
            if (dst_rm) dst_rm[(long)r * ldrm + c0 + tx] = (OutT)v;
        }
        tile[i][tx] = v;
        csum += v;
    }
    __syncthreads();
    if (dst_t) {
        for (int i = ty; i < 64; i += 4) {
            const int r = r0 + tx;   // transposed: tx walks the source rows
            if (r < rows_pad) dst_t[(long)(c0 + i) * ldt + r] = (OutT)tile[tx][i];
        }
    }
    if (colsum) {
        __syncthreads();
        tile[ty][tx] = csum;     // rows 0..3 of the tile are free after the transpose reads
        __syncthreads();
        if (ty == 0) atomicAdd(colsum + c0 + tx, (tile[0][tx] + tile[1][tx]) + (tile[2][tx] + tile[3][tx]));
    }
}

// Backward of a = dropout(SiLU(GroupNorm(u))), 4 columns x rpt consecutive rows per thread:
//   d_u = rstd * (dxh - mean_g(dxh) - xh * mean_g(dxh * xh)),  dxh = dA * keep * silu'(n) * gamma,  n = xh * gamma + beta
//   dgamma[c] += sum_rows dA * keep * silu'(n) * xh,   dbeta[c] += sum_rows dA * keep * silu'(n)
// dst = (accumulate ? dst : 0) + d_u + add.
// A workgroup is c4t = min(C / 4, 256) column lanes x 256 / c4t row lanes (no idle lanes for narrow layers); the host picks
// rpt (8..32) so that the grid has enough workgroups; rows are taken two at a time so that four loads are in flight.
template <int LPG>
__global__ __launch_bounds__(256) void gn_silu_bwd_kernel(const float* __restrict__ u, long ldu, const float* __restrict__ dA, long ldda,
                                                          int rows, int c4t, int rpt, int lanes_per_group, float inv_gs,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                          float drop_p, uint64_t drop_seed, uint32_t drop_layer, int drop_col0,
                                                          const float* __restrict__ add, long ldadd, float* __restrict__ dst, long lddst,
                                                          int accumulate, float* __restrict__ part, long part_stride,
                                                          bf16_t* __restrict__ dst_op, long lddo, int want_colsum) {
    __shared__ float red[3][4 * 256];
    const int cq = threadIdx.x % c4t, ry = threadIdx.x / c4t, ryc = 256 / c4t;
    const int col = (blockIdx.x * c4t + cq) * 4;
    const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + col);
    const f32x4 b = *reinterpret_cast<const f32x4*>(beta + col);
    f32x4 dg = {0, 0, 0, 0}, db = {0, 0, 0, 0}, cs = {0, 0, 0, 0};
    const int rbeg = (blockIdx.y * ryc + ry) * rpt;
    const int rend = rbeg + rpt < rows ? rbeg + rpt : rows;
    for (int r0 = rbeg; r0 < rend; r0 += 2) {
        // the second row of the pair repeats the first when the range ends on an odd row; its results are discarded
        const bool two = r0 + 1 < rend;
        const int rr[2] = {r0, two ? r0 + 1 : r0};
        f32x4 xs[2], das[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            xs[k] = *reinterpret_cast<const f32x4*>(u + (long)rr[k] * ldu + col);
            das[k] = *reinterpret_cast<const f32x4*>(dA + (long)rr[k] * ldda + col);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int r = rr[k];
            const f32x4 x = xs[k], da = das[k];
            const float s = lanes_allsum<LPG>((x[0] + x[1]) + (x[2] + x[3]), lanes_per_group);
            const float mean = s * inv_gs;
            const f32x4 d = {x[0] - mean, x[1] - mean, x[2] - mean, x[3] - mean};
            const float v = lanes_allsum<LPG>((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]), lanes_per_group);
            const float rstd = 1.0f / __builtin_sqrtf(v * inv_gs + eps);
            f32x4 keep = {1.0f, 1.0f, 1.0f, 1.0f};
            if (drop_p > 0.0f) keep = dropout_keep4(drop_seed, drop_layer, (uint64_t)r, (uint32_t)((drop_col0 + col) >> 2), drop_p);
            f32x4 xh, dxh;
            float s1 = 0.0f, s2 = 0.0f;
            const bool count = k == 0 || two;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                xh[i] = d[i] * rstd;
                const float n = xh[i] * g[i] + b[i];
                // v_exp_f32 / v_rcp_f32 (1 ulp each) instead of expf and a full division: the kernel is bound by its arithmetic,
                // and a gradient does not need the forward pass's last bit
                const float sig = __builtin_amdgcn_rcpf(1.0f + __expf(-n));
                const float dn = da[i] * keep[i] * (sig * (1.0f + n * (1.0f - sig)));
                if (count) { dg[i] += dn * xh[i]; db[i] += dn; }
                dxh[i] = dn * g[i];
                s1 += dxh[i];
                s2 += dxh[i] * xh[i];
            }
            s1 = lanes_allsum<LPG>(s1, lanes_per_group); s2 = lanes_allsum<LPG>(s2, lanes_per_group);
            s1 *= inv_gs; s2 *= inv_gs;
            if (count) {
                f32x4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = rstd * (dxh[i] - s1 - xh[i] * s2);
                if (add) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(add + (long)r * ldadd + col);
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] += a[i];
                }
                float* dp = dst + (long)r * lddst + col;
                if (accumulate) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(dp);
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] += a[i];
                }
                *reinterpret_cast<f32x4*>(dp) = o;
                // the finished value also in bf16 (the operand of the two backward products it enters) and into its column sums
                // (the gradient of the bias of the layer that produced u's tensor)
                if (dst_op) {
                    bf16x4 ob;
#pragma unroll
                    for (int i = 0; i < 4; ++i) ob[i] = (bf16_t)o[i];
                    *reinterpret_cast<bf16x4*>(dst_op + (long)r * lddo + col) = ob;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) cs[i] += o[i];
            }
        }
    }
    // d(gamma), d(beta): sum the workgroup's row lanes in LDS (fixed order), then one partial row per workgroup; the column
    // sums over the partial rows are taken by colsum_partials_kernel -- no atomics (thousands of float atomics on the few
    // cache lines of a [C] vector serialise in L2 and cost more than the rest of this kernel)
#pragma unroll
    for (int i = 0; i < 4; ++i) { red[0][(ry * c4t + cq) * 4 + i] = dg[i]; red[1][(ry * c4t + cq) * 4 + i] = db[i]; red[2][(ry * c4t + cq) * 4 + i] = cs[i]; }
    __syncthreads();
    if (ry == 0) {
        f32x4 sg = {0, 0, 0, 0}, sb = {0, 0, 0, 0}, sc = {0, 0, 0, 0};
        for (int y = 0; y < ryc; ++y)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                sg[i] += red[0][(y * c4t + cq) * 4 + i]; sb[i] += red[1][(y * c4t + cq) * 4 + i]; sc[i] += red[2][(y * c4t + cq) * 4 + i];
            }
        float* pr = part + (long)blockIdx.y * part_stride + col;   // a partial row = (d gamma | d beta | column sums), a third of the stride each
        *reinterpret_cast<f32x4*>(pr) = sg;
        *reinterpret_cast<f32x4*>(pr + part_stride / 3) = sb;
        if (want_colsum) *reinterpret_cast<f32x4*>(pr + 2 * (part_stride / 3)) = sc;
    }
}

// The column sums over the partial rows of EVERY gn_silu_bwd launch of a backward pass, in one launch at its end (they are
// parameter gradients: nothing in the pass reads them).  An item = one launch's partial rows; three vectors per item
// (d gamma, d beta, the column sums), each added to up to two destinations.  A workgroup takes 64 columns of one vector of one
// item: 16 row lanes that each sum every 16th partial row in index order, combined through LDS in lane order (deterministic).
struct ColsumItem {
    const float* part;
    long stride;          // floats per partial row (3 C)
    int nrows, C;
    float* dst[3][2];     // null = skip
    int first_block;      // prefix of 3 * ceil(C / 64) over the earlier items
};
constexpr int COLSUM_MAX_ITEMS = 32;
struct ColsumTable {      // travels as the kernel's argument (2.6 KB): no copy to the device, nothing to keep alive
    ColsumItem items[COLSUM_MAX_ITEMS];
    int n;
};

__global__ __launch_bounds__(256) void colsum_batched_kernel(const ColsumTable tab) {
    __shared__ float red[16][64];
    const ColsumItem* const items = tab.items;
    int lo = 0, hi = tab.n - 1;
    const int blk = (int)blockIdx.x;
    while (lo < hi) {   // last item whose first block <= blk (workgroup-uniform)
        const int mid = (lo + hi + 1) >> 1;
        if (items[mid].first_block <= blk) lo = mid; else hi = mid - 1;
    }
    const ColsumItem it = items[lo];
    const int nb = (it.C + 63) / 64;
    const int local = blk - it.first_block, which = local / nb, cb = local % nb;
    float* const d0 = it.dst[which][0];
    float* const d1 = it.dst[which][1];
    if (!d0 && !d1) return;
    const int cx = threadIdx.x & 15, ly = threadIdx.x >> 4;
    const int c = cb * 64 + cx * 4;
    f32x4 s = {0, 0, 0, 0};
    if (c < it.C) {
        const float* src = it.part + which * (it.stride / 3) + c;
        for (int r = ly; r < it.nrows; r += 16) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(src + (long)r * it.stride);
            s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) red[ly][cx * 4 + i] = s[i];
    __syncthreads();
    if (threadIdx.x < 64 && cb * 64 + threadIdx.x < it.C) {
        float t = 0.0f;
#pragma unroll
        for (int y = 0; y < 16; ++y) t += red[y][threadIdx.x];
        if (d0) d0[cb * 64 + threadIdx.x] += t;
        if (d1) d1[cb * 64 + threadIdx.x] += t;
    }
}

// dx = dy * silu'(x), elementwise over [rows, cols] (all leading dimensions = cols).
__global__ __launch_bounds__(256) void silu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, long n) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const float v = x[e];
    const float sig = 1.0f / (1.0f + expf(-v));
    dx[e] = dy[e] * (sig * (1.0f + v * (1.0f - sig)));
}

// table[t[r], col0 + c] += src[r, c]
__global__ __launch_bounds__(256) void scatter_add_rows_kernel(const float* __restrict__ src, long lds_, const int64_t* __restrict__ t, int rows,
                                                               int cols, float* __restrict__ table, long ldt, int col0, int table_rows) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)rows * cols) return;
    const long r = e / cols;
    const int c = (int)(e - r * cols);
    long tr = t[r];
    tr = tr < 0 ? 0 : tr >= table_rows ? table_rows - 1 : tr;   // as the forward pass clamps it
    atomicAdd(table + tr * ldt + col0 + c, src[r * lds_ + c]);
}

// d_pred = 2 * scale * (pred - target) / n;  *loss += scale * sum((target - pred)^2) / n   (F.mse_loss, mean reduction)
__global__ __launch_bounds__(256) void mse_grad_kernel(const float* __restrict__ pred, long ldp, const float* __restrict__ target, long ldt_,
                                                       float* __restrict__ dpred, long ldd, int rows, int cols, float scale, float* loss) {
    __shared__ float part[4];
    const long n = (long)rows * cols;
    float sq = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; ++k) {   // 4096 elements per workgroup: one atomic on the loss per 4096 elements
        const long e = ((long)blockIdx.x * 16 + k) * 256 + threadIdx.x;
        if (e < n) {
            const long r = e / cols;
            const int c = (int)(e - r * cols);
            const float d = pred[r * ldp + c] - target[r * ldt_ + c];
            dpred[r * ldd + c] = 2.0f * scale * d / (float)n;
            sq += d * d;
        }
    }
    for (int m = 32; m > 0; m >>= 1) sq += __shfl_xor(sq, m);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = sq;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss, scale * ((part[0] + part[1]) + (part[2] + part[3])) / (float)n);
}

// dst[r, c] += src[r, c] for r < rows (leading dimensions lds_ / ldd)
__global__ __launch_bounds__(256) void add_rows_kernel(const float* __restrict__ src, long lds_, float* __restrict__ dst, long ldd, int rows, int cols) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)rows * cols) return;
    const long r = e / cols;
    const int c = (int)(e - r * cols);
    dst[r * ldd + c] += src[r * lds_ + c];
}

// dz[r, j] = ds[r, j] * silu'(te_table[t_r, j] + cemb[r, j]): backward of s = SiLU(t_emb + c_emb) (diffusion_model.py:766, 859-862)
__global__ __launch_bounds__(256) void cond_silu_bwd_kernel(const float* __restrict__ te_table, int ldt, int table_rows, const int64_t* __restrict__ t_rows,
                                                            const float* __restrict__ cemb, int ldce, const float* __restrict__ ds,
                                                            float* __restrict__ dz, int rows, int H) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)rows * H) return;
    const long r = e / H;
    const int j = (int)(e - r * H);
    long t = (long)t_rows[r];
    t = t < 0 ? 0 : t >= table_rows ? table_rows - 1 : t;   // as the forward pass clamps it
    const float v = te_table[t * ldt + j] + cemb[r * ldce + j];
    const float sig = 1.0f / (1.0f + expf(-v));
    dz[e] = ds[e] * (sig * (1.0f + v * (1.0f - sig)));
}

__global__ __launch_bounds__(256) void copy_f32_kernel(const float* __restrict__ src, float* __restrict__ dst, long n, int accumulate) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e < n) dst[e] = (accumulate ? dst[e] : 0.0f) + src[e];
}

// ------------------------------------------------------------------ host helpers
template <typename InT>
static int cast_transpose(const InT* src, long lds_, int rows, int rows_pad, int cols, int out_dtype, void* dst_rm, long ldrm, void* dst_t,
                          long ldt, float* colsum, hipStream_t st) {
    if (cols % 64 != 0) return fail(ADCDIFF_E_BADARG, "cast_transpose: cols=%d must be a multiple of 64", cols);
    TimingScope ts(ADCDIFF_KIND_OTHER, rows, cols, 0, 0.0, (double)rows * cols * (sizeof(InT) + (out_dtype == ADCDIFF_BF16 ? 2.0 : 4.0) * ((dst_rm ? 1 : 0) + (dst_t ? 1 : 0))), st);
    const dim3 grid((unsigned)((rows_pad + 63) / 64), (unsigned)(cols / 64));
    if (out_dtype == ADCDIFF_BF16)
        hipLaunchKernelGGL((cast_transpose_kernel<InT, bf16_t>), grid, dim3(256), 0, st, src, lds_, rows, rows_pad, cols,
                           reinterpret_cast<bf16_t*>(dst_rm), ldrm, reinterpret_cast<bf16_t*>(dst_t), ldt, colsum);
    else
        hipLaunchKernelGGL((cast_transpose_kernel<InT, float>), grid, dim3(256), 0, st, src, lds_, rows, rows_pad, cols,
                           reinterpret_cast<float*>(dst_rm), ldrm, reinterpret_cast<float*>(dst_t), ldt, colsum);
    return check_launch("cast_transpose");
}

// float32 [rows, cols] -> bf16 row-major, column sums accumulated into colsum and copied to colsum2 (a bias that two layers
// share).  (A variant with 16-byte loads, both sums as atomics of one kernel and the per-timestep scatter folded in measured
// slower on the same box: atomics count in vmcnt in issue order, so loads wait for the atomics issued before them.)
static int cast_rows(const float* src, long lds_, int rows, int rows_pad, int cols, void* dst, long ldd, float* colsum, float* colsum2,
                     hipStream_t st) {
    TRY(cast_transpose<float>(src, lds_, rows, rows_pad, cols, ADCDIFF_BF16, dst, ldd, nullptr, 0, colsum, st));
    if (colsum2) {
        TimingScope ts(ADCDIFF_KIND_OTHER, 1, 0, 0, 0.0, (double)cols * 8.0, st);
        hipLaunchKernelGGL(copy_f32_kernel, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, st, colsum, colsum2, (long)cols, 0);
        return check_launch("copy_f32");
    }
    return ADCDIFF_OK;
}

// transposes an operand-typed [rows, cols] activation into [cols, rows_pad]
static int transpose_op(const void* src, long lds_, int rows, int rows_pad, int cols, int dtype, void* dst_t, hipStream_t st) {
    if (dtype == ADCDIFF_BF16)
        return cast_transpose<bf16_t>(reinterpret_cast<const bf16_t*>(src), lds_, rows, rows_pad, cols, dtype, nullptr, 0, dst_t, rows_pad, nullptr, st);
    return cast_transpose<float>(reinterpret_cast<const float*>(src), lds_, rows, rows_pad, cols, dtype, nullptr, 0, dst_t, rows_pad, nullptr, st);
}

// geometry of a gn_silu_bwd launch: column lanes per workgroup, rows per thread, partial rows
struct GnBwdGeom { int c4t, ryc, rpt, nby; };
static GnBwdGeom gn_bwd_geom(int rows, int C) {
    const int c4 = C / 4;
    GnBwdGeom g;
    g.c4t = c4 < 256 ? c4 : 256;
    g.ryc = 256 / g.c4t;
    g.rpt = 32;
    while (g.rpt > 8 && (long)(c4 / g.c4t) * ((rows + g.ryc * g.rpt - 1) / (g.ryc * g.rpt)) < 1024) g.rpt >>= 1;
    g.nby = (rows + g.ryc * g.rpt - 1) / (g.ryc * g.rpt);
    while (g.nby > 512) { g.rpt <<= 1; g.nby = (rows + g.ryc * g.rpt - 1) / (g.ryc * g.rpt); }   // bounds the partial rows of a huge batch
    return g;
}
static size_t gn_bwd_part_bytes(int rows, int C) { return round_up((size_t)gn_bwd_geom(rows, C).nby * 3 * C * 4, 256); }

struct ColsumQueue {
    ColsumTable tab;
    ColsumItem (&items)[COLSUM_MAX_ITEMS] = tab.items;
    int n = 0, blocks = 0;
    Bump parts{nullptr, 0};   // the partial rows of every launch stay until the batched column sums have run
};

// d(u) of a = dropout(SiLU(GroupNorm(u))) (see the kernel).  The partial rows of d gamma / d beta (and, with bias0, of the column
// sums of the finished dst, which are the bias gradient of the layer that produced u: added to bias0 and bias1) go to the queue's
// buffer and are summed by ONE launch at the end of the pass (colsum_flush).  dst_op: the finished dst in bf16 as well.
static int gn_silu_bwd(const float* u, long ldu, const float* dA, long ldda, int rows, int C, int groups, const float* gamma,
                       const float* beta, float drop_p, uint64_t drop_seed, uint32_t drop_layer, int drop_col0, const float* add,
                       long ldadd, float* dst, long lddst, int accumulate, float* dgamma, float* dbeta, ColsumQueue& cq, bool dry, hipStream_t st,
                       void* dst_op = nullptr, long lddo = 0, float* bias0 = nullptr, float* bias1 = nullptr) {
    const int gs = C / groups;
    if (gs < 4 || gs > 256 || (gs & (gs - 1)) != 0)
        return fail(ADCDIFF_E_BADARG, "training needs GroupNorm groups that are powers of two in 4..256 (got %d)", gs);
    const int c4 = C / 4;
    // column lanes per workgroup: all of them up to 256 (a power of two, so that it divides 256 and holds whole groups)
    if (C % 4 != 0 || (c4 <= 256 ? (c4 & (c4 - 1)) != 0 : c4 % 256 != 0))
        return fail(ADCDIFF_E_BADARG, "training needs a channel count that is a power of two up to 1024 or a multiple of 1024 (got %d)", C);
    const GnBwdGeom g = gn_bwd_geom(rows, C);
    if (dry) return ADCDIFF_OK;
    float* part = (float*)cq.parts.take((size_t)g.nby * 3 * C * 4);
    if (cq.parts.off > cq.parts.cap || cq.n >= COLSUM_MAX_ITEMS) return fail(ADCDIFF_E_WORKSPACE, "unet_backward: internal sizing error (GroupNorm partial rows)");
    TimingScope ts(ADCDIFF_KIND_OTHER, rows, C, 0, 0.0, (double)rows * C * (12.0 + (dst_op ? 2.0 : 0.0)), st);
    const dim3 grid((unsigned)(c4 / g.c4t), (unsigned)g.nby);
    // the group size as a template argument: the four group sums per row then run on the cross-lane data path (same bits)
#define ADCDIFF_GN_BWD(LPG)                                                                                                                 \
    hipLaunchKernelGGL(gn_silu_bwd_kernel<LPG>, grid, dim3(256), 0, st, u, ldu, dA, ldda, rows, g.c4t, g.rpt, gs / 4, 1.0f / (float)gs, gamma, \
                       beta, 1e-5f, drop_p, drop_seed, drop_layer, drop_col0, add, ldadd, dst, lddst, accumulate, part, (long)3 * C,          \
                       reinterpret_cast<bf16_t*>(dst_op), lddo, bias0 ? 1 : 0)
    switch (gs / 4) {
        case 1: ADCDIFF_GN_BWD(1); break;
        case 2: ADCDIFF_GN_BWD(2); break;
        case 4: ADCDIFF_GN_BWD(4); break;
        case 8: ADCDIFF_GN_BWD(8); break;
        case 16: ADCDIFF_GN_BWD(16); break;
        case 32: ADCDIFF_GN_BWD(32); break;
        default: ADCDIFF_GN_BWD(64); break;
    }
#undef ADCDIFF_GN_BWD
    ColsumItem& it = cq.items[cq.n++];
    it.part = part; it.stride = (long)3 * C; it.nrows = g.nby; it.C = C;
    it.dst[0][0] = dgamma; it.dst[0][1] = nullptr; it.dst[1][0] = dbeta; it.dst[1][1] = nullptr; it.dst[2][0] = bias0; it.dst[2][1] = bias1;
    it.first_block = cq.blocks;
    cq.blocks += 3 * ((C + 63) / 64);
    return check_launch("gn_silu_bwd");
}

// the queued column sums in one launch
static int colsum_flush(ColsumQueue& cq, hipStream_t st) {
    if (cq.n == 0) return ADCDIFF_OK;
    cq.tab.n = cq.n;
    TimingScope ts(ADCDIFF_KIND_OTHER, cq.n, 0, 0, 0.0, (double)cq.parts.off, st);
    hipLaunchKernelGGL(colsum_batched_kernel, dim3((unsigned)cq.blocks), dim3(256), 0, st, cq.tab);
    cq.n = 0; cq.blocks = 0;
    return check_launch("colsum_batched");
}

static int gemm_simple(int dt, const void* A, int lda, const void* W, int ldw, int M, int m_valid, int N, int K, float* out, int ldo,
                       hipStream_t st) {
    adcdiff_gemm_args g = gemm_args(dt, M, m_valid, N);
    g.nseg = 1; g.A[0] = A; g.lda[0] = lda; g.K[0] = K; g.W = W; g.ldw = ldw; g.out_f32 = out; g.ldo_f32 = ldo;
    return gemm_dispatch(g, st);
}

// dst[e] = sum over s of part[s * stride + e], slices in index order (n % 4 == 0, 16-byte aligned)
// (the partials are [rows, cols] contiguous; dst rows are ldd floats apart)
__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ part, int slices, long stride, long n4, float* __restrict__ dst,
                                                           int cols, long ldd) {
    const long q = (long)blockIdx.x * 256 + threadIdx.x;
    if (q >= n4) return;
    f32x4 acc = *reinterpret_cast<const f32x4*>(part + q * 4);
    for (int s = 1; s < slices; ++s) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(part + (long)s * stride + q * 4);
        acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3];
    }
    const long e = q * 4, r = e / cols, c = e - r * cols;
    *reinterpret_cast<f32x4*>(dst + r * ldd + c) = acc;
}

// K slices for a weight-gradient GEMM out[M, N] = A[M, K] W[N, K]^T whose K is the batch: enough that slices x tiles fill
// the chip, each slice at least 512 long.  Depends on the shapes only, so the sizing dry run takes the same decision.
static int dw_slices(int M, int N, int K) {
    const long tiles = (long)((M + 127) / 128) * ((N + 127) / 128);
    int s = 1;
    while (tiles * s < 256 && K % (s * 2 * 128) == 0 && K / (s * 2) >= 512) s *= 2;
    return s;
}

// d(weight)[M, N] = A[M, K] W[N, K]^T (both operands K-major), split over K when M x N alone leaves most CUs idle: the
// slices' partial products go to the workspace and are summed in slice order (deterministic, unlike atomics).
static int gemm_dw(int dt, const void* A, const void* W, int M, int N, int K, float* out, Bump& ws, bool dry, hipStream_t st) {
    const int slices = dw_slices(M, N, K);
    if (slices == 1) {
        if (dry) return ADCDIFF_OK;
        return gemm_simple(dt, A, K, W, K, M, 0, N, K, out, N, st);
    }
    const size_t keep = ws.off;
    float* part = (float*)ws.take((size_t)slices * M * N * 4);
    ws.off = keep;   // the partials are dead once summed; the next take may reuse them (the peak is recorded)
    if (dry) return ADCDIFF_OK;
    adcdiff_gemm_args g = gemm_args(dt, M, 0, N);
    g.nseg = 1; g.A[0] = A; g.lda[0] = K; g.K[0] = K; g.W = W; g.ldw = K; g.out_f32 = part; g.ldo_f32 = N;
    g.split_k = slices; g.split_stride = (long long)M * N;
    TRY(gemm_dispatch(g, st));
    const long n4 = (long)M * N / 4;
    TimingScope ts(ADCDIFF_KIND_OTHER, M, N, 0, 0.0, (double)M * N * 4.0 * (slices + 1), st);
    hipLaunchKernelGGL(sum_partials_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, part, slices, (long)M * N, n4, out, N, (long)N);
    return check_launch("sum_partials");
}

// d(weight)[M, N] (rows ldo apart) = dY[K, M]^T X[K, N] on the operands as the forward pass stored them (K = batch rows), the
// contraction cut into slices when M x N alone leaves most CUs idle; partials are summed in slice order.
struct DwQueue {
    TrProblem probs[TR_MAX_GROUP];
    int n = 0;
};

// queue: the product is recorded and runs later, with all the others of the backward pass, as one launch (gemm_tr_grouped) -- its
// operands and its output must stay untouched until then; without a queue it is launched now
static int gemm_dw_direct(const void* dY, long ldy, const void* X, long ldx, int M, int N, int K, float* out, long ldo, int n_whole, Bump& ws,
                          bool dry, hipStream_t st, DwQueue* queue = nullptr) {
    if (queue) {
        if (dry) return ADCDIFF_OK;
        if (queue->n >= TR_MAX_GROUP) return fail(ADCDIFF_E_BADARG, "unet_backward: weight-gradient queue overflow");
        queue->probs[queue->n++] = TrProblem{dY, ldy, X, ldx, out, ldo, M, N, K};
        return ADCDIFF_OK;
    }
    // the slicing of the transposing path (which takes a concatenated input in one product of n_whole columns): same bits
    const int slices = dw_slices(M, n_whole, K);
    if (slices == 1) {
        if (dry) return ADCDIFF_OK;
        return gemm_tr(true, dY, ldy, X, ldx, out, ldo, M, N, K, 1, 0, st);
    }
    const size_t keep = ws.off;
    float* part = (float*)ws.take((size_t)slices * M * N * 4);
    ws.off = keep;
    if (dry) return ADCDIFF_OK;
    TRY(gemm_tr(true, dY, ldy, X, ldx, part, N, M, N, K, slices, (long)M * N, st));
    const long n4 = (long)M * N / 4;
    TimingScope ts(ADCDIFF_KIND_OTHER, M, N, 0, 0.0, (double)M * N * 4.0 * (slices + 1), st);
    hipLaunchKernelGGL(sum_partials_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, part, slices, (long)M * N, n4, out, N, ldo);
    return check_launch("sum_partials");
}

// debug switch train_transpose = 1 keeps the backward pass on the NT kernel with transposed copies (a test compares the two)
static bool direct_enabled() { return debug_value(DBG_TRAIN_TRANSPOSE) == 0; }

static int copy_f32(const float* src, float* dst, long n, int accumulate, hipStream_t st) {
    TimingScope ts(ADCDIFF_KIND_OTHER, 1, 0, 0, 0.0, (double)n * 8.0, st);
    hipLaunchKernelGGL(copy_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, dst, n, accumulate);
    return check_launch("copy_f32");
}

static int add_rows(const float* src, long lds_, float* dst, long ldd, int rows, int cols, hipStream_t st) {
    TimingScope ts(ADCDIFF_KIND_OTHER, rows, cols, 0, 0.0, (double)rows * cols * 12.0, st);
    const long total = (long)rows * cols;
    hipLaunchKernelGGL(add_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, src, lds_, dst, ldd, rows, cols);
    return check_launch("add_rows");
}

static int silu_bwd(const float* x, const float* dy, float* dx, long nel, hipStream_t st) {
    TimingScope ts(ADCDIFF_KIND_OTHER, 1, 0, 0, 0.0, (double)nel * 12.0, st);
    hipLaunchKernelGGL(silu_bwd_kernel, dim3((unsigned)((nel + 255) / 256)), dim3(256), 0, st, x, dy, dx, nel);
    return check_launch("silu_bwd");
}

// condition_embed (diffusion_model.py:694-698) inside a training forward, with what its backward needs left in the workspace:
// the condition in the operand type, the hidden pre-activation (float32), the hidden activation (operand type), the output.
static int cond_embed_taped(adcdiff_unet* n, const float* cond, int ldc, int B, Bump& ws, bool dry, hipStream_t st, Tape* tape) {
    const int Bp = (int)round_up((size_t)B, 128), H = n->H, D = n->D, dt = n->dtype;
    const size_t e = (size_t)n->esz;
    void* c_op = ws.take((size_t)Bp * D * e);
    float* pre = (float*)ws.take((size_t)Bp * H * 4);
    void* hid = ws.take((size_t)Bp * H * e);
    float* cemb = (float*)ws.take((size_t)Bp * H * 4);
    if (dry) return ADCDIFF_OK;
    TRY(cast_pack(cond, ldc, dt, c_op, D, 0, B, D, st));
    adcdiff_gemm_args g = gemm_args(dt, Bp, B, H);
    g.nseg = 1; g.A[0] = c_op; g.lda[0] = D; g.K[0] = D; g.W = n->w_c0; g.ldw = D; g.bias = n->prm.cond0_b;
    g.out_f32 = pre; g.ldo_f32 = H;
    TRY(gemm_dispatch(g, st));
    TRY(temb_silu(pre, H, B, nullptr, -1, nullptr, 0, dt, hid, H, B, H, st));   // hid = SiLU(pre), row r of the "table" is row r
    g = gemm_args(dt, Bp, B, H);
    g.nseg = 1; g.A[0] = hid; g.lda[0] = H; g.K[0] = H; g.W = n->w_c2; g.ldw = H; g.bias = n->prm.cond2_b;
    g.out_f32 = cemb; g.ldo_f32 = H;
    TRY(gemm_dispatch(g, st));
    tape->cond = true; tape->c_op = c_op; tape->c_pre = pre; tape->c_hid = hid; tape->cemb = cemb;
    return ADCDIFF_OK;
}

}  // namespace adcdiff

using namespace adcdiff;

static size_t train_forward_bytes(adcdiff_unet* n, int B) {
    Bump ws(nullptr, 0);
    cond_embed_taped(n, nullptr, 0, B, ws, true, nullptr, nullptr);   // sized for the conditional case, the larger one
    const size_t base = ws.off;
    forward_unfused(n, nullptr, 0, B, nullptr, 0, nullptr, 0, nullptr, 0, ws, true, nullptr);
    size_t need = ws.off;
    if (n->fused) {   // either forward may run (debug switch train_fused_fwd): size for the larger carving
        Bump w2(nullptr, 0);
        w2.off = base;
        forward_taped(n, nullptr, 0, B, nullptr, nullptr, 0, nullptr, 0, w2, true, nullptr, nullptr, 0.0f, 0);
        if (w2.off > need) need = w2.off;
    }
    return need;
}

static int backward_impl(adcdiff_unet* n, const float* d_eps, int lde, const adcdiff_unet_grads* gr, int B, Bump& ws, bool dry, hipStream_t st);

// Backward workspace = a dry run of the same allocation sequence.
static size_t train_backward_bytes(adcdiff_unet* n, int B) {
    static const adcdiff_unet_grads none{};
    Bump ws(nullptr, 0);
    backward_impl(n, nullptr, 0, &none, B, ws, true, nullptr);
    return ws.peak + 256;
}

extern "C" size_t adcdiff_unet_train_workspace_bytes(int input_dim, int hidden_dim, int output_dim, int num_timesteps, int dtype, int batch) {
    adcdiff_unet n{};
    if (batch <= 0 || plan_dims(&n, input_dim, hidden_dim, output_dim, num_timesteps, dtype) != ADCDIFF_OK) return 0;
    return train_forward_bytes(&n, batch) + train_backward_bytes(&n, batch);
}

extern "C" int adcdiff_unet_train_forward(adcdiff_unet* n, const float* x, int ldx, int B, const int64_t* t_rows, const float* cond, int ldc,
                                          float dropout_p, uint64_t dropout_seed, float* eps, int lde, void* workspace, size_t workspace_bytes,
                                          adcdiff_stream_t stream_) {
    if (!n || !n->packed) return fail(ADCDIFF_E_BADARG, "unet_train_forward: plan missing or not packed");
    if (!x || !eps || !t_rows || B <= 0 || ldx < n->D || lde < n->O || !aligned16(x) || !aligned16(eps) || ldx % 4 || lde % 4)
        return fail(ADCDIFF_E_BADARG, "unet_train_forward: bad x / t / eps arguments");
    if (dropout_p < 0.0f || dropout_p >= 1.0f) return fail(ADCDIFF_E_BADARG, "unet_train_forward: dropout probability %f", dropout_p);
    if (!workspace || !aligned16(workspace)) return fail(ADCDIFF_E_BADARG, "unet_train_forward: workspace null or misaligned");
    const size_t need = train_forward_bytes(n, B) + train_backward_bytes(n, B);
    if (workspace_bytes < need) return fail(ADCDIFF_E_WORKSPACE, "unet_train_forward: workspace of %zu bytes, %zu needed", workspace_bytes, need);
    for (int i = 0; i < 10; ++i)
        if (n->blk[i].cout % 128 != 0) return fail(ADCDIFF_E_BADARG, "training needs layer widths that are multiples of 128 (got %d)", n->blk[i].cout);
    if (cond && (ldc < n->D || !aligned16(cond) || ldc % 4)) return fail(ADCDIFF_E_BADARG, "unet_train_forward: condition misaligned or ldc=%d < %d", ldc, n->D);
    Bump ws(workspace, workspace_bytes);
    n->tape = Tape{};
    if (cond) TRY(cond_embed_taped(n, cond, ldc, B, ws, false, as_stream(stream_), &n->tape));
    else cond_embed_taped(n, nullptr, 0, B, ws, true, nullptr, nullptr);   // the same carving with and without a condition
    if (n->fused && debug_value(DBG_TRAIN_FUSED_FWD) != 0)
        return forward_taped(n, x, ldx, B, t_rows, n->tape.cemb, n->H, eps, lde, ws, false, as_stream(stream_), &n->tape, dropout_p, dropout_seed);
    return forward_unfused(n, x, ldx, B, t_rows, 0, n->tape.cemb, n->H, eps, lde, ws, false, as_stream(stream_), &n->tape, dropout_p, dropout_seed);
}

extern "C" int adcdiff_mse_loss_grad(const float* pred, int ldp, const float* target, int ldt, float* dpred, int ldd, int B, int D,
                                     float scale, float* loss_accum, adcdiff_stream_t stream_) {
    if (!pred || !target || !dpred || !loss_accum || B <= 0 || D <= 0 || ldp < D || ldt < D || ldd < D)
        return fail(ADCDIFF_E_BADARG, "mse_loss_grad: bad arguments");
    const long nel = (long)B * D;
    TimingScope ts(ADCDIFF_KIND_OTHER, B, D, 0, 0.0, (double)B * D * 12.0, as_stream(stream_));
    hipLaunchKernelGGL(mse_grad_kernel, dim3((unsigned)((nel + 4095) / 4096)), dim3(256), 0, as_stream(stream_), pred, (long)ldp, target, (long)ldt,
                       dpred, (long)ldd, B, D, scale, loss_accum);
    return check_launch("mse_loss_grad");
}

extern "C" int adcdiff_unet_backward(adcdiff_unet* n, const float* d_eps, int lde, const adcdiff_unet_grads* gr, void* workspace,
                                     size_t workspace_bytes, adcdiff_stream_t stream_) {
    if (!n || !n->tape.valid) return fail(ADCDIFF_E_BADARG, "unet_backward: no training forward recorded on this plan");
    if (!d_eps || !gr || lde < n->O) return fail(ADCDIFF_E_BADARG, "unet_backward: bad arguments");
    const size_t fwd = round_up(n->tape.ws_used, 256);
    if (workspace_bytes < fwd + train_backward_bytes(n, n->tape.B)) return fail(ADCDIFF_E_WORKSPACE, "unet_backward: workspace too small");
    Bump ws(reinterpret_cast<char*>(workspace) + fwd, workspace_bytes - fwd);
    const int rc = backward_impl(n, d_eps, lde, gr, n->tape.B, ws, false, as_stream(stream_));
    if (rc == ADCDIFF_OK && ws.peak > ws.cap) return fail(ADCDIFF_E_WORKSPACE, "unet_backward: internal sizing error (%zu > %zu)", ws.peak, ws.cap);
    return rc;
}

// debug switch sync = 1 synchronises after every backward launch and names the first one that fails.
static bool debug_sync() { return debug_value(DBG_SYNC) != 0; }
#define RUN(expr)                                                                                              \
    do {                                                                                                       \
        if (!dry) {                                                                                            \
            TRY(expr);                                                                                         \
            if (debug_sync()) {                                                                                \
                hipError_t e_ = hipStreamSynchronize(st);                                                      \
                if (e_ != hipSuccess) return fail(ADCDIFF_E_LAUNCH, "%s (block %d): %s", #expr, dbg_block, hipGetErrorString(e_)); \
            }                                                                                                  \
        }                                                                                                      \
    } while (0)

static int backward_impl(adcdiff_unet* n, const float* d_eps, int lde, const adcdiff_unet_grads* gr, int B, Bump& ws, bool dry, hipStream_t st) {
    const Tape& tp = n->tape;
    const int Bp = (int)round_up((size_t)B, 128), dt = n->dtype, H = n->H, O = n->O, Tp = n->Tp, SN = n->sumN;
    const size_t e = (size_t)n->esz;
    const adcdiff_unet_params& p = n->prm;
    int dbg_block = 10;

    // gradients of the block outputs, and of the per-timestep table
    float* dO[10];
    bool dO_written[10];
    for (int i = 0; i < 10; ++i) { dO[i] = (float*)ws.take((size_t)Bp * n->blk[i].cout * 4); dO_written[i] = false; }
    float* d_tb = (float*)ws.take((size_t)Tp * SN * 4);
    // conditional forward (per-row condition): the time term of a row is time_proj(s_r), s_r = SiLU(t_emb[t_r] + c_emb_r), an
    // operand of every block's first GEMM; its gradient d(s) [Bp, H] is summed over the blocks, nothing goes through the
    // per-timestep table of projections
    const bool cond = tp.cond;
    float* d_s_rows = (float*)ws.take((size_t)Bp * H * 4);
    void* sT = ws.take((size_t)H * Bp * e);      // s transposed, for the path on transposed copies
    if (!dry) {
        if (hipMemsetAsync(cond ? d_s_rows : d_tb, 0, cond ? (size_t)Bp * H * 4 : (size_t)Tp * SN * 4, st) != hipSuccess)
            return fail(ADCDIFF_E_LAUNCH, "unet_backward: memset failed");
        if (cond && (!gr->cond0_w || !gr->cond0_b || !gr->cond2_w || !gr->cond2_b))
            return fail(ADCDIFF_E_BADARG, "unet_backward: the forward was conditional, condition_embed gradient destinations are needed");
    }

    // ---- output head: eps = out_layer(a_out), a_out = SiLU(output_norm(o9)) (diffusion_model.py:786-788)
    // bf16 with whole 64-row contraction tiles: the backward GEMMs read d(y), the saved activations and the packed weights as
    // they are (gemm_tr); otherwise the NT kernel on transposed copies
    const bool direct = dt == ADCDIFF_BF16 && B % 64 == 0 && direct_enabled();
    if (cond && !direct) RUN(transpose_op(tp.s_op, H, B, Bp, H, dt, sT, st));
    // The weight gradients of the blocks and of the output layer feed nothing in this pass: they are queued and run as ONE launch
    // behind the last block (thousands of tiles instead of 4 to 256 per product, no split over the batch, no partial sums).
    // Their d(y) operands therefore get buffers of their own per block, alive to the end.
    DwQueue dwq;
    DwQueue* const q_dw = direct && debug_value(DBG_TRAIN_GROUP_DW) != 0 ? &dwq : nullptr;
    void* dY_keep[10][2] = {};
    if (direct) {
        for (int i = 0; i < 10; ++i)
            for (int k = 0; k < 2; ++k) dY_keep[i][k] = ws.take((size_t)Bp * n->blk[i].cout * e);
        // rows B .. Bp of these operands are read by the d(x) products (M = Bp): zero, as the cast kernels left them
        if (!dry && Bp > B)
            for (int i = 0; i < 10; ++i)
                for (int k = 0; k < 2; ++k)
                    if (hipMemsetAsync((char*)dY_keep[i][k] + (size_t)B * n->blk[i].cout * e, 0, (size_t)(Bp - B) * n->blk[i].cout * e, st) != hipSuccess)
                        return fail(ADCDIFF_E_LAUNCH, "unet_backward: memset failed");
    }
    // every GroupNorm backward launch leaves partial rows (d gamma | d beta | column sums) that one launch sums at the end
    ColsumQueue cq;
    {
        size_t need = gn_bwd_part_bytes(B, H);
        for (int i = 0; i < 10; ++i) {
            need += gn_bwd_part_bytes(B, n->blk[i].cout);
            for (int k = 0; k < n->blk[i].nparts; ++k) need += gn_bwd_part_bytes(B, n->blk[i].part_c[k]);
        }
        cq.parts = Bump(ws.take(need), need);
    }
    // direct path: the bf16 copy of a finished d(block output) / d(h1) and its column sums (bias gradients) come out of the
    // GroupNorm backward kernel that finishes it, instead of a cast pass over it
    const bool fuse_cast = direct && debug_value(DBG_TRAIN_FUSE_CAST) != 0;
    if (direct) {
        void* dE_rm = ws.take((size_t)Bp * O * e);
        RUN(cast_rows(d_eps, lde, B, Bp, O, dE_rm, O, gr->out_b, nullptr, st));
        TRY(gemm_dw_direct(dE_rm, O, tp.a_out, H, O, H, B, gr->out_w, H, H, ws, dry, st, q_dw));
        float* d_a = (float*)ws.take((size_t)Bp * H * 4);
        RUN(gemm_tr(false, dE_rm, O, n->w_out, H, d_a, H, Bp, H, O, 1, 0, st));
        TRY(gn_silu_bwd(tp.blk[9].o, n->blk[9].cout, d_a, H, B, H, num_groups(H), p.out_norm_w, p.out_norm_b, 0.0f, 0, 0, 0, nullptr, 0, dO[9],
                        n->blk[9].cout, 0, gr->out_norm_w, gr->out_norm_b, cq, dry, st, fuse_cast ? dY_keep[9][0] : nullptr, n->blk[9].cout,
                        fuse_cast ? gr->blocks[9].linear2_b : nullptr, fuse_cast && n->blk[9].has_res ? gr->blocks[9].residual_b : nullptr));
        dO_written[9] = true;
    } else {
        void* dE_rm = ws.take((size_t)Bp * O * e);
        void* dE_t = ws.take((size_t)O * Bp * e);
        RUN(cast_transpose<float>(d_eps, lde, B, Bp, O, dt, dE_rm, O, dE_t, Bp, gr->out_b, st));
        void* aT = ws.take((size_t)H * Bp * e);
        RUN(transpose_op(tp.a_out, H, B, Bp, H, dt, aT, st));
        TRY(gemm_dw(dt, dE_t, aT, O, H, Bp, gr->out_w, ws, dry, st));
        void* wT = ws.take((size_t)H * O * e);
        RUN(cast_transpose<float>(p.out_w, H, O, O, H, dt, nullptr, 0, wT, O, nullptr, st));
        float* d_a = (float*)ws.take((size_t)Bp * H * 4);
        RUN(gemm_simple(dt, dE_rm, O, wT, O, Bp, B, H, O, d_a, H, st));
        TRY(gn_silu_bwd(tp.blk[9].o, n->blk[9].cout, d_a, H, B, H, num_groups(H), p.out_norm_w, p.out_norm_b, 0.0f, 0, 0, 0, nullptr, 0, dO[9],
                        n->blk[9].cout, 0, gr->out_norm_w, gr->out_norm_b, cq, dry, st));
        dO_written[9] = true;
    }
    const size_t mark = ws.off;
    for (int i = 9; i >= 0; --i) {
        ws.off = mark;   // per-block temporaries are reused
        dbg_block = i;
        const BlockPlan& b = n->blk[i];
        const BlockTape& bt = tp.blk[i];
        const adcdiff_block_params& q = p.blocks[i];
        const adcdiff_block_grads& g = gr->blocks[i];
        const int cin = b.cin, cout = b.cout;
        float* d_raw = nullptr;
        float* d_a1 = nullptr;
        long ld_raw = cin, ld_a1 = cin;
        if (direct) {
            // (1) dO in bf16; its column sums are d(linear2.bias) (= d(residual_proj.bias))
            void* dY = dY_keep[i][0];
            if (!fuse_cast) RUN(cast_rows(dO[i], cout, B, Bp, cout, dY, cout, g.linear2_b, b.has_res ? g.residual_b : nullptr, st));
            // (2) d(linear2.weight) = dO^T a2; (3) d(residual_proj.weight) = dO^T x, part by part
            TRY(gemm_dw_direct(dY, cout, bt.a2, cout, cout, cout, B, g.linear2_w, cout, cout, ws, dry, st, q_dw));
            if (b.has_res) {
                int col0 = 0;
                for (int k = 0; k < b.nparts; ++k) {
                    TRY(gemm_dw_direct(dY, cout, bt.in_raw[k], b.part_c[k], cout, b.part_c[k], B, g.residual_w + col0, cin, cin, ws, dry, st, q_dw));
                    col0 += b.part_c[k];
                }
            }
            // (4) [d(a2) | d(x) through residual_proj] = dO [W2 | Wres] in one product on the packed weight
            const int n2 = b.has_res ? cout + cin : cout;
            float* d_both = (float*)ws.take((size_t)Bp * n2 * 4);
            RUN(gemm_tr(false, dY, cout, b.w2cat, b.ldw2, d_both, n2, Bp, n2, cout, 1, 0, st));
            if (b.has_res) { d_raw = d_both + cout; ld_raw = n2; }
            // (5) through dropout, SiLU and norm2 to d(h1)
            float* d_h1 = (float*)ws.take((size_t)Bp * cout * 4);
            TRY(gn_silu_bwd(bt.h1, cout, d_both, n2, B, cout, b.g2, q.norm2_w, q.norm2_b, tp.drop_p, tp.drop_seed, (uint32_t)i, 0, nullptr, 0, d_h1,
                            cout, 0, g.norm2_w, g.norm2_b, cq, dry, st, fuse_cast ? dY_keep[i][1] : nullptr, cout, fuse_cast ? g.linear1_b : nullptr,
                            fuse_cast ? g.time_proj_b : nullptr));
            // (6) d(h1) in bf16: column sums are d(linear1.bias) = d(time_proj.bias); (7) rows go to the timestep table
            dY = dY_keep[i][1];
            if (!fuse_cast) RUN(cast_rows(d_h1, cout, B, Bp, cout, dY, cout, g.linear1_b, g.time_proj_b, st));
            if (!cond) {
                const long total = (long)B * cout;
                if (!dry) {
                    TimingScope ts(ADCDIFF_KIND_OTHER, B, cout, 0, 0.0, (double)total * 12.0, st);
                    hipLaunchKernelGGL(scatter_add_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, d_h1, (long)cout, tp.t_rows, B,
                                       cout, d_tb, (long)SN, b.tb_off, n->T);
                    TRY(check_launch("scatter_add_rows"));
                }
            }
            // (8) d(linear1.weight) = d_h1^T a1; (9) d(a1) = d_h1 W1 (the first cin columns of the packed [W1 | Wtp])
            TRY(gemm_dw_direct(dY, cout, bt.a1, cin, cout, cin, B, g.linear1_w, cin, cin, ws, dry, st, q_dw));
            // with a condition also d(time_proj.weight) = d_h1^T s, and [d(a1) | d(s)] = d_h1 [W1 | Wtp] in one product
            if (cond || dry) TRY(gemm_dw_direct(dY, cout, tp.s_op, H, cout, H, B, g.time_proj_w, H, H, ws, dry, st, q_dw));
            ld_a1 = cond ? cin + H : cin;
            d_a1 = (float*)ws.take((size_t)Bp * (cin + H) * 4);
            RUN(gemm_tr(false, dY, cout, b.w1cat, b.ldw1, d_a1, ld_a1, Bp, (int)ld_a1, cout, 1, 0, st));
            if (cond) RUN(add_rows(d_a1 + cin, ld_a1, d_s_rows, H, B, H, st));
        } else {
            // (1) dO in the operand type, row-major and transposed; its column sums are d(linear2.bias) (= d(residual_proj.bias))
            void* dO_rm = ws.take((size_t)Bp * cout * e);
            void* dO_t = ws.take((size_t)cout * Bp * e);
            RUN(cast_transpose<float>(dO[i], cout, B, Bp, cout, dt, dO_rm, cout, dO_t, Bp, g.linear2_b, st));
            if (b.has_res) RUN(copy_f32(g.linear2_b, g.residual_b, cout, 0, st));
            // (2) d(linear2.weight) = dO^T a2
            void* xT = ws.take((size_t)(cin > cout ? cin : cout) * Bp * e);
            RUN(transpose_op(bt.a2, cout, B, Bp, cout, dt, xT, st));
            TRY(gemm_dw(dt, dO_t, xT, cout, cout, Bp, g.linear2_w, ws, dry, st));
            // (3) residual_proj: d(weight) = dO^T x, d(x) = dO Wres
            void* wT = ws.take((size_t)(cin > cout ? cin : cout) * cout * e);
            if (b.has_res) {
                int row0 = 0;
                for (int k = 0; k < b.nparts; ++k) {
                    RUN(transpose_op(bt.in_raw[k], b.part_c[k], B, Bp, b.part_c[k], dt, (char*)xT + (size_t)row0 * Bp * e, st));
                    row0 += b.part_c[k];
                }
                TRY(gemm_dw(dt, dO_t, xT, cout, cin, Bp, g.residual_w, ws, dry, st));
                RUN(cast_transpose<float>(q.residual_w, cin, cout, cout, cin, dt, nullptr, 0, wT, cout, nullptr, st));
                d_raw = (float*)ws.take((size_t)Bp * cin * 4);
                RUN(gemm_simple(dt, dO_rm, cout, wT, cout, Bp, B, cin, cout, d_raw, cin, st));
            }
            // (4) d(a2) = dO W2; (5) through dropout, SiLU and norm2 to d(h1)
            RUN(cast_transpose<float>(q.linear2_w, cout, cout, cout, cout, dt, nullptr, 0, wT, cout, nullptr, st));
            float* d_a2 = (float*)ws.take((size_t)Bp * cout * 4);
            RUN(gemm_simple(dt, dO_rm, cout, wT, cout, Bp, B, cout, cout, d_a2, cout, st));
            float* d_h1 = (float*)ws.take((size_t)Bp * cout * 4);
            TRY(gn_silu_bwd(bt.h1, cout, d_a2, cout, B, cout, b.g2, q.norm2_w, q.norm2_b, tp.drop_p, tp.drop_seed, (uint32_t)i, 0, nullptr, 0, d_h1,
                            cout, 0, g.norm2_w, g.norm2_b, cq, dry, st));
            // (6) d(h1): column sums are d(linear1.bias) = d(time_proj.bias); (7) rows go to the timestep table
            RUN(cast_transpose<float>(d_h1, cout, B, Bp, cout, dt, dO_rm, cout, dO_t, Bp, g.linear1_b, st));
            RUN(copy_f32(g.linear1_b, g.time_proj_b, cout, 0, st));
            if (!cond) {
                const long total = (long)B * cout;
                if (!dry) {
                    TimingScope ts(ADCDIFF_KIND_OTHER, B, cout, 0, 0.0, (double)total * 12.0, st);
                    hipLaunchKernelGGL(scatter_add_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, d_h1, (long)cout, tp.t_rows, B,
                                       cout, d_tb, (long)SN, b.tb_off, n->T);
                    TRY(check_launch("scatter_add_rows"));
                }
            }
            if (cond || dry) {
                // with a condition: d(time_proj.weight) = d_h1^T s and d(s) += d_h1 Wtp
                TRY(gemm_dw(dt, dO_t, sT, cout, H, Bp, g.time_proj_w, ws, dry, st));
                float* d_si = (float*)ws.take((size_t)Bp * H * 4);
                if (cond) {
                    RUN(cast_transpose<float>(q.time_proj_w, H, cout, cout, H, dt, nullptr, 0, wT, cout, nullptr, st));
                    RUN(gemm_simple(dt, dO_rm, cout, wT, cout, Bp, B, H, cout, d_si, H, st));
                    RUN(add_rows(d_si, H, d_s_rows, H, B, H, st));
                }
            }
            // (8) d(linear1.weight) = d_h1^T a1; (9) d(a1) = d_h1 W1
            RUN(transpose_op(bt.a1, cin, B, Bp, cin, dt, xT, st));
            TRY(gemm_dw(dt, dO_t, xT, cout, cin, Bp, g.linear1_w, ws, dry, st));
            RUN(cast_transpose<float>(q.linear1_w, cin, cout, cout, cin, dt, nullptr, 0, wT, cout, nullptr, st));
            d_a1 = (float*)ws.take((size_t)Bp * cin * 4);
            RUN(gemm_simple(dt, dO_rm, cout, wT, cout, Bp, B, cin, cout, d_a1, cin, st));
        }
        // (10) through SiLU and norm1 of each input part; the result plus the residual-path gradient is the gradient of the
        // tensor that part came from (previous block, or the encoder output of a skip connection)
        float* scratch = (float*)ws.take((size_t)Bp * b.part_c[0] * 4);
        int col = 0;
        for (int k = 0; k < b.nparts; ++k) {
            const int src = k == 0 ? i - 1 : 9 - i;    // producer of this part (-1: the network input)
            float* dst = src >= 0 ? dO[src] : scratch;
            const int lddst = b.part_c[k];
            const int acc = src >= 0 && dO_written[src] ? 1 : 0;
            const float* add = b.has_res ? d_raw + col : dO[i];     // Identity residual passes dO straight through
            const long ldadd = b.has_res ? ld_raw : cout;
            // part 0 of a block's input is the previous block's output, and this launch is the last to write its gradient (a skip
            // consumer, one of the decoder blocks, ran earlier): it finishes d(output of block i - 1)
            const bool fin = fuse_cast && k == 0 && src >= 0;
            TRY(gn_silu_bwd(bt.in_f32[k], bt.in_ld[k], d_a1 + col, ld_a1, B, b.part_c[k], b.part_c[k] / b.gs1, q.norm1_w + col, q.norm1_b + col, 0.0f,
                            0, 0, 0, add, ldadd, dst, lddst, acc, g.norm1_w + col, g.norm1_b + col, cq, dry, st, fin ? dY_keep[src][0] : nullptr, lddst,
                            fin ? gr->blocks[src].linear2_b : nullptr, fin && n->blk[src].has_res ? gr->blocks[src].residual_b : nullptr));
            if (src >= 0) dO_written[src] = true;
            col += b.part_c[k];
        }
    }
    if (q_dw && !dry) TRY(gemm_tr_grouped(dwq.probs, dwq.n, st));
    // ---- time branch: tb_table = SiLU(te_table) Wtp_all^T + btp_all; te_table = te_hidden W2^T + b2; te_hidden = SiLU(te_pre);
    //      te_pre = sinusoid W1^T + b1 (all float32, M = Tp rows of which only the drawn timesteps carry gradient)
    ws.off = mark;
    {
        float* d_te = (float*)ws.take((size_t)Tp * H * 4);   // gradient of the TimeEmbedding table (rows = timesteps)
        const size_t tail = ws.off;
        if (!cond || dry) {
            float* d_s = nullptr;
            if (direct) {
                // the two large products of the branch (contraction over the Tp timesteps and over all sumN time_proj outputs) on bf16
                // copies of d_tb, s_table and the stacked time_proj weights, read as stored
                void* dtb_op = ws.take((size_t)Tp * SN * e);
                RUN(cast_transpose<float>(d_tb, SN, Tp, Tp, SN, dt, dtb_op, SN, nullptr, 0, nullptr, st));
                void* s_op = ws.take((size_t)Tp * H * e);
                RUN(cast_transpose<float>(n->s_table, H, Tp, Tp, H, dt, s_op, H, nullptr, 0, nullptr, st));
                for (int i = 0; i < 10; ++i)
                    TRY(gemm_dw_direct((const char*)dtb_op + (size_t)n->blk[i].tb_off * e, SN, s_op, H, n->blk[i].cout, H, Tp, gr->blocks[i].time_proj_w, H,
                                       H, ws, dry, st));
                void* wtp_op = ws.take((size_t)SN * H * e);
                RUN(cast_pack(n->wtp_all, H, dt, wtp_op, H, 0, SN, H, st));
                d_s = (float*)ws.take((size_t)Tp * H * 4);
                int slices = 1;
                for (int c = 16; c > 1; --c)
                    if (SN % (64 * c) == 0) { slices = c; break; }
                float* part = slices > 1 ? (float*)ws.take((size_t)slices * Tp * H * 4) : d_s;
                RUN(gemm_tr(false, dtb_op, SN, wtp_op, H, part, H, Tp, H, SN, slices, (long)Tp * H, st));
                if (slices > 1 && !dry) {
                    const long n4 = (long)Tp * H / 4;
                    TimingScope ts(ADCDIFF_KIND_OTHER, Tp, H, 0, 0.0, (double)Tp * H * 4.0 * (slices + 1), st);
                    hipLaunchKernelGGL(sum_partials_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, part, slices, (long)Tp * H, n4, d_s, H, (long)H);
                    TRY(check_launch("sum_partials"));
                }
            } else {
                float* dTB_t = (float*)ws.take((size_t)SN * Tp * 4);
                RUN(cast_transpose<float>(d_tb, SN, Tp, Tp, SN, ADCDIFF_F32, nullptr, 0, dTB_t, Tp, nullptr, st));
                float* s_tab_t = (float*)ws.take((size_t)H * Tp * 4);
                RUN(cast_transpose<float>(n->s_table, H, Tp, Tp, H, ADCDIFF_F32, nullptr, 0, s_tab_t, Tp, nullptr, st));
                for (int i = 0; i < 10; ++i)
                    RUN(gemm_simple(ADCDIFF_F32, dTB_t + (size_t)n->blk[i].tb_off * Tp, Tp, s_tab_t, Tp, n->blk[i].cout, 0, H, Tp, gr->blocks[i].time_proj_w, H, st));
                float* wtpT = (float*)ws.take((size_t)H * SN * 4);
                RUN(cast_transpose<float>(n->wtp_all, H, SN, SN, H, ADCDIFF_F32, nullptr, 0, wtpT, SN, nullptr, st));
                d_s = (float*)ws.take((size_t)Tp * H * 4);
                RUN(gemm_simple(ADCDIFF_F32, d_tb, SN, wtpT, SN, Tp, 0, H, SN, d_s, H, st));
            }
            if (!dry && !cond) {
                TimingScope ts(ADCDIFF_KIND_OTHER, Tp, H, 0, 0.0, (double)Tp * H * 12.0, st);
                hipLaunchKernelGGL(silu_bwd_kernel, dim3((unsigned)(((long)Tp * H + 255) / 256)), dim3(256), 0, st, n->te_table, d_s, d_te, (long)Tp * H);
                TRY(check_launch("silu_bwd"));
            }
        }
        if (cond || dry) {
            // d(z) = d(s) * silu'(z), z = t_emb[t] + c_emb: its rows are scatter-added into the table gradient (the time MLP's
            // backward below) and are the gradient of c_emb (condition_embed's backward here, diffusion_model.py:694-698)
            ws.off = tail;
            const int D = n->D;
            float* d_z = (float*)ws.take((size_t)Bp * H * 4);
            if (!dry && cond) {
                if (hipMemsetAsync(d_te, 0, (size_t)Tp * H * 4, st) != hipSuccess) return fail(ADCDIFF_E_LAUNCH, "unet_backward: memset failed");
                TimingScope ts(ADCDIFF_KIND_OTHER, B, H, 0, 0.0, (double)B * H * 16.0, st);
                const long total = (long)B * H;
                hipLaunchKernelGGL(cond_silu_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, n->te_table, H, n->T, tp.t_rows, tp.cemb, H,
                                   d_s_rows, d_z, B, H);
                hipLaunchKernelGGL(scatter_add_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, d_z, (long)H, tp.t_rows, B, H, d_te, (long)H, 0, n->T);
                TRY(check_launch("cond_silu_bwd"));
            }
            float* d_hid = (float*)ws.take((size_t)Bp * H * 4);
            float* d_cpre = (float*)ws.take((size_t)Bp * H * 4);
            void* dz_rm = ws.take((size_t)Bp * H * e);
            const bool run = cond && !dry;
            if (direct) {
                if (run) TRY(cast_rows(d_z, H, B, Bp, H, dz_rm, H, gr->cond2_b, nullptr, st));
                TRY(gemm_dw_direct(dz_rm, H, tp.c_hid, H, H, H, B, gr->cond2_w, H, H, ws, !run, st));
                if (run) {
                    TRY(gemm_tr(false, dz_rm, H, n->w_c2, H, d_hid, H, Bp, H, H, 1, 0, st));
                    TRY(silu_bwd(tp.c_pre, d_hid, d_cpre, (long)B * H, st));
                    TRY(cast_rows(d_cpre, H, B, Bp, H, dz_rm, H, gr->cond0_b, nullptr, st));
                }
                TRY(gemm_dw_direct(dz_rm, H, tp.c_op, D, H, D, B, gr->cond0_w, D, D, ws, !run, st));
            } else {
                void* dz_t = ws.take((size_t)H * Bp * e);
                void* xT = ws.take((size_t)(D > H ? D : H) * Bp * e);
                void* wT = ws.take((size_t)H * H * e);
                if (run) {
                    TRY(cast_transpose<float>(d_z, H, B, Bp, H, dt, dz_rm, H, dz_t, Bp, gr->cond2_b, st));
                    TRY(transpose_op(tp.c_hid, H, B, Bp, H, dt, xT, st));
                }
                TRY(gemm_dw(dt, dz_t, xT, H, H, Bp, gr->cond2_w, ws, !run, st));
                if (run) {
                    TRY(cast_transpose<float>(p.cond2_w, H, H, H, H, dt, nullptr, 0, wT, H, nullptr, st));
                    TRY(gemm_simple(dt, dz_rm, H, wT, H, Bp, B, H, H, d_hid, H, st));
                    TRY(silu_bwd(tp.c_pre, d_hid, d_cpre, (long)B * H, st));
                    TRY(cast_transpose<float>(d_cpre, H, B, Bp, H, dt, dz_rm, H, dz_t, Bp, gr->cond0_b, st));
                    TRY(transpose_op(tp.c_op, D, B, Bp, D, dt, xT, st));
                }
                TRY(gemm_dw(dt, dz_t, xT, H, D, Bp, gr->cond0_w, ws, !run, st));
            }
        }
        ws.off = tail;
        float* dTE_t = (float*)ws.take((size_t)H * Tp * 4);
        RUN(cast_transpose<float>(d_te, H, Tp, Tp, H, ADCDIFF_F32, nullptr, 0, dTE_t, Tp, gr->time_lin2_b, st));
        float* hidT = (float*)ws.take((size_t)4 * H * Tp * 4);
        RUN(cast_transpose<float>(n->te_hidden, 4 * H, Tp, Tp, 4 * H, ADCDIFF_F32, nullptr, 0, hidT, Tp, nullptr, st));
        RUN(gemm_simple(ADCDIFF_F32, dTE_t, Tp, hidT, Tp, H, 0, 4 * H, Tp, gr->time_lin2_w, 4 * H, st));
        float* w2T = (float*)ws.take((size_t)4 * H * H * 4);
        RUN(cast_transpose<float>(p.time_lin2_w, 4 * H, H, H, 4 * H, ADCDIFF_F32, nullptr, 0, w2T, H, nullptr, st));
        float* d_hid = (float*)ws.take((size_t)Tp * 4 * H * 4);
        RUN(gemm_simple(ADCDIFF_F32, d_te, H, w2T, H, Tp, 0, 4 * H, H, d_hid, 4 * H, st));
        float* d_pre = (float*)ws.take((size_t)Tp * 4 * H * 4);
        if (!dry) {
            TimingScope ts(ADCDIFF_KIND_OTHER, Tp, 4 * H, 0, 0.0, (double)Tp * 4 * H * 12.0, st);
            hipLaunchKernelGGL(silu_bwd_kernel, dim3((unsigned)(((long)Tp * 4 * H + 255) / 256)), dim3(256), 0, st, n->te_pre, d_hid, d_pre, (long)Tp * 4 * H);
            TRY(check_launch("silu_bwd"));
        }
        float* dPRE_t = (float*)ws.take((size_t)4 * H * Tp * 4);
        RUN(cast_transpose<float>(d_pre, 4 * H, Tp, Tp, 4 * H, ADCDIFF_F32, nullptr, 0, dPRE_t, Tp, gr->time_lin1_b, st));
        float* sinT = (float*)ws.take((size_t)H * Tp * 4);
        RUN(cast_transpose<float>(n->sin_table, H, Tp, Tp, H, ADCDIFF_F32, nullptr, 0, sinT, Tp, nullptr, st));
        RUN(gemm_simple(ADCDIFF_F32, dPRE_t, Tp, sinT, Tp, 4 * H, 0, H, Tp, gr->time_lin1_w, H, st));
    }
    if (!dry) TRY(colsum_flush(cq, st));
    return ADCDIFF_OK;
}
