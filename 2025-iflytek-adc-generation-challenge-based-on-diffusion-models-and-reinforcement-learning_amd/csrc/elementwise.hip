// HBM-bound kernels of the DDPM path: GroupNorm+SiLU, the fused schedule update (posterior mean,
// variance, Philox noise), DDIM update, q_sample, Philox fill, and the small table helpers.
// Built with -ffp-contract=off so that the update formulas round exactly like the reference's
// separate float32 torch ops.
#include <atomic>
#include <cmath>
#include <cstring>
#include <mutex>

#include "unet_plan.h"

#define TRY_RC(expr) do { int rc_ = (expr); if (rc_ != ADCDIFF_OK) return rc_; } while (0)

namespace adcdiff {

char* error_buffer() {
    static thread_local char buf[512] = "";
    return buf;
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

// ------------------------------------------------------------------ per-launch timing
// One event in FRONT of every recorded launch and one behind the last: a launch's time is the span to the next
// launch's event, so the records of a window add up to the window (dispatch gaps are charged to the launch in front of them).
namespace {
struct TimingState {
    bool on = false;
    int cap = 0, used = 0;
    hipEvent_t* ev = nullptr;            // cap + 1 events
    adcdiff_timing_record* rec = nullptr;
    hipStream_t last_stream = nullptr;   // where the closing event is recorded
} g_timing;
}  // namespace

TimingScope::TimingScope(int kind, int M, int N, int K, double flops, double bytes, hipStream_t st) : slot(-1), stream(st) {
    if (!g_timing.on || g_timing.used >= g_timing.cap) return;
    slot = g_timing.used++;
    g_timing.rec[slot] = adcdiff_timing_record{kind, M, N, K, 0.0f, flops, bytes};
    g_timing.last_stream = stream;
    (void)hipEventRecord(g_timing.ev[slot], stream);
}

TimingScope::~TimingScope() {}

// ------------------------------------------------------------------ GroupNorm + SiLU
// One float4 (four channels) per thread; a group of gs channels spans gs/4 adjacent lanes, so the two
// statistics passes (mean, then centred sum of squares) are xor-shuffle reductions over gs/4 lanes.
// Reads x once, fully coalesced; writes the activation (and optionally a converted copy of x).
template <typename OutT, int LPG>
__global__ __launch_bounds__(256) void gn_silu_vec_kernel(const float* __restrict__ x, int ldx, long total4, int c4,
                                                          int lanes_per_group, float inv_gs, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float eps, OutT* __restrict__ out,
                                                          int ldo, OutT* __restrict__ raw, int ldraw, float drop_p, uint64_t drop_seed,
                                                          uint32_t drop_layer) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    const bool live = e < total4;
    const long ee = live ? e : 0;
    const long row = ee / c4;
    const int col = (int)(ee - row * c4) * 4;
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + row * ldx + col);
    const float s = lanes_allsum<LPG>((v[0] + v[1]) + (v[2] + v[3]), lanes_per_group);
    const float mean = s * inv_gs;
    const f32x4 d = {v[0] - mean, v[1] - mean, v[2] - mean, v[3] - mean};
    const float q = lanes_allsum<LPG>((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]), lanes_per_group);
    const float rstd = 1.0f / __builtin_sqrtf(q * inv_gs + eps);
    if (!live) return;
    const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + col);
    const f32x4 b = *reinterpret_cast<const f32x4*>(beta + col);
    f32x4 y;
#pragma unroll
    for (int i = 0; i < 4; ++i) y[i] = silu_f(d[i] * rstd * g[i] + b[i]);
    if (drop_p > 0.0f) {   // training-mode dropout (diffusion_model.py:898): keep with probability 1 - p, scale by 1 / (1 - p)
        const f32x4 keep = dropout_keep4(drop_seed, drop_layer, (uint64_t)row, (uint32_t)(col >> 2), drop_p);
#pragma unroll
        for (int i = 0; i < 4; ++i) y[i] *= keep[i];
    }
    if constexpr (sizeof(OutT) == 4) {
        *reinterpret_cast<f32x4*>(out + row * ldo + col) = y;
        if (raw) *reinterpret_cast<f32x4*>(raw + row * ldraw + col) = v;
    } else {
        bf16x4 o = {(bf16_t)y[0], (bf16_t)y[1], (bf16_t)y[2], (bf16_t)y[3]};
        *reinterpret_cast<bf16x4*>(out + row * ldo + col) = o;
        if (raw) {
            bf16x4 r = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
            *reinterpret_cast<bf16x4*>(raw + row * ldraw + col) = r;
        }
    }
}

// Any group size: one thread per (row, group), serial over the group's channels.
template <typename OutT>
__global__ __launch_bounds__(256) void gn_silu_generic_kernel(const float* __restrict__ x, int ldx, long total_groups, int groups,
                                                              int gs, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              float eps, OutT* __restrict__ out, int ldo, OutT* __restrict__ raw,
                                                              int ldraw) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total_groups) return;
    const long row = e / groups;
    const int c0 = (int)(e - row * groups) * gs;
    const float* xr = x + row * ldx + c0;
    float s = 0.0f;
    for (int i = 0; i < gs; ++i) s += xr[i];
    const float mean = s / (float)gs;
    float q = 0.0f;
    for (int i = 0; i < gs; ++i) { const float d = xr[i] - mean; q += d * d; }
    const float rstd = 1.0f / __builtin_sqrtf(q / (float)gs + eps);
    for (int i = 0; i < gs; ++i) {
        const float y = silu_f((xr[i] - mean) * rstd * gamma[c0 + i] + beta[c0 + i]);
        out[row * ldo + c0 + i] = (OutT)y;
        if (raw) raw[row * ldraw + c0 + i] = (OutT)xr[i];
    }
}

template <typename OutT>
static int gn_launch(const float* x, int ldx, int M, int C, int groups, const float* gamma, const float* beta, float eps,
                     OutT* out, int ldo, OutT* raw, int ldraw, hipStream_t stream, float drop_p, uint64_t drop_seed, uint32_t drop_layer) {
    const int gs = C / groups;
    const bool pow2 = (gs & (gs - 1)) == 0;
    const bool vec = pow2 && gs >= 4 && gs <= 256 && (ldx % 4 == 0) && (ldo % 4 == 0) && (!raw || ldraw % 4 == 0) &&
                     aligned16(x) && aligned16(gamma) && aligned16(beta) && ((reinterpret_cast<uintptr_t>(out) & 7u) == 0) &&
                     (sizeof(OutT) == 2 || (aligned16(out) && (!raw || aligned16(raw)))) && (!raw || (reinterpret_cast<uintptr_t>(raw) & 7u) == 0);
    if (vec) {
        const long total4 = (long)M * (C / 4);
        const long blocks = (total4 + 255) / 256;
        // the group size as a template argument: the group sums then run on the cross-lane data path (same bits)
#define ADCDIFF_GN_VEC(LPG)                                                                                                          \
    hipLaunchKernelGGL((gn_silu_vec_kernel<OutT, LPG>), dim3((unsigned)blocks), dim3(256), 0, stream, x, ldx, total4, C / 4, gs / 4, \
                       1.0f / (float)gs, gamma, beta, eps, out, ldo, raw, ldraw, drop_p, drop_seed, drop_layer)
        switch (gs / 4) {
            case 1: ADCDIFF_GN_VEC(1); break;
            case 2: ADCDIFF_GN_VEC(2); break;
            case 4: ADCDIFF_GN_VEC(4); break;
            case 8: ADCDIFF_GN_VEC(8); break;
            case 16: ADCDIFF_GN_VEC(16); break;
            case 32: ADCDIFF_GN_VEC(32); break;
            default: ADCDIFF_GN_VEC(64); break;
        }
#undef ADCDIFF_GN_VEC
    } else {
        if (drop_p > 0.0f) return fail(ADCDIFF_E_BADARG, "groupnorm_silu: dropout needs group sizes that are powers of two >= 4");
        const long total = (long)M * groups;
        const long blocks = (total + 255) / 256;
        hipLaunchKernelGGL((gn_silu_generic_kernel<OutT>), dim3((unsigned)blocks), dim3(256), 0, stream, x, ldx, total, groups, gs,
                           gamma, beta, eps, out, ldo, raw, ldraw);
    }
    return check_launch("groupnorm_silu");
}

int groupnorm_silu(const float* x, int ldx, int M, int C, int groups, const float* gamma, const float* beta, float eps,
                   int out_dtype, void* out, int ldo, void* raw_out, int ldraw, hipStream_t stream, float drop_p, uint64_t drop_seed,
                   uint32_t drop_layer) {
    if (!x || !gamma || !beta || !out || M <= 0 || C <= 0 || groups <= 0 || C % groups != 0 || ldx < C || ldo < C)
        return fail(ADCDIFF_E_BADARG, "groupnorm_silu: bad arguments (M=%d C=%d groups=%d ldx=%d ldo=%d)", M, C, groups, ldx, ldo);
    const double osz = out_dtype == ADCDIFF_BF16 ? 2.0 : 4.0;
    TimingScope ts(ADCDIFF_KIND_GROUPNORM, M, C, 0, 0.0, (double)M * C * (4.0 + osz * (raw_out ? 2 : 1)), stream);
    if (raw_out && ldraw < C) return fail(ADCDIFF_E_BADARG, "groupnorm_silu: ldraw=%d < C=%d", ldraw, C);
    if (out_dtype == ADCDIFF_F32)
        return gn_launch<float>(x, ldx, M, C, groups, gamma, beta, eps, reinterpret_cast<float*>(out), ldo,
                                reinterpret_cast<float*>(raw_out), ldraw, stream, drop_p, drop_seed, drop_layer);
    if (out_dtype == ADCDIFF_BF16)
        return gn_launch<bf16_t>(x, ldx, M, C, groups, gamma, beta, eps, reinterpret_cast<bf16_t*>(out), ldo,
                                 reinterpret_cast<bf16_t*>(raw_out), ldraw, stream, drop_p, drop_seed, drop_layer);
    return fail(ADCDIFF_E_BADARG, "groupnorm_silu: unknown dtype %d", out_dtype);
}

// ------------------------------------------------------------------ Philox fill
__global__ __launch_bounds__(256) void philox_normal_kernel(float* __restrict__ out, int ld, long total4, int q4, uint64_t seed,
                                                            uint32_t stream_id, uint32_t step, uint64_t row_offset) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total4) return;
    const long row = e / q4;
    const int quad = (int)(e - row * q4);
    const f32x4 z = philox_normal4(seed, stream_id, step, row_offset + (uint64_t)row, (uint32_t)quad);
    *reinterpret_cast<f32x4*>(out + row * ld + quad * 4) = z;
}

// ------------------------------------------------------------------ schedule update (one reverse step)
// mean = sra * (x - beta * eps / s1m); out = mean + sigma * z.  12 bytes of HBM traffic per element with
// in-kernel noise (read x, read eps, write out), 16 with injected noise.
template <bool INJECT>
__global__ __launch_bounds__(256) void ddpm_step_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ eps, int lde,
                                                        const float* __restrict__ noise, int ldn, float* __restrict__ out, int ldo,
                                                        long total4, int q4, adcdiff_ddpm_coef c, uint64_t seed, uint32_t step,
                                                        uint64_t row_offset) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total4) return;
    const long row = e / q4;
    const int quad = (int)(e - row * q4);
    const f32x4 xv = *reinterpret_cast<const f32x4*>(x + row * ldx + quad * 4);
    const f32x4 ev = *reinterpret_cast<const f32x4*>(eps + row * lde + quad * 4);
    f32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = c.sqrt_recip_alpha * (xv[i] - (c.beta * ev[i]) / c.sqrt_one_minus_ac);
    if (c.sigma != 0.0f) {
        f32x4 z;
        if constexpr (INJECT) z = *reinterpret_cast<const f32x4*>(noise + row * ldn + quad * 4);
        else z = philox_normal4(seed, ADCDIFF_STREAM_STEP, step, row_offset + (uint64_t)row, (uint32_t)quad);
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = o[i] + c.sigma * z[i];
    }
    *reinterpret_cast<f32x4*>(out + row * ldo + quad * 4) = o;
}

template <bool INJECT>
__global__ __launch_bounds__(256) void ddim_step_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ eps, int lde,
                                                        const float* __restrict__ noise, int ldn, float* __restrict__ out, int ldo,
                                                        long total4, int q4, adcdiff_ddim_coef c, uint64_t seed, uint32_t step,
                                                        uint64_t row_offset) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total4) return;
    const long row = e / q4;
    const int quad = (int)(e - row * q4);
    const f32x4 xv = *reinterpret_cast<const f32x4*>(x + row * ldx + quad * 4);
    const f32x4 ev = *reinterpret_cast<const f32x4*>(eps + row * lde + quad * 4);
    f32x4 dir;
#pragma unroll
    for (int i = 0; i < 4; ++i) dir[i] = c.dir_coef * ev[i];
    if (c.sigma != 0.0f) {
        f32x4 z;
        if constexpr (INJECT) z = *reinterpret_cast<const f32x4*>(noise + row * ldn + quad * 4);
        else z = philox_normal4(seed, ADCDIFF_STREAM_STEP, step, row_offset + (uint64_t)row, (uint32_t)quad);
#pragma unroll
        for (int i = 0; i < 4; ++i) dir[i] = dir[i] + c.sigma * z[i];
    }
    f32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float pred_x0 = (xv[i] - c.c_eps * ev[i]) / c.sqrt_ac_t;
        o[i] = c.sqrt_ac_prev * pred_x0 + dir[i];
    }
    *reinterpret_cast<f32x4*>(out + row * ldo + quad * 4) = o;
}

template <bool INJECT>
__global__ __launch_bounds__(256) void q_sample_kernel(const float* __restrict__ x0, int ldx, const int64_t* __restrict__ t,
                                                       const float* __restrict__ sqrt_ac, const float* __restrict__ sqrt_1mac,
                                                       const float* __restrict__ noise, int ldn, float* __restrict__ out, int ldo,
                                                       float* __restrict__ noise_out, int ldno, long total4, int q4, uint64_t seed,
                                                       uint32_t step, uint64_t row_offset, int num_timesteps) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total4) return;
    const long row = e / q4;
    const int quad = (int)(e - row * q4);
    long tr = t[row];
    tr = tr < 0 ? 0 : tr >= num_timesteps ? num_timesteps - 1 : tr;   // never read outside the schedule tables
    const float a = sqrt_ac[tr], s = sqrt_1mac[tr];
    const f32x4 xv = *reinterpret_cast<const f32x4*>(x0 + row * ldx + quad * 4);
    f32x4 z;
    if constexpr (INJECT) z = *reinterpret_cast<const f32x4*>(noise + row * ldn + quad * 4);
    else z = philox_normal4(seed, ADCDIFF_STREAM_QNOISE, step, row_offset + (uint64_t)row, (uint32_t)quad);
    f32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = a * xv[i] + s * z[i];
    *reinterpret_cast<f32x4*>(out + row * ldo + quad * 4) = o;
    if (noise_out) *reinterpret_cast<f32x4*>(noise_out + row * ldno + quad * 4) = z;
}

static bool rowvec_ok(const void* p, int ld, int D) { return p && aligned16(p) && ld >= D && ld % 4 == 0; }

// ------------------------------------------------------------------ table helpers
// sinusoid[t, j] = sin(t * freq[j]) for j < half, cos(t * freq[j - half]) above (diffusion_model.py:815-816).
__global__ __launch_bounds__(256) void sinusoid_kernel(float* __restrict__ out, int ld, int rows, int half,
                                                       const float* __restrict__ freq) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)rows * half) return;
    const int t = (int)(e / half), j = (int)(e - (long)t * half);
    const float arg = (float)t * freq[j];
    out[(long)t * ld + j] = sinf(arg);
    out[(long)t * ld + half + j] = cosf(arg);
}

// s[r, :] = SiLU(te_table[t_r, :] + cemb[r, :]) in the GEMM operand type: the A operand of time_proj
// (diffusion_model.py:766, 859-862) when a condition makes the time term row dependent.
template <typename OutT>
__global__ __launch_bounds__(256) void temb_silu_kernel(const float* __restrict__ te_table, int ldt, int table_rows,
                                                        const int64_t* __restrict__ t_rows, int t_uniform, const float* __restrict__ cemb, int ldce,
                                                        OutT* __restrict__ out, int ldo, int rows, int H) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)rows * H) return;
    const long r = e / H;
    const int j = (int)(e - r * H);
    long t = t_rows ? (long)t_rows[r] : (t_uniform < 0 ? r : (long)t_uniform);   // t_uniform < 0: row r is timestep r
    t = t < 0 ? 0 : t >= table_rows ? table_rows - 1 : t;                        // never read outside the table
    float v = te_table[t * ldt + j];
    if (cemb) v = v + cemb[r * ldce + j];
    out[r * ldo + j] = (OutT)silu_f(v);
}

// dst[n, col0 + k] = (T) src[n, k]: packs float32 nn.Linear weights into a K-concatenated operand.
template <typename OutT>
__global__ __launch_bounds__(256) void cast_pack_kernel(const float* __restrict__ src, int lds_, OutT* __restrict__ dst, int ldd,
                                                        int col0, int rows, int cols) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)rows * cols) return;
    const long r = e / cols;
    const int k = (int)(e - r * cols);
    dst[r * ldd + col0 + k] = (OutT)src[r * lds_ + k];
}

__global__ __launch_bounds__(256) void add_vec_kernel(const float* a, const float* b, float* out, int n) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < n) out[e] = a[e] + (b ? b[e] : 0.0f);
}

int sinusoid_table(float* out, int ld, int rows, int half, const float* freq, hipStream_t stream) {
    const long total = (long)rows * half;
    hipLaunchKernelGGL(sinusoid_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, out, ld, rows, half, freq);
    return check_launch("sinusoid");
}

int temb_silu(const float* te_table, int ldt, int table_rows, const int64_t* t_rows, int t_uniform, const float* cemb, int ldce, int dtype,
              void* out, int ldo, int rows, int H, hipStream_t stream) {
    TimingScope ts(ADCDIFF_KIND_OTHER, rows, H, 0, 0.0, (double)rows * H * (dtype == ADCDIFF_BF16 ? 6.0 : 8.0), stream);
    const long total = (long)rows * H;
    const dim3 grid((unsigned)((total + 255) / 256));
    if (dtype == ADCDIFF_BF16)
        hipLaunchKernelGGL((temb_silu_kernel<bf16_t>), grid, dim3(256), 0, stream, te_table, ldt, table_rows, t_rows, t_uniform, cemb, ldce,
                           reinterpret_cast<bf16_t*>(out), ldo, rows, H);
    else
        hipLaunchKernelGGL((temb_silu_kernel<float>), grid, dim3(256), 0, stream, te_table, ldt, table_rows, t_rows, t_uniform, cemb, ldce,
                           reinterpret_cast<float*>(out), ldo, rows, H);
    return check_launch("temb_silu");
}

int cast_pack(const float* src, int lds_, int dtype, void* dst, int ldd, int col0, int rows, int cols, hipStream_t stream) {
    TimingScope ts(ADCDIFF_KIND_OTHER, rows, cols, 0, 0.0, (double)rows * cols * (dtype == ADCDIFF_BF16 ? 6.0 : 8.0), stream);
    const long total = (long)rows * cols;
    const dim3 grid((unsigned)((total + 255) / 256));
    if (dtype == ADCDIFF_BF16)
        hipLaunchKernelGGL((cast_pack_kernel<bf16_t>), grid, dim3(256), 0, stream, src, lds_, reinterpret_cast<bf16_t*>(dst), ldd, col0,
                           rows, cols);
    else
        hipLaunchKernelGGL((cast_pack_kernel<float>), grid, dim3(256), 0, stream, src, lds_, reinterpret_cast<float*>(dst), ldd, col0,
                           rows, cols);
    return check_launch("cast_pack");
}

// every weight cast and bias sum of a pack in one launch: a workgroup takes one PACK_CHUNK-element chunk of one entry
__global__ __launch_bounds__(256) void pack_items_kernel(const PackItem* __restrict__ items, int n_items) {
    int lo = 0, hi = n_items - 1;
    const long long chunk = blockIdx.x;
    while (lo < hi) {   // last entry whose chunk_start <= chunk (workgroup-uniform)
        const int mid = (lo + hi + 1) >> 1;
        if (items[mid].chunk_start <= chunk) lo = mid; else hi = mid - 1;
    }
    const PackItem it = items[lo];
#pragma unroll
    for (int h = 0; h < PACK_CHUNK / 1024; ++h) {
        const long long e = (chunk - it.chunk_start) * PACK_CHUNK + h * 1024 + threadIdx.x * 4;
        if (e >= it.numel) continue;
        f32x4 v = *reinterpret_cast<const f32x4*>(it.src + e);
        if (it.add) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(it.add + e);
            v[0] += w[0]; v[1] += w[1]; v[2] += w[2]; v[3] += w[3];
        }
        const long long r = e / it.cols;
        const long long o = r * it.ldd + it.col0 + (e - r * it.cols);
        if (it.dtype == ADCDIFF_BF16) {
            bf16x4 b = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
            *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(it.dst) + o) = b;
        } else {
            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(it.dst) + o) = v;
        }
    }
}

int pack_items(const PackItem* items_dev, int n_items, long long total_chunks, hipStream_t stream) {
    if (n_items <= 0 || total_chunks <= 0) return ADCDIFF_OK;
    TimingScope ts(ADCDIFF_KIND_OTHER, n_items, 0, 0, 0.0, (double)total_chunks * PACK_CHUNK * 6.0, stream);
    hipLaunchKernelGGL(pack_items_kernel, dim3((unsigned)total_chunks), dim3(256), 0, stream, items_dev, n_items);
    return check_launch("pack_items");
}

int add_vec(const float* a, const float* b, float* out, int n, hipStream_t stream) {
    hipLaunchKernelGGL(add_vec_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, a, b, out, n);
    return check_launch("add_vec");
}

}  // namespace adcdiff

using namespace adcdiff;

namespace adcdiff {
namespace {
struct DebugEntry { const char* name; long long def; };
const DebugEntry kDebug[DBG_NKEYS] = {
    {"tile_hints", 0},        // 1: adcdiff_gemm_args.tile_hint is honoured (0: a non-zero hint is rejected)
    {"gemm_tile", 0},         // 128 / 256 / 2561: force that tile kernel for every bf16 launch the shape allows
    {"gemm_ring", 0},         // 1: the ring main loop of the eight-wave 256-tile kernel
    {"gemm_ring2", 1},        // 0: never the two-stage ring of the 128-row kernels
    {"gemm_ring_max", 512},   // launches with at most this many 128-row workgroups take a ring of LDS stages
    {"gemm_w4", 1},           // 0: the eight-wave kernel instead of the four-wave 256-tile kernel
    {"gemm_w4_min_n", 512},   // the narrowest layer (output columns) the four-wave kernel takes
    {"gemm_big_min_k", 256},  // the shortest K for which the four-wave kernel is taken by default (512: the rule before)
    {"gemm_w1", 1},           // 0: the tile kernels instead of the one-wave kernels for few live rows
    {"gemm_w1_rows", 1024},   // live rows up to which the one-wave kernels are taken
    {"gemm_cl", 1},           // 0: never the loader-wave variant of the one-wave kernel
    {"gemm_splitk", 1},       // 0: never the few-row kernel that splits K over workgroups (results then bit-identical to the tile kernels)
    {"gemm_pf_w", 1},         // 0: the four-wave kernel's W loaders do not prefetch their W rows into L2 (the A loaders always prefetch theirs)
    {"gemm_pf_next", 1},      // 0: the four-wave kernel does not touch its successor's first operand lines near the end of its epilogue
    {"step_fold", 1},         // 0: the reverse step as a launch of its own instead of the output GEMM's epilogue
    {"train_transpose", 0},   // 1: backward products on transposed copies (the NT kernel) instead of gemm_tr
    {"train_group_dw", 1},    // 0: every weight-gradient product of the backward pass as a launch of its own
    {"train_fuse_cast", 1},   // 0: bf16 copies and bias column sums of the backward pass by cast launches of their own
    {"train_fused_fwd", 1},   // 0: the training forward with every GroupNorm as a launch of its own
    {"sync", 0},              // 1: synchronise and check after every training launch
};
std::atomic<long long> g_debug[DBG_NKEYS];
std::atomic<bool> g_debug_init{false};
void debug_init() {
    if (g_debug_init.load(std::memory_order_acquire)) return;
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    if (g_debug_init.load(std::memory_order_relaxed)) return;
    for (int i = 0; i < DBG_NKEYS; ++i) g_debug[i].store(kDebug[i].def, std::memory_order_relaxed);
    g_debug_init.store(true, std::memory_order_release);
}
}  // namespace
long long debug_value(int key) {
    debug_init();
    return g_debug[key].load(std::memory_order_relaxed);
}
}  // namespace adcdiff

extern "C" int adcdiff_debug_set(const char* key, long long value) {
    using namespace adcdiff;
    if (!key) return fail(ADCDIFF_E_BADARG, "debug_set: null key");
    debug_init();
    for (int i = 0; i < DBG_NKEYS; ++i)
        if (strcmp(key, kDebug[i].name) == 0) {
            g_debug[i].store(value, std::memory_order_relaxed);
            return ADCDIFF_OK;
        }
    return fail(ADCDIFF_E_BADARG, "debug_set: unknown key '%s'", key);
}
extern "C" int adcdiff_debug_get(const char* key, long long* value) {
    using namespace adcdiff;
    if (!key || !value) return fail(ADCDIFF_E_BADARG, "debug_get: null argument");
    debug_init();
    for (int i = 0; i < DBG_NKEYS; ++i)
        if (strcmp(key, kDebug[i].name) == 0) {
            *value = g_debug[i].load(std::memory_order_relaxed);
            return ADCDIFF_OK;
        }
    return fail(ADCDIFF_E_BADARG, "debug_get: unknown key '%s'", key);
}

extern "C" int adcdiff_abi_version(void) { return ADCDIFF_ABI_VERSION; }
extern "C" const char* adcdiff_last_error(void) { return error_buffer(); }

extern "C" int adcdiff_groupnorm_silu(const float* x, int ldx, int M, int C, int groups, const float* gamma, const float* beta,
                                      float eps, int out_dtype, void* out, int ldo, void* raw_out, int ldraw,
                                      adcdiff_stream_t stream) {
    return groupnorm_silu(x, ldx, M, C, groups, gamma, beta, eps, out_dtype, out, ldo, raw_out, ldraw, as_stream(stream), 0.0f, 0, 0);
}

extern "C" int adcdiff_philox_normal(float* out, int ld, int rows, int cols, uint64_t seed, uint32_t stream_id, uint32_t step,
                                     uint64_t row_offset, adcdiff_stream_t stream) {
    if (rows <= 0 || cols <= 0 || cols % 4 != 0 || !rowvec_ok(out, ld, cols))
        return fail(ADCDIFF_E_BADARG, "philox_normal: rows=%d cols=%d (multiple of 4) ld=%d", rows, cols, ld);
    const long total4 = (long)rows * (cols / 4);
    TimingScope ts(ADCDIFF_KIND_OTHER, rows, cols, 0, 0.0, (double)rows * cols * 4.0, as_stream(stream));
    hipLaunchKernelGGL(philox_normal_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, as_stream(stream), out, ld, total4,
                       cols / 4, seed, stream_id, step, row_offset);
    return check_launch("philox_normal");
}

extern "C" int adcdiff_ddpm_step(const float* x, int ldx, const float* eps, int lde, const float* noise, int ldn, float* out, int ldo,
                                 int B, int D, adcdiff_ddpm_coef coef, uint64_t seed, uint32_t step, uint64_t row_offset,
                                 adcdiff_stream_t stream) {
    if (B <= 0 || D <= 0 || D % 4 != 0 || !rowvec_ok(x, ldx, D) || !rowvec_ok(eps, lde, D) || !rowvec_ok(out, ldo, D) ||
        (noise && !rowvec_ok(noise, ldn, D)))
        return fail(ADCDIFF_E_BADARG, "ddpm_step: B=%d D=%d (multiple of 4), pointers 16-byte aligned, ld >= D", B, D);
    const long total4 = (long)B * (D / 4);
    const dim3 grid((unsigned)((total4 + 255) / 256));
    TimingScope ts(ADCDIFF_KIND_UPDATE, B, D, 0, 0.0, (double)B * D * (noise ? 16.0 : 12.0), as_stream(stream));
    if (noise)
        hipLaunchKernelGGL((ddpm_step_kernel<true>), grid, dim3(256), 0, as_stream(stream), x, ldx, eps, lde, noise, ldn, out, ldo,
                           total4, D / 4, coef, seed, step, row_offset);
    else
        hipLaunchKernelGGL((ddpm_step_kernel<false>), grid, dim3(256), 0, as_stream(stream), x, ldx, eps, lde, noise, ldn, out, ldo,
                           total4, D / 4, coef, seed, step, row_offset);
    return check_launch("ddpm_step");
}

extern "C" int adcdiff_ddim_step(const float* x, int ldx, const float* eps, int lde, const float* noise, int ldn, float* out, int ldo,
                                 int B, int D, adcdiff_ddim_coef coef, uint64_t seed, uint32_t step, uint64_t row_offset,
                                 adcdiff_stream_t stream) {
    if (B <= 0 || D <= 0 || D % 4 != 0 || !rowvec_ok(x, ldx, D) || !rowvec_ok(eps, lde, D) || !rowvec_ok(out, ldo, D) ||
        (noise && !rowvec_ok(noise, ldn, D)))
        return fail(ADCDIFF_E_BADARG, "ddim_step: B=%d D=%d (multiple of 4), pointers 16-byte aligned, ld >= D", B, D);
    const long total4 = (long)B * (D / 4);
    const dim3 grid((unsigned)((total4 + 255) / 256));
    TimingScope ts(ADCDIFF_KIND_UPDATE, B, D, 0, 0.0, (double)B * D * (noise ? 16.0 : 12.0), as_stream(stream));
    if (noise)
        hipLaunchKernelGGL((ddim_step_kernel<true>), grid, dim3(256), 0, as_stream(stream), x, ldx, eps, lde, noise, ldn, out, ldo,
                           total4, D / 4, coef, seed, step, row_offset);
    else
        hipLaunchKernelGGL((ddim_step_kernel<false>), grid, dim3(256), 0, as_stream(stream), x, ldx, eps, lde, noise, ldn, out, ldo,
                           total4, D / 4, coef, seed, step, row_offset);
    return check_launch("ddim_step");
}

extern "C" int adcdiff_q_sample(const float* x0, int ldx, const int64_t* t, const float* sqrt_ac, const float* sqrt_one_minus_ac,
                                int num_timesteps, const float* noise, int ldn, float* out, int ldo, float* noise_out, int ldno,
                                int B, int D, uint64_t seed, uint32_t step, uint64_t row_offset, adcdiff_stream_t stream) {
    if (B <= 0 || D <= 0 || D % 4 != 0 || num_timesteps <= 0 || !t || !sqrt_ac || !sqrt_one_minus_ac || !rowvec_ok(x0, ldx, D) ||
        !rowvec_ok(out, ldo, D) || (noise && !rowvec_ok(noise, ldn, D)) || (noise_out && !rowvec_ok(noise_out, ldno, D)))
        return fail(ADCDIFF_E_BADARG, "q_sample: B=%d D=%d (multiple of 4), pointers 16-byte aligned, ld >= D", B, D);
    const long total4 = (long)B * (D / 4);
    const dim3 grid((unsigned)((total4 + 255) / 256));
    TimingScope ts(ADCDIFF_KIND_OTHER, B, D, 0, 0.0, (double)B * D * ((noise ? 12.0 : 8.0) + (noise_out ? 4.0 : 0.0)), as_stream(stream));
    if (noise)
        hipLaunchKernelGGL((q_sample_kernel<true>), grid, dim3(256), 0, as_stream(stream), x0, ldx, t, sqrt_ac, sqrt_one_minus_ac,
                           noise, ldn, out, ldo, noise_out, ldno, total4, D / 4, seed, step, row_offset, num_timesteps);
    else
        hipLaunchKernelGGL((q_sample_kernel<false>), grid, dim3(256), 0, as_stream(stream), x0, ldx, t, sqrt_ac, sqrt_one_minus_ac,
                           noise, ldn, out, ldo, noise_out, ldno, total4, D / 4, seed, step, row_offset, num_timesteps);
    return check_launch("q_sample");
}

// ---------------------------------------------------------------- latent summaries (numpy-ordered row sums)
namespace adcdiff {

// One wave per row.  Lane (leaf, j) = (lane >> 3, lane & 7) accumulates numpy's partial sum r[j] of 128-element block
// `leaf` (a[j] + a[8 + j] + ... in index order); xor-shuffles 1, 2, 4 combine the eight partial sums in numpy's order,
// 8, 16, 32 the blocks as a balanced tree; rows longer than 1024 take several passes combined the same way.
__global__ __launch_bounds__(256) void latent_summary_kernel(const float* __restrict__ x, long ldx, int B, int D, float* __restrict__ out, long ldo) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B) return;
    const int lane = threadIdx.x & 63, leaf = lane >> 3, j = lane & 7;
    const float* a = x + (long)row * ldx;
    const int nleaf = D >> 7;                 // 1, 2, 4, 8, 16 or 32
    const int passes = nleaf > 8 ? nleaf >> 3 : 1;
    float pass_sum[4];
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        pass_sum[ps] = 0.0f;
        if (ps < passes) {
            const int lf = ps * 8 + leaf;
            float r = 0.0f;
            if (lf < nleaf) {
                const float* b = a + lf * 128 + j;
                r = b[0];
#pragma unroll
                for (int i = 1; i < 16; ++i) r += b[8 * i];
            }
            r += __shfl_xor(r, 1);
            r += __shfl_xor(r, 2);
            r += __shfl_xor(r, 4);
            if (nleaf >= 2) r += __shfl_xor(r, 8);
            if (nleaf >= 4) r += __shfl_xor(r, 16);
            if (nleaf >= 8) r += __shfl_xor(r, 32);
            pass_sum[ps] = r;
        }
    }
    float sum = pass_sum[0];
    if (passes == 2) sum = pass_sum[0] + pass_sum[1];
    if (passes == 4) sum = (pass_sum[0] + pass_sum[1]) + (pass_sum[2] + pass_sum[3]);
    if (lane == 0) {
        float* o = out + (long)row * ldo;
        o[0] = sum / (float)D;
        o[1] = sum;
        o[2] = a[3]; o[3] = a[5]; o[4] = a[8];
    }
}

}  // namespace adcdiff

extern "C" int adcdiff_latent_summary(const float* x, int ldx, int B, int D, float* out, int ld_out, adcdiff_stream_t stream_) {
    using namespace adcdiff;
    if (!x || !out || B <= 0 || ldx < D || ld_out < 5) return fail(ADCDIFF_E_BADARG, "latent_summary: null pointers, B=%d, ldx=%d < D or ld_out=%d < 5", B, ldx, ld_out);
    if (D < 128 || D > 4096 || (D & (D - 1)) != 0)
        return fail(ADCDIFF_E_BADARG, "latent_summary: D=%d; numpy's pairwise order is reproduced for D = 128 * 2^k up to 4096", D);
    TimingScope ts(ADCDIFF_KIND_OTHER, B, D, 0, 0.0, (double)B * (D * 4.0 + 20.0), as_stream(stream_));
    hipLaunchKernelGGL(latent_summary_kernel, dim3((B + 3) / 4), dim3(256), 0, as_stream(stream_), x, (long)ldx, B, D, out, (long)ld_out);
    return check_launch("latent_summary");
}

// ---------------------------------------------------------------- clip_grad_norm_ + AdamW over a tensor list
namespace adcdiff {

// chunk index -> tensor: the last descriptor whose chunk_start <= chunk (descriptors are in chunk order)
__device__ __forceinline__ int opt_find_tensor(const adcdiff_opt_tensor* __restrict__ ts, int n, long long chunk) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (ts[mid].chunk_start <= chunk) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// One workgroup per chunk of ADCDIFF_OPT_CHUNK gradient elements: partial[chunk] = sum of squares, summed in a fixed
// order (per thread in index order, then a butterfly over the wave, then the four waves in order).
__global__ __launch_bounds__(256) void opt_sumsq_kernel(const adcdiff_opt_tensor* __restrict__ ts, int n_tensors, float* __restrict__ partial) {
    __shared__ float wsum[4];
    const long long chunk = blockIdx.x;
    const adcdiff_opt_tensor d = ts[opt_find_tensor(ts, n_tensors, chunk)];
    const long long base = (chunk - d.chunk_start) * ADCDIFF_OPT_CHUNK;
    const long long left = d.numel - base;
    const int cnt = left < ADCDIFF_OPT_CHUNK ? (int)left : ADCDIFF_OPT_CHUNK;
    const float* g = d.grad + base;
    float s = 0.0f;
    if ((reinterpret_cast<uintptr_t>(g) & 15u) == 0) {
        for (int i = threadIdx.x * 4; i + 3 < cnt; i += 1024) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(g + i);
            s += v[0] * v[0]; s += v[1] * v[1]; s += v[2] * v[2]; s += v[3] * v[3];
        }
        for (int i = (cnt & ~3) + threadIdx.x; i < cnt; i += 256) s += g[i] * g[i];
    } else {
        for (int i = threadIdx.x; i < cnt; i += 256) s += g[i] * g[i];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[chunk] = ((wsum[0] + wsum[1]) + wsum[2]) + wsum[3];
}

// scal[0] = ||g||_2, scal[1] = clip factor.  One workgroup; double accumulation in a fixed order.
__global__ __launch_bounds__(1024) void opt_norm_kernel(const float* __restrict__ partial, long long n, float max_norm, float* __restrict__ scal,
                                                        float* __restrict__ grad_norm) {
    __shared__ double wsum[16];
    double s = 0.0;
    for (long long i = threadIdx.x; i < n; i += 1024) s += (double)partial[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int w = 0; w < 16; ++w) tot += wsum[w];
        const float norm = (float)sqrt(tot);
        float clip = 1.0f;
        if (max_norm > 0.0f) {
            clip = max_norm / (norm + 1e-6f);   // torch.nn.utils.clip_grad_norm_: clamp(max_norm / (total_norm + 1e-6), max=1)
            clip = clip > 1.0f ? 1.0f : clip;
        }
        scal[0] = norm; scal[1] = clip;
        if (grad_norm) *grad_norm = norm;
    }
}

struct OptScalars { float decay, w1, beta2, w2, inv_bc2_sqrt, eps, neg_step, l2; };

__device__ __forceinline__ void opt_update1(float& p, float g, float& m, float& v, float clip, const OptScalars& c) {
    g = g * clip;
    if (c.l2 != 0.0f) g = g + c.l2 * p;        // Adam: grad.add(param, alpha=weight_decay)
    p = p * c.decay;                            // AdamW: param.mul_(1 - lr * weight_decay); decay = 1 for Adam
    m = m + c.w1 * (g - m);                     // exp_avg.lerp_(grad, 1 - beta1)
    v = v * c.beta2;                            // exp_avg_sq.mul_(beta2)
    v = v + c.w2 * (g * g);                     //           .addcmul_(grad, grad, value=1 - beta2)
    const float denom = __builtin_sqrtf(v) * c.inv_bc2_sqrt + c.eps;   // (exp_avg_sq.sqrt() / sqrt(bias_correction2)).add_(eps)
    p = p + c.neg_step * (m / denom);           // param.addcdiv_(exp_avg, denom, value=-lr / bias_correction1)
}

__global__ __launch_bounds__(256) void opt_adamw_kernel(const adcdiff_opt_tensor* __restrict__ ts, int n_tensors, const float* __restrict__ scal,
                                                        const OptScalars c) {
    const long long chunk = blockIdx.x;
    const adcdiff_opt_tensor d = ts[opt_find_tensor(ts, n_tensors, chunk)];
    const long long base = (chunk - d.chunk_start) * ADCDIFF_OPT_CHUNK;
    const long long left = d.numel - base;
    const int cnt = left < ADCDIFF_OPT_CHUNK ? (int)left : ADCDIFF_OPT_CHUNK;
    const float clip = scal[1];
    float* p = d.param + base;
    const float* g = d.grad + base;
    float* m = d.exp_avg + base;
    float* v = d.exp_avg_sq + base;
    const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                       reinterpret_cast<uintptr_t>(v)) & 15u) == 0;
    if (vec) {
        for (int i = threadIdx.x * 4; i + 3 < cnt; i += 1024) {
            f32x4 pv = *reinterpret_cast<f32x4*>(p + i), mv = *reinterpret_cast<f32x4*>(m + i), vv = *reinterpret_cast<f32x4*>(v + i);
            const f32x4 gv = *reinterpret_cast<const f32x4*>(g + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float pe = pv[e], me = mv[e], ve = vv[e];
                opt_update1(pe, gv[e], me, ve, clip, c);
                pv[e] = pe; mv[e] = me; vv[e] = ve;
            }
            *reinterpret_cast<f32x4*>(p + i) = pv;
            *reinterpret_cast<f32x4*>(m + i) = mv;
            *reinterpret_cast<f32x4*>(v + i) = vv;
        }
    }
    for (int i = (vec ? (cnt & ~3) : 0) + threadIdx.x; i < cnt; i += 256) opt_update1(p[i], g[i], m[i], v[i], clip, c);
}

}  // namespace adcdiff

extern "C" size_t adcdiff_clip_adamw_workspace_bytes(long long total_chunks) {
    return total_chunks <= 0 ? 0 : (size_t)total_chunks * 4 + 256;
}

namespace adcdiff {

// The three launches of the clipped step.  scal: 64 floats ([0] norm, [1] clip factor); partial: one float per chunk.  Every
// word of both that a kernel reads was written earlier in the same call, so neither has to be cleared.
static int clip_adamw_launch(const adcdiff_opt_tensor* tensors, int n_tensors, long long total_chunks, const adcdiff_adamw_args* a,
                             float* grad_norm, float* scal, float* partial, hipStream_t st) {
    TimingScope ts(ADCDIFF_KIND_OTHER, n_tensors, 0, 0, 0.0, (double)total_chunks * ADCDIFF_OPT_CHUNK * 32.0, st);
    hipLaunchKernelGGL(opt_sumsq_kernel, dim3((unsigned)total_chunks), dim3(256), 0, st, tensors, n_tensors, partial);
    hipLaunchKernelGGL(opt_norm_kernel, dim3(1), dim3(1024), 0, st, partial, total_chunks, (float)a->max_norm, scal, grad_norm);
    // scalars as torch forms them: python floats (double) rounded to float where they meet the float32 tensors
    const double bc1 = 1.0 - pow(a->beta1, a->step), bc2 = 1.0 - pow(a->beta2, a->step);
    OptScalars c;
    c.decay = a->decoupled ? (float)(1.0 - a->lr * a->weight_decay) : 1.0f;
    c.l2 = a->decoupled ? 0.0f : (float)a->weight_decay;
    c.w1 = (float)(1.0 - a->beta1);
    c.beta2 = (float)a->beta2;
    c.w2 = (float)(1.0 - a->beta2);
    c.inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
    c.eps = (float)a->eps;
    c.neg_step = (float)(-a->lr / bc1);
    hipLaunchKernelGGL(opt_adamw_kernel, dim3((unsigned)total_chunks), dim3(256), 0, st, tensors, n_tensors, scal, c);
    return check_launch("clip_adamw_step");
}

static int clip_adamw_check(const char* who, const adcdiff_opt_tensor* tensors, int n_tensors, long long total_chunks, const adcdiff_adamw_args* a) {
    if (!tensors || !a || n_tensors <= 0 || total_chunks <= 0 || total_chunks > 0x7fffffffLL)
        return fail(ADCDIFF_E_BADARG, "%s: null table / args or %d tensors, %lld chunks", who, n_tensors, total_chunks);
    if (a->step < 1 || !(a->beta1 >= 0.0 && a->beta1 < 1.0) || !(a->beta2 >= 0.0 && a->beta2 < 1.0) || !(a->eps >= 0.0))
        return fail(ADCDIFF_E_BADARG, "%s: step %d, betas (%g, %g), eps %g", who, a->step, a->beta1, a->beta2, a->eps);
    return ADCDIFF_OK;
}

// Scratch of the entry point that allocates nothing: 64 scalars + one partial sum per chunk, in the code object's own
// device memory (every GPU the library runs on holds its copy until the process ends).
static __device__ __align__(256) float scratch[64 + ADCDIFF_OPT_NOALLOC_MAX_CHUNKS];

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
float* g_scratch_ptr[kMaxDevices] = {};   // device address of `scratch` on each device, looked up on first use
}  // namespace

// Device address of this device's copy of `scratch` (the array's name is only a host-side handle for the lookup).
static int scratch_for_current_device(float** out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return fail(ADCDIFF_E_LAUNCH, "clip_adamw_step_noalloc: hipGetDevice: %s", hipGetErrorString(e));
    if (dev < 0 || dev >= kMaxDevices) return fail(ADCDIFF_E_BADARG, "clip_adamw_step_noalloc: device %d outside 0..%d", dev, kMaxDevices - 1);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        float* ptr = nullptr;
        e = hipGetSymbolAddress(reinterpret_cast<void**>(&ptr), HIP_SYMBOL(scratch));
        if (e != hipSuccess || !ptr) return fail(ADCDIFF_E_LAUNCH, "clip_adamw_step_noalloc: hipGetSymbolAddress: %s", hipGetErrorString(e));
        g_scratch_ptr[dev] = ptr;
    }
    *out = g_scratch_ptr[dev];
    return ADCDIFF_OK;
}

}  // namespace adcdiff

extern "C" int adcdiff_clip_adamw_step(const adcdiff_opt_tensor* tensors, int n_tensors, long long total_chunks, const adcdiff_adamw_args* a,
                                       float* grad_norm, void* workspace, size_t workspace_bytes, adcdiff_stream_t stream_) {
    using namespace adcdiff;
    TRY_RC(clip_adamw_check("clip_adamw_step", tensors, n_tensors, total_chunks, a));
    if (!workspace || !aligned16(workspace) || workspace_bytes < adcdiff_clip_adamw_workspace_bytes(total_chunks))
        return fail(ADCDIFF_E_WORKSPACE, "clip_adamw_step: workspace of %zu bytes, %zu needed", workspace_bytes,
                    adcdiff_clip_adamw_workspace_bytes(total_chunks));
    float* scal = reinterpret_cast<float*>(workspace);          // [0] norm, [1] clip factor
    return clip_adamw_launch(tensors, n_tensors, total_chunks, a, grad_norm, scal, scal + 64, as_stream(stream_));
}

extern "C" int adcdiff_clip_adamw_step_noalloc(const adcdiff_opt_tensor* tensors, int n_tensors, long long total_chunks,
                                               const adcdiff_adamw_args* a, float* grad_norm, adcdiff_stream_t stream_) {
    using namespace adcdiff;
    TRY_RC(clip_adamw_check("clip_adamw_step_noalloc", tensors, n_tensors, total_chunks, a));
    if (total_chunks > ADCDIFF_OPT_NOALLOC_MAX_CHUNKS)
        return fail(ADCDIFF_E_WORKSPACE, "clip_adamw_step_noalloc: %lld chunks exceed the built-in scratch of %d chunks (%d parameters); use "
                    "adcdiff_clip_adamw_step with a workspace", total_chunks, ADCDIFF_OPT_NOALLOC_MAX_CHUNKS,
                    ADCDIFF_OPT_NOALLOC_MAX_CHUNKS * ADCDIFF_OPT_CHUNK);
    float* scal = nullptr;
    TRY_RC(scratch_for_current_device(&scal));
    return clip_adamw_launch(tensors, n_tensors, total_chunks, a, grad_norm, scal, scal + 64, as_stream(stream_));
}

extern "C" int adcdiff_timing_begin(int max_records) {
    if (max_records <= 0) return fail(ADCDIFF_E_BADARG, "timing_begin: max_records=%d", max_records);
    if (g_timing.on) return fail(ADCDIFF_E_BADARG, "timing_begin: already active");
    if (g_timing.cap < max_records) {
        for (int i = 0; i < (g_timing.cap ? g_timing.cap + 1 : 0); ++i) (void)hipEventDestroy(g_timing.ev[i]);
        delete[] g_timing.ev;
        delete[] g_timing.rec;
        g_timing.ev = new hipEvent_t[(size_t)max_records + 1];
        g_timing.rec = new adcdiff_timing_record[max_records];
        g_timing.cap = 0;
        for (int i = 0; i < max_records + 1; ++i) {
            hipError_t e = hipEventCreate(&g_timing.ev[i]);
            if (e != hipSuccess) return fail(ADCDIFF_E_LAUNCH, "timing_begin: hipEventCreate: %s", hipGetErrorString(e));
        }
        g_timing.cap = max_records;
    }
    g_timing.used = 0;
    g_timing.on = true;
    return ADCDIFF_OK;
}

extern "C" int adcdiff_timing_end(adcdiff_timing_record* out, int capacity, int* count) {
    if (!g_timing.on) return fail(ADCDIFF_E_BADARG, "timing_end: not active");
    g_timing.on = false;
    if (!out || !count) return fail(ADCDIFF_E_BADARG, "timing_end: null output");
    const int used = g_timing.used;
    if (used > 0) {
        hipError_t e = hipEventRecord(g_timing.ev[used], g_timing.last_stream);   // closes the last launch's span
        if (e == hipSuccess) e = hipEventSynchronize(g_timing.ev[used]);
        if (e != hipSuccess) return fail(ADCDIFF_E_LAUNCH, "timing_end: %s", hipGetErrorString(e));
    }
    const int n = used < capacity ? used : capacity;
    for (int i = 0; i < n; ++i) {
        hipError_t e = hipEventElapsedTime(&g_timing.rec[i].ms, g_timing.ev[i], g_timing.ev[i + 1]);
        if (e != hipSuccess) return fail(ADCDIFF_E_LAUNCH, "timing_end: %s", hipGetErrorString(e));
        out[i] = g_timing.rec[i];
    }
    *count = n;
    return ADCDIFF_OK;
}

// ---- what the matrix cores sustain inside the power limit ------------------------------------------------------------------
// 64 v_mfma_f32_16x16x32_bf16 per iteration and wave on register operands with random signs, exponents and mantissas (no memory
// traffic at all), one wave per SIMD holding 256 accumulator registers, as in the four-wave GEMM.  The shader clock (s_memtime
// ticks per 10 ns of the constant 100 MHz counter) is read by every workgroup.
namespace adcdiff {
typedef __bf16 probe_bf16x8 __attribute__((ext_vector_type(8)));
typedef float probe_f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 1) void matrix_core_probe_kernel(float* out, int iters, long long* ticks) {
    probe_bf16x8 a[8], b[8];
    for (int i = 0; i < 8; ++i)
        for (int k = 0; k < 8; ++k) {   // |x| in [0.25, 4)
            unsigned h = (threadIdx.x * 64 + i * 8 + k + blockIdx.x * 16384) * 2654435761u;
            h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
            a[i][k] = __builtin_bit_cast(__bf16, (unsigned short)((h & 0x807f) | ((125 + ((h >> 8) & 3)) << 7)));
            b[i][k] = __builtin_bit_cast(__bf16, (unsigned short)(((h >> 16) & 0x807f) | ((125 + ((h >> 28) & 3)) << 7)));
        }
    const long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    probe_f32x4 acc[8][8] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    float s = 0.0f;
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 8; ++j) s += acc[i][j][0] + acc[i][j][3];
    const long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { ticks[blockIdx.x * 2] = c1 - c0; ticks[blockIdx.x * 2 + 1] = r1 - r0; }
}
}  // namespace adcdiff

extern "C" int adcdiff_matrix_core_probe(int iters, int launches, double* tflops, double* clock_mhz) {
    using namespace adcdiff;
    if (iters <= 0 || launches <= 0 || !tflops || !clock_mhz) return fail(ADCDIFF_E_BADARG, "matrix_core_probe: iters=%d launches=%d", iters, launches);
    int dev = 0, cus = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess || cus <= 0) return fail(ADCDIFF_E_LAUNCH, "matrix_core_probe: device query: %s", hipGetErrorString(e));
    float* out = nullptr;
    long long* ticks = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = ADCDIFF_OK;
    auto step = [&](hipError_t err, const char* what) {
        if (rc == ADCDIFF_OK && err != hipSuccess) rc = fail(ADCDIFF_E_LAUNCH, "matrix_core_probe: %s: %s", what, hipGetErrorString(err));
        return rc == ADCDIFF_OK;
    };
    if (step(hipMalloc(&out, (size_t)cus * 256 * sizeof(float)), "hipMalloc") && step(hipMalloc(&ticks, (size_t)cus * 16), "hipMalloc") &&
        step(hipEventCreate(&e0), "hipEventCreate") && step(hipEventCreate(&e1), "hipEventCreate")) {
        const int settle = launches / 4 + 1;   // untimed launches in front: the clock needs some milliseconds to settle
        for (int w = 0; w < settle; ++w) hipLaunchKernelGGL(matrix_core_probe_kernel, dim3(cus), dim3(256), 0, nullptr, out, iters, ticks);
        step(hipEventRecord(e0, nullptr), "hipEventRecord");
        for (int w = 0; w < launches; ++w) hipLaunchKernelGGL(matrix_core_probe_kernel, dim3(cus), dim3(256), 0, nullptr, out, iters, ticks);
        step(hipEventRecord(e1, nullptr), "hipEventRecord");
        step(hipEventSynchronize(e1), "hipEventSynchronize");
        float ms = 0.0f;
        std::vector<long long> h((size_t)cus * 2);
        if (step(hipEventElapsedTime(&ms, e0, e1), "hipEventElapsedTime") &&
            step(hipMemcpy(h.data(), ticks, h.size() * sizeof(long long), hipMemcpyDeviceToHost), "hipMemcpy")) {
            double mhz = 0.0;
            for (int i = 0; i < cus; ++i) mhz += 100.0 * (double)h[2 * i] / (double)(h[2 * i + 1] > 0 ? h[2 * i + 1] : 1);
            *clock_mhz = mhz / cus;
            // per iteration and wave: 64 instructions of 2 * 16 * 16 * 32 flop
            *tflops = 64.0 * 16384.0 * iters * 4.0 * cus * launches / ((double)ms * 1e-3) * 1e-12;
        }
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (out) (void)hipFree(out);
    if (ticks) (void)hipFree(ticks);
    return rc;
}
