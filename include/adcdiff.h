/*
 * adcdiff.h -- C ABI of the MI355X (gfx950) DDPM sampling library.
 *
 * The reference (zhangmask/2025-iFlytek-ADC-Generation-Challenge-...) has no FFI, plugin or
 * operator layer: its hot path is the Python class surface of src/models/diffusion_model.py.
 * Each entry point below therefore cites the reference Python it replaces (file:line under
 * the reference root); the Python mirror of that class surface (package
 * models/diffusion_model.py) is the only caller.
 *
 * Conventions
 *   - every function returns 0 (ADCDIFF_OK) or a negative ADCDIFF_E_* code and never throws;
 *     adcdiff_last_error() returns a thread-local message for the last failure;
 *   - all tensor arguments are caller-owned DEVICE pointers, row-major, leading dimensions in
 *     elements; the library allocates no device memory: packed weights live in a caller
 *     provided arena, intermediates in a caller provided workspace;
 *   - every call only enqueues work on the given stream (a hipStream_t passed as void*, NULL =
 *     the null stream) and returns without synchronising;
 *   - ADCDIFF_F32 computes with float32 operands on v_mfma_f32_32x32x2_f32, ADCDIFF_BF16 with
 *     bfloat16 operands on v_mfma_f32_16x16x32_bf16; accumulation, GroupNorm statistics, the
 *     latent x_t and the schedule update are float32 in both.
 */
#ifndef ADCDIFF_H
#define ADCDIFF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ADCDIFF_ABI_VERSION 3

enum { ADCDIFF_OK = 0, ADCDIFF_E_BADARG = -1, ADCDIFF_E_LAUNCH = -2, ADCDIFF_E_ARCH = -3, ADCDIFF_E_WORKSPACE = -4 };
enum { ADCDIFF_F32 = 0, ADCDIFF_BF16 = 1 };

/* Philox stream ids of the noise definition (counter = (col/4, row, step, stream)). */
enum { ADCDIFF_STREAM_XT = 1, ADCDIFF_STREAM_STEP = 2, ADCDIFF_STREAM_QNOISE = 3, ADCDIFF_STREAM_TDRAW = 4 };

typedef void* adcdiff_stream_t;

int adcdiff_abi_version(void);
const char* adcdiff_last_error(void);

/* ------------------------------------------------------------------ single kernels ---- */

/* v = [A0 | A1 | A2] * W^T + bias + rowbias + residual: nn.Linear (diffusion_model.py:852-853, 799-800,
 * 695-697, 733, 871) with the K dimension split over up to three A operands, which is how
 * torch.cat([h, skip]) (:782) and linear2 + residual_proj (:885, :899, :907) are consumed without
 * materialising the concatenation.  W is [N, K0+K1+K2] row-major (nn.Linear layout).
 * Outputs (any subset): v in float32, v in the operand type (the next block's residual_proj operand),
 * and up to two SiLU(GroupNorm(v)) in the operand type -- the GroupNorm + SiLU that the CONSUMER of this
 * layer applies first (norm2 before linear2 :896-897, the next block's norm1 :888-889, the decoder's
 * norm1 over its skip half, output_norm :786-787), computed here from the float32 accumulators so the
 * layer output is never re-read to be normalised.
 * Requirements: M % 128 == 0, N % 64 == 0, every K_s a multiple of 32 (F32) / 64 (BF16), all pointers
 * 16-byte aligned, every leading dimension * element size a multiple of 16 bytes. */
typedef struct adcdiff_gn_out {
    const float* gamma;        /* GroupNorm weight for this GEMM's N columns */
    const float* beta;         /* GroupNorm bias */
    void* out;                 /* [M, ldo] in the operand type */
    int ldo;
    int group_size;            /* channels per group: a power of two, 4..128 (4..64 when N % 128 != 0), dividing N */
} adcdiff_gn_out;

/* Schedule update applied to v in the GEMM's epilogue, before any output is formed: the denoiser's output layer (whose v is
 * the predicted noise, diffusion_model.py:788) can apply the reverse step of p_sample (:207-216) or ddim_step (:350-369) to the
 * latent itself, so that the predicted noise never goes to memory and the step's outputs are what the NEXT step reads:
 *   kind 1 (DDPM): v <- c[0] * (x - c[1] * v / c[2]) + c[3] * z          c = sqrt_recip_alpha, beta, sqrt_one_minus_ac, sigma
 *   kind 2 (DDIM): v <- c[2] * ((x - c[0] * v) / c[1]) + (c[3] * v + c[4] * z)   c = c_eps, sqrt_ac_t, sqrt_ac_prev, dir_coef, sigma
 * in the operation order of adcdiff_ddpm_step / adcdiff_ddim_step (bit-identical results); x [M, ldx] float32 is read per row
 * (out_f32 may alias it: a row's elements are read before they are written); z = noise[r, c] when noise != NULL, else Philox
 * (seed, ADCDIFF_STREAM_STEP, step, row_offset + r); c[sigma] == 0 draws nothing.  Not combined with act, split_k. */
typedef struct adcdiff_step_update {
    int kind;
    float c[5];
    const float* x;
    int ldx;
    const float* noise;
    int ldn;
    uint64_t seed;
    uint32_t step;
    uint64_t row_offset;
} adcdiff_step_update;

typedef struct adcdiff_gemm_args {
    int dtype;                 /* operand type of A, W, residual, out_op and gn outputs: ADCDIFF_F32 / ADCDIFF_BF16 */
    int M, N;
    int m_valid;               /* rows >= m_valid are computed but read no per-row side input and store nothing; 0 = M */
    int nseg;                  /* 1..3 */
    const void* A[3];
    int lda[3];
    int K[3];
    const void* W;
    int ldw;
    const float* bias;         /* [N] or NULL */
    const float* rowbias;      /* NULL, or a table added per output row: row r reads
                                  rowbias[(rowbias_idx ? rowbias_idx[r] : r) * ldrb + n];
                                  ldrb == 0 broadcasts one row (the hoisted time_proj term, :893) */
    long long ldrb;
    const int64_t* rowbias_idx;/* [M] timestep per row, or NULL */
    int rowbias_rows;          /* rows of the rowbias table when rowbias_idx is given: indices are clamped to 0..rowbias_rows-1
                                  (the reference raises an IndexError there; a kernel must not read out of bounds); 0 = no clamp */
    const void* residual;      /* [M, N] in the operand type, added to v (Identity residual, :873), or NULL */
    int ldr;
    int act;                   /* 0 none, 1 SiLU on v before it is stored (:820); not combined with gn outputs */
    float* out_f32;            /* optional float32 output of v */
    int ldo_f32;
    void* out_op;              /* optional output of v in the operand type */
    int ldo_op;
    int n_gn;                  /* 0..2 fused SiLU(GroupNorm(v)) outputs */
    adcdiff_gn_out gn[2];
    float gn_eps;
    int tile_hint;             /* 0 = the library chooses the kernel from the shape (the only value a product caller passes).  Non-zero
                                  values force one of the kernels for bit-identity tests and A/B measurements and are rejected unless
                                  adcdiff_debug_set("tile_hints", 1) was called; their meaning is listed at adcdiff_debug_set. */
    int split_k;               /* 0 / 1 = off.  S > 1: the K dimension (one segment, a multiple of S * 64 bf16 / S * 32 f32) is cut
                                  into S equal slices computed by separate workgroups; slice s writes its partial product to
                                  out_f32 + s * split_stride and the caller sums the S partials in a fixed order.  For
                                  weight-gradient GEMMs (K = batch) whose M x N gives too few tiles to fill the chip.  Only with
                                  out_f32 as the sole output and no bias / rowbias / residual / act. */
    long long split_stride;    /* floats between consecutive partial outputs */
    adcdiff_step_update update;/* kind 0 = none; see adcdiff_step_update */
    void* splitk_scratch;      /* optional device buffer (16-byte aligned) that lets a bf16 launch with at most 64 live rows split K over
                                  workgroups inside the library: the first 4096 bytes are arrival counters that must be ZERO before the
                                  first launch that uses the buffer (every launch leaves them zero), the rest holds partial sums,
                                  4 * S * 64 * N bytes for S <= 16 splits (4096 + 4 MiB covers N <= 1024, 8 MiB N <= 2048).  Launches that
                                  share the buffer must be ordered by one stream.  The result then equals the other kernels' up to float32
                                  summation order (S partial sums added in a fixed order), not bit for bit.  NULL = never split. */
    long long splitk_scratch_bytes;
} adcdiff_gemm_args;
int adcdiff_gemm(const adcdiff_gemm_args* args, adcdiff_stream_t stream);

/* bf16 GEMM with contraction-major operands, float32 output: C[M, N] = sum_k A(m, k) * B[k, N].
 * B is [K, ldb] row-major (the contraction index is the ROW); A is [K, lda] row-major when ta != 0, else [M, lda] row-major.
 * These are the two products of the backward pass of y = x W^T (diffusion_model.py:436 loss.backward() through the nn.Linear
 * layers of :852-853, 867-870, 878) taken on the operands as the forward pass left them, with no transposed copies:
 *   d(x) = d(y) W        ta = 0, A = d(y) [rows, out], B = W [out, in]
 *   d(W) = d(y)^T x      ta = 1, A = d(y) [rows, out], B = x [rows, in]     (K = rows)
 * M, N multiples of 128; K a multiple of 64 * slices.  slices > 1 cuts the contraction into equal parts computed by separate
 * workgroups, part s writing C + s * c_slice (the caller sums the parts in order). */
int adcdiff_gemm_tr(int ta, const void* A, long long lda, const void* B, long long ldb, float* C, long long ldc,
                    int M, int N, int K, int slices, long long c_slice, adcdiff_stream_t stream);

/* out = SiLU(GroupNorm(x)) for x [M, C] float32 with `groups` contiguous channel groups, biased variance,
 * eps inside the root (nn.GroupNorm + F.silu, diffusion_model.py:888-889, 896-897, 786-787).  out has
 * out_dtype; raw_out (optional, same dtype) receives x itself converted to that type: the
 * residual_proj operand (:885). */
int adcdiff_groupnorm_silu(const float* x, int ldx, int M, int C, int groups, const float* gamma,
                           const float* beta, float eps, int out_dtype, void* out, int ldo,
                           void* raw_out, int ldraw, adcdiff_stream_t stream);

/* N(0,1) noise of the shared definition: out[r, c] for global row row_offset + r. cols % 4 == 0. */
int adcdiff_philox_normal(float* out, int ld, int rows, int cols, uint64_t seed, uint32_t stream_id,
                          uint32_t step, uint64_t row_offset, adcdiff_stream_t stream);

/* One reverse step for a batch-uniform timestep (diffusion_model.py:197-216):
 *   mean = sqrt_recip_alpha * (x - beta * eps / sqrt_one_minus_ac);  out = mean + sigma * z
 * z = noise[r, c] when noise != NULL, else Philox (seed, ADCDIFF_STREAM_STEP, step, row_offset + r);
 * sigma == 0 (the t == 0 branch, :211-212) draws nothing.  out may alias x. */
typedef struct adcdiff_ddpm_coef { float sqrt_recip_alpha, beta, sqrt_one_minus_ac, sigma; } adcdiff_ddpm_coef;
int adcdiff_ddpm_step(const float* x, int ldx, const float* eps, int lde, const float* noise, int ldn,
                      float* out, int ldo, int B, int D, adcdiff_ddpm_coef coef, uint64_t seed,
                      uint32_t step, uint64_t row_offset, adcdiff_stream_t stream);

/* One DDIM step (diffusion_model.py:350-369):
 *   pred_x0 = (x - c_eps * eps) / sqrt_ac_t;  out = sqrt_ac_prev * pred_x0 + dir_coef * eps + sigma * z */
typedef struct adcdiff_ddim_coef { float c_eps, sqrt_ac_t, sqrt_ac_prev, dir_coef, sigma; } adcdiff_ddim_coef;
int adcdiff_ddim_step(const float* x, int ldx, const float* eps, int lde, const float* noise, int ldn,
                      float* out, int ldo, int B, int D, adcdiff_ddim_coef coef, uint64_t seed,
                      uint32_t step, uint64_t row_offset, adcdiff_stream_t stream);

/* Forward diffusion with per-row timesteps (diffusion_model.py:144-163); timesteps outside 0..num_timesteps-1 are clamped:
 *   out = sqrt_ac[t_r] * x0 + sqrt_one_minus_ac[t_r] * z;  z = noise or Philox (ADCDIFF_STREAM_QNOISE, step).
 * When noise_out != NULL the z that was used is written there (p_losses needs it as the target, :176-183). */
int adcdiff_q_sample(const float* x0, int ldx, const int64_t* t, const float* sqrt_ac, const float* sqrt_one_minus_ac,
                     int num_timesteps, const float* noise, int ldn, float* out, int ldo, float* noise_out,
                     int ldno, int B, int D, uint64_t seed, uint32_t step, uint64_t row_offset,
                     adcdiff_stream_t stream);

/* ------------------------------------------------------------------ denoiser -------- */

/* Float32 parameters of one ResidualBlock in the reference's layout (diffusion_model.py:849-873). */
typedef struct adcdiff_block_params {
    const float *norm1_w, *norm1_b, *norm2_w, *norm2_b;
    const float *linear1_w, *linear1_b, *linear2_w, *linear2_b;
    const float *time_proj_w, *time_proj_b;
    const float *residual_w, *residual_b;   /* NULL when in_dim == out_dim (nn.Identity) */
} adcdiff_block_params;

/* UNetModel parameters (diffusion_model.py:690-733); blocks = encoder.0-3, middle.0-1, decoder.0-3. */
typedef struct adcdiff_unet_params {
    int input_dim, hidden_dim, output_dim, num_timesteps;
    const float *time_lin1_w, *time_lin1_b, *time_lin2_w, *time_lin2_b;
    const float *cond0_w, *cond0_b, *cond2_w, *cond2_b;
    adcdiff_block_params blocks[10];
    const float *out_norm_w, *out_norm_b, *out_w, *out_b;
    const float* sinusoid_freq;   /* [hidden_dim / 2] float32: exp(arange(half) * -ln(1e4)/(half-1)) (:812-814) */
} adcdiff_unet_params;

typedef struct adcdiff_unet adcdiff_unet;   /* host-side plan: sizes and pointers into the arena */

/* Bytes of device arena for packed weights and the per-timestep tables, and of workspace for a batch. */
size_t adcdiff_unet_arena_bytes(int input_dim, int hidden_dim, int output_dim, int num_timesteps, int dtype);
size_t adcdiff_unet_workspace_bytes(int input_dim, int hidden_dim, int output_dim, int dtype, int batch);

/* Create a plan over caller-owned parameters and arena (host-side only: nothing is copied to the device or launched). */
int adcdiff_unet_create(const adcdiff_unet_params* params, int dtype, void* arena, size_t arena_bytes,
                        adcdiff_unet** out);
void adcdiff_unet_destroy(adcdiff_unet* net);

/* (Re)pack the weights into the arena (the first call also uploads the plan's table of packing entries, a few KB, as a
 * stream-ordered host-to-device copy) and rebuild the tables that depend only on the timestep:
 * TimeEmbedding(t) (diffusion_model.py:802-823) and every block's time_proj(SiLU(.)) (:859-862, :893)
 * for t = 0..num_timesteps-1.  Call after the parameters change.  workspace as for batch 1024. */
int adcdiff_unet_pack(adcdiff_unet* net, void* workspace, size_t workspace_bytes, adcdiff_stream_t stream);

/* condition_embed (diffusion_model.py:694-698, :765): cemb [B, hidden] float32. */
int adcdiff_unet_condition_embed(adcdiff_unet* net, const float* cond, int ldc, int B, float* cemb, int ldce,
                                 void* workspace, size_t workspace_bytes, adcdiff_stream_t stream);

/* UNetModel.forward (diffusion_model.py:749-789): eps [B, output_dim] float32.
 * t_rows (device int64 [B]) gives a timestep per row; when NULL every row uses t_uniform.
 * cemb is the output of adcdiff_unet_condition_embed or NULL (unconditional). */
int adcdiff_unet_forward(adcdiff_unet* net, const float* x, int ldx, int B, const int64_t* t_rows,
                         int t_uniform, const float* cemb, int ldce, float* eps, int lde, void* workspace,
                         size_t workspace_bytes, adcdiff_stream_t stream);

/* p_sample_loop (diffusion_model.py:253-261): x holds x_T on entry and the result on return.
 * coef is a HOST array indexed by timestep; steps run t_start, t_start-1, ..., t_stop.
 * noise (optional, device) holds the injected noise of step t at noise + (t_start - t) * noise_step_stride
 * as [B, D] with leading dimension D; NULL selects in-kernel Philox with (seed, row_offset).
 * cemb: NULL, [B, hidden] with leading dimension ldce >= hidden, or ONE row with ldce = 0 = the whole batch shares the
 * condition (the generator's case): t_emb + c_emb is then the same for every row and every block's time projection of it
 * becomes a per-timestep row bias, as without a condition, built once per call into a table that belongs to the PLAN (its
 * arena), not to the workspace: sampling calls that share a plan and pass a one-row condition must share a stream. */
int adcdiff_ddpm_sample_loop(adcdiff_unet* net, float* x, int ldx, int B, const float* cemb, int ldce,
                             const adcdiff_ddpm_coef* coef, int t_start, int t_stop, const float* noise,
                             long long noise_step_stride, uint64_t seed, uint64_t row_offset, void* workspace,
                             size_t workspace_bytes, adcdiff_stream_t stream);

/* ddim_sample_loop (diffusion_model.py:325-340): nsteps steps, timesteps[i] / coef[i] HOST arrays in
 * execution order; noise layout as above indexed by execution step i; cemb / ldce as above. */
int adcdiff_ddim_sample_loop(adcdiff_unet* net, float* x, int ldx, int B, const float* cemb, int ldce,
                             const int* timesteps, const adcdiff_ddim_coef* coef, int nsteps, const float* noise,
                             long long noise_step_stride, uint64_t seed, uint64_t row_offset, void* workspace,
                             size_t workspace_bytes, adcdiff_stream_t stream);

/* ------------------------------------------------------------------ training -------- */

/* Gradient destinations, float32, same shapes as the parameters of adcdiff_unet_params.  Bias and GroupNorm
 * gradients are ACCUMULATED with atomics and must be zero on entry; weight gradients are overwritten.
 * The condition_embed gradients (cond*) are written only after a conditional training forward (p_losses with a condition,
 * diffusion_model.py:165-184 with :764-766); the reference's train_step itself never passes one (:427, :450). */
typedef struct adcdiff_block_grads {
    float *norm1_w, *norm1_b, *norm2_w, *norm2_b;
    float *linear1_w, *linear1_b, *linear2_w, *linear2_b;
    float *time_proj_w, *time_proj_b;
    float *residual_w, *residual_b;
} adcdiff_block_grads;
typedef struct adcdiff_unet_grads {
    float *time_lin1_w, *time_lin1_b, *time_lin2_w, *time_lin2_b;
    adcdiff_block_grads blocks[10];
    float *out_norm_w, *out_norm_b, *out_w, *out_b;
    float *cond0_w, *cond0_b, *cond2_w, *cond2_b;
} adcdiff_unet_grads;

/* Workspace for one training forward + backward at this batch (activations stay in it between the two calls). */
size_t adcdiff_unet_train_workspace_bytes(int input_dim, int hidden_dim, int output_dim, int num_timesteps, int dtype, int batch);

/* UNetModel.forward in training mode (diffusion_model.py:749-789 under model.train(), :413) with per-row timesteps:
 * dropout(p) after the norm2 activation of every ResidualBlock (:898) from the counter RNG (dropout_seed, layer, row,
 * column); leaves the activations the backward pass needs in the workspace.  Requires adcdiff_unet_pack after the last
 * parameter change (the timestep tables are part of the differentiated graph).
 * cond: NULL, or the condition rows [B, input_dim] float32 (leading dimension ldc): condition_embed (:694-698) runs inside the
 * call and its output joins the time embedding of every row (:764-766); adcdiff_unet_backward then also produces the
 * condition_embed gradients. */
int adcdiff_unet_train_forward(adcdiff_unet* net, const float* x, int ldx, int B, const int64_t* t_rows, const float* cond, int ldc,
                               float dropout_p, uint64_t dropout_seed, float* eps, int lde, void* workspace, size_t workspace_bytes,
                               adcdiff_stream_t stream);

/* F.mse_loss(target, pred) with mean reduction (diffusion_model.py:183): *loss_accum += scale * mean((target - pred)^2),
 * dpred = d(scale * loss) / d(pred).  scale = 1 / gradient_accumulation_steps (:428, :451). */
int adcdiff_mse_loss_grad(const float* pred, int ldp, const float* target, int ldt, float* dpred, int ldd, int B, int D, float scale,
                          float* loss_accum, adcdiff_stream_t stream);

/* loss.backward() through the denoiser (diffusion_model.py:431, :453): d_eps [B, output_dim] -> parameter gradients. */
int adcdiff_unet_backward(adcdiff_unet* net, const float* d_eps, int lde, const adcdiff_unet_grads* grads, void* workspace,
                          size_t workspace_bytes, adcdiff_stream_t stream);

/* ------------------------------------------------------------------ latent summaries ------ */

/* What the reference's latent -> SMILES heuristic reads of a latent row (molecule_generator.py:1080-1082 np.sum /
 * np.mean of the float32 row, :1114-1126 features[5], [8], [3], :1158-1161 the fallback's np.sum):
 *   out[row] = { mean, sum, x[3], x[5], x[8] }    (5 floats, ld_out >= 5)
 * so that 20 bytes per sample leave the GPU (or cross xGMI) instead of the 4 * D byte row.  The sum reproduces
 * numpy's float32 pairwise summation bit for bit (blocks of 128 elements, eight strided partial sums per block combined
 * as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), blocks combined as a balanced binary tree), which is defined here for
 * D = 128 * 2^k, 128 <= D <= 4096 (the tree is then perfect); other D are rejected.  mean = sum / D. */
int adcdiff_latent_summary(const float* x, int ldx, int B, int D, float* out, int ld_out, adcdiff_stream_t stream);

/* ------------------------------------------------------------------ optimizer step ------ */

/* torch.nn.utils.clip_grad_norm_(parameters, max_norm) followed by optimizer.step() of AdamW / Adam
 * (diffusion_model.py:69-74 construction, :434-436 and :456-459 use) over all parameter tensors in three launches:
 * per-chunk sums of squares of the gradients, a fixed-order reduction to the global L2 norm and
 * clip = min(1, max_norm / (norm + 1e-6)), and the update
 *   g = clip * grad (+ weight_decay * p when decoupled == 0: Adam's L2 form)
 *   p *= 1 - lr * weight_decay (decoupled != 0: AdamW);  m += (1 - beta1) * (g - m);  v = beta2 * v + (1 - beta2) * g * g
 *   p -= (lr / bias_correction1) * m / (sqrt(v) / sqrt(bias_correction2) + eps),  bias_correction_i = 1 - beta_i^step
 * in torch's operation order.  No host synchronisation; deterministic (fixed chunking, fixed summation order).
 * `tensors` is a DEVICE array of n_tensors descriptors (the caller uploads it once and again when a pointer changes);
 * chunk_start = number of ADCDIFF_OPT_CHUNK-element chunks in all earlier tensors.  grad_norm (device, optional)
 * receives the norm before clipping, as clip_grad_norm_ returns it. */
#define ADCDIFF_OPT_CHUNK 4096
typedef struct adcdiff_opt_tensor {
    float* param;
    const float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    long long numel;
    long long chunk_start;
} adcdiff_opt_tensor;
typedef struct adcdiff_adamw_args {
    double lr, beta1, beta2, eps, weight_decay;   /* the optimizer's python floats: 1 - beta2 must not pass through float32 */
    double max_norm;       /* <= 0: no clipping */
    int step;              /* 1-based count of this update */
    int decoupled;         /* 1 = AdamW, 0 = Adam with L2 weight decay */
} adcdiff_adamw_args;
size_t adcdiff_clip_adamw_workspace_bytes(long long total_chunks);
int adcdiff_clip_adamw_step(const adcdiff_opt_tensor* tensors, int n_tensors, long long total_chunks, const adcdiff_adamw_args* args,
                            float* grad_norm, void* workspace, size_t workspace_bytes, adcdiff_stream_t stream);

/* adcdiff_clip_adamw_step for callers that cannot allocate (code under HIP graph capture, serving loops, harnesses that forbid
 * hipMalloc); it has no counterpart in the reference.  Same arguments and results, but no workspace: the per-chunk partial sums
 * and the (norm, clip) scalars live in a fixed array inside the library's code object, 64 + ADCDIFF_OPT_NOALLOC_MAX_CHUNKS
 * floats (256.25 KiB) that every GPU the library runs on holds until the process ends.  Limit: total_chunks <=
 * ADCDIFF_OPT_NOALLOC_MAX_CHUNKS (268 M parameters; the 44.3 M-parameter denoiser has 10 880); more is rejected with
 * ADCDIFF_E_WORKSPACE.  Calls no hipMalloc / hipFree and no allocating library routine; the first call on a device looks
 * the array's address up (hipGetSymbolAddress).  All calls on one GPU share that array: calls that overlap on one GPU (two
 * streams, two threads) race; calls on one stream are ordered by the stream.  Nothing in it needs clearing between calls. */
#define ADCDIFF_OPT_NOALLOC_MAX_CHUNKS 65536
int adcdiff_clip_adamw_step_noalloc(const adcdiff_opt_tensor* tensors, int n_tensors, long long total_chunks,
                                    const adcdiff_adamw_args* args, float* grad_norm, adcdiff_stream_t stream);

/* ------------------------------------------------------------------ measurement ------ */

/* Measurement and test switches; no counterpart in the reference.  With no call of adcdiff_debug_set the library reads no
 * environment variable and its kernel choice is a function of the call's arguments alone.  Keys (default):
 *   tile_hints (0)       1 = honour adcdiff_gemm_args.tile_hint: 128 / 256 force that tile kernel where the shape allows (256 =
 *                        the eight-wave 256 x 256 kernel, 2560 / 2561 its phased / ring main loop, 2564 the four-wave 256 x 256
 *                        kernel with the assembly main loop, 2565 / 2566 / 2568 the same with epilogue batches of 1 / 2 / 8
 *                        rows, 2567 the same without the epilogue variants specialised on the launch's side inputs); 1 / 3 the 128-row kernel with one LDS stage / the three-stage ring; 32 the one-wave kernel,
 *                        325 its variant with loader waves, 327 the split-K variant (needs splitk_scratch)
 *   gemm_tile (0)        128 / 256 / 2561: as the hint of that value, for every bf16 launch
 *   gemm_ring (0), gemm_ring2 (1), gemm_ring_max (512), gemm_w4 (1), gemm_w4_min_n (512), gemm_big_min_k (256), gemm_w1 (1), gemm_w1_rows (1024),
 *   gemm_cl (1)          kernel-family switches of adcdiff_gemm's dispatch (gemm_w4_min_n: the narrowest layer the four-wave kernel takes)
 *   gemm_pf_w (1)        0 = the four-wave kernel prefetches only its A rows into L2, not the W rows
 *   gemm_pf_next (1)     0 = the four-wave kernel does not touch, near the end of its epilogue, the operand lines that the first two
 *                        K tiles of its successor workgroup on the same XCD will read
 *   gemm_splitk (1)      0 = never the few-row kernel that splits K over workgroups (splitk_scratch is ignored): small batches are
 *                        then bit-identical to large ones
 *   step_fold (1)        0 = the reverse step and encoder.0's GroupNorm as launches of their own in the sampling loops
 *   train_transpose (0)  1 = backward products on transposed copies instead of adcdiff_gemm_tr
 *   train_group_dw (1)   0 = every weight-gradient product of adcdiff_unet_backward as a launch of its own (split over the batch,
 *                        partial sums) instead of one grouped launch at the end of the pass
 *   train_fuse_cast (1)  0 = the bf16 operand copies and bias gradients of adcdiff_unet_backward by cast launches of their own instead
 *                        of the GroupNorm backward kernel that finishes the tensor
 *   train_fused_fwd (1)  0 = adcdiff_unet_train_forward with every GroupNorm as a launch of its own instead of in the epilogue of the
 *                        GEMM that produces its input (only norm2 with dropout and the latent's norm1 are stand-alone launches)
 *   sync (0)             1 = synchronise after every launch of adcdiff_unet_backward and name the first that fails
 * Values are process-wide and read by every later call; unknown keys are rejected. */
int adcdiff_debug_set(const char* key, long long value);
int adcdiff_debug_get(const char* key, long long* value);

/* Per-launch timing for benchmarks; no counterpart in the reference.  Between adcdiff_timing_begin and
 * adcdiff_timing_end every GEMM, GroupNorm+SiLU and schedule-update launch made through this library has a
 * hipEvent recorded in front of it on the stream it is launched on (up to max_records launches; later ones
 * are not recorded); adcdiff_timing_end records one more event behind the last launch, synchronises it and
 * returns one record per launch whose ms is the span from its event to the next launch's event.  The records
 * of a window therefore add up to the window's device time; the dispatch gap behind a kernel (1-2 us) is charged
 * to it.  All launches of a window must go to ONE stream.  Not thread safe; not for use under graph capture. */
enum { ADCDIFF_KIND_GEMM_F32 = 0, ADCDIFF_KIND_GEMM_BF16 = 1, ADCDIFF_KIND_GROUPNORM = 2, ADCDIFF_KIND_UPDATE = 3,
       ADCDIFF_KIND_GEMM_BF16_T256 = 4 /* bf16 GEMM that ran on the 256 x 256-tile kernel */,
       ADCDIFF_KIND_OTHER = 5 /* every other kernel launch of the library (casts, transposes, backward element-wise kernels, optimizer) */,
       ADCDIFF_KIND_GEMM_BF16_W256 = 6 /* bf16 GEMM that ran on the four-wave 256 x 256-tile kernel (assembly main loop) */ };
typedef struct adcdiff_timing_record {
    int kind;
    int M, N, K;        /* GEMM: rows, outputs, total K; others: rows, columns, 0 */
    float ms;           /* device time between the two events */
    double flops;       /* 2*M*N*K for a GEMM, 0 otherwise */
    double bytes;       /* algorithmic HBM bytes of the launch (operands read once + outputs written once) */
} adcdiff_timing_record;
int adcdiff_timing_begin(int max_records);
int adcdiff_timing_end(adcdiff_timing_record* out, int capacity, int* count);

/* What the matrix cores of the current GPU sustain inside its power limit; no counterpart in the reference.  Runs `launches`
 * launches (after launches / 4 + 1 untimed ones) of a kernel that issues iters x 64 v_mfma_f32_16x16x32_bf16 per wave on register
 * operands with random bits (no memory traffic), one wave per SIMD on every CU, on the null stream, and returns the rate in
 * TFLOP/s and the mean shader clock in MHz.  The data sheet's dense bf16 rate assumes the top clock; under this load the clock
 * settles lower, and a GEMM -- which also moves its operands -- cannot exceed the rate measured here.  Allocates and frees
 * two small device buffers; synchronises the device. */
int adcdiff_matrix_core_probe(int iters, int launches, double* tflops, double* clock_mhz);

#ifdef __cplusplus
}
#endif
#endif /* ADCDIFF_H */
