// Host-side plan structures of the denoiser shared by unet.hip (inference) and train.hip (training step).
#pragma once
#include <cstring>
#include <new>

#include "common.h"

namespace adcdiff {

int gemm_dispatch(const adcdiff_gemm_args& a, hipStream_t stream);
// gemm_tr.hip: bf16 product on contraction-major operands (see adcdiff_gemm_tr)
int gemm_tr(bool ta, const void* A, long lda, const void* B, long ldb, float* C, long ldc, int M, int N, int K, int slices, long c_slice,
            hipStream_t stream);
// gemm_tr.hip: queued transposed-A products C = A^T B run as one launch (the weight gradients of a backward pass)
struct TrProblem {
    const void* A; long lda;    // [K, M] row-major bf16
    const void* B; long ldb;    // [K, N] row-major bf16
    float* C; long ldc;         // [M, N] float32
    int M, N, K;
};
constexpr int TR_MAX_GROUP = 64;
int gemm_tr_grouped(const TrProblem* probs, int nprob, hipStream_t stream);
int groupnorm_silu(const float* x, int ldx, int M, int C, int groups, const float* gamma, const float* beta, float eps,
                   int out_dtype, void* out, int ldo, void* raw_out, int ldraw, hipStream_t stream, float drop_p = 0.0f,
                   uint64_t drop_seed = 0, uint32_t drop_layer = 0);
int sinusoid_table(float* out, int ld, int rows, int half, const float* freq, hipStream_t stream);
int temb_silu(const float* te_table, int ldt, int table_rows, const int64_t* t_rows, int t_uniform, const float* cemb, int ldce, int dtype,
              void* out, int ldo, int rows, int H, hipStream_t stream);
int cast_pack(const float* src, int lds_, int dtype, void* dst, int ldd, int col0, int rows, int cols, hipStream_t stream);
int add_vec(const float* a, const float* b, float* out, int n, hipStream_t stream);

// One entry of the batched weight packing: dst[r * ldd + col0 + c] = (dtype) (src[r * cols + c] + (add ? add[r * cols + c] : 0)),
// r * cols + c < numel; cols, ldd, col0 multiples of 4.  chunk_start = PACK_CHUNK-element chunks of all earlier entries.
struct PackItem {
    const float* src;
    const float* add;
    void* dst;
    int cols, ldd, col0, dtype;
    long long numel, chunk_start;
};
constexpr int PACK_CHUNK = 2048;
constexpr int PACK_MAX_ITEMS = 128;
int pack_items(const PackItem* items_dev, int n_items, long long total_chunks, hipStream_t stream);

static inline size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static inline int num_groups(int dim) {   // diffusion_model.py:723-730, 840-847
    if (dim < 4) return 1;
    for (int g = dim < 32 ? dim : 32; g > 0; --g)
        if (dim % g == 0) return g;
    return 1;
}

struct Bump {
    char* base;
    size_t off = 0, cap, peak = 0;   // peak: high-water mark (callers may rewind `off` to reuse temporaries)
    Bump(void* b, size_t c) : base(reinterpret_cast<char*>(b)), cap(c) {}
    void* take(size_t bytes) {
        void* p = base + off;   // in a dry run base is null and only the offsets matter
        off += round_up(bytes, 256);
        if (off > peak) peak = off;
        return p;
    }
};

struct BlockPlan {
    int cin, cout, nparts, part_c[2];
    bool has_res;
    int g1, g2, tb_off, ldw1, ldw2;
    void *w1cat, *w2cat;
    float *bias1, *bias1_cond, *bias2;
    int gs1, gs2;   // channels per GroupNorm group of norm1 (over the whole concatenated input) and norm2
};

// What a training forward leaves in the workspace for the backward pass (pointers into the caller's workspace).
struct BlockTape {
    const float* in_f32[2];   // block input part(s), float32 [rows, C] (leading dimension in_ld)
    int in_ld[2], in_c[2], nparts;
    const void* in_raw[2];    // the same in the operand type, padded rows (null when the block has an Identity residual and float32 operands)
    const void* a1;           // SiLU(GroupNorm1(input)) in the operand type, [Bp, cin]
    const float* h1;          // linear1 output + time term, float32 [Bp, cout]
    const void* a2;           // dropout(SiLU(GroupNorm2(h1))) in the operand type, [Bp, cout]
    const float* o;           // block output, float32 [Bp, cout]
};

struct Tape {
    bool valid = false;
    int B = 0;
    const int64_t* t_rows = nullptr;
    float drop_p = 0.0f;
    uint64_t drop_seed = 0;
    BlockTape blk[10];
    const void* a_out = nullptr;   // SiLU(output_norm(decoder.3 output)) in the operand type
    size_t ws_used = 0;
    // a conditional training forward (per-row condition): the condition in the operand type [Bp, D], condition_embed's hidden
    // pre-activation (float32) and activation (operand type) [Bp, H], its output c_emb (float32 [Bp, H]) and
    // s = SiLU(t_emb[t] + c_emb) in the operand type [Bp, H], which every block's time_proj consumes
    bool cond = false;
    const void* c_op = nullptr;
    const float* c_pre = nullptr;
    const void* c_hid = nullptr;
    const float* cemb = nullptr;
    const void* s_op = nullptr;
};

}  // namespace adcdiff

struct adcdiff_unet {
    adcdiff_unet_params prm;
    int dtype, esz, D, H, O, T, Tp, sumN;
    adcdiff::BlockPlan blk[10];
    float *sin_table, *te_pre, *te_hidden, *te_table, *s_table, *wtp_all, *btp_all, *tb_table;
    float *s_cond, *tb_cond;   // the sampling loops' table for a one-row condition: SiLU(t_emb + c_emb) and every block's time_proj of it, per timestep
    void *w_out, *w_c0, *w_c2;
    void* sk_scratch;          // split-K scratch of the few-row GEMMs: 4096 bytes of arrival counters, then partial sums
    size_t sk_bytes;
    size_t arena_used;
    bool packed;
    adcdiff::PackItem* pack_table;      // device copy of the packing entries (in the arena), n_pack entries, pack_chunks chunks
    int n_pack;
    long long pack_chunks;
    adcdiff::PackItem pack_host[adcdiff::PACK_MAX_ITEMS];   // the entries on the host: adcdiff_unet_pack uploads them on its stream (the first time)
    bool pack_uploaded;
    bool sin_built;            // the sinusoid table depends on constants only: built by the first pack
    bool fused;     // every GroupNorm can be applied in a producing GEMM's epilogue
    bool fold;      // fused, and the output GEMM can apply the sampler's update and encoder.0's norm1 of the next step
    adcdiff::Tape tape;
};

namespace adcdiff {
int plan_dims(adcdiff_unet* n, int input_dim, int hidden_dim, int output_dim, int num_timesteps, int dtype);
int forward_unfused(adcdiff_unet* n, const float* x, int ldx, int B, const int64_t* t_rows, int t_uniform, const float* cemb,
                    int ldce, float* eps, int lde, Bump& ws, bool dry, hipStream_t st, Tape* tape = nullptr, float drop_p = 0.0f,
                    uint64_t drop_seed = 0);
int forward_taped(adcdiff_unet* n, const float* x, int ldx, int B, const int64_t* t_rows, const float* cemb, int ldce, float* eps,
                  int lde, Bump& ws, bool dry, hipStream_t st, Tape* tape, float drop_p, uint64_t drop_seed);
adcdiff_gemm_args gemm_args(int dtype, int M, int m_valid, int N);
}  // namespace adcdiff

#define TRY(expr) do { int rc_ = (expr); if (rc_ != ADCDIFF_OK) return rc_; } while (0)
