#include "unet_plan.h"

using namespace adcdiff;

int adcdiff::plan_dims(adcdiff_unet* n, int input_dim, int hidden_dim, int output_dim, int num_timesteps, int dtype) {
    if (dtype != ADCDIFF_F32 && dtype != ADCDIFF_BF16) return fail(ADCDIFF_E_BADARG, "unet: unknown dtype %d", dtype);
    if (input_dim <= 0 || hidden_dim <= 0 || output_dim <= 0 || input_dim % 64 || hidden_dim % 64 || output_dim % 64)
        return fail(ADCDIFF_E_BADARG, "unet: input_dim=%d hidden_dim=%d output_dim=%d must be positive multiples of 64", input_dim,
                    hidden_dim, output_dim);
    if (num_timesteps <= 0) return fail(ADCDIFF_E_BADARG, "unet: num_timesteps=%d", num_timesteps);
    n->dtype = dtype; n->esz = dtype == ADCDIFF_BF16 ? 2 : 4;
    n->D = input_dim; n->H = hidden_dim; n->O = output_dim; n->T = num_timesteps;
    n->Tp = (int)round_up((size_t)num_timesteps, 128);
    const int h = hidden_dim;
    const int cin[10] = {input_dim, h, 2 * h, 4 * h, 8 * h, 8 * h, 16 * h, 8 * h, 4 * h, 2 * h};
    const int cout[10] = {h, 2 * h, 4 * h, 8 * h, 8 * h, 8 * h, 4 * h, 2 * h, h, h};
    int off = 0;
    for (int i = 0; i < 10; ++i) {
        BlockPlan& b = n->blk[i];
        b.cin = cin[i]; b.cout = cout[i]; b.has_res = cin[i] != cout[i];
        b.nparts = i >= 6 ? 2 : 1;
        b.part_c[0] = i >= 6 ? cin[i] / 2 : cin[i];
        b.part_c[1] = i >= 6 ? cin[i] / 2 : 0;
        b.g1 = num_groups(b.cin); b.g2 = num_groups(b.cout);
        const int gs = b.cin / b.g1;
        if (b.nparts == 2 && b.part_c[0] % gs != 0)
            return fail(ADCDIFF_E_BADARG, "unet: GroupNorm groups of %d channels straddle the skip concatenation at width %d", gs, b.cin);
        b.tb_off = off; off += b.cout;
        b.ldw1 = b.cin + h;
        b.ldw2 = b.cout + (b.has_res ? b.cin : 0);
        b.gs1 = gs; b.gs2 = b.cout / b.g2;
    }
    n->sumN = off;
    auto fusable = [](int gs, int width) { return gs >= 4 && gs <= (width % 128 == 0 ? 128 : 64) && (gs & (gs - 1)) == 0; };
    n->fused = fusable(hidden_dim / num_groups(hidden_dim), hidden_dim);   // output_norm, applied by decoder.3's GEMM
    for (int i = 0; i < 10; ++i) {
        const BlockPlan& b = n->blk[i];
        n->fused = n->fused && fusable(b.gs2, b.cout);
        if (i > 0) n->fused = n->fused && fusable(b.gs1, n->blk[i - 1].cout);           // applied by the previous block
        if (i >= 6) n->fused = n->fused && fusable(b.gs1, n->blk[9 - i].cout);          // and by the skip's encoder block
    }
    n->fold = n->fused && input_dim == output_dim && fusable(n->blk[0].gs1, output_dim);
    return ADCDIFF_OK;
}

// Carve the arena; with a null base this is the size computation.
static void carve_arena(adcdiff_unet* n, void* arena, size_t cap) {
    Bump a(arena, cap);
    const size_t e = (size_t)n->esz;
    for (int i = 0; i < 10; ++i) {
        BlockPlan& b = n->blk[i];
        b.w1cat = a.take((size_t)b.cout * b.ldw1 * e);
        b.w2cat = a.take((size_t)b.cout * b.ldw2 * e);
        b.bias1 = (float*)a.take((size_t)b.cout * 4);
        b.bias1_cond = (float*)a.take((size_t)b.cout * 4);
        b.bias2 = (float*)a.take((size_t)b.cout * 4);
    }
    n->w_out = a.take((size_t)n->O * n->H * e);
    n->w_c0 = a.take((size_t)n->H * n->D * e);
    n->w_c2 = a.take((size_t)n->H * n->H * e);
    n->sin_table = (float*)a.take((size_t)n->Tp * n->H * 4);
    n->te_pre = (float*)a.take((size_t)n->Tp * 4 * n->H * 4);
    n->te_hidden = (float*)a.take((size_t)n->Tp * 4 * n->H * 4);
    n->te_table = (float*)a.take((size_t)n->Tp * n->H * 4);
    n->s_table = (float*)a.take((size_t)n->Tp * n->H * 4);
    n->wtp_all = (float*)a.take((size_t)n->sumN * n->H * 4);
    n->btp_all = (float*)a.take((size_t)n->sumN * 4);
    n->tb_table = (float*)a.take((size_t)n->Tp * n->sumN * 4);
    n->s_cond = (float*)a.take((size_t)n->Tp * n->H * 4);
    n->tb_cond = (float*)a.take((size_t)n->Tp * n->sumN * 4);
    n->pack_table = (PackItem*)a.take(PACK_MAX_ITEMS * sizeof(PackItem));
    // few-row GEMMs split K over workgroups through this buffer (adcdiff_gemm_args.splitk_scratch): arrival counters, zeroed by
    // adcdiff_unet_pack and left zero by every launch, and the partial sums of 16 splits x 64 rows x the widest layer
    int widest = n->O;
    for (int i = 0; i < 10; ++i) widest = n->blk[i].cout > widest ? n->blk[i].cout : widest;
    n->sk_bytes = 4096 + (size_t)16 * 64 * widest * 4;
    n->sk_scratch = a.take(n->sk_bytes);
    n->arena_used = a.off;
}

extern "C" size_t adcdiff_unet_arena_bytes(int input_dim, int hidden_dim, int output_dim, int num_timesteps, int dtype) {
    adcdiff_unet n{};
    if (plan_dims(&n, input_dim, hidden_dim, output_dim, num_timesteps, dtype) != ADCDIFF_OK) return 0;
    carve_arena(&n, nullptr, 0);
    return n.arena_used;
}

extern "C" int adcdiff_unet_create(const adcdiff_unet_params* p, int dtype, void* arena, size_t arena_bytes, adcdiff_unet** out) {
    if (!p || !arena || !out) return fail(ADCDIFF_E_BADARG, "unet_create: null argument");
    adcdiff_unet* n = new (std::nothrow) adcdiff_unet{};
    if (!n) return fail(ADCDIFF_E_BADARG, "unet_create: out of host memory");
    int rc = plan_dims(n, p->input_dim, p->hidden_dim, p->output_dim, p->num_timesteps, dtype);
    if (rc != ADCDIFF_OK) { delete n; return rc; }
    if (p->input_dim != p->output_dim) {
        // the reverse loop feeds the [B, output_dim] latent back as the [B, input_dim] input (diffusion_model.py:299, 702)
        delete n;
        return fail(ADCDIFF_E_BADARG, "unet_create: input_dim=%d != output_dim=%d cannot be sampled", p->input_dim, p->output_dim);
    }
    n->prm = *p;
    if (!aligned16(arena)) { delete n; return fail(ADCDIFF_E_BADARG, "unet_create: arena must be 16-byte aligned"); }
    carve_arena(n, arena, arena_bytes);
    if (n->arena_used > arena_bytes) {
        const size_t need = n->arena_used;
        delete n;
        return fail(ADCDIFF_E_WORKSPACE, "unet_create: arena of %zu bytes, %zu needed", arena_bytes, need);
    }
    const float* req[] = {p->time_lin1_w, p->time_lin1_b, p->time_lin2_w, p->time_lin2_b, p->cond0_w, p->cond0_b, p->cond2_w,
                          p->cond2_b, p->out_norm_w, p->out_norm_b, p->out_w, p->out_b, p->sinusoid_freq};
    for (const float* q : req)
        if (!q || !aligned16(q)) { delete n; return fail(ADCDIFF_E_BADARG, "unet_create: null or misaligned parameter pointer"); }
    for (int i = 0; i < 10; ++i) {
        const adcdiff_block_params& b = p->blocks[i];
        const float* rb[] = {b.norm1_w, b.norm1_b, b.norm2_w, b.norm2_b, b.linear1_w, b.linear1_b, b.linear2_w, b.linear2_b,
                             b.time_proj_w, b.time_proj_b};
        for (const float* q : rb)
            if (!q || !aligned16(q)) { delete n; return fail(ADCDIFF_E_BADARG, "unet_create: block %d: null or misaligned parameter", i); }
        if (n->blk[i].has_res != (b.residual_w != nullptr) || (b.residual_w && !b.residual_b)) {
            delete n;
            return fail(ADCDIFF_E_BADARG, "unet_create: block %d: residual_proj presence does not match its dimensions", i);
        }
    }
    // the packing entries: every weight cast into the concatenated operand layouts, every bias sum, the stacked time_proj copies
    {
        PackItem* const items = n->pack_host;
        int k = 0;
        long long chunks = 0;
        auto add_item = [&](const float* src, const float* add, void* dst, int rows, int cols, int ldd, int col0, int dt) {
            PackItem& it = items[k++];
            it.src = src; it.add = add; it.dst = dst; it.cols = cols; it.ldd = ldd; it.col0 = col0; it.dtype = dt;
            it.numel = (long long)rows * cols; it.chunk_start = chunks;
            chunks += (it.numel + PACK_CHUNK - 1) / PACK_CHUNK;
        };
        const int H = n->H;
        for (int i = 0; i < 10; ++i) {
            const BlockPlan& b = n->blk[i];
            const adcdiff_block_params& q = p->blocks[i];
            add_item(q.linear1_w, nullptr, b.w1cat, b.cout, b.cin, b.ldw1, 0, dtype);
            add_item(q.time_proj_w, nullptr, b.w1cat, b.cout, H, b.ldw1, b.cin, dtype);
            add_item(q.linear2_w, nullptr, b.w2cat, b.cout, b.cout, b.ldw2, 0, dtype);
            if (b.has_res) add_item(q.residual_w, nullptr, b.w2cat, b.cout, b.cin, b.ldw2, b.cout, dtype);
            add_item(q.linear1_b, nullptr, b.bias1, 1, b.cout, b.cout, 0, ADCDIFF_F32);
            add_item(q.linear1_b, q.time_proj_b, b.bias1_cond, 1, b.cout, b.cout, 0, ADCDIFF_F32);
            add_item(q.linear2_b, b.has_res ? q.residual_b : nullptr, b.bias2, 1, b.cout, b.cout, 0, ADCDIFF_F32);
            add_item(q.time_proj_w, nullptr, n->wtp_all + (size_t)b.tb_off * H, b.cout, H, H, 0, ADCDIFF_F32);
            add_item(q.time_proj_b, nullptr, n->btp_all + b.tb_off, 1, b.cout, b.cout, 0, ADCDIFF_F32);
        }
        add_item(p->out_w, nullptr, n->w_out, n->O, H, H, 0, dtype);
        add_item(p->cond0_w, nullptr, n->w_c0, H, n->D, n->D, 0, dtype);
        add_item(p->cond2_w, nullptr, n->w_c2, H, H, H, 0, dtype);
        n->n_pack = k; n->pack_chunks = chunks;
        n->pack_uploaded = false;   // create copies and launches nothing: the first adcdiff_unet_pack uploads the entries on its stream
    }
    n->packed = false;
    n->sin_built = false;
    *out = n;
    return ADCDIFF_OK;
}

extern "C" void adcdiff_unet_destroy(adcdiff_unet* n) { delete n; }

adcdiff_gemm_args adcdiff::gemm_args(int dtype, int M, int m_valid, int N) {
    adcdiff_gemm_args g;
    memset(&g, 0, sizeof(g));
    g.dtype = dtype; g.M = M; g.m_valid = m_valid; g.N = N;
    return g;
}

extern "C" int adcdiff_unet_pack(adcdiff_unet* n, void* workspace, size_t workspace_bytes, adcdiff_stream_t stream_) {
    (void)workspace; (void)workspace_bytes;
    if (!n) return fail(ADCDIFF_E_BADARG, "unet_pack: null plan");
    hipStream_t st = as_stream(stream_);
    const adcdiff_unet_params& p = n->prm;
    const int H = n->H;
    if (!n->pack_uploaded) {
        // stream-ordered copy out of the plan's own host memory (pageable: the runtime stages it before returning)
        if (hipMemcpyAsync(n->pack_table, n->pack_host, (size_t)n->n_pack * sizeof(PackItem), hipMemcpyHostToDevice, st) != hipSuccess)
            return fail(ADCDIFF_E_LAUNCH, "unet_pack: copying the packing table into the arena failed (is the arena device memory?)");
        n->pack_uploaded = true;
    }
    TRY(pack_items(n->pack_table, n->n_pack, n->pack_chunks, st));
    if (hipMemsetAsync(n->sk_scratch, 0, 4096, st) != hipSuccess) return fail(ADCDIFF_E_LAUNCH, "unet_pack: clearing the split-K counters failed");

    // TimeEmbedding(t) for every timestep (diffusion_model.py:802-823), float32 throughout
    if (!n->sin_built) TRY(sinusoid_table(n->sin_table, H, n->Tp, H / 2, p.sinusoid_freq, st));
    n->sin_built = true;
    adcdiff_gemm_args g = gemm_args(ADCDIFF_F32, n->Tp, 0, 4 * H);
    g.nseg = 1; g.A[0] = n->sin_table; g.lda[0] = H; g.K[0] = H; g.W = p.time_lin1_w; g.ldw = H; g.bias = p.time_lin1_b;
    g.out_f32 = n->te_pre; g.ldo_f32 = 4 * H;   // pre-activation kept for the training step's backward
    TRY(gemm_dispatch(g, st));
    TRY(temb_silu(n->te_pre, 4 * H, n->Tp, nullptr, -1, nullptr, 0, ADCDIFF_F32, n->te_hidden, 4 * H, n->Tp, 4 * H, st));
    g = gemm_args(ADCDIFF_F32, n->Tp, 0, H);
    g.nseg = 1; g.A[0] = n->te_hidden; g.lda[0] = 4 * H; g.K[0] = 4 * H; g.W = p.time_lin2_w; g.ldw = 4 * H; g.bias = p.time_lin2_b;
    g.out_f32 = n->te_table; g.ldo_f32 = H;
    TRY(gemm_dispatch(g, st));
    // every block's time_proj(SiLU(t_emb)) for every timestep (diffusion_model.py:859-862, 893)
    TRY(temb_silu(n->te_table, H, n->Tp, nullptr, -1, nullptr, 0, ADCDIFF_F32, n->s_table, H, n->Tp, H, st));
    g = gemm_args(ADCDIFF_F32, n->Tp, 0, n->sumN);
    g.nseg = 1; g.A[0] = n->s_table; g.lda[0] = H; g.K[0] = H; g.W = n->wtp_all; g.ldw = H; g.bias = n->btp_all;
    g.out_f32 = n->tb_table; g.ldo_f32 = n->sumN;
    TRY(gemm_dispatch(g, st));
    n->packed = true;
    return ADCDIFF_OK;
}

namespace {

struct Part {
    const float* f32;   // [rows, C] float32 activations (leading dimension ld)
    int C, ld;
    const void* raw;    // the same values in the GEMM operand type with padded rows, or null if not made yet
};

// One denoiser forward.  With dry = true nothing is launched and only the bump offset advances.
}  // namespace

int adcdiff::forward_unfused(adcdiff_unet* n, const float* x, int ldx, int B, const int64_t* t_rows, int t_uniform, const float* cemb,
                             int ldce, float* eps, int lde, Bump& ws, bool dry, hipStream_t st, Tape* tape, float drop_p,
                             uint64_t drop_seed) {
    const int Bp = (int)round_up((size_t)B, 128);
    const size_t e = (size_t)n->esz;
    const int H = n->H, dt = n->dtype;
    const adcdiff_unet_params& p = n->prm;
    const bool cond = cemb != nullptr;

    const void* s_op = nullptr;
    if (cond || dry) {   // the dry run sizes for the conditional case, the larger one
        void* s = ws.take((size_t)Bp * H * e);
        if (!dry) TRY(temb_silu(n->te_table, H, n->T, t_rows, t_uniform, cemb, ldce, dt, s, H, B, H, st));
        s_op = s;
    }

    Part outs[10];
    Part cur[2];
    cur[0] = Part{x, n->D, ldx, nullptr};
    int ncur = 1;
    for (int i = 0; i < 10; ++i) {
        const BlockPlan& b = n->blk[i];
        const adcdiff_block_params& q = p.blocks[i];
        if (i >= 6) {   // decoder input = cat([h, skip]) with skip = encoder output 3, 2, 1, 0 (diffusion_model.py:781-782)
            cur[1] = outs[9 - i];
            ncur = 2;
        }
        BlockTape* bt = tape ? &tape->blk[i] : nullptr;
        // a1 = SiLU(GroupNorm1(input)), written as one [Bp, cin] operand
        char* a1 = (char*)ws.take((size_t)Bp * b.cin * e);
        const int gs = b.cin / b.g1;
        int col = 0;
        for (int k = 0; k < ncur; ++k) {
            void* raw = nullptr;
            if (!cur[k].raw) raw = ws.take((size_t)Bp * cur[k].C * e);   // residual_proj operand or Identity residual
            if (!dry)
                TRY(groupnorm_silu(cur[k].f32, cur[k].ld, B, cur[k].C, cur[k].C / gs, q.norm1_w + col, q.norm1_b + col, 1e-5f, dt,
                                   a1 + (size_t)col * e, b.cin, raw, cur[k].C, st));
            if (raw) cur[k].raw = raw;
            if (bt) { bt->in_f32[k] = cur[k].f32; bt->in_ld[k] = cur[k].ld; bt->in_c[k] = cur[k].C; bt->in_raw[k] = cur[k].raw; }
            col += cur[k].C;
        }
        if (bt) { bt->nparts = ncur; bt->a1 = a1; }
        // h1 = linear1(a1) + time_proj(SiLU(t_emb))
        float* h1 = (float*)ws.take((size_t)Bp * b.cout * 4);
        if (!dry) {
            adcdiff_gemm_args g = gemm_args(dt, Bp, B, b.cout);
            g.nseg = 1; g.A[0] = a1; g.lda[0] = b.cin; g.K[0] = b.cin;
            g.W = b.w1cat; g.ldw = b.ldw1;
            if (cond) {
                g.nseg = 2; g.A[1] = s_op; g.lda[1] = H; g.K[1] = H;
                g.bias = b.bias1_cond;
            } else {
                g.bias = b.bias1;
                if (t_rows) { g.rowbias = n->tb_table + b.tb_off; g.ldrb = n->sumN; g.rowbias_idx = t_rows; g.rowbias_rows = n->T; }
                else { g.rowbias = n->tb_table + (size_t)t_uniform * n->sumN + b.tb_off; g.ldrb = 0; }
            }
            g.out_f32 = h1; g.ldo_f32 = b.cout;
            TRY(gemm_dispatch(g, st));
        }
        // a2 = SiLU(GroupNorm2(h1)); dropout is the identity in eval mode (diffusion_model.py:896-898)
        void* a2 = ws.take((size_t)Bp * b.cout * e);
        // in a training forward the dropout of diffusion_model.py:898 is applied here (counter RNG keyed by layer and row)
        if (!dry) TRY(groupnorm_silu(h1, b.cout, B, b.cout, b.g2, q.norm2_w, q.norm2_b, 1e-5f, dt, a2, b.cout, nullptr, 0, st, drop_p, drop_seed, (uint32_t)i));
        if (bt) { bt->h1 = h1; bt->a2 = a2; }
        // o = linear2(a2) + residual_proj(input)
        float* o = (float*)ws.take((size_t)Bp * b.cout * 4);
        if (!dry) {
            adcdiff_gemm_args g = gemm_args(dt, Bp, B, b.cout);
            g.nseg = 1; g.A[0] = a2; g.lda[0] = b.cout; g.K[0] = b.cout;
            g.W = b.w2cat; g.ldw = b.ldw2; g.bias = b.bias2;
            if (b.has_res) {
                for (int k = 0; k < ncur; ++k) {
                    g.A[g.nseg] = cur[k].raw; g.lda[g.nseg] = cur[k].C; g.K[g.nseg] = cur[k].C;
                    ++g.nseg;
                }
            } else {
                g.residual = cur[0].raw; g.ldr = cur[0].C;
            }
            g.out_f32 = o; g.ldo_f32 = b.cout;
            TRY(gemm_dispatch(g, st));
        }
        if (bt) bt->o = o;
        // in float32 the block output is its own GEMM operand (padded rows included)
        outs[i] = Part{o, b.cout, b.cout, dt == ADCDIFF_F32 ? (const void*)o : nullptr};
        if (i >= 1 && i <= 4 && cur[0].raw) outs[i - 1].raw = cur[0].raw;   // keep the converted copy of a skip tensor
        cur[0] = outs[i];
        ncur = 1;
    }
    // output head: Linear(SiLU(GroupNorm(h))) (diffusion_model.py:786-788)
    void* a = ws.take((size_t)Bp * H * e);
    if (tape) {
        tape->a_out = a; tape->B = B; tape->t_rows = t_rows; tape->drop_p = drop_p; tape->drop_seed = drop_seed;
        tape->s_op = cond ? s_op : nullptr;
        tape->valid = !dry;
    }
    if (!dry) {
        TRY(groupnorm_silu(cur[0].f32, cur[0].ld, B, H, num_groups(H), p.out_norm_w, p.out_norm_b, 1e-5f, dt, a, H, nullptr, 0, st));
        adcdiff_gemm_args g = gemm_args(dt, Bp, B, n->O);
        g.nseg = 1; g.A[0] = a; g.lda[0] = H; g.K[0] = H; g.W = n->w_out; g.ldw = H; g.bias = p.out_b;
        g.out_f32 = eps; g.ldo_f32 = lde;
        TRY(gemm_dispatch(g, st));
    }
    if (tape) tape->ws_used = ws.off;
    return ADCDIFF_OK;
}

// The training forward where every GroupNorm can run in a GEMM epilogue (n->fused): the same tape as forward_unfused -- float32 h1
// and o per block, the operand-typed a1 / a2 / block inputs -- with norm1 of the next block, the consuming decoder's norm1 slice of
// a skip tensor and output_norm applied by the producing block's second GEMM (written straight into the concatenated a1 of the
// consumer), and norm2 by the first GEMM when there is no dropout.  One stand-alone GroupNorm (the latent's) plus ten with
// dropout, instead of 25.  The statistics come from the float32 accumulators that out_f32 stores, so the backward pass, which
// recomputes them from the tape, sees the same inputs (diffusion_model.py:875-907).
int adcdiff::forward_taped(adcdiff_unet* n, const float* x, int ldx, int B, const int64_t* t_rows, const float* cemb, int ldce,
                           float* eps, int lde, Bump& ws, bool dry, hipStream_t st, Tape* tape, float drop_p, uint64_t drop_seed) {
    const int Bp = (int)round_up((size_t)B, 128);
    const size_t e = (size_t)n->esz;
    const int H = n->H, dt = n->dtype;
    const adcdiff_unet_params& p = n->prm;
    const bool cond = cemb != nullptr;
    const bool f32 = dt == ADCDIFF_F32;

    const void* s_op = nullptr;
    if (cond || dry) {   // the dry run sizes for the conditional case, the larger one
        void* s = ws.take((size_t)Bp * H * e);
        if (!dry) TRY(temb_silu(n->te_table, H, n->T, t_rows, 0, cemb, ldce, dt, s, H, B, H, st));
        s_op = s;
    }
    // every block's first operand a1 = SiLU(GroupNorm1(cat(inputs))), [Bp, cin], and the output head's
    char* a1[10];
    for (int i = 0; i < 10; ++i) a1[i] = (char*)ws.take((size_t)Bp * n->blk[i].cin * e);
    void* a_out = ws.take((size_t)Bp * H * e);
    void* x_raw = ws.take((size_t)Bp * n->D * e);
    if (!dry) TRY(groupnorm_silu(x, ldx, B, n->D, n->blk[0].g1, p.blocks[0].norm1_w, p.blocks[0].norm1_b, 1e-5f, dt, a1[0], n->D, x_raw, n->D, st));

    const float* o_f32[10];
    const void* o_raw[10];
    for (int i = 0; i < 10; ++i) {
        const BlockPlan& b = n->blk[i];
        const adcdiff_block_params& q = p.blocks[i];
        // the block's input parts: the previous output (the latent for encoder.0) and, in the decoder, the skip
        const float* in_f32[2] = {i == 0 ? x : o_f32[i - 1], i >= 6 ? o_f32[9 - i] : nullptr};
        const int in_ld[2] = {i == 0 ? ldx : n->blk[i - 1].cout, i >= 6 ? n->blk[9 - i].cout : 0};
        const void* in_raw[2] = {i == 0 ? x_raw : o_raw[i - 1], i >= 6 ? o_raw[9 - i] : nullptr};
        const int nparts = i >= 6 ? 2 : 1;
        BlockTape* bt = tape ? &tape->blk[i] : nullptr;
        if (bt) {
            for (int k = 0; k < nparts; ++k) { bt->in_f32[k] = in_f32[k]; bt->in_ld[k] = in_ld[k]; bt->in_c[k] = b.part_c[k]; bt->in_raw[k] = in_raw[k]; }
            bt->nparts = nparts; bt->a1 = a1[i];
        }
        // h1 = linear1(a1) + time_proj(SiLU(t_emb)); without dropout the epilogue also writes a2 = SiLU(GroupNorm2(h1))
        float* h1 = (float*)ws.take((size_t)Bp * b.cout * 4);
        void* a2 = ws.take((size_t)Bp * b.cout * e);
        if (!dry) {
            adcdiff_gemm_args g = gemm_args(dt, Bp, B, b.cout);
            g.nseg = 1; g.A[0] = a1[i]; g.lda[0] = b.cin; g.K[0] = b.cin;
            g.W = b.w1cat; g.ldw = b.ldw1;
            if (cond) {
                g.nseg = 2; g.A[1] = s_op; g.lda[1] = H; g.K[1] = H;
                g.bias = b.bias1_cond;
            } else {
                g.bias = b.bias1;
                g.rowbias = n->tb_table + b.tb_off; g.ldrb = n->sumN; g.rowbias_idx = t_rows; g.rowbias_rows = n->T;
            }
            g.out_f32 = h1; g.ldo_f32 = b.cout;
            if (drop_p <= 0.0f) {
                g.n_gn = 1; g.gn_eps = 1e-5f;
                g.gn[0] = adcdiff_gn_out{q.norm2_w, q.norm2_b, a2, b.cout, b.gs2};
            }
            TRY(gemm_dispatch(g, st));
            // the dropout of diffusion_model.py:898 (counter RNG keyed by layer and row) stays with the stand-alone kernel
            if (drop_p > 0.0f) TRY(groupnorm_silu(h1, b.cout, B, b.cout, b.g2, q.norm2_w, q.norm2_b, 1e-5f, dt, a2, b.cout, nullptr, 0, st, drop_p, drop_seed, (uint32_t)i));
        }
        if (bt) { bt->h1 = h1; bt->a2 = a2; }
        // o = linear2(a2) + residual_proj(input), normalised for its consumers on the way out
        float* o = (float*)ws.take((size_t)Bp * b.cout * 4);
        void* raw = (!f32 && i < 9) ? ws.take((size_t)Bp * b.cout * e) : nullptr;   // in float32 o is its own operand
        if (!dry) {
            adcdiff_gemm_args g = gemm_args(dt, Bp, B, b.cout);
            g.nseg = 1; g.A[0] = a2; g.lda[0] = b.cout; g.K[0] = b.cout;
            g.W = b.w2cat; g.ldw = b.ldw2; g.bias = b.bias2;
            if (b.has_res) {
                for (int k = 0; k < nparts; ++k) { g.A[g.nseg] = in_raw[k]; g.lda[g.nseg] = b.part_c[k]; g.K[g.nseg] = b.part_c[k]; ++g.nseg; }
            } else {
                g.residual = in_raw[0]; g.ldr = b.part_c[0];
            }
            g.out_f32 = o; g.ldo_f32 = b.cout;
            if (raw) { g.out_op = raw; g.ldo_op = b.cout; }
            g.gn_eps = 1e-5f;
            if (i < 9) {
                const BlockPlan& nb = n->blk[i + 1];
                g.gn[g.n_gn++] = adcdiff_gn_out{p.blocks[i + 1].norm1_w, p.blocks[i + 1].norm1_b, a1[i + 1], nb.cin, nb.gs1};
            } else {
                g.gn[g.n_gn++] = adcdiff_gn_out{p.out_norm_w, p.out_norm_b, a_out, H, H / num_groups(H)};
            }
            if (i <= 3) {   // decoder 9 - i reads this output as the second half of its input (diffusion_model.py:781-782)
                const BlockPlan& db = n->blk[9 - i];
                g.gn[g.n_gn++] = adcdiff_gn_out{p.blocks[9 - i].norm1_w + db.part_c[0], p.blocks[9 - i].norm1_b + db.part_c[0],
                                                a1[9 - i] + (size_t)db.part_c[0] * e, db.cin, db.gs1};
            }
            TRY(gemm_dispatch(g, st));
        }
        if (bt) bt->o = o;
        o_f32[i] = o;
        o_raw[i] = f32 ? (const void*)o : raw;
    }
    if (tape) {
        tape->a_out = a_out; tape->B = B; tape->t_rows = t_rows; tape->drop_p = drop_p; tape->drop_seed = drop_seed;
        tape->s_op = cond ? s_op : nullptr;
        tape->valid = !dry;
    }
    if (!dry) {   // output head (diffusion_model.py:786-788): its GroupNorm + SiLU was applied by decoder.3's second GEMM
        adcdiff_gemm_args g = gemm_args(dt, Bp, B, n->O);
        g.nseg = 1; g.A[0] = a_out; g.lda[0] = H; g.K[0] = H; g.W = n->w_out; g.ldw = H; g.bias = p.out_b;
        g.out_f32 = eps; g.ldo_f32 = lde;
        TRY(gemm_dispatch(g, st));
    }
    if (tape) tape->ws_used = ws.off;
    return ADCDIFF_OK;
}

namespace {

// The fused path (see the header comment).  Activations are kept only in the operand type.
struct Act {
    const void* raw;    // layer output o in the operand type, [Bp, C]
    const void* n_next; // SiLU(GroupNorm(o)) under the next block's norm1 (or output_norm)
    const void* n_skip; // SiLU(GroupNorm(o)) under the consuming decoder's norm1 slice (encoder outputs only)
    int C;
};

// A sampling loop's step folded into the forward: the output GEMM applies the schedule update to the latent in place (the
// predicted noise is never stored) and writes what encoder.0 reads in the NEXT step -- the latent in the operand type and
// SiLU(GroupNorm1(latent)) -- into the same workspace slots this step read them from.  Consecutive steps carve the
// workspace identically, so from the second step on the stand-alone GroupNorm is not launched (have_in).
struct StepFold {
    adcdiff_step_update upd;
    float* x_out;    // = x of the forward
    bool have_in;
};

// tb_table: the [T, sumN] table of every block's time_proj output per timestep that the row bias comes from when there is no
// per-row condition: the plan's (no condition) or the caller's (one condition row for the whole batch, uniform_cond_table)
int forward_fused(adcdiff_unet* n, const float* x, int ldx, int B, const int64_t* t_rows, int t_uniform, const float* cemb,
                  int ldce, float* eps, int lde, Bump& ws, bool dry, hipStream_t st, const StepFold* fold = nullptr,
                  const float* tb_table = nullptr) {
    if (!tb_table) tb_table = n->tb_table;
    const int Bp = (int)round_up((size_t)B, 128);
    const size_t e = (size_t)n->esz;
    const int H = n->H, dt = n->dtype;
    const adcdiff_unet_params& p = n->prm;
    const bool cond = cemb != nullptr;

    const void* s_op = nullptr;
    if (cond || dry) {
        void* s = ws.take((size_t)Bp * H * e);
        if (!dry) TRY(temb_silu(n->te_table, H, n->T, t_rows, t_uniform, cemb, ldce, dt, s, H, B, H, st));
        s_op = s;
    }
    // at most 64 rows (the generator's batch is 32): the GEMMs may split K over workgroups through the plan's scratch buffer
    void* const sk = B <= 64 ? n->sk_scratch : nullptr;
    // the latent enters through the one stand-alone GroupNorm of the step: encoder.0's norm1 (diffusion_model.py:888-889)
    Act in;
    void *in_nrm, *in_raw;
    {
        const BlockPlan& b0 = n->blk[0];
        void* nrm = in_nrm = ws.take((size_t)Bp * n->D * e);
        void* raw = in_raw = ws.take((size_t)Bp * n->D * e);
        if (!dry && !(fold && fold->have_in)) TRY(groupnorm_silu(x, ldx, B, n->D, b0.g1, p.blocks[0].norm1_w, p.blocks[0].norm1_b, 1e-5f, dt, nrm, n->D, raw, n->D, st));
        in = Act{raw, nrm, nullptr, n->D};
    }
    Act outs[10];
    for (int i = 0; i < 10; ++i) {
        const BlockPlan& b = n->blk[i];
        const adcdiff_block_params& q = p.blocks[i];
        const Act& c0 = i == 0 ? in : outs[i - 1];
        const Act* c1 = i >= 6 ? &outs[9 - i] : nullptr;   // decoder skip = encoder output 3, 2, 1, 0 (diffusion_model.py:781-782)
        void* a2 = ws.take((size_t)Bp * b.cout * e);
        if (!dry) {
            adcdiff_gemm_args g = gemm_args(dt, Bp, B, b.cout);
            g.nseg = 0;
            g.A[g.nseg] = c0.n_next; g.lda[g.nseg] = c0.C; g.K[g.nseg] = c0.C; ++g.nseg;
            if (c1) { g.A[g.nseg] = c1->n_skip; g.lda[g.nseg] = c1->C; g.K[g.nseg] = c1->C; ++g.nseg; }
            g.W = b.w1cat; g.ldw = b.ldw1;
            if (cond) {
                g.A[g.nseg] = s_op; g.lda[g.nseg] = H; g.K[g.nseg] = H; ++g.nseg;
                g.bias = b.bias1_cond;
            } else {
                g.bias = b.bias1;
                if (t_rows) { g.rowbias = tb_table + b.tb_off; g.ldrb = n->sumN; g.rowbias_idx = t_rows; g.rowbias_rows = n->T; }
                else { g.rowbias = tb_table + (size_t)t_uniform * n->sumN + b.tb_off; g.ldrb = 0; }
            }
            g.n_gn = 1; g.gn_eps = 1e-5f;
            g.gn[0] = adcdiff_gn_out{q.norm2_w, q.norm2_b, a2, b.cout, b.gs2};
            g.splitk_scratch = sk; g.splitk_scratch_bytes = (long long)n->sk_bytes;
            TRY(gemm_dispatch(g, st));
        }
        // second GEMM: linear2 + residual, emitting what the consumers of this block's output read
        void* raw = i < 9 ? ws.take((size_t)Bp * b.cout * e) : nullptr;
        void* n_next = ws.take((size_t)Bp * b.cout * e);
        void* n_skip = i <= 3 ? ws.take((size_t)Bp * b.cout * e) : nullptr;
        // At most 64 rows: a GroupNorm output with 128-channel groups would force 128-column tiles on the few-row kernels (22 us per
        // launch against 8.6 / 11.2 us with 32- / 64-column tiles and K split over workgroups).  Such an output is left to a
        // GroupNorm launch on the float32 result instead (about 4 us); its statistics see the same float32 values.
        const int gs_next = i < 9 ? n->blk[i + 1].gs1 : H / num_groups(H);
        const int gs_skip = i <= 3 ? n->blk[9 - i].gs1 : 0;
        // bf16 batches whose GEMMs may split K (float32 has no few-row kernels, and its small batches stay bit-identical to large ones);
        // a function of B and the plan alone: the dry run sizes the workspace with it
        const bool few = dt == ADCDIFF_BF16 && B <= 64 && debug_value(DBG_GEMM_SPLITK) != 0;
        const bool late_next = few && gs_next > 64, late_skip = few && gs_skip > 64;
        float* o_f32 = (late_next || late_skip) ? (float*)ws.take((size_t)Bp * b.cout * 4) : nullptr;
        if (!dry) {
            adcdiff_gemm_args g = gemm_args(dt, Bp, B, b.cout);
            g.nseg = 1; g.A[0] = a2; g.lda[0] = b.cout; g.K[0] = b.cout;
            g.W = b.w2cat; g.ldw = b.ldw2; g.bias = b.bias2;
            if (b.has_res) {
                g.A[g.nseg] = c0.raw; g.lda[g.nseg] = c0.C; g.K[g.nseg] = c0.C; ++g.nseg;
                if (c1) { g.A[g.nseg] = c1->raw; g.lda[g.nseg] = c1->C; g.K[g.nseg] = c1->C; ++g.nseg; }
            } else {
                g.residual = c0.raw; g.ldr = c0.C;
            }
            g.out_op = raw; g.ldo_op = b.cout;
            if (o_f32) { g.out_f32 = o_f32; g.ldo_f32 = b.cout; }
            g.gn_eps = 1e-5f;
            const float* const next_w = i < 9 ? p.blocks[i + 1].norm1_w : p.out_norm_w;
            const float* const next_b = i < 9 ? p.blocks[i + 1].norm1_b : p.out_norm_b;
            if (!late_next) g.gn[g.n_gn++] = adcdiff_gn_out{next_w, next_b, n_next, b.cout, gs_next};
            const float *skip_w = nullptr, *skip_b = nullptr;
            if (i <= 3) {
                const BlockPlan& db = n->blk[9 - i];
                skip_w = p.blocks[9 - i].norm1_w + db.part_c[0]; skip_b = p.blocks[9 - i].norm1_b + db.part_c[0];
                if (!late_skip) g.gn[g.n_gn++] = adcdiff_gn_out{skip_w, skip_b, n_skip, b.cout, gs_skip};
            }
            g.splitk_scratch = sk; g.splitk_scratch_bytes = (long long)n->sk_bytes;
            TRY(gemm_dispatch(g, st));
            if (late_next) TRY(groupnorm_silu(o_f32, b.cout, B, b.cout, b.cout / gs_next, next_w, next_b, 1e-5f, dt, n_next, b.cout, nullptr, 0, st));
            if (late_skip) TRY(groupnorm_silu(o_f32, b.cout, B, b.cout, b.cout / gs_skip, skip_w, skip_b, 1e-5f, dt, n_skip, b.cout, nullptr, 0, st));
        }
        outs[i] = Act{raw, n_next, n_skip, b.cout};
    }
    if (!dry) {   // output head (diffusion_model.py:786-788): its GroupNorm + SiLU was applied by decoder.3's GEMM
        adcdiff_gemm_args g = gemm_args(dt, Bp, B, n->O);
        g.nseg = 1; g.A[0] = outs[9].n_next; g.lda[0] = H; g.K[0] = H; g.W = n->w_out; g.ldw = H; g.bias = p.out_b;
        if (fold) {
            g.update = fold->upd;
            g.out_f32 = fold->x_out; g.ldo_f32 = ldx;
            g.out_op = in_raw; g.ldo_op = n->D;
            g.n_gn = 1; g.gn_eps = 1e-5f;
            g.gn[0] = adcdiff_gn_out{p.blocks[0].norm1_w, p.blocks[0].norm1_b, in_nrm, n->D, n->blk[0].gs1};
        } else {
            g.out_f32 = eps; g.ldo_f32 = lde;
        }
        g.splitk_scratch = sk; g.splitk_scratch_bytes = (long long)n->sk_bytes;
        TRY(gemm_dispatch(g, st));
    }
    return ADCDIFF_OK;
}

// debug switch step_fold = 0 keeps the sampler's update and encoder.0's GroupNorm as launches of their own
bool fold_enabled() { return debug_value(DBG_STEP_FOLD) != 0; }

int forward_impl(adcdiff_unet* n, const float* x, int ldx, int B, const int64_t* t_rows, int t_uniform, const float* cemb,
                 int ldce, float* eps, int lde, Bump& ws, bool dry, hipStream_t st) {
    return n->fused ? forward_fused(n, x, ldx, B, t_rows, t_uniform, cemb, ldce, eps, lde, ws, dry, st)
                    : forward_unfused(n, x, ldx, B, t_rows, t_uniform, cemb, ldce, eps, lde, ws, dry, st);
}

size_t forward_bytes(adcdiff_unet* n, int B) {
    Bump ws(nullptr, 0);
    forward_impl(n, nullptr, 0, B, nullptr, 0, nullptr, 0, nullptr, 0, ws, true, nullptr);
    // + the eps buffer of the sampling loops and the operand copy of the condition
    const size_t Bp = round_up((size_t)B, 128);
    return ws.off + round_up(Bp * n->O * 4, 256) + round_up(Bp * n->D * n->esz, 256) + round_up(Bp * n->H * n->esz, 256);
}

// One condition row for the whole batch (the caller passes ldce = 0; molecule_generator.py:238-245 expands one protein's features
// to the batch): t_emb + c_emb is then the same for every row, so time_proj(SiLU(t_emb + c_emb)) is a row bias per timestep, as
// without a condition, instead of a K segment of every block's first GEMM.  Built once per call for all T timesteps in float32,
// like the plan's own table (two launches, ~70 us), in the plan's second table: calls that share a plan must share a stream.
int uniform_cond_table(adcdiff_unet* n, const float* cemb, hipStream_t st) {
    TRY(temb_silu(n->te_table, n->H, n->Tp, nullptr, -1, cemb, 0, ADCDIFF_F32, n->s_cond, n->H, n->Tp, n->H, st));
    adcdiff_gemm_args g = gemm_args(ADCDIFF_F32, n->Tp, 0, n->sumN);
    g.nseg = 1; g.A[0] = n->s_cond; g.lda[0] = n->H; g.K[0] = n->H; g.W = n->wtp_all; g.ldw = n->H; g.bias = n->btp_all;
    g.out_f32 = n->tb_cond; g.ldo_f32 = n->sumN;
    return gemm_dispatch(g, st);
}

int check_forward_args(adcdiff_unet* n, const char* who, int B, const void* ws, size_t ws_bytes) {
    if (!n) return fail(ADCDIFF_E_BADARG, "%s: null plan", who);
    if (!n->packed) return fail(ADCDIFF_E_BADARG, "%s: adcdiff_unet_pack has not been called", who);
    if (B <= 0) return fail(ADCDIFF_E_BADARG, "%s: batch %d", who, B);
    if (!ws || !aligned16(ws)) return fail(ADCDIFF_E_BADARG, "%s: workspace null or misaligned", who);
    const size_t need = forward_bytes(n, B);
    if (ws_bytes < need) return fail(ADCDIFF_E_WORKSPACE, "%s: workspace of %zu bytes, %zu needed for batch %d", who, ws_bytes, need, B);
    return ADCDIFF_OK;
}

}  // namespace

extern "C" size_t adcdiff_unet_workspace_bytes(int input_dim, int hidden_dim, int output_dim, int dtype, int batch) {
    adcdiff_unet n{};
    if (batch <= 0 || plan_dims(&n, input_dim, hidden_dim, output_dim, 1, dtype) != ADCDIFF_OK) return 0;
    return forward_bytes(&n, batch);
}

extern "C" int adcdiff_unet_condition_embed(adcdiff_unet* n, const float* cond, int ldc, int B, float* cemb, int ldce,
                                            void* workspace, size_t workspace_bytes, adcdiff_stream_t stream_) {
    TRY(check_forward_args(n, "unet_condition_embed", B, workspace, workspace_bytes));
    if (!cond || !cemb || ldc < n->D || ldce < n->H) return fail(ADCDIFF_E_BADARG, "unet_condition_embed: bad cond / cemb arguments");
    hipStream_t st = as_stream(stream_);
    const int Bp = (int)round_up((size_t)B, 128);
    Bump ws(workspace, workspace_bytes);
    void* c_op = ws.take((size_t)Bp * n->D * n->esz);
    void* hid = ws.take((size_t)Bp * n->H * n->esz);
    TRY(cast_pack(cond, ldc, n->dtype, c_op, n->D, 0, B, n->D, st));
    adcdiff_gemm_args g = gemm_args(n->dtype, Bp, B, n->H);
    g.nseg = 1; g.A[0] = c_op; g.lda[0] = n->D; g.K[0] = n->D; g.W = n->w_c0; g.ldw = n->D; g.bias = n->prm.cond0_b; g.act = 1;
    g.out_op = hid; g.ldo_op = n->H;
    TRY(gemm_dispatch(g, st));
    g = gemm_args(n->dtype, Bp, B, n->H);
    g.nseg = 1; g.A[0] = hid; g.lda[0] = n->H; g.K[0] = n->H; g.W = n->w_c2; g.ldw = n->H; g.bias = n->prm.cond2_b;
    g.out_f32 = cemb; g.ldo_f32 = ldce;
    return gemm_dispatch(g, st);
}

extern "C" int adcdiff_unet_forward(adcdiff_unet* n, const float* x, int ldx, int B, const int64_t* t_rows, int t_uniform,
                                    const float* cemb, int ldce, float* eps, int lde, void* workspace, size_t workspace_bytes,
                                    adcdiff_stream_t stream_) {
    TRY(check_forward_args(n, "unet_forward", B, workspace, workspace_bytes));
    if (!x || !eps || ldx < n->D || lde < n->O || !aligned16(x) || !aligned16(eps) || ldx % 4 || lde % 4)
        return fail(ADCDIFF_E_BADARG, "unet_forward: x / eps null, misaligned or leading dimension too small");
    if (!t_rows && (t_uniform < 0 || t_uniform >= n->T)) return fail(ADCDIFF_E_BADARG, "unet_forward: timestep %d outside 0..%d", t_uniform, n->T - 1);
    if (cemb && (ldce < n->H || !aligned16(cemb))) return fail(ADCDIFF_E_BADARG, "unet_forward: bad condition embedding");
    Bump ws(workspace, workspace_bytes);
    return forward_impl(n, x, ldx, B, t_rows, t_uniform, cemb, ldce, eps, lde, ws, false, as_stream(stream_));
}

extern "C" int adcdiff_ddpm_sample_loop(adcdiff_unet* n, float* x, int ldx, int B, const float* cemb, int ldce,
                                        const adcdiff_ddpm_coef* coef, int t_start, int t_stop, const float* noise,
                                        long long noise_step_stride, uint64_t seed, uint64_t row_offset, void* workspace,
                                        size_t workspace_bytes, adcdiff_stream_t stream_) {
    TRY(check_forward_args(n, "ddpm_sample_loop", B, workspace, workspace_bytes));
    if (!x || !coef || ldx < n->D || !aligned16(x) || ldx % 4) return fail(ADCDIFF_E_BADARG, "ddpm_sample_loop: bad x / coef");
    if (t_start >= n->T || t_stop < 0 || t_stop > t_start) return fail(ADCDIFF_E_BADARG, "ddpm_sample_loop: steps %d..%d outside 0..%d", t_start, t_stop, n->T - 1);
    if (cemb && ((ldce != 0 && ldce < n->H) || !aligned16(cemb))) return fail(ADCDIFF_E_BADARG, "ddpm_sample_loop: bad condition embedding");
    const int Bp = (int)round_up((size_t)B, 128);
    const float* tb_table = nullptr;
    if (cemb && ldce == 0 && n->fused) {
        TRY(uniform_cond_table(n, cemb, as_stream(stream_)));
        tb_table = n->tb_cond;
        cemb = nullptr;
    }
    for (int t = t_start; t >= t_stop; --t) {
        Bump ws(workspace, workspace_bytes);
        float* eps = (float*)ws.take((size_t)Bp * n->O * 4);
        const float* nz = noise ? noise + (size_t)(t_start - t) * (size_t)noise_step_stride : nullptr;
        if (n->fold && fold_enabled()) {
            StepFold f{};
            f.upd.kind = 1;
            f.upd.c[0] = coef[t].sqrt_recip_alpha; f.upd.c[1] = coef[t].beta; f.upd.c[2] = coef[t].sqrt_one_minus_ac; f.upd.c[3] = coef[t].sigma;
            f.upd.x = x; f.upd.ldx = ldx; f.upd.noise = nz; f.upd.ldn = n->O;
            f.upd.seed = seed; f.upd.step = (uint32_t)t; f.upd.row_offset = row_offset;
            f.x_out = x; f.have_in = t != t_start;
            TRY(forward_fused(n, x, ldx, B, nullptr, t, cemb, ldce, nullptr, 0, ws, false, as_stream(stream_), &f, tb_table));
            continue;
        }
        if (tb_table) TRY(forward_fused(n, x, ldx, B, nullptr, t, nullptr, 0, eps, n->O, ws, false, as_stream(stream_), nullptr, tb_table));
        else TRY(forward_impl(n, x, ldx, B, nullptr, t, cemb, ldce, eps, n->O, ws, false, as_stream(stream_)));
        TRY(adcdiff_ddpm_step(x, ldx, eps, n->O, nz, n->O, x, ldx, B, n->O, coef[t], seed, (uint32_t)t, row_offset, stream_));
    }
    return ADCDIFF_OK;
}

extern "C" int adcdiff_ddim_sample_loop(adcdiff_unet* n, float* x, int ldx, int B, const float* cemb, int ldce, const int* timesteps,
                                        const adcdiff_ddim_coef* coef, int nsteps, const float* noise, long long noise_step_stride,
                                        uint64_t seed, uint64_t row_offset, void* workspace, size_t workspace_bytes,
                                        adcdiff_stream_t stream_) {
    TRY(check_forward_args(n, "ddim_sample_loop", B, workspace, workspace_bytes));
    if (!x || !coef || !timesteps || nsteps <= 0 || ldx < n->D || !aligned16(x) || ldx % 4)
        return fail(ADCDIFF_E_BADARG, "ddim_sample_loop: bad x / coef / timesteps");
    if (cemb && ((ldce != 0 && ldce < n->H) || !aligned16(cemb))) return fail(ADCDIFF_E_BADARG, "ddim_sample_loop: bad condition embedding");
    const int Bp = (int)round_up((size_t)B, 128);
    const float* tb_table = nullptr;
    if (cemb && ldce == 0 && n->fused) {
        TRY(uniform_cond_table(n, cemb, as_stream(stream_)));
        tb_table = n->tb_cond;
        cemb = nullptr;
    }
    for (int i = 0; i < nsteps; ++i) {
        const int t = timesteps[i];
        if (t < 0 || t >= n->T) return fail(ADCDIFF_E_BADARG, "ddim_sample_loop: timestep %d outside 0..%d", t, n->T - 1);
        Bump ws(workspace, workspace_bytes);
        float* eps = (float*)ws.take((size_t)Bp * n->O * 4);
        const float* nz = noise ? noise + (size_t)i * (size_t)noise_step_stride : nullptr;
        if (n->fold && fold_enabled()) {
            StepFold f{};
            f.upd.kind = 2;
            f.upd.c[0] = coef[i].c_eps; f.upd.c[1] = coef[i].sqrt_ac_t; f.upd.c[2] = coef[i].sqrt_ac_prev; f.upd.c[3] = coef[i].dir_coef;
            f.upd.c[4] = coef[i].sigma;
            f.upd.x = x; f.upd.ldx = ldx; f.upd.noise = nz; f.upd.ldn = n->O;
            f.upd.seed = seed; f.upd.step = (uint32_t)t; f.upd.row_offset = row_offset;
            f.x_out = x; f.have_in = i != 0;
            TRY(forward_fused(n, x, ldx, B, nullptr, t, cemb, ldce, nullptr, 0, ws, false, as_stream(stream_), &f, tb_table));
            continue;
        }
        if (tb_table) TRY(forward_fused(n, x, ldx, B, nullptr, t, nullptr, 0, eps, n->O, ws, false, as_stream(stream_), nullptr, tb_table));
        else TRY(forward_impl(n, x, ldx, B, nullptr, t, cemb, ldce, eps, n->O, ws, false, as_stream(stream_)));
        TRY(adcdiff_ddim_step(x, ldx, eps, n->O, nz, n->O, x, ldx, B, n->O, coef[i], seed, (uint32_t)t, row_offset, stream_));
    }
    return ADCDIFF_OK;
}
