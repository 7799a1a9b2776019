// model.cpp -- host orchestration of the VAE step on one MI355X and the C ABI (include/argsim_vae.h).
//
// Restates the dataflow of reference src/model.py:75-189 as a fixed sequence of kernel launches
// on one HIP stream: prep -> gather -> 3x(bidirectional GRU) -> latent -> 3x GRU decoder ->
// out affine -> tied logits -> softmax-CE, then the hand-derived backward in reverse order and
// TF-style Adam.  No tracing compiler, no autograd: every buffer lives in one workspace laid out
// by a bump allocator.
#include "../../include/argsim_vae.h"
#include "kernels.h"

#include <fcntl.h>
#include <sys/file.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

using namespace avae;

namespace {

struct ParamEntry {
    std::string name;
    int64_t offset;
    int ndim;
    int64_t shape[4];
    int g16;          // rows are stored gate-interleaved (GRU W/R/bW/bR)
    int bucket;
};

struct GruP { int64_t W, R, bW, bR; };     // offsets into the flat state

static std::string g_create_err;

}  // namespace

struct avae_ctx {
    avae_config cfg{};
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    std::vector<ParamEntry> params;
    int64_t numel = 0;
    std::vector<std::pair<int64_t, int64_t>> buckets;    // (offset, count) in completion order
    float *P = nullptr, *G = nullptr, *M = nullptr, *Vv = nullptr;
    int64_t step = 0;
    avae_grad_hook hook = nullptr; void* hook_user = nullptr;
    std::vector<int> hook_pending;     // buckets complete but not yet announced (see hook_flush)
    int persistent = 1;
    bool first_step_checked = false;
    int gru_ablate = 0, gru_force_slow = 0, gru_stagger = 0, gru_item = 2;
    int skinny = 1;       // a few rows (latent block, one-step top layer; backward: remainder rows, small products): one 32x32 tile per workgroup,
                          // K split over its waves (gemm_f32.hip); 0: the tiled forms
    const int* expect_ptr[3] = {nullptr, nullptr, nullptr}; int expect_val[3] = {0, 0, 0};      // (dyn_expected)
    int rows_form = 0;    // one-shot: the next gemm() call's rows are the batch rows -- skinny form whatever the batch size (see gemm)
    int compact = 2;      // encoder activations stored over the REAL rows only (row_map / GruArgs::rowmap): padded rows of a ragged batch cost nothing in the
                          // encoder's GEMMs.  0 off, 1 on, 2 auto: on where the share of real positions the previous calls reported is below 0.85 (fill_hint)
    int skip_pad = 1;     // team GRU kernels skip the steps behind a row block's longest row (rows sorted by length, ops.hip row_order); 0: every step of every row
    int shared_device = 0; int lock_fd = -1;      // option shared_device: persistent launches are taken one at a time ACROSS processes (DeviceTurn)
    int dyn_split = 1;    // ragged batches: narrow backward GEMMs over few expected rows split K instead of leaving the chip at one workgroup per CU (gemm())
    int dyn_thin = 1;     // device-row-count GEMMs with a narrow output run 64x64 tiles (gemm())
    int enc_top1 = 1;     // the top encoder layer's backward direction runs its ONE live step only (gru.hip "one step from a zero state"); 0: all S steps like the reference's graph
    int table_l1 = 1;     // layers fed by embedding rows project the TABLE once and gather / scatter by id where a batch has more tokens than the vocabulary (use_table)
    int bf16_act = 1;     // compute_dtype 1: h / h_prev row-major copies written as bf16 by the forward team kernels (the GEMM operands as they stand)
    int bf16_sv = 1;      // compute_dtype 1: saved gates as bf16 where a layer's forward and backward both run the team kernels
    int bf16_tn = 1;      // compute_dtype 1: the BPTT team kernels write the gate gradients as bf16 and the weight-gradient GEMMs read row-major bf16 operands through transposing LDS loads (gemm_bf16_tn): no transposed copies
    int logits16 = 1;     // compute_dtype 1, training forward: the logits leave the phased GEMM as an fp16 panel that softmax_ce turns into the bf16 gradient in place (no fp32 logits)
    int bf16_nt8 = 1;     // compute_dtype 1: NT GEMMs on the phased LDS-DMA kernel (gemm_bf16_p8.hip) where the shape allows (0: the register-staged 256x256 kernel)
    int bf16_direct = 0;  // (measured at configs[2]: 50.2 ms with it, 43.3 ms with the conversion passes + 256x256 NT kernel: off)
    //  compute_dtype 1: GEMMs read their fp32 operands directly and round to bf16 while staging (0: conversion passes + NT kernel)
    int gru_spec = 2;     // team kernels load a consumer's operand at once, without a probe round trip in front of it: 0 never, 1 always, 2 where few rows are alive per
                          // step (spec_pick: RAGGED 256 x 64 10.41 -> 10.23 ms; always-on costs a FULL 100 x 512 batch 1 %, 71.5 -> 72.3 ms, and a FULL 256 x 64 nothing)
    int bwd_rs = 2;       // fp32 BPTT team kernels in the reduce-scatter form (gru_rs.hip: own gate columns x resident R slice, partial dH summed through the
                          // exchange): 0 never, 1 wherever the geometry allows, 2 auto -- where few rows are alive per step (rs_pick)
    int gru_bf16 = 1;     // compute_dtype 1 only: the recurrent product of the team kernels takes bf16 operands too (0: fp32 recurrence)
    // offsets
    int64_t oE = 0, oKout = 0, oBout = 0, oWmu = 0, oBmu = 0, oWlv = 0, oBlv = 0, oWex = 0, oBex = 0;
    std::vector<GruP> enc;     // per layer: W = [fwd;bwd] (6D,In), R = [fwd;bwd], bW (6D), bR (6D)
    std::vector<GruP> dec;
    // small persistent device state
    float* losses = nullptr;   // [3]
    float* acc = nullptr;      // [2] sum loss_gen_samp, sum kld
    int* errw = nullptr;       // GRU spin time-out word
    unsigned* counters = nullptr;
    float* scratch = nullptr;  // staging for get/set tensor
    int64_t scratch_n = 0;
    // workspace
    char* ws = nullptr; size_t ws_cap = 0;
    // last forward geometry
    int B = 0, Ss = 0, St = 0;
    // optional per-kernel-class timing with HIP events on the launch stream (bench.py roofline leg)
    int timing = 0, timing_on = 0;
    // bf16-operand GEMM mode (compute_dtype = 1): converted operand panels
    unsigned short *bfA = nullptr, *bfB = nullptr; size_t bfA_cap = 0, bfB_cap = 0;
    float* slab = nullptr; size_t slab_floats = 0;     // bf16 mode: the K slices' partial tiles of the weight-gradient GEMMs (gemm_bf16_p8.hip; 512 tiles of 256 x 256)
    unsigned short* keep_a16 = nullptr;       // one-shot: the next bf16-mode GEMM converts its (k-contiguous) A operand HERE and leaves it for the backward (gemm_raw)
    unsigned short* bfP = nullptr; size_t bfP_cap = 0;     // bf16 mode: (softmax - onehot)/N as written by softmax_ce_kernel, (N,V) bf16
    // (dyn / dyn_max: a GEMM whose M or K is a device-side count -- its FLOPs are scaled by count / static bound at collection)
    struct Stamp { hipEvent_t a, b; int cls; double flops; const int* dyn; int dyn_max; };
    std::vector<Stamp> stamps; size_t stamps_used = 0;
    // fill hint: the real source positions of an earlier call, copied to pinned host memory without a synchronisation (whatever has
    // arrived is read; it only ever decides the LAYOUT, never a value) and the padded positions of the call that issued the copy
    int32_t* hint_dev = nullptr; volatile int32_t* hint_host = nullptr;
    const int32_t *cnt_src = nullptr, *cnt_tgt = nullptr;     // present-id counts of the last forward (table-fed layers), device
};

namespace {

#define AV_CHECK(expr)                                                                              \
    do { hipError_t e_ = (expr); if (e_ != hipSuccess) {                                            \
        char b_[512]; snprintf(b_, sizeof b_, "%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); \
        h->err = b_; return 1; } } while (0)
#define AV_TRY(expr) do { int r_ = (expr); if (r_) return r_; } while (0)

int fail(avae_ctx* h, const std::string& m) { h->err = m; return 1; }

// persistent GRU launches: the residency check of gru.hip answers hipErrorCooperativeLaunchTooLarge
#define AV_GRU(expr)                                                                                                \
    do { hipError_t e_ = (expr);                                                                                    \
         if (e_ == hipErrorCooperativeLaunchTooLarge)                                                               \
             return fail(h, "persistent GRU kernel: its workgroups cannot all be resident on this device at once (occupancy query x CU count < grid); " \
                            "run with avae_set_option(\"persistent\", 0)");                                         \
         if (e_ != hipSuccess) { char b_[512]; snprintf(b_, sizeof b_, "%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); \
             h->err = b_; return 1; } } while (0)

// A persistent GRU launch needs every CU, so two PROCESSES computing on one device (several data-parallel ranks rehearsed on one
// GPU, a second job) can each get part of the chip and both run into the 2 s exchange time-out.  Option shared_device = 1: every
// persistent launch is taken in turn across processes -- an exclusive flock on a per-device lock file from before the launch
// is enqueued until it has COMPLETED (one stream synchronisation per launch: slower, never wrong).  Non-persistent kernels of another
// process only delay a persistent launch; they cannot strand it.
struct DeviceTurn {
    avae_ctx* h; bool held = false;
    explicit DeviceTurn(avae_ctx* h_) : h(h_) {
        if (!h->shared_device || !h->persistent) return;
        if (h->lock_fd < 0) {
            char path[64]; snprintf(path, sizeof path, "/tmp/argsim_vae_dev%d.lock", h->device);
            h->lock_fd = open(path, O_CREAT | O_RDWR, 0666);
        }
        if (h->lock_fd >= 0 && flock(h->lock_fd, LOCK_EX) == 0) held = true;
    }
    ~DeviceTurn() { if (held) { (void)hipStreamSynchronize(h->stream); (void)flock(h->lock_fd, LOCK_UN); } }
};

// kernel classes for the timing hook: 0 = MFMA GEMM, 1 = GRU forward, 2 = GRU backward
struct Timed {
    avae_ctx* h; avae_ctx::Stamp* s = nullptr;
    Timed(avae_ctx* h_, int cls, double flops, const int* dyn = nullptr, int dyn_max = 0) : h(h_) {
        if (!h->timing) return;
        if (h->stamps_used == h->stamps.size()) {
            avae_ctx::Stamp n{};
            if (hipEventCreate(&n.a) != hipSuccess || hipEventCreate(&n.b) != hipSuccess) return;
            h->stamps.push_back(n);
        }
        s = &h->stamps[h->stamps_used++];
        s->cls = cls; s->flops = flops; s->dyn = dyn; s->dyn_max = dyn_max;
        (void)hipEventRecord(s->a, h->stream);
    }
    ~Timed() { if (s) (void)hipEventRecord(s->b, h->stream); }
};

// -------------------------------------------------------------------------------- workspace
struct Ws {
    // ints
    int32_t *src_tm, *lens_src, *lens_tgt, *lead, *gold, *rank, *cidx, *ntok, *pred;
    // forward
    float *emb_src, *emb_tgt, *ew, *dew;
    std::vector<float*> e_gi, e_hs, e_sv[2], e_hp[2];
    std::vector<float*> d_gi, d_hd, d_sv, d_hp;
    float *hpick, *mu, *lv, *z, *eps, *kld, *h0;
    float *xlast, *gib, *svb, *dgib, *dghb, *dxl;     // one-step top backward direction (top_one_step): (B,2D) (B,3D) (B,D,4) (B,3D) (B,3D) (B,2D)
    float *hc, *ho, *logits;
    float *loss_samp, *errt_samp;
    // backward
    float *dho, *dhc, *dhd[2], *dgi_d, *dgh_d, *dh0, *carry, *dh0sum, *dz, *dmu, *dlv, *dhpick;
    float *dhs[2], *dgi_e, *dgh_e, *demb_src, *demb_tgt;
    std::vector<unsigned short*> e_hs16, d_hd16, e_hp16[2], d_hp16;      // bf16 mode: h / h_prev as the forward team kernels write them (bf16_act)
    std::vector<char> act_e, act_d, acth_e, acth_d;                       // per layer: hs16 / hp16 in use this call
    std::vector<unsigned short*> x16_e, x16_d;                 // bf16 mode: the layer inputs as the forward GEMMs converted them (row-major: the backward's TN operand)
    const unsigned short* x16_kept_e(int i) const { return x16_valid ? x16_e[i] : nullptr; }
    const unsigned short* x16_kept_d(int i) const { return x16_valid ? x16_d[i] : nullptr; }
    bool x16_valid = false;
    unsigned short *dgi16_d, *dgh16_d, *dgi16_e, *dgh16_e;      // bf16 mode: the gate gradients as the BPTT team kernels write them (bf16_tn)
    int32_t* scat;                        // embed_scatter_add2's token lists
    int32_t *grp_src, *grp_tgt;           // id_groups_build scratch of the two id sources (use_table)
    int32_t *tokrow_src, *tokrow_tgt;
    float* xbuf; size_t xbuf_floats;      // exchange scratch of the GRU team kernels (GruArgs::xbuf)
    // row orders of the padding-skipping team kernels (build_row_orders): 0 = encoder, both directions; 1 = encoder, one job
    // (top layer); 2 = decoder.  ord_ok: built for this call with geometry (ord_T, ord_cpj)
    int32_t *ord_perm[3], *ord_slens[3]; int ord_T[3], ord_cpj[3]; bool ord_ok[3];
    // compact encoder layout (build_compact): map_src[(t, b)] = row among the real source positions or -1, nsrc = how many
    int32_t *map_src, *nact_src, *nsrc; bool compact;
    // the decoder's: map_tgt[(t, b)] over the positions t <= (last non-eos target position of row b) + 1, ntgt = how many
    int32_t *map_tgt, *nact_tgt, *ntgt; bool compact_d;
    // rows of the GRU team kernels' launch geometry (gru_team_batch): = B where B itself has one, else the next row count that has;
    // the slots beyond B hold phantom rows (GruArgs::Bx), which exist through the row order + the compact layout only
    int Bx;
    int bx_enc() const { return compact ? Bx : 0; }
    int bx_dec() const { return compact_d ? Bx : 0; }
};

struct Bump {
    char* base; size_t off = 0;
    template <class T> T* take(size_t n) {
        off = (off + 255) & ~(size_t)255;
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += n * sizeof(T);
        return p;
    }
};

// A layer whose input is an embedding row (encoder layer 1: E[src]; decoder layer 1: E[lead]) computes W E[id].  With more
// tokens than vocabulary entries the projection is taken once over the U <= V ids present in the batch (their E rows
// gathered, one GEMM with a device-side row count) and gathered by id; the backward sums the per-token gate gradients by id
// (rows_group_sum) and runs dW = (sum)^T E_present and dE[present] += (sum) W over U rows.  The same products grouped by id:
// exact algebra, a different summation order in the backward.  The compact E rows live in emb_src / emb_tgt (unused
// otherwise in this mode), the per-id gradient of E in demb_src / demb_tgt.
// A batch WITHOUT a team-kernel geometry of its own reaches the team kernels through the compact layout, which
// needs the first layers table-fed: such a batch takes the table path below the vocabulary size as well (never more rows than
// tokens: U <= rows).
static bool phantom_batch(const avae_ctx* h, int B)
{
    return h->compact && h->skip_pad && h->persistent && h->cfg.dim_emb == 512 && gru_team_batch(B) > B;
}
static bool use_table(const avae_ctx* h, int rows, int B)
{
    if (!h->table_l1 || !id_groups_supported(h->cfg.dim_tgt)) return false;
    return rows >= h->cfg.dim_tgt || (rows >= 1024 && phantom_batch(h, B));
}

void layout(avae_ctx* h, Bump& b, Ws& w, int B, int Ss, int St, bool train)
{
    const int D = h->cfg.dim_emb, V = h->cfg.dim_tgt, R = h->cfg.dim_rep, L = h->cfg.rnn_layers;
    const size_t T = St + 1, rs = (size_t)Ss * B, rt = T * B;
    w.src_tm = b.take<int32_t>(rs); w.lens_src = b.take<int32_t>(B); w.lens_tgt = b.take<int32_t>(B);
    w.lead = b.take<int32_t>(rt); w.gold = b.take<int32_t>(rt); w.rank = b.take<int32_t>(rt);
    w.cidx = b.take<int32_t>(rt); w.ntok = b.take<int32_t>(4 + kPrepChunks); w.pred = b.take<int32_t>(rt);      // (ntok[4..]: prep_ids' chunk counts)
    w.emb_src = b.take<float>(rs * D); w.emb_tgt = b.take<float>(rt * D);
    // table-fed first layers (use_table), taken only where this geometry runs them: W E over the present ids ((U, 6D) encoder
    // layer 1, then (U, 3D) decoder layer 1), the token groups of the id source, the projection row of every token
    const bool tab_s = use_table(h, (int)rs, B), tab_t = use_table(h, (int)rt, B);
    w.ew = b.take<float>(tab_s ? (size_t)V * 6 * D : (tab_t ? (size_t)V * 3 * D : 0));
    w.grp_src = b.take<int32_t>(tab_s ? id_groups_ints(rs, V) : 0); w.grp_tgt = b.take<int32_t>(tab_t ? id_groups_ints(rt, V) : 0);
    w.tokrow_src = b.take<int32_t>(tab_s ? rs : 0); w.tokrow_tgt = b.take<int32_t>(tab_t ? rt : 0);
    w.e_gi.resize(L); w.e_hs.resize(L);
    for (int d = 0; d < 2; ++d) { w.e_sv[d].resize(L); w.e_hp[d].resize(L); }
    w.d_gi.resize(L); w.d_hd.resize(L); w.d_sv.resize(L); w.d_hp.resize(L);
    w.e_hs16.assign(L, nullptr); w.d_hd16.assign(L, nullptr); w.e_hp16[0].assign(L, nullptr); w.e_hp16[1].assign(L, nullptr); w.d_hp16.assign(L, nullptr);
    w.act_e.assign(L, 0); w.act_d.assign(L, 0); w.acth_e.assign(L, 0); w.acth_d.assign(L, 0);
    if (train && h->cfg.compute_dtype == 1 && h->bf16_act && h->bf16_tn && D % 8 == 0)
        for (int i = 0; i < L; ++i) {
            w.e_hs16[i] = b.take<unsigned short>(rs * 2 * D); w.e_hp16[0][i] = b.take<unsigned short>(rs * D); w.e_hp16[1][i] = b.take<unsigned short>(rs * D);
            if (i < L - 1) w.d_hd16[i] = b.take<unsigned short>(rt * D);
            w.d_hp16[i] = b.take<unsigned short>(rt * D);
        }
    w.x16_e.assign(L, nullptr); w.x16_d.assign(L, nullptr);
    w.x16_valid = train && h->cfg.compute_dtype == 1 && h->bf16_tn && !h->bf16_direct && D % 8 == 0;
    if (w.x16_valid)
        for (int i = 1; i < L; ++i) { w.x16_e[i] = b.take<unsigned short>(rs * 2 * D); w.x16_d[i] = b.take<unsigned short>(rt * D); }
    for (int i = 0; i < L; ++i) {
        w.e_gi[i] = b.take<float>(rs * 6 * D);
        w.e_hs[i] = b.take<float>(rs * 2 * D);
        for (int d = 0; d < 2; ++d) {
            w.e_sv[d][i] = train ? b.take<float>(rs * 4 * D) : nullptr;
            w.e_hp[d][i] = train ? b.take<float>(rs * D) : nullptr;
        }
    }
    for (int i = 0; i < L; ++i) {
        w.d_gi[i] = b.take<float>(rt * 3 * D);
        w.d_hd[i] = b.take<float>(rt * D);
        w.d_sv[i] = train ? b.take<float>(rt * 4 * D) : nullptr;
        w.d_hp[i] = train ? b.take<float>(rt * D) : nullptr;
    }
    {   // every GRU launch exchanges through the same scratch: the largest launch is both encoder directions (forward:
        // D floats per row and step, backward: 3D) or one decoder layer
        const int bx = gru_team_batch(B);
        w.Bx = bx > 0 ? bx : B;
        const size_t rows = std::max<size_t>(2 * Ss, T) * (size_t)w.Bx;
        w.xbuf_floats = rows * D * (train ? 3 : 1);
        if (train && D == 512 && h->bwd_rs) w.xbuf_floats = std::max(w.xbuf_floats, gru_bwd_rs_xbuf_floats(2, w.Bx));      // (the reduce-scatter BPTT's ring, gru_rs.hip)
        w.xbuf = b.take<float>(w.xbuf_floats);
    }
    w.map_src = b.take<int32_t>(rs); w.nact_src = b.take<int32_t>(Ss + 1); w.nsrc = b.take<int32_t>(4); w.compact = false;
    w.map_tgt = b.take<int32_t>(rt); w.nact_tgt = b.take<int32_t>(T + 1); w.ntgt = b.take<int32_t>(4); w.compact_d = false;
    for (int k = 0; k < 3; ++k) { w.ord_perm[k] = b.take<int32_t>(w.Bx); w.ord_slens[k] = b.take<int32_t>(w.Bx); w.ord_ok[k] = false; w.ord_T[k] = w.ord_cpj[k] = 0; }
    w.hpick = b.take<float>((size_t)B * 2 * D);
    w.xlast = b.take<float>((size_t)B * 2 * D); w.gib = b.take<float>((size_t)B * 3 * D); w.svb = b.take<float>((size_t)B * 4 * D);
    w.dgib = b.take<float>(train ? (size_t)B * 3 * D : 0); w.dghb = b.take<float>(train ? (size_t)B * 3 * D : 0); w.dxl = b.take<float>(train ? (size_t)B * 2 * D : 0);
    w.mu = b.take<float>((size_t)B * R); w.lv = b.take<float>((size_t)B * R); w.z = b.take<float>((size_t)B * R);
    w.eps = b.take<float>((size_t)B * R); w.kld = b.take<float>((size_t)B * R);
    w.h0 = b.take<float>((size_t)B * D);
    w.hc = b.take<float>(rt * D); w.ho = b.take<float>(rt * D); w.logits = b.take<float>(rt * V);
    w.loss_samp = b.take<float>(rt); w.errt_samp = b.take<float>(rt);
    if (train) {
        w.dho = b.take<float>(rt * D); w.dhc = b.take<float>(rt * D);
        w.dhd[0] = b.take<float>(rt * D); w.dhd[1] = b.take<float>(rt * D);
        w.dgi_d = b.take<float>(rt * 3 * D); w.dgh_d = b.take<float>(rt * 3 * D);
        w.dh0 = b.take<float>((size_t)L * B * D); w.carry = b.take<float>((size_t)3 * w.Bx * D);
        w.dh0sum = b.take<float>((size_t)B * D);
        w.dz = b.take<float>((size_t)B * R); w.dmu = b.take<float>((size_t)B * R); w.dlv = b.take<float>((size_t)B * R);
        w.dhpick = b.take<float>((size_t)B * 2 * D);
        w.dhs[0] = b.take<float>(rs * 2 * D); w.dhs[1] = b.take<float>(rs * 2 * D);
        w.dgi_e = b.take<float>(rs * 6 * D); w.dgh_e = b.take<float>(rs * 6 * D);
        w.demb_src = b.take<float>(rs * D); w.demb_tgt = b.take<float>(rt * D);
        const bool g16 = h->cfg.compute_dtype == 1;
        w.dgi16_d = b.take<unsigned short>(g16 ? rt * 3 * D : 0); w.dgh16_d = b.take<unsigned short>(g16 ? rt * 3 * D : 0);
        w.dgi16_e = b.take<unsigned short>(g16 ? rs * 6 * D : 0); w.dgh16_e = b.take<unsigned short>(g16 ? rs * 6 * D : 0);
        w.scat = b.take<int32_t>(embed_scatter_scratch_ints(rs + rt, V));
        w.dew = b.take<float>(tab_s ? (size_t)V * 6 * D : (tab_t ? (size_t)V * 3 * D : 0));  // gate gradients of a table-fed layer summed by id
    }
}

int get_ws(avae_ctx* h, Ws& w, int B, int Ss, int St, bool train)
{
    Bump probe{nullptr};
    layout(h, probe, w, B, Ss, St, train);
    size_t need = probe.off + 4096;
    if (need > h->ws_cap) {
        AV_CHECK(hipStreamSynchronize(h->stream));
        if (h->ws) AV_CHECK(hipFree(h->ws));
        h->ws = nullptr; h->ws_cap = 0;
        h->cnt_src = h->cnt_tgt = nullptr;                    // (pointed into the old arena)
        for (auto& sp : h->stamps) sp.dyn = nullptr;
        size_t cap = need + need / 8;
        AV_CHECK(hipMalloc(reinterpret_cast<void**>(&h->ws), cap));
        h->ws_cap = cap;
    }
    Bump real{h->ws};
    layout(h, real, w, B, Ss, St, train);
    return 0;
}

// -------------------------------------------------------------------------------- helpers
int grow_bf16(avae_ctx* h, unsigned short** buf, size_t* cap, size_t need)
{
    if (need <= *cap) return 0;
    AV_CHECK(hipStreamSynchronize(h->stream));
    if (*buf) AV_CHECK(hipFree(*buf));
    *buf = nullptr; *cap = 0;
    size_t n = need + need / 8;
    AV_CHECK(hipMalloc(reinterpret_cast<void**>(buf), n * sizeof(unsigned short)));
    *cap = n;
    return 0;
}

// what the host expects a device-side row count to be (the compact layout's counts: an earlier call's fill, build_compact); 0 = unknown
static int dyn_expected(const avae_ctx* h, const int* dyn, int dyn_kind)
{
    if (dyn_kind != 1 || !dyn) return 0;
    for (int i = 0; i < 3; ++i) if (dyn == h->expect_ptr[i]) return h->expect_val[i];
    return 0;
}
// second problem of a pair (same shape, layout, scalars): see GemmArgs in kernels.h
struct Pair { const float* A; const float* B; float* C; const float* bias; };

int gemm_raw(avae_ctx* h, bool a_mc, bool b_nc, const float* A, int lda, const float* Bm, int ldb, float* C, int ldc,
             int M, int N, int K, float alpha, const float* bias, int accumulate, int split_k, const int* dyn, int dyn_kind,
             int thin = 0, const Pair* pair = nullptr, unsigned short* c16 = nullptr)
{
    if (pair && h->cfg.compute_dtype != 0) {     // the other GEMM kernels take one problem per launch
        AV_TRY(gemm_raw(h, a_mc, b_nc, A, lda, Bm, ldb, C, ldc, M, N, K, alpha, bias, accumulate, split_k, dyn, dyn_kind, thin));
        return gemm_raw(h, a_mc, b_nc, pair->A, lda, pair->B, ldb, pair->C, ldc, M, N, K, alpha, pair->bias, accumulate, split_k, dyn, dyn_kind, thin);
    }
    GemmArgs g{A, Bm, C, bias, M, N, K, lda, ldb, ldc, alpha, accumulate, split_k, dyn, dyn_kind, dyn_expected(h, dyn, dyn_kind), thin,
               pair ? pair->A : nullptr, pair ? pair->B : nullptr, pair ? pair->C : nullptr, pair ? pair->bias : nullptr};
    Timed t(h, 0, 2.0 * M * N * K * (pair ? 2 : 1), dyn, dyn_kind == 1 ? M : (dyn_kind == 2 ? K : 0));
    if (h->cfg.compute_dtype == 1 && h->bf16_direct) {
        // bf16 operands rounded on the way into LDS, straight from the fp32 operands in whatever layout: no conversion passes
        AV_CHECK(gemm_bf16_direct(h->stream, a_mc, b_nc, g));
        return 0;
    }
    if (h->cfg.compute_dtype == 1) {
        // bf16 operands: convert (transposing [k][x] operands) into k-contiguous panels, then one NT kernel
        const int Kp = (K + 7) & ~7;
        unsigned short* a16 = nullptr;
        if (h->keep_a16 && !a_mc && Kp == K) { a16 = h->keep_a16; }      // a layer input: its bf16 copy [M][K] stays for the weight-gradient GEMM of the backward
        h->keep_a16 = nullptr;
        if (!a16) { AV_TRY(grow_bf16(h, &h->bfA, &h->bfA_cap, (size_t)M * Kp)); a16 = h->bfA; }
        AV_TRY(grow_bf16(h, &h->bfB, &h->bfB_cap, (size_t)N * Kp));
        AV_CHECK(cvt_bf16(h->stream, A, lda, a_mc, a_mc ? K : M, a_mc ? M : K, a16, Kp));
        AV_CHECK(cvt_bf16(h->stream, Bm, ldb, b_nc, b_nc ? K : N, b_nc ? N : K, h->bfB, Kp));
        g.nt8 = h->bf16_nt8;
        g.c16 = c16;
        AV_CHECK(gemm_bf16_nt(h->stream, a16, Kp, h->bfB, Kp, g));
        return 0;
    }
    if (c16) return fail(h, "the fp16 output panel exists in compute_dtype 1 only");
    // compute_dtype 2: fp32 operands split into 3 x bf16 on the fly (6 partial products, fp32-accurate); thin
    // row panels (a few rows, little work) stay on the exact-fp32 kernel's 32x128 tiles
    if (h->cfg.compute_dtype == 2 && !thin) AV_CHECK(gemm_f32s(h->stream, a_mc, b_nc, g));
    else AV_CHECK(gemm_f32(h->stream, a_mc, b_nc, g));
    return 0;
}

// bf16 mode, A already bf16 and row-major ((rows, lda16), written by the producer -- the softmax gradient): as the A panel
// itself (a_mc = false: k-contiguous) or transposed once from the 2-byte source (a_mc = true); B converted as usual.
int gemm_bf16_pre(avae_ctx* h, const unsigned short* A16, int lda16, bool a_mc, const float* Bm, int ldb, bool b_nc, float* C, int ldc,
                  int M, int N, int K, float alpha, int accumulate, int split_k, const int* dyn, int dyn_kind, const float* bias = nullptr)
{
    GemmArgs g{nullptr, Bm, C, bias, M, N, K, lda16, ldb, ldc, alpha, accumulate, split_k, dyn, dyn_kind, 0, 0, nullptr, nullptr, nullptr, nullptr};
    Timed t(h, 0, 2.0 * M * N * K, dyn, dyn_kind == 1 ? M : (dyn_kind == 2 ? K : 0));
    const int Kp = (K + 7) & ~7;
    const unsigned short* Ap = A16; int lda_p = lda16;
    if (a_mc) {
        AV_TRY(grow_bf16(h, &h->bfA, &h->bfA_cap, (size_t)M * Kp));
        AV_CHECK(transpose_bf16(h->stream, A16, lda16, K, M, h->bfA, Kp));
        Ap = h->bfA; lda_p = Kp;
    }
    AV_TRY(grow_bf16(h, &h->bfB, &h->bfB_cap, (size_t)N * Kp));
    AV_CHECK(cvt_bf16(h->stream, Bm, ldb, b_nc, b_nc ? K : N, b_nc ? N : K, h->bfB, Kp));
    g.nt8 = h->bf16_nt8;
    AV_CHECK(gemm_bf16_nt(h->stream, Ap, lda_p, h->bfB, Kp, g));
    return 0;
}

int grad_split(int M, int N, int K);
// bf16 mode, weight gradient C (M x N) += alpha * A^T B over K rows with BOTH operands row-major [k][x]: bf16 as a producer
// wrote them (A16 / B16) or fp32 converted row by row (no transpose); the GEMM reads them through transposing LDS loads
// (gemm_bf16_tn).  C holds the zero-filled gradient; dynk: device-side K.
int gemm_tn16(avae_ctx* h, const unsigned short* A16, const float* A32, int lda, const unsigned short* B16, const float* B32, int ldb,
              float* C, int ldc, int M, int N, int K, float alpha, const int* dynk)
{
    const int s = grad_split(M, N, K);
    GemmArgs g{nullptr, nullptr, C, nullptr, M, N, K, lda, ldb, ldc, alpha, s > 1 ? 0 : 1, s, dynk, dynk ? 2 : 0, 0, 0, nullptr, nullptr, nullptr, nullptr};
    Timed t(h, 0, 2.0 * M * N * K, dynk, dynk ? K : 0);
    int la = lda, lb = ldb;
    if (!A16) {
        la = (M + 7) & ~7;
        AV_TRY(grow_bf16(h, &h->bfA, &h->bfA_cap, (size_t)K * la));
        AV_CHECK(cvt_bf16(h->stream, A32, lda, false, K, M, h->bfA, la));
        A16 = h->bfA;
    }
    if (!B16) {
        lb = (N + 7) & ~7;
        AV_TRY(grow_bf16(h, &h->bfB, &h->bfB_cap, (size_t)K * lb));
        AV_CHECK(cvt_bf16(h->stream, B32, ldb, false, K, N, h->bfB, lb));
        B16 = h->bfB;
    }
    g.nt8 = h->bf16_nt8;
    if (h->bf16_nt8 && s > 1) {
        if (!h->slab) {
            const size_t n = (size_t)512 << 16;
            AV_CHECK(hipMalloc(reinterpret_cast<void**>(&h->slab), n * sizeof(float)));
            h->slab_floats = n;
        }
        g.slab = h->slab; g.slab_floats = h->slab_floats;
    }
    AV_CHECK(gemm_bf16_tn(h->stream, A16, la, B16, lb, g));
    return 0;
}
static bool tn16_ok(const avae_ctx* h, int M, int N) { return h->cfg.compute_dtype == 1 && h->bf16_tn && ((M | N) & 7) == 0 && (size_t)M * N >= (size_t)1 << 19; }

// C = alpha * op(A) op(B) (+bias) with launch shaping for the 256-CU chip (k-contiguous A only):
//  * thin outputs (M <= 512): 32x128 block tiles so that the few rows still spread over many CUs;
//    where float atomics are acceptable (backward) a long K is split over ~768 workgroups instead;
//  * a tile count just above a multiple of 256 (M = 65*256 rows -> 130 row tiles): the rows that make
//    whole rounds of 256 tiles run as one launch and the thin remainder as 32x128 tiles, instead of a
//    few CUs carrying an extra full tile while the rest idle.
// Both forward forms are deterministic (no atomics): z and the per-token losses stay bit-reproducible.
int gemm(avae_ctx* h, bool a_mc, bool b_nc, const float* A, int lda, const float* Bm, int ldb, float* C, int ldc,
         int M, int N, int K, float alpha = 1.f, const float* bias = nullptr, int accumulate = 0, int split_k = 0,
         const int* dyn = nullptr, int dyn_kind = 0, bool allow_atomic = false)
{
    const int mt = (M + 127) / 128, nt = (N + 127) / 128, tiles = mt * nt;
    unsigned short* const keep = h->keep_a16;           // (bf16 mode: where the converted A operand is to stay, see gemm_raw)
    if (split_k != 0 || a_mc || dyn_kind == 2)
        return gemm_raw(h, a_mc, b_nc, A, lda, Bm, ldb, C, ldc, M, N, K, alpha, bias, accumulate, split_k ? split_k : 1, dyn, dyn_kind);
    // The skinny form (gemm_f32.hip: one 32x32 tile per workgroup, K split over its waves) sums K in another order than the tiled
    // kernels, so WHICH form a forward product takes must not depend on the batch: z and the per-token losses of a row are the
    // same bits in a batch of 16 and of 1024 (test_large_batch_rows_are_independent).  Forward: only the call sites whose rows
    // are the batch rows themselves ask for it (rows_form), for every batch size.  Backward (allow_atomic: the gradients carry
    // float-atomic order anyway): a few rows over a moderate K take it instead of a zero fill + split-K atomics.
    const bool rows_form = h->rows_form != 0; h->rows_form = 0;
    if (rows_form && h->skinny && h->cfg.compute_dtype != 1 && !a_mc && split_k == 0 && dyn_kind == 0)
        return gemm_raw(h, a_mc, b_nc, A, lda, Bm, ldb, C, ldc, M, N, K, alpha, bias, accumulate, 1, nullptr, 0, 3);
    // Ragged batches (compact layout: the host knows roughly how many rows are real): a backward GEMM with a narrow output whose real rows
    // make fewer 128x128 tiles than the chip has CUs (dho = dlogits E over 7.4 k of 16.6 k token rows: 232 tiles, ONE workgroup per CU,
    // 105 TFLOP/s) splits K over ~768 workgroups instead (float atomics into the zero-filled output: the gradients carry that order anyway).
    // The expectation only shapes the launch; rows beyond the device-side count are never touched either way.
    if (allow_atomic && h->dyn_split && dyn_kind == 1 && !accumulate && ldc == N && h->cfg.compute_dtype == 0 && K >= 1536) {
        const int expect = dyn_expected(h, dyn, dyn_kind);
        if (expect > 0) {
            const int eff_tiles = ((expect + 127) / 128) * nt;
            const int sk = std::min(768 / std::max(eff_tiles, 1), K / 512);
            if (eff_tiles <= 320 && sk >= 2) {
                AV_CHECK(zero_rows_dyn(h->stream, C, dyn, M, N));
                return gemm_raw(h, a_mc, b_nc, A, lda, Bm, ldb, C, ldc, M, N, K, alpha, bias, 0, sk, dyn, dyn_kind);
            }
        }
    }
    const bool prefer_skinny = allow_atomic && h->skinny && M <= 512 && K <= 2048 && h->cfg.compute_dtype != 1;
    const int thin_form = (allow_atomic && h->skinny) ? 3 : 1;
    if (tiles <= 96) {
        if (allow_atomic && !accumulate && ldc == N && K >= 512 && !prefer_skinny) {
            int s = std::min(768 / tiles, K / 128);
            if (s >= 2) {
                AV_CHECK(zero_fill(h->stream, C, sizeof(float) * (size_t)M * N));
                return gemm_raw(h, a_mc, b_nc, A, lda, Bm, ldb, C, ldc, M, N, K, alpha, bias, 0, s, dyn, dyn_kind);
            }
        }
        if (M <= 512)
            return gemm_raw(h, a_mc, b_nc, A, lda, Bm, ldb, C, ldc, M, N, K, alpha, bias, accumulate, 1, dyn, dyn_kind, thin_form);      // (3: the skinny form where it applies, else 32x128 tiles)
    }
    // A GEMM whose row count is only known on the device (the ids present in the batch: about V / 2 of the static bound
    // of V rows) with a narrow output: 128x128 tiles over the rows that exist are fewer than one round of the chip (dE of
    // the table-fed layers: 112 tiles for 768 slots, 49 TFLOP/s).  64x64 tiles: four times the tiles, deterministic.
    if (h->dyn_thin && dyn_kind == 1 && h->cfg.compute_dtype == 0 && nt <= 4 && tiles <= 512 && K >= 1024) {
        // (backward -- dE of the present ids, K = 3D or 6D: the K range split over 2-4 slices as well, float atomics into the zeroed rows:
        //  3 584 x 512 x 3 072: 113 -> 96 us; fewer present ids, a ragged batch: more)
        const int sk = std::min(4, K / 768);
        if (allow_atomic && h->dyn_split && !accumulate && ldc == N && sk >= 2) {
            AV_CHECK(zero_rows_dyn(h->stream, C, dyn, M, N));
            return gemm_raw(h, a_mc, b_nc, A, lda, Bm, ldb, C, ldc, M, N, K, alpha, bias, 0, sk, dyn, dyn_kind, 2);
        }
        return gemm_raw(h, a_mc, b_nc, A, lda, Bm, ldb, C, ldc, M, N, K, alpha, bias, accumulate, 1, dyn, dyn_kind, 2);
    }
    // The forward projection of the present target ids (static bound V rows x 3D: 768 tiles of 128x128, about 40 % of them real): 32x128
    // tiles fill the chip with the rows that exist (78 -> 54 us; the encoder's, 1536 static tiles, is faster on 128x128).  Any tile
    // form keeps a row's K order: the bits of gi do not move.
    if (h->dyn_thin && dyn_kind == 1 && h->cfg.compute_dtype == 0 && tiles <= 768 && nt > 4 && K <= 512 && !accumulate)
        return gemm_raw(h, a_mc, b_nc, A, lda, Bm, ldb, C, ldc, M, N, K, alpha, bias, accumulate, 1, dyn, dyn_kind, 1);
    if (tiles > 256 && tiles % 256 != 0) {
        int main_mt = mt;
        while (main_mt > 0 && (main_mt * nt) % 256 != 0) --main_mt;
        const int tail_tiles = (mt - main_mt) * nt;
        const double frac = (double)tiles / 256.0;
        if (main_mt > 0 && tail_tiles < 200 && (std::ceil(frac) - frac) >= 0.3) {
            const int main_rows = main_mt * 128, tail_rows = M - main_rows;
            AV_TRY(gemm_raw(h, a_mc, b_nc, A, lda, Bm, ldb, C, ldc, main_rows, N, K, alpha, bias, accumulate, 1, dyn, dyn_kind));
            // rows beyond the device-side row count hold unread garbage either way: the tail keeps the static bound
            const float* At = A + (size_t)main_rows * lda; float* Ct = C + (size_t)main_rows * ldc;
            if (keep) h->keep_a16 = keep + (size_t)main_rows * K;
            if (allow_atomic && !accumulate && ldc == N && K >= 1024 && !(h->skinny && K <= 2048 && h->cfg.compute_dtype != 1)) {
                // backward only: a few rows x a long K (dho: 256 rows x K = 8192 took 0.2 ms on 32 thin tiles):
                // K split over ~768 workgroups of full tiles with float atomics instead
                const int s = std::min(768 / (((tail_rows + 127) / 128) * nt), K / 128);
                if (s >= 2) {
                    AV_CHECK(zero_fill(h->stream, Ct, sizeof(float) * (size_t)tail_rows * N));
                    return gemm_raw(h, a_mc, b_nc, At, lda, Bm, ldb, Ct, ldc, tail_rows, N, K, alpha, bias, 0, s, nullptr, 0);
                }
            }
            return gemm_raw(h, a_mc, b_nc, At, lda, Bm, ldb, Ct, ldc, tail_rows, N, K, alpha, bias, accumulate, 1, nullptr, 0, thin_form);
        }
    }
    return gemm_raw(h, a_mc, b_nc, A, lda, Bm, ldb, C, ldc, M, N, K, alpha, bias, accumulate, 1, dyn, dyn_kind);
}
// K-split so that a weight-gradient GEMM (few output tiles, very long K) fills the chip: aim at 768
// co-resident workgroups (3 per CU), every slice at least 4 K-tiles deep
int grad_split(int M, int N, int K)
{
    int tiles = ((M + 127) / 128) * ((N + 127) / 128);
    int s = 768 / tiles;
    int kmax = K / 128; if (kmax < 1) kmax = 1;
    if (s > kmax) s = kmax;
    return s < 1 ? 1 : s;
}
// dW (M x N) += A^T B over K rows; A [k][m] lda, B [k][n] ldb.  grads are zero-filled beforehand.
int gemm_tn_grad(avae_ctx* h, const float* A, int lda, const float* Bm, int ldb, float* C, int ldc, int M, int N, int K,
                 float alpha = 1.f, const int* dynk = nullptr, const Pair* pair = nullptr)
{
    if (h->cfg.compute_dtype != 0) {
        const int s = grad_split(M, N, K);
        return gemm_raw(h, true, true, A, lda, Bm, ldb, C, ldc, M, N, K, alpha, nullptr, s > 1 ? 0 : 1, s, dynk, dynk ? 2 : 0, 0, pair);
    }
    // exact-fp32 kernel.  Few output tiles over a very long K: the K split's float atomics (~1.3 TB/s chip-wide, all
    // workgroups at once at the end of one synchronous round) are the overhead, and they scale with tile bytes x
    // slices.  64x64 tiles give four times the tiles, so a quarter of the slices fill the chip (decoder dW/dR pair,
    // the two directions' dR: +6 %, gpurun_out/ab8.log); ~1536 workgroups = 6 per CU.  A pair shares them.
    // A small output (decode/out/kernel, latent) uses 32x128 tiles.
    const int np = pair ? 2 : 1;
    const int t128 = ((M + 127) / 128) * ((N + 127) / 128);
    int thin = 2, tiles = ((M + 63) / 64) * ((N + 63) / 64) * np, target = 1536;
    if (t128 * np > 96) { thin = 0; tiles = t128 * np; target = 768; }      // enough full tiles: 128x128 (measured: 64x64 loses 3-8 % there)
    if (t128 <= 16) { thin = 1; tiles = ((M + 31) / 32) * ((N + 127) / 128) * np; target = 768; }
    int s = (target + tiles / 2) / tiles;
    s = std::max(1, std::min(s, std::max(1, K / 128)));
    // (gradients were zero-filled: one slice may store, several add)
    return gemm_raw(h, true, true, A, lda, Bm, ldb, C, ldc, M, N, K, alpha, nullptr, 0, s, dynk, dynk ? 2 : 0, thin, pair);
}

void gru_geometry(int D, int njobs, int B, int* G, int* rpg)
{
    int HT = D / 16;
    int gmax = 512 / (njobs * HT); if (gmax < 1) gmax = 1; if (gmax > 16) gmax = 16;
    int g = (B + 15) / 16; if (g > gmax) g = gmax; if (g < 1) g = 1;
    int r = (B + g - 1) / g; r = (r + 15) / 16 * 16;
    g = (B + r - 1) / r;
    *G = g; *rpg = r;
}

struct Sched { float keepwd, anneal, lr; };
Sched schedule(const avae_ctx* h)
{
    // src/model.py:77-80, float32 like the TF graph
    float rate = h->cfg.accelerate * (float)h->step;
    Sched s;
    s.keepwd = 1.f / (1.f + expf(-rate));
    s.anneal = tanhf(rate);
    s.lr = h->cfg.learn_rate / (sqrtf(rate) + 1.f);
    return s;
}

// The top encoder layer's backward direction is consumed at ONE position only -- the pick at len_b - 1 (model.py:135), the
// first step of the reversed sequence, from h = 0: its other S - 1 steps, their input projection and their whole BPTT are
// dead in the reference's graph (zero gradient; dR of that direction is exactly 0, oracle fixtures gnorm/encode/rnnL/bwd/R).
// The build computes that one step for B rows (gru_first_step_*) and runs the layer's GRU launches with the forward
// direction alone.  Same values as the full form (option enc_top1 = 0), a sixth of the encoder's work not executed.
static bool top_one_step(const avae_ctx* h) { return h->enc_top1 && h->cfg.rnn_layers >= 2; }

// The share of the padded source positions that are real, as the host last saw it (the asynchronous hint of build_row_orders;
// 1.0 while nothing has arrived).  Shapes launches only, never a value.
static double expected_fill(const avae_ctx* h)
{
    const uint64_t pair = h->hint_host ? *reinterpret_cast<const volatile uint64_t*>(h->hint_host) : 0xffffffffull;
    const int32_t real = (int32_t)(uint32_t)(pair & 0xffffffffull), rows = (int32_t)(uint32_t)(pair >> 32);
    return (real > 0 && rows > 0) ? std::min(1.0, (double)real / (double)rows) : 1.0;
}
// Which BPTT team kernel a launch takes (option bwd_rs = 2).  The reduce-scatter form stores 64 KB per live row and step where the other
// form stores 6 KB, and wins only where a workgroup's teams run out of rows at different times so that most steps belong to one or two
// lone chains -- a ragged batch with a long tail.  Measured: batch 100 x 512 ragged (fill 0.30) -8 % of the step; RAGGED 256 x 64 (fill
// 0.44) a tie; FULL batches lose at every size (64 x 64 +1.8 %, 128 x 64 +4.7 %, 100 x 512 +2.3 %, 256 x 64 +5 %).  So the fill decides,
// not the row count.
static int rs_pick(const avae_ctx* h, int njobs, int B)
{
    (void)njobs; (void)B;
    if (h->bwd_rs != 2) return h->bwd_rs;
    return expected_fill(h) < 0.40 ? 1 : 0;
}
// the same regime in the exchange of every team kernel (option gru_spec = 2): a consumer's first operand load goes out without a probe
// round trip in front of it (GruArgs::spec): RAGGED 256 x 64 (fill 0.44) -1.5 %; forced on, a FULL 100 x 512 batch loses 1 %.
static int spec_pick(const avae_ctx* h, int njobs, int B)
{
    (void)njobs; (void)B;
    if (h->gru_spec != 2) return h->gru_spec;
    return expected_fill(h) < 0.60 ? 1 : 0;
}

// -------------------------------------------------------------------------------- forward pieces
// GRU launch arguments common to every call site
static void gru_common(avae_ctx* h, const Ws& w, GruArgs& a, int njobs, int S, int B, int ldg, int ldh, const int32_t* lens)
{
    const int D = h->cfg.dim_emb;
    a.njobs = njobs; a.S = S; a.B = B; a.D = D; a.ldg = ldg; a.ldh = ldh; a.lens = lens;
    a.Bx = w.Bx;        // (shape queries: the geometry the team kernels would run with the row order and the compact layout in place)
    gru_geometry(D, njobs, B, &a.G, &a.rows_per_group);
    a.p_begin = 0; a.p_end = S; a.counters = h->counters; a.err = h->errw; a.ablate = h->gru_ablate; a.force_slow = h->gru_force_slow;
    a.bf16 = h->cfg.compute_dtype == 1 && h->gru_bf16; a.stagger = h->gru_stagger; a.item_pipeline = h->gru_item;
    a.xbuf = w.xbuf; a.xbuf_floats = w.xbuf_floats; a.stamps = reinterpret_cast<unsigned long long*>(h->errw + 16); a.bwd_rs = h->bwd_rs != 0;
}
// Row orders of this call (one launch, after prep_ids has the lengths): for every GRU launch shape of the step whose team
// kernels can skip padding, the batch rows sorted by length and dealt over the workgroups.  with_dec: the decoder runs too.
int build_row_orders(avae_ctx* h, Ws& w, int B, int Ss, int T, bool with_dec)
{
    if (!h->skip_pad || !h->persistent || w.Bx % 16) return 0;
    if (w.Bx != B && !h->compact) return 0;                   // (phantom rows need the compact layout)
    const int D = h->cfg.dim_emb;
    RowOrder ord[3]; int n = 0, which[3];
    auto want = [&](int k, int njobs, int S, int ldg, int ldh, const int32_t* lens, int add) {
        GruArgs a{};
        gru_common(h, w, a, njobs, S, B, ldg, ldh, lens);
        int Tm = 0, cpj = 0, nrb = 0;
        if (S < 2 || !gru_team_shape(a, true, true, &Tm, &cpj, &nrb)) return;
        ord[n] = RowOrder{lens, add, Tm, cpj, w.ord_perm[k], w.ord_slens[k]};
        which[n++] = k; w.ord_T[k] = Tm; w.ord_cpj[k] = cpj;
    };
    want(0, 2, Ss, 6 * D, 2 * D, w.lens_src, 0);          // (every layer but a one-step top layer carries both directions)
    if (top_one_step(h)) want(1, 1, Ss, 6 * D, 2 * D, w.lens_src, 0);
    if (with_dec) want(2, 1, T, 3 * D, D, w.lens_tgt, 1);
    if (!n) return 0;
    const bool hint = which[0] == 0 && h->compact == 2;        // (order 0 sorts the source rows: its step sum = the real source positions)
    if (hint && !h->hint_host) {
        int32_t* hp = nullptr;
        if (hipHostMalloc(reinterpret_cast<void**>(&hp), 64, hipHostMallocDefault) == hipSuccess) { hp[0] = -1; hp[1] = 0; h->hint_host = hp; }
    }
    h->hint_dev = reinterpret_cast<int32_t*>(h->errw + 100);     // (spare words of the error block)
    hipError_t e = row_order(h->stream, ord, n, B, w.Bx, std::max(Ss, T), hint ? h->hint_dev : nullptr, Ss * B);
    if (e == hipErrorInvalidValue) return 0;                 // (a batch beyond the kernel's LDS: no order, every step runs)
    AV_CHECK(e);
    if (hint && h->hint_host) {
        AV_CHECK(hipMemcpyAsync(const_cast<int32_t*>(h->hint_host), h->hint_dev, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    }
    for (int i = 0; i < n; ++i) w.ord_ok[which[i]] = true;
    return 0;
}
// bf16 mode: does a layer's forward AND backward launch both run the bf16 team kernels?  Then what passes between them (saved
// gates, h_prev) and on to the GEMMs (h, gate gradients) can be 16-bit.
static bool both_team_bf16(avae_ctx* h, const GruArgs& a, bool train)
{
    int T = 0, cpj = 0, nrb = 0;
    GruArgs q = a; q.sv16 = 0; q.slens = nullptr; q.perm = nullptr;
    for (int i = 0; i < q.njobs; ++i) { q.job[i].dgi16 = nullptr; q.job[i].dgh16 = nullptr; q.job[i].gi_rows = nullptr; q.job[i].hs16 = nullptr; q.job[i].hp16 = nullptr; }
    q.p_begin = 0; q.p_end = q.S;
    return train && a.bf16 && a.S > 1 && gru_team_shape(q, true, h->persistent != 0, &T, &cpj, &nrb) && gru_team_shape(q, false, h->persistent != 0, &T, &cpj, &nrb);
}
static void attach_sv16(avae_ctx* h, GruArgs& a, bool train) { a.sv16 = h->bf16_sv && both_team_bf16(h, a, train); }
static void attach_order(avae_ctx* h, const Ws& w, GruArgs& a, bool fwd, int k)
{
    int T = 0, cpj = 0, nrb = 0;
    if (!w.ord_ok[k] || !gru_team_shape(a, fwd, h->persistent != 0, &T, &cpj, &nrb) || T != w.ord_T[k] || cpj != w.ord_cpj[k]) return;
    a.slens = w.ord_slens[k]; a.perm = w.ord_perm[k];
}

// Compact layout of this call, decided for the encoder stack and, with T > 1, for the decoder stack: possible where
// every GRU launch of the stack runs the team kernels (they address the external arrays through GruArgs::rowmap) and its first
// layer is table-fed (its per-token arrays keep the padded order).  Taken when the fill hint says the batch is ragged, and
// always for a batch without a team-kernel geometry of its own, which reaches the team kernels through it (Ws::Bx).
int build_compact(avae_ctx* h, Ws& w, int B, int Ss, int T, bool train)
{
    w.compact = false; w.compact_d = false;
    h->expect_ptr[0] = h->expect_ptr[1] = h->expect_ptr[2] = nullptr;
    if (!h->compact || !h->persistent || Ss < 2 || !use_table(h, Ss * B, B)) return 0;
    const bool phantom = w.Bx != B;        // a batch without a team geometry of its own: the compact layout is what lets it run the team kernels at all
    if (phantom && !(w.ord_ok[0] && (w.ord_ok[1] || !top_one_step(h)))) return 0;
    if (h->compact == 2 && !phantom) {
        // auto: the layout pays where a good share of the padded positions is padding; on FULL batches the static row counts shape
        // the GEMM launches a little better (16.48 vs 16.71 ms at configs[1]; break-even at a fill of 0.93).  The hint is an EARLIER call's count (no synchronisation).
        // (both words in ONE 8-byte load: the copy that lands them is 8 bytes, so a pair is never half of one call and half of another)
        const uint64_t pair = h->hint_host ? *reinterpret_cast<const volatile uint64_t*>(h->hint_host) : 0xffffffffull;
        const int32_t real = (int32_t)(uint32_t)(pair & 0xffffffffull), rows = (int32_t)(uint32_t)(pair >> 32);
        if (real < 0 || rows <= 0 || (double)real >= 0.92 * (double)rows) return 0;
    }
    const int D = h->cfg.dim_emb;
    for (int njobs = 1; njobs <= 2; ++njobs) {
        if (njobs == 1 && !top_one_step(h)) continue;
        GruArgs q{};
        gru_common(h, w, q, njobs, Ss, B, 6 * D, 2 * D, w.lens_src);
        int T = 0, cpj = 0, nrb = 0;
        if (!gru_team_shape(q, true, true, &T, &cpj, &nrb) || (train && !gru_team_shape(q, false, true, &T, &cpj, &nrb))) return 0;
    }
    AV_CHECK(row_map(h->stream, w.lens_src, 0, Ss, B, w.map_src, w.nact_src, w.nsrc));
    w.compact = true;
    {   // the fill the host last saw, scaled to this call's rows: shapes the launches of the GEMMs over the compact rows
        const uint64_t pair = h->hint_host ? *reinterpret_cast<const volatile uint64_t*>(h->hint_host) : 0xffffffffull;
        const int32_t real = (int32_t)(uint32_t)(pair & 0xffffffffull), rows = (int32_t)(uint32_t)(pair >> 32);
        const double fill = (real > 0 && rows > 0) ? std::min(1.0, (double)real / (double)rows) : 0.0;
        h->expect_ptr[0] = w.nsrc; h->expect_val[0] = (int)(fill * Ss * B);
        h->expect_ptr[1] = w.ntgt; h->expect_val[1] = (int)(fill * T * B);
        h->expect_ptr[2] = w.ntok; h->expect_val[2] = (int)(fill * T * B);      // (the unmasked decoder tokens: the same share of T x B for prefix masks)
    }
    // the decoder stack the same way (training / evaluation calls: T > 1): a row's steps end one behind its last non-eos target id
    if (T < 2 || !use_table(h, T * B, B) || h->compact == 3) return 0;      // (3: the encoder alone, for measurements)
    if (phantom && !w.ord_ok[2]) return 0;
    GruArgs q{};
    gru_common(h, w, q, 1, T, B, 3 * D, D, nullptr);
    int Tm = 0, cpj = 0, nrb = 0;
    if (!gru_team_shape(q, true, true, &Tm, &cpj, &nrb) || (train && !gru_team_shape(q, false, true, &Tm, &cpj, &nrb))) return 0;
    AV_CHECK(row_map(h->stream, w.lens_tgt, 1, T, B, w.map_tgt, w.nact_tgt, w.ntgt));
    w.compact_d = true;
    return 0;
}

int run_encoder(avae_ctx* h, Ws& w, int B, int Ss, bool save)
{
    const int D = h->cfg.dim_emb, V = h->cfg.dim_tgt, L = h->cfg.rnn_layers;
    const int rs = Ss * B;
    const bool table = use_table(h, rs, B);
    h->cnt_src = table ? id_groups_count(w.grp_src, rs, V) : nullptr;
    // compact layout (build_compact): every array between the GEMMs and the GRU launches holds the real rows only; the GEMMs over
    // them take the device-side row count
    const int32_t* const cdyn = w.compact ? w.nsrc : nullptr;
    const int32_t* const cmap = w.compact ? w.map_src : nullptr;
    if (!table) AV_CHECK(embed_gather(h->stream, h->P + h->oE, w.src_tm, w.emb_src, rs, D, V));
    const float* x = w.emb_src; int In = D;
    for (int i = 0; i < L; ++i) {
        const GruP& p = h->enc[i];
        const bool top1 = i == L - 1 && top_one_step(h);      // forward direction only; the backward direction's one live step follows the loop
        if (i == 0 && table) {
            const int32_t* cnt = id_groups_count(w.grp_src, rs, V);
            AV_CHECK(id_groups_build(h->stream, w.src_tm, rs, V, w.grp_src, save));
            AV_CHECK(rows_gather(h->stream, w.emb_src, h->P + h->oE, id_groups_uid(w.grp_src, rs, V), cnt, std::min(V, rs), D));
            AV_TRY(gemm(h, false, false, w.emb_src, D, h->P + p.W, D, w.ew, 6 * D, std::min(V, rs), 6 * D, D, 1.f, h->P + p.bW, 0, 0, cnt, 1));
        } else if (i > 0 && w.act_e[i - 1]) {       // the layer below wrote its output as bf16: the A operand as it stands
            AV_TRY(gemm_bf16_pre(h, w.e_hs16[i - 1], In, false, h->P + p.W, In, false, w.e_gi[i], 6 * D, rs, top1 ? 3 * D : 6 * D, In, 1.f, 0, 1, cdyn, cdyn ? 1 : 0, h->P + p.bW));
        } else {
        h->keep_a16 = save ? w.x16_e[i] : nullptr;
        AV_TRY(gemm(h, false, false, x, In, h->P + p.W, In, w.e_gi[i], 6 * D, rs, top1 ? 3 * D : 6 * D, In, 1.f, h->P + p.bW, 0, 0, cdyn, cdyn ? 1 : 0));
        h->keep_a16 = nullptr;
        }
        GruArgs a{};
        a.njobs = top1 ? 1 : 2; a.S = Ss; a.B = B; a.D = D; a.ldg = 6 * D; a.ldh = 2 * D; a.lens = w.lens_src;
        a.Bx = w.bx_enc();
        gru_geometry(D, a.njobs, B, &a.G, &a.rows_per_group);
        a.p_begin = 0; a.p_end = Ss; a.counters = h->counters; a.err = h->errw; a.ablate = h->gru_ablate; a.force_slow = h->gru_force_slow; a.bf16 = h->cfg.compute_dtype == 1 && h->gru_bf16; a.stagger = h->gru_stagger; a.item_pipeline = h->gru_item; a.xbuf = w.xbuf; a.xbuf_floats = w.xbuf_floats; a.stamps = reinterpret_cast<unsigned long long*>(h->errw + 16); a.bwd_rs = h->bwd_rs != 0;
        // table-fed layer: the team kernels read gi straight out of the per-id projection through a row index per token;
        // the other kernel forms get a per-token copy
        const bool table0 = i == 0 && table, indirect = table0 && gru_forward_uses_team(a, h->persistent != 0);
        if (indirect) AV_CHECK(rank_rows(h->stream, w.tokrow_src, w.src_tm, id_groups_rank(w.grp_src, rs, V), rs, V));
        else if (table0) AV_CHECK(rows_gather_ranked(h->stream, w.e_gi[0], w.ew, w.src_tm, id_groups_rank(w.grp_src, rs, V), rs, 6 * D, V));
        for (int d = 0; d < a.njobs; ++d) {
            GruJob& j = a.job[d];
            j.gi = (indirect ? w.ew : w.e_gi[i]) + d * 3 * D;
            j.gi_rows = indirect ? w.tokrow_src : nullptr;
            j.R = h->P + p.R + (int64_t)d * 3 * D * D;
            j.bR = h->P + p.bR + d * 3 * D;
            j.h0 = nullptr;
            j.hs = w.e_hs[i] + d * D;
            j.sv = save ? w.e_sv[d][i] : nullptr;
            j.hp = save ? w.e_hp[d][i] : nullptr;
            j.reverse = d;
        }
        attach_sv16(h, a, save);
        if (w.e_hs16[i] && tn16_ok(h, 3 * D, D) && both_team_bf16(h, a, save)) {      // h / h_prev as bf16 (a table-fed layer keeps h_prev fp32: its dR runs the fp32-operand path)
            w.act_e[i] = 1; w.acth_e[i] = 1;
            for (int d = 0; d < a.njobs; ++d) { a.job[d].hs16 = w.e_hs16[i] + d * D; a.job[d].hp16 = w.e_hp16[d][i]; }
        }
        attach_order(h, w, a, true, top1 ? 1 : 0);
        a.rowmap = cmap;
        a.spec = spec_pick(h, a.njobs, B);
        { Timed t(h, 1, 2.0 * a.njobs * Ss * (double)B * D * 3 * D);
          DeviceTurn turn(h);
          AV_GRU(gru_forward(h->stream, a, h->persistent != 0)); }
        if (top1) {
            // the backward direction at position len_b - 1: gi = W_b x[len_b - 1] + bW_b for B rows, then one cell step from h = 0
            const int64_t oWb = p.W + (int64_t)3 * D * In;
            if (w.act_e[i - 1]) AV_CHECK(pick_last16(h->stream, w.xlast, w.e_hs16[i - 1], w.lens_src, B, In, cmap));
            else
            AV_CHECK(pick_last(h->stream, w.xlast, x, w.lens_src, B, In, cmap));
            h->rows_form = 1;
            AV_TRY(gemm(h, false, false, w.xlast, In, h->P + oWb, In, w.gib, 3 * D, B, 3 * D, In, 1.f, h->P + p.bW + 3 * D));
        }
        x = w.e_hs[i]; In = 2 * D;
    }
    if (w.act_e[L - 1]) AV_CHECK(pick_last16(h->stream, w.hpick, w.e_hs16[L - 1], w.lens_src, B, 2 * D, cmap));
    else
    AV_CHECK(pick_last(h->stream, w.hpick, w.e_hs[L - 1], w.lens_src, B, 2 * D, cmap));
    if (top_one_step(h))      // (the pick copied the never-written backward half of the top layer's rows: overwritten here)
        AV_CHECK(gru_first_step_fwd(h->stream, w.gib, h->P + h->enc[L - 1].bR + 3 * D, w.hpick + D, 2 * D, save ? w.svb : nullptr, B, D, w.lens_src));
    return 0;
}

int run_latent(avae_ctx* h, Ws& w, int B, bool train, uint64_t seed, const float* eps)
{
    const int D = h->cfg.dim_emb, R = h->cfg.dim_rep;
    {   // mu and lv (model.py:149-150): two affines of the same input, one launch
        const Pair lv{w.hpick, h->P + h->oWlv, w.lv, h->P + h->oBlv};
        AV_TRY(gemm_raw(h, false, true, w.hpick, 2 * D, h->P + h->oWmu, R, w.mu, R, B, R, 2 * D, 1.f, h->P + h->oBmu, 0, 1, nullptr, 0, h->skinny ? 3 : (B <= 512 ? 1 : 0), &lv));      // (rows = the batch rows: the skinny form for every batch size)
    }
    AV_CHECK(latent_fwd(h->stream, w.mu, w.lv, eps, w.eps, w.z, w.kld, B * R, train ? 1 : 0, seed, h->cfg.free_bits, nullptr));      // (KL scalar: finalize_losses, fixed order)
    return 0;
}

// decoder GRU stack over T steps from per-layer initial states (state stride: layer * B * D; 0 = shared h0)
int run_decoder_rnn(avae_ctx* h, Ws& w, int B, int T, const float* state_in, int64_t state_stride, bool save, const int32_t* ids0 = nullptr, bool compact = false)
{
    // compact layout (build_compact): the arrays between the GEMMs and the GRU launches hold the real rows only
    const int32_t* const cdyn = compact ? w.ntgt : nullptr;
    const int32_t* const cmap = compact ? w.map_tgt : nullptr;
    const int D = h->cfg.dim_emb, L = h->cfg.rnn_layers;
    const int rt = T * B;
    const float* x = w.emb_tgt;
    h->cnt_tgt = ids0 ? id_groups_count(w.grp_tgt, rt, h->cfg.dim_tgt) : nullptr;
    for (int i = 0; i < L; ++i) {
        const GruP& p = h->dec[i];
        if (i == 0 && ids0) {       // (ids0: the layer input is E[ids0], not yet gathered -- use_table decided by the caller)
            const int V = h->cfg.dim_tgt;
            const int32_t* cnt = id_groups_count(w.grp_tgt, rt, V);
            AV_CHECK(id_groups_build(h->stream, ids0, rt, V, w.grp_tgt, save));
            AV_CHECK(rows_gather(h->stream, w.emb_tgt, h->P + h->oE, id_groups_uid(w.grp_tgt, rt, V), cnt, std::min(V, rt), D));
            AV_TRY(gemm(h, false, false, w.emb_tgt, D, h->P + p.W, D, w.ew, 3 * D, std::min(V, rt), 3 * D, D, 1.f, h->P + p.bW, 0, 0, cnt, 1));
        } else if (i > 0 && w.act_d[i - 1]) {
            AV_TRY(gemm_bf16_pre(h, w.d_hd16[i - 1], D, false, h->P + p.W, D, false, w.d_gi[i], 3 * D, rt, 3 * D, D, 1.f, 0, 1, cdyn, cdyn ? 1 : 0, h->P + p.bW));
        } else {
        h->keep_a16 = save ? w.x16_d[i] : nullptr;
        AV_TRY(gemm(h, false, false, x, D, h->P + p.W, D, w.d_gi[i], 3 * D, rt, 3 * D, D, 1.f, h->P + p.bW, 0, 0, cdyn, cdyn ? 1 : 0));
        h->keep_a16 = nullptr;
        }
        GruArgs a{};
        a.njobs = 1; a.S = T; a.B = B; a.D = D; a.ldg = 3 * D; a.ldh = D; a.lens = nullptr;
        a.Bx = compact ? w.Bx : 0;
        gru_geometry(D, 1, B, &a.G, &a.rows_per_group);
        a.p_begin = 0; a.p_end = T; a.counters = h->counters; a.err = h->errw; a.ablate = h->gru_ablate; a.force_slow = h->gru_force_slow; a.bf16 = h->cfg.compute_dtype == 1 && h->gru_bf16; a.stagger = h->gru_stagger; a.item_pipeline = h->gru_item; a.xbuf = w.xbuf; a.xbuf_floats = w.xbuf_floats; a.stamps = reinterpret_cast<unsigned long long*>(h->errw + 16); a.bwd_rs = h->bwd_rs != 0;
        GruJob& j = a.job[0];
        const bool table0 = i == 0 && ids0, indirect = table0 && gru_forward_uses_team(a, h->persistent != 0);
        if (indirect) AV_CHECK(rank_rows(h->stream, w.tokrow_tgt, ids0, id_groups_rank(w.grp_tgt, rt, h->cfg.dim_tgt), rt, h->cfg.dim_tgt));
        else if (table0) AV_CHECK(rows_gather_ranked(h->stream, w.d_gi[0], w.ew, ids0, id_groups_rank(w.grp_tgt, rt, h->cfg.dim_tgt), rt, 3 * D, h->cfg.dim_tgt));
        j.gi = indirect ? w.ew : w.d_gi[i]; j.gi_rows = indirect ? w.tokrow_tgt : nullptr;
        j.R = h->P + p.R; j.bR = h->P + p.bR;
        j.h0 = state_in + state_stride * i;
        j.hs = w.d_hd[i];
        j.sv = save ? w.d_sv[i] : nullptr;
        j.hp = save ? w.d_hp[i] : nullptr;
        j.reverse = 0;
        attach_sv16(h, a, save);
        if (w.d_hp16[i] && tn16_ok(h, 3 * D, D) && both_team_bf16(h, a, save)) {      // (the top layer keeps its fp32 output: the compaction and the out affine's gradient read it)
            if (i < L - 1) { w.act_d[i] = 1; j.hs16 = w.d_hd16[i]; }
            w.acth_d[i] = 1; j.hp16 = w.d_hp16[i];       // (a table-fed layer too: its dR reads bf16 dgh and this, gemm_tn16)
        }
        if (T > 1) attach_order(h, w, a, true, 2);
        a.rowmap = cmap;
        a.spec = spec_pick(h, 1, B);
        { Timed t(h, 1, 2.0 * T * (double)B * D * 3 * D);
          DeviceTurn turn(h);
          AV_GRU(gru_forward(h->stream, a, h->persistent != 0)); }
        x = w.d_hd[i];
    }
    return 0;
}

int forward(avae_ctx* h, Ws& w, const int32_t* src, const int32_t* tgt, int B, int Ss, int St, bool train,
            uint64_t seed, const uint8_t* keep_mask, const float* eps, float inv_n)
{
    const int D = h->cfg.dim_emb, V = h->cfg.dim_tgt, R = h->cfg.dim_rep, L = h->cfg.rnn_layers;
    const int T = St + 1, rt = T * B;
    Sched sc = schedule(h);
    PrepArgs p{};
    p.src = src; p.tgt = tgt; p.B = B; p.Ss = Ss; p.St = St; p.eos = h->cfg.eos; p.bos = h->cfg.bos;
    p.train = train ? 1 : 0; p.keepwd = sc.keepwd; p.seed = seed; p.keep_mask = keep_mask;
    p.src_tm = w.src_tm; p.lens_src = w.lens_src; p.lens_tgt = w.lens_tgt; p.lead = w.lead; p.gold = w.gold;
    p.rank = w.rank; p.cidx = w.cidx; p.ntok = w.ntok; p.chunk_counts = w.ntok + 4; p.zero2 = nullptr;
    AV_CHECK(prep_ids(h->stream, p));
    AV_TRY(build_row_orders(h, w, B, Ss, T, true));
    AV_TRY(build_compact(h, w, B, Ss, T, train));
    AV_TRY(run_encoder(h, w, B, Ss, train));
    AV_TRY(run_latent(h, w, B, train, seed, eps));
    h->rows_form = 1;
    AV_TRY(gemm(h, false, true, w.z, R, h->P + h->oWex, D, w.h0, D, B, D, R, 1.f, h->P + h->oBex));
    if (use_table(h, rt, B)) AV_TRY(run_decoder_rnn(h, w, B, T, w.h0, 0, train, w.lead, w.compact_d));
    else {
        AV_CHECK(embed_gather(h->stream, h->P + h->oE, w.lead, w.emb_tgt, rt, D, V));
        AV_TRY(run_decoder_rnn(h, w, B, T, w.h0, 0, train));
    }
    AV_CHECK(rows_gather(h->stream, w.hc, w.d_hd[L - 1], w.cidx, w.ntok, rt, D, w.compact_d ? w.map_tgt : nullptr));
    AV_TRY(gemm(h, false, true, w.hc, D, h->P + h->oKout, D, w.ho, D, rt, D, D, 1.f, h->P + h->oBout, 0, 0, w.ntok, 1));
    CeArgs c{};
    c.logits = w.logits; c.gold = w.gold; c.cidx = w.cidx; c.n_dev = w.ntok; c.n_max = rt; c.V = V;
    c.write_grad = train ? 1 : 0; c.inv_n = inv_n;
    if (train && h->cfg.compute_dtype == 1 && (V & 7) == 0) {      // bf16 mode: the gradient is written as the backward GEMMs' bf16 operand
        AV_TRY(grow_bf16(h, &h->bfP, &h->bfP_cap, (size_t)rt * V));
        c.grad16 = h->bfP;
        // ... and where the phased GEMM takes the whole product, the logits themselves go THERE as fp16 (2^-12 relative, finer than the bf16
        // gradient they become): no fp32 logits are written or read -- 4.3 GB of 10.8 GB at configs[2]
        GemmArgs probe{nullptr, nullptr, nullptr, nullptr, rt, V, D, D, D, V, 1.f, 0, 1, w.ntok, 1, 0, 0, nullptr, nullptr, nullptr, nullptr};
        probe.nt8 = h->bf16_nt8;
        if (h->logits16 && !h->bf16_direct && (D & 7) == 0 && gemm_bf16_c16_ok(probe, D, D)) c.logits16 = 1;
    }
    if (c.logits16) AV_TRY(gemm_raw(h, false, false, w.ho, D, h->P + h->oE, D, w.logits, V, rt, V, D, 1.f / sqrtf((float)D), nullptr, 0, 1, w.ntok, 1, 0, nullptr, h->bfP));
    else
    AV_TRY(gemm(h, false, false, w.ho, D, h->P + h->oE, D, w.logits, V, rt, V, D, 1.f / sqrtf((float)D), nullptr, 0, 0, w.ntok, 1));
    c.loss_samp = w.loss_samp; c.errt_samp = w.errt_samp; c.pred = w.pred; c.loss_acc = nullptr;      // (the scalar is summed from loss_samp in a fixed order: finalize_losses)
    AV_CHECK(softmax_ce(h->stream, c));
    float beta = h->cfg.kl_beta;
    AV_CHECK(finalize_losses(h->stream, h->losses, w.loss_samp, w.ntok, rt, w.kld, B * R, h->cfg.free_bits, 1.f / ((float)B * R), sc.anneal * beta));
    return 0;
}

// Data-parallel hook protocol (include/argsim_vae.h).  A bucket whose gradients are final is not announced at once
// but right AFTER the next persistent GRU launch has been enqueued (or at the end of backward): a collective the
// callee starts then is ordered behind that launch and overlaps the GEMM phase that follows it.  Just BEFORE every
// persistent GRU launch the hook is called with bucket = AVAE_HOOK_FENCE so that the callee makes the compute stream
// wait for the collectives in flight: a persistent launch needs every CU (its workgroups exchange data inside the
// launch) and must never share the device with a kernel that may wait on a peer GPU.
void fire_hook(avae_ctx* h, int bucket)
{
    if (h->hook && bucket >= 0 && bucket < (int)h->buckets.size()) h->hook_pending.push_back(bucket);
}
void hook_fence(avae_ctx* h)
{
    if (h->hook) h->hook(h->hook_user, AVAE_HOOK_FENCE, 0, 0);
}
void hook_flush(avae_ctx* h)
{
    for (int b : h->hook_pending) h->hook(h->hook_user, b, h->buckets[b].first, h->buckets[b].second);
    h->hook_pending.clear();
}

int backward(avae_ctx* h, Ws& w, int B, int Ss, int St, float b_global)
{
    const int D = h->cfg.dim_emb, V = h->cfg.dim_tgt, R = h->cfg.dim_rep, L = h->cfg.rnn_layers;
    const int T = St + 1, rt = T * B, rs = Ss * B;
    const float isd = 1.f / sqrtf((float)D);
    Sched sc = schedule(h);
    hipStream_t st = h->stream;
    float* G = h->G; const float* P = h->P;
    h->hook_pending.clear();
    AV_CHECK(zero_fill(st, G, sizeof(float) * h->numel));

    // logits: dho = dlogits E / sqrt(D);  dE = dlogits^T ho / sqrt(D)
    if (h->cfg.compute_dtype == 1 && (V & 7) == 0) {
        // bf16 mode: softmax_ce_kernel left the gradient as bf16 (h->bfP): the first GEMM reads it as its A panel, the
        // second transposes the 2-byte source once -- no fp32 gradient is written or converted (10.7 GB less traffic per
        // step at configs[2]); same values as rounding the fp32 gradient, so the results do not change
        AV_TRY(gemm_bf16_pre(h, h->bfP, V, false, P + h->oE, D, true, w.dho, D, rt, D, V, isd, 0, 1, w.ntok, 1));
        if (tn16_ok(h, V, D)) AV_TRY(gemm_tn16(h, h->bfP, nullptr, V, nullptr, w.ho, D, G + h->oE, D, V, D, rt, isd, w.ntok));      // (no transposed copy of the 2-byte gradient)
        else {
        const int s = grad_split(V, D, rt);
        AV_TRY(gemm_bf16_pre(h, h->bfP, V, true, w.ho, D, true, G + h->oE, D, V, D, rt, isd, s > 1 ? 0 : 1, s, w.ntok, 2));
        }
    } else {
    AV_TRY(gemm(h, false, true, w.logits, V, P + h->oE, D, w.dho, D, rt, D, V, isd, nullptr, 0, 0, w.ntok, 1, true));
    // (V x D output over K = N rows: 256 tiles of 128x128 x 3 K slices, float atomics into the zero-filled G; the gather
    //  part is scatter-added at the end)
    AV_TRY(gemm_tn_grad(h, w.logits, V, w.ho, D, G + h->oE, D, V, D, rt, isd, w.ntok));
    }
    // out affine
    AV_TRY(gemm_tn_grad(h, w.hc, D, w.dho, D, G + h->oKout, D, D, D, rt, 1.f, w.ntok));
    AV_CHECK(colsum(st, w.dho, rt, D, D, G + h->oBout, w.ntok));
    AV_TRY(gemm(h, false, false, w.dho, D, P + h->oKout, D, w.dhc, D, rt, D, D, 1.f, nullptr, 0, 0, w.ntok, 1, true));
    fire_hook(h, 0);
    const int32_t* const ddyn = w.compact_d ? w.ntgt : nullptr;        // compact decoder layout (build_compact)
    const int32_t* const dmap = w.compact_d ? w.map_tgt : nullptr;
    AV_CHECK(rows_expand(st, w.dhd[0], w.dhc, w.rank, rt, D, dmap));

    // decoder GRU stack, top layer first
    int cur = 0;
    for (int i = L - 1; i >= 0; --i) {
        const GruP& p = h->dec[i];
        GruArgs a{};
        a.njobs = 1; a.S = T; a.B = B; a.D = D; a.ldg = 3 * D; a.ldh = D; a.lens = nullptr;
        a.Bx = w.bx_dec();
        gru_geometry(D, 1, B, &a.G, &a.rows_per_group);
        a.p_begin = 0; a.p_end = T; a.counters = h->counters; a.err = h->errw; a.ablate = h->gru_ablate; a.force_slow = h->gru_force_slow; a.bf16 = h->cfg.compute_dtype == 1 && h->gru_bf16; a.stagger = h->gru_stagger; a.item_pipeline = h->gru_item; a.xbuf = w.xbuf; a.xbuf_floats = w.xbuf_floats; a.stamps = reinterpret_cast<unsigned long long*>(h->errw + 16); a.bwd_rs = h->bwd_rs != 0;
        GruJob& j = a.job[0];
        j.R = P + p.R; j.sv = w.d_sv[i]; j.hp = w.d_hp[i]; j.reverse = 0;
        j.dh_out = w.dhd[cur]; j.dgi = w.dgi_d; j.dgh = w.dgh_d;
        j.dh0 = w.dh0 + (size_t)i * B * D; j.carry = w.carry;
        j.dbW = G + p.bW; j.dbR = G + p.bR;
        // bf16 mode: the team kernels write the gate gradients as bf16, the operand of the three GEMMs below as it stands
        // (a table-fed layer keeps fp32: its gradients are summed by id first)
        const bool g16 = tn16_ok(h, 3 * D, D) && a.bf16 && !(i == 0 && use_table(h, rt, B)) && gru_backward_uses_team(a, h->persistent != 0);
        // (a table-fed layer: dgh alone as bf16 -- its dR GEMM reads it and the bf16 h_prev as they stand; dgi stays fp32 for the sum by id)
        const bool gh16 = !g16 && w.dgh16_d && tn16_ok(h, 3 * D, D) && a.bf16 && w.acth_d[i] && gru_backward_uses_team(a, h->persistent != 0);
        if (g16) { j.dgi16 = w.dgi16_d; j.dgh16 = w.dgh16_d; }
        if (gh16) j.dgh16 = w.dgh16_d;
        if (!g16 && !gh16 && w.acth_d[i]) return fail(h, "internal: the forward kept this layer's h_prev as bf16 only and the backward cannot read it");
        if (w.acth_d[i]) j.hp16 = w.d_hp16[i];
        attach_sv16(h, a, true);
        attach_order(h, w, a, false, 2);
        a.rowmap = dmap;
        if (dmap && i == 0) j.dgi_by_pos = 1;       // (table-fed: its gate gradients are summed by token id)
        a.bwd_rs = rs_pick(h, 1, B); a.spec = spec_pick(h, 1, B);
        hook_fence(h);
        { Timed t(h, 2, 2.0 * T * (double)B * D * 3 * D);
          DeviceTurn turn(h);
          AV_GRU(gru_backward(st, a, h->persistent != 0)); }
        hook_flush(h);
        if (i == 0 && use_table(h, rt, B)) {
            // table-fed layer: gate gradients summed by id, then dW = (sum)^T E and dE += (sum) W over V rows
            const int32_t* cnt = id_groups_count(w.grp_tgt, rt, V); const int U = std::min(V, rt);
            AV_CHECK(rows_group_sum(st, w.dew, w.lead, w.dgi_d, rt, 3 * D, V, w.grp_tgt));
            AV_TRY(gemm_tn_grad(h, w.dew, 3 * D, w.emb_tgt, D, G + p.W, D, 3 * D, D, U, 1.f, cnt));
            if (gh16) AV_TRY(gemm_tn16(h, w.dgh16_d, nullptr, 3 * D, w.d_hp16[i], nullptr, D, G + p.R, D, 3 * D, D, rt, 1.f, ddyn));
            else
            AV_TRY(gemm_tn_grad(h, w.dgh_d, 3 * D, w.d_hp[i], D, G + p.R, D, 3 * D, D, rt, 1.f, ddyn));
            AV_TRY(gemm(h, false, true, w.dew, 3 * D, P + p.W, D, w.demb_tgt, D, U, D, 3 * D, 1.f, nullptr, 0, 0, cnt, 1, true));
            AV_CHECK(rows_add_indexed(st, G + h->oE, w.demb_tgt, id_groups_uid(w.grp_tgt, rt, V), cnt, U, D));
        } else if (g16) {
            const float* x = i == 0 ? w.emb_tgt : w.d_hd[i - 1];
            AV_TRY(gemm_tn16(h, w.dgi16_d, nullptr, 3 * D, (i > 0 && w.act_d[i - 1]) ? w.d_hd16[i - 1] : w.x16_kept_d(i), x, D, G + p.W, D, 3 * D, D, rt, 1.f, ddyn));
            AV_TRY(gemm_tn16(h, w.dgh16_d, nullptr, 3 * D, w.acth_d[i] ? w.d_hp16[i] : nullptr, w.d_hp[i], D, G + p.R, D, 3 * D, D, rt, 1.f, ddyn));
            float* dx = i == 0 ? w.demb_tgt : w.dhd[cur ^ 1];
            AV_TRY(gemm_bf16_pre(h, w.dgi16_d, 3 * D, false, P + p.W, D, true, dx, D, rt, D, 3 * D, 1.f, 0, 1, ddyn, ddyn ? 1 : 0));
        } else {
        const float* x = i == 0 ? w.emb_tgt : w.d_hd[i - 1];
        {   // dW = dgi^T x and dR = dgh^T h_prev: same shape over the same rows, one launch
            const Pair dR{w.dgh_d, w.d_hp[i], G + p.R, nullptr};
            AV_TRY(gemm_tn_grad(h, w.dgi_d, 3 * D, x, D, G + p.W, D, 3 * D, D, rt, 1.f, ddyn, &dR));
        }
        float* dx = i == 0 ? w.demb_tgt : w.dhd[cur ^ 1];
        AV_TRY(gemm(h, false, true, w.dgi_d, 3 * D, P + p.W, D, dx, D, rt, D, 3 * D, 1.f, nullptr, 0, 0, ddyn, ddyn ? 1 : 0, true));
        }
        cur ^= 1;
        fire_hook(h, 1 + (L - 1 - i));
    }

    // latent
    AV_CHECK(add3(st, w.dh0sum, w.dh0, L > 1 ? w.dh0 + (size_t)B * D : nullptr, L > 2 ? w.dh0 + (size_t)2 * B * D : nullptr, (int64_t)B * D));
    for (int i = 3; i < L; ++i) AV_CHECK(add3(st, w.dh0sum, w.dh0sum, w.dh0 + (size_t)i * B * D, nullptr, (int64_t)B * D));
    AV_TRY(gemm_tn_grad(h, w.z, R, w.dh0sum, D, G + h->oWex, D, R, D, B));
    AV_CHECK(colsum(st, w.dh0sum, B, D, D, G + h->oBex, nullptr));
    AV_TRY(gemm(h, false, false, w.dh0sum, D, P + h->oWex, D, w.dz, R, B, R, D));
    float bg = b_global > 0.f ? b_global : (float)B;
    AV_CHECK(latent_bwd(st, w.dz, w.mu, w.lv, w.eps, w.dmu, w.dlv, B, R, sc.anneal * h->cfg.kl_beta / (bg * R), h->cfg.free_bits));
    {
        const Pair dlv{w.hpick, w.dlv, G + h->oWlv, nullptr};
        AV_TRY(gemm_tn_grad(h, w.hpick, 2 * D, w.dmu, R, G + h->oWmu, R, 2 * D, R, B, 1.f, nullptr, &dlv));
    }
    AV_CHECK(colsum(st, w.dmu, B, R, R, G + h->oBmu, nullptr));
    AV_CHECK(colsum(st, w.dlv, B, R, R, G + h->oBlv, nullptr));
    AV_TRY(gemm(h, false, false, w.dmu, R, P + h->oWmu, R, w.dhpick, 2 * D, B, 2 * D, R));
    AV_TRY(gemm(h, false, false, w.dlv, R, P + h->oWlv, R, w.dhpick, 2 * D, B, 2 * D, R, 1.f, nullptr, 1));
    fire_hook(h, 1 + L);
    const int32_t* const cdyn = w.compact ? w.nsrc : nullptr;        // compact encoder layout (build_compact)
    const int32_t* const cmap = w.compact ? w.map_src : nullptr;
    if (w.compact) {
        AV_CHECK(zero_rows_dyn(st, w.dhs[0], w.nsrc, rs, 2 * D));
        AV_CHECK(pick_last_add(st, w.dhs[0], w.dhpick, w.lens_src, B, 2 * D, cmap));
    } else
    AV_CHECK(pick_last_bwd(st, w.dhs[0], w.dhpick, w.lens_src, Ss, B, 2 * D));

    // encoder stack
    cur = 0;
    for (int i = L - 1; i >= 0; --i) {
        const GruP& p = h->enc[i];
        const int In = i == 0 ? D : 2 * D;
        const bool top1 = i == L - 1 && top_one_step(h);
        const int64_t oWb = p.W + (int64_t)3 * D * In;
        if (top1) {
            // backward direction of the top layer: BPTT of its one live step (dH = the pick's gradient, nothing carried),
            // bias gradients = column sums over the B rows; dR of this direction stays exactly zero (h_prev = 0)
            AV_CHECK(gru_first_step_bwd(st, w.dhpick + D, 2 * D, w.svb, w.dgib, w.dghb, B, D, w.lens_src));
            AV_CHECK(colsum(st, w.dgib, B, 3 * D, 3 * D, G + p.bW + 3 * D, nullptr));
            AV_CHECK(colsum(st, w.dghb, B, 3 * D, 3 * D, G + p.bR + 3 * D, nullptr));
        }
        GruArgs a{};
        a.njobs = top1 ? 1 : 2; a.S = Ss; a.B = B; a.D = D; a.ldg = 6 * D; a.ldh = 2 * D; a.lens = w.lens_src;
        a.Bx = w.bx_enc();
        gru_geometry(D, a.njobs, B, &a.G, &a.rows_per_group);
        a.p_begin = 0; a.p_end = Ss; a.counters = h->counters; a.err = h->errw; a.ablate = h->gru_ablate; a.force_slow = h->gru_force_slow; a.bf16 = h->cfg.compute_dtype == 1 && h->gru_bf16; a.stagger = h->gru_stagger; a.item_pipeline = h->gru_item; a.xbuf = w.xbuf; a.xbuf_floats = w.xbuf_floats; a.stamps = reinterpret_cast<unsigned long long*>(h->errw + 16); a.bwd_rs = h->bwd_rs != 0;
        for (int d = 0; d < a.njobs; ++d) {
            GruJob& j = a.job[d];
            j.R = P + p.R + (int64_t)d * 3 * D * D; j.sv = w.e_sv[d][i]; j.hp = w.e_hp[d][i]; j.reverse = d;
            j.dh_out = w.dhs[cur] + d * D; j.dgi = w.dgi_e + d * 3 * D; j.dgh = w.dgh_e + d * 3 * D;
            j.dh0 = nullptr; j.carry = w.carry + (size_t)d * w.Bx * D;
            j.dbW = G + p.bW + d * 3 * D; j.dbR = G + p.bR + d * 3 * D;
        }
        const bool g16 = tn16_ok(h, 3 * D, D) && a.bf16 && !(i == 0 && use_table(h, rs, B)) && gru_backward_uses_team(a, h->persistent != 0);
        const bool gh16 = !g16 && w.dgh16_e && tn16_ok(h, 3 * D, D) && a.bf16 && w.acth_e[i] && gru_backward_uses_team(a, h->persistent != 0);      // (table-fed layer: see the decoder)
        if (g16) for (int d = 0; d < a.njobs; ++d) { a.job[d].dgi16 = w.dgi16_e + d * 3 * D; a.job[d].dgh16 = w.dgh16_e + d * 3 * D; }
        if (gh16) for (int d = 0; d < a.njobs; ++d) a.job[d].dgh16 = w.dgh16_e + d * 3 * D;
        if (!g16 && !gh16 && w.acth_e[i]) return fail(h, "internal: the forward kept this layer's h_prev as bf16 only and the backward cannot read it");
        if (w.acth_e[i]) for (int d = 0; d < a.njobs; ++d) a.job[d].hp16 = w.e_hp16[d][i];
        attach_sv16(h, a, true);
        attach_order(h, w, a, false, top1 ? 1 : 0);
        a.rowmap = cmap;
        if (cmap && i == 0) for (int d = 0; d < a.njobs; ++d) a.job[d].dgi_by_pos = 1;      // (table-fed: its gate gradients are summed by token id)
        a.bwd_rs = rs_pick(h, a.njobs, B); a.spec = spec_pick(h, a.njobs, B);
        hook_fence(h);
        { Timed t(h, 2, 2.0 * a.njobs * (Ss - 1) * (double)B * D * 3 * D);
          DeviceTurn turn(h);
          AV_GRU(gru_backward(st, a, h->persistent != 0)); }
        hook_flush(h);
        const bool table = i == 0 && use_table(h, rs, B);
        const float* x = i == 0 ? w.emb_src : w.e_hs[i - 1];
        float* dx = i == 0 ? w.demb_src : w.dhs[cur ^ 1];
        if (g16) {
            // bf16 mode, gate gradients written as bf16 by the BPTT kernels: every GEMM of the layer reads them as they stand
            const int Gc = a.njobs * 3 * D;                          // gate columns of the directions that ran
            if (i == 0) {     // first layer (per-token form): input gradient, scatter, embedding bucket -- the fixed order of announcements
                AV_TRY(gemm_bf16_pre(h, w.dgi16_e, 6 * D, false, P + p.W, In, true, dx, In, rs, In, Gc, 1.f, 0, 1, nullptr, 0));
                AV_CHECK(embed_scatter_add2(st, G + h->oE, w.src_tm, w.demb_src, rs, w.lead, w.demb_tgt, use_table(h, rt, B) ? 0 : rt, D, V, w.scat));
                fire_hook(h, 2 + 2 * L);
                hook_flush(h);
            }
            AV_TRY(gemm_tn16(h, w.dgi16_e, nullptr, 6 * D, (i > 0 && w.act_e[i - 1]) ? w.e_hs16[i - 1] : w.x16_kept_e(i), x, In, G + p.W, In, Gc, In, rs, 1.f, cdyn));
            if (top1) AV_TRY(gemm_tn_grad(h, w.dgib, 3 * D, w.xlast, In, G + oWb, In, 3 * D, In, B));
            for (int d = 0; d < a.njobs; ++d)
                AV_TRY(gemm_tn16(h, w.dgh16_e + d * 3 * D, nullptr, 6 * D, w.acth_e[i] ? w.e_hp16[d][i] : nullptr, w.e_hp[d][i], D, G + p.R + (int64_t)d * 3 * D * D, D, 3 * D, D, rs, 1.f, cdyn));
            if (i > 0) AV_TRY(gemm_bf16_pre(h, w.dgi16_e, 6 * D, false, P + p.W, In, true, dx, In, rs, In, Gc, 1.f, 0, 1, cdyn, cdyn ? 1 : 0));
            if (top1) {
                AV_TRY(gemm(h, false, true, w.dgib, 3 * D, P + oWb, In, w.dxl, In, B, In, 3 * D, 1.f, nullptr, 0, 0, nullptr, 0, true));
                AV_CHECK(pick_last_add(st, dx, w.dxl, w.lens_src, B, In, cmap));
            }
            cur ^= 1;
            fire_hook(h, 2 + L + (L - 1 - i));
            continue;
        }
        if (i == 0) {
            // First encoder layer: its input gradient completes the embedding gradient.  That bucket is ALWAYS announced here,
            // before this layer's two weight-gradient GEMMs (its all-reduce runs beside them), and this layer's own bucket
            // ends backward -- in the table-fed form and in the per-token form alike.  The order of announcements must not
            // depend on the batch shape: data-parallel ranks pad their shards to their own longest row, so one rank can be
            // on either side of use_table() while its peer is on the other, and the collectives are paired by call order.
            if (table) {
                // table-fed layer (use_table): gate gradients summed by id, dE[present] += (sum) W over U rows
                const int32_t* cnt = id_groups_count(w.grp_src, rs, V); const int U = std::min(V, rs);
                AV_CHECK(rows_group_sum(st, w.dew, w.src_tm, w.dgi_e, rs, 6 * D, V, w.grp_src));
                AV_TRY(gemm(h, false, true, w.dew, 6 * D, P + p.W, D, w.demb_src, D, U, D, 6 * D, 1.f, nullptr, 0, 0, cnt, 1, true));
                AV_CHECK(rows_add_indexed(st, G + h->oE, w.demb_src, id_groups_uid(w.grp_src, rs, V), cnt, U, D));
            } else
            AV_TRY(gemm(h, false, true, w.dgi_e, 6 * D, P + p.W, In, dx, In, rs, In, 6 * D, 1.f, nullptr, 0, 0, nullptr, 0, true));
            // gather gradients of the per-token forms on top of the logits term (a table-fed side has added its rows already)
            AV_CHECK(embed_scatter_add2(st, G + h->oE, w.src_tm, w.demb_src, table ? 0 : rs, w.lead, w.demb_tgt, use_table(h, rt, B) ? 0 : rt, D, V, w.scat));
            fire_hook(h, 2 + 2 * L);
            hook_flush(h);
            if (table) {
                const int32_t* cnt = id_groups_count(w.grp_src, rs, V); const int U = std::min(V, rs);
                AV_TRY(gemm_tn_grad(h, w.dew, 6 * D, w.emb_src, D, G + p.W, D, 6 * D, D, U, 1.f, cnt));
            } else
            AV_TRY(gemm_tn_grad(h, w.dgi_e, 6 * D, x, In, G + p.W, In, 6 * D, In, rs));
        } else if (top1) {
            AV_TRY(gemm_tn_grad(h, w.dgi_e, 6 * D, x, In, G + p.W, In, 3 * D, In, rs, 1.f, cdyn));       // forward direction's W
            AV_TRY(gemm_tn_grad(h, w.dgib, 3 * D, w.xlast, In, G + oWb, In, 3 * D, In, B));              // backward direction's: B rows
        } else
        AV_TRY(gemm_tn_grad(h, w.dgi_e, 6 * D, x, In, G + p.W, In, 6 * D, In, rs, 1.f, cdyn));
        if (gh16) {
            for (int d = 0; d < a.njobs; ++d)
                AV_TRY(gemm_tn16(h, w.dgh16_e + d * 3 * D, nullptr, 6 * D, w.e_hp16[d][i], nullptr, D, G + p.R + (int64_t)d * 3 * D * D, D, 3 * D, D, rs, 1.f, cdyn));
        } else if (top1) AV_TRY(gemm_tn_grad(h, w.dgh_e, 6 * D, w.e_hp[0][i], D, G + p.R, D, 3 * D, D, rs, 1.f, cdyn));
        else {   // dR of the two directions: same shape, one launch
            const Pair bwd{w.dgh_e + 3 * D, w.e_hp[1][i], G + p.R + (int64_t)3 * D * D, nullptr};
            AV_TRY(gemm_tn_grad(h, w.dgh_e, 6 * D, w.e_hp[0][i], D, G + p.R, D, 3 * D, D, rs, 1.f, cdyn, &bwd));
        }
        if (i > 0)
        AV_TRY(gemm(h, false, true, w.dgi_e, 6 * D, P + p.W, In, dx, In, rs, In, top1 ? 3 * D : 6 * D, 1.f, nullptr, 0, 0, cdyn, cdyn ? 1 : 0, true));
        if (top1) {       // the backward direction's input gradient lands on the rows at len_b - 1
            AV_TRY(gemm(h, false, true, w.dgib, 3 * D, P + oWb, In, w.dxl, In, B, In, 3 * D, 1.f, nullptr, 0, 0, nullptr, 0, true));      // (B rows: the skinny form, 43 -> 16 us)
            AV_CHECK(pick_last_add(st, dx, w.dxl, w.lens_src, B, In, cmap));
        }
        cur ^= 1;
        fire_hook(h, 2 + L + (L - 1 - i));
    }
    hook_flush(h);
    return 0;
}

// -------------------------------------------------------------------------------- parameter table
void add_param(avae_ctx* h, const std::string& name, std::initializer_list<int64_t> shape, int g16, int bucket, int64_t* off_out)
{
    ParamEntry e; e.name = name; e.offset = h->numel; e.ndim = (int)shape.size(); e.g16 = g16; e.bucket = bucket;
    int64_t n = 1; int k = 0;
    for (int i = 0; i < 4; ++i) e.shape[i] = 1;
    for (auto s : shape) { e.shape[k++] = s; n *= s; }
    if (off_out) *off_out = e.offset;
    h->numel += (n + 3) / 4 * 4;
    h->params.push_back(e);
}

void build_params(avae_ctx* h)
{
    const int64_t D = h->cfg.dim_emb, V = h->cfg.dim_tgt, R = h->cfg.dim_rep; const int L = h->cfg.rnn_layers;
    h->enc.resize(L); h->dec.resize(L);
    auto close_bucket = [&](int64_t start) { h->buckets.push_back({start, h->numel - start}); };
    int bucket = 0; int64_t start = 0;
    add_param(h, "decode/out/kernel", {D, D}, 0, bucket, &h->oKout);
    add_param(h, "decode/out/bias", {D}, 0, bucket, &h->oBout);
    close_bucket(start);
    for (int i = L - 1; i >= 0; --i) {
        ++bucket; start = h->numel;
        std::string p = "decode/rnn/l" + std::to_string(i + 1) + "/";
        add_param(h, p + "W", {3 * D, D}, 1, bucket, &h->dec[i].W);
        add_param(h, p + "R", {3 * D, D}, 1, bucket, &h->dec[i].R);
        add_param(h, p + "bW", {3 * D}, 1, bucket, &h->dec[i].bW);
        add_param(h, p + "bR", {3 * D}, 1, bucket, &h->dec[i].bR);
        close_bucket(start);
    }
    ++bucket; start = h->numel;
    add_param(h, "latent/ex/kernel", {R, D}, 0, bucket, &h->oWex);
    add_param(h, "latent/ex/bias", {D}, 0, bucket, &h->oBex);
    add_param(h, "latent/mu/kernel", {2 * D, R}, 0, bucket, &h->oWmu);
    add_param(h, "latent/mu/bias", {R}, 0, bucket, &h->oBmu);
    add_param(h, "latent/lv/kernel", {2 * D, R}, 0, bucket, &h->oWlv);
    add_param(h, "latent/lv/bias", {R}, 0, bucket, &h->oBlv);
    close_bucket(start);
    for (int i = L - 1; i >= 0; --i) {
        ++bucket; start = h->numel;
        const int64_t In = i == 0 ? D : 2 * D;
        std::string p = "encode/rnn" + std::to_string(i + 1) + "/";
        int64_t o;
        add_param(h, p + "fwd/W", {3 * D, In}, 1, bucket, &h->enc[i].W);
        add_param(h, p + "bwd/W", {3 * D, In}, 1, bucket, &o);
        add_param(h, p + "fwd/R", {3 * D, D}, 1, bucket, &h->enc[i].R);
        add_param(h, p + "bwd/R", {3 * D, D}, 1, bucket, &o);
        add_param(h, p + "fwd/bW", {3 * D}, 1, bucket, &h->enc[i].bW);
        add_param(h, p + "bwd/bW", {3 * D}, 1, bucket, &o);
        add_param(h, p + "fwd/bR", {3 * D}, 1, bucket, &h->enc[i].bR);
        add_param(h, p + "bwd/bR", {3 * D}, 1, bucket, &o);
        close_bucket(start);
    }
    ++bucket; start = h->numel;
    add_param(h, "embed/embedding", {V, D}, 0, bucket, &h->oE);
    close_bucket(start);
}

const ParamEntry* find_param(avae_ctx* h, const char* name)
{
    for (auto& e : h->params) if (e.name == name) return &e;
    return nullptr;
}

float* state_buf(avae_ctx* h, int kind)
{
    switch (kind) { case AVAE_PARAM: return h->P; case AVAE_GRAD: return h->G; case AVAE_ADAM_M: return h->M; case AVAE_ADAM_V: return h->Vv; }
    return nullptr;
}

int check_bound(avae_ctx* h)
{
    if (!h->P || !h->G || !h->M || !h->Vv) return fail(h, "state buffers not bound (avae_bind_state)");
    return 0;
}

int check_gru_err(avae_ctx* h)
{
    int e = 0;
    AV_CHECK(hipMemcpyAsync(&e, h->errw, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    AV_CHECK(hipStreamSynchronize(h->stream));
    if (e) {
        (void)hipMemsetAsync(h->errw, 0, sizeof(int), h->stream);
        return fail(h, "GRU persistent kernel: an exchange wait timed out -- its workgroups were not all resident at once.  A persistent "
                       "launch needs every CU of the device (one handle = one GPU = one process, include/argsim_vae.h): another process or "
                       "stream computing on this GPU holds CUs the launch is waiting for.  Give the handle the device to itself, or "
                       "run with avae_set_option(\"persistent\", 0) (one launch per time step)");
    }
    return 0;
}

}  // namespace

// ================================================================================= C ABI
extern "C" {

int avae_create(const avae_config* cfg, int device, avae_handle* out)
{
    if (!cfg || !out) { g_create_err = "null argument"; return 1; }
    *out = nullptr;
    if (!gru_dim_supported(cfg->dim_emb)) { g_create_err = "dim_emb must be one of 16, 32, 64, 128, 256, 512 (the GRU kernels are instantiated for these widths only; the reference leaves dim_emb free, config.json uses 512)"; return 1; }
    if (cfg->compute_dtype < 0 || cfg->compute_dtype > 2) { g_create_err = "compute_dtype must be 0 (fp32 MFMA), 1 (bf16 GEMM operands) or 2 (fp32 via split bf16 MFMA)"; return 1; }
    if (cfg->dim_rep % 4 || cfg->dim_tgt % 4 || cfg->rnn_layers < 1 || cfg->rnn_layers > 8) { g_create_err = "dim_rep and dim_tgt must be multiples of 4 (16-byte rows); 1 <= rnn_layers <= 8"; return 1; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { g_create_err = "no HIP device available: the gfx950 kernels cannot run (no CPU fallback)"; return 1; }
    if (device < 0 || device >= ndev) { g_create_err = "bad device index"; return 1; }
    avae_ctx* h = new avae_ctx();
    h->cfg = *cfg; h->device = device;
    if (h->cfg.kl_beta == 0.f) h->cfg.kl_beta = 1.f;
    build_params(h);
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&h->losses), 64 * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&h->errw), 128 * sizeof(int));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&h->counters), 2048 * sizeof(unsigned));
    if (e == hipSuccess) e = hipMemset(h->losses, 0, 64 * sizeof(float));
    if (e == hipSuccess) e = hipMemset(h->errw, 0, 128 * sizeof(int));
    if (e == hipSuccess) e = hipMemset(h->counters, 0, 2048 * sizeof(unsigned));
    if (e != hipSuccess) { g_create_err = std::string("hip init failed: ") + hipGetErrorString(e); delete h; return 1; }
    h->acc = h->losses + 8;
    *out = h;
    return 0;
}

void avae_destroy(avae_handle h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream); else (void)hipDeviceSynchronize();
    if (h->ws) (void)hipFree(h->ws);
    if (h->losses) (void)hipFree(h->losses);
    if (h->errw) (void)hipFree(h->errw);
    if (h->hint_host) (void)hipHostFree(const_cast<int32_t*>(h->hint_host));
    if (h->counters) (void)hipFree(h->counters);
    if (h->scratch) (void)hipFree(h->scratch);
    if (h->bfA) (void)hipFree(h->bfA);
    if (h->slab) (void)hipFree(h->slab);
    if (h->bfP) (void)hipFree(h->bfP);
    if (h->bfB) (void)hipFree(h->bfB);
    if (h->lock_fd >= 0) (void)close(h->lock_fd);
    delete h;
}

const char* avae_last_error(avae_handle h) { return h ? h->err.c_str() : g_create_err.c_str(); }

int avae_set_stream(avae_handle h, void* s) { if (!h) return 1; h->stream = reinterpret_cast<hipStream_t>(s); return 0; }

int avae_get_dims(avae_handle h, int32_t* V, int32_t* D, int32_t* R, int32_t* L)
{
    if (!h) return 1;
    if (V) *V = h->cfg.dim_tgt; if (D) *D = h->cfg.dim_emb; if (R) *R = h->cfg.dim_rep; if (L) *L = h->cfg.rnn_layers;
    return 0;
}

int64_t avae_state_numel(avae_handle h) { return h ? h->numel : 0; }

int avae_bind_state(avae_handle h, float* p, float* g, float* m, float* v)
{
    if (!h) return 1;
    if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) return fail(h, "state buffers must be 16-byte aligned");
    h->P = p; h->G = g; h->M = m; h->Vv = v;
    return 0;
}

int avae_param_count(avae_handle h) { return h ? (int)h->params.size() : 0; }
const char* avae_param_name(avae_handle h, int i) { return (h && i >= 0 && i < (int)h->params.size()) ? h->params[i].name.c_str() : nullptr; }

int avae_param_info(avae_handle h, const char* name, int64_t* offset, int32_t* ndim, int64_t shape[4])
{
    if (!h) return 1;
    const ParamEntry* e = find_param(h, name);
    if (!e) return fail(h, std::string("unknown variable: ") + (name ? name : "(null)"));
    if (offset) *offset = e->offset; if (ndim) *ndim = e->ndim;
    if (shape) for (int i = 0; i < 4; ++i) shape[i] = e->shape[i];
    return 0;
}

static int xfer_tensor(avae_handle h, const char* name, int kind, float* buf, bool get)
{
    if (!h) return 1;
    AV_TRY(check_bound(h));
    const ParamEntry* e = find_param(h, name);
    if (!e) return fail(h, std::string("unknown variable: ") + (name ? name : "(null)"));
    float* base = state_buf(h, kind);
    if (!base) return fail(h, "bad tensor kind");
    float* flat = base + e->offset;
    int64_t n = e->shape[0] * e->shape[1] * e->shape[2] * e->shape[3];
    if (!e->g16) {
        AV_CHECK(hipMemcpyAsync(get ? buf : flat, get ? flat : buf, n * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    } else {
        int D = h->cfg.dim_emb, cols = (int)(e->ndim == 2 ? e->shape[1] : 1);
        if (get) AV_CHECK(g16_permute(h->stream, buf, flat, D, cols, false));
        else     AV_CHECK(g16_permute(h->stream, flat, buf, D, cols, true));
    }
    return 0;
}
int avae_get_tensor(avae_handle h, const char* name, int kind, float* buf) { return xfer_tensor(h, name, kind, buf, true); }
int avae_set_tensor(avae_handle h, const char* name, int kind, const float* buf) { return xfer_tensor(h, name, kind, const_cast<float*>(buf), false); }

int avae_get_step(avae_handle h, int64_t* s) { if (!h || !s) return 1; *s = h->step; return 0; }
int avae_set_step(avae_handle h, int64_t s) { if (!h) return 1; h->step = s; return 0; }
int avae_get_schedule(avae_handle h, float out[3])
{
    if (!h || !out) return 1;
    Sched s = schedule(h); out[0] = s.keepwd; out[1] = s.anneal; out[2] = s.lr;
    return 0;
}

int avae_set_grad_hook(avae_handle h, avae_grad_hook hook, void* user) { if (!h) return 1; h->hook = hook; h->hook_user = user; return 0; }

// undocumented knob used by tests/bench: 1 = persistent GRU kernels (default), 0 = one launch per time step
int avae_set_option(avae_handle h, const char* key, int value)
{
    if (!h || !key) return 1;
    if (!strcmp(key, "persistent")) { h->persistent = value; return 0; }
    if (!strcmp(key, "gru_item")) { h->gru_item = value; return 0; }
    if (!strcmp(key, "gru_stagger")) { h->gru_stagger = value; return 0; }
    if (!strcmp(key, "gru_force_slow")) { h->gru_force_slow = value; return 0; }
    if (!strcmp(key, "gru_bf16")) { h->gru_bf16 = value != 0; return 0; }
    if (!strcmp(key, "bwd_rs")) { h->bwd_rs = value; return 0; }
    if (!strcmp(key, "dyn_split")) { h->dyn_split = value != 0; return 0; }
    if (!strcmp(key, "shared_device")) { h->shared_device = value != 0; return 0; }
    if (!strcmp(key, "gru_spec")) { h->gru_spec = value; return 0; }
    if (!strcmp(key, "bf16_nt8")) { h->bf16_nt8 = value != 0; return 0; }
    if (!strcmp(key, "logits16")) { h->logits16 = value != 0; return 0; }
    if (!strcmp(key, "bf16_direct")) { h->bf16_direct = value != 0; return 0; }
    if (!strcmp(key, "bf16_tn")) { h->bf16_tn = value != 0; return 0; }
    if (!strcmp(key, "bf16_sv")) { h->bf16_sv = value != 0; return 0; }
    if (!strcmp(key, "bf16_act")) { h->bf16_act = value != 0; return 0; }
    if (!strcmp(key, "table_l1")) { h->table_l1 = value != 0; return 0; }
    if (!strcmp(key, "enc_top1")) { h->enc_top1 = value != 0; return 0; }
    if (!strcmp(key, "dyn_thin")) { h->dyn_thin = value != 0; return 0; }
    if (!strcmp(key, "skip_pad")) { h->skip_pad = value != 0; return 0; }
    if (!strcmp(key, "compact")) { h->compact = value; return 0; }
    if (!strcmp(key, "skinny")) { h->skinny = value; return 0; }
    if (!strcmp(key, "gru_ablate")) {
        // timing experiments that change results exist only in the diagnostic build (make DIAG=1)
        if (value && !gru_diag_build()) return fail(h, "gru_ablate needs the diagnostic build of libargsim_vae.so (make -C argsim_amd/csrc DIAG=1)");
        h->gru_ablate = value; return 0;
    }
    if (!strcmp(key, "timing")) { h->timing = value; h->timing_on = value; h->stamps_used = 0; return 0; }
    if (!strcmp(key, "timing_pause")) { h->timing = value ? 0 : h->timing_on; return 0; }
    return fail(h, "unknown option");
}
// the row / depth count a dyn-count GEMM ran with, as a fraction of the static bound its stamp was priced at.  Only the
// last step's count is still on the device (the bench repeats one batch, so it is every stamped step's count); each
// distinct count word is read once per collection.
static int dyn_fraction(avae_ctx* h, const avae_ctx::Stamp& s, std::vector<std::pair<const int*, int>>& seen, double* f)
{
    *f = 1.0;
    if (!s.dyn || s.dyn_max <= 0) return 0;
    int c = -1;
    for (auto& e : seen) if (e.first == s.dyn) c = e.second;
    if (c < 0) {
        AV_CHECK(hipMemcpy(&c, s.dyn, sizeof(int), hipMemcpyDeviceToHost));
        if (c < 0) c = 0;
        seen.push_back({s.dyn, c});
    }
    *f = (double)std::min(c, s.dyn_max) / (double)s.dyn_max;
    return 0;
}
// synchronises, sums the HIP-event durations recorded since timing was switched on / last collected:
// out[3*c + 0..2] = total ms, launches, EXECUTED FLOPs of kernel class c (0 GEMM, 1 GRU fwd, 2 GRU bwd): a GEMM whose
// row count or depth is a device-side count (the table-fed layers' present ids, the kept tokens) exits at that count,
// so its 2MNK is scaled by count / static bound
int avae_timing_collect(avae_handle h, double* out)
{
    if (!h || !out) return 1;
    AV_CHECK(hipStreamSynchronize(h->stream));
    for (int i = 0; i < 9; ++i) out[i] = 0.0;
    std::vector<std::pair<const int*, int>> seen;
    for (size_t i = 0; i < h->stamps_used; ++i) {
        float ms = 0.f;
        AV_CHECK(hipEventElapsedTime(&ms, h->stamps[i].a, h->stamps[i].b));
        double f = 1.0;
        AV_TRY(dyn_fraction(h, h->stamps[i], seen, &f));
        int c = h->stamps[i].cls;
        out[3 * c] += ms; out[3 * c + 1] += 1.0; out[3 * c + 2] += h->stamps[i].flops * f;
    }
    h->stamps_used = 0;
    return 0;
}
// rows of the launch geometry the GRU team kernels take for a batch of B rows (gru_team_batch: B itself, the next row count with a
// geometry -- the slots beyond B hold phantom rows --, or 0).  Host arithmetic only: callable without a GPU.
int avae_debug_team_batch(int32_t B) { return B > 0 ? gru_team_batch(B) : 0; }
// ids present in the last forward's two id sources (encoder input, decoder input) where those layers were table-fed
// (use_table), else -1: out[0] = src, out[1] = tgt.  Synchronises.
int avae_debug_present_ids(avae_handle h, int32_t out[2])
{
    if (!h || !out) return 1;
    AV_CHECK(hipStreamSynchronize(h->stream));
    out[0] = out[1] = -1;
    if (h->cnt_src) AV_CHECK(hipMemcpy(&out[0], h->cnt_src, sizeof(int), hipMemcpyDeviceToHost));
    if (h->cnt_tgt) AV_CHECK(hipMemcpy(&out[1], h->cnt_tgt, sizeof(int), hipMemcpyDeviceToHost));
    return 0;
}
// test hook: the per-token cross-entropy (model.py:180 loss_gen_samp) of the LAST avae_forward_backward / avae_train_step -- the
// TRAIN forward, word dropout and the latent draw live -- copied to out (device memory, max_n floats); *n_out = its token count.
// (The workspace layout is a pure function of the call geometry, so the array is found again without keeping a pointer.)
int avae_debug_train_ce(avae_handle h, float* out, int32_t max_n, int32_t* n_out)
{
    if (!h || !out || !n_out) return 1;
    if (h->B < 1) return fail(h, "avae_debug_train_ce: no training forward has run on this handle");
    Ws w;
    AV_TRY(get_ws(h, w, h->B, h->Ss, h->St, true));
    int n = 0;
    AV_CHECK(hipMemcpyAsync(&n, w.ntok, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    AV_CHECK(hipStreamSynchronize(h->stream));
    n = std::min(n, (int)max_n);
    AV_CHECK(hipMemcpyAsync(out, w.loss_samp, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    *n_out = n;
    return 0;
}
// diagnostic: per-launch (class, ms, FLOPs) triples of the stamps recorded since timing was switched on, in launch
// order (does not reset them); returns the number of stamps through *n
int avae_debug_timing(avae_handle h, double* out, int max_n, int* n)
{
    if (!h || !out || !n) return 1;
    AV_CHECK(hipStreamSynchronize(h->stream));
    *n = (int)std::min<size_t>(h->stamps_used, (size_t)max_n);
    std::vector<std::pair<const int*, int>> seen;
    for (int i = 0; i < *n; ++i) {
        float ms = 0.f;
        AV_CHECK(hipEventElapsedTime(&ms, h->stamps[i].a, h->stamps[i].b));
        double f = 1.0;
        AV_TRY(dyn_fraction(h, h->stamps[i], seen, &f));
        const double fl = h->stamps[i].flops * f;
        out[3 * i] = h->stamps[i].cls; out[3 * i + 1] = ms; out[3 * i + 2] = fl;
    }
    return 0;
}
// diagnostic: reads and clears the 32 GRU phase-stamp words (option gru_ablate bit 32)
int avae_debug_stamps(avae_handle h, unsigned long long* out)
{
    if (!h || !out) return 1;
    AV_CHECK(hipStreamSynchronize(h->stream));
    AV_CHECK(hipMemcpy(out, h->errw + 16, 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    AV_CHECK(hipMemset(h->errw + 16, 0, 32 * sizeof(unsigned long long)));
    return 0;
}
// test hook: the MFMA GEMM on caller buffers (see kernels.h for the operand conventions)
int avae_debug_gemm(avae_handle h, int a_mc, int b_nc, const float* A, const float* Bm, float* Cm, const float* bias,
                    int M, int N, int K, int lda, int ldb, int ldc, float alpha, int accumulate, int split_k)
{
    if (!h) return 1;
    // split_k == -1 selects the thin (32x128 tile) variant, -3 the skinny form, 1000 + s the 64x64-tile variant with s K slices
    return gemm_raw(h, a_mc != 0, b_nc != 0, A, lda, Bm, ldb, Cm, ldc, M, N, K, alpha, bias, accumulate, split_k < 0 ? 1 : (split_k >= 1000 ? split_k - 1000 : split_k), nullptr, 0, split_k == -3 ? 3 : (split_k < 0 ? 1 : (split_k >= 1000 ? 2 : 0)));
}
// test hook: C = A B^T (A (M, K), B (N, K) row-major) over the first *rows rows of A only, rows read on the DEVICE (dyn_kind 1)
int avae_debug_gemm_dyn(avae_handle h, const float* A, const float* Bm, float* Cm, int M, int N, int K, const int* rows)
{
    if (!h) return 1;
    return gemm_raw(h, false, false, A, K, Bm, K, Cm, N, M, N, K, 1.f, nullptr, 0, 1, rows, 1, 0);
}
// test hook (compute_dtype 1): the fp16 output panel of the phased NT GEMM, C16 (M x N) = fp16(alpha * A B^T) over the first *rows rows (rows == nullptr: all);
// returns 3 where the phased kernel does not take the shape
int avae_debug_gemm_c16(avae_handle h, const float* A, const float* Bm, unsigned short* C16, int M, int N, int K, float alpha, const int* rows)
{
    if (!h) return 1;
    if (h->cfg.compute_dtype != 1 || (K & 7)) return 3;
    GemmArgs probe{nullptr, nullptr, nullptr, nullptr, M, N, K, K, K, N, alpha, 0, 1, rows, rows ? 1 : 0, 0, 0, nullptr, nullptr, nullptr, nullptr};
    probe.nt8 = h->bf16_nt8;
    if (!gemm_bf16_c16_ok(probe, K, K)) return 3;
    return gemm_raw(h, false, false, A, K, Bm, K, nullptr, N, M, N, K, alpha, nullptr, 0, 1, rows, rows ? 1 : 0, 0, nullptr, C16);
}
// test hook: C (M x N) += alpha * A^T B with A (K x M, lda), B (K x N, ldb) fp32 row-major, operands rounded to bf16 row by
// row and read through the transposing-LDS-load GEMM (gemm_tn16 / gemm_bf16_tn); C must hold the value to add onto
int avae_debug_gemm_tn16(avae_handle h, const float* A, const float* Bm, float* Cm, int M, int N, int K, int lda, int ldb, int ldc, float alpha)
{
    if (!h) return 1;
    return gemm_tn16(h, nullptr, A, lda, nullptr, Bm, ldb, Cm, ldc, M, N, K, alpha, nullptr);
}
int avae_bucket_count(avae_handle h) { return h ? (int)h->buckets.size() : 0; }
int avae_bucket_info(avae_handle h, int i, int64_t* offset, int64_t* count)
{
    if (!h || i < 0 || i >= (int)h->buckets.size()) return 1;
    *offset = h->buckets[i].first; *count = h->buckets[i].second; return 0;
}

int avae_forward_backward(avae_handle h, const int32_t* src, const int32_t* tgt, int32_t B, int32_t Ss, int32_t St,
                          uint64_t seed, const uint8_t* keep_mask, const float* eps, float n_tok_global, float b_global)
{
    if (!h) return 1;
    AV_TRY(check_bound(h));
    if (B < 1 || Ss < 1 || St < 1) return fail(h, "empty batch");
    AV_CHECK(hipSetDevice(h->device));
    Ws w;
    AV_TRY(get_ws(h, w, B, Ss, St, true));
    h->B = B; h->Ss = Ss; h->St = St;
    AV_TRY(forward(h, w, src, tgt, B, Ss, St, true, seed, keep_mask, eps, n_tok_global > 0.f ? 1.f / n_tok_global : 0.f));
    AV_TRY(backward(h, w, B, Ss, St, b_global));
    if (h->persistent && !h->first_step_checked) {
        // the first step of a handle is checked at once (one synchronisation, once): a device shared with another process
        // fails HERE with the message above instead of training on past time-outs until the losses are next fetched
        h->first_step_checked = true;
        AV_TRY(check_gru_err(h));
    }
    return 0;
}

int avae_adam_step(avae_handle h)
{
    if (!h) return 1;
    AV_TRY(check_bound(h));
    Sched sc = schedule(h);
    const double b1 = 0.9, b2 = 0.999;
    double t = (double)h->step + 1.0;
    AdamArgs a{h->P, h->G, h->M, h->Vv, h->numel, (float)(sc.lr * std::sqrt(1.0 - std::pow(b2, t)) / (1.0 - std::pow(b1, t))), 0.9f, 0.999f, 1e-8f, h->errw};
    AV_CHECK(adam_tf(h->stream, a));
    h->step += 1;
    return 0;
}

int avae_train_step(avae_handle h, const int32_t* src, const int32_t* tgt, int32_t B, int32_t Ss, int32_t St,
                    uint64_t seed, const uint8_t* keep_mask, const float* eps)
{
    AV_TRY(avae_forward_backward(h, src, tgt, B, Ss, St, seed, keep_mask, eps, 0.f, 0.f));
    return avae_adam_step(h);
}

int avae_get_losses(avae_handle h, float out[3])
{
    if (!h || !out) return 1;
    AV_CHECK(hipMemcpyAsync(out, h->losses, 3 * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    AV_TRY(check_gru_err(h));
    return 0;
}

int avae_eval(avae_handle h, const int32_t* src, const int32_t* tgt, int32_t B, int32_t Ss, int32_t St,
              float* errt_samp, float* loss_gen_samp, float* loss_kld_samp, int32_t* n_out)
{
    if (!h) return 1;
    AV_TRY(check_bound(h));
    if (B < 1 || Ss < 1 || St < 1) return fail(h, "empty batch");
    AV_CHECK(hipSetDevice(h->device));
    Ws w;
    AV_TRY(get_ws(h, w, B, Ss, St, false));
    AV_TRY(forward(h, w, src, tgt, B, Ss, St, false, 0, nullptr, nullptr, 0.f));
    const size_t rt = (size_t)(St + 1) * B;
    if (errt_samp) AV_CHECK(hipMemcpyAsync(errt_samp, w.errt_samp, rt * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    if (loss_gen_samp) AV_CHECK(hipMemcpyAsync(loss_gen_samp, w.loss_samp, rt * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    if (loss_kld_samp) AV_CHECK(hipMemcpyAsync(loss_kld_samp, w.kld, (size_t)B * h->cfg.dim_rep * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    int n = 0;
    AV_CHECK(hipMemcpyAsync(&n, w.ntok, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    AV_TRY(check_gru_err(h));
    if (n_out) *n_out = n;
    return 0;
}

int avae_encode(avae_handle h, const int32_t* src, int32_t b, int32_t t, float* z_out, float* lv_out)
{
    if (!h) return 1;
    AV_TRY(check_bound(h));
    if (b < 1 || t < 1) return fail(h, "empty batch");
    AV_CHECK(hipSetDevice(h->device));
    Ws w;
    AV_TRY(get_ws(h, w, b, t, 1, false));
    PrepArgs p{};
    p.src = src; p.tgt = src; p.B = b; p.Ss = t; p.St = 1; p.eos = h->cfg.eos; p.bos = h->cfg.bos;
    p.src_tm = w.src_tm; p.lens_src = w.lens_src; p.lens_tgt = w.lens_tgt; p.lead = w.lead; p.gold = w.gold;
    p.rank = w.rank; p.cidx = w.cidx; p.ntok = w.ntok; p.chunk_counts = w.ntok + 4;
    // tgt is unused by the encoder; feed the first column of src as a 1-wide dummy target
    AV_CHECK(prep_ids(h->stream, p));
    AV_TRY(build_row_orders(h, w, b, t, 2, false));
    AV_TRY(build_compact(h, w, b, t, 0, false));
    AV_TRY(run_encoder(h, w, b, t, false));
    AV_TRY(run_latent(h, w, b, false, 0, nullptr));
    const size_t n = (size_t)b * h->cfg.dim_rep * sizeof(float);
    if (z_out) AV_CHECK(hipMemcpyAsync(z_out, w.mu, n, hipMemcpyDeviceToDevice, h->stream));
    if (lv_out) AV_CHECK(hipMemcpyAsync(lv_out, w.lv, n, hipMemcpyDeviceToDevice, h->stream));
    return check_gru_err(h);       // synchronises: a z computed past a timed-out wait must not be handed out silently
}

int avae_decode_init(avae_handle h, const float* z, int32_t b, float* state_out)
{
    if (!h) return 1;
    AV_TRY(check_bound(h));
    AV_CHECK(hipSetDevice(h->device));
    const int D = h->cfg.dim_emb, R = h->cfg.dim_rep, L = h->cfg.rnn_layers;
    h->rows_form = 1;
    AV_TRY(gemm(h, false, true, z, R, h->P + h->oWex, D, state_out, D, b, D, R, 1.f, h->P + h->oBex));
    for (int i = 1; i < L; ++i)
        AV_CHECK(hipMemcpyAsync(state_out + (size_t)i * b * D, state_out, (size_t)b * D * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    return 0;
}

static int decode_step_ws(avae_handle h, Ws& w, const int32_t* lead, const float* state_in, int b, int32_t* pred_out, float* state_out)
{
    const int D = h->cfg.dim_emb, V = h->cfg.dim_tgt, L = h->cfg.rnn_layers;
    AV_CHECK(embed_gather(h->stream, h->P + h->oE, lead, w.emb_tgt, b, D, V));
    AV_TRY(run_decoder_rnn(h, w, b, 1, state_in, (int64_t)b * D, false));
    for (int i = 0; i < L; ++i)
        AV_CHECK(hipMemcpyAsync(state_out + (size_t)i * b * D, w.d_hd[i], (size_t)b * D * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    AV_TRY(gemm(h, false, true, w.d_hd[L - 1], D, h->P + h->oKout, D, w.ho, D, b, D, D, 1.f, h->P + h->oBout));
    AV_TRY(gemm(h, false, false, w.ho, D, h->P + h->oE, D, w.logits, V, b, V, D, 1.f / sqrtf((float)D)));
    AV_CHECK(argmax_rows(h->stream, w.logits, pred_out, b, V));
    return 0;
}

int avae_decode_step(avae_handle h, const int32_t* lead, const float* state_in, int32_t b, int32_t* pred_out, float* state_out)
{
    if (!h) return 1;
    AV_TRY(check_bound(h));
    AV_CHECK(hipSetDevice(h->device));
    Ws w;
    AV_TRY(get_ws(h, w, b, 1, 1, false));
    AV_TRY(decode_step_ws(h, w, lead, state_in, b, pred_out, state_out));
    return check_gru_err(h);
}

// one launch sequence per token with a host check every 16 tokens: the fallback where the persistent kernel's geometry
// does not fit (decode.hip) and the reference form for tests (option "persistent" = 0)
static int decode_greedy_stepwise(avae_handle h, const float* z, int32_t b, int32_t steps, int32_t* out_ids, int32_t* n_steps)
{
    const int D = h->cfg.dim_emb, L = h->cfg.rnn_layers;
    Ws w;
    AV_TRY(get_ws(h, w, b, 1, 1, false));
    const size_t sn = (size_t)L * b * D;
    float* state[2]; int32_t* ids_tm = nullptr;
    {
        size_t need = 2 * sn * sizeof(float) + (size_t)(steps + 1) * b * sizeof(int32_t);
        if (h->scratch_n < (int64_t)need) {
            AV_CHECK(hipStreamSynchronize(h->stream));
            if (h->scratch) AV_CHECK(hipFree(h->scratch));
            h->scratch = nullptr; h->scratch_n = 0;
            AV_CHECK(hipMalloc(reinterpret_cast<void**>(&h->scratch), need));
            h->scratch_n = (int64_t)need;
        }
        state[0] = h->scratch; state[1] = h->scratch + sn;
        ids_tm = reinterpret_cast<int32_t*>(h->scratch + 2 * sn);
    }
    AV_TRY(avae_decode_init(h, z, b, state[0]));
    std::vector<int32_t> host((size_t)(steps + 1) * b);
    for (int i = 0; i < b; ++i) host[i] = h->cfg.bos;
    AV_CHECK(hipMemcpyAsync(ids_tm, host.data(), b * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
    int done = 0, kept = steps, cur = 0;
    const int chunk = 16;
    while (done < steps) {
        int n = std::min(chunk, steps - done);
        for (int s = 0; s < n; ++s) {
            AV_TRY(decode_step_ws(h, w, ids_tm + (size_t)(done + s) * b, state[cur], b, ids_tm + (size_t)(done + s + 1) * b, state[cur ^ 1]));
            cur ^= 1;
        }
        AV_CHECK(hipMemcpyAsync(host.data() + (size_t)(done + 1) * b, ids_tm + (size_t)(done + 1) * b, (size_t)n * b * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
        AV_CHECK(hipStreamSynchronize(h->stream));
        bool stop = false;
        for (int s = 0; s < n && !stop; ++s) {
            bool all = true;
            for (int i = 0; i < b; ++i) all &= host[(size_t)(done + s + 1) * b + i] == h->cfg.eos;
            if (all) { kept = done + s; stop = true; }      // model.py:217: break before appending
        }
        done += n;
        if (stop) break;
    }
    if (kept > done) kept = done;
    // transpose (kept, b) time-major -> (b, steps) row-major on the host (tiny), eos-fill the rest
    std::vector<int32_t> outv((size_t)b * steps, h->cfg.eos);
    for (int s = 0; s < kept; ++s) for (int i = 0; i < b; ++i) outv[(size_t)i * steps + s] = host[(size_t)(s + 1) * b + i];
    AV_CHECK(hipMemcpyAsync(out_ids, outv.data(), outv.size() * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
    AV_CHECK(hipStreamSynchronize(h->stream));
    if (n_steps) *n_steps = kept;
    return check_gru_err(h);
}

int avae_decode_greedy(avae_handle h, const float* z, int32_t b, int32_t steps, int32_t* out_ids, int32_t* n_steps)
{
    if (!h) return 1;
    AV_TRY(check_bound(h));
    if (b < 1 || steps < 1) return fail(h, "empty batch");
    AV_CHECK(hipSetDevice(h->device));
    // measured at D = 512, V = 8192, steps = 512 (scripts/decode_bench.py, profiles/r03_decode_bench.txt): the persistent launch
    // takes 42 / 74 / 145 us per token at b = 1 / 16 / 64, the launch-per-token loop 116-130 us at any b <= 128 (its
    // GEMMs are far from full): one launch up to 32 rows, the per-token loop above
    if (!h->persistent || b > 32) return decode_greedy_stepwise(h, z, b, steps, out_ids, n_steps);
    // the whole loop in ONE persistent launch (decode.hip); state, partial maxima and the id log live in the scratch buffer
    const int D = h->cfg.dim_emb, V = h->cfg.dim_tgt, L = h->cfg.rnn_layers;
    const int G = decode_workgroups();
    if (G < 1) return fail(h, "no HIP device");
    const size_t sn = (size_t)L * b * D;
    const size_t nf = 2 * sn + (size_t)b * D + (size_t)G * b;                                 // floats: state x2, o, part_val
    const size_t ni = (size_t)G * b + (size_t)(steps + 1) * b + 16;                            // ints: part_idx, ids_tm, kept, barrier
    const size_t need = (nf + ni) * 4;
    if (h->scratch_n < (int64_t)need) {
        AV_CHECK(hipStreamSynchronize(h->stream));
        if (h->scratch) AV_CHECK(hipFree(h->scratch));
        h->scratch = nullptr; h->scratch_n = 0;
        AV_CHECK(hipMalloc(reinterpret_cast<void**>(&h->scratch), need));
        h->scratch_n = (int64_t)need;
    }
    DecodeArgs a{};
    a.E = h->P + h->oE;
    for (int l = 0; l < L; ++l) { a.W[l] = h->P + h->dec[l].W; a.R[l] = h->P + h->dec[l].R; a.bW[l] = h->P + h->dec[l].bW; a.bR[l] = h->P + h->dec[l].bR; }
    a.Kout = h->P + h->oKout; a.bout = h->P + h->oBout;
    a.state[0] = h->scratch; a.state[1] = h->scratch + sn;
    a.o = h->scratch + 2 * sn; a.part_val = a.o + (size_t)b * D;
    int32_t* ip = reinterpret_cast<int32_t*>(h->scratch + nf);
    a.part_idx = ip; a.ids_tm = ip + (size_t)G * b;
    a.kept = a.ids_tm + (size_t)(steps + 1) * b; a.bar = reinterpret_cast<unsigned*>(a.kept + 8);
    a.out_ids = out_ids; a.err = h->errw;
    a.b = b; a.steps = steps; a.D = D; a.V = V; a.L = L; a.eos = h->cfg.eos; a.isd = 1.f / sqrtf((float)D);
    AV_TRY(avae_decode_init(h, z, b, a.state[0]));
    std::vector<int32_t> bos((size_t)b, h->cfg.bos);
    AV_CHECK(hipMemcpyAsync(a.ids_tm, bos.data(), b * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
    AV_CHECK(hipMemsetAsync(a.kept, 0, 16 * sizeof(int32_t), h->stream));
    int grid = 0;
    hipError_t e = decode_greedy(h->stream, a, &grid);
    if (e == hipErrorInvalidValue) {                       // geometry outside the persistent kernel: same results, more launches
        AV_CHECK(hipStreamSynchronize(h->stream));           // (bos.data() is still being read)
        return decode_greedy_stepwise(h, z, b, steps, out_ids, n_steps);
    }
    if (e == hipErrorCooperativeLaunchTooLarge) return fail(h, "greedy decode kernel: one workgroup per CU does not fit this device");
    AV_CHECK(e);
    int kept = 0;
    AV_CHECK(hipMemcpyAsync(&kept, a.kept, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    AV_TRY(check_gru_err(h));                               // synchronises
    if (n_steps) *n_steps = kept;
    return 0;
}

}  // extern "C"
