// gcrnn.hip - CRNN forward / backward orchestration of the GENERIC kernel set (gen.h) behind the same C-ABI entry
// points as crnn.hip (which dispatches here for every configuration other than C = 64 / H = 64 / fp32).
//
// Operator sequence = CRNN.forward (baseline/models/CRNN.py:59-84) with nb_filters = [C, C, C], n_RNN_cell = H:
//   block 0 (blk0.hip, templated on C; always fp32) -> [conv3x3 + BN sums (gconv.hip) -> BN / GLU / dropout / pool
//   (gglu.hip)] x 2 -> BiGRU: input projection (gemm.hip) + recurrence (ggru.hip for H = 256, gru.hip for H = 64)
//   -> heads (heads.hip, templated on 2 H).
#include <string.h>
#include "common.h"
#include "kernels.h"
#include "gkernels.h"
#include "gpack.h"

static inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }
static inline size_t esz(const Geo& g) { return g.mode == SED_DTYPE_BF16 ? 2 : 4; }
// bytes per element of the conv-block activations / gradients in HBM (p0, y1, p1, y2; dz1, dz2, dp0, dp1): bf16 in
// SED_DTYPE_BF16 mode, fp32 otherwise (gen.h Stor<>); p2 - the GRU input - and dp2 are always fp32
static inline size_t ssz(const Geo& g) { return g.mode == SED_DTYPE_BF16 ? 2 : 4; }
static inline int gru_splitk(const Geo& g) { return g.H == 64 ? SED_GRU_SPLITK : 4; }
// split-K partials of the GRU weight-gradient batch: the W_ih problems have N = nin (C, or 2H for layer 1), the W_hh
// problems N = H - the batch's stride is set by the LARGEST of them (with one layer and H > C that is H, not C)
static inline size_t gru_gemm_part_floats(const Geo& g) {
    int max_n = g.H > g.C ? g.H : g.C;
    if (g.L > 1 && 2 * g.H > max_n) max_n = 2 * g.H;
    return gemm_part_floats(4, gru_splitk(g), 3 * g.H, max_n + 1);
}

struct GCtx {
    size_t acc0, mom0, stat1, stat2;          // fp64: patch moments of block 0 | BatchNorm sums of blocks 1, 2
    size_t wz0, wl0, bn0, mompart, p0;
    size_t wpk[3], wpkT[3], wg[3], wgT[3], bg[3], y[3], bn[3], p[3];      // index 1, 2
    size_t ph[2], yh[3];                      // SED_DTYPE_F16: the forward chain's fp16 p0, p1 / y1, y2 (p[], y[] are then the bf16 copies the backward reads)
    size_t gi[2], gates[2], out[2], whh[2], whhT[2];
    size_t wihT[2];                           // [nin][6H]: the two W_ih stacked along K and transposed (dX GEMM operand)
    size_t xch[2], epoch[2], err;             // cluster recurrence: exchange granules, launch epochs, spin-timeout flag
    size_t logits_s, strong_sv, weak_sv, den_sv;
    size_t mask[3];
    size_t sg0;                               // bf16 family: block 0's GLU gate, one byte per element (blk0.hip SG), 0 bytes otherwise
    size_t total;
};
// debug bit 28: block 0's forward SAVES its GLU gates (one byte per element) and the backward reads them instead of recomputing
// z on the MFMA + exp2 + rcp per element.  Built and measured in round 6, NOT the default: at B = 64 the backward kernel goes
// 89 -> 81 us but the student's forward 109 -> 128 us (164 MB of byte stores from a kernel that runs beside the teacher's) -
// waveform-bf16 0.8008 -> 0.8211 ms, mt-bf16 0.4929 -> 0.4944 (profiles/r06_blk0_saved_gates_ab.txt).
static inline bool blk0_saves_gates(const Geo& g) { return g.mode == SED_DTYPE_BF16 && (g_sed_debug & 268435456); }
static size_t mask_bytes(size_t Q, int C) { return ((Q + 3) / 4) * (size_t)(C / 32) * 64 * sizeof(uint16_t); }

static GCtx make_gctx(const Geo& g) {
    GCtx L;
    size_t o = 0;
    auto put = [&](size_t& f, size_t bytes) { f = o; o = al(o + bytes); };
    const size_t C = g.C, H = g.H, E = esz(g);
    const size_t n0 = (size_t)g.B * g.H1 * g.W1 * C, n1 = (size_t)g.B * g.H2 * g.W2 * C, n2 = (size_t)g.B * g.T3 * C;
    put(L.acc0, (64 + 4 * C) * sizeof(double));
    L.mom0 = L.acc0; L.stat1 = L.acc0 + 64 * sizeof(double); L.stat2 = L.stat1 + 2 * C * sizeof(double);
    put(L.wz0, C * 12 * 4); put(L.wl0, C * 12 * 4); put(L.bn0, 4 * C * 4);
    put(L.mompart, (size_t)x_moments_parts(g) * 54 * sizeof(double));
    const size_t SS = ssz(g);
    put(L.p0, n0 * SS);
    L.p[0] = L.p0;
    const size_t nn[3] = {0, n0, n1}, np[3] = {n0, n1, n2};
    for (int i = 1; i <= 2; ++i) {
        put(L.wpk[i], 9 * C * C * E); put(L.wpkT[i], 9 * C * C * E); put(L.wg[i], C * C * E); put(L.wgT[i], C * C * E);
        put(L.bg[i], C * 4); put(L.y[i], nn[i] * SS); put(L.bn[i], 4 * C * 4); put(L.p[i], np[i] * (i == 2 ? 4 : SS));
    }
    L.ph[0] = L.ph[1] = 0; L.yh[0] = L.yh[1] = L.yh[2] = 0;
    if (g.f16) { put(L.ph[0], n0 * 2); put(L.yh[1], n0 * 2); put(L.ph[1], n1 * 2); put(L.yh[2], n1 * 2); }
    const size_t bt = (size_t)g.B * g.T3;
    for (int l = 0; l < 2; ++l) {
        put(L.gi[l], g.H == 64 ? 0 : bt * 6 * H * 4);          // (H = 64: the projection runs inside gru.hip's kernel)
        put(L.gates[l], bt * 8 * H * 4); put(L.out[l], bt * 2 * H * 4);
        put(L.whh[l], g.H == 64 ? 0 : 2 * 3 * H * H * 4); put(L.whhT[l], g.H == 64 ? 0 : 2 * 3 * H * H * 4);
        put(L.xch[l], g.H == 256 ? gclu_xch_bytes(g.B, g.H, 0) : 0); put(L.epoch[l], (size_t)2 * g.B * 4);
        put(L.wihT[l], g.H == 64 ? 0 : (size_t)(l == 0 ? C : 2 * H) * 6 * H * 4);
    }
    put(L.err, 256);
    put(L.logits_s, bt * g.NC * 4); put(L.strong_sv, bt * g.NC * 4);
    put(L.weak_sv, (size_t)g.B * g.NC * 4); put(L.den_sv, (size_t)g.B * g.NC * 4);
    put(L.mask[0], mask_bytes((size_t)g.B * g.H1 * g.W1, g.C)); put(L.mask[1], mask_bytes((size_t)g.B * g.H2 * g.W2, g.C));
    put(L.mask[2], mask_bytes((size_t)g.B * g.T3, g.C));
    // (sized by the MODE alone, never by the debug bit: the caller allocated ctx from sed_crnn_ctx_bytes before any bit was set)
    put(L.sg0, g.mode == SED_DTYPE_BF16 ? mask_bytes((size_t)g.B * g.H1 * g.W1, g.C) * 8 : 0);
    L.total = o;
    return L;
}

struct GWs {
    size_t d_out, dgi[2], dgh[2], hprev[2], d_in, heads_part;
    size_t dp[3], dz[3], coef[3];             // dp[i]: gradient w.r.t. block i's pooled output; dz / coef: blocks 1, 2
    size_t glu_part, glu_part2, de0, wg_part, gemm_part;
    size_t xch[2], epoch[2];
    size_t total;
};
static GWs make_gws(const Geo& g) {
    GWs W;
    size_t o = 0;
    auto put = [&](size_t& f, size_t bytes) { f = o; o = al(o + bytes); };
    const size_t C = g.C, H = g.H, bt = (size_t)g.B * g.T3;
    const size_t n0 = (size_t)g.B * g.H1 * g.W1 * C, n1 = (size_t)g.B * g.H2 * g.W2 * C;
    put(W.d_out, bt * 2 * H * 4);
    for (int l = 0; l < 2; ++l) { put(W.dgi[l], bt * 6 * H * 4); put(W.dgh[l], bt * 6 * H * 4); put(W.hprev[l], bt * 2 * H * 4); }
    put(W.d_in, 2 * bt * 2 * H * 4);          // (H = 64: two direction planes, gru.hip; generic: one tensor)
    put(W.heads_part, heads_part_floats(g.B, g.T3, g.NC, 2 * H) * 4);
    const size_t SS = ssz(g);
    put(W.dp[2], 2 * bt * C * 4); put(W.dz[2], n1 * SS); put(W.dp[1], n1 * SS); put(W.dz[1], n0 * SS); put(W.dp[0], n0 * SS);
    W.dz[0] = 0; W.coef[0] = 0;
    put(W.coef[1], 3 * C * 4); put(W.coef[2], 3 * C * 4);
    put(W.glu_part, (size_t)(g.mode == SED_DTYPE_BF16 ? bglu_bwd_grid(g.C, g.B, g.H1, g.W1) : gglu_bwd_grid(g.B, g.H1, g.W1)) * (C * C + 3 * C) * 4);
    put(W.glu_part2, (size_t)GPART_SLICES * (C * C + 3 * C) * 4);
    put(W.de0, 2 * C * 10 * sizeof(double));
    put(W.wg_part, (size_t)gwgrad_slabs(g.C) * 9 * C * C * 4);
    put(W.gemm_part, 2 * gru_gemm_part_floats(g) * 4);     // one per GRU layer: their batches may overlap in time
    for (int l = 0; l < 2; ++l) { put(W.xch[l], g.H == 256 ? gclu_xch_bytes(g.B, g.H, 1) : 0); put(W.epoch[l], (size_t)2 * g.B * 4); }
    W.total = o;
    return W;
}

size_t gen_ctx_bytes(const Geo& g) { return make_gctx(g).total; }
size_t gen_ws_bytes(const Geo& g) { return make_gws(g).total; }

int gen_ctx_view(const Geo& g, const char* name, size_t* offset, size_t* bytes) {
    const GCtx L = make_gctx(g);
    const size_t C = g.C, H = g.H, bt = (size_t)g.B * g.T3;
    const size_t n0 = (size_t)g.B * g.H1 * g.W1 * C * ssz(g), n1 = (size_t)g.B * g.H2 * g.W2 * C * ssz(g);
    struct { const char* n; size_t o, b; } tab[] = {
        {"mom0", L.mom0, 64 * 8}, {"bn0", L.bn0, 4 * C * 4}, {"p0", L.p[0], n0}, {"y1", L.y[1], n0}, {"stat1", L.stat1, 2 * C * 8},
        {"bn1", L.bn[1], 4 * C * 4}, {"p1", L.p[1], n1}, {"y2", L.y[2], n1}, {"stat2", L.stat2, 2 * C * 8}, {"bn2", L.bn[2], 4 * C * 4},
        {"p2", L.p[2], bt * C * 4}, {"gates0", L.gates[0], bt * 8 * H * 4}, {"gates1", L.gates[1], bt * 8 * H * 4},
        {"gru0", L.out[0], bt * 2 * H * 4}, {"gru1", L.out[1], bt * 2 * H * 4}, {"logits_s", L.logits_s, bt * g.NC * 4},
        {"den", L.den_sv, (size_t)g.B * g.NC * 4}, {"gru_err", L.err, 4},
    };
    for (auto& t : tab)
        if (strcmp(t.n, name) == 0) { *offset = t.o; *bytes = t.b; return SED_OK; }
    sed_set_error("sed_crnn_ctx_view: unknown buffer '%s'", name);
    return SED_ERR_BAD_ARG;
}

// Regions the library needs INITIALISED before the first forward / backward on a fresh buffer (everything else is written
// before it is read): the cluster recurrence's exchange granules + launch epochs (a stale word that happened to carry a
// matching tag would be consumed as a fresh value) and the sticky spin-timeout counter.
int gen_buffers_init(const Geo& g, void* ctx, size_t ctx_bytes, void* ws, size_t ws_bytes, hipStream_t st) {
    if (ctx) {
        const GCtx L = make_gctx(g);
        if (ctx_bytes < L.total) { sed_set_error("sed_crnn_buffers_init: ctx has %zu bytes, needs %zu", ctx_bytes, L.total); return SED_ERR_WORKSPACE; }
        // xch[0] .. err are laid out back to back apart from the small per-layer W_ih transposes: clear each piece
        for (int l = 0; l < 2; ++l) {
            if (g.H == 256) SED_CHECK_HIP(hipMemsetAsync((char*)ctx + L.xch[l], 0, gclu_xch_bytes(g.B, g.H, 0), st));
            SED_CHECK_HIP(hipMemsetAsync((char*)ctx + L.epoch[l], 0, (size_t)2 * g.B * 4, st));
        }
        SED_CHECK_HIP(hipMemsetAsync((char*)ctx + L.err, 0, 256, st));
    }
    if (ws) {
        const GWs W = make_gws(g);
        if (ws_bytes < W.total) { sed_set_error("sed_crnn_buffers_init: ws has %zu bytes, needs %zu", ws_bytes, W.total); return SED_ERR_WORKSPACE; }
        for (int l = 0; l < 2; ++l) {
            if (g.H == 256) SED_CHECK_HIP(hipMemsetAsync((char*)ws + W.xch[l], 0, gclu_xch_bytes(g.B, g.H, 1), st));
            SED_CHECK_HIP(hipMemsetAsync((char*)ws + W.epoch[l], 0, (size_t)2 * g.B * 4, st));
        }
    }
    return SED_OK;
}

#define CTXF(off) ((float*)((char*)ctx + (off)))
#define CTXD(off) ((double*)((char*)ctx + (off)))
#define CTXV(off) ((void*)((char*)ctx + (off)))
#define CTXM(off) ((uint16_t*)((char*)ctx + (off)))
#define WSF(off) ((float*)((char*)ws + (off)))
#define WSD(off) ((double*)((char*)ws + (off)))

int gen_mompart(const Geo& g, void* ctx, size_t ctx_bytes, double** out) {
    const GCtx L = make_gctx(g);
    if (ctx_bytes < L.total) {
        sed_set_error("sed_crnn_moments: ctx has %zu bytes, needs %zu", ctx_bytes, L.total);
        return SED_ERR_WORKSPACE;
    }
    *out = (double*)((char*)ctx + L.mompart);
    return SED_OK;
}

int gen_forward(const Geo& g, const ParamOff& P, const float* params, float* bn_running, int64_t* bn_tracked, const float* x,
                int train, int update_bn, const uint64_t* seed_dev, void* ctx, size_t ctx_bytes, float* strong, float* weak,
                hipStream_t st, hipStream_t ss, hipEvent_t ev_fork, hipEvent_t ev_join) {
    const int mom_ready = (train & 4) ? 1 : 0;      // (sed_crnn_forward's train bit 2)
    train &= 3;
    const GCtx L = make_gctx(g);
    if (ctx_bytes < L.total) {
        sed_set_error("sed_crnn_forward: ctx has %zu bytes, needs %zu", ctx_bytes, L.total);
        return SED_ERR_WORKSPACE;
    }
    const int C = g.C, H = g.H;
    const int gm = (g.mode == SED_DTYPE_BF16X3) ? SED_DTYPE_F32 : g.mode;      // arithmetic of everything but the 3x3 convolutions
    const int use_drop = (train && g.p > 0.f) ? 1 : 0;
    const int upd = (train && update_bn) ? 1 : 0;
    // train & 2: train-mode arithmetic, but no backward will ever run on this ctx (the teacher's forward, main.py:87-89): what only
    // a backward reads need not be written - today the bf16 activation copies of SED_DTYPE_F16
    const bool keep_b16 = train && !(train & 2);
    int64_t* trk[3] = {bn_tracked ? bn_tracked + 0 : nullptr, bn_tracked ? bn_tracked + 1 : nullptr,
                       bn_tracked ? bn_tracked + 2 : nullptr};
    // ---- weight packing (conv panels, GLU weights folded with the BatchNorm affine, GRU streaming layout) + the fp64
    //      BatchNorm accumulators of blocks 1 and 2 ----------------------------------------------------------------------
    GenPackArgs pk = {};
    pk.C = C;
    pk.w1 = params + P.conv_w[1]; pk.w2 = params + P.conv_w[2];
    pk.wpk1 = CTXV(L.wpk[1]); pk.wpk2 = CTXV(L.wpk[2]);
    pk.wpkT1 = train ? CTXV(L.wpkT[1]) : nullptr; pk.wpkT2 = train ? CTXV(L.wpkT[2]) : nullptr;
    pk.glu_w1 = params + P.glu_w[1]; pk.glu_w2 = params + P.glu_w[2]; pk.glu_b1 = params + P.glu_b[1]; pk.glu_b2 = params + P.glu_b[2];
    pk.gamma1 = params + P.bn_g[1]; pk.gamma2 = params + P.bn_g[2]; pk.beta1 = params + P.bn_b[1]; pk.beta2 = params + P.bn_b[2];
    pk.wg1 = CTXV(L.wg[1]); pk.wg2 = CTXV(L.wg[2]);
    pk.wgT1 = train ? CTXV(L.wgT[1]) : nullptr; pk.wgT2 = train ? CTXV(L.wgT[2]) : nullptr;
    pk.bg1 = CTXF(L.bg[1]); pk.bg2 = CTXF(L.bg[2]);
    pk.zero = CTXD(L.stat1); pk.n_zero = train ? 4 * C : 0;
    pk.err = nullptr;                         // sticky: never cleared by a forward (sed_crnn_buffers_init does)
    pk.f16 = g.f16 ? 1 : 0;
    // (The packing is independent of block 0, but forking it onto the helper stream is not an option: a forward that
    // itself runs on a forked stream - the teacher's, next to the student's - would fork a second time inside the same
    // hipGraph capture, and ROCm 7.0's hipStreamEndCapture segfaults on that nested fork.  It stays on the caller's stream.)
    (void)ev_fork; (void)ev_join;
    ss = st;
    // Training forwards: the packing (and the W_ih transposes below) ride in spare workgroups of block 0's moments launch
    // (gpack.h; debug bit 15 keeps the stand-alone kernels for A/B timing).  Eval forwards have no moments launch.
    const bool cluster_ = (H == 256) && !(g_sed_debug & 1024);
    const bool aux_pack = train && (H == 64 || cluster_) && !(g_sed_debug & 32768);
    GenAuxPack aux = {};
    if (aux_pack) {
        aux.pk = pk; aux.mode = g.mode; aux.n_gnt = 0;
        if (H != 64)
            for (int l = 0; l < g.L && l < 2; ++l) {
                aux.gw0[l] = params + P.w_ih[l][0]; aux.gw1[l] = params + P.w_ih[l][1]; aux.gout[l] = CTXF(L.wihT[l]);
                aux.gR[l] = 3 * H; aux.gN[l] = (l == 0) ? C : 2 * H;
                aux.n_gnt = l + 1;
            }
        // the W_hh layouts of the one-CU bf16 recurrence too (k_grec_pack: 6 us in front of every recurrence launch)
        if (H == 256 && g.mode == SED_DTYPE_BF16 && !(g_sed_debug & (1024 | 65536)))
            for (int l = 0; l < g.L && l < 2; ++l) {
                aux.rw0[l] = params + P.w_hh[l][0]; aux.rw1[l] = params + P.w_hh[l][1];
                aux.rwp[l] = CTXV(L.whh[l]); aux.rwpT[l] = CTXV(L.whhT[l]);
                aux.n_grec = l + 1;
            }
    } else {
        SED_TRY(launch_gen_pack(pk, g.mode, ss));
    }
    // (debug bit 10: the streaming recurrence kernels instead of the cluster ones - A/B timing)
    const bool cluster = (H == 256) && !(g_sed_debug & 1024);
    // SED_DTYPE_BF16 at H = 256: one workgroup per chain, W_hh as bf16 in registers (grec.hip; debug bit 16 falls back to
    // the fp32 cluster kernels)
    const bool rec16 = cluster && g.mode == SED_DTYPE_BF16 && !(g_sed_debug & 65536);
    if (H != 64 && !cluster)
        for (int l = 0; l < g.L; ++l)
            SED_TRY(launch_ggru_pack(params + P.w_hh[l][0], params + P.w_hh[l][1], CTXF(L.whh[l]), train ? CTXF(L.whhT[l]) : nullptr, H, ss));
    if (H != 64 && train && !aux_pack)
        for (int l = 0; l < g.L; ++l)
            SED_TRY(launch_gnt_pack_t(params + P.w_ih[l][0], params + P.w_ih[l][1], CTXF(L.wihT[l]), 3 * H, l == 0 ? C : 2 * H, ss));

    // ---- conv block 0 -------------------------------------------------------------------------------------------------
    SED_TRY(launch_blk0_forward(g, x, params + P.conv_w[0], params + P.conv_b[0], params + P.bn_g[0], params + P.bn_b[0],
                                params + P.glu_w[0], params + P.glu_b[0], bn_running + 0, bn_running + C, trk[0], train, upd,
                                seed_dev, CTXD(L.mom0), CTXD(L.mompart), CTXF(L.wz0), CTXF(L.wl0), CTXF(L.bn0),
                                g.f16 ? CTXF(L.ph[0]) : CTXF(L.p[0]),
                                use_drop ? CTXM(L.mask[0]) : nullptr, nullptr, st, 0, aux_pack ? &aux : nullptr,
                                (g.f16 && keep_b16) ? CTXV(L.p[0]) : nullptr, mom_ready,
                                (train && !(train & 2) && blk0_saves_gates(g)) ? CTXV(L.sg0) : nullptr));
    // ---- conv blocks 1, 2 -----------------------------------------------------------------------------------------------
    const size_t so[3] = {0, L.stat1, L.stat2};
    const int Hs[3] = {0, g.H1, g.H2}, Wd[3] = {0, g.W1, g.W2};
    for (int i = 1; i <= 2; ++i) {
        if (g.f16) {
            // SED_DTYPE_F16: the forward chain runs on the fp16 tensors ph / yh; a training forward also leaves the bf16 copies
            // y[i] / p[i] that the (bf16-family) backward kernels read
            SED_TRY(launch_bconv_fwd(2, C, CTXV(L.ph[i - 1]), CTXV(L.wpk[i]), params + P.conv_b[i], CTXV(L.yh[i]),
                                     train ? CTXD(so[i]) : nullptr, g.B, Hs[i], Wd[i], st));
            GBnArgs bnf;
            bnf.stat = CTXD(so[i]); bnf.N = (double)g.B * Hs[i] * Wd[i]; bnf.gamma = params + P.bn_g[i]; bnf.beta = params + P.bn_b[i];
            bnf.run_mean = bn_running + (2 * i) * C; bnf.run_var = bn_running + (2 * i + 1) * C; bnf.tracked = trk[i];
            bnf.train = train; bnf.update = upd; bnf.eps = g.eps; bnf.momentum = g.mom; bnf.bn = CTXF(L.bn[i]);
            SED_TRY(launch_bglu_fwd(C, CTXV(L.yh[i]), bnf, params + P.glu_w[i], params + P.glu_b[i], i == 1 ? CTXV(L.ph[1]) : CTXV(L.p[2]),
                                    i == 1 ? 1 : 0, g.B, Hs[i], Wd[i], i, use_drop, g.p, seed_dev, use_drop ? CTXM(L.mask[i]) : nullptr,
                                    train ? CTXV(L.wg[i]) : nullptr, train ? CTXF(L.bg[i]) : nullptr, st, 1,
                                    (i == 1 && keep_b16) ? CTXV(L.p[1]) : nullptr, keep_b16 ? CTXV(L.y[i]) : nullptr));
            continue;
        }
        if (g.mode != SED_DTYPE_F32)      // bf16 (bf16 storage) / bf16x3 (fp32 storage, split operands): bconv.hip
            SED_TRY(launch_bconv_fwd(g.mode == SED_DTYPE_BF16X3, C, CTXV(L.p[i - 1]), CTXV(L.wpk[i]), params + P.conv_b[i], CTXV(L.y[i]),
                                     train ? CTXD(so[i]) : nullptr, g.B, Hs[i], Wd[i], st));
        else
            SED_TRY(launch_gconv_fwd(g.mode, C, CTXF(L.p[i - 1]), CTXV(L.wpk[i]), params + P.conv_b[i], CTXF(L.y[i]),
                                     train ? CTXD(so[i]) : nullptr, g.B, Hs[i], Wd[i], st));
        GBnArgs bn;
        bn.stat = CTXD(so[i]); bn.N = (double)g.B * Hs[i] * Wd[i]; bn.gamma = params + P.bn_g[i]; bn.beta = params + P.bn_b[i];
        bn.run_mean = bn_running + (2 * i) * C; bn.run_var = bn_running + (2 * i + 1) * C; bn.tracked = trk[i];
        bn.train = train; bn.update = upd; bn.eps = g.eps; bn.momentum = g.mom; bn.bn = CTXF(L.bn[i]);
        if (g.mode == SED_DTYPE_BF16 && !(g_sed_debug & 262144))      // (debug bit 18: the round-2 GLU kernels, A/B timing)
            SED_TRY(launch_bglu_fwd(C, CTXV(L.y[i]), bn, params + P.glu_w[i], params + P.glu_b[i], CTXV(L.p[i]), i == 1 ? 1 : 0, g.B,
                                    Hs[i], Wd[i], i, use_drop, g.p, seed_dev, use_drop ? CTXM(L.mask[i]) : nullptr,
                                    train ? CTXV(L.wg[i]) : nullptr, train ? CTXF(L.bg[i]) : nullptr, st));
        else if (g.mode == SED_DTYPE_BF16X3 && !(g_sed_debug & 8388608))      // (debug bit 23: the exact-fp32 GLU forward in this mode, A/B)
            SED_TRY(launch_bglu_fwd_x3(C, CTXV(L.y[i]), bn, params + P.glu_w[i], params + P.glu_b[i], CTXV(L.p[i]), g.B, Hs[i], Wd[i], i,
                                       use_drop, g.p, seed_dev, use_drop ? CTXM(L.mask[i]) : nullptr, st));
        else
        SED_TRY(launch_gglu_fwd(gm, C, CTXV(L.y[i]), bn, CTXV(L.wg[i]), CTXF(L.bg[i]), CTXV(L.p[i]),
                                (g.mode == SED_DTYPE_BF16 && i == 1) ? 1 : 0, g.B, Hs[i], Wd[i], i,
                                use_drop, g.p, seed_dev, use_drop ? CTXM(L.mask[i]) : nullptr, st));
    }
    // ---- BiGRU ----------------------------------------------------------------------------------------------------------
    SED_TRY(sed_fork_point(st));
    const float* in = CTXF(L.p[2]);
    int nin = C;
    const int BT = g.B * g.T3;
    for (int l = 0; l < g.L; ++l) {
        if (H == 64) {
            SED_TRY(launch_gru_fwd(in, nin, params + P.w_ih[l][0], params + P.w_ih[l][1], params + P.b_ih[l][0], params + P.b_ih[l][1],
                                   params + P.w_hh[l][0], params + P.w_hh[l][1], params + P.b_hh[l][0], params + P.b_hh[l][1],
                                   CTXF(L.out[l]), train ? CTXF(L.gates[l]) : nullptr, g.B, g.T3, st));
        } else {
            // gi[bt][dir][3H] = x W_ih[dir]^T + b_ih[dir]: both directions in one launch (ggemm.hip)
            GntBatch gb;
            gb.n_prob = 2;
            for (int dir = 0; dir < 2; ++dir)
                gb.p[dir] = GntProb{in, nin, params + P.w_ih[l][dir], nin, CTXF(L.gi[l]) + dir * 3 * H, 6 * H, params + P.b_ih[l][dir], BT, 3 * H, nin};
            SED_TRY(g.mode != SED_DTYPE_F32 ? launch_gnt_gemm_bf16(gb, st, g.f16 ? 2 : (g.mode == SED_DTYPE_BF16X3 ? 1 : 0)) : launch_gnt_gemm(gb, st));
            if (rec16) {
                if (!(aux_pack && l < aux.n_grec))
                    SED_TRY(launch_grec_pack(params + P.w_hh[l][0], params + P.w_hh[l][1], CTXV(L.whh[l]), train ? CTXV(L.whhT[l]) : nullptr, st, g.f16 ? 1 : 0));
                SED_TRY(launch_grec_fwd(CTXF(L.gi[l]), CTXV(L.whh[l]), params + P.b_hh[l][0], params + P.b_hh[l][1], CTXF(L.out[l]),
                                        train ? CTXF(L.gates[l]) : nullptr, g.B, g.T3, st, g.f16 ? 1 : 0));
            } else if (cluster)
                SED_TRY(launch_gclu_fwd(CTXF(L.gi[l]), params + P.w_hh[l][0], params + P.w_hh[l][1], params + P.b_hh[l][0],
                                        params + P.b_hh[l][1], CTXF(L.out[l]), train ? CTXF(L.gates[l]) : nullptr, CTXV(L.xch[l]),
                                        (unsigned int*)CTXV(L.epoch[l]), (int*)CTXV(L.err), g.B, g.T3, st));
            else
                SED_TRY(launch_ggru_fwd(H, CTXF(L.gi[l]), CTXF(L.whh[l]), params + P.b_hh[l][0], params + P.b_hh[l][1], CTXF(L.out[l]),
                                        train ? CTXF(L.gates[l]) : nullptr, g.B, g.T3, st));
        }
        in = CTXF(L.out[l]);
        nin = 2 * H;
    }
    // ---- heads ----------------------------------------------------------------------------------------------------------
    if (strong == nullptr) return SED_OK;                 // deferred to sed_mt_step_backward (crnn.hip)
    SED_TRY(launch_heads_fwd(in, params + P.dense_w, params + P.dense_b, params + P.soft_w, params + P.soft_b, strong, weak,
                             train ? CTXF(L.strong_sv) : nullptr, train ? CTXF(L.weak_sv) : nullptr, CTXF(L.logits_s),
                             CTXF(L.den_sv), g.B, g.T3, g.NC, use_drop, g.p, seed_dev, st, 2 * H));
    return SED_OK;
}

// Backward.  `side`: the helper stream of the caller's stream (fork / join events owned by crnn.hip), or the caller's own.
int gen_backward(const Geo& g, const ParamOff& P, const float* params, const float* x, const uint64_t* seed_dev, void* ctx,
                 size_t ctx_bytes, const float* d_strong, const float* d_weak, float* grads, void* ws, size_t ws_bytes, int parts,
                 hipStream_t st, hipStream_t ss, hipEvent_t ev_fork, hipEvent_t ev_join, hipStream_t ss2, hipEvent_t ev_join2,
                 const HeadsLoss* hl, const HeadsOut* ho) {
    const GCtx L = make_gctx(g);
    const GWs W = make_gws(g);
    if (ctx_bytes < L.total || ws_bytes < W.total) {
        sed_set_error("sed_crnn_backward: ctx %zu/%zu bytes, ws %zu/%zu bytes", ctx_bytes, L.total, ws_bytes, W.total);
        return SED_ERR_WORKSPACE;
    }
    const int C = g.C, H = g.H, BT = g.B * g.T3;
    const int gm = (g.mode == SED_DTYPE_BF16X3) ? SED_DTYPE_F32 : g.mode;
    const int use_drop = (g.p > 0.f) ? 1 : 0;
    const bool have_side = (ss != st);
    const bool defer_gru_w = (parts & 4) != 0;
    // (H = 64: measured NEUTRAL to slightly negative - 0.602 against 0.592 ms for the base bf16 step: the conv-block backward is
    // throughput-bound, the GEMMs only trade places with the weight-gradient kernels and slow the latency-bound recurrence)
    // (H = 64, re-measured with the round-3 kernels: 0.580 against 0.569 ms - the GEMMs stretch the lower layer's recurrence from
    // 40 to 52 us)
    const bool early_gru_w = parts == 3 && have_side && (H != 64 || (g_sed_debug & 536870912)) && !(g_sed_debug & 131072);    // (debug bit 17: old schedule; bit 29: early at H = 64 too)
    bool forked = false, forked2 = false;
    auto fork = [&]() -> int {
        if (!have_side) return SED_OK;
        SED_CHECK_HIP(hipEventRecord(ev_fork, st));
        SED_CHECK_HIP(hipStreamWaitEvent(ss, ev_fork, 0));
        forked = true;
        return SED_OK;
    };
    // weight + bias gradients of one GRU layer, both directions (split-K MFMA GEMMs)
    auto gru_weight_grads_layer = [&](int l, hipStream_t s2) -> int {
        const int nin = (l == 0) ? C : 2 * H;
        const float* input = (l == 0) ? CTXF(L.p[2]) : CTXF(L.out[l - 1]);
        GemmBatch gb;
        gb.n_prob = 4; gb.splits = gru_splitk(g); gb.part = WSF(W.gemm_part) + (size_t)l * gru_gemm_part_floats(g);
        gb.part_floats = gru_gemm_part_floats(g); gb.part_stride = 0;
        gb.bf16 = g.mode == SED_DTYPE_BF16 ? 1 : (g.mode == SED_DTYPE_BF16X3 ? 2 : 0);     // GRU weight gradients: operand mode of the MFMA
        for (int dir = 0; dir < 2; ++dir) {
            gb.p[2 * dir] = gemm_prob(WSF(W.dgi[l]) + dir * 3 * H, 1, 6 * H, input, nin, 1, grads + P.w_ih[l][dir], nin, 3 * H, nin, BT);
            gb.p[2 * dir].Cones = grads + P.b_ih[l][dir];
            gb.p[2 * dir + 1] = gemm_prob(WSF(W.dgh[l]) + dir * 3 * H, 1, 6 * H, WSF(W.hprev[l]) + dir * H, 2 * H, 1,
                                          grads + P.w_hh[l][dir], H, 3 * H, H, BT);
            gb.p[2 * dir + 1].Cones = grads + P.b_hh[l][dir];
        }
        return launch_gemm_batch(gb, s2);
    };
    auto gru_weight_grads = [&](hipStream_t s2) -> int {
        for (int l = g.L - 1; l >= 0; --l) SED_TRY(gru_weight_grads_layer(l, s2));
        return SED_OK;
    };
    // ho != null: heads deferred by the forward (sed_mt_step_backward, crnn.hip).  H = 64: fused into the top layer's backward
    // recurrence (hfuse.h); H = 256 (or debug bit 24): k_heads_fwd here, then the two-kernel form.
    const int head_cols = 2 * (g.NC * 2 * H + g.NC);
    const bool fuse = ho && hl && (parts & 1) && heads_fusable(H, g.T3) && !(g_sed_debug & 16777216) && !hl->d_strong_out && !hl->d_weak_out;
    auto heads_colsum = [&](hipStream_t s2) -> int {
        if (fuse) return launch_heads_fin(WSF(W.heads_part), grads + P.dense_w, g.B, g.T3, g.NC, head_cols, *hl, s2);
        return launch_heads_colsum(WSF(W.heads_part), grads + P.dense_w, g.B * heads_bwd_chunks(2 * H, g.T3), g.NC, s2, 2 * H);
    };
    const bool defer_colsum = ((parts & 2) && have_side) || defer_gru_w;
    if (parts & 1) {
        // ---- heads ------------------------------------------------------------------------------------------------------
        if (ho && !fuse)
            SED_TRY(launch_heads_fwd(CTXF(L.out[g.L - 1]), params + P.dense_w, params + P.dense_b, params + P.soft_w, params + P.soft_b,
                                     ho->strong, ho->weak, CTXF(L.strong_sv), CTXF(L.weak_sv), CTXF(L.logits_s), CTXF(L.den_sv), g.B, g.T3,
                                     g.NC, use_drop, g.p, seed_dev, st, 2 * H));
        if (!fuse)
        SED_TRY(launch_heads_bwd(CTXF(L.out[g.L - 1]), params + P.dense_w, params + P.soft_w, CTXF(L.strong_sv), CTXF(L.weak_sv),
                                 CTXF(L.logits_s), CTXF(L.den_sv), d_strong, d_weak, WSF(W.d_out), WSF(W.heads_part),
                                 grads + P.dense_w, grads + P.dense_b, grads + P.soft_w, grads + P.soft_b, g.B, g.T3, g.NC, use_drop,
                                 g.p, seed_dev, (parts & 2) ? WSD(W.de0) : nullptr, 2 * C * 10,
                                 defer_colsum ? 1 : 0, hl, st, 2 * H));
        // ---- BiGRU --------------------------------------------------------------------------------------------------------
        const float* d_cur = WSF(W.d_out);
        const float* d_cur2 = nullptr;
        for (int l = g.L - 1; l >= 0; --l) {
            const int nin = (l == 0) ? C : 2 * H;
            float* d_in = (l == 0) ? WSF(W.dp[2]) : WSF(W.d_in);
            if (H == 64) {
                if (fuse && l == g.L - 1) {
                    HeadsFuse hf = {};
                    hf.wd = params + P.dense_w; hf.strong = ho->strong; hf.weak = ho->weak; hf.part = WSF(W.heads_part);
                    hf.NC = g.NC; hf.use_drop = use_drop; hf.p_drop = g.p; hf.seed = seed_dev;
                    hf.zero = (parts & 2) ? WSD(W.de0) : nullptr; hf.n_zero = (parts & 2) ? 2 * C * 10 : 0;
                    hf.hl = *hl;
                    SED_TRY(launch_gru_bwd_heads(CTXF(L.out[l]), CTXF(L.gates[l]), params + P.w_hh[l][0], params + P.w_hh[l][1],
                                                 params + P.w_ih[l][0], params + P.w_ih[l][1], nin, WSF(W.dgi[l]), WSF(W.dgh[l]),
                                                 WSF(W.hprev[l]), d_in, g.B, g.T3, hf, st));
                    if (!defer_colsum) SED_TRY(heads_colsum(st));
                    else if (defer_gru_w) SED_TRY(launch_heads_fin(WSF(W.heads_part), grads + P.dense_w, g.B, g.T3, g.NC, 0, *hl, st));
                } else
                SED_TRY(launch_gru_bwd(d_cur, d_cur2, CTXF(L.out[l]), CTXF(L.gates[l]), params + P.w_hh[l][0], params + P.w_hh[l][1],
                                       params + P.w_ih[l][0], params + P.w_ih[l][1], nin, WSF(W.dgi[l]), WSF(W.dgh[l]), WSF(W.hprev[l]),
                                       d_in, g.B, g.T3, st));
                // this layer's weight-gradient GEMMs start on the helper stream right behind its recurrence: the rest of the GRU
                // chain keeps 48 CUs busy, and left for the conv blocks' fork they ended the step ~70 us after the caller's
                // stream had finished (profiles/r03_a_mt-bf16_step_timeline.txt)
                if (early_gru_w) {
                    SED_TRY(fork());
                    if (l == g.L - 1) SED_TRY(heads_colsum(ss));
                    SED_TRY(gru_weight_grads_layer(l, ss));
                }
                d_cur = d_in;
                d_cur2 = d_in + (size_t)BT * nin;
            } else {
                if (H == 256 && g.mode == SED_DTYPE_BF16 && !(g_sed_debug & (1024 | 65536)))
                    SED_TRY(launch_grec_bwd(d_cur, CTXF(L.out[l]), CTXF(L.gates[l]), CTXV(L.whhT[l]), WSF(W.dgi[l]), WSF(W.dgh[l]),
                                            WSF(W.hprev[l]), g.B, g.T3, st));
                else if (H == 256 && !(g_sed_debug & 1024))
                    SED_TRY(launch_gclu_bwd(d_cur, CTXF(L.out[l]), CTXF(L.gates[l]), params + P.w_hh[l][0], params + P.w_hh[l][1],
                                            WSF(W.dgi[l]), WSF(W.dgh[l]), WSF(W.hprev[l]), (void*)((char*)ws + W.xch[l]),
                                            (unsigned int*)((char*)ws + W.epoch[l]), (int*)CTXV(L.err), g.B, g.T3, st));
                else
                    SED_TRY(launch_ggru_bwd(H, d_cur, CTXF(L.out[l]), CTXF(L.gates[l]), CTXF(L.whhT[l]), WSF(W.dgi[l]), WSF(W.dgh[l]),
                                            WSF(W.hprev[l]), g.B, g.T3, st));
                // dX[bt][i] = sum_dir sum_g dgi[bt][dir][g] W_ih[dir][g][i]: K = 6H, the two W_ih stacked along K (transposed
                // copy made by the forward)
                // H = 256: this layer's weight-gradient GEMMs (100 - 160 us of split-K work) start on the helper stream NOW, next
                // to the rest of the recurrence chain - which keeps 48 of 256 CUs busy for another ~300 us - instead of after
                // the conv blocks' fork, where they used to be the tail of the step (profiles/r03_*_wide-bf16_step_timeline.txt)
                // The fork event is recorded in front of the dX GEMM, but the GEMM - the critical chain - is CAPTURED FIRST: the
                // graph executor keeps the first-captured child of a node on its parent's hardware queue, and with the helper
                // stream's kernels captured first the dX GEMM hopped to another queue, ~10 us of cross-queue latency per layer
                // (profiles/r05b_wide-bf16_step_timeline.txt: "idle 9.9" in front of it).  Both still start together.
                if (early_gru_w && have_side) SED_CHECK_HIP(hipEventRecord(ev_fork, st));
                GntBatch gb;
                gb.n_prob = 1;
                gb.p[0] = GntProb{WSF(W.dgi[l]), 6 * H, CTXF(L.wihT[l]), 6 * H, d_in, nin, nullptr, BT, nin, 6 * H};
                SED_TRY(g.mode != SED_DTYPE_F32 ? launch_gnt_gemm_bf16(gb, st, g.mode == SED_DTYPE_BF16X3) : launch_gnt_gemm(gb, st));
                if (early_gru_w) {
                    if (have_side) { SED_CHECK_HIP(hipStreamWaitEvent(ss, ev_fork, 0)); forked = true; }
                    if (l == g.L - 1) SED_TRY(heads_colsum(ss));
                    SED_TRY(gru_weight_grads_layer(l, ss));
                }
                d_cur = d_in;
                d_cur2 = nullptr;
            }
        }
    }
    if (parts == 1) SED_TRY(gru_weight_grads(st));
    if (parts == 8) {
        SED_TRY(launch_heads_colsum(WSF(W.heads_part), grads + P.dense_w, g.B * heads_bwd_chunks(2 * H, g.T3), g.NC, st, 2 * H));
        SED_TRY(gru_weight_grads(st));
        return SED_OK;
    }
    if (!(parts & 2)) return SED_OK;
    // ---- conv blocks 2, 1 -------------------------------------------------------------------------------------------------
    if (!(parts & 1)) SED_CHECK_HIP(hipMemsetAsync(WSD(W.de0), 0, 2 * C * 10 * sizeof(double), st));
    const int Hs[3] = {0, g.H1, g.H2}, Wd[3] = {0, g.W1, g.W2};
    for (int i = 2; i >= 1; --i) {
        // (H = 64: the GRU's dX arrives as two direction planes; the GLU backward adds them while loading)
        const bool new_glu = g.mode == SED_DTYPE_BF16 && !(g_sed_debug & 262144);
        if (new_glu)
            SED_TRY(launch_bglu_bwd(C, CTXV(L.y[i]), CTXF(L.bn[i]), CTXV(L.wg[i]), CTXF(L.bg[i]), CTXV(L.wgT[i]), WSF(W.dp[i]), i == 1 ? 1 : 0, WSF(W.dz[i]), WSF(W.glu_part), g.B, Hs[i], Wd[i],
                                    use_drop, g.p, CTXM(L.mask[i]), st, (i == 2 && H == 64) ? WSF(W.dp[2]) + (size_t)BT * C : nullptr));
        else
        SED_TRY(launch_gglu_bwd(gm, C, CTXV(L.y[i]), CTXF(L.bn[i]), params + P.bn_g[i], params + P.bn_b[i], CTXV(L.wg[i]),
                                CTXV(L.wgT[i]), CTXF(L.bg[i]), WSF(W.dp[i]), (g.mode == SED_DTYPE_BF16 && i == 1) ? 1 : 0,
                                WSF(W.dz[i]), WSF(W.glu_part), g.B, Hs[i], Wd[i], use_drop,
                                g.p, CTXM(L.mask[i]), st, (i == 2 && H == 64) ? WSF(W.dp[2]) + (size_t)BT * C : nullptr));
        GBnBwdArgs pa;
        pa.part = WSF(W.glu_part); pa.n_part = new_glu ? bglu_bwd_grid(C, g.B, Hs[i], Wd[i]) : gglu_bwd_grid(g.B, Hs[i], Wd[i]); pa.C = C; pa.N = (double)g.B * Hs[i] * Wd[i];
        pa.part2 = WSF(W.glu_part2);
        pa.gamma = params + P.bn_g[i]; pa.beta = params + P.bn_b[i]; pa.bn = CTXF(L.bn[i]); pa.coef = WSF(W.coef[i]);
        pa.g_gamma = grads + P.bn_g[i]; pa.g_beta = grads + P.bn_b[i]; pa.g_wglu = grads + P.glu_w[i]; pa.g_bglu = grads + P.glu_b[i];
        pa.g_convb = grads + P.conv_b[i];
        SED_TRY(launch_gbn_bwd_prep(pa, st));
        // The weight gradient (helper stream) and the data gradient (caller's stream) both depend on the coefficients only.
        // The data gradient is on the critical path: it is CAPTURED FIRST - a replayed hipGraph serialises nodes that its
        // executor maps to the same hardware queue in creation order, and with the weight gradient created first the 145 us
        // wgrad kernel sat in front of dgrad1 + block 0 on one queue (profiles/r03_a_wide-bf16_step_timeline.txt)
        if (have_side) SED_CHECK_HIP(hipEventRecord(ev_fork, st));
        if (g.mode != SED_DTYPE_F32)
            SED_TRY(launch_bconv_dgrad(g.mode == SED_DTYPE_BF16X3, C, WSF(W.dz[i]), CTXV(L.y[i]), WSF(W.coef[i]), CTXV(L.wpkT[i]),
                                       WSF(W.dp[i - 1]), g.B, Hs[i], Wd[i], st));
        else
            SED_TRY(launch_gconv_dgrad(g.mode, C, WSF(W.dz[i]), CTXF(L.y[i]), WSF(W.coef[i]), CTXV(L.wpkT[i]), WSF(W.dp[i - 1]), g.B, Hs[i],
                                       Wd[i], st));
        if (have_side) { SED_CHECK_HIP(hipStreamWaitEvent(ss, ev_fork, 0)); forked = true; }
        SED_TRY(launch_gwgrad(g.mode, C, WSF(W.dz[i]), CTXF(L.y[i]), WSF(W.coef[i]), CTXF(L.p[i - 1]), WSF(W.wg_part), grads + P.conv_w[i], g.B,
                              Hs[i], Wd[i], ss));
        if (i == 2 && parts == 3 && !early_gru_w) {
            // on the second helper stream, so that they do not sit in front of wgrad1 on the first (crnn.hip)
            hipStream_t sg = ss;
            if (have_side && ss2 != nullptr) {
                SED_CHECK_HIP(hipStreamWaitEvent(ss2, ev_fork, 0));
                sg = ss2;
                forked2 = true;
            }
            if (have_side) SED_TRY(heads_colsum(sg));
            SED_TRY(gru_weight_grads(sg));
        }
    }
    // ---- conv block 0 -----------------------------------------------------------------------------------------------------
    SED_TRY(launch_blk0_backward(g, x, params + P.conv_w[0], params + P.conv_b[0], params + P.bn_g[0], params + P.bn_b[0],
                                 params + P.glu_w[0], CTXM(L.mask[0]), CTXD(L.mom0), CTXF(L.wz0), CTXF(L.wl0), CTXF(L.bn0), WSF(W.dp[0]),
                                 WSD(W.de0), 0, grads + P.conv_w[0], grads + P.conv_b[0], grads + P.bn_g[0], grads + P.bn_b[0],
                                 grads + P.glu_w[0], grads + P.glu_b[0], st, blk0_saves_gates(g) ? CTXV(L.sg0) : nullptr));
    if (forked) {
        SED_CHECK_HIP(hipEventRecord(ev_join, ss));
        SED_CHECK_HIP(hipStreamWaitEvent(st, ev_join, 0));
    }
    if (forked2) {
        SED_CHECK_HIP(hipEventRecord(ev_join2, ss2));
        SED_CHECK_HIP(hipStreamWaitEvent(st, ev_join2, 0));
    }
    return SED_OK;
}

