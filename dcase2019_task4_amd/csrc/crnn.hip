// crnn.hip - CRNN forward / backward orchestration behind the C-ABI (include/dcase_sed.h).
//
// Mirrors the operator sequence of CRNN.forward (baseline/models/CRNN.py:59-84):
//   conv block 0 (blk0.hip, fully fused) -> [conv3x3 + BN stats (conv.hip) -> BN/GLU/dropout/pool
//   (bnglu.hip)] x 2 -> 2-layer BiGRU (gru.hip: input projection + recurrence per layer) -> heads (heads.hip)
// All launches go to the caller's stream; no allocation, no synchronisation.
#include <stdarg.h>
#include <map>
#include <mutex>
#include <utility>
#include <stdio.h>
#include <string.h>
#include "common.h"
#include "kernels.h"
#include "gkernels.h"

static thread_local char g_err[512] = "";
void sed_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* sed_last_error(void) { return g_err; }
extern "C" int sed_version(void) { return 100; }
int g_sed_debug = 0;
extern "C" int sed_debug_set(int flags) { const int old = g_sed_debug; g_sed_debug = flags; return old; }

int sed_validate_dims(const sed_dims* d) {
    SED_CHECK_ARG(d != nullptr, "null dims");
    if (d->F != 64 || (d->C != 64 && d->C != 128) || (d->H != 64 && d->H != 256) ||
        (d->dtype != SED_DTYPE_F32 && d->dtype != SED_DTYPE_BF16 && d->dtype != SED_DTYPE_BF16X3 && d->dtype != SED_DTYPE_F16)) {
        sed_set_error("supported: F = 64, C in {64, 128}, H in {64, 256}, dtype in {f32, bf16, bf16x3, f16} (got F=%d C=%d H=%d dtype=%d)",
                      d->F, d->C, d->H, d->dtype);
        return SED_ERR_UNSUPPORTED;
    }
    SED_CHECK_ARG(d->B >= 1 && d->T >= 16, "need B >= 1 and T >= 16");
    SED_CHECK_ARG(d->nclass >= 1 && d->nclass <= 16, "nclass must be in [1, 16]");
    SED_CHECK_ARG(d->n_layers_rnn == 1 || d->n_layers_rnn == 2, "n_layers_rnn must be 1 or 2");
    SED_CHECK_ARG(d->p_drop >= 0.f && d->p_drop < 1.f, "p_drop must be in [0, 1)");
    SED_CHECK_ARG((int64_t)d->B * d->T * d->F * d->C < (1ll << 31), "batch too large for 32-bit pixel indices");
    return SED_OK;
}

ParamOff make_param_off(const Geo& g, int64_t* out) {
    ParamOff P;
    int64_t o = 0;
    int k = 0;
    auto put = [&](int64_t& field, int64_t n) {
        field = o;
        if (out) out[k] = o;
        ++k;
        o += n;
    };
    for (int i = 0; i < 3; ++i) {
        const int cin = (i == 0) ? 1 : g.C;
        put(P.conv_w[i], (int64_t)g.C * cin * 9);
        put(P.conv_b[i], g.C);
        put(P.bn_g[i], g.C);
        put(P.bn_b[i], g.C);
        put(P.glu_w[i], (int64_t)g.C * g.C);
        put(P.glu_b[i], g.C);
    }
    for (int l = 0; l < g.L; ++l) {
        const int nin = (l == 0) ? g.C : 2 * g.H;
        for (int dir = 0; dir < 2; ++dir) {
            put(P.w_ih[l][dir], (int64_t)3 * g.H * nin);
            put(P.w_hh[l][dir], (int64_t)3 * g.H * g.H);
            put(P.b_ih[l][dir], 3 * g.H);
            put(P.b_hh[l][dir], 3 * g.H);
        }
    }
    put(P.dense_w, (int64_t)g.NC * 2 * g.H);
    put(P.dense_b, g.NC);
    put(P.soft_w, (int64_t)g.NC * 2 * g.H);
    put(P.soft_b, g.NC);
    P.total = o;
    P.count = k;
    if (out) out[k] = o;
    return P;
}

static inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

CtxLayout make_ctx_layout(const Geo& g) {
    CtxLayout L;
    size_t o = 0;
    auto put = [&](size_t& f, size_t bytes) { f = o; o = al(o + bytes); };
    const size_t n0 = (size_t)g.B * g.H1 * g.W1 * 64, n1 = (size_t)g.B * g.H2 * g.W2 * 64, n2 = (size_t)g.B * g.T3 * 64;
    put(L.acc0, 320 * sizeof(double));
    L.mom0 = L.acc0; L.stat1 = L.acc0 + 64 * sizeof(double); L.stat2 = L.acc0 + 192 * sizeof(double);
    put(L.wz0, 64 * 12 * 4); put(L.wl0, 64 * 12 * 4); put(L.bn0, 256 * 4);
    put(L.mompart, (size_t)x_moments_parts(g) * 54 * sizeof(double));
    put(L.p0, n0 * 4);
    put(L.wpk1, (9 + 16) * 4096 * 4); put(L.wpkT1, (9 + 16) * 4096 * 4); put(L.y1, n0 * 4); put(L.bn1, 256 * 4); put(L.p1, n1 * 4);
    put(L.wpk2, (9 + 16) * 4096 * 4); put(L.wpkT2, (9 + 16) * 4096 * 4); put(L.y2, n1 * 4); put(L.bn2, 256 * 4); put(L.p2, n2 * 4);
    const size_t bt = (size_t)g.B * g.T3;
    for (int l = 0; l < 2; ++l) {
        put(L.gates[l], bt * 512 * 4); put(L.out[l], bt * 128 * 4);      // (gi only exists in LDS, gru.hip)
    }
    put(L.logits_s, bt * g.NC * 4); put(L.strong_sv, bt * g.NC * 4);
    put(L.weak_sv, (size_t)g.B * g.NC * 4); put(L.den_sv, (size_t)g.B * g.NC * 4);
    auto mask_bytes = [](size_t Q) { return ((Q + 3) / 4) * 2 * 64 * sizeof(uint16_t); };
    put(L.mask0, mask_bytes((size_t)g.B * g.H1 * g.W1)); put(L.mask1, mask_bytes((size_t)g.B * g.H2 * g.W2));
    put(L.mask2, mask_bytes((size_t)g.B * g.T3));
    L.total = o;
    return L;
}

WsLayout make_ws_layout(const Geo& g) {
    WsLayout W;
    size_t o = 0;
    auto put = [&](size_t& f, size_t bytes) { f = o; o = al(o + bytes); };
    const size_t n0 = (size_t)g.B * g.H1 * g.W1 * 64, n1 = (size_t)g.B * g.H2 * g.W2 * 64, bt = (size_t)g.B * g.T3;
    put(W.d_out, bt * 128 * 4);
    for (int l = 0; l < 2; ++l) { put(W.dgi[l], bt * 384 * 4); put(W.dgh[l], bt * 384 * 4); put(W.hprev[l], bt * 128 * 4); }
    put(W.d_in, 2 * bt * 128 * 4); put(W.heads_part, (size_t)g.B * 2 * (g.NC * 128 + g.NC) * 4);
    put(W.dp2, 2 * bt * 64 * 4); put(W.dz2, n1 * 4); put(W.dp1, n1 * 4); put(W.dz1, n0 * 4); put(W.dp0, n0 * 4);
    put(W.bnb, 256 * sizeof(double)); W.coef[0] = 0; put(W.coef[1], 192 * 4); put(W.coef[2], 192 * 4);
    put(W.bwd_acc, (2 * SED_GLUACC_N + 2 * 64 * 10) * sizeof(double));
    W.gluacc1 = W.bwd_acc; W.gluacc2 = W.bwd_acc + SED_GLUACC_N * sizeof(double); W.de0 = W.bwd_acc + 2 * SED_GLUACC_N * sizeof(double);
    W.wgrad_blocks = SED_WGRAD_MAX_BLOCKS;
    put(W.wg_part, (size_t)W.wgrad_blocks * 9 * 4096 * 4);
    put(W.gemm_part, gemm_part_floats(4, SED_GRU_SPLITK, 192, 129) * 4);
    W.total = o;
    return W;
}

extern "C" int sed_param_count(const sed_dims* d) {
    if (sed_validate_dims(d) != SED_OK) return -1;
    const Geo g = make_geo(d);
    return make_param_off(g, nullptr).count;
}
extern "C" int sed_param_layout(const sed_dims* d, int64_t* offsets) {
    SED_TRY(sed_validate_dims(d));
    SED_CHECK_ARG(offsets != nullptr, "null offsets");
    const Geo g = make_geo(d);
    make_param_off(g, offsets);
    return SED_OK;
}
extern "C" size_t sed_crnn_ctx_bytes(const sed_dims* d) {
    if (sed_validate_dims(d) != SED_OK) return 0;
    const Geo g = make_geo(d);
    return g.generic ? gen_ctx_bytes(g) : make_ctx_layout(g).total;
}
extern "C" size_t sed_crnn_bwd_ws_bytes(const sed_dims* d) {
    if (sed_validate_dims(d) != SED_OK) return 0;
    const Geo g = make_geo(d);
    return g.generic ? gen_ws_bytes(g) : make_ws_layout(g).total;
}

extern "C" int sed_crnn_buffers_init(const sed_dims* d, void* ctx, size_t ctx_bytes, void* ws, size_t ws_bytes, void* stream) {
    SED_TRY(sed_validate_dims(d));
    const Geo g = make_geo(d);
    if (g.generic) return gen_buffers_init(g, ctx, ctx_bytes, ws, ws_bytes, (hipStream_t)stream);
    return SED_OK;                             // the specialised kernel set writes everything before it reads it
}

extern "C" int sed_crnn_ctx_view(const sed_dims* d, const char* name, size_t* offset, size_t* bytes) {
    SED_TRY(sed_validate_dims(d));
    SED_CHECK_ARG(name && offset && bytes, "null argument");
    const Geo g = make_geo(d);
    if (g.generic) return gen_ctx_view(g, name, offset, bytes);
    const CtxLayout L = make_ctx_layout(g);
    const size_t n0 = (size_t)g.B * g.H1 * g.W1 * 64 * 4, n1 = (size_t)g.B * g.H2 * g.W2 * 64 * 4, bt = (size_t)g.B * g.T3;
    struct { const char* n; size_t o, b; } tab[] = {
        {"mom0", L.mom0, 64 * 8}, {"wz0", L.wz0, 64 * 12 * 4}, {"wl0", L.wl0, 64 * 12 * 4}, {"bn0", L.bn0, 1024},
        {"p0", L.p0, n0}, {"y1", L.y1, n0}, {"stat1", L.stat1, 1024}, {"bn1", L.bn1, 1024}, {"p1", L.p1, n1},
        {"y2", L.y2, n1}, {"stat2", L.stat2, 1024}, {"bn2", L.bn2, 1024}, {"p2", L.p2, bt * 64 * 4},
        {"gates0", L.gates[0], bt * 512 * 4}, {"gates1", L.gates[1], bt * 512 * 4},
        {"gru0", L.out[0], bt * 128 * 4}, {"gru1", L.out[1], bt * 128 * 4},
        {"logits_s", L.logits_s, bt * g.NC * 4}, {"den", L.den_sv, (size_t)g.B * g.NC * 4},
    };
    for (auto& t : tab)
        if (strcmp(t.n, name) == 0) { *offset = t.o; *bytes = t.b; return SED_OK; }
    sed_set_error("sed_crnn_ctx_view: unknown buffer '%s'", name);
    return SED_ERR_BAD_ARG;
}

// ---- side stream: weight-gradient work that is off the backward critical path ---------------------
// The dX chain (heads -> GRU -> dgrad2 -> dgrad1 -> block 0) is serial; the GRU dW/db GEMMs and the conv
// wgrads only feed the optimiser.  They are forked onto a helper stream and joined before the call returns, so
// the caller still sees one stream-ordered op and a hipGraph capture records the fork/join as graph edges.
// One helper stream + fork/join event pair exists per (device, caller stream): two host threads (or two models on two
// GPUs of one process) that call in on different streams never share events.  The pool is created under a mutex, on
// first use or - preferably, so that nothing is created while the caller's stream is being captured -
// by sed_stream_prepare(stream).
struct SideStream {
    hipStream_t s = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;          // backward
    hipStream_t s2 = nullptr;                           // second helper: the GRU / heads weight-gradient GEMMs
    hipEvent_t join2 = nullptr;
    bool ok = false;
};
static std::mutex g_side_mu;
static std::map<std::pair<int, hipStream_t>, SideStream*> g_side;
struct ForkHook { void (*fn)(void*) = nullptr; void* user = nullptr; };        // sed_crnn_fork_callback, below
static std::map<std::pair<int, hipStream_t>, ForkHook> g_fork_hooks;
static SideStream& side_stream(hipStream_t caller) {
    static SideStream none;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return none;
    std::lock_guard<std::mutex> lk(g_side_mu);
    SideStream*& slot = g_side[std::make_pair(dev, caller)];
    if (slot == nullptr) {
        slot = new SideStream();
        // (default priority on purpose: a lowest-priority side stream, meant to let the critical-path kernels win the
        // CUs, made the replayed step 70 % slower - 1.79 vs 1.06 ms)
        if (hipStreamCreateWithFlags(&slot->s, hipStreamNonBlocking) == hipSuccess &&
            hipEventCreateWithFlags(&slot->fork, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&slot->join, hipEventDisableTiming) == hipSuccess &&
            hipStreamCreateWithFlags(&slot->s2, hipStreamNonBlocking) == hipSuccess &&
            hipEventCreateWithFlags(&slot->join2, hipEventDisableTiming) == hipSuccess)
            slot->ok = true;
    }
    return *slot;
}
extern "C" int sed_stream_prepare(void* stream) {
    SideStream& sd = side_stream((hipStream_t)stream);
    if (!sd.ok) {
        sed_set_error("sed_stream_prepare: could not create the helper stream / events");
        return SED_ERR_LAUNCH;
    }
    return SED_OK;
}
// Releases what sed_stream_prepare / first use created for `stream` on the current device: both helper streams, the three
// events and a pending fork hook.  The caller guarantees that no work of the library is in flight or being captured on the
// stream (synchronise it first); the stream can be prepared again afterwards.  A caller stream that was never prepared is a
// no-op.  hipStreamDestroy on a helper stream that still has work queued would complete it asynchronously, which is why the
// helpers are synchronised here first - this is the ONE entry point of the library that blocks the host.
extern "C" int sed_stream_release(void* stream) {
    int dev = 0;
    SED_CHECK_HIP(hipGetDevice(&dev));
    SideStream* sd = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_side_mu);
        const auto key = std::make_pair(dev, (hipStream_t)stream);
        auto it = g_side.find(key);
        if (it != g_side.end()) { sd = it->second; g_side.erase(it); }
        g_fork_hooks.erase(key);
    }
    if (sd == nullptr) return SED_OK;
    int rc = SED_OK;
    auto note = [&](hipError_t e) { if (e != hipSuccess && rc == SED_OK) { sed_set_error("sed_stream_release: %s", hipGetErrorString(e)); rc = SED_ERR_LAUNCH; } };
    if (sd->s) { note(hipStreamSynchronize(sd->s)); note(hipStreamDestroy(sd->s)); }
    if (sd->s2) { note(hipStreamSynchronize(sd->s2)); note(hipStreamDestroy(sd->s2)); }
    if (sd->fork) note(hipEventDestroy(sd->fork));
    if (sd->join) note(hipEventDestroy(sd->join));
    if (sd->join2) note(hipEventDestroy(sd->join2));
    delete sd;
    return rc;
}
// One-shot hook for a caller with independent work to run BESIDE the recurrent part of a forward (the waveform front-end:
// the next batch's STFT, features.WaveformFrontEnd): the next sed_crnn_forward on `stream` calls fn(user) on the calling host
// thread after enqueueing its last conv-block kernel and before enqueueing its first recurrence kernel.  The callback forks
// its own stream off `stream` there (event record + wait) and enqueues its work, so the fork sits at that point of the
// enqueue ORDER too: a hipGraph captured around the forward submits its nodes in creation order, and a dependency on a node in
// the middle of another stream's chain that is submitted late starts late (measured: a fork expressed only as an event
// recorded here and waited for after the forward returned started 300 us late, at the first backward kernel).
extern "C" int sed_crnn_fork_callback(void* stream, void (*fn)(void*), void* user) {
    int dev = 0;
    SED_CHECK_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_side_mu);
    const auto key = std::make_pair(dev, (hipStream_t)stream);
    if (fn == nullptr) { g_fork_hooks.erase(key); return SED_OK; }
    ForkHook& h = g_fork_hooks[key];
    h.fn = fn;
    h.user = user;
    return SED_OK;
}
int sed_fork_point(hipStream_t st) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return SED_OK;
    ForkHook h;
    {
        std::lock_guard<std::mutex> lk(g_side_mu);
        auto it = g_fork_hooks.find(std::make_pair(dev, st));
        if (it == g_fork_hooks.end()) return SED_OK;
        h = it->second;
        g_fork_hooks.erase(it);
    }
    if (h.fn) h.fn(h.user);
    return SED_OK;
}
#define SIDE_FORK(main_st)                                          \
    do {                                                            \
        SED_CHECK_HIP(hipEventRecord(sd.fork, (main_st)));          \
        SED_CHECK_HIP(hipStreamWaitEvent(sd.s, sd.fork, 0));        \
    } while (0)
#define SIDE_JOIN(main_st)                                          \
    do {                                                            \
        SED_CHECK_HIP(hipEventRecord(sd.join, sd.s));               \
        SED_CHECK_HIP(hipStreamWaitEvent((main_st), sd.join, 0));   \
    } while (0)

#define CTXF(off) ((float*)((char*)ctx + (off)))
#define CTXD(off) ((double*)((char*)ctx + (off)))
#define WSF(off) ((float*)((char*)ws + (off)))
#define WSD(off) ((double*)((char*)ws + (off)))
#define CTXM(off) ((uint16_t*)((char*)ctx + (off)))

extern "C" int sed_crnn_forward(const sed_dims* d, const float* params, float* bn_running, int64_t* bn_tracked,
                                const float* x, int train, int update_bn, const uint64_t* seed_dev, void* ctx,
                                size_t ctx_bytes, float* strong, float* weak, void* stream) {
    SED_TRY(sed_validate_dims(d));
    SED_CHECK_ARG(params && bn_running && x && ctx, "sed_crnn_forward: null argument");
    // train bit 2 (value 4): this batch's patch moments are already in ctx (sed_crnn_moments ran for it, e.g. during the previous
    // step): no moments launch at the head of the forward
    const int mom_ready = (train & 4) ? 1 : 0;
    train &= 3;
    // strong == weak == NULL (train mode only): the output heads are left to sed_mt_step_backward, which runs them together
    // with the loss and their backward in front of the top layer's backward recurrence
    SED_CHECK_ARG((strong && weak) || (!strong && !weak && train), "sed_crnn_forward: strong and weak must both be given (or both NULL in train mode: heads deferred)");
    const Geo g = make_geo(d);
    const ParamOff P = make_param_off(g, nullptr);
    if (g.generic) {
        SED_CHECK_ARG(!(train && g.p > 0.f) || seed_dev, "sed_crnn_forward: dropout enabled but seed_dev is null");
        hipStream_t st0 = (hipStream_t)stream;
        SideStream& sd0 = side_stream(st0);
        return gen_forward(g, P, params, bn_running, bn_tracked, x, train | (mom_ready ? 4 : 0), update_bn, seed_dev, ctx, ctx_bytes, strong,
                           weak, st0, sd0.ok ? sd0.s : st0, sd0.fork, sd0.join);
    }
    const CtxLayout L = make_ctx_layout(g);
    if (ctx_bytes < L.total) {
        sed_set_error("sed_crnn_forward: ctx has %zu bytes, needs %zu", ctx_bytes, L.total);
        return SED_ERR_WORKSPACE;
    }
    const int use_drop = (train && g.p > 0.f) ? 1 : 0;
    SED_CHECK_ARG(!use_drop || seed_dev, "sed_crnn_forward: dropout enabled but seed_dev is null");
    hipStream_t st = (hipStream_t)stream;
    const int upd = (train && update_bn) ? 1 : 0;
    int64_t* trk[3] = {bn_tracked ? bn_tracked + 0 : nullptr, bn_tracked ? bn_tracked + 1 : nullptr,
                       bn_tracked ? bn_tracked + 2 : nullptr};

    // conv1 / conv2 weights -> [tap][ci][co] + Winograd panels (+ flipped/transposed copies for dgrad); the same threads
    // also zero the fp64 BatchNorm sums of blocks 1 and 2.  In a training forward this rides along in the k_x_moments
    // launch (independent work, extra workgroups) instead of being a launch of its own on the chain.
    const ConvPackArgs pack = {params + P.conv_w[1], params + P.conv_w[2], CTXF(L.wpk1), CTXF(L.wpk2),
                               train ? CTXF(L.wpkT1) : nullptr, train ? CTXF(L.wpkT2) : nullptr, CTXD(L.stat1), train ? 256 : 0};
    // ---- conv block 0 ---------------------------------------------------------------------------
    SED_TRY(launch_blk0_forward(g, x, params + P.conv_w[0], params + P.conv_b[0], params + P.bn_g[0], params + P.bn_b[0],
                                params + P.glu_w[0], params + P.glu_b[0], bn_running + 0, bn_running + 64, trk[0], train,
                                upd, seed_dev, CTXD(L.mom0), CTXD(L.mompart), CTXF(L.wz0), CTXF(L.wl0), CTXF(L.bn0), CTXF(L.p0),
                                use_drop ? CTXM(L.mask0) : nullptr, train ? &pack : nullptr, st, 0, nullptr, nullptr, mom_ready));
    if (!train) SED_TRY(launch_conv_pack(pack, st));      // (training: done by extra workgroups of k_x_moments / k_blk0_prep_pack)
    // ---- conv blocks 1, 2 -----------------------------------------------------------------------
    const float* in = CTXF(L.p0);
    const size_t wpk[3] = {0, L.wpk1, L.wpk2}, yo[3] = {0, L.y1, L.y2}, so[3] = {0, L.stat1, L.stat2},
                 bo[3] = {0, L.bn1, L.bn2}, po[3] = {0, L.p1, L.p2}, mo[3] = {L.mask0, L.mask1, L.mask2};
    const int Hs[3] = {0, g.H1, g.H2}, Ws[3] = {0, g.W1, g.W2};
    for (int i = 1; i <= 2; ++i) {
        SED_TRY(launch_conv_fwd(in, CTXF(wpk[i]), params + P.conv_b[i], CTXF(yo[i]), train ? CTXD(so[i]) : nullptr, 0, g.B,
                                Hs[i], Ws[i], st));
        SED_TRY(launch_glu_pool_fwd(CTXF(yo[i]), CTXD(so[i]), (double)g.B * Hs[i] * Ws[i], params + P.bn_g[i], params + P.bn_b[i],
                                    bn_running + (2 * i) * 64, bn_running + (2 * i + 1) * 64, trk[i], train, upd, g.eps, g.mom,
                                    CTXF(bo[i]), params + P.glu_w[i], params + P.glu_b[i], CTXF(po[i]), g.B, Hs[i], Ws[i], i,
                                    use_drop, g.p, seed_dev, use_drop ? CTXM(mo[i]) : nullptr, st));
        in = CTXF(po[i]);
    }
    // ---- BiGRU ----------------------------------------------------------------------------------
    SED_TRY(sed_fork_point(st));
    int nin = 64;
    for (int l = 0; l < g.L; ++l) {
        // the input projection x W_ih^T + b_ih runs inside the recurrence kernel (gi only exists in LDS)
        SED_TRY(launch_gru_fwd(in, nin, params + P.w_ih[l][0], params + P.w_ih[l][1], params + P.b_ih[l][0], params + P.b_ih[l][1],
                               params + P.w_hh[l][0], params + P.w_hh[l][1], params + P.b_hh[l][0], params + P.b_hh[l][1],
                               CTXF(L.out[l]), train ? CTXF(L.gates[l]) : nullptr, g.B, g.T3, st));
        in = CTXF(L.out[l]);
        nin = 128;
    }
    // ---- heads ----------------------------------------------------------------------------------
    if (strong == nullptr) return SED_OK;                 // deferred (sed_mt_step_backward)
    SED_TRY(launch_heads_fwd(in, params + P.dense_w, params + P.dense_b, params + P.soft_w, params + P.soft_b, strong, weak,
                             train ? CTXF(L.strong_sv) : nullptr, train ? CTXF(L.weak_sv) : nullptr, CTXF(L.logits_s),
                             CTXF(L.den_sv), g.B, g.T3, g.NC, use_drop, g.p, seed_dev, st));
    return SED_OK;
}

// The train-mode BatchNorm statistics of conv block 0 come from the 9 + 45 first / second moments of the 3x3 input patch
// (blk0.hip): a function of the BATCH only, not of any parameter.  A caller that has the next batch resident while the current
// step runs (features.WaveformFrontEnd: one batch ahead; a constant / resident batch) computes them THEN - beside the
// recurrences, which leave most of the chip idle - and passes train | 4 to the forward that consumes them: 16 us (B = 24) of
// launch + round trip leave the head of the step's critical chain.  Same kernel, same partials, same order: bit-identical.
extern "C" int sed_crnn_moments(const sed_dims* d, const float* x, void* ctx, size_t ctx_bytes, void* stream) {
    SED_TRY(sed_validate_dims(d));
    SED_CHECK_ARG(x && ctx, "sed_crnn_moments: null argument");
    const Geo g = make_geo(d);
    double* mompart = nullptr;
    if (g.generic) {
        SED_TRY(gen_mompart(g, ctx, ctx_bytes, &mompart));
    } else {
        const CtxLayout L = make_ctx_layout(g);
        if (ctx_bytes < L.total) {
            sed_set_error("sed_crnn_moments: ctx has %zu bytes, needs %zu", ctx_bytes, L.total);
            return SED_ERR_WORKSPACE;
        }
        mompart = CTXD(L.mompart);
    }
    return launch_x_moments(g, x, mompart, nullptr, (hipStream_t)stream);
}

static int crnn_backward_impl(const sed_dims* d, const float* params, const float* x, const uint64_t* seed_dev,
                              void* ctx, size_t ctx_bytes, const float* d_strong, const float* d_weak, float* grads,
                              void* ws, size_t ws_bytes, int parts, void* stream, const HeadsLoss* hl,
                              const HeadsOut* ho = nullptr) {
    SED_TRY(sed_validate_dims(d));
    SED_CHECK_ARG(params && x && ctx && (hl || (d_strong && d_weak)) && grads && ws, "sed_crnn_backward: null argument");
    SED_CHECK_ARG(parts == 1 || parts == 2 || parts == 3 || parts == 5 || parts == 8,
                  "sed_crnn_backward: parts must be 1, 2, 3, 5 or 8");
    const Geo g = make_geo(d);
    const ParamOff P = make_param_off(g, nullptr);
    if (g.generic) {
        SED_CHECK_ARG(!(g.p > 0.f) || seed_dev, "sed_crnn_backward: dropout enabled but seed_dev is null");
        hipStream_t st0 = (hipStream_t)stream;
        SideStream& sd0 = side_stream(st0);
        // (debug bit 30: no helper stream - every kernel on the caller's stream, for near-solo kernel times under rocprofv3)
        return gen_backward(g, P, params, x, seed_dev, ctx, ctx_bytes, d_strong, d_weak, grads, ws, ws_bytes, parts, st0,
                            (sd0.ok && !(g_sed_debug & 1073741824)) ? sd0.s : st0, sd0.fork, sd0.join, (sd0.ok && (g_sed_debug & 2048)) ? sd0.s2 : nullptr, sd0.join2, hl, ho);
    }
    const CtxLayout L = make_ctx_layout(g);
    const WsLayout W = make_ws_layout(g);
    if (ctx_bytes < L.total || ws_bytes < W.total) {
        sed_set_error("sed_crnn_backward: ctx %zu/%zu bytes, ws %zu/%zu bytes", ctx_bytes, L.total, ws_bytes, W.total);
        return SED_ERR_WORKSPACE;
    }
    const int use_drop = (g.p > 0.f) ? 1 : 0;
    SED_CHECK_ARG(!use_drop || seed_dev, "sed_crnn_backward: dropout enabled but seed_dev is null");
    hipStream_t st = (hipStream_t)stream;
    const int BT = g.B * g.T3;
    SideStream& sd = side_stream(st);
    hipStream_t ss = (sd.ok && !(g_sed_debug & 1073741824)) ? sd.s : st;       // without a side stream everything stays on the caller's
    const bool defer_gru_w = (parts & 4) != 0;           // parts == 5: the caller runs them later (parts == 8)
    bool forked = false, forked2 = false;

    // weight + bias gradients of a GRU layer, both directions (split-K MFMA GEMMs, low occupancy):
    //   dW_ih[g][i] = sum_bt dgi[bt][g] input[bt][i],  db_ih[g] = sum_bt dgi[bt][g]
    //   dW_hh[g][j] = sum_bt dgh[bt][g] hprev[bt][j],  db_hh[g] = sum_bt dgh[bt][g]
    auto gru_weight_grads_layer = [&](int l, hipStream_t s2) -> int {
        const int nin = (l == 0) ? 64 : 128;
        const float* input = (l == 0) ? CTXF(L.p2) : CTXF(L.out[l - 1]);
        GemmBatch gb;
        gb.n_prob = 4; gb.splits = SED_GRU_SPLITK; gb.part = WSF(W.gemm_part); gb.part_floats = gemm_part_floats(4, SED_GRU_SPLITK, 192, 129); gb.part_stride = 0;
        for (int dir = 0; dir < 2; ++dir) {
            gb.p[2 * dir] = gemm_prob(WSF(W.dgi[l]) + dir * 192, 1, 384, input, nin, 1, grads + P.w_ih[l][dir], nin, 192, nin, BT);
            gb.p[2 * dir].Cones = grads + P.b_ih[l][dir];
            gb.p[2 * dir + 1] = gemm_prob(WSF(W.dgh[l]) + dir * 192, 1, 384, WSF(W.hprev[l]) + dir * 64, 128, 1,
                                          grads + P.w_hh[l][dir], 64, 192, 64, BT);
            gb.p[2 * dir + 1].Cones = grads + P.b_hh[l][dir];
        }
        return launch_gemm_batch(gb, s2);
    };
    auto gru_weight_grads = [&](hipStream_t s2) -> int {
        for (int l = g.L - 1; l >= 0; --l) SED_TRY(gru_weight_grads_layer(l, s2));
        return SED_OK;
    };
    // Debug bit 14 (timing experiment, measured twice and rejected): each layer's weight-gradient GEMMs right behind that
    // layer's recurrence kernel on the side stream - the upper layer's next to the lower layer's recurrence (48 workgroups
    // on 256 CUs) - instead of between the two conv weight-gradient kernels.  Round 2, with the four-SIMD recurrence:
    // 0.816 ms against 0.766 ms per step.  Every extra cross-stream edge of the captured graph becomes a completion
    // signal between two hardware queues, and two of them on the recurrence chain cost more than the GEMMs' 40 us of
    // otherwise idle GPU time give back.
    const bool early_gru_w = parts == 3 && sd.ok && (g_sed_debug & 16384);

    // ho != null: the forward left the output heads to this call (sed_mt_step_backward).  Fused form (hfuse.h): heads forward +
    // loss + heads backward run as the prologue phase of the top layer's backward recurrence, the meters' clip sums, the
    // step-state advance and the head weight gradients' column sum in k_heads_fin where the column sum alone used to be.
    // Otherwise (debug bit 24, T / 8 > 128, gradient outputs asked for): k_heads_fwd here, then the two-kernel form.
    const int head_cols = 2 * (g.NC * 128 + g.NC);
    const bool fuse = ho && hl && (parts & 1) && heads_fusable(64, g.T3) && !(g_sed_debug & 16777216) && !hl->d_strong_out && !hl->d_weak_out;
    auto heads_colsum = [&](hipStream_t s2) -> int {
        if (fuse) return launch_heads_fin(WSF(W.heads_part), grads + P.dense_w, g.B, g.T3, g.NC, head_cols, *hl, s2);
        return launch_heads_colsum(WSF(W.heads_part), grads + P.dense_w, g.B, g.NC, s2);
    };
    const bool defer_colsum = ((parts & 2) && sd.ok) || defer_gru_w;
    if (parts & 1) {
    // ---- heads ----------------------------------------------------------------------------------
    const float* h_last = CTXF(L.out[g.L - 1]);
    if (ho && !fuse)
        SED_TRY(launch_heads_fwd(h_last, params + P.dense_w, params + P.dense_b, params + P.soft_w, params + P.soft_b, ho->strong, ho->weak,
                                 CTXF(L.strong_sv), CTXF(L.weak_sv), CTXF(L.logits_s), CTXF(L.den_sv), g.B, g.T3, g.NC, use_drop, g.p,
                                 seed_dev, st));
    if (!fuse)
    SED_TRY(launch_heads_bwd(h_last, params + P.dense_w, params + P.soft_w, CTXF(L.strong_sv), CTXF(L.weak_sv),
                             CTXF(L.logits_s), CTXF(L.den_sv), d_strong, d_weak, WSF(W.d_out), WSF(W.heads_part),
                             grads + P.dense_w, grads + P.dense_b, grads + P.soft_w, grads + P.soft_b, g.B, g.T3, g.NC,
                             use_drop, g.p, seed_dev, (parts & 2) ? WSD(W.bwd_acc) : nullptr, 2 * SED_GLUACC_N + 2 * 64 * 10,
                             defer_colsum ? 1 : 0, hl, st));
    // ---- BiGRU ----------------------------------------------------------------------------------
    // The gradient w.r.t. each layer's input is produced INSIDE the recurrence kernel (two extra waves, one block of
    // steps behind), as two direction planes [2][B*T'][nin] that the consumer adds while loading: the layer below's
    // recurrence kernel, or k_glu_pool_bwd for layer 0 (planes in W.dp2).
    const float* d_cur = WSF(W.d_out);
    const float* d_cur2 = nullptr;
    for (int l = g.L - 1; l >= 0; --l) {
        const int nin = (l == 0) ? 64 : 128;
        float* d_in = (l == 0) ? WSF(W.dp2) : WSF(W.d_in);
        if (fuse && l == g.L - 1) {
            HeadsFuse hf = {};
            hf.wd = params + P.dense_w; hf.strong = ho->strong; hf.weak = ho->weak; hf.part = WSF(W.heads_part);
            hf.NC = g.NC; hf.use_drop = use_drop; hf.p_drop = g.p; hf.seed = seed_dev;
            hf.zero = (parts & 2) ? WSD(W.bwd_acc) : nullptr; hf.n_zero = (parts & 2) ? 2 * SED_GLUACC_N + 2 * 64 * 10 : 0;
            hf.hl = *hl;
            SED_TRY(launch_gru_bwd_heads(CTXF(L.out[l]), CTXF(L.gates[l]), params + P.w_hh[l][0], params + P.w_hh[l][1],
                                         params + P.w_ih[l][0], params + P.w_ih[l][1], nin, WSF(W.dgi[l]), WSF(W.dgh[l]),
                                         WSF(W.hprev[l]), d_in, g.B, g.T3, hf, st));
            // the meters / step-state advance (+ column sum unless deferred to the side stream or a parts = 8 call)
            if (!defer_colsum) SED_TRY(heads_colsum(st));
            else if (defer_gru_w) SED_TRY(launch_heads_fin(WSF(W.heads_part), grads + P.dense_w, g.B, g.T3, g.NC, 0, *hl, st));
        } else
        SED_TRY(launch_gru_bwd(d_cur, d_cur2, CTXF(L.out[l]), CTXF(L.gates[l]), params + P.w_hh[l][0], params + P.w_hh[l][1],
                               params + P.w_ih[l][0], params + P.w_ih[l][1], nin, WSF(W.dgi[l]), WSF(W.dgh[l]),
                               WSF(W.hprev[l]), d_in, g.B, g.T3, st));
        d_cur = d_in;
        d_cur2 = d_in + (size_t)BT * nin;
        if (early_gru_w) {
            SIDE_FORK(st);
            forked = true;
            if (l == g.L - 1) SED_TRY(heads_colsum(ss));
            SED_TRY(gru_weight_grads_layer(l, ss));
        }
    }
    }
    // parts == 1 (data-parallel: the GRU + heads gradient bucket must be complete when this call returns so that
    // its all-reduce can start): the GEMMs follow the dX chain on the caller's stream.  parts == 3: they are
    // deferred to the side stream of the conv-block backward below, where they overlap k_glu_pool_bwd.
    if (parts == 1) SED_TRY(gru_weight_grads(st));
    if (parts == 8) {
        // the weight-gradient tail of a parts == 5 call: head column sum + every GRU dW / db, on the CALLER's stream (which
        // a data-parallel host makes a second stream, so that this bucket and its all-reduce overlap the conv backward)
        SED_TRY(launch_heads_colsum(WSF(W.heads_part), grads + P.dense_w, g.B, g.NC, st));
        SED_TRY(gru_weight_grads(st));
        return SED_OK;
    }
    if (!(parts & 2)) {
        if (forked) SIDE_JOIN(st);
        return SED_OK;
    }
    // ---- conv blocks 2, 1 -----------------------------------------------------------------------
    const size_t wpkT[3] = {0, L.wpkT1, L.wpkT2}, yo[3] = {0, L.y1, L.y2}, bo[3] = {0, L.bn1, L.bn2};
    const size_t pin[3] = {0, L.p0, L.p1}, gacc[3] = {0, W.gluacc1, W.gluacc2}, mo[3] = {L.mask0, L.mask1, L.mask2};
    const size_t dzo[3] = {0, W.dz1, W.dz2}, dpo[3] = {W.dp0, W.dp1, W.dp2};
    const int Hs[3] = {0, g.H1, g.H2}, Wd[3] = {0, g.W1, g.W2};
    // every fp64 accumulator of the conv-block backward: zeroed by k_heads_bwd when this call also ran part 1
    if (!(parts & 1)) SED_CHECK_HIP(hipMemsetAsync(WSD(W.bwd_acc), 0, (2 * SED_GLUACC_N + 2 * 64 * 10) * sizeof(double), st));
    // Stream schedule (kernel timeline of one step, tools/timeline.py): the dgrad chain is the critical path.
    //   main: glu2_bwd  prep | dgrad2            | glu1_bwd  prep | dgrad1          | blk0_bwd  finalize |
    //   side:                | wgrad2  GRU dW/db |                | wgrad1  reduce                       | join
    // (Forking right behind each recurrence kernel - the upper layer's GEMMs next to the lower layer's 48-workgroup
    // recurrence - measured slower still, 1.043 ms: every event record splits the critical chain.)
    // (Forking before glu2_bwd so that the GRU GEMMs run first and wgrad1 starts on time measured slower, 1.061 vs
    // 1.029 ms: they then compete with the critical-path kernels glu2_bwd / dgrad2 / glu1_bwd.)
    // (Same experiment again with the Winograd convolutions, where the side stream has become the tail of the step: still
    // slower, 28.0 vs 28.3 k clips/s.)
    // (Starting wgrad1 only after dgrad1, next to the VALU-bound k_blk0_bwd, measured the same: dgrad1 drops from
    // 175 to 93 us but k_blk0_bwd, left with one wave per SIMD beside the wgrad wave, goes from 86 to 177 us.)
    // BatchNorm-backward coefficients: derived by the conv dgrad / wgrad kernels themselves from the reduction sums (no
    // 1-workgroup k_bn_bwd_prep + launch gap between k_glu_pool_bwd and the dgrad, twice per step); bit 9 of the debug
    // knob, or any of the A/B conv kernels, brings the separate kernel back
#ifdef SED_AB
    const bool fuse_prep = (g_sed_debug & (4 | 8 | 64 | 128 | 512)) == 0;
#else
    const bool fuse_prep = (g_sed_debug & 512) == 0;
#endif
    BnBwdPrepArgs prep[3] = {};
    for (int i = 2; i >= 1; --i) {
        const BnBwdPrepArgs* pp = fuse_prep ? &prep[i] : nullptr;
        // (the BatchNorm-backward coefficients and the block's parameter gradients are produced by the last workgroup
        // of k_glu_pool_bwd: no separate 1-workgroup kernel between it and the conv dgrad / wgrad)
        SED_TRY(launch_glu_pool_bwd(CTXF(yo[i]), CTXF(bo[i]), params + P.glu_w[i], params + P.glu_b[i], WSF(dpo[i]),
                                    i == 2 ? WSF(dpo[i]) + (size_t)BT * 64 : nullptr,
                                    WSF(dzo[i]), WSD(gacc[i]), 0, g.B, Hs[i], Wd[i], i, use_drop, g.p, CTXM(mo[i]),
                                    params + P.bn_g[i], WSF(W.coef[i]), grads + P.bn_g[i], grads + P.bn_b[i],
                                    grads + P.glu_w[i], grads + P.glu_b[i], grads + P.conv_b[i], fuse_prep ? &prep[i] : nullptr, st));
        if (i == 2) {
            // The fork event is recorded here, but the dgrad - the critical chain - is CAPTURED FIRST: the graph executor keeps the
            // first-captured child of a node on its parent's hardware queue; with the helper stream's wgrad captured first the
            // dgrad hopped to another queue and started 10 us after k_glu_pool_bwd8 had finished
            // (profiles/r05b_mt-f32_step_timeline.txt: 447.6 -> 457.7 us).  Both still depend on the same event.
            if (sd.ok) SED_CHECK_HIP(hipEventRecord(sd.fork, st));
            SED_TRY(launch_conv_dgrad(WSF(dzo[i]), CTXF(yo[i]), WSF(W.coef[i]), CTXF(wpkT[i]), WSF(dpo[i - 1]), g.B, Hs[i], Wd[i], pp, st));
            if (sd.ok) { SED_CHECK_HIP(hipStreamWaitEvent(sd.s, sd.fork, 0)); forked = true; }
            SED_TRY(launch_conv_wgrad(WSF(dzo[i]), CTXF(yo[i]), WSF(W.coef[i]), CTXF(pin[i]), WSF(W.wg_part), W.wgrad_blocks,
                                      grads + P.conv_w[i], g.B, Hs[i], Wd[i], pp, ss));
            if (parts == 3 && !early_gru_w) {
                // the head weight-gradient column sum (deferred from part 1): behind wgrad2, long before the tail of the step
                // (queued right in front of wgrad1 it sat 60 us behind the persistent dgrad kernel and held wgrad1 back; at
                // the very end of the side stream it was 4 us on the step's tail)
                // (debug bit 11: on a helper stream of their OWN.  Queued on the conv-wgrad helper they sit between wgrad2 and
                // wgrad1 and hold wgrad1 - the tail of the step - back by ~80 us (r02_a step timeline: wgrad1 starts at 678 us,
                // its inputs are ready at 594 us); giving them their own stream measured SLOWER all the same - 0.829 vs
                // 0.815 ms (fp32), 0.775 vs 0.742 ms (bf16 operands): the step is throughput-bound, a third stream only
                // takes CUs from the dgrad / block-0 chain.)
                hipStream_t sg = ss;
                if (sd.ok && (g_sed_debug & 2048)) {
                    SED_CHECK_HIP(hipStreamWaitEvent(sd.s2, sd.fork, 0));
                    sg = sd.s2;
                    forked2 = true;
                }
                if (sd.ok) SED_TRY(heads_colsum(sg));
                SED_TRY(gru_weight_grads(sg));
            }
        } else {
            if (sd.ok) { SIDE_FORK(st); forked = true; }
            SED_TRY(launch_conv_dgrad(WSF(dzo[i]), CTXF(yo[i]), WSF(W.coef[i]), CTXF(wpkT[i]), WSF(dpo[i - 1]), g.B, Hs[i], Wd[i], pp, st));
            SED_TRY(launch_conv_wgrad(WSF(dzo[i]), CTXF(yo[i]), WSF(W.coef[i]), CTXF(pin[i]), WSF(W.wg_part), W.wgrad_blocks,
                                      grads + P.conv_w[i], g.B, Hs[i], Wd[i], pp, ss));
        }
    }
    // ---- conv block 0 ---------------------------------------------------------------------------
    SED_TRY(launch_blk0_backward(g, x, params + P.conv_w[0], params + P.conv_b[0], params + P.bn_g[0], params + P.bn_b[0],
                                 params + P.glu_w[0], CTXM(L.mask0), CTXD(L.mom0), CTXF(L.wz0), CTXF(L.wl0), CTXF(L.bn0),
                                 WSF(W.dp0), WSD(W.de0), 0, grads + P.conv_w[0], grads + P.conv_b[0], grads + P.bn_g[0],
                                 grads + P.bn_b[0], grads + P.glu_w[0], grads + P.glu_b[0], st));
    if (forked) SIDE_JOIN(st);
    if (forked2) {
        SED_CHECK_HIP(hipEventRecord(sd.join2, sd.s2));
        SED_CHECK_HIP(hipStreamWaitEvent(st, sd.join2, 0));
    }
    return SED_OK;
}

extern "C" int sed_crnn_backward(const sed_dims* d, const float* params, const float* x, const uint64_t* seed_dev,
                                 void* ctx, size_t ctx_bytes, const float* d_strong, const float* d_weak, float* grads,
                                 void* ws, size_t ws_bytes, int parts, void* stream) {
    return crnn_backward_impl(d, params, x, seed_dev, ctx, ctx_bytes, d_strong, d_weak, grads, ws, ws_bytes, parts, stream, nullptr);
}

extern "C" int sed_mt_loss_backward(const sed_dims* d, const float* params, const float* x, const uint64_t* seed_dev,
                                    void* ctx, size_t ctx_bytes, const float* strong_ema, const float* weak_ema,
                                    const float* target, int weak_lo, int weak_hi, int strong_lo, int strong_hi,
                                    sed_step_state* state_dev, int advance_state, float* losses, float* d_strong,
                                    float* d_weak, float* grads, void* ws, size_t ws_bytes, int parts, void* stream) {
    SED_CHECK_ARG(d && strong_ema && weak_ema && target && state_dev && losses, "sed_mt_loss_backward: null argument");
    SED_CHECK_ARG(parts == 1 || parts == 3 || parts == 5, "sed_mt_loss_backward: parts must include the heads (1, 3 or 5)");
    SED_CHECK_ARG(weak_lo >= 0 && weak_hi <= d->B && weak_lo <= weak_hi && strong_lo >= 0 && strong_hi <= d->B &&
                      strong_lo <= strong_hi, "sed_mt_loss_backward: bad mask range");
    HeadsLoss hl = {strong_ema, weak_ema, target, weak_lo, weak_hi, strong_lo, strong_hi, state_dev, losses, d_strong, d_weak,
                    advance_state ? state_dev : nullptr};
    return crnn_backward_impl(d, params, x, seed_dev, ctx, ctx_bytes, nullptr, nullptr, grads, ws, ws_bytes, parts, stream, &hl);
}

extern "C" int sed_mt_step_backward(const sed_dims* d, const float* params, const float* x, const uint64_t* seed_dev,
                                    void* ctx, size_t ctx_bytes, float* strong, float* weak, const float* strong_ema,
                                    const float* weak_ema, const float* target, int weak_lo, int weak_hi, int strong_lo,
                                    int strong_hi, sed_step_state* state_dev, int advance_state, float* losses, float* d_strong,
                                    float* d_weak, float* grads, void* ws, size_t ws_bytes, int parts, void* stream) {
    SED_CHECK_ARG(d && strong && weak && strong_ema && weak_ema && target && state_dev && losses, "sed_mt_step_backward: null argument");
    SED_CHECK_ARG(parts == 1 || parts == 3 || parts == 5, "sed_mt_step_backward: parts must include the heads (1, 3 or 5)");
    SED_CHECK_ARG(weak_lo >= 0 && weak_hi <= d->B && weak_lo <= weak_hi && strong_lo >= 0 && strong_hi <= d->B &&
                      strong_lo <= strong_hi, "sed_mt_step_backward: bad mask range");
    HeadsLoss hl = {strong_ema, weak_ema, target, weak_lo, weak_hi, strong_lo, strong_hi, state_dev, losses, d_strong, d_weak,
                    advance_state ? state_dev : nullptr};
    HeadsOut ho = {strong, weak};
    return crnn_backward_impl(d, params, x, seed_dev, ctx, ctx_bytes, nullptr, nullptr, grads, ws, ws_bytes, parts, stream, &hl, &ho);
}

extern "C" int sed_kernel_replay(const char* name, const sed_dims* d, const float* params, const float* x,
                                 const uint64_t* seed_dev, void* ctx, size_t ctx_bytes, float* grads, void* ws,
                                 size_t ws_bytes, void* stream) {
    SED_TRY(sed_validate_dims(d));
    SED_CHECK_ARG(name && params && x && ctx && grads && ws, "sed_kernel_replay: null argument");
    const Geo g = make_geo(d);
    if (g.generic) {
        sed_set_error("sed_kernel_replay: only the C = 64 / H = 64 / fp32 kernel set is replayable");
        return SED_ERR_UNSUPPORTED;
    }
    const ParamOff P = make_param_off(g, nullptr);
    const CtxLayout L = make_ctx_layout(g);
    const WsLayout W = make_ws_layout(g);
    if (ctx_bytes < L.total || ws_bytes < W.total) {
        sed_set_error("sed_kernel_replay: buffers too small");
        return SED_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const int use_drop = (g.p > 0.f) ? 1 : 0;
    const int BT = g.B * g.T3;
    const size_t wpk[3] = {0, L.wpk1, L.wpk2}, wpkT[3] = {0, L.wpkT1, L.wpkT2}, yo[3] = {0, L.y1, L.y2},
                 so[3] = {0, L.stat1, L.stat2}, bo[3] = {0, L.bn1, L.bn2}, po[3] = {L.p0, L.p1, L.p2},
                 mo[3] = {L.mask0, L.mask1, L.mask2};
    const size_t dzo[3] = {0, W.dz1, W.dz2}, dpo[3] = {W.dp0, W.dp1, W.dp2}, gacc[3] = {0, W.gluacc1, W.gluacc2};
    const int Hs[3] = {0, g.H1, g.H2}, Wd[3] = {0, g.W1, g.W2};
    auto is = [&](const char* n) { return strcmp(name, n) == 0; };
    if (is("x_moments")) return launch_x_moments(g, x, CTXD(L.mompart), nullptr, st);
    if (is("blk0_fwd")) {
        const int tpc = (g.H1 + 3) / 4;
        (void)tpc;
        return launch_blk0_forward(g, x, params + P.conv_w[0], params + P.conv_b[0], params + P.bn_g[0], params + P.bn_b[0],
                                   params + P.glu_w[0], params + P.glu_b[0], WSF(W.coef[1]), WSF(W.coef[1]) + 64, nullptr, 1, 0,
                                   seed_dev, CTXD(L.mom0), CTXD(L.mompart), CTXF(L.wz0), CTXF(L.wl0), CTXF(L.bn0), CTXF(L.p0),
                                   use_drop ? CTXM(L.mask0) : nullptr, nullptr, st, 1);
    }
    for (int i = 1; i <= 2; ++i) {
        char nm[32];
        snprintf(nm, sizeof nm, "conv%d_fwd", i);
        if (is(nm)) return launch_conv_fwd(CTXF(po[i - 1]), CTXF(wpk[i]), params + P.conv_b[i], CTXF(yo[i]), CTXD(so[i]), 1, g.B, Hs[i], Wd[i], st);
        snprintf(nm, sizeof nm, "glu%d_fwd", i);
        if (is(nm)) return launch_glu_pool_fwd(CTXF(yo[i]), CTXD(so[i]), (double)g.B * Hs[i] * Wd[i], params + P.bn_g[i], params + P.bn_b[i],
                                               WSF(W.coef[1]), WSF(W.coef[1]) + 64, nullptr, 1, 0, g.eps, g.mom, CTXF(bo[i]),
                                               params + P.glu_w[i], params + P.glu_b[i], CTXF(po[i]), g.B, Hs[i], Wd[i], i, use_drop,
                                               g.p, seed_dev, use_drop ? CTXM(mo[i]) : nullptr, st);
        snprintf(nm, sizeof nm, "glu%d_bwd", i);
        if (is(nm)) return launch_glu_pool_bwd(CTXF(yo[i]), CTXF(bo[i]), params + P.glu_w[i], params + P.glu_b[i], WSF(dpo[i]), i == 2 ? WSF(dpo[i]) + (size_t)BT * 64 : nullptr, WSF(dzo[i]), WSD(gacc[i]), 1, g.B, Hs[i], Wd[i], i, use_drop, g.p, CTXM(mo[i]),
                                               params + P.bn_g[i], WSF(W.coef[i]), grads + P.bn_g[i], grads + P.bn_b[i], grads + P.glu_w[i], grads + P.glu_b[i], grads + P.conv_b[i], nullptr, st);
        snprintf(nm, sizeof nm, "conv%d_wgrad", i);
        if (is(nm)) return launch_conv_wgrad(WSF(dzo[i]), CTXF(yo[i]), WSF(W.coef[i]), CTXF(po[i - 1]), WSF(W.wg_part), W.wgrad_blocks, grads + P.conv_w[i], g.B, Hs[i], Wd[i], nullptr, st);
        snprintf(nm, sizeof nm, "conv%d_dgrad", i);
        if (is(nm)) return launch_conv_dgrad(WSF(dzo[i]), CTXF(yo[i]), WSF(W.coef[i]), CTXF(wpkT[i]), WSF(dpo[i - 1]), g.B, Hs[i], Wd[i], nullptr, st);
    }
    for (int l = 0; l < g.L; ++l) {
        char nm[32];
        snprintf(nm, sizeof nm, "gru%d_fwd", l);
        if (is(nm)) return launch_gru_fwd(l == 0 ? CTXF(L.p2) : CTXF(L.out[l - 1]), l == 0 ? 64 : 128, params + P.w_ih[l][0], params + P.w_ih[l][1],
                                          params + P.b_ih[l][0], params + P.b_ih[l][1], params + P.w_hh[l][0], params + P.w_hh[l][1],
                                          params + P.b_hh[l][0], params + P.b_hh[l][1], CTXF(L.out[l]), CTXF(L.gates[l]), g.B, g.T3, st);
        snprintf(nm, sizeof nm, "gru%d_bwd", l);
        if (is(nm)) {
            const int nin = (l == 0) ? 64 : 128;
            const bool top = (l == g.L - 1);
            return launch_gru_bwd(top ? WSF(W.d_out) : WSF(W.d_in), top ? nullptr : WSF(W.d_in) + (size_t)BT * 128, CTXF(L.out[l]),
                                  CTXF(L.gates[l]), params + P.w_hh[l][0], params + P.w_hh[l][1], params + P.w_ih[l][0],
                                  params + P.w_ih[l][1], nin, WSF(W.dgi[l]), WSF(W.dgh[l]), WSF(W.hprev[l]),
                                  l == 0 ? WSF(W.dp2) : WSF(W.d_in), g.B, g.T3, st);
        }
    }
    if (is("heads_fwd"))
        return launch_heads_fwd(CTXF(L.out[g.L - 1]), params + P.dense_w, params + P.dense_b, params + P.soft_w, params + P.soft_b,
                                CTXF(L.strong_sv), CTXF(L.weak_sv), nullptr, nullptr, CTXF(L.logits_s), CTXF(L.den_sv), g.B, g.T3, g.NC,
                                use_drop, g.p, seed_dev, st);
    if (is("blk0_bwd"))
        return launch_blk0_backward(g, x, params + P.conv_w[0], params + P.conv_b[0], params + P.bn_g[0], params + P.bn_b[0],
                                    params + P.glu_w[0], CTXM(L.mask0), CTXD(L.mom0), CTXF(L.wz0), CTXF(L.wl0), CTXF(L.bn0), WSF(W.dp0),
                                    WSD(W.de0), 1, grads + P.conv_w[0], grads + P.conv_b[0], grads + P.bn_g[0], grads + P.bn_b[0],
                                    grads + P.glu_w[0], grads + P.glu_b[0], st);
    (void)BT;
    sed_set_error("sed_kernel_replay: unknown kernel '%s'", name);
    return SED_ERR_BAD_ARG;
}
