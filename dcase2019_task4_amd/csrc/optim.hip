// optim.hip - fused Adam + EMA-teacher update over the flat parameter buffer, and the device-side
// step state (step counters, consistency ramp, EMA alpha, Adam bias corrections, dropout seeds).
//
// Reference ops: torch.optim.Adam(lr=1e-3, betas=(0.9,0.999)).step() (baseline/main.py:154,289-290)
// followed by update_ema_variables(model, ema_model, 0.999, global_step) (main.py:45-49,156-157).
// The reference launches ~38 x 6 tiny kernels for this; here it is ONE pass of 5 reads + 4 writes
// per parameter (HBM-bound, 7.7 MB for the base model), with every per-step scalar read from
// device memory so a captured hipGraph can be replayed without host involvement.
#include <math.h>
#include "common.h"
#include "kernels.h"

template <bool EMA>
__global__ __launch_bounds__(256) void k_adam_ema(int64_t n, float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, float* __restrict__ pe,
                                                   const sed_step_state* __restrict__ st, float grad_scale) {
    const float b1 = (float)st->beta1, b2 = (float)st->beta2, eps = (float)st->eps;
    const float step_size = st->adam_step_size, sqrt_bc2 = st->adam_sqrt_bc2, alpha = st->ema_alpha;
    const int64_t n4 = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 P = ((float4*)p)[i], G = ((const float4*)g)[i], M = ((float4*)m)[i], V = ((float4*)v)[i];
        float4 E = EMA ? ((float4*)pe)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
#define ADAM1(x)                                                         \
    {                                                                    \
        const float gg = G.x * grad_scale;                               \
        M.x = b1 * M.x + (1.0f - b1) * gg;                               \
        V.x = b2 * V.x + (1.0f - b2) * gg * gg;                          \
        P.x = P.x - step_size * (M.x / (sqrtf(V.x) / sqrt_bc2 + eps));   \
        E.x = alpha * E.x + (1.0f - alpha) * P.x;                        \
    }
        ADAM1(x) ADAM1(y) ADAM1(z) ADAM1(w)
#undef ADAM1
        ((float4*)p)[i] = P; ((float4*)m)[i] = M; ((float4*)v)[i] = V;
        if (EMA) ((float4*)pe)[i] = E;
    }
    // tail (n not a multiple of 4)
    const int64_t i = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const float gg = g[i] * grad_scale;
        const float mm = b1 * m[i] + (1.0f - b1) * gg;
        const float vv = b2 * v[i] + (1.0f - b2) * gg * gg;
        const float pp = p[i] - step_size * (mm / (sqrtf(vv) / sqrt_bc2 + eps));
        m[i] = mm; v[i] = vv; p[i] = pp;
        if (EMA) pe[i] = alpha * pe[i] + (1.0f - alpha) * pp;
    }
}

__global__ __launch_bounds__(256) void k_ema(int64_t n, const float* __restrict__ p, float* __restrict__ pe, float alpha) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        pe[i] = alpha * pe[i] + (1.0f - alpha) * p[i];
}

// (splitmix64, step_state_derive*, step_state_advance_early: kernels.h - the heads backward advances the state too)

__global__ void k_step_state_init(sed_step_state* s, uint64_t base_seed, int64_t rampup_length, double lr, double beta1,
                                  double beta2, double eps, double ema_decay, double max_cons_cost) {
    s->global_step = 0; s->opt_step = 1; s->rampup_length = rampup_length; s->base_seed = base_seed;
    s->lr = lr; s->beta1 = beta1; s->beta2 = beta2; s->eps = eps; s->ema_decay = ema_decay; s->max_cons_cost = max_cons_cost;
    step_state_derive(s);
}
__global__ void k_step_state_advance(sed_step_state* s) {
    s->global_step += 1;
    s->opt_step += 1;
    step_state_derive(s);
}

// flags bit 0: replace base_seed (and re-derive this step's dropout keys); bit 1: replace lr
__global__ void k_step_state_update(sed_step_state* s, uint64_t base_seed, double lr, int flags) {
    if (flags & 1) s->base_seed = base_seed;
    if (flags & 2) s->lr = lr;
    step_state_derive(s);
}

// global_step = epoch * len(train_loader) + i is RECOMPUTED from the epoch argument at every call of main.train (main.py:74):
// a run that enters train(epoch = k) with a fresh step object must see the ramp-up, EMA alpha and dropout keys of step
// k * len(loader), not of step 0.  The Adam step counter is the optimiser's own and is left alone.
__global__ void k_step_state_set_global_step(sed_step_state* s, int64_t gs) {
    s->global_step = gs;
    step_state_derive(s);
}

extern "C" int sed_adam_ema(int64_t n, float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                            float* ema_params, const sed_step_state* state_dev, float grad_scale, void* stream) {
    SED_CHECK_ARG(n > 0 && params && grads && exp_avg && exp_avg_sq && state_dev, "sed_adam_ema: bad argument");
    SED_CHECK_ARG(((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq | (uintptr_t)ema_params) % 16 == 0,
                  "sed_adam_ema: buffers must be 16-byte aligned");
    int64_t blocks = (n / 4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    if (ema_params) k_adam_ema<true><<<(int)blocks, 256, 0, (hipStream_t)stream>>>(n, params, grads, exp_avg, exp_avg_sq, ema_params, state_dev, grad_scale);
    else k_adam_ema<false><<<(int)blocks, 256, 0, (hipStream_t)stream>>>(n, params, grads, exp_avg, exp_avg_sq, nullptr, state_dev, grad_scale);
    SED_CHECK_LAUNCH();
    return SED_OK;
}

extern "C" int sed_ema_update(int64_t n, const float* params, float* ema_params, float alpha, void* stream) {
    SED_CHECK_ARG(n > 0 && params && ema_params, "sed_ema_update: bad argument");
    int64_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    k_ema<<<(int)blocks, 256, 0, (hipStream_t)stream>>>(n, params, ema_params, alpha);
    SED_CHECK_LAUNCH();
    return SED_OK;
}

extern "C" int sed_step_state_init(sed_step_state* state_dev, uint64_t base_seed, int64_t rampup_length, double lr,
                                   double beta1, double beta2, double eps, double ema_decay, double max_cons_cost,
                                   void* stream) {
    SED_CHECK_ARG(state_dev, "sed_step_state_init: null state");
    k_step_state_init<<<1, 1, 0, (hipStream_t)stream>>>(state_dev, base_seed, rampup_length, lr, beta1, beta2, eps, ema_decay,
                                                        max_cons_cost);
    SED_CHECK_LAUNCH();
    return SED_OK;
}

extern "C" int sed_step_state_update(sed_step_state* state_dev, uint64_t base_seed, double lr, int flags, void* stream) {
    SED_CHECK_ARG(state_dev, "sed_step_state_update: null state");
    k_step_state_update<<<1, 1, 0, (hipStream_t)stream>>>(state_dev, base_seed, lr, flags);
    SED_CHECK_LAUNCH();
    return SED_OK;
}

extern "C" int sed_step_state_set_global_step(sed_step_state* state_dev, int64_t global_step, void* stream) {
    SED_CHECK_ARG(state_dev && global_step >= 0, "sed_step_state_set_global_step: bad argument");
    k_step_state_set_global_step<<<1, 1, 0, (hipStream_t)stream>>>(state_dev, global_step);
    SED_CHECK_LAUNCH();
    return SED_OK;
}

extern "C" int sed_step_state_advance(sed_step_state* state_dev, void* stream) {
    SED_CHECK_ARG(state_dev, "sed_step_state_advance: null state");
    k_step_state_advance<<<1, 1, 0, (hipStream_t)stream>>>(state_dev);
    SED_CHECK_LAUNCH();
    return SED_OK;
}
