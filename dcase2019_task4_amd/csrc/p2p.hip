// p2p.hip - the data-parallel step's gradient all-reduce as ONE kernel over peer-mapped device memory (round 5).
//
// The reference is single-process; what it fixes is the batch contract the data-parallel step preserves
// (baseline/main.py:238-247, DataLoad.py:562-571: every rank takes its share of each stream) and that the update uses the mean
// gradient of the global batch (main.py:152-154).  The message is small - 857 KB for cfg.crnn_kwargs, 8.5 MB for the wide model,
// in two buckets - so a ring all-reduce is latency: 2 (W - 1) dependent hops of a library kernel the step cannot see into, and on
// a one-GPU box RCCL refuses two ranks altogether (the captured-collective schedule had never run with world > 1).  Here every
// rank owns a communication buffer that all W ranks of the node map (hipIpcMemHandle; xGMI is point-to-point, every peer is one
// hop away), and one launch does reduce-scatter + all-gather with direct remote loads / stores:
//
//   workgroup g of rank r (G workgroups, the same G on every rank; element chunks of 4 floats dealt round-robin to the
//   (slice, workgroup) pairs so that every remote access is a 16-byte vector):
//     1. publish : copies ITS share of the local gradient bucket into stage[e & 1] of r's own buffer, release fence,
//                  then writes ready[r][g] = e into every rank's flag page
//     2. reduce  : waits for ready[p][g] == e of every rank p, sums ITS share of slice r over the ranks IN RANK ORDER
//                  (p = 0 .. W-1: every rank ends with the bit-identical sum - replicas must not drift apart) reading the
//                  peers' stage buffers over xGMI, and writes the sums into result[e & 1] of EVERY rank; release fence,
//                  done[r][g] = e everywhere
//     3. gather  : waits for done[p][g] == e of every p, copies its share of result[e & 1] back over the gradient bucket.
//   Only same-index workgroups of different ranks ever wait for each other, and every wait is on a flag whose writer has no
//   wait in front of it that depends on the waiter (publish precedes every wait): no cycle, whatever the residency.  A rank's
//   stage / result halves alternate with the epoch e (a per-workgroup device word, so a replayed hipGraph needs no host
//   value): reuse of a half is two epochs away, and an epoch cannot start before every peer has finished the previous one.
//   Every spin is bounded (wall clock) and reports through a sticky error word instead of hanging the GPU.
//
// Memory: the communication buffers are allocated HERE (sed_p2p_alloc: fine-grained device memory, the kind peers may read
// and write coherently while kernels run) - the one place the library owns device memory, because it has to be created with
// flags torch's allocator does not offer and exported through hipIpcGetMemHandle.  The gradient buffer itself stays torch's.
#include <string.h>
#include "common.h"

#define P2P_THREADS 256
#define P2P_MAX_WORLD 16
#define P2P_MAX_WG 64
#define P2P_FLAG_WORDS (2 * P2P_MAX_WORLD * P2P_MAX_WG)       // ready | done, [rank][workgroup]
#define P2P_HDR_BYTES 16384                                   // flags (8 KB) | epoch[G] | error word | padding
#define P2P_TIMEOUT_TICKS (300ull * 1000 * 1000)              // 3 s of the 100 MHz wall clock

struct P2PHeader {
    unsigned int flags[P2P_FLAG_WORDS];
    unsigned int epoch[P2P_MAX_WG];
    unsigned int error;            // sticky: number of waits that timed out
    unsigned int magic;
};
static_assert(sizeof(P2PHeader) <= P2P_HDR_BYTES, "header page");

struct P2PPeers { char* buf[P2P_MAX_WORLD]; };

__device__ __forceinline__ unsigned int p2p_flag_load(const unsigned int* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void p2p_flag_store(unsigned int* p, unsigned int v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// stage / result halves: [2][n_cap] floats each, behind the header
__device__ __forceinline__ float* p2p_stage(char* buf, size_t n_cap, unsigned int e) { return (float*)(buf + P2P_HDR_BYTES) + (size_t)(e & 1u) * n_cap; }
__device__ __forceinline__ float* p2p_result(char* buf, size_t n_cap, unsigned int e) { return (float*)(buf + P2P_HDR_BYTES) + (size_t)(2u + (e & 1u)) * n_cap; }

// data: the local gradient bucket (n floats, 16-byte aligned); n4 = ceil(n / 4) vector chunks; chunk c belongs to slice
// c % W and, inside the slice, to workgroup (c / W) % G.
__global__ __launch_bounds__(P2P_THREADS) void k_p2p_allreduce(float* __restrict__ data, long long n, int rank, int W, P2PPeers peers,
                                                                size_t n_cap) {
    const int g = blockIdx.x, G = gridDim.x, tid = threadIdx.x;
    // (the peer table goes to LDS: sixteen 64-bit kernel arguments indexed by a run-time rank cost 85 spilled scalar registers)
    __shared__ char* s_buf[P2P_MAX_WORLD];
    __shared__ unsigned int s_e;
    __shared__ int s_bad;
    if (tid < P2P_MAX_WORLD) s_buf[tid] = peers.buf[tid < W ? tid : 0];
    __syncthreads();
    char* mine = s_buf[rank];
    P2PHeader* hdr = (P2PHeader*)mine;
    if (tid == 0) { s_e = hdr->epoch[g] + 1u; s_bad = 0; }
    __syncthreads();
    const unsigned int e = s_e;
    const long long n4 = (n + 3) / 4;
    const long long per_wg = (n4 + (long long)W * G - 1) / ((long long)W * G);     // chunks of one (slice, workgroup) pair
    auto chunk_of = [&](int slice, long long k) -> long long { return ((k * G + g) * W + slice); };
    auto ld4 = [&](const float* p, long long c) -> f32x4 {
        if (4 * c + 3 < n) return *(const f32x4*)(p + 4 * c);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < 4; ++i) if (4 * c + i < n) v[i] = p[4 * c + i];
        return v;
    };
    auto st4 = [&](float* p, long long c, f32x4 v) {
        if (4 * c + 3 < n) *(f32x4*)(p + 4 * c) = v;
        else for (int i = 0; i < 4; ++i) if (4 * c + i < n) p[4 * c + i] = v[i];
    };
    // bounded wait for flag words [kind][p][g] == e of every rank p (thread p polls rank p's word in MY flag page)
    auto wait_all = [&](int kind) {
        if (tid < W) {
            const unsigned int* f = &hdr->flags[(kind * P2P_MAX_WORLD + tid) * P2P_MAX_WG + g];
            const unsigned long long t0 = wall_clock64();
            while (p2p_flag_load(f) != e) {
                __builtin_amdgcn_s_sleep(2);
                if (wall_clock64() - t0 > P2P_TIMEOUT_TICKS) { atomicAdd(&hdr->error, 1u); s_bad = 1; break; }
            }
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");               // system scope: what the flags' writers released
    };
    auto signal_all = [&](int kind) {
        __syncthreads();                                             // every thread's stores issued and counted (vmcnt drained)
        if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");  // system scope
        __syncthreads();
        if (tid < W) p2p_flag_store(&((P2PHeader*)s_buf[tid])->flags[(kind * P2P_MAX_WORLD + rank) * P2P_MAX_WG + g], e);
    };
    // ---- 1. publish my share of every slice ---------------------------------------------------------------------------------
    {
        float* st = p2p_stage(mine, n_cap, e);
        for (int s = 0; s < W; ++s)
            for (long long k = tid; k < per_wg; k += P2P_THREADS) {
                const long long c = chunk_of(s, k);
                if (c < n4) st4(st, c, ld4(data, c));
            }
    }
    signal_all(0);
    // ---- 2. reduce my share of slice `rank` over the ranks, in rank order; write it to every rank --------------------------
    wait_all(0);
    for (long long k = tid; k < per_wg; k += P2P_THREADS) {
        const long long c = chunk_of(rank, k);
        if (c >= n4) continue;
        // eight remote loads in flight per batch (clamped rank: the surplus loads re-read the last rank and are not added); the
        // sum runs p = 0, 1, .. W-1 on every rank
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int p0 = 0; p0 < W; p0 += 8) {
            f32x4 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = ld4(p2p_stage(s_buf[min(p0 + i, W - 1)], n_cap, e), c);
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (p0 + i < W) acc += v[i];
        }
        for (int p = 0; p < W; ++p) st4(p2p_result(s_buf[p], n_cap, e), c, acc);
    }
    signal_all(1);
    // ---- 3. gather: every slice's reduced share of this workgroup back into the gradient bucket ------------------------------
    wait_all(1);
    {
        const float* rs = p2p_result(mine, n_cap, e);
        for (int s = 0; s < W; ++s)
            for (long long k = tid; k < per_wg; k += P2P_THREADS) {
                const long long c = chunk_of(s, k);
                if (c < n4) st4(data, c, ld4(rs, c));
            }
    }
    if (tid == 0) hdr->epoch[g] = e;
    (void)s_bad;
}

// ---- host side ---------------------------------------------------------------------------------------------------------------
extern "C" size_t sed_p2p_buffer_bytes(long long n_floats_max) {
    if (n_floats_max <= 0) return 0;
    const size_t n_cap = ((size_t)n_floats_max + 63) & ~(size_t)63;
    return P2P_HDR_BYTES + 4 * n_cap * sizeof(float);
}

// Allocates + zeroes a communication buffer on the current device and exports it.  handle_out: 64 bytes (hipIpcMemHandle_t).
// fine_grained != 0: hipExtMallocWithFlags(hipDeviceMallocFinegrained) - coherent for peers while kernels run; 0: plain hipMalloc
// (what sed_p2p_alloc falls back to if the fine-grained allocation or its export fails; *fine_grained_out says which it was).
extern "C" int sed_p2p_alloc(size_t bytes, int fine_grained, void** ptr_out, void* handle_out, int* fine_grained_out) {
    SED_CHECK_ARG(ptr_out && handle_out && bytes >= P2P_HDR_BYTES, "sed_p2p_alloc: bad argument");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "handle size");
    void* p = nullptr;
    hipIpcMemHandle_t h;
    int fg = 0;
    if (fine_grained) {
        if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained) == hipSuccess) {
            if (hipIpcGetMemHandle(&h, p) == hipSuccess) fg = 1;
            else { (void)hipFree(p); p = nullptr; }
        }
        (void)hipGetLastError();
    }
    if (p == nullptr) {
        SED_CHECK_HIP(hipMalloc(&p, bytes));
        hipError_t e = hipIpcGetMemHandle(&h, p);
        if (e != hipSuccess) {
            (void)hipFree(p);
            sed_set_error("sed_p2p_alloc: hipIpcGetMemHandle -> %s", hipGetErrorString(e));
            return SED_ERR_LAUNCH;
        }
    }
    SED_CHECK_HIP(hipMemset(p, 0, bytes));
    SED_CHECK_HIP(hipDeviceSynchronize());
    memcpy(handle_out, &h, sizeof h);
    *ptr_out = p;
    if (fine_grained_out) *fine_grained_out = fg;
    return SED_OK;
}
extern "C" int sed_p2p_open(const void* handle, void** ptr_out) {
    SED_CHECK_ARG(handle && ptr_out, "sed_p2p_open: null argument");
    hipIpcMemHandle_t h;
    memcpy(&h, handle, sizeof h);
    SED_CHECK_HIP(hipIpcOpenMemHandle(ptr_out, h, hipIpcMemLazyEnablePeerAccess));
    return SED_OK;
}
extern "C" int sed_p2p_close(void* peer_ptr) {
    if (peer_ptr) SED_CHECK_HIP(hipIpcCloseMemHandle(peer_ptr));
    return SED_OK;
}
extern "C" int sed_p2p_free(void* own_ptr) {
    if (own_ptr) SED_CHECK_HIP(hipFree(own_ptr));
    return SED_OK;
}
// 1 if the current device can map device `peer_device`'s memory (or it IS that device)
extern "C" int sed_p2p_can_access(int peer_device) {
    int dev = 0, can = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (dev == peer_device) return 1;
    if (hipDeviceCanAccessPeer(&can, dev, peer_device) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return can;
}
// the sticky timeout counter of a communication buffer (blocking 4-byte copy; test / health-check use)
extern "C" int sed_p2p_errors(const void* own_ptr, unsigned int* out) {
    SED_CHECK_ARG(own_ptr && out, "sed_p2p_errors: null argument");
    SED_CHECK_HIP(hipMemcpy(out, (const char*)own_ptr + offsetof(P2PHeader, error), sizeof(unsigned int), hipMemcpyDeviceToHost));
    return SED_OK;
}

// In-place sum all-reduce of data[0, n) over the `world` ranks whose communication buffers are bufs[0 .. world) (bufs[rank] =
// this rank's own; every one created with the SAME n_floats_max); every rank must enqueue the same sequence of calls.
// One launch of `workgroups` workgroups (0: default 16; the same value on every rank), capturable, no host synchronisation.
extern "C" int sed_p2p_allreduce(float* data, long long n, int rank, int world, void* const* bufs, long long n_floats_max,
                                 int workgroups, void* stream) {
    SED_CHECK_ARG(data && bufs && n >= 0 && world >= 1 && world <= P2P_MAX_WORLD && rank >= 0 && rank < world, "sed_p2p_allreduce: bad argument");
    SED_CHECK_ARG(n <= n_floats_max, "sed_p2p_allreduce: message larger than the communication buffers");
    SED_CHECK_ARG(((uintptr_t)data & 15) == 0, "sed_p2p_allreduce: data must be 16-byte aligned");
    if (n == 0) return SED_OK;
    int G = workgroups > 0 ? workgroups : 16;
    if (G > P2P_MAX_WG) G = P2P_MAX_WG;
    P2PPeers peers = {};
    for (int p = 0; p < world; ++p) {
        SED_CHECK_ARG(bufs[p] != nullptr, "sed_p2p_allreduce: null peer buffer");
        peers.buf[p] = (char*)bufs[p];
    }
    const size_t n_cap = ((size_t)n_floats_max + 63) & ~(size_t)63;
    k_p2p_allreduce<<<G, P2P_THREADS, 0, (hipStream_t)stream>>>(data, n, rank, world, peers, n_cap);
    SED_CHECK_LAUNCH();
    return SED_OK;
}
