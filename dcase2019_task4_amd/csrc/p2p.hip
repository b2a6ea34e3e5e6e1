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
//     1. publish : copies ITS share of the local gradient bucket into stage[e & 1] of r's own buffer (system-scope stores),
//                  drains them, then writes ready[r][g] = e into every rank's flag page
//     2. reduce  : waits for ready[p][g] == e of every rank p, sums ITS share of slice r over the ranks IN RANK ORDER
//                  (p = 0 .. W-1: every rank ends with the bit-identical sum - replicas must not drift apart) reading the
//                  peers' stage buffers over xGMI, and writes the sums into result[e & 1] of EVERY rank; drain,
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
#define P2P_MAX_WG 256
#define P2P_FLAG_WORDS (2 * P2P_MAX_WORLD * P2P_MAX_WG)       // ready | done, [rank][workgroup]
#define P2P_HDR_BYTES 49152                                   // flags (32 KB) | epoch[G] | error word | padding
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

// Payload accesses to the communication buffers: relaxed SYSTEM-scope 8-byte atomics (global_load / store_dwordx2 sc0 sc1), two per
// 16-byte chunk.  They bypass / write through the caches on both sides, so the hand-over needs NO cache-maintenance fence: the
// producer drains its stores (s_waitcnt vmcnt(0) of the workgroup barrier) and stores the flag, the consumer sees the flag and
// loads.  The first version used plain accesses between system-scope release / acquire fences: a release fence writes back every
// dirty line of the XCD's L2 - mostly the conv backward's, which runs beside the tail bucket's all-reduce - and the one-rank
// launch-structure measurement came out at +51 us per step for the two calls.
typedef unsigned long long p2p_u64;
typedef __attribute__((ext_vector_type(2))) p2p_u64 p2p_u64x2;
__device__ __forceinline__ f32x4 ld_sys(const float* p) {
    p2p_u64x2 v;
    v[0] = __hip_atomic_load((const p2p_u64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    v[1] = __hip_atomic_load((const p2p_u64*)p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return __builtin_bit_cast(f32x4, v);
}
__device__ __forceinline__ void st_sys(float* p, f32x4 x) {
    const p2p_u64x2 v = __builtin_bit_cast(p2p_u64x2, x);
    __hip_atomic_store((p2p_u64*)p, v[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store((p2p_u64*)p + 1, v[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ float ld_sys1(const float* p) { return __builtin_bit_cast(float, __hip_atomic_load((const unsigned int*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)); }
__device__ __forceinline__ void st_sys1(float* p, float x) { __hip_atomic_store((unsigned int*)p, __builtin_bit_cast(unsigned int, x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// stage / result halves: [2][n_cap] floats each, behind the header
__device__ __forceinline__ float* p2p_stage(char* buf, size_t n_cap, unsigned int e) { return (float*)(buf + P2P_HDR_BYTES) + (size_t)(e & 1u) * n_cap; }
__device__ __forceinline__ float* p2p_result(char* buf, size_t n_cap, unsigned int e) { return (float*)(buf + P2P_HDR_BYTES) + (size_t)(2u + (e & 1u)) * n_cap; }

// data: the local gradient bucket (n floats, 16-byte aligned); n4 = n / 4 whole vector chunks (+ at most one partial tail chunk);
// chunk c belongs to slice c % W and, inside the slice, to workgroup (c / W) % G.
// Every loop keeps EIGHT 16-byte accesses per lane in flight (loads first, then the stores): the communication buffers are
// fine-grained (uncached) memory, so a load-then-store loop pays a full memory round trip per iteration - the first version,
// one chunk per iteration, took ~30 us per call for 500 KB on ONE rank.  WC: compile-time world size (1, 2, 4, 8; 0 = any).
template <int WC>
__global__ __launch_bounds__(P2P_THREADS) void k_p2p_allreduce(float* __restrict__ data, long long n, int rank, int W_, P2PPeers peers,
                                                                size_t n_cap) {
    const int W = WC ? WC : W_;
    const int g = blockIdx.x, G = gridDim.x, tid = threadIdx.x;
    // (the peer table goes to LDS: sixteen 64-bit kernel arguments indexed by a run-time rank cost 85 spilled scalar registers)
    __shared__ char* s_buf[P2P_MAX_WORLD];
    __shared__ unsigned int s_e;
    if (tid < P2P_MAX_WORLD) s_buf[tid] = peers.buf[tid < W ? tid : 0];
    __syncthreads();
    char* mine = s_buf[rank];
    P2PHeader* hdr = (P2PHeader*)mine;
    if (tid == 0) s_e = hdr->epoch[g] + 1u;
    __syncthreads();
    const unsigned int e = s_e;
    const long long n4 = n / 4;                                                     // whole chunks
    const long long nck = (n + 3) / 4;                                              // incl. the partial tail chunk (index n4) if n % 4
    const long long per_wg = (nck + (long long)W * G - 1) / ((long long)W * G);     // chunks of one (slice, workgroup) pair
    auto chunk_of = [&](int slice, long long k) -> long long { return ((k * G + g) * W + slice); };
    // the partial tail chunk (n % 4 floats) is moved element-wise by whoever owns it
    // (to_comm: local gradient -> own staging half, system-scope stores; else own result half -> local gradient, system-scope loads)
    auto tail_copy = [&](const float* src, float* dst, bool to_comm) {
        for (long long i = 4 * n4; i < n; ++i) { if (to_comm) st_sys1(dst + i, src[i]); else dst[i] = ld_sys1(src + i); }
    };
    // bounded wait for flag words [kind][p][g] == e of every rank p (thread p polls rank p's word in MY flag page)
    auto wait_all = [&](int kind) {
        if (tid < W) {
            const unsigned int* f = &hdr->flags[(kind * P2P_MAX_WORLD + tid) * P2P_MAX_WG + g];
            const unsigned long long t0 = wall_clock64();
            while (p2p_flag_load(f) != e) {
                __builtin_amdgcn_s_sleep(2);
                if (wall_clock64() - t0 > P2P_TIMEOUT_TICKS) { atomicAdd(&hdr->error, 1u); break; }
            }
        }
        __syncthreads();                                             // (the payload loads that follow are system-scope themselves)
    };
    auto signal_all = [&](int kind) {
        __syncthreads();                                             // every wave's system-scope stores issued AND acknowledged (vmcnt drained)
        if (tid < W) p2p_flag_store(&((P2PHeader*)s_buf[tid])->flags[(kind * P2P_MAX_WORLD + rank) * P2P_MAX_WG + g], e);
    };
    // this workgroup's chunks of ALL slices, flattened: j = k * W + slice; eight per lane and trip
    auto copy_all = [&](const float* src, float* dst, bool to_comm) {
        const long long nj = per_wg * W;
        for (long long j0 = tid; j0 < nj; j0 += 8 * P2P_THREADS) {
            f32x4 v[8];
            long long c[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const long long j = j0 + (long long)i * P2P_THREADS;
                c[i] = chunk_of((int)(j % W), j / W);
                const bool ok = j < nj && c[i] < n4;
                const float* sp = src + 4 * (ok ? c[i] : 0);
                v[i] = to_comm ? *(const f32x4*)sp : ld_sys(sp);
                if (!ok) c[i] = -1;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (c[i] >= 0) { if (to_comm) st_sys(dst + 4 * c[i], v[i]); else *(f32x4*)(dst + 4 * c[i]) = v[i]; }
        }
        if (n4 != nck && tid == 0 && (int)((n4 / W) % G) == g) tail_copy(src, dst, to_comm);
    };
    // ---- 1. publish my share of every slice ---------------------------------------------------------------------------------
    copy_all(data, p2p_stage(mine, n_cap, e), true);
    signal_all(0);
    // ---- 2. reduce my share of slice `rank` over the ranks, in rank order (p = 0 .. W - 1 on every rank: bit-identical sums);
    //         write it to every rank.  KB chunks x W ranks = eight remote loads per lane in flight -------------------------------
    wait_all(0);
    if constexpr (WC != 0) {
        constexpr int KB = 8 / WC;
        for (long long k0 = tid; k0 < per_wg; k0 += (long long)KB * P2P_THREADS) {
            f32x4 v[KB][WC];
            long long c[KB];
#pragma unroll
            for (int kk = 0; kk < KB; ++kk) {
                const long long k = k0 + (long long)kk * P2P_THREADS;
                c[kk] = chunk_of(rank, k);
                const bool ok = k < per_wg && c[kk] < n4;
#pragma unroll
                for (int p = 0; p < WC; ++p) v[kk][p] = ld_sys(p2p_stage(s_buf[p], n_cap, e) + 4 * (ok ? c[kk] : 0));
                if (!ok) c[kk] = -1;
            }
#pragma unroll
            for (int kk = 0; kk < KB; ++kk) {
                f32x4 acc = v[kk][0];
#pragma unroll
                for (int p = 1; p < WC; ++p) acc += v[kk][p];
                if (c[kk] >= 0) {
#pragma unroll
                    for (int p = 0; p < WC; ++p) st_sys(p2p_result(s_buf[p], n_cap, e) + 4 * c[kk], acc);
                }
            }
        }
    } else {
        for (long long k = tid; k < per_wg; k += P2P_THREADS) {
            const long long c = chunk_of(rank, k);
            if (c >= n4) continue;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int p0 = 0; p0 < W; p0 += 8) {
                f32x4 v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = ld_sys(p2p_stage(s_buf[min(p0 + i, W - 1)], n_cap, e) + 4 * c);
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (p0 + i < W) acc += v[i];
            }
            for (int p = 0; p < W; ++p) st_sys(p2p_result(s_buf[p], n_cap, e) + 4 * c, acc);
        }
    }
    if (n4 != nck && tid == 0 && (int)(n4 % W) == rank && (int)((n4 / W) % G) == g) {      // the partial tail chunk of my slice
        for (long long i = 4 * n4; i < n; ++i) {
            float acc = 0.f;
            for (int p = 0; p < W; ++p) acc += ld_sys1(p2p_stage(s_buf[p], n_cap, e) + i);
            for (int p = 0; p < W; ++p) st_sys1(p2p_result(s_buf[p], n_cap, e) + i, acc);
        }
    }
    signal_all(1);
    // ---- 3. gather: every slice's reduced share of this workgroup back into the gradient bucket ------------------------------
    wait_all(1);
    copy_all(p2p_result(mine, n_cap, e), data, false);
    if (tid == 0) hdr->epoch[g] = e;
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
// One launch of `workgroups` workgroups (0: default, derived from n; the same value on every rank), capturable, no host synchronisation.
extern "C" int sed_p2p_allreduce(float* data, long long n, int rank, int world, void* const* bufs, long long n_floats_max,
                                 int workgroups, void* stream) {
    SED_CHECK_ARG(data && bufs && n >= 0 && world >= 1 && world <= P2P_MAX_WORLD && rank >= 0 && rank < world, "sed_p2p_allreduce: bad argument");
    SED_CHECK_ARG(n <= n_floats_max, "sed_p2p_allreduce: message larger than the communication buffers");
    SED_CHECK_ARG(((uintptr_t)data & 15) == 0, "sed_p2p_allreduce: data must be 16-byte aligned");
    if (n == 0) return SED_OK;
    // default: one workgroup per 2 K floats of the buffers' capacity, at least 32, at most P2P_MAX_WG - the same on every rank and call.  One-rank
    // launch-structure measurements (tools/dp1_wgs.sh; the buffers are uncached fine-grained memory, so every phase is round
    // trips, and more lanes share them): the base model's 857 KB message 36 us on 32 workgroups, 25 on 64, 19 on 128; the wide
    // model's 8.5 MB 310 us on 32, 74 on 256.
    // (derived from the buffers' CAPACITY, not from this call's n: the stage / result halves alternate with a per-workgroup epoch,
    // which is only safe when every call on a set of buffers uses the same workgroups for the same chunks)
    int G = workgroups > 0 ? workgroups : (int)((n_floats_max + 2047) / 2048);
    if (G < 32 && workgroups <= 0) G = 32;
    // (default capped at 128: same-index workgroups of different ranks spin on each other; two ranks SHARING one GPU - the one-GPU
    // test set-up - must pass 32 explicitly, as dist.PeerAllReduce does: 2 x 105 .. 256 spinning workgroups beside both ranks'
    // persistent kernels did not all become resident before the bounded waits ran out)
    if (G > 128 && workgroups <= 0) G = 128;
    if (G > P2P_MAX_WG) G = P2P_MAX_WG;
    P2PPeers peers = {};
    for (int p = 0; p < world; ++p) {
        SED_CHECK_ARG(bufs[p] != nullptr, "sed_p2p_allreduce: null peer buffer");
        peers.buf[p] = (char*)bufs[p];
    }
    const size_t n_cap = ((size_t)n_floats_max + 63) & ~(size_t)63;
    hipStream_t st = (hipStream_t)stream;
    switch (world) {
        case 1: k_p2p_allreduce<1><<<G, P2P_THREADS, 0, st>>>(data, n, rank, world, peers, n_cap); break;
        case 2: k_p2p_allreduce<2><<<G, P2P_THREADS, 0, st>>>(data, n, rank, world, peers, n_cap); break;
        case 4: k_p2p_allreduce<4><<<G, P2P_THREADS, 0, st>>>(data, n, rank, world, peers, n_cap); break;
        case 8: k_p2p_allreduce<8><<<G, P2P_THREADS, 0, st>>>(data, n, rank, world, peers, n_cap); break;
        default: k_p2p_allreduce<0><<<G, P2P_THREADS, 0, st>>>(data, n, rank, world, peers, n_cap); break;
    }
    SED_CHECK_LAUNCH();
    return SED_OK;
}
