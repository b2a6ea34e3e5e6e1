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
#include <stdlib.h>
#include <string.h>
#include "common.h"

#define P2P_THREADS 256
#define P2P_MAX_WORLD 16
#define P2P_MAX_WG 256
#define P2P_FLAG_WORDS (2 * P2P_MAX_WORLD * P2P_MAX_WG)       // ready | done, [rank][workgroup]
#define P2P_HDR_BYTES 49152                                   // flags (32 KB) | epoch[G] | error word | configuration | padding
#define P2P_TICKS_PER_S (100ull * 1000 * 1000)                // the 100 MHz wall clock
#define P2P_DEFAULT_TIMEOUT_S 600.0                           // SED_P2P_TIMEOUT_S / sed_p2p_configure override it

struct P2PHeader {
    unsigned int flags[P2P_FLAG_WORDS];
    unsigned int epoch[P2P_MAX_WG];
    unsigned int error;                 // sticky: number of waits that timed out
    unsigned int magic;
    unsigned long long timeout_ticks;   // wait budget of ONE cross-rank wait (sed_p2p_alloc: SED_P2P_TIMEOUT_S or 600 s)
    unsigned int* host_err;             // optional word in pinned host memory that a timed-out wait also raises (sed_p2p_configure):
                                        // the host polls it without synchronising the device
};
static_assert(sizeof(P2PHeader) <= P2P_HDR_BYTES, "header page");

struct P2PPeers { char* buf[P2P_MAX_WORLD]; };

__device__ __forceinline__ unsigned int p2p_flag_load(const unsigned int* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// (assembly with a marker comment, like the drain in signal_all: tests/test_abi.py checks in the compiler's output that nothing
// but the barrier sits between the two)
__device__ __forceinline__ void p2p_flag_store(unsigned int* p, unsigned int v) {
    asm volatile("global_store_dword %0, %1, off sc0 sc1 ; p2p_flag_store" :: "v"(p), "v"(v) : "memory");
}

// Payload accesses to the communication buffers are 16-byte SYSTEM-scope accesses (global_load / store_dwordx4 sc0 sc1): they
// bypass / write through the caches on both sides, so the hand-over needs no cache-maintenance fence (a system-scope release
// fence writes back every dirty line of the XCD's L2 - mostly the conv backward's, which runs beside the tail bucket's
// all-reduce: +51 us per step in round 5's first version) - but it DOES need the producer to wait until its stores are
// acknowledged before it raises the flag.  That wait is an explicit `s_waitcnt vmcnt(0)` on every wave in signal_all(): the
// workgroup barrier does NOT contain one on gfx950 (back-off barrier: the compiler emits s_waitcnt lgkmcnt(0); s_barrier only -
// round 5 relied on it and a flag could overtake payload stores still in flight across xGMI; tests/test_abi.py greps the
// disassembly for the drain).  This is the AMDGPU memory model's own system-scope release sequence minus the L2 write-back
// that only non-write-through stores need.
//
// The accesses are inline assembly because the compiler cannot express them pipelined: relaxed system-scope ATOMICS are at most
// 8 bytes (two half-line stores per chunk), and it put `s_waitcnt vmcnt(0)` - which on gfx9 also waits for every earlier STORE's
// acknowledgement - in front of each store pair, so every lane had one uncached round trip in flight at a time (round 5:
// 8.5 MB in 100 us on one rank).  Here a lane issues eight loads, waits once, then issues its stores back to back.
// ONE statement = eight loads + the wait for them: nothing the compiler schedules or copies can come between a load and the
// wait that makes its destination registers valid (the first version had the wait as a separate statement tied to the registers;
// this form leaves no room for a register copy in between).  P2P_ST_*: one 16-byte store, no wait.
#define P2P_LD8(BITS, v, a)                                                                                              \
    asm volatile("global_load_dwordx4 %0, %8, off" BITS "\n\tglobal_load_dwordx4 %1, %9, off" BITS                       \
                 "\n\tglobal_load_dwordx4 %2, %10, off" BITS "\n\tglobal_load_dwordx4 %3, %11, off" BITS                 \
                 "\n\tglobal_load_dwordx4 %4, %12, off" BITS "\n\tglobal_load_dwordx4 %5, %13, off" BITS                 \
                 "\n\tglobal_load_dwordx4 %6, %14, off" BITS "\n\tglobal_load_dwordx4 %7, %15, off" BITS                 \
                 "\n\ts_waitcnt vmcnt(0)"                                                                                \
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])  \
                 : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7])                  \
                 : "memory")
#define P2P_LD8_SYS(v, a) P2P_LD8(" sc0 sc1", v, a)
#define P2P_LD8_PLAIN(v, a) P2P_LD8("", v, a)
// The s_nop behind every store is a HARDWARE hazard the compiler only resolves for its own instructions: a VMEM store of more
// than 8 bytes reads its data registers a cycle or two after issue, and a VALU write to one of them in that window lands in
// the stored data (the first 16-byte version did exactly that: ~1 call in 5 stored a zero, a NaN constant or the next
// address temporary in ONE dword of a few chunks - found by tools/p2p_debug.py at world 2, never at world 1).
#define P2P_ST_SYS(p, v)   asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 2" :: "v"(p), "v"(v) : "memory")
#define P2P_ST_PLAIN(p, v) asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 2" :: "v"(p), "v"(v) : "memory")
__device__ __forceinline__ float ld_sys1(const float* p) { return __builtin_bit_cast(float, __hip_atomic_load((const unsigned int*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)); }
__device__ __forceinline__ void st_sys1(float* p, float x) { __hip_atomic_store((unsigned int*)p, __builtin_bit_cast(unsigned int, x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// stage / result halves: [2][n_cap] floats each, behind the header
__device__ __forceinline__ float* p2p_stage(char* buf, size_t n_cap, unsigned int e) { return (float*)(buf + P2P_HDR_BYTES) + (size_t)(e & 1u) * n_cap; }
__device__ __forceinline__ float* p2p_result(char* buf, size_t n_cap, unsigned int e) { return (float*)(buf + P2P_HDR_BYTES) + (size_t)(2u + (e & 1u)) * n_cap; }

// data: the local gradient bucket (n floats, 16-byte aligned); n4 = n / 4 whole vector chunks (+ at most one partial tail chunk);
// chunk c belongs to slice c % W and, inside the slice, to workgroup (c / W) % G.
// A rank's OWN slice never goes through the communication buffers: nobody else reads slice r of rank r's staging half, so the
// reduce phase takes that operand from the gradient bucket itself (still summed in rank order), writes rank r's copy of the
// result straight back into the bucket, and publish / gather skip the slice - 1 / W of the local traffic, and a one-rank group
// moves nothing through uncached memory at all.  WC: compile-time world size (1, 2, 4, 8; 0 = any).
template <int WC>
__global__ __launch_bounds__(P2P_THREADS) void k_p2p_allreduce(float* __restrict__ data, long long n, int rank, int W_, P2PPeers peers,
                                                                size_t n_cap) {
    const int W = WC ? WC : W_;
    const int g = blockIdx.x, G = gridDim.x, tid = threadIdx.x;
    // (lane predicates on the thread index are RE-formed where they are used: kept live across the phases they cost the any-W
    // instantiation spilled scalar register pairs)
    auto tid_now = [&]() { int t = tid; asm volatile("" : "+v"(t)); return t; };
    auto has_tail = [&]() { int r = (int)(n & 3); asm volatile("" : "+s"(r)); return r != 0; };      // (likewise: n % 4 != 0)
    // (the peer table goes to LDS: sixteen 64-bit kernel arguments indexed by a run-time rank cost 85 spilled scalar registers)
    __shared__ char* s_buf[P2P_MAX_WORLD];
    __shared__ unsigned int s_e, s_bad;
    __shared__ unsigned long long s_budget;
#pragma unroll
    for (int i = 0; i < P2P_MAX_WORLD; ++i)      // (compile-time indices: a run-time index loads all sixteen pointers into SGPRs at once)
        if (tid == i) s_buf[i] = (i < W) ? peers.buf[i] : peers.buf[0];
    __syncthreads();
    char* mine = s_buf[rank];
    P2PHeader* hdr = (P2PHeader*)mine;
    if (tid == 0) { s_e = hdr->epoch[g] + 1u; s_bad = 0u; s_budget = hdr->timeout_ticks; }
    __syncthreads();
    const unsigned int e = s_e;
    // (32-bit chunk arithmetic: the host checks n < 2^31; 64-bit divisions by a run-time W were a third of the any-W kernel's code)
    const int n4 = (int)(n / 4);                                                    // whole chunks
    const int nck = (int)((n + 3) / 4);                                             // incl. the partial tail chunk (index n4) if n % 4
    const int per_wg = (nck + W * G - 1) / (W * G);                                 // chunks of one (slice, workgroup) pair
    auto chunk_of = [&](int slice, int k) -> int { return ((k * G + g) * W + slice); };
    float* my_stage = p2p_stage(mine, n_cap, e);
    float* my_result = p2p_result(mine, n_cap, e);
    if constexpr (WC == 0) {      // (the any-W form runs out of scalar registers: keep the loop-invariant pointers in vector registers)
        asm volatile("" : "+v"(my_stage), "+v"(my_result), "+v"(data), "+v"(n_cap), "+v"(hdr));
    }
    // the partial tail chunk (n % 4 floats) is moved element-wise by whoever owns it, through the buffers on every rank
    // (to_comm: local gradient -> own staging half, system-scope stores; else own result half -> local gradient, system-scope loads)
    auto tail_copy = [&](const float* src, float* dst, bool to_comm) {
        for (long long i = 4ll * n4; i < n; ++i) { if (to_comm) st_sys1(dst + i, src[i]); else dst[i] = ld_sys1(src + i); }
    };
    // Bounded wait for flag words [kind][p][g] to reach epoch e on every rank p (thread p polls rank p's word in MY flag page).
    // "Reach", not "equal": a rank that gave up on an epoch and moved on must not make every later wait of a late peer time out
    // as well (the difference is taken modulo 2^32).  A wait that runs out of its budget raises the sticky error word (and the
    // host-visible one) and POISONS this launch's output with NaN, below: a timed-out all-reduce must never look like a result.
    auto wait_all = [&](int kind) {
        const int tw = tid_now();
        if (tw < W) {
            const unsigned int* f = &hdr->flags[(kind * P2P_MAX_WORLD + tw) * P2P_MAX_WG + g];
            const unsigned long long t0 = wall_clock64(), budget = s_budget;
            while ((int)(p2p_flag_load(f) - e) < 0) {
                __builtin_amdgcn_s_sleep(2);
                if (wall_clock64() - t0 > budget) {
                    atomicAdd(&hdr->error, 1u);
                    if (hdr->host_err) __hip_atomic_fetch_add(hdr->host_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    s_bad = 1u;
                    break;
                }
            }
        }
        __syncthreads();                                             // (the payload loads that follow are system-scope themselves)
    };
    auto signal_all = [&](int kind) {
        asm volatile("s_waitcnt vmcnt(0) ; p2p_signal_drain" ::: "memory");   // EVERY wave: its payload stores are acknowledged ...
        __syncthreads();                                             // ... before any flag of this workgroup is raised
        const int tw = tid_now();
        if (tw < W) p2p_flag_store(&((P2PHeader*)s_buf[tw])->flags[(kind * P2P_MAX_WORLD + rank) * P2P_MAX_WG + g], e);
    };
    // this workgroup's chunks of all slices but its own, flattened: j = k * W + slice; eight per lane and trip.
    // (masked-off lanes load from the communication buffer's first chunk - always mapped - and store nothing)
    auto copy_all = [&](const float* src, float* dst, bool to_comm, bool nan_out) {
        const int nj = per_wg * W;
        for (int j0 = tid_now(); j0 < nj; j0 += 8 * P2P_THREADS) {
            f32x4 v[8];
            int c[8];
            const float* a[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int j = j0 + i * P2P_THREADS;
                const int slice = j % W;
                c[i] = chunk_of(slice, j / W);
                const bool ok = j < nj && c[i] < n4 && slice != rank;
                if (!ok) c[i] = -1;
                a[i] = ok ? src + 4 * (size_t)c[i] : my_stage;
            }
            if (to_comm) P2P_LD8_PLAIN(v, a); else P2P_LD8_SYS(v, a);
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (c[i] >= 0) {
                    if (nan_out) v[i] = f32x4{__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf("")};
                    float* dp = dst + 4 * (size_t)c[i];
                    if (to_comm) P2P_ST_SYS(dp, v[i]); else P2P_ST_PLAIN(dp, v[i]);
                }
        }
        if (has_tail() && tid_now() == 0 && (int)((n4 / W) % G) == g) {
            tail_copy(src, dst, to_comm);
            if (nan_out) for (long long i = 4ll * n4; i < n; ++i) dst[i] = __builtin_nanf("");
        }
    };
    // ---- 1. publish my share of every other rank's slice --------------------------------------------------------------------
    copy_all(data, my_stage, true, false);
    signal_all(0);
    // ---- 2. reduce my share of slice `rank` over the ranks, in rank order (p = 0 .. W - 1: run-to-run identical sums; every
    //         rank receives the same bits because each slice is summed once); write it to every rank - my own copy straight into
    //         the gradient bucket.  KB chunks x W ranks = eight loads per lane in flight -----------------------------------------
    wait_all(0);
    const bool bad0 = s_bad != 0u;
    {
        constexpr int WL = WC ? WC : 8;                               // ranks per load batch
        constexpr int KB = 8 / WL;
        for (int k0 = tid_now(); k0 < per_wg; k0 += KB * P2P_THREADS) {
            int c[KB];
            f32x4 acc[KB], own[KB];
#pragma unroll
            for (int kk = 0; kk < KB; ++kk) {                         // my own operand comes from the gradient bucket itself
                const int k = k0 + kk * P2P_THREADS;
                c[kk] = chunk_of(rank, k);
                if (!(k < per_wg && c[kk] < n4)) c[kk] = -1;
                own[kk] = c[kk] >= 0 ? *(const f32x4*)(data + 4 * (size_t)c[kk]) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            for (int p0 = 0; p0 < W; p0 += WL) {                      // (one trip when WC != 0)
                f32x4 v[8];
                const float* a[8];
#pragma unroll
                for (int kk = 0; kk < KB; ++kk)
#pragma unroll
                    for (int pp = 0; pp < WL; ++pp) {
                        const int p = p0 + pp;                        // (slots of my own rank, of ranks >= W and of masked-off
                        char* sb = s_buf[p < W ? p : 0];
                        a[kk * WL + pp] = (p < W && p != rank && c[kk] >= 0)      //  chunks read one always-mapped line)
                                              ? p2p_stage(sb, n_cap, e) + 4 * (size_t)c[kk] : my_stage;
                    }
                P2P_LD8_SYS(v, a);
#pragma unroll
                for (int kk = 0; kk < KB; ++kk) {
#pragma unroll
                    for (int pp = 0; pp < WL; ++pp) {
                        const int p = p0 + pp;
                        if (p >= W) continue;
                        const f32x4 x = p == rank ? own[kk] : v[kk * WL + pp];
                        if (p == 0) acc[kk] = x; else acc[kk] += x;
                    }
                }
            }
#pragma unroll
            for (int kk = 0; kk < KB; ++kk) {
                if (c[kk] < 0) continue;
                if (bad0) acc[kk] = f32x4{__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf("")};
                for (int p = 0; p < W; ++p) {
                    if (p == rank) { float* dp = data + 4 * (size_t)c[kk]; P2P_ST_PLAIN(dp, acc[kk]); }
                    else { float* dp = p2p_result(s_buf[p], n_cap, e) + 4 * (size_t)c[kk]; P2P_ST_SYS(dp, acc[kk]); }
                }
            }
        }
    }
    if (has_tail() && tid_now() == 0 && (int)(n4 % W) == rank && (int)((n4 / W) % G) == g) {      // the partial tail chunk of my slice
        for (long long i = 4ll * n4; i < n; ++i) {
            float acc = 0.f;
            for (int p = 0; p < W; ++p) acc += ld_sys1(p2p_stage(s_buf[p], n_cap, e) + i);
            if (bad0) acc = __builtin_nanf("");
            for (int p = 0; p < W; ++p) st_sys1(p2p_result(s_buf[p], n_cap, e) + i, acc);
        }
    }
    signal_all(1);
    // ---- 3. gather: the other slices' reduced shares of this workgroup back into the gradient bucket ---------------------------
    wait_all(1);
    const bool bad1 = s_bad != 0u;
    copy_all(my_result, data, false, bad1);
    if (bad1) {                                                       // ... and my own slice, already in place, is void as well
        int Gs = G, Ws = W;
        asm volatile("" : "+s"(Gs), "+s"(Ws));      // (re-formed products: W * G kept live to here was the any-W kernel's last spilled SGPR)
        for (int k = tid; k < per_wg; k += P2P_THREADS) {
            const int c = (k * Gs + g) * Ws + rank;
            if (c < n4) *(f32x4*)(data + 4 * (size_t)c) = f32x4{__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf("")};
        }
    }
    if (tid_now() == 0) hdr->epoch[g] = e;
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
    // the wait budget of the kernel's cross-rank waits: minutes by default (a rank that writes a checkpoint, validates or
    // re-captures its graph while the others enter the next step is skew, not failure - RCCL would wait for ever)
    double timeout_s = P2P_DEFAULT_TIMEOUT_S;
    if (const char* ev = getenv("SED_P2P_TIMEOUT_S")) { const double v = atof(ev); if (v > 0.0) timeout_s = v; }
    const unsigned long long ticks = (unsigned long long)(timeout_s * (double)P2P_TICKS_PER_S);
    SED_CHECK_HIP(hipMemcpy((char*)p + offsetof(P2PHeader, timeout_ticks), &ticks, sizeof ticks, hipMemcpyHostToDevice));
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
// Per-buffer configuration (blocking; not for use inside a capture): timeout_s > 0 replaces the wait budget of one cross-rank
// wait (default: SED_P2P_TIMEOUT_S or 600 s); host_err, if not NULL, is the HOST address of a 4-byte word in pinned, mapped host
// memory (hipHostMalloc / torch pin_memory) that every timed-out wait increments besides the sticky counter in the buffer - the
// host can poll it per step without synchronising.  NULL leaves the current word in place; pass clear_host_err != 0 to remove it.
extern "C" int sed_p2p_configure(void* own_ptr, double timeout_s, unsigned int* host_err, int clear_host_err) {
    SED_CHECK_ARG(own_ptr, "sed_p2p_configure: null buffer");
    SED_CHECK_HIP(hipDeviceSynchronize());
    if (timeout_s > 0.0) {
        const unsigned long long ticks = (unsigned long long)(timeout_s * (double)P2P_TICKS_PER_S);
        SED_CHECK_HIP(hipMemcpy((char*)own_ptr + offsetof(P2PHeader, timeout_ticks), &ticks, sizeof ticks, hipMemcpyHostToDevice));
    }
    if (host_err != nullptr || clear_host_err) {
        unsigned int* v = nullptr;
        if (!clear_host_err) SED_CHECK_HIP(hipHostGetDevicePointer((void**)&v, host_err, 0));   // (pinned + mapped, or this fails)
        SED_CHECK_HIP(hipMemcpy((char*)own_ptr + offsetof(P2PHeader, host_err), &v, sizeof v, hipMemcpyHostToDevice));
    }
    SED_CHECK_HIP(hipDeviceSynchronize());
    return SED_OK;
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
    SED_CHECK_ARG(n < (1ll << 31), "sed_p2p_allreduce: at most 2^31 - 1 floats per call");
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
