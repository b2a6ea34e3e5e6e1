// philox.h - Philox4x32-10 counter RNG and the dropout / noise stream definition.
// Mirrored bit-for-bit by oracle/philox.py (test infrastructure) so train-mode parity tests run
// with dropout enabled.  Stands in for nn.Dropout's generator (models/CNN.py:59-61, CRNN.py:74)
// and np.random.normal in AugmentGaussianNoise (DataLoad.py:285).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PHILOX_M0 0xD2511F53u
#define PHILOX_M1 0xCD9E8D57u
#define PHILOX_W0 0x9E3779B9u
#define PHILOX_W1 0xBB67AE85u
#define PHILOX_TAG 0x5ED0u

struct u32x4 { uint32_t x, y, z, w; };

__device__ __forceinline__ u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                               uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        uint32_t hi0 = __umulhi(PHILOX_M0, c0), lo0 = PHILOX_M0 * c0;
        uint32_t hi1 = __umulhi(PHILOX_M1, c2), lo1 = PHILOX_M1 * c2;
        uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += PHILOX_W0; k1 += PHILOX_W1;
    }
    u32x4 r = {c0, c1, c2, c3};
    return r;
}

// 16 x 8-bit uniforms for (index, stream): byte i = (word[i>>2] >> 8*(i&3)) & 0xff.
// keep(byte) = byte >= thresh8(p) with thresh8 = round(p * 256); kept values are scaled by
// 256 / (256 - thresh8)  (the exact keep probability of the 8-bit draw; p = 0.5 and 0.25 are exact).
__device__ __forceinline__ u32x4 philox_stream(uint32_t index, uint32_t stream, uint64_t seed) {
    return philox4x32_10(index, 0u, stream, PHILOX_TAG, (uint32_t)seed, (uint32_t)(seed >> 32));
}
__device__ __forceinline__ uint32_t philox_byte(const u32x4& o, int i) {
    uint32_t w = (i >> 2) == 0 ? o.x : (i >> 2) == 1 ? o.y : (i >> 2) == 2 ? o.z : o.w;
    return (w >> (8 * (i & 3))) & 0xffu;
}
__host__ __device__ __forceinline__ uint32_t drop_thresh8(float p) { return (uint32_t)(p * 256.0f + 0.5f); }
__device__ __forceinline__ float drop_scale8(float p) { return 256.0f / (256.0f - (float)drop_thresh8(p)); }
// 16 keep bits (bit i = byte i kept) of one draw
__device__ __forceinline__ uint32_t philox_keep16(const u32x4& o, uint32_t thr) {
    uint32_t m = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) m |= (philox_byte(o, i) >= thr ? 1u : 0u) << i;
    return m;
}

// p == 0.5 (thresh8 == 128, the reference's dropout rate) needs ONE random bit per element, so a single draw
// serves 8 (row block, channel half) units of a lane instead of 1:
//   draw(index = (rb >> 2) * 64 + lane, stream = 32 + block)  ->  eight 16-bit fields,
//   field (rb & 3) * 2 + h = the keep bits of row block rb, channels 32h..32h+31, this lane
//   (bit r <-> pooled pixel r >> 2, df = r & 3, as in the 8-bit stream).
// The 32-bit multiplies of Philox are quarter rate; with 8-bit draws they were ~40 % of the VALU cycles of
// the block-0 forward kernel.
#define PHILOX_STREAM_1BIT 32u
__device__ __forceinline__ u32x4 philox_stream_1bit(uint32_t rb, int lane, int block, uint64_t seed) {
    return philox_stream((rb >> 2) * 64u + (uint32_t)lane, PHILOX_STREAM_1BIT + (uint32_t)block, seed);
}
__device__ __forceinline__ uint32_t philox_field16(const u32x4& o, int f) {
    const int wi = f >> 1;
    const uint32_t w = wi == 0 ? o.x : wi == 1 ? o.y : wi == 2 ? o.z : o.w;
    return (w >> (16 * (f & 1))) & 0xffffu;
}
