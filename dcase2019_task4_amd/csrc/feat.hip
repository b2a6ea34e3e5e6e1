// feat.hip - feature front-end: STFT + mel projection, and the log / noise / pad / normalise chain.
//
// Reference ops:
//  * DatasetDcase2019Task4.calculate_mel_spec (baseline/DatasetDcase2019Task4.py:197-231):
//      np.hamming(2048) -> librosa.stft(center=True, pad_mode='reflect') -> |.| ->
//      librosa.feature.melspectrogram(S=|.|, htk=False, norm=None) -> .T -> float32
//    librosa runs this in float64 (soundfile hands it float64 audio); so does k_stft_mel: the
//    front-end is ~0.12 GFLOP per clip, three orders of magnitude below the CRNN, so fp64 costs
//    nothing and keeps parity at the 1e-12 level instead of fp32-FFT noise near the -80 dB floor.
//  * get_transforms chain (baseline/utils/utils.py:397-412; DataLoad.py:262-350; Scaler.py:99-105):
//      [x + |N(0, 0.25)|] -> amplitude_to_db (amin 1e-5, top_db 80 per clip) -> pad/trunc ->
//      float32 -> (x - mean) / std in float64 -> float32
//
// k_stft_mel: one workgroup per frame.  The 2048 windowed samples (reflect-padded on the fly) are
// packed as a 1024-point complex sequence, transformed by a 5-stage radix-4 Stockham FFT held
// entirely in LDS (2 x 16 KB ping-pong, fp64), unpacked to the 1025 real-FFT magnitudes in LDS,
// and projected on the 64 mel filters straight from LDS - the 1025 x 628 spectrogram never
// exists in HBM.  HBM traffic per clip: 640 KB of waveform in (L2-shared between overlapping
// frames), 161 KB of mel out.
#include <math.h>
#include "common.h"
#include "philox.h"
#include "kernels.h"

SED_TS_DEFINE(feat)
#define NFFT 2048
#define NH 1024

// Besides the twiddles and the window: the support [lo, hi) of every mel band.  librosa's triangular filters overlap
// pairwise only, so the dense [n_mels][1025] basis the ABI takes holds ~2 x 1025 non-zeros; walking it densely made every
// frame's workgroup pull 262 KB through L2 (10.5 GB for a batch of 64 clips: 2.3 of the 3.6 ms of a from-waveform step).
// Skipping exact zeros leaves every fp64 partial sum bit-identical (x + 0 * mag == x).
__global__ void k_feat_tables(double2* __restrict__ tw, double* __restrict__ win, const float* __restrict__ window,
                              const float* __restrict__ mel_basis, int n_mels, int* __restrict__ band) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_mels * 64) {      // one wave per band: first / last non-zero bin by ballot
        const int m = i >> 6, lane = i & 63;
        const float* row = mel_basis + (size_t)m * (NH + 1);
        int lo = NH + 1, hi = 0;
        for (int k = lane; k <= NH; k += 64)
            if (row[k] != 0.f) { lo = min(lo, k); hi = max(hi, k + 1); }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { lo = min(lo, __shfl_xor(lo, o)); hi = max(hi, __shfl_xor(hi, o)); }
        if (lane == 0) { band[2 * m] = min(lo, hi); band[2 * m + 1] = hi; }
    }
    if (i < NFFT) {
        double s, c;
        sincospi(-2.0 * (double)i / (double)NFFT, &s, &c);      // W_2048^i = exp(-2 pi i / 2048)
        tw[i] = make_double2(c, s);
        // np.hamming(n): 0.54 - 0.46 cos(2 pi k / (n - 1))   (symmetric)
        win[i] = window ? (double)window[i] : 0.54 - 0.46 * cospi(2.0 * (double)i / (double)(NFFT - 1));
    }
}

__device__ __forceinline__ double2 cmul(double2 a, double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

__global__ __launch_bounds__(256) void k_stft_mel(const float* __restrict__ wave, int n_samples, int hop, int frames,
                                                   const double2* __restrict__ tw, const double* __restrict__ win,
                                                   const float* __restrict__ mel_basis, const int* __restrict__ band, int n_mels,
                                                   float* __restrict__ mel) {
    __shared__ double2 bufA[NH];
    __shared__ double2 bufB[NH];
    const int tid = threadIdx.x;
    const int f = blockIdx.x, clip = blockIdx.y;
    const float* w = wave + (size_t)clip * n_samples;
    // ---- windowed, reflect-padded frame packed as z[m] = x[2m] + i x[2m+1] ------------------------
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int m = tid + 256 * it;
        double v[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int n = 2 * m + h;
            int idx = f * hop + n - NFFT / 2;
            if (idx < 0) idx = -idx;
            if (idx >= n_samples) idx = 2 * (n_samples - 1) - idx;
            v[h] = (double)w[idx] * win[n];
        }
        bufA[m] = make_double2(v[0], v[1]);
    }
    __syncthreads();
    // ---- radix-4 Stockham, N = 1024 -------------------------------------------------------------
    double2* src = bufA;
    double2* dst = bufB;
#pragma unroll
    for (int p = 1; p < NH; p <<= 2) {
        const int i = tid, k = i & (p - 1), j = ((i - k) << 2) + k;
        const int tstep = 512 / p;
        const double2 u0 = src[i];
        const double2 u1 = cmul(src[i + 256], tw[(k * tstep) & (NFFT - 1)]);
        const double2 u2 = cmul(src[i + 512], tw[(2 * k * tstep) & (NFFT - 1)]);
        const double2 u3 = cmul(src[i + 768], tw[(3 * k * tstep) & (NFFT - 1)]);
        const double2 a02 = make_double2(u0.x + u2.x, u0.y + u2.y), s02 = make_double2(u0.x - u2.x, u0.y - u2.y);
        const double2 a13 = make_double2(u1.x + u3.x, u1.y + u3.y), s13 = make_double2(u1.x - u3.x, u1.y - u3.y);
        dst[j] = make_double2(a02.x + a13.x, a02.y + a13.y);
        dst[j + p] = make_double2(s02.x + s13.y, s02.y - s13.x);          // u0 - i u1 - u2 + i u3
        dst[j + 2 * p] = make_double2(a02.x - a13.x, a02.y - a13.y);
        dst[j + 3 * p] = make_double2(s02.x - s13.y, s02.y + s13.x);      // u0 + i u1 - u2 - i u3
        __syncthreads();
        double2* t = src; src = dst; dst = t;
    }
    // ---- real-FFT unpack -> magnitudes (into dst, reinterpreted as double[]) ----------------------
    double* mag = (double*)dst;
    for (int k = tid; k <= NH; k += 256) {
        const double2 Zk = src[k & (NH - 1)];
        const double2 Zm = src[(NH - k) & (NH - 1)];
        const double2 Zc = make_double2(Zm.x, -Zm.y);
        const double2 e = make_double2(0.5 * (Zk.x + Zc.x), 0.5 * (Zk.y + Zc.y));
        const double2 o = make_double2(0.5 * (Zk.x - Zc.x), 0.5 * (Zk.y - Zc.y));
        const double2 t = cmul(tw[k], o);                                   // W^k * o
        const double re = e.x + t.y, im = e.y - t.x;                        // e - i * t
        mag[k] = sqrt(re * re + im * im);
    }
    __syncthreads();
    // ---- mel projection: thread = (mel m, quarter q) ----------------------------------------------
    for (int m0 = 0; m0 < n_mels; m0 += 64) {
        const int m = m0 + (tid >> 2), q = tid & 3;
        double s = 0.0;
        if (m < n_mels) {
            const float* row = mel_basis + (size_t)m * (NH + 1);
            const int lo = band[2 * m], hi = band[2 * m + 1];
            // same partition of the bins over the 4 lanes as the dense walk (k = q mod 4), restricted to the band's support
            for (int k = lo + ((q - lo) & 3); k < hi; k += 4) s += (double)row[k] * mag[k];
        }
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        if (q == 0 && m < n_mels) mel[((size_t)clip * frames + f) * n_mels + m] = (float)s;
    }
}

// ---- round 3: one WAVE per frame, radix 16 x 16 x 4 with the 16-point transforms in registers ------------------------------
// k_stft_mel above moves 16 KB through LDS five times per frame with ds_write_b128 at 64-byte strides (4-way bank
// conflicts; the LDS write path peaks at ~80 B / clock / CU anyway): 444 us for a batch of 64 clips against ~36 us of fp64
// arithmetic - the LDS write path is what bounds it.  Here a frame belongs to ONE wave (no workgroup barriers), thread t
// loads its 16 samples z[64 n1 + t] straight from global memory into registers, and 1024 = 16 x 16 x 4:
//   A  16-point DFT over n1 in registers, twiddle W_1024^(t k1)                 -> exchange 1 (LDS, [k1][t], rows of 65)
//   B1 thread (k1, m2): 16-point DFT over m1 of A[4 m1 + m2][k1], twiddle W_64^(m2 j1) -> exchange 2 ([m2][17 k1 + j1])
//   B2 thread (k1, j1 = T / 16 + 4 i): 4-point DFT over m2 -> X[k1 + 16 (j1 + 16 j2)]  -> exchange 3 (natural order)
// then the same real-FFT unpack and banded mel projection as above.  Three 16 KB LDS round trips instead of five, every
// one of them conflict-free by layout (the paddings 65 / 17 put the 16 lanes of a ds_*_b128 group on distinct bank groups).
struct cd { double x, y; };
__device__ __forceinline__ cd cdmul(cd a, cd b) { return cd{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
// forward 4-point DFT in place (W_4 = -i): a <- a + b + c + d, b <- (a - c) - i (b - d), c <- a - b + c - d, d <- (a - c) + i (b - d)
__device__ __forceinline__ void dft4(cd& a, cd& b, cd& c, cd& d) {
    const cd s02 = {a.x + c.x, a.y + c.y}, d02 = {a.x - c.x, a.y - c.y};
    const cd s13 = {b.x + d.x, b.y + d.y}, d13 = {b.x - d.x, b.y - d.y};
    a = cd{s02.x + s13.x, s02.y + s13.y};
    c = cd{s02.x - s13.x, s02.y - s13.y};
    b = cd{d02.x + d13.y, d02.y - d13.x};
    d = cd{d02.x - d13.y, d02.y + d13.x};
}
// forward 16-point DFT in place; afterwards X[k] sits in v[(k >> 2) + 4 (k & 3)]
__device__ __forceinline__ void dft16(cd (&v)[16]) {
    constexpr double C8 = 0.92387953251128673848, S8 = 0.38268343236508978178, R2 = 0.70710678118654752440;
#pragma unroll
    for (int n0 = 0; n0 < 4; ++n0) dft4(v[n0], v[n0 + 4], v[n0 + 8], v[n0 + 12]);        // v[n0 + 4 k0] = t[n0][k0]
    // t[n0][k0] *= W_16^(n0 k0)
    v[1 + 4] = cdmul(v[1 + 4], cd{C8, -S8});     // 1
    v[1 + 8] = cdmul(v[1 + 8], cd{R2, -R2});     // 2
    v[1 + 12] = cdmul(v[1 + 12], cd{S8, -C8});   // 3
    v[2 + 4] = cdmul(v[2 + 4], cd{R2, -R2});     // 2
    v[2 + 8] = cd{v[2 + 8].y, -v[2 + 8].x};      // 4: -i
    v[2 + 12] = cdmul(v[2 + 12], cd{-R2, -R2});  // 6
    v[3 + 4] = cdmul(v[3 + 4], cd{S8, -C8});     // 3
    v[3 + 8] = cdmul(v[3 + 8], cd{-R2, -R2});    // 6
    v[3 + 12] = cdmul(v[3 + 12], cd{-C8, S8});   // 9
#pragma unroll
    for (int k0 = 0; k0 < 4; ++k0) dft4(v[4 * k0], v[4 * k0 + 1], v[4 * k0 + 2], v[4 * k0 + 3]);   // X[k0 + 4 k1] at v[k1 + 4 k0]
}
#define DFT16_AT(k) (((k) >> 2) + 4 * ((k) & 3))

__global__ __launch_bounds__(64) void k_stft_mel16(const float* __restrict__ wave, int n_samples, int hop, int frames,
                                                    const double2* __restrict__ tw, const double* __restrict__ win,
                                                    const float* __restrict__ mel_basis, const int* __restrict__ band, int n_mels,
                                                    float* __restrict__ mel) {
    __shared__ __attribute__((aligned(16))) cd buf[4 * 272];            // exchange 1: 16 x 65, exchange 2: 4 x 272, exchange 3: 1024
    __shared__ double mag[NH + 1];
    const int t = threadIdx.x;
    const int f = blockIdx.x, clip = blockIdx.y;
    const float* w = wave + (size_t)clip * n_samples;
    auto twd = [&](int i) { const double2 v = tw[i]; return cd{v.x, v.y}; };
    cd v[16];
    TSC(0);
    // ---- windowed, reflect-padded frame: z[m] = x[2m] + i x[2m + 1], m = 64 n1 + t ---------------------------------------------
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) {
        const int m = 64 * n1 + t;
        double s[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int n = 2 * m + h;
            int idx = f * hop + n - NFFT / 2;
            if (idx < 0) idx = -idx;
            if (idx >= n_samples) idx = 2 * (n_samples - 1) - idx;
            s[h] = (double)w[idx] * win[n];
        }
        v[n1] = cd{s[0], s[1]};
    }
    TSC(1);
    // ---- A ---------------------------------------------------------------------------------------------------------------------
    dft16(v);
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) {
        const cd a = (k1 == 0) ? v[DFT16_AT(0)] : cdmul(v[DFT16_AT(k1)], twd(2 * t * k1));     // W_1024^(t k1) = W_2048^(2 t k1)
        buf[k1 * 65 + t] = a;
    }
    TSC(2);
    __syncthreads();
    // ---- B1 --------------------------------------------------------------------------------------------------------------------
    const int k1l = t & 15, m2 = t >> 4;
#pragma unroll
    for (int m1 = 0; m1 < 16; ++m1) v[m1] = buf[k1l * 65 + 4 * m1 + m2];
    dft16(v);
    __syncthreads();
#pragma unroll
    for (int j1 = 0; j1 < 16; ++j1) {
        const cd b = (j1 == 0) ? v[DFT16_AT(0)] : cdmul(v[DFT16_AT(j1)], twd(32 * m2 * j1));    // W_64^(m2 j1) = W_2048^(32 m2 j1)
        buf[m2 * 272 + k1l * 17 + j1] = b;
    }
    __syncthreads();
    TSC(3);
    // ---- B2: thread (k1 = t & 15, j1 = (t >> 4) + 4 i) ---------------------------------------------------------------------------
    cd x4[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int j1 = (t >> 4) + 4 * i;
#pragma unroll
        for (int q = 0; q < 4; ++q) x4[i][q] = buf[q * 272 + k1l * 17 + j1];
        dft4(x4[i][0], x4[i][1], x4[i][2], x4[i][3]);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int j1 = (t >> 4) + 4 * i;
#pragma unroll
        for (int j2 = 0; j2 < 4; ++j2) buf[k1l + 16 * (j1 + 16 * j2)] = x4[i][j2];
    }
    __syncthreads();
    TSC(4);
    // ---- real-FFT unpack -> magnitudes ---------------------------------------------------------------------------------------------
    auto unpack = [&](int k) {
        const cd Zk = buf[k & (NH - 1)];
        const cd Zm = buf[(NH - k) & (NH - 1)];
        const cd Zc = {Zm.x, -Zm.y};
        const cd e = {0.5 * (Zk.x + Zc.x), 0.5 * (Zk.y + Zc.y)};
        const cd o = {0.5 * (Zk.x - Zc.x), 0.5 * (Zk.y - Zc.y)};
        const cd tt = cdmul(twd(k), o);                                  // W^k * o
        const double re = e.x + tt.y, im = e.y - tt.x;                   // e - i * t
        mag[k] = sqrt(re * re + im * im);
    };
#pragma unroll
    for (int i = 0; i < 16; ++i) unpack(t + 64 * i);                     // (unrolled: the 16 twiddle loads are issued together)
    if (t == 0) unpack(NH);
    TSC(5);
    __syncthreads();
    // ---- mel projection: thread = (mel m, quarter q), 16 bands per pass --------------------------------------------------------------
    for (int m0 = 0; m0 < n_mels; m0 += 16) {
        const int m = m0 + (t >> 2), q = t & 3;
        double sacc = 0.0;
        if (m < n_mels) {
            const float* row = mel_basis + (size_t)m * (NH + 1);
            const int lo = band[2 * m], hi = band[2 * m + 1];
            // 8 weights per trip, all eight loads issued before the first is used (clamped index, zero weight past the band)
            for (int k0 = lo + ((q - lo) & 3); k0 < hi; k0 += 32) {
                float wv[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) wv[i] = row[min(k0 + 4 * i, NH)];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int k = k0 + 4 * i;
                    sacc += (k < hi) ? (double)wv[i] * mag[min(k, NH)] : 0.0;
                }
            }
        }
        sacc += __shfl_xor(sacc, 1);
        sacc += __shfl_xor(sacc, 2);
        if (q == 0 && m < n_mels) mel[((size_t)clip * frames + f) * n_mels + m] = (float)sacc;
    }
    TSC(6);
}

// ---- log / noise / pad / normalise ----------------------------------------------------------------
__device__ __forceinline__ double teacher_noise(uint32_t e_global, uint64_t seed) {
    const u32x4 o = philox4x32_10(e_global >> 1, 0u, 16u, PHILOX_TAG, (uint32_t)seed, (uint32_t)(seed >> 32));
    const uint32_t w0 = (e_global & 1) ? o.z : o.x, w1 = (e_global & 1) ? o.w : o.y;
    const double u1 = ((double)w0 + 1.0) * 2.3283064365386963e-10, u2 = (double)w1 * 2.3283064365386963e-10;
    const double g = sqrt(-2.0 * log(u1)) * cospi(2.0 * u2);
    return (double)(float)fabs(0.25 * g);
}
__device__ __forceinline__ double amp_db(double a) {
    // librosa.amplitude_to_db(ref=1, amin=1e-5): 10 log10(max(amin^2, a^2)) - 10 log10(max(amin^2, 1))
    return 10.0 * log10(fmax(1e-10, a * a));
}

// Two launches over the WHOLE chip instead of one workgroup per clip (round 2: 64 of 256 CUs, one wave per SIMD, the
// Box-Muller noise formed twice per element: 283 us for a batch of 64 clips):
//   k_logmel_max   LM_CHUNKS workgroups per clip: partial maxima of |mel| and |mel + noise| over the clip (amplitude_to_db
//                  clamps at max - 80 dB over the WHOLE clip, before padding) -> part[clip][chunk][2]; the noise values it
//                  draws are fp32 by definition (AugmentGaussianNoise adds a float32 array, DataLoad.py:189-207) and are
//                  parked in out_noisy at the element's own output position, so the second pass does not redo Philox +
//                  log + sqrt + cospi in fp64
//   k_logmel_apply every output element: dB, clamp, pad, normalise; folds the LM_CHUNKS partials of its clip itself
// No atomics, no initialisation, fixed reduction order: bit-reproducible.
#define LM_CHUNKS 16
__global__ __launch_bounds__(256) void k_logmel_max(const float* __restrict__ mel, int frames, int n_mels, int max_frames,
                                                     const uint64_t* __restrict__ seed_ptr, double* __restrict__ part,
                                                     float* __restrict__ out_noisy) {
    __shared__ double red[2][4];
    const int tid = threadIdx.x, chunk = blockIdx.x, clip = blockIdx.y, lane = tid & 63, wv = tid >> 6;
    const int n = frames * n_mels, n_keep = min(frames, max_frames) * n_mels;
    const float* src = mel + (size_t)clip * n;
    const uint64_t seed = out_noisy ? seed_ptr[0] : 0ull;
    const int per = (n + LM_CHUNKS - 1) / LM_CHUNKS, e0 = chunk * per, e1 = min(n, e0 + per);
    double mc = 0.0, mn = 0.0;
    for (int e = e0 + tid; e < e1; e += 256) {
        const double a = (double)src[e];
        mc = fmax(mc, fabs(a));
        if (out_noisy) {
            const double nz = teacher_noise((uint32_t)(clip * n + e), seed);
            mn = fmax(mn, fabs(a + nz));
            if (e < n_keep) out_noisy[(size_t)clip * max_frames * n_mels + e] = (float)nz;       // exact: nz is a float
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mc = fmax(mc, __shfl_xor(mc, o)); mn = fmax(mn, __shfl_xor(mn, o)); }
    if (lane == 0) { red[0][wv] = mc; red[1][wv] = mn; }
    __syncthreads();
    if (tid < 2) part[((size_t)clip * LM_CHUNKS + chunk) * 2 + tid] = fmax(fmax(red[tid][0], red[tid][1]), fmax(red[tid][2], red[tid][3]));
}

__global__ __launch_bounds__(256) void k_logmel_apply(const float* __restrict__ mel, int frames, int n_mels, int max_frames,
                                                       const double* __restrict__ mean, const double* __restrict__ stdv,
                                                       const double* __restrict__ part, float* __restrict__ out_clean,
                                                       float* __restrict__ out_noisy) {
    const int tid = threadIdx.x, clip = blockIdx.y;
    const int n = frames * n_mels, n_out = max_frames * n_mels;
    const float* src = mel + (size_t)clip * n;
    double mc = 0.0, mn = 0.0;
#pragma unroll
    for (int k = 0; k < LM_CHUNKS; ++k) {
        mc = fmax(mc, part[((size_t)clip * LM_CHUNKS + k) * 2]);
        mn = fmax(mn, part[((size_t)clip * LM_CHUNKS + k) * 2 + 1]);
    }
    const double floor_c = amp_db(mc) - 80.0, floor_n = amp_db(mn) - 80.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = (blockIdx.x * 4 + i) * 256 + tid;
        if (e >= n_out) break;
        const int t = e / n_mels, m = e - t * n_mels;
        float vc = 0.f, vn = 0.f;                               // PadOrTrunc pads with 0 (dB) AFTER the log
        if (t < frames) {
            const double a = (double)src[e];
            vc = (float)fmax(amp_db(a), floor_c);               // ToTensor: .float()
            if (out_noisy) vn = (float)fmax(amp_db(a + (double)out_noisy[(size_t)clip * n_out + e]), floor_n);
        }
        if (mean) {                                             // Scaler.normalize in float64, torch.Tensor() -> fp32
            vc = (float)(((double)vc - mean[m]) / stdv[m]);
            vn = (float)(((double)vn - mean[m]) / stdv[m]);
        }
        out_clean[(size_t)clip * n_out + e] = vc;
        if (out_noisy) out_noisy[(size_t)clip * n_out + e] = vn;
    }
}

extern "C" size_t sed_mel_spec_ws_bytes(int n_clips, int n_samples, int hop, int n_fft, int n_mels) {
    (void)n_clips; (void)n_samples; (void)hop; (void)n_mels;
    if (n_fft != NFFT) return 0;
    return (size_t)NFFT * sizeof(double2) + (size_t)NFFT * sizeof(double) + (size_t)2 * (n_mels > 0 ? n_mels : 0) * sizeof(int);
}

extern "C" int sed_mel_spec(const float* wave, int n_clips, int n_samples, int hop, int n_fft, const float* window,
                            const float* mel_basis, int n_mels, float* mel, void* ws, size_t ws_bytes, void* stream) {
    SED_CHECK_ARG(wave && mel_basis && mel && ws, "sed_mel_spec: null argument");
    if (n_fft != NFFT) {
        sed_set_error("sed_mel_spec: n_fft must be 2048 (config.py:18), got %d", n_fft);
        return SED_ERR_UNSUPPORTED;
    }
    SED_CHECK_ARG(n_clips >= 1 && hop >= 1 && n_mels >= 1 && n_samples > NFFT / 2, "sed_mel_spec: bad sizes (reflect padding needs n_samples > n_fft/2)");
    if (ws_bytes < sed_mel_spec_ws_bytes(n_clips, n_samples, hop, n_fft, n_mels)) {
        sed_set_error("sed_mel_spec: workspace too small");
        return SED_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    double2* tw = (double2*)ws;
    double* win = (double*)((char*)ws + (size_t)NFFT * sizeof(double2));
    int* band = (int*)((char*)ws + (size_t)NFFT * sizeof(double2) + (size_t)NFFT * sizeof(double));
    const int tb = (NFFT > n_mels * 64 ? NFFT : n_mels * 64);
    k_feat_tables<<<(tb + 255) / 256, 256, 0, st>>>(tw, win, window, mel_basis, n_mels, band);
    SED_CHECK_LAUNCH();
    const int frames = 1 + n_samples / hop;
    // (debug bit 19: the round-2 kernel - one 256-thread workgroup per frame, five radix-4 passes through LDS - for A/B timing)
    if (g_sed_debug & 524288) k_stft_mel<<<dim3(frames, n_clips), 256, 0, st>>>(wave, n_samples, hop, frames, tw, win, mel_basis, band, n_mels, mel);
    else k_stft_mel16<<<dim3(frames, n_clips), 64, 0, st>>>(wave, n_samples, hop, frames, tw, win, mel_basis, band, n_mels, mel);
    SED_CHECK_LAUNCH();
    return SED_OK;
}

// The front-end's own noise-key chain: key += golden-ratio stride, one thread.  Lets a caller that computes the NEXT batch's
// features on a side stream (features.WaveformFrontEnd) advance the key in stream order without touching the train step's
// device state, which the step itself advances concurrently.
__global__ void k_seed_advance(uint64_t* key) { key[0] += 0x9E3779B97F4A7C15ull; }
extern "C" int sed_seed_advance(uint64_t* key_dev, void* stream) {
    SED_CHECK_ARG(key_dev != nullptr, "sed_seed_advance: null key");
    k_seed_advance<<<1, 1, 0, (hipStream_t)stream>>>(key_dev);
    SED_CHECK_LAUNCH();
    return SED_OK;
}

extern "C" size_t sed_logmel_transform_ws_bytes(int n_clips) { return (size_t)(n_clips > 0 ? n_clips : 0) * LM_CHUNKS * 2 * sizeof(double); }

extern "C" int sed_logmel_transform(const float* mel, int n_clips, int frames, int n_mels, int max_frames,
                                    const double* mean, const double* std, const uint64_t* seed_dev, float* out_clean,
                                    float* out_noisy, void* ws, size_t ws_bytes, void* stream) {
    SED_CHECK_ARG(mel && out_clean, "sed_logmel_transform: null argument");
    SED_CHECK_ARG((mean == nullptr) == (std == nullptr), "sed_logmel_transform: mean and std go together");
    SED_CHECK_ARG(!out_noisy || seed_dev, "sed_logmel_transform: noise requested but seed_dev is null");
    SED_CHECK_ARG(n_clips >= 1 && frames >= 1 && n_mels >= 1 && max_frames >= 1, "sed_logmel_transform: bad sizes");
    SED_CHECK_ARG((int64_t)n_clips * frames * n_mels < (1ll << 32), "sed_logmel_transform: too many elements for the noise stream");
    SED_CHECK_ARG(ws != nullptr, "sed_logmel_transform: null workspace");
    if (ws_bytes < sed_logmel_transform_ws_bytes(n_clips)) {
        sed_set_error("sed_logmel_transform: workspace has %zu bytes, needs %zu", ws_bytes, sed_logmel_transform_ws_bytes(n_clips));
        return SED_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    k_logmel_max<<<dim3(LM_CHUNKS, n_clips), 256, 0, st>>>(mel, frames, n_mels, max_frames, seed_dev, (double*)ws, out_noisy);
    SED_CHECK_LAUNCH();
    const int n_out = max_frames * n_mels;
    k_logmel_apply<<<dim3((n_out + 1023) / 1024, n_clips), 256, 0, st>>>(mel, frames, n_mels, max_frames, mean, std, (const double*)ws,
                                                                         out_clean, out_noisy);
    SED_CHECK_LAUNCH();
    return SED_OK;
}

// ---- Scaler statistics pass (baseline/utils/Scaler.py:34-87) ---------------------------------------------------------
// The reference walks the whole training set on the host before training starts (main.py:249-250) and averages the
// per-clip per-band means of x and x^2.  All clips have the same shape there (it raises otherwise), so that is the
// plain column mean over every row of the set: one streaming pass, fp64 accumulation, per-workgroup partial sums in
// registers -> LDS -> one fp64 atomic per column per workgroup.
__global__ __launch_bounds__(256) void k_scaler_stats(const float* __restrict__ x, long long n_rows, int n_cols,
                                                       double* __restrict__ sums /* [2][n_cols] */) {
    __shared__ double red[2][256];
    const int tid = threadIdx.x;
    const int rows_per_pass = 256 / n_cols;              // n_cols divides 256 (checked by the launcher)
    const int c = tid % n_cols, r0 = tid / n_cols;
    double s = 0.0, q = 0.0;
    for (long long r = (long long)blockIdx.x * rows_per_pass + r0; r < n_rows; r += (long long)gridDim.x * rows_per_pass) {
        const double v = (double)x[r * n_cols + c];
        s += v;
        q += v * v;
    }
    red[0][tid] = s; red[1][tid] = q;
    __syncthreads();
    if (tid < n_cols) {
        double a = 0.0, b = 0.0;
        for (int k = 0; k < rows_per_pass; ++k) { a += red[0][tid + k * n_cols]; b += red[1][tid + k * n_cols]; }
        atomicAdd(&sums[tid], a);
        atomicAdd(&sums[n_cols + tid], b);
    }
}

extern "C" int sed_scaler_stats(const float* x, long long n_rows, int n_cols, double* sums, void* stream) {
    SED_CHECK_ARG(x && sums && n_rows >= 1, "sed_scaler_stats: bad argument");
    SED_CHECK_ARG(n_cols >= 1 && n_cols <= 256 && 256 % n_cols == 0, "sed_scaler_stats: n_cols must divide 256");
    const int rpp = 256 / n_cols;
    long long blocks = (n_rows + rpp - 1) / rpp;
    if (blocks > 2048) blocks = 2048;
    k_scaler_stats<<<(int)blocks, 256, 0, (hipStream_t)stream>>>(x, n_rows, n_cols, sums);
    SED_CHECK_LAUNCH();
    return SED_OK;
}

// ---- band-limited resampling (read_audio, baseline/utils/utils.py:175-193) ------------------------------------------
// librosa.resample(audio, orig_sr, target_sr) of the reference's era = resampy's windowed-sinc interpolation
// (resampy/interpn.py resample_f) + fix_length.  Every output sample is an independent dot product of the input with
// the filter table sampled at a fractional offset (linear interpolation between table entries): one thread per output
// sample, fp64 like the reference, the 256 KB table stays in L2.  ~350 taps per sample for 44.1 kHz -> 16 kHz.
__global__ __launch_bounds__(256) void k_resample(const double* __restrict__ x, int n_in, double ratio,
                                                   const double* __restrict__ win, int nwin, int num_table,
                                                   const double* __restrict__ time_reg, double* __restrict__ y,
                                                   int n_resampled, int n_out) {
    const int clip = blockIdx.y;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n_out) return;
    const double* xc = x + (size_t)clip * n_in;
    double acc = 0.0;
    if (t < n_resampled) {                       // samples beyond resampy's own length are fix_length's zero padding
        const double scale = ratio < 1.0 ? ratio : 1.0;
        const int index_step = (int)(scale * num_table);
        // resampy accumulates time_register += 1/ratio sample after sample; the rounding of that running sum decides
        // on which side of an input sample an output instant falls, and - because the table step is truncated to an
        // integer - the two sides differ by ~1e-4.  The caller passes the exact running sums (a host cumsum).
        const double time_register = time_reg ? time_reg[t] : (double)t * (1.0 / ratio);
        const int n = (int)time_register;
        double frac = scale * (time_register - n);
        double index_frac = frac * num_table;
        int offset = (int)index_frac;
        double eta = index_frac - offset;
        int i_max = min(n + 1, (nwin - offset) / index_step);
        for (int i = 0; i < i_max; ++i) {
            const int j = offset + i * index_step;
            const double w0 = win[j], w1 = (j + 1 < nwin) ? win[j + 1] : w0;      // interp_delta[last] = 0
            acc += (w0 + eta * (w1 - w0)) * xc[n - i];
        }
        frac = scale - frac;
        index_frac = frac * num_table;
        offset = (int)index_frac;
        eta = index_frac - offset;
        int k_max = min(n_in - n - 1, (nwin - offset) / index_step);
        for (int k = 0; k < k_max; ++k) {
            const int j = offset + k * index_step;
            const double w0 = win[j], w1 = (j + 1 < nwin) ? win[j + 1] : w0;
            acc += (w0 + eta * (w1 - w0)) * xc[n + k + 1];
        }
    }
    y[(size_t)clip * n_out + t] = acc;
}

extern "C" int sed_resample(const double* x, int n_clips, int n_in, double ratio, const double* interp_win, int nwin,
                            int num_table, const double* time_reg, double* y, int n_out, void* stream) {
    SED_CHECK_ARG(x && interp_win && y, "sed_resample: null argument");
    SED_CHECK_ARG(n_clips >= 1 && n_in >= 1 && n_out >= 1 && ratio > 0.0 && nwin >= 2 && num_table >= 1, "sed_resample: bad sizes");
    const double scale = ratio < 1.0 ? ratio : 1.0;
    SED_CHECK_ARG((int)(scale * num_table) >= 1, "sed_resample: ratio too small for this filter table");
    const int n_resampled = (int)((double)n_in * ratio);
    dim3 grid((n_out + 255) / 256, n_clips);
    k_resample<<<grid, 256, 0, (hipStream_t)stream>>>(x, n_in, ratio, interp_win, nwin, num_table, time_reg, y, n_resampled, n_out);
    SED_CHECK_LAUNCH();
    return SED_OK;
}
