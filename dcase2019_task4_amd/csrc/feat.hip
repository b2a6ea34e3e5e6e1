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

// Second table pass (one workgroup), for the persistent kernel's mel projection: bands are served 16 at a time (4 lanes per
// band, lane q takes bins lo + q + 4 j), every band of a 16-band pass padded with zero weights to the pass's trip count
// (a multiple of 4), so that the whole wave runs the same fully unrolled chunks with no predicate.
//   band[2 n_mels + p]      offset of pass p's weights in melw, laid out [trip j][lane = 4 (m - 16 p) + q]: lane t of the
//                           wave reads melw[off + 64 j + t] - consecutive addresses, no bank conflict (band-major rows of
//                           4 * trips floats put all 16 bands of a pass on the same banks: 16-way conflicts, measured)
//   band[3 n_mels]          total padded size (0: does not fit -> the kernel walks the dense basis)
//   band[3 n_mels + 1 + p]  trips of pass p
__global__ __launch_bounds__(256) void k_feat_pack(const float* __restrict__ mel_basis, int n_mels, int* __restrict__ band,
                                                    float* __restrict__ melw, int cap) {
    __shared__ int s_total;
    const int n_pass = (n_mels + 15) / 16;
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int p = 0; p < n_pass; ++p) {
            int w = 0;
            for (int m = 16 * p; m < min(n_mels, 16 * p + 16); ++m) w = max(w, band[2 * m + 1] - band[2 * m]);
            const int trips = ((w + 3) / 4 + 3) & ~3;
            band[3 * n_mels + 1 + p] = trips;
            band[2 * n_mels + p] = acc;
            acc += 64 * trips;
            // the kernel reads magnitudes at lo + q + 4 j without a clamp: the padded trips must stay inside the wave's buffer
            for (int m = 16 * p; m < min(n_mels, 16 * p + 16); ++m)
                if (band[2 * m] + 3 + 4 * (trips - 1) >= 1056) acc = cap + 1;
        }
        if (acc > cap || n_mels > 64) acc = 0;
        band[3 * n_mels] = acc;
        s_total = acc;
    }
    __syncthreads();
    if (s_total == 0) return;
    for (int p = 0; p < n_pass; ++p) {
        const int off = band[2 * n_mels + p], n = 64 * band[3 * n_mels + 1 + p];
        for (int i = threadIdx.x; i < n; i += 256) {
            const int j = i >> 6, m = 16 * p + ((i & 63) >> 2), q = i & 3;
            float w = 0.f;
            if (m < n_mels) {
                const int k = band[2 * m] + q + 4 * j;
                if (k < band[2 * m + 1]) w = mel_basis[(size_t)m * (NH + 1) + k];
            }
            melw[off + i] = w;
        }
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

// ---- round 4: persistent STFT, tables resident in LDS ------------------------------------------------------------------------
// k_stft_mel16 launches one 64-thread workgroup per frame: 40 192 waves for a batch of 64 clips, each of which re-fetches its
// 32 window values, 47 twiddles and ~50 filterbank weights from global memory (shader-clock stamps: 13.5 k of the 21 k cycles
// of a frame are those table walks at 1.5 waves per SIMD), and - worse - fills every CU's LDS for 350 us, so that a recurrence
// kernel of the train step launched beside it waits for workgroup slots (k_gru4_bwd 43 -> 368 us).
// k_stft_mel_p is a PERSISTENT kernel on a fixed number of workgroups (one per CU by its LDS footprint; the launcher caps the
// grid, so the rest of the chip stays free for whatever else is running): 8 waves per workgroup, every wave walks its own
// sequence of frames with no workgroup barrier after the table fill.
//   * window, stage-A twiddles W_1024^(t k1), stage-B twiddles W_64^(m2 j1), unpack twiddles and the band-compressed
//     filterbank live in LDS in the order the lanes read them (conflict-free ds_read_b64 / b128);
//   * the next frame's 32 samples per lane are requested before the current frame's arithmetic starts;
//   * the three exchanges move the real and the imaginary halves one after the other through ONE 8.4 KB buffer per wave
//     (8 waves x 17 KB of complex buffers would not fit beside the tables); the third exchange only fetches the mirror
//     half Z[1024 - k]: after stage B2 lane t already holds Z[t + 64 n], so every lane unpacks the PAIR (k, 1024 - k) from one
//     (e, W^k o) product - half the unpack arithmetic of k_stft_mel16;
//   * same fp64 arithmetic, same operation order per output as k_stft_mel16 (parity with the oracle unchanged at 2e-6).
// Layout checks (8-byte elements; ds_read_b64: the 32 lanes of a half wave on distinct (a/8) mod 32, ds_write_b64: 16
// contiguous lanes on distinct (a/8) mod 16): exchange 1 rows of 66 (reader lanes (k1, m2): 2 k1 + m2), exchange 2 dense
// [m2][j1][k1] (reader lanes (k1, c): k1 + 16 c), exchange 3 [n - 8][t] read back at 64 - t.
#ifndef STP_WAVES
#define STP_WAVES 12
#endif
#define STP_THREADS (64 * STP_WAVES)
#define STP_XB 1056            // 8-byte elements per wave: 16 x 66 (exchange 1) >= 1025 magnitudes
#define STP_MELW 3584          // padded band-compressed filterbank weights held in LDS (64 Slaney bands over 1025 bins: 3 328)
#define STP_MELS 128           // bands whose support table fits

// The kernel is written once for two arithmetic types:
//   R = double  the reference's arithmetic (librosa runs the STFT in float64): parity with the oracle at 2e-6
//   R = float   SED_FFT_F32, a STATED mode like sed_dims.dtype: fp32 butterflies (twiddles and window still generated in
//               float64 and rounded once), twice the lanes per instruction through the packed-fp32 pipe, complex values as one
//               8-byte LDS element (no real / imaginary split).  Error bound asserted in tests/test_gpu_features.py.
template <typename R> struct cx { R x, y; };
template <typename R> __device__ __forceinline__ cx<R> cxmul(cx<R> a, cx<R> b) { return cx<R>{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
template <typename R> __device__ __forceinline__ void tdft4(cx<R>& a, cx<R>& b, cx<R>& c, cx<R>& d) {
    const cx<R> s02 = {a.x + c.x, a.y + c.y}, d02 = {a.x - c.x, a.y - c.y};
    const cx<R> s13 = {b.x + d.x, b.y + d.y}, d13 = {b.x - d.x, b.y - d.y};
    a = cx<R>{s02.x + s13.x, s02.y + s13.y};
    c = cx<R>{s02.x - s13.x, s02.y - s13.y};
    b = cx<R>{d02.x + d13.y, d02.y - d13.x};
    d = cx<R>{d02.x - d13.y, d02.y + d13.x};
}
// forward 16-point DFT in place; afterwards X[k] sits in v[DFT16_AT(k)] (same schedule as dft16 above)
template <typename R> __device__ __forceinline__ void tdft16(cx<R> (&v)[16]) {
    constexpr R C8 = (R)0.92387953251128673848, S8 = (R)0.38268343236508978178, R2 = (R)0.70710678118654752440;
#pragma unroll
    for (int n0 = 0; n0 < 4; ++n0) tdft4(v[n0], v[n0 + 4], v[n0 + 8], v[n0 + 12]);
    v[1 + 4] = cxmul(v[1 + 4], cx<R>{C8, -S8});
    v[1 + 8] = cxmul(v[1 + 8], cx<R>{R2, -R2});
    v[1 + 12] = cxmul(v[1 + 12], cx<R>{S8, -C8});
    v[2 + 4] = cxmul(v[2 + 4], cx<R>{R2, -R2});
    v[2 + 8] = cx<R>{v[2 + 8].y, -v[2 + 8].x};
    v[2 + 12] = cxmul(v[2 + 12], cx<R>{-R2, -R2});
    v[3 + 4] = cxmul(v[3 + 4], cx<R>{S8, -C8});
    v[3 + 8] = cxmul(v[3 + 8], cx<R>{-R2, -R2});
    v[3 + 12] = cxmul(v[3 + 12], cx<R>{-C8, S8});
#pragma unroll
    for (int k0 = 0; k0 < 4; ++k0) tdft4(v[4 * k0], v[4 * k0 + 1], v[4 * k0 + 2], v[4 * k0 + 3]);
}
// sqrt for the magnitudes.  fp64: v_rsq_f64 (~2^-23) and ONE Goldschmidt step -> ~2^-45 relative, five instructions instead of
// the ~17 of the correctly rounded sqrt(); the result is summed with fp32 weights and rounded to fp32.  fp32: v_sqrt_f32 (1 ulp).
__device__ __forceinline__ double stp_sqrt(double x) {
    x = fmax(x, 1e-280);
    const double y = __builtin_amdgcn_rsq(x);
    const double g = x * y, h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    return fma(g, r, g);
}
__device__ __forceinline__ float stp_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
// xor-1 / xor-2 butterflies inside a quad on the VALU (DPP quad_perm) - __shfl_xor goes through ds_bpermute, one LDS round trip each
__device__ __forceinline__ float quad_xor(float v, int which) {
    const int i = __float_as_int(v);
    return __int_as_float(which == 1 ? __builtin_amdgcn_mov_dpp(i, 0xB1, 0xF, 0xF, true) : __builtin_amdgcn_mov_dpp(i, 0x4E, 0xF, 0xF, true));
}
__device__ __forceinline__ double quad_xor(double v, int which) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    if (which == 1) { lo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true); hi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true); }
    else { lo = __builtin_amdgcn_mov_dpp(lo, 0x4E, 0xF, 0xF, true); hi = __builtin_amdgcn_mov_dpp(hi, 0x4E, 0xF, 0xF, true); }
    return __hiloint2double(hi, lo);
}

template <typename R> struct StpLds {
    R win[NFFT];
    cx<R> twA[15 * 64];
    cx<R> twU[8 * 64];
    cx<R> twB[64];
    float melw[STP_MELW];
    int lo[STP_MELS], off[STP_MELS / 16], trips[STP_MELS / 16];
    double xb[STP_WAVES][STP_XB];                       // 8-byte elements: doubles (fp64: split exchanges) or complex floats
};
__device__ __forceinline__ void wave_lds_sync() {
    // LDS operations of one wave execute in issue order; this only stops the compiler from moving them across the exchange
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
struct __attribute__((packed, aligned(4))) f32pair { float a, b; };

__device__ __forceinline__ void stp_load_frame(const float* __restrict__ w, int n_samples, int base, int t, float (&s)[32]) {
    if (base >= 0 && base + NFFT <= n_samples) {          // wave-uniform: no reflection in this frame
        const float* p = w + base + 2 * t;
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) {
            const f32pair v = *(const f32pair*)(p + 128 * n1);
            s[2 * n1] = v.a; s[2 * n1 + 1] = v.b;
        }
    } else {
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                int idx = base + 128 * n1 + 2 * t + h;
                if (idx < 0) idx = -idx;
                if (idx >= n_samples) idx = 2 * (n_samples - 1) - idx;
                s[2 * n1 + h] = w[idx];
            }
        }
    }
}

// One 16-value exchange through the wave's buffer: lane writes src[i] at wi(i), reads dst[i] from ri(i).
template <typename R, typename FW, typename FR>
__device__ __forceinline__ void stp_exchange16(double* xb, const cx<R> (&src)[16], cx<R> (&dst)[16], FW wi, FR ri) {
    if constexpr (sizeof(R) == 8) {
#pragma unroll
        for (int i = 0; i < 16; ++i) xb[wi(i)] = src[i].x;
        wave_lds_sync();
#pragma unroll
        for (int i = 0; i < 16; ++i) dst[i].x = xb[ri(i)];
        wave_lds_sync();
#pragma unroll
        for (int i = 0; i < 16; ++i) xb[wi(i)] = src[i].y;
        wave_lds_sync();
#pragma unroll
        for (int i = 0; i < 16; ++i) dst[i].y = xb[ri(i)];
        wave_lds_sync();
    } else {
        cx<R>* xc = reinterpret_cast<cx<R>*>(xb);
#pragma unroll
        for (int i = 0; i < 16; ++i) xc[wi(i)] = src[i];
        wave_lds_sync();
#pragma unroll
        for (int i = 0; i < 16; ++i) dst[i] = xc[ri(i)];
        wave_lds_sync();
    }
}

template <typename R>
__global__ __launch_bounds__(STP_THREADS) void k_stft_mel_p(const float* __restrict__ wave, int n_clips, int n_samples, int hop,
                                                             int frames, const double2* __restrict__ tw,
                                                             const double* __restrict__ win, const float* __restrict__ mel_basis,
                                                             const int* __restrict__ band, const float* __restrict__ melw_g,
                                                             int n_mels, float* __restrict__ mel) {
    extern __shared__ __attribute__((aligned(16))) unsigned char stp_raw[];
    StpLds<R>& L = *reinterpret_cast<StpLds<R>*>(stp_raw);
    typedef cx<R> C;
    TS(10);
    const int tid = threadIdx.x, t = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);    // (wave-uniform: scalar registers)
    const int n_total = n_clips * frames, n_wv = gridDim.x * STP_WAVES;
    // (clip, frame) of this wave's current frame, advanced without a division per frame (all wave-uniform)
    int g = blockIdx.x * STP_WAVES + wv;
    int clip = g / frames, f = g - clip * frames;
    const int dq = n_wv / frames, dr = n_wv - dq * frames;
    float s[32];
    if (g < n_total) stp_load_frame(wave + (size_t)clip * n_samples, n_samples, f * hop - NFFT / 2, t, s);   // in flight while the tables are filled
    // ---- tables -> LDS (once per workgroup) ------------------------------------------------------------------------------------
    const int nnz = band[3 * n_mels];                                             // padded size; 0: no LDS table
    const bool w_in_lds = n_mels <= 64 && nnz > 0 && nnz <= STP_MELW;
    for (int i = tid; i < NFFT; i += STP_THREADS) L.win[i] = (R)win[i];
    for (int i = tid; i < 15 * 64; i += STP_THREADS) {
        const int k1 = 1 + (i >> 6), tt = i & 63;
        const double2 v = tw[(2 * tt * k1) & (NFFT - 1)];                         // W_1024^(t k1) = W_2048^(2 t k1)
        L.twA[i] = C{(R)v.x, (R)v.y};
    }
    for (int i = tid; i < 8 * 64; i += STP_THREADS) { const double2 v = tw[i]; L.twU[i] = C{(R)v.x, (R)v.y}; }     // W_2048^k, k = t + 64 n
    if (tid < 64) { const double2 v = tw[(32 * (tid >> 4) * (tid & 15)) & (NFFT - 1)]; L.twB[tid] = C{(R)v.x, (R)v.y}; }   // W_64^(m2 j1)
    if (w_in_lds) {
        for (int i = tid; i < n_mels; i += STP_THREADS) L.lo[i] = band[2 * i];
        if (tid < (n_mels + 15) / 16) { L.trips[tid] = band[3 * n_mels + 1 + tid]; L.off[tid] = band[2 * n_mels + tid]; }
        for (int i = tid; i < nnz; i += STP_THREADS) L.melw[i] = melw_g[i];
    }
    __syncthreads();
    double* xb = L.xb[wv];
    const int k1l = t & 15, m2 = t >> 4, q = t & 3;
    // mel pass parameters are frame-invariant: trips and offsets in scalar registers; the support start of this lane's band is
    // re-read from LDS per pass (four more live registers across the FFT cost the fp64 kernel its third wave per SIMD)
    int mtr[4], moff[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        mtr[p] = __builtin_amdgcn_readfirstlane((w_in_lds && 16 * p < n_mels) ? L.trips[p] : 0);
        moff[p] = __builtin_amdgcn_readfirstlane((w_in_lds && 16 * p < n_mels) ? L.off[p] : 0);
    }
    TSC(0);
    TS(8);                                                                    // (wall clock beside the shader clock: tools/ts_feat.py derives the clock the kernel runs at)
    while (g < n_total) {
        C v[16];
        // ---- windowed frame: z[m] = x[2m] + i x[2m + 1], m = 64 n1 + t ---------------------------------------------------------------
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) {
            const C wn = *reinterpret_cast<const C*>(&L.win[128 * n1 + 2 * t]);
            v[n1] = C{(R)s[2 * n1] * wn.x, (R)s[2 * n1 + 1] * wn.y};
        }
        const int row_cur = clip * frames + f;
        g += n_wv; clip += dq; f += dr;
        if (f >= frames) { f -= frames; ++clip; }
        TSC(1);
        // ---- A: 16-point DFT over n1, twiddle, exchange 1 ([k1][t], rows of 66) ----------------------------------------------------------
        tdft16(v);
        {
            C a[16];
            a[0] = v[DFT16_AT(0)];
#pragma unroll
            for (int k1 = 1; k1 < 16; ++k1) a[k1] = cxmul(v[DFT16_AT(k1)], L.twA[(k1 - 1) * 64 + t]);
            stp_exchange16<R>(xb, a, v, [&](int k1) { return k1 * 66 + t; }, [&](int m1) { return k1l * 66 + 4 * m1 + m2; });
        }
        TSC(2);
        // ---- B1: thread (k1, m2): 16-point DFT over m1, twiddle W_64^(m2 j1), exchange 2 ([m2][j1][k1]) ---------------------------------
        tdft16(v);
        C x4[16];                                                               // x4[4 i + q'] = B1 output (k1, j1 = m2 + 4 i, m2' = q')
        {
            C b[16];
            b[0] = v[DFT16_AT(0)];
#pragma unroll
            for (int j1 = 1; j1 < 16; ++j1) b[j1] = cxmul(v[DFT16_AT(j1)], L.twB[m2 * 16 + j1]);
            stp_exchange16<R>(xb, b, x4, [&](int j1) { return 256 * m2 + 16 * j1 + k1l; },
                              [&](int e) { return 256 * (e & 3) + 16 * (m2 + 4 * (e >> 2)) + k1l; });
        }
        // ---- B2: 4-point DFT over m2: x4[4 i + j2] = Z[t + 64 (i + 4 j2)] -----------------------------------------------------------------
#pragma unroll
        for (int i = 0; i < 4; ++i) tdft4(x4[4 * i], x4[4 * i + 1], x4[4 * i + 2], x4[4 * i + 3]);
        TSC(3);
        // ---- exchange 3: the mirror half.  Z[1024 - (t + 64 n)], n = 0..7, sits in lane (64 - t) & 63 as its n' = 15 - n (16 - n for t = 0)
#define STP_Z(n) x4[4 * ((n) & 3) + ((n) >> 2)]
        C zm[8];
        if constexpr (sizeof(R) == 8) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
#pragma unroll
                for (int n = 8; n < 16; ++n) xb[(n - 8) * 64 + t] = c ? STP_Z(n).y : STP_Z(n).x;
                if (t == 0) xb[512] = c ? STP_Z(0).y : STP_Z(0).x;                // Z[1024] = Z[0]
                wave_lds_sync();
#pragma unroll
                for (int n = 0; n < 8; ++n) {
                    const double r = xb[(7 - n) * 64 + 64 - t];
                    if (c) zm[n].y = r; else zm[n].x = r;
                }
                wave_lds_sync();
            }
        } else {
            C* xc = reinterpret_cast<C*>(xb);
#pragma unroll
            for (int n = 8; n < 16; ++n) xc[(n - 8) * 64 + t] = STP_Z(n);
            if (t == 0) xc[512] = STP_Z(0);
            wave_lds_sync();
#pragma unroll
            for (int n = 0; n < 8; ++n) zm[n] = xc[(7 - n) * 64 + 64 - t];
            wave_lds_sync();
        }
        TSC(4);
        // ---- real-FFT unpack of the pair (k, 1024 - k), k = t + 64 n: X[k] = e - i W^k o, X[1024 - k] = conj(e) - i conj(W^k o) -------------
        // (the 0.5 of e and o is left out: every magnitude comes out doubled, the band sums are halved at the end - exact)
        // (the mirror values are in registers: the magnitudes can overwrite the exchange buffer pair by pair)
        R* mag = reinterpret_cast<R*>(xb);
        R* mag_m = mag + (576 - t);
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            const C Zk = STP_Z(n);
            const C Zc = {zm[n].x, -zm[n].y};
            const C e = {Zk.x + Zc.x, Zk.y + Zc.y};
            const C o = {Zk.x - Zc.x, Zk.y - Zc.y};
            const C tt = cxmul(L.twU[n * 64 + t], o);
            const R re = e.x + tt.y, im = e.y - tt.x;
            const R rm = e.x - tt.y, imm = e.y + tt.x;                          // (|.| of the conjugate partner: sign of im irrelevant)
            mag[n * 64 + t] = stp_sqrt(re * re + im * im);
            mag_m[(7 - n) * 64] = stp_sqrt(rm * rm + imm * imm);                 // mag[1024 - 64 n - t] (one base, offsets >= 0)
        }
        const C Zh = STP_Z(8);                                                  // lane 0: Z[512]; |X[512]| = |Z[512]|
        if (t == 0) mag[512] = (R)2 * stp_sqrt(Zh.x * Zh.x + Zh.y * Zh.y);
        // the zero-weight padding trips of the packed projection read up to mag[1055] (k_feat_pack's guard): whatever the exchanges
        // left there - or, for the two fp64 elements no exchange ever writes, whatever the LDS held before the launch - must not be
        // an Inf / NaN pattern (0 x NaN would poison a band sum)
        if (t < STP_XB - (NH + 1)) mag[NH + 1 + t] = (R)0;
#undef STP_Z
        wave_lds_sync();
        // the next frame's samples are requested here: the FFT's registers are free from this point on (held across the
        // exchanges, the 32 values cost a third wave per SIMD), and the projection below covers most of their latency
        if (g < n_total) stp_load_frame(wave + (size_t)clip * n_samples, n_samples, f * hop - NFFT / 2, t, s);
        TSC(5);
        // ---- mel projection: lane = (band 16 p + t / 4, quarter q) in pass p; bins lo + q + 4 j, zero-padded weights [j][lane] ---------------
        float* outp = mel + (size_t)row_cur * n_mels;
        if (w_in_lds) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                R acc = (R)0;
                const int m_p = 16 * p + (t >> 2);
                const R* mp = mag + (m_p < n_mels ? L.lo[m_p] + q : 0);
                const float* wp = L.melw + moff[p] + t;
                for (int j0 = 0; j0 < mtr[p]; j0 += 4) {                        // (trips: multiples of 4, wave-uniform)
                    float wvv[4]; R mv[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) { wvv[i] = wp[64 * i]; mv[i] = mp[4 * i]; }
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc += (R)wvv[i] * mv[i];
                    mp += 16; wp += 256;
                }
                acc += quad_xor(acc, 1);
                acc += quad_xor(acc, 2);
                if (q == 0 && m_p < n_mels) outp[m_p] = (float)((R)0.5 * acc);
            }
        } else {                                                                // general basis: walk the dense rows (supports from `band`)
            for (int m0 = 0; m0 < n_mels; m0 += 16) {
                const int m = m0 + (t >> 2);
                R sacc = (R)0;
                if (m < n_mels) {
                    const float* row = mel_basis + (size_t)m * (NH + 1);
                    const int lo = band[2 * m], hi = band[2 * m + 1];
                    for (int k = lo + q; k < hi; k += 4) sacc += (R)row[k] * mag[k];
                }
                sacc += quad_xor(sacc, 1);
                sacc += quad_xor(sacc, 2);
                if (q == 0 && m < n_mels) outp[m] = (float)((R)0.5 * sacc);
            }
        }
        wave_lds_sync();                                                        // the magnitudes are read before the next frame's exchange 1 overwrites them
        TSC(6);
    }
    TSC(7);
    TS(9);
}

// ---- log / noise / pad / normalise ----------------------------------------------------------------
// |N(0, 0.25)| for the element PAIR (2 i, 2 i + 1) of the flattened [clip][frame][mel] tensor: one Philox4x32-10 draw (counter i,
// stream 16), one Box-Muller transform - both of its outputs are used (r cos, r sin: two independent normals), so a pair
// costs one Philox, one log, one sqrt and one sincospi in fp64 (round 3 drew and transformed once per ELEMENT and used half of
// each draw).  Mirrored bit for bit by oracle/philox.py teacher_noise.
__device__ __forceinline__ void teacher_noise_pair(uint32_t pair, uint64_t seed, double& n0, double& n1) {
    const u32x4 o = philox4x32_10(pair, 0u, 16u, PHILOX_TAG, (uint32_t)seed, (uint32_t)(seed >> 32));
    const double u1 = ((double)o.x + 1.0) * 2.3283064365386963e-10, u2 = (double)o.y * 2.3283064365386963e-10;
    const double r = sqrt(-2.0 * log(u1));
    double sn, cs;
    sincospi(2.0 * u2, &sn, &cs);
    n0 = (double)(float)fabs(0.25 * (r * cs));
    n1 = (double)(float)fabs(0.25 * (r * sn));
}
// SED_FFT_F32 (the front-end's stated fp32 mode): the same draw and transform in fp32 arithmetic - the noise values then differ
// from the float64-derived ones by an ulp or two of fp32 (relative ~1e-7 of a value that is added to a linear mel amplitude)
__device__ __forceinline__ void teacher_noise_pair_f32(uint32_t pair, uint64_t seed, float& n0, float& n1) {
    const u32x4 o = philox4x32_10(pair, 0u, 16u, PHILOX_TAG, (uint32_t)seed, (uint32_t)(seed >> 32));
    const float u1 = fminf(((float)o.x + 1.0f) * 2.3283064365386963e-10f, 1.0f), u2 = (float)o.y * 2.3283064365386963e-10f;
    const float r = sqrtf(-2.0f * logf(u1));
    float sn, cs;
    sincospif(2.0f * u2, &sn, &cs);
    n0 = fabsf(0.25f * (r * cs));
    n1 = fabsf(0.25f * (r * sn));
}
__device__ __forceinline__ float amp_db_f32(float a) { return 10.0f * log10f(fmaxf(1e-10f, a * a)); }
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
template <int F32>
__global__ __launch_bounds__(256) void k_logmel_max(const float* __restrict__ mel, int frames, int n_mels, int max_frames,
                                                     const uint64_t* __restrict__ seed_ptr, double* __restrict__ part,
                                                     float* __restrict__ out_noisy) {
    __shared__ double red[2][4];
    const int tid = threadIdx.x, chunk = blockIdx.x, clip = blockIdx.y, lane = tid & 63, wv = tid >> 6;
    const int n = frames * n_mels, n_keep = min(frames, max_frames) * n_mels;
    const float* src = mel + (size_t)clip * n;
    const uint64_t seed = out_noisy ? seed_ptr[0] : 0ull;
    const int per = (((n + LM_CHUNKS - 1) / LM_CHUNKS) + 1) & ~1, e0 = chunk * per, e1 = min(n, e0 + per);   // even chunks: pairs never straddle
    double mc = 0.0, mn = 0.0;
    for (int e = e0 + 2 * tid; e < e1; e += 512) {       // (n = frames * n_mels; an odd n leaves a last single element)
        const bool two = e + 1 < e1;
        const double a0 = (double)src[e], a1 = two ? (double)src[e + 1] : 0.0;
        mc = fmax(mc, fmax(fabs(a0), fabs(a1)));
        if (out_noisy) {
            double z0, z1;
            auto draw = [&](uint32_t pair, double& a_, double& b_) {
                if constexpr (F32 != 0) { float f0, f1; teacher_noise_pair_f32(pair, seed, f0, f1); a_ = f0; b_ = f1; }
                else teacher_noise_pair(pair, seed, a_, b_);
            };
            draw((uint32_t)(((size_t)clip * n + e) >> 1), z0, z1);
            if ((((size_t)clip * n + e) & 1) != 0) {      // odd clip size x odd clip index: the pair starts one element earlier
                // (only reachable when n is odd; keep the definition element-wise exact)
                double y0, y1;
                draw((uint32_t)((((size_t)clip * n + e) >> 1) + 1), y0, y1);
                z0 = z1; z1 = y0;
            }
            mn = fmax(mn, fabs(a0 + z0));
            if (e < n_keep) out_noisy[(size_t)clip * max_frames * n_mels + e] = (float)z0;       // exact: the noise is a float
            if (two) {
                mn = fmax(mn, fabs(a1 + z1));
                if (e + 1 < n_keep) out_noisy[(size_t)clip * max_frames * n_mels + e + 1] = (float)z1;
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mc = fmax(mc, __shfl_xor(mc, o)); mn = fmax(mn, __shfl_xor(mn, o)); }
    if (lane == 0) { red[0][wv] = mc; red[1][wv] = mn; }
    __syncthreads();
    if (tid < 2) part[((size_t)clip * LM_CHUNKS + chunk) * 2 + tid] = fmax(fmax(red[tid][0], red[tid][1]), fmax(red[tid][2], red[tid][3]));
}

template <int F32>
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
            if constexpr (F32 != 0) {
                const float a = src[e];
                vc = fmaxf(amp_db_f32(a), (float)floor_c);
                if (out_noisy) vn = fmaxf(amp_db_f32(a + out_noisy[(size_t)clip * n_out + e]), (float)floor_n);
            } else {
                const double a = (double)src[e];
                vc = (float)fmax(amp_db(a), floor_c);           // ToTensor: .float()
                if (out_noisy) vn = (float)fmax(amp_db(a + (double)out_noisy[(size_t)clip * n_out + e]), floor_n);
            }
        }
        if (mean) {                                             // Scaler.normalize in float64, torch.Tensor() -> fp32
            if constexpr (F32 != 0) {
                const float mu = (float)mean[m], rs = 1.0f / (float)stdv[m];
                vc = (vc - mu) * rs;
                vn = (vn - mu) * rs;
            } else {
                vc = (float)(((double)vc - mean[m]) / stdv[m]);
                vn = (float)(((double)vn - mean[m]) / stdv[m]);
            }
        }
        out_clean[(size_t)clip * n_out + e] = vc;
        if (out_noisy) out_noisy[(size_t)clip * n_out + e] = vn;
    }
}

// workspace: W_2048 table | window | band supports (lo, hi) + offsets + total | compressed filterbank weights
static size_t mel_ws_band_off() { return (size_t)NFFT * sizeof(double2) + (size_t)NFFT * sizeof(double); }
static size_t mel_ws_w_off(int n_mels) { return mel_ws_band_off() + (((size_t)(3 * n_mels + 2 + (n_mels + 15) / 16) * sizeof(int) + 15) & ~(size_t)15); }
extern "C" size_t sed_mel_spec_ws_bytes(int n_clips, int n_samples, int hop, int n_fft, int n_mels) {
    (void)n_clips; (void)n_samples; (void)hop;
    if (n_fft != NFFT) return 0;
    return mel_ws_w_off(n_mels > 0 ? n_mels : 0) + (size_t)STP_MELW * sizeof(float);
}

static int mel_check(int n_fft, int n_mels, const void* mel_basis, const void* ws, size_t ws_bytes, const char* who) {
    if (n_fft != NFFT) {
        sed_set_error("%s: n_fft must be 2048 (config.py:18), got %d", who, n_fft);
        return SED_ERR_UNSUPPORTED;
    }
    if (!mel_basis || !ws || n_mels < 1) {
        sed_set_error("%s: null argument / n_mels < 1", who);
        return SED_ERR_BAD_ARG;
    }
    if (ws_bytes < sed_mel_spec_ws_bytes(1, 0, 1, n_fft, n_mels)) {
        sed_set_error("%s: workspace too small", who);
        return SED_ERR_WORKSPACE;
    }
    return SED_OK;
}

extern "C" int sed_mel_tables(int n_fft, const float* window, const float* mel_basis, int n_mels, void* ws, size_t ws_bytes,
                              void* stream) {
    SED_TRY(mel_check(n_fft, n_mels, mel_basis, ws, ws_bytes, "sed_mel_tables"));
    hipStream_t st = (hipStream_t)stream;
    double2* tw = (double2*)ws;
    double* win = (double*)((char*)ws + (size_t)NFFT * sizeof(double2));
    int* band = (int*)((char*)ws + mel_ws_band_off());
    float* melw = (float*)((char*)ws + mel_ws_w_off(n_mels));
    const int tb = (NFFT > n_mels * 64 ? NFFT : n_mels * 64);
    k_feat_tables<<<(tb + 255) / 256, 256, 0, st>>>(tw, win, window, mel_basis, n_mels, band);
    SED_CHECK_LAUNCH();
    k_feat_pack<<<1, 256, 0, st>>>(mel_basis, n_mels, band, melw, STP_MELW);
    SED_CHECK_LAUNCH();
    return SED_OK;
}

static int stp_num_cus() {
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!cus[dev]) {
        int n = 0;
        cus[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    }
    return cus[dev];
}

template <typename R>
static int launch_stft_p(const float* wave, int n_clips, int n_samples, int hop, int frames, const double2* tw, const double* win,
                         const float* mel_basis, const int* band, const float* melw, int n_mels, float* mel, int max_workgroups,
                         hipStream_t st) {
    static thread_local SedAttrOnce attr_done;
    if (attr_done.need())
        SED_CHECK_HIP(hipFuncSetAttribute((const void*)k_stft_mel_p<R>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(StpLds<R>)));
    const long long n_total = (long long)n_clips * frames;
    SED_CHECK_ARG(n_total < (1ll << 31), "sed_mel_frames: too many frames");
    long long grid = (n_total + STP_WAVES - 1) / STP_WAVES;
    const int cap = max_workgroups > 0 ? max_workgroups : stp_num_cus();        // one workgroup per CU (by LDS / registers); default: the whole chip
    if (grid > cap) grid = cap;
    k_stft_mel_p<R><<<(int)grid, STP_THREADS, sizeof(StpLds<R>), st>>>(wave, n_clips, n_samples, hop, frames, tw, win, mel_basis, band,
                                                                      melw, n_mels, mel);
    SED_CHECK_LAUNCH();
    return SED_OK;
}

extern "C" int sed_mel_frames(const float* wave, int n_clips, int n_samples, int hop, int n_fft, const float* mel_basis,
                              int n_mels, float* mel, const void* ws, size_t ws_bytes, int fft_dtype, int max_workgroups,
                              void* stream) {
    SED_TRY(mel_check(n_fft, n_mels, mel_basis, ws, ws_bytes, "sed_mel_frames"));
    SED_CHECK_ARG(wave && mel, "sed_mel_frames: null argument");
    SED_CHECK_ARG(n_clips >= 1 && hop >= 1 && n_samples > NFFT / 2, "sed_mel_frames: bad sizes (reflect padding needs n_samples > n_fft/2)");
    SED_CHECK_ARG(fft_dtype == SED_FFT_F64 || fft_dtype == SED_FFT_F32, "sed_mel_frames: fft_dtype must be SED_FFT_F64 or SED_FFT_F32");
    hipStream_t st = (hipStream_t)stream;
    const double2* tw = (const double2*)ws;
    const double* win = (const double*)((const char*)ws + (size_t)NFFT * sizeof(double2));
    const int* band = (const int*)((const char*)ws + mel_ws_band_off());
    const float* melw = (const float*)((const char*)ws + mel_ws_w_off(n_mels));
    const int frames = 1 + n_samples / hop;
    // debug bit 19: round 2's kernel (one 256-thread workgroup per frame, five radix-4 passes through LDS); bit 21: round 3's
    // (one wave per frame, tables walked in global memory) - both fp64, kept for A/B timing and as second implementations in the tests
    if (g_sed_debug & 524288) {
        k_stft_mel<<<dim3(frames, n_clips), 256, 0, st>>>(wave, n_samples, hop, frames, tw, win, mel_basis, band, n_mels, mel);
    } else if (g_sed_debug & 2097152) {
        k_stft_mel16<<<dim3(frames, n_clips), 64, 0, st>>>(wave, n_samples, hop, frames, tw, win, mel_basis, band, n_mels, mel);
    } else if (fft_dtype == SED_FFT_F32) {
        return launch_stft_p<float>(wave, n_clips, n_samples, hop, frames, tw, win, mel_basis, band, melw, n_mels, mel, max_workgroups, st);
    } else {
        return launch_stft_p<double>(wave, n_clips, n_samples, hop, frames, tw, win, mel_basis, band, melw, n_mels, mel, max_workgroups, st);
    }
    SED_CHECK_LAUNCH();
    return SED_OK;
}

extern "C" int sed_mel_spec(const float* wave, int n_clips, int n_samples, int hop, int n_fft, const float* window,
                            const float* mel_basis, int n_mels, float* mel, void* ws, size_t ws_bytes, void* stream) {
    SED_TRY(sed_mel_tables(n_fft, window, mel_basis, n_mels, ws, ws_bytes, stream));
    return sed_mel_frames(wave, n_clips, n_samples, hop, n_fft, mel_basis, n_mels, mel, ws, ws_bytes, SED_FFT_F64, 0, stream);
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
                                    float* out_noisy, void* ws, size_t ws_bytes, int math_dtype, void* stream) {
    SED_CHECK_ARG(math_dtype == SED_FFT_F64 || math_dtype == SED_FFT_F32, "sed_logmel_transform: math_dtype must be SED_FFT_F64 or SED_FFT_F32");
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
    if (math_dtype == SED_FFT_F32) k_logmel_max<1><<<dim3(LM_CHUNKS, n_clips), 256, 0, st>>>(mel, frames, n_mels, max_frames, seed_dev, (double*)ws, out_noisy);
    else k_logmel_max<0><<<dim3(LM_CHUNKS, n_clips), 256, 0, st>>>(mel, frames, n_mels, max_frames, seed_dev, (double*)ws, out_noisy);
    SED_CHECK_LAUNCH();
    const int n_out = max_frames * n_mels;
    const dim3 ga((n_out + 1023) / 1024, n_clips);
    if (math_dtype == SED_FFT_F32) k_logmel_apply<1><<<ga, 256, 0, st>>>(mel, frames, n_mels, max_frames, mean, std, (const double*)ws, out_clean, out_noisy);
    else k_logmel_apply<0><<<ga, 256, 0, st>>>(mel, frames, n_mels, max_frames, mean, std, (const double*)ws, out_clean, out_noisy);
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
