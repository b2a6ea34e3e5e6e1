// post.hip - strong-posterior post-processing of the reference's evaluation loop, on the device.
//
// Reference ops (baseline/evaluation_measures.py:203-231, utils/utils.py:146-162), per clip, host-side numpy:
//   pred = ProbabilityEncoder().binarization(pred_strong, "global_threshold", threshold=0.5)   (dcase_util: p > thr)
//   pred = scipy.ndimage.filters.median_filter(pred, (median_window, 1))                        (mode "reflect")
//   for each class column: DecisionEncoder().find_contiguous_regions(column) -> [onset, offset) frame pairs
// The reference does this one clip at a time after a batch-1 forward; here one wave owns one (clip, class) column of a
// whole batch: threshold, rank filter (for 0/1 data the median is a count), run-length decode with wave ballots.
// Output is the compact event list the host turns into the reference's DataFrame / TSV.
#include "common.h"
#include "kernels.h"

#define PP_MAXT 2048      // frames per column held in LDS (T/8: clips up to 16 384 input frames)

__global__ __launch_bounds__(64) void k_postprocess(const float* __restrict__ strong, int T, int NC, float threshold,
                                                    int window, uint8_t* __restrict__ binary, int32_t* __restrict__ ev_count,
                                                    int32_t* __restrict__ ev_pairs, int max_ev) {
    __shared__ uint8_t raw[PP_MAXT];
    __shared__ uint8_t flt[PP_MAXT + 1];
    const int b = blockIdx.x / NC, c = blockIdx.x % NC, lane = threadIdx.x;
    const float* p = strong + (size_t)b * T * NC + c;
    for (int t = lane; t < T; t += 64) raw[t] = (p[(size_t)t * NC] > threshold) ? 1 : 0;
    __syncthreads();
    // scipy rank filter: origin 0 -> window covers [t - w/2, t - w/2 + w); reflect: -1 -> 0, -2 -> 1, T -> T-1, T+1 -> T-2;
    // median = sorted[w/2]: for 0/1 data that is 1 iff #ones >= w - w/2
    const int lo = window / 2, need = window - window / 2;
    for (int t = lane; t < T; t += 64) {
        int ones = 0;
        for (int d = 0; d < window; ++d) {
            int i = t - lo + d;
            // reflect (repeatedly, for windows longer than the column)
            while (i < 0 || i >= T) i = (i < 0) ? -i - 1 : 2 * T - i - 1;
            ones += raw[i];
        }
        const uint8_t v = ones >= need ? 1 : 0;
        flt[t] = v;
        if (binary) binary[((size_t)b * T + t) * NC + c] = v;
    }
    if (lane == 0) flt[T] = 0;
    __syncthreads();
    // find_contiguous_regions: onset where 0 -> 1 (or t = 0 active), offset (exclusive) where 1 -> 0 (or the end).
    // Onsets and offsets alternate along the column, so the k-th onset pairs with the k-th offset; two running counters
    // because an event may span a 64-frame chunk boundary.
    int n_on = 0, n_off = 0;
    int32_t* out = ev_pairs + (size_t)blockIdx.x * max_ev * 2;
    for (int t0 = 0; t0 < T; t0 += 64) {
        const int t = t0 + lane;
        const bool act = t < T && flt[t];
        const bool prev = act && t > 0 && flt[t - 1];
        const bool next = act && flt[t + 1];              // flt[T] = 0
        const bool is_on = act && !prev, is_off = act && !next;
        const unsigned long long m_on = __ballot(is_on), m_off = __ballot(is_off);
        const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
        if (is_on) {
            const int k = n_on + __popcll(m_on & below);
            if (k < max_ev) out[2 * k] = t;
        }
        if (is_off) {
            const int k = n_off + __popcll(m_off & below);
            if (k < max_ev) out[2 * k + 1] = t + 1;
        }
        n_on += __popcll(m_on);
        n_off += __popcll(m_off);
    }
    const int n_ev = n_on;
    if (lane == 0) ev_count[blockIdx.x] = n_ev;
}

extern "C" int sed_postprocess(const float* strong, int n_clips, int T, int nclass, float threshold, int median_window,
                               uint8_t* binary, int32_t* ev_count, int32_t* ev_pairs, int max_events, void* stream) {
    SED_CHECK_ARG(strong && ev_count && ev_pairs, "sed_postprocess: null argument");
    SED_CHECK_ARG(n_clips >= 1 && nclass >= 1 && T >= 1 && T <= PP_MAXT, "sed_postprocess: need 1 <= T <= 2048 output frames");
    SED_CHECK_ARG(median_window >= 1 && median_window <= 63, "sed_postprocess: median_window must be in [1, 63]");
    SED_CHECK_ARG(max_events >= (T + 1) / 2, "sed_postprocess: max_events must be >= ceil(T / 2)");
    k_postprocess<<<n_clips * nclass, 64, 0, (hipStream_t)stream>>>(strong, T, nclass, threshold, median_window, binary, ev_count,
                                                                      ev_pairs, max_events);
    SED_CHECK_LAUNCH();
    return SED_OK;
}
