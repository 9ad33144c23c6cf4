// Spectrum rendering: LightSpectrum.render (optrace/tracer/spectrum/light_spectrum.py:41-79) on the device.
// The reference histograms the float32 wavelengths of the selected rays with np.histogram(wl, bins=N, weights=w,
// range=[wl0, wl1]).  With float32 data and float32 range NumPy keeps the whole bin search in float32
// (bin_type = result_type(first, last, a)); the kernels below follow that arithmetic so that a ray lands in the
// same bin.  Rays are given dense (weight 0 = not selected), as ot_detector_hits leaves them, or as a compact list
// (`fill`: 1024 pieces of hit_piece_len(n) entries, piece k holding fill[k] entries at its front, ot_detector_req.fill).
#pragma once
#include "ot_detector.hpp"
#include "ot_device.hpp"

#define OT_SPEC_SLICE 8192  // entries of a compact list a workgroup takes at a time

// pass 1: wavelength range and number of selected rays.  stats = {min wl, max wl} (pre-set to +inf / -inf),
// count[0] += rays with w > 0 (np.count_nonzero(w) over the selected rays, light_spectrum.py:60,70)
// index walk shared by both passes: dense = grid-stride over [0, n); compact = the workgroup takes whole pieces
template <class F>
OT_DEV void spectrum_for_each(int64_t n, const unsigned int* __restrict__ fill, F&& body) {
    if (!fill) {
        const int64_t stride = (int64_t)gridDim.x * blockDim.x;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) body(i);
        return;
    }
    // work items = (piece, slice of OT_SPEC_SLICE entries): a workgroup per whole piece left the 1024 pieces of a long list
    // (2e8 rays: 7e4 filled entries each) to 1024 workgroups walking them serially -- 0.54 / 0.84 ms for the two passes
    // over 0.56 GB; slices beyond a piece's fill cost one comparison
    const int shift = hit_piece_shift(n);
    const int64_t slices = (((int64_t)1 << shift) + OT_SPEC_SLICE - 1) / OT_SPEC_SLICE;  // per piece
    for (int64_t it = blockIdx.x; it < OT_HIT_PIECES_N * slices; it += gridDim.x) {
        const int64_t sl = it / OT_HIT_PIECES_N, pc = it % OT_HIT_PIECES_N;  // slice-major: the filled front slices spread over all workgroups
        const int64_t i0 = (pc << shift) + sl * OT_SPEC_SLICE;
        int64_t i1 = (pc << shift) + (int64_t)fill[pc];
        if (i1 > i0 + OT_SPEC_SLICE) i1 = i0 + OT_SPEC_SLICE;
        for (int64_t i = i0 + threadIdx.x; i < i1; i += blockDim.x) body(i);
    }
}

__global__ __launch_bounds__(256) void spectrum_stats_kernel(int64_t n, const float* __restrict__ wl, const float* __restrict__ w,
                                                             double* __restrict__ stats, unsigned long long* __restrict__ count,
                                                             const unsigned int* __restrict__ fill) {
    const double inf = __builtin_inf();
    double lo = inf, hi = -inf;
    unsigned long long c = 0;
    spectrum_for_each(n, fill, [&](int64_t i) {
        if (w[i] > 0.f) {
            double l = (double)wl[i];
            lo = fmin(lo, l);
            hi = fmax(hi, l);
            c++;
        }
    });
    lo = wave_min(lo);
    hi = wave_max(hi);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if (__lane_id() == 0 && c) {
        atomic_min_f64(&stats[0], lo);
        atomic_max_f64(&stats[1], hi);
        atomicAdd(count, c);
    }
}

// pass 2: weighted histogram over the float32 edges `edges` (nbins + 1 values, np.linspace of the reference).
// Bin search of numpy/lib/_histograms_impl.py (uniform-bin fast path): scaled index in float32, then one-step
// corrections against the edges; the last bin is closed on the right.  Sums are kept in float64.
// LDS-privatised when the bins fit (lds_bins > 0), otherwise global atomics.
__global__ __launch_bounds__(1024) void spectrum_hist_kernel(int64_t n, const float* __restrict__ wl, const float* __restrict__ w,
                                                             const float* __restrict__ edges, int nbins, int lds_bins,
                                                             double* __restrict__ hist, const unsigned int* __restrict__ fill) {
    extern __shared__ double sh[];  // [lds_bins] sums, then [lds_bins + 1] edges as float
    float* sedge = (float*)(sh + lds_bins);
    if (lds_bins) {
        for (int i = threadIdx.x; i < lds_bins; i += blockDim.x) sh[i] = 0.0;
        for (int i = threadIdx.x; i <= lds_bins; i += blockDim.x) sedge[i] = edges[i];
        __syncthreads();
    }
    const float* e = lds_bins ? sedge : edges;
    const float first = edges[0], last = edges[nbins];
    const float denom = last - first;
    const float fn = (float)nbins;
    spectrum_for_each(n, fill, [&](int64_t i) {
        float wi = w[i];
        if (!(wi > 0.f)) return;
        float x = wl[i];
        if (!(x >= first && x <= last)) return;
        int idx = (int)(((x - first) / denom) * fn);
        if (idx == nbins) idx -= 1;
        if (x < e[idx]) idx -= 1;
        if (x >= e[idx + 1] && idx != nbins - 1) idx += 1;
        if (lds_bins)
            unsafeAtomicAdd(&sh[idx], (double)wi);
        else
            unsafeAtomicAdd(&hist[idx], (double)wi);
    });
    if (lds_bins) {
        __syncthreads();
        for (int i = threadIdx.x; i < lds_bins; i += blockDim.x) {
            double v = sh[i];
            if (v != 0.0) unsafeAtomicAdd(&hist[i], v);
        }
    }
}
