// Focus search on the device: Raytracer.focus_search (optrace/tracer/raytracer.py:1354-1640).
//
// The reference extracts, per ray, the section that crosses the search region, writes it as the line
// ph(z) = pa + sb * z and then evaluates a cost function at hundreds of z positions with NumPy passes over all
// rays.  Here the line parameters stay in HBM (pax, pay, sbx, sby f64, w f32; 36 B per ray) and one cost
// evaluation is a short chain of streaming passes; nothing goes back to the host between the passes, a whole
// batch of z samples leaves one cost value each in a device array.
//
// Rays are dense: w < 0 marks a ray that is not part of the search (already absorbed in front of the region).
// Rays with w == 0 stay in: the reference counts them for the pixel number and the image extent.
#pragma once
#include "ot_detector.hpp"
#include "ot_device.hpp"

// workspace slots (doubles)
#define OT_FS_XMIN 0
#define OT_FS_XMAX 1
#define OT_FS_YMIN 2
#define OT_FS_YMAX 3
#define OT_FS_W 4     // sum w
#define OT_FS_W2 5    // sum w^2
#define OT_FS_WX 6    // sum w x
#define OT_FS_WY 7    // sum w y
#define OT_FS_VX 8    // sum w (x - avg)^2
#define OT_FS_VY 9
#define OT_FS_I0 10   // image pass 1: sum (IRR_VAR: of non-empty pixels; CENTER: of the windowed image)
#define OT_FS_I1 11   // image pass 1: count of non-empty pixels / sum of squared gradients
#define OT_FS_I2 12   // image pass 2: sum of squared deviations

OT_DEV double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// one atomic per workgroup for each of NV partial sums
template <int NV>
OT_DEV void block_atomic_add(double (&v)[NV], double* __restrict__ dst, const int* __restrict__ slot) {
    __shared__ double part[NV][16];
    const int lane = __lane_id(), wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int k = 0; k < NV; k++) {
        double s = wave_sum(v[k]);
        if (lane == 0) part[k][wave] = s;
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        double s = 0.0;
        for (int j = 0; j < nw; j++) s += part[threadIdx.x][j];
        if (s != 0.0) unsafeAtomicAdd(&dst[slot[threadIdx.x]], s);
    }
}

// raytracer.py:1552-1583: section index pos = argmax(z < p_z) - 1 of every ray of [first, first + count),
// rays without such a section are left out; pa = p - s / s_z * p_z, sb = s / s_z with the normalised section
// direction.  out: pasb[4 * count] = pax | pay | sbx | sby, w[count] (-1: left out), n_use += rays used.
__global__ __launch_bounds__(256) void focus_prepare_kernel(ot_rays R, int64_t first, int64_t count, double z,
                                                            double* __restrict__ pasb, float* __restrict__ w,
                                                            unsigned long long* __restrict__ n_use) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = q < count;
    bool used = false;
    if (active) {
        const int64_t r = first + q, N = R.N;
        const int nt = R.nt;
        int k = -1;
        for (int j = 0; j < nt; j++) {
            if (z < R.p[r + N * (j + 2 * (int64_t)nt)]) {
                k = j - 1;  // j == 0: argmax(...) - 1 == -1, ray starts behind the region
                break;
            }
        }
        float wi = -1.f;
        double pax = 0.0, pay = 0.0, sbx = 0.0, sby = 0.0;
        if (k >= 0) {
            V3 p = {R.p[r + N * k], R.p[r + N * (k + (int64_t)nt)], R.p[r + N * (k + 2 * (int64_t)nt)]};
            V3 s = section_dir(R, r, k);
            sbx = s.x / s.z;
            sby = s.y / s.z;
            pax = p.x - sbx * p.z;
            pay = p.y - sby * p.z;
            wi = R.w[r + N * k];
            used = true;
        }
        pasb[q] = pax;
        pasb[q + count] = pay;
        pasb[q + 2 * count] = sbx;
        pasb[q + 3 * count] = sby;
        w[q] = wi;
    }
    unsigned long long m = __ballot(used);
    if (__lane_id() == 0 && m) atomicAdd(n_use, (unsigned long long)__popcll(m));
}

__global__ void focus_init_kernel(double* __restrict__ ws) {
    const double inf = __builtin_inf();
    int i = threadIdx.x;
    if (i < OT_FOCUS_WS) ws[i] = (i == OT_FS_XMIN || i == OT_FS_YMIN) ? inf : (i == OT_FS_XMAX || i == OT_FS_YMAX) ? -inf : 0.0;
}

// pass 1: extent of the hit positions and the weighted first moments (np.cov / np.average inputs)
__global__ __launch_bounds__(256) void focus_stats_kernel(int64_t n, const double* __restrict__ pasb, const float* __restrict__ w,
                                                          double z, double* __restrict__ ws) {
    const double inf = __builtin_inf();
    double xmin = inf, xmax = -inf, ymin = inf, ymax = -inf;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float wf = w[i];
        if (wf < 0.f) continue;
        double x = pasb[i] + pasb[i + 2 * n] * z, y = pasb[i + n] + pasb[i + 3 * n] * z;
        double wd = (double)wf;
        xmin = fmin(xmin, x);
        xmax = fmax(xmax, x);
        ymin = fmin(ymin, y);
        ymax = fmax(ymax, y);
        acc[0] += wd;
        acc[1] += wd * wd;
        acc[2] += x * wd;
        acc[3] += y * wd;
    }
    xmin = wave_min(xmin);
    xmax = wave_max(xmax);
    ymin = wave_min(ymin);
    ymax = wave_max(ymax);
    if (__lane_id() == 0 && xmin <= xmax) {
        atomic_min_f64(&ws[OT_FS_XMIN], xmin);
        atomic_max_f64(&ws[OT_FS_XMAX], xmax);
        atomic_min_f64(&ws[OT_FS_YMIN], ymin);
        atomic_max_f64(&ws[OT_FS_YMAX], ymax);
    }
    const int slot[4] = {OT_FS_W, OT_FS_W2, OT_FS_WX, OT_FS_WY};
    __shared__ int sslot[4];
    if (threadIdx.x < 4) sslot[threadIdx.x] = slot[threadIdx.x];
    __syncthreads();
    block_atomic_add<4>(acc, ws, sslot);
}

// pass 2 (RMS Spot Size): centred second moments, np.cov(x, aweights=w): sum (x - avg) * ((x - avg) * w)
__global__ __launch_bounds__(256) void focus_var_kernel(int64_t n, const double* __restrict__ pasb, const float* __restrict__ w,
                                                        double z, double* __restrict__ ws) {
    const double sw = ws[OT_FS_W];
    const double ax = ws[OT_FS_WX] / sw, ay = ws[OT_FS_WY] / sw;
    double acc[2] = {0.0, 0.0};
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float wf = w[i];
        if (wf < 0.f) continue;
        double x = pasb[i] + pasb[i + 2 * n] * z, y = pasb[i + n] + pasb[i + 3 * n] * z;
        double wd = (double)wf, dx = x - ax, dy = y - ay;
        acc[0] += dx * (dx * wd);
        acc[1] += dy * (dy * wd);
    }
    __shared__ int sslot[2];
    if (threadIdx.x == 0) {
        sslot[0] = OT_FS_VX;
        sslot[1] = OT_FS_VY;
    }
    __syncthreads();
    block_atomic_add<2>(acc, ws, sslot);
}

// pass 2 (image methods): bin the hits into the npx x npx image over their own extent
// (misc.binning_indices_2d misc.py:59-91, np.add.at).  LDS hash privatisation as in render_kernel: near the
// focus nearly all rays share a handful of pixels.
#define OT_FHASH_N 4096
__global__ __launch_bounds__(1024) void focus_bin_kernel(int64_t n, const double* __restrict__ pasb, const float* __restrict__ w,
                                                         double z, const double* __restrict__ ws, int npx,
                                                         double* __restrict__ img) {
    __shared__ double hval[OT_FHASH_N];
    __shared__ int hkey[OT_FHASH_N];
    for (int i = threadIdx.x; i < OT_FHASH_N; i += blockDim.x) {
        hkey[i] = -1;
        hval[i] = 0.0;
    }
    __syncthreads();
    const double x0 = ws[OT_FS_XMIN], x1 = ws[OT_FS_XMAX], y0 = ws[OT_FS_YMIN], y1 = ws[OT_FS_YMAX];
    const double fx = (double)npx / (x1 - x0), fy = (double)npx / (y1 - y0);
    // contiguous piece of the rays per workgroup, as in render_kernel (neighbouring rays share their source)
    const int64_t chunk = ((n + gridDim.x - 1) / gridDim.x + blockDim.x - 1) / blockDim.x * blockDim.x;
    const int64_t i_end = ((int64_t)(blockIdx.x + 1) * chunk < n) ? (int64_t)(blockIdx.x + 1) * chunk : n;
    for (int64_t i = (int64_t)blockIdx.x * chunk + threadIdx.x; i < i_end; i += blockDim.x) {
        float wf = w[i];
        if (!(wf > 0.f)) continue;  // left out, or weight 0: adds nothing
        double x = pasb[i] + pasb[i + 2 * n] * z, y = pasb[i + n] + pasb[i + 3 * n] * z;
        double gx = floor(fx * (x - x0)), gy = floor(fy * (y - y0));
        if (y == y1) gy = (double)(npx - 1);
        if (x == x1) gx = (double)(npx - 1);
        if (!(gx >= 0.0 && gy >= 0.0 && gx < (double)npx && gy < (double)npx)) continue;
        const int pix = (int)gy * npx + (int)gx;
        unsigned int h = ((unsigned int)pix * 2654435761u) >> (32 - 12);
        int slot = -1;
#pragma unroll
        for (int pr = 0; pr < 4; pr++) {
            int sidx = (int)((h + pr) & (OT_FHASH_N - 1));
            int k = hkey[sidx];
            if (k == -1) k = atomicCAS(&hkey[sidx], -1, pix);
            if (k == -1 || k == pix) {
                slot = sidx;
                break;
            }
        }
        if (slot >= 0)
            unsafeAtomicAdd(&hval[slot], (double)wf);
        else
            unsafeAtomicAdd(&img[pix], (double)wf);
    }
    __syncthreads();
    for (int sidx = threadIdx.x; sidx < OT_FHASH_N; sidx += blockDim.x) {
        int k = hkey[sidx];
        if (k != -1 && hval[sidx] != 0.0) unsafeAtomicAdd(&img[k], hval[sidx]);
    }
}

// rotationally symmetric Hann window on the [-1, 1]^2 pixel grid (raytracer.py:1398-1402)
OT_DEV double focus_window(int ix, int iy, int npx) {
    const double step = 2.0 / (double)(npx - 1);
    double X = (double)ix * step + -1.0, Y = (double)iy * step + -1.0;
    double R = sqrt(X * X + Y * Y);
    return R > 1.0 ? 0.0 : 1.0 + cos(R * M_PI);
}

// image pass 1.  IRR_VAR: sum and count of the non-empty pixels.  SHARPNESS / CENTER_SHARPNESS: sum of the
// squared forward differences in both directions (raytracer.py:1412) of the (windowed) image and its sum.
__global__ __launch_bounds__(256) void focus_image1_kernel(const double* __restrict__ img, int npx, int mode, double* __restrict__ ws) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t np2 = (int64_t)npx * npx;
    double acc[2] = {0.0, 0.0};
    if (i < np2) {
        const int iy = (int)(i / npx), ix = (int)(i - (int64_t)iy * npx);
        if (mode == OT_FOCUS_IRR_VAR) {
            double v = img[i];
            if (v > 0.0) {
                acc[0] = v;
                acc[1] = 1.0;
            }
        } else {
            const bool win = mode == OT_FOCUS_CENTER_SHARPNESS;
            double v = img[i] * (win ? focus_window(ix, iy, npx) : 1.0);
            double g = 0.0;
            if (ix + 1 < npx) {
                double d = img[i + 1] * (win ? focus_window(ix + 1, iy, npx) : 1.0) - v;
                g += d * d;
            }
            if (iy + 1 < npx) {
                double d = img[i + npx] * (win ? focus_window(ix, iy + 1, npx) : 1.0) - v;
                g += d * d;
            }
            acc[0] = v;
            acc[1] = g;
        }
    }
    __shared__ int sslot[2];
    if (threadIdx.x == 0) {
        sslot[0] = OT_FS_I0;
        sslot[1] = OT_FS_I1;
    }
    __syncthreads();
    block_atomic_add<2>(acc, ws, sslot);
}

// image pass 2 (IRR_VAR): squared deviations of the non-empty pixels from their mean (ndarray.var)
__global__ __launch_bounds__(256) void focus_image2_kernel(const double* __restrict__ img, int npx, double* __restrict__ ws) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const double mean = ws[OT_FS_I0] / ws[OT_FS_I1];
    double acc[1] = {0.0};
    if (i < (int64_t)npx * npx) {
        double v = img[i];
        if (v > 0.0) acc[0] = (v - mean) * (v - mean);
    }
    __shared__ int sslot[1];
    if (threadIdx.x == 0) sslot[0] = OT_FS_I2;
    __syncthreads();
    block_atomic_add<1>(acc, ws, sslot);
}

// cost value of this z sample from the workspace sums (raytracer.py:1376-1379, 1404-1418)
__global__ void focus_finalize_kernel(int mode, int npx, const double* __restrict__ ws, double* __restrict__ cost) {
    if (threadIdx.x || blockIdx.x) return;
    double c;
    if (mode == OT_FOCUS_RMS) {
        const double sw = ws[OT_FS_W];
        const double fact = sw - 1.0 * ws[OT_FS_W2] / sw;  // np.cov: w_sum - ddof * sum(w * aweights) / w_sum
        const double f = 1.0 / fact;
        c = sqrt(ws[OT_FS_VX] * f + ws[OT_FS_VY] * f);
    } else if (mode == OT_FOCUS_IRR_VAR) {
        const double var = ws[OT_FS_I2] / ws[OT_FS_I1];
        const double ap = (ws[OT_FS_XMAX] - ws[OT_FS_XMIN]) * (ws[OT_FS_YMAX] - ws[OT_FS_YMIN]) / ((double)npx * (double)npx);
        c = -log(var / (ap * ap));
    } else if (mode == OT_FOCUS_SHARPNESS) {
        c = -ws[OT_FS_I1];
    } else {
        const double s = ws[OT_FS_I0];
        c = (s != 0.0) ? -(ws[OT_FS_I1] / (s * s)) : -ws[OT_FS_I1];
    }
    cost[0] = c;
}

// direct RMS solution (raytracer.py:1420-1460), second pass: with the weighted mean line through the bounds
// known (sums[0..4] = sum w, w pax, w pay, w sbx, w sby), accumulate sum w^2 (dtx^2 + dty^2) and
// sum w^2 (dtx dx + dty dy) into sums[5], sums[6].
__global__ __launch_bounds__(256) void focus_moments1_kernel(int64_t n, const double* __restrict__ pasb, const float* __restrict__ w,
                                                             double* __restrict__ sums) {
    double acc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float wf = w[i];
        if (wf < 0.f) continue;
        double wd = (double)wf;
        acc[0] += wd;
        acc[1] += pasb[i] * wd;
        acc[2] += pasb[i + n] * wd;
        acc[3] += pasb[i + 2 * n] * wd;
        acc[4] += pasb[i + 3 * n] * wd;
        acc[5] += wd * wd;
    }
    __shared__ int sslot[6];
    if (threadIdx.x < 5) sslot[threadIdx.x] = threadIdx.x;
    if (threadIdx.x == 5) sslot[5] = 7;
    __syncthreads();
    block_atomic_add<6>(acc, sums, sslot);
}

// second pass: the two sums of the direct solution (sums[5], sums[6]) and the centred second moments of the hit
// line about the weighted mean line, taken at z0 = (b0 + b1) / 2 (sums[8..13]): with x'(z) = x0' + sb' (z - z0),
// sum w x'^2 = S8 + 2 (z - z0) S9 + (z - z0)^2 S10 for every z -- the whole RMS cost curve from one pass.
__global__ __launch_bounds__(256) void focus_moments2_kernel(int64_t n, const double* __restrict__ pasb, const float* __restrict__ w,
                                                             double b0, double b1, double* __restrict__ sums) {
    const double sw = sums[0];
    const double mpx = sums[1] / sw, mpy = sums[2] / sw, msx = sums[3] / sw, msy = sums[4] / sw;
    // mean position at the bounds, direction of the mean position
    const double pb0x = mpx + msx * b0, pb0y = mpy + msy * b0;
    const double pb1x = mpx + msx * b1, pb1y = mpy + msy * b1;
    const double vz = b1 - b0, vxz = (pb1x - pb0x) / vz, vyz = (pb1y - pb0y) / vz;
    const double z0 = 0.5 * (b0 + b1);
    const double m0x = mpx + msx * z0, m0y = mpy + msy * z0;
    double acc[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float wf = w[i];
        if (wf < 0.f) continue;
        const double pax = pasb[i], pay = pasb[i + n], sbx = pasb[i + 2 * n], sby = pasb[i + 3 * n];
        double wd = (double)wf, w2 = wd * wd;
        double dx = pax - pb0x, dy = pay - pb0y;
        double dtx = sbx - vxz, dty = sby - vyz;
        acc[0] += w2 * (dtx * dtx) + w2 * (dty * dty);
        acc[1] += dtx * dx * w2 + dty * dy * w2;
        const double x0 = (pax + sbx * z0) - m0x, y0 = (pay + sby * z0) - m0y, sx = sbx - msx, sy = sby - msy;
        acc[2] += wd * (x0 * x0);
        acc[3] += wd * (x0 * sx);
        acc[4] += wd * (sx * sx);
        acc[5] += wd * (y0 * y0);
        acc[6] += wd * (y0 * sy);
        acc[7] += wd * (sy * sy);
    }
    __shared__ int sslot[8];
    if (threadIdx.x < 2) sslot[threadIdx.x] = 5 + threadIdx.x;
    if (threadIdx.x >= 2 && threadIdx.x < 8) sslot[threadIdx.x] = 6 + threadIdx.x;
    __syncthreads();
    block_atomic_add<8>(acc, sums, sslot);
}
