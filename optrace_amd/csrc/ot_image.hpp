// Image conversion of the rendered XYZW histogram: RenderImage.get render_image.py:131-222 with
// color.xyz_to_srgb / xyz_to_srgb_linear srgb.py:267-407, color.xyz_to_luv / luv_to_xyz / luv_* luv.py,
// color.xyz_to_xyY xyz.py, the chroma clipping helper _triangle_intersect srgb.py:133-183 and _get_chroma_scale
// srgb.py:186-222.  One lane per (down-binned) pixel; the few image-wide quantities (maxima, any-flags, the
// minimum chroma factor) are reduced on the device between the passes (wave shuffle + one atomic per wave).
#pragma once
#include "ot_detector.hpp"

#define OT_IMG_IRRADIANCE 0
#define OT_IMG_ILLUMINANCE 1
#define OT_IMG_SRGB_ABSOLUTE 2
#define OT_IMG_SRGB_PERCEPTUAL 3
#define OT_IMG_OUTSIDE_GAMUT 4
#define OT_IMG_LIGHTNESS 5
#define OT_IMG_HUE 6
#define OT_IMG_CHROMA 7
#define OT_IMG_SATURATION 8

// slots of the reduction scratch (doubles)
#define OT_RED_YMAX 0        // nanmax(Y | Y > 0)                                  luv.py: Yn (normalize=True)
#define OT_RED_RGBMAX 1      // nanmax(M XYZ) of the unmodified image               srgb.py:_to_srgb
#define OT_RED_ANY_INV 2     // any(RGBL < 0)                                       srgb.py:317
#define OT_RED_LMAX 3        // max L (Luv, normalize=False) of the clipped image   srgb.py:249
#define OT_RED_ANY_GAMUT 4   // any(in_gamut)                                       srgb.py:209
#define OT_RED_CRMIN 5       // min cr_fact2 over valid & L > L_th * Lmax           srgb.py:250-251
#define OT_RED_RGBMAX2 6     // nanmax(M XYZ') of the corrected image
#define OT_RED_N 8

OT_DEV void to_rgbl(double X, double Y, double Z, double& r, double& g, double& b) {  // srgb.py:124-128
    r = 3.2404542 * X + -1.5371385 * Y + -0.4985314 * Z;
    g = -0.9692660 * X + 1.8760108 * Y + 0.0415560 * Z;
    b = 0.0556434 * X + -0.2040259 * Y + 1.0572252 * Z;
}

// luv.py xyz_to_luv for one pixel (xyz already clipped at 0 by the caller where the reference clips)
OT_DEV void xyz_to_luv1(double X, double Y, double Z, double Yn, double& L, double& u, double& v) {
    X = fmax(X, 0.0);
    Y = fmax(Y, 0.0);
    Z = fmax(Z, 0.0);
    L = u = v = 0.0;
    if (!(Y > 0)) return;
    const double un = 0.19783982, vn = 0.4683363;
    double t = 1 / Yn * Y;
    L = (t > 0.008856) ? 116 * cbrt(t) - 16 : 903.3 * t;
    double D = 1 / (X + 15 * Y + 3 * Z);
    double uu = 4 * X * D, vv = 9 * Y * D;
    double L13 = 13 * L;
    u = L13 * (uu - un);
    v = L13 * (vv - vn);
}

OT_DEV void luv_to_xyz1(double L, double u, double v, double& X, double& Y, double& Z) {  // luv.py luv_to_xyz
    X = Y = Z = 0.0;
    if (!(L > 0)) return;
    const double un = 0.19783982, vn = 0.4683363;
    if (L > 903.3 * 0.008856) {
        double q = 1.0 / 116 * (L + 16);
        Y = q * q * q;
    } else {
        Y = 1 / 903.3 * L;
    }
    double L13 = 13 * L;
    X = 9.0 / 4 * Y * (u + L13 * un) / (v + L13 * vn);
    Z = 3 * Y * (L13 / (v + L13 * vn) - 5.0 / 3) - 1.0 / 3 * X;
}

// srgb.py:_triangle_intersect: project (x, y) towards the whitepoint w onto the gamut triangle r, g, b
OT_DEV void triangle_intersect(double rx, double ry, double gx, double gy, double bx, double by, double wx, double wy,
                               double& x, double& y) {
    double phir = atan2(ry - wy, rx - wx);
    double phig = atan2(gy - wy, gx - wx);
    double phib = atan2(by - wy, bx - wx) + 2 * M_PI;
    double phi = atan2(y - wy, x - wx);
    if (phi < 0) phi += 2 * M_PI;
    double aw = tan(phi);
    double abg = (gy - by) / (gx - bx), abr = (ry - by) / (rx - bx), agr = (ry - gy) / (rx - gx);
    bool is_bg = (phi <= phib) && (phi > phig);
    bool is_gr = (phi <= phig) && (phi > phir);
    if (is_bg) {
        x = (y - x * aw + (bx * abg - by)) / (abg - aw);
        y = x * abg + (by - bx * abg);
    } else if (is_gr) {
        x = (y - x * aw + (gx * agr - gy)) / (agr - aw);
        y = x * agr + (gy - gx * agr);
    } else {
        x = (y - x * aw + (bx * abr - by)) / (abr - aw);
        y = x * abr + (by - bx * abr);
    }
}

OT_DEV double srgb_gamma(double v) {  // srgb_linear_to_srgb srgb.py:358-376
    double a = 0.055, av = fabs(v);
    if (av <= 0.0031308) return v * 12.92;
    double sg = (v > 0) - (v < 0);
    return sg * ((1 + a) * pow(av, 1 / 2.4) - a);
}

OT_DEV void wave_atomic_max(double* addr, double v) {  // NaN-ignoring maximum (np.nanmax)
    double m = wave_max(isnan(v) ? -__builtin_inf() : v);
    if (__lane_id() == 0 && m > -__builtin_inf()) atomic_max_f64(addr, m);
}

OT_DEV void wave_atomic_min(double* addr, double v) {
    double m = wave_min(isnan(v) ? __builtin_inf() : v);
    if (__lane_id() == 0 && m < __builtin_inf()) atomic_min_f64(addr, m);
}

// INTER_AREA down-binning by an integer factor = mean of fact x fact bins (render_image.py:174), 4 channels
__global__ __launch_bounds__(256) void img_downbin_kernel(const double* __restrict__ hist, int Nx, int Ny, int fact,
                                                          double* __restrict__ out) {
    const int nx = Nx / fact, ny = Ny / fact;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)nx * ny) return;
    const int y = (int)(i / nx), x = (int)(i - (int64_t)y * nx);
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    for (int dy = 0; dy < fact; dy++)
        for (int dx = 0; dx < fact; dx++) {
            const double* h = hist + (((int64_t)(y * fact + dy)) * Nx + (x * fact + dx)) * 4;
            a0 += h[0];
            a1 += h[1];
            a2 += h[2];
            a3 += h[3];
        }
    const double inv = 1.0 / ((double)fact * fact);
    out[i * 4 + 0] = a0 * inv;
    out[i * 4 + 1] = a1 * inv;
    out[i * 4 + 2] = a2 * inv;
    out[i * 4 + 3] = a3 * inv;
}

// pass 1: image-wide quantities of the unmodified image
__global__ __launch_bounds__(256) void img_reduce1_kernel(const double* __restrict__ img, int64_t npx, double* __restrict__ red) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool act = i < npx;
    double X = 0, Y = 0, Z = 0;
    if (act) {
        X = img[i * 4];
        Y = img[i * 4 + 1];
        Z = img[i * 4 + 2];
    }
    double r, g, b;
    to_rgbl(X, Y, Z, r, g, b);
    const double ninf = -__builtin_inf();
    wave_atomic_max(&red[OT_RED_YMAX], (act && fmax(Y, 0.0) > 0) ? fmax(Y, 0.0) : ninf);
    wave_atomic_max(&red[OT_RED_RGBMAX], act ? fmax(fmax(r, g), b) : ninf);
    bool inv = act && (r < 0 || g < 0 || b < 0);
    if (__ballot(inv) && __lane_id() == 0) red[OT_RED_ANY_INV] = 1.0;
    // Luv of the clipped image with Yn = 1 (normalize=False) for the perceptual intent
    double L, u, v;
    xyz_to_luv1(X, Y, Z, 1.0, L, u, v);
    wave_atomic_max(&red[OT_RED_LMAX], act ? L : ninf);
}

// srgb.py:_get_chroma_scale for one pixel: valid-colour mask and squared chroma factor towards the sRGB triangle
OT_DEV void chroma_scale1(double L, double u, double v, bool& in_gamut, double& cr2) {
    const double un = 0.19783982, vn = 0.4683363;
    double u_ = un, v_ = vn;
    if (L > 0) {
        u_ += 1.0 / 13 * u / L;
        v_ += 1.0 / 13 * v / L;
    }
    bool l1 = v_ > (0.5065 - 0.013) / (0.6235 - 0.255) * (u_ - 0.2555) + 0.01373;
    bool l2 = v_ < (0.5065 - 0.6) / (0.6235 - 0.0) * u_ + 0.6;
    bool l3 = u_ > 0;
    bool l4 = v_ > (0.013 - 0.28) / (0.255 - 0) * u_ + 0.28;
    bool l5 = v_ > (0.0 - 0.48) / (0.18 - 0) * u_ + 0.48;
    in_gamut = l1 && l2 && l3 && l4 && l5;
    double cr0 = (u_ - un) * (u_ - un) + (v_ - vn) * (v_ - vn);
    triangle_intersect(0.4507042254, 0.5228873239, 0.125, 0.5625, 0.1754385965, 0.1578947368, un, vn, u_, v_);
    double cr1 = (u_ - un) * (u_ - un) + (v_ - vn) * (v_ - vn);
    cr2 = cr1 / (cr0 + 1e-9);
}

// pass 2 (perceptual intent): any(in_gamut) and the minimum squared chroma factor over valid, bright pixels
__global__ __launch_bounds__(256) void img_reduce2_kernel(const double* __restrict__ img, int64_t npx, double L_th,
                                                          double* __restrict__ red) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool act = i < npx;
    double L = 0, u = 0, v = 0;
    if (act) xyz_to_luv1(img[i * 4], img[i * 4 + 1], img[i * 4 + 2], 1.0, L, u, v);
    bool in_gamut;
    double cr2;
    chroma_scale1(L, u, v, in_gamut, cr2);
    in_gamut = in_gamut && act;
    if (__ballot(in_gamut) && __lane_id() == 0) red[OT_RED_ANY_GAMUT] = 1.0;
    bool use = in_gamut && (L > L_th * red[OT_RED_LMAX]);
    wave_atomic_min(&red[OT_RED_CRMIN], use ? cr2 : __builtin_inf());
}

// pass 3: corrected XYZ' per pixel (in place in img[..., :3]) and nanmax(M XYZ')
//   intent 0 = Ignore, 1 = Absolute, 2 = Perceptual with the final per-image chroma_scale (srgb.py:313-352)
__global__ __launch_bounds__(256) void img_correct_kernel(double* __restrict__ img, int64_t npx, int intent,
                                                          double chroma_scale, int use_ones, double* __restrict__ red) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool act = i < npx;
    double X = 0, Y = 0, Z = 0;
    if (act) {
        X = img[i * 4];
        Y = img[i * 4 + 1];
        Z = img[i * 4 + 2];
    }
    double r, g, b;
    to_rgbl(X, Y, Z, r, g, b);
    if (intent == 1) {
        if (r < 0 || g < 0 || b < 0) {  // chroma-clip towards the whitepoint in xy (srgb.py:322-330)
            double s = X + Y + Z;
            double x = 0.31272, y = 0.32903;
            if (s > 0) {
                x = X / s;
                y = Y / s;
            }
            triangle_intersect(0.64, 0.33, 0.30, 0.60, 0.15, 0.06, 0.31272, 0.32903, x, y);
            double k = Y / ((y > 0) ? y : __builtin_inf());
            X = k * x;
            Z = k * (1 - x - y);
        }
    } else if (intent == 2) {
        double L, u, v;
        xyz_to_luv1(X, Y, Z, 1.0, L, u, v);
        bool in_gamut;
        double cr2 = 1.0;
        if (!use_ones) chroma_scale1(L, u, v, in_gamut, cr2);  // srgb.py:209-210: all ones if nothing is in gamut
        double cr = sqrt(cr2);
        if (cr > chroma_scale) cr = chroma_scale;
        luv_to_xyz1(L, u * cr, v * cr, X, Y, Z);
    }
    if (act) {
        img[i * 4] = X;
        img[i * 4 + 1] = Y;
        img[i * 4 + 2] = Z;
    }
    to_rgbl(X, Y, Z, r, g, b);
    wave_atomic_max(&red[OT_RED_RGBMAX2], act ? fmax(fmax(r, g), b) : -__builtin_inf());
}

// final pass: write the requested quantity
__global__ __launch_bounds__(256) void img_final_kernel(const double* __restrict__ img, int64_t npx, int mode, double apx,
                                                        double K, const double* __restrict__ red, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npx) return;
    const double X = img[i * 4], Y = img[i * 4 + 1], Z = img[i * 4 + 2], W = img[i * 4 + 3];
    const bool normalize = !(mode & OT_IMG_FLAG_NO_NORMALIZE), clip = !(mode & OT_IMG_FLAG_NO_CLIP);
    mode &= ~(OT_IMG_FLAG_NO_NORMALIZE | OT_IMG_FLAG_NO_CLIP);
    switch (mode) {
        case OT_IMG_IRRADIANCE: out[i] = 1 / apx * W; return;
        case OT_IMG_ILLUMINANCE: out[i] = K / apx * Y; return;
        case OT_IMG_SRGB_ABSOLUTE:
        case OT_IMG_SRGB_PERCEPTUAL: {
            double r, g, b;
            to_rgbl(X, Y, Z, r, g, b);
            double nmax = red[OT_RED_RGBMAX2];
            if (normalize && nmax != 0 && !isnan(nmax) && isfinite(nmax)) {  // `if normalize and (nmax := np.nanmax(RGBL_))`
                double s = 1 / nmax;
                r *= s;
                g *= s;
                b *= s;
            }
            if (clip) {  // color.xyz_to_srgb srgb.py:403-404
                r = fmin(fmax(r, 0.0), 1.0);
                g = fmin(fmax(g, 0.0), 1.0);
                b = fmin(fmax(b, 0.0), 1.0);
            }
            out[i * 3 + 0] = srgb_gamma(r);
            out[i * 3 + 1] = srgb_gamma(g);
            out[i * 3 + 2] = srgb_gamma(b);
            return;
        }
        case OT_IMG_OUTSIDE_GAMUT: {
            double r, g, b;
            to_rgbl(X, Y, Z, r, g, b);
            double nmax = red[OT_RED_RGBMAX];
            if (nmax != 0 && isfinite(nmax)) {
                double s = 1 / nmax;
                r *= s;
                g *= s;
                b *= s;
            }
            out[i] = (r < -1e-6 || g < -1e-6 || b < -1e-6) ? 1.0 : 0.0;
            return;
        }
        default: {
            double Yn = red[OT_RED_YMAX];
            double L = 0, u = 0, v = 0;
            if (isfinite(Yn)) xyz_to_luv1(X, Y, Z, Yn, L, u, v);  // no pixel with Y > 0: all zero (luv.py:34-35)
            if (mode == OT_IMG_LIGHTNESS) out[i] = L;
            else if (mode == OT_IMG_CHROMA) out[i] = sqrt(u * u + v * v);
            else if (mode == OT_IMG_SATURATION) out[i] = (L > 0) ? sqrt(u * u + v * v) / L : 0.0;
            else {
                double hue = 180 / M_PI * atan2(v, u);
                if (hue < 0) hue += 360;
                out[i] = hue;
            }
        }
    }
}

// RenderImage._apply_rayleigh_filter render_image.py:257-296: "same"-size convolution of every channel with the
// (2ps+1)^2 Airy kernel.  The reference uses scipy.signal.fftconvolve; the kernel is small against the image, so
// a direct sum per output pixel (zero-padded borders, zero taps skipped, kernel staged in LDS) gives the same
// result without FFT round-off (the reference clamps its negative FFT noise to 0 afterwards).
__global__ __launch_bounds__(256) void img_convolve_kernel(const double* __restrict__ in, int Nx, int Ny,
                                                           const double* __restrict__ psf, int ps, double* __restrict__ out) {
    extern __shared__ double kern[];
    const int side = 2 * ps + 1;
    for (int k = threadIdx.x; k < side * side; k += blockDim.x) kern[k] = psf[k];
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)Nx * Ny) return;
    const int y = (int)(i / Nx), x = (int)(i - (int64_t)y * Nx);
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    for (int j = 0; j < side; j++) {
        const int yy = y + ps - j;  // (f * g)[y] = sum_j g[j] f[y - (j - ps)]
        if (yy < 0 || yy >= Ny) continue;
        for (int k = 0; k < side; k++) {
            const int xx = x + ps - k;
            if (xx < 0 || xx >= Nx) continue;
            const double g = kern[j * side + k];
            if (g == 0.0) continue;
            const double* h = in + ((int64_t)yy * Nx + xx) * 4;
            a0 += g * h[0];
            a1 += g * h[1];
            a2 += g * h[2];
            a3 += g * h[3];
        }
    }
    double* o = out + i * 4;
    o[0] = fmax(a0, 0.0);
    o[1] = fmax(a1, 0.0);
    o[2] = fmax(a2, 0.0);
    o[3] = fmax(a3, 0.0);
}
