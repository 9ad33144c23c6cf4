// FITPACK B-spline evaluation for DataSurface1D/2D (data_surface_2d.py:10-227).
//
// The reference builds scipy.interpolate.InterpolatedUnivariateSpline(k=4) / RectBivariateSpline(kx=ky=4) once
// per surface and evaluates them (and their first derivatives) for every ray through FITPACK (splev / splder,
// bispeu / pardeu).  The fit itself is one-off host set-up and stays with SciPy; the per-ray evaluation is
// restated here in FITPACK's own operation order (fpbspl's recurrence, splev's/fpbisp's summation order), so
// device values agree with SciPy's to the last bits.  Knots and coefficients live in HBM (SurfDev::tab); the
// knot interval is found from a uniform-spacing guess corrected by FITPACK's own search loops.
// Fortran indices (1-based) are kept in the comments: t(i) is t[i - 1].
#pragma once
// included by ot_device.hpp after its OT_HD / OT_DEV macros

// fpbspl.f: the K+1 non-zero B-splines of degree K at x, with t(l) <= x < t(l+1)
template <int K, class TP>
OT_HD void fpbspl(TP t, double x, int l, double* h) {
    double hh[K > 0 ? K : 1];
    h[0] = 1.0;
#pragma unroll
    for (int j = 1; j <= K; j++) {
#pragma unroll
        for (int i = 0; i < j; i++) hh[i] = h[i];
        h[0] = 0.0;
#pragma unroll
        for (int i = 1; i <= j; i++) {
            const double tli = t[l + i - 1], tlj = t[l + i - j - 1];
            if (tli == tlj) {
                h[i] = 0.0;
            } else {
                const double f = hh[i - 1] / (tli - tlj);
                h[i - 1] = h[i - 1] + f * (tli - x);
                h[i] = f * (x - tlj);
            }
        }
    }
}

// knot interval of splev.f / splder.f / fpbisp.f: l in [k1, nk1] with t(l) <= x < t(l+1) (ends clamped).
// k1 = degree of the *original* spline + 1; inv_h = approximate knots per unit length (first guess only).
template <class TP>
OT_HD int spl_interval(TP t, int n, int k1, double inv_h, double x) {
    const int nk1 = n - k1;
    double g = (x - t[k1 - 1]) * inv_h;
    int l = k1 + (g > 0.0 ? (g < (double)(nk1 - k1) ? (int)g : nk1 - k1) : 0);
    while (x < t[l - 1] && l != k1) l--;
    while (x >= t[l] && l != nk1) l++;
    return l;
}

// splev.f (der = 0, K = 4) and splder.f (der = 1: K = 3 on the same knots with the derivative coefficients);
// ext = 0: values outside the knot range are extrapolated from the end polynomials.
template <int K, class TP>
OT_HD double spl1_eval(TP t, int n, TP c, double inv_h, double x) {
    const int k1 = OT_SPL_K + 1;
    const int l = spl_interval(t, n, k1, inv_h, x);
    double h[K + 1];
    fpbspl<K>(t, x, l, h);
    double sp = 0.0;
#pragma unroll
    for (int j = 0; j <= K; j++) sp = sp + c[l - k1 + j] * h[j];
    return sp;
}

// bispeu.f / pardeu.f -> fpbisp.f for one point.  tx has nx knots of degree KX, ty ny knots of degree KY,
// c is (nx - KX - 1) x (ny - KY - 1) with y fastest.  Arguments are clamped to the knot range (no extrapolation).
template <int KX, int KY, class TP>
OT_HD double spl2_eval(TP tx, int nx, TP ty, int ny, TP c, double inv_h, double x, double y) {
    const int kx1 = KX + 1, ky1 = KY + 1, nkx1 = nx - kx1, nky1 = ny - ky1;
    double ax = x, ay = y;
    if (ax < tx[kx1 - 1]) ax = tx[kx1 - 1];
    if (ax > tx[nkx1]) ax = tx[nkx1];
    if (ay < ty[ky1 - 1]) ay = ty[ky1 - 1];
    if (ay > ty[nky1]) ay = ty[nky1];
    const int lx = spl_interval(tx, nx, kx1, inv_h, ax);
    const int ly = spl_interval(ty, ny, ky1, inv_h, ay);
    double hx[KX + 1], hy[KY + 1];
    fpbspl<KX>(tx, ax, lx, hx);
    fpbspl<KY>(ty, ay, ly, hy);
    double sp = 0.0;
    int l1 = (lx - kx1) * nky1 + (ly - ky1);
#pragma unroll
    for (int i1 = 0; i1 <= KX; i1++) {
#pragma unroll
        for (int j1 = 0; j1 <= KY; j1++) sp = sp + c[l1 + j1] * hx[i1] * hy[j1];
        l1 += nky1;
    }
    return sp;
}
