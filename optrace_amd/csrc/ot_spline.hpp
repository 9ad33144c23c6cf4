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

// The same K+1 basis values on UNIFORM knots, as polynomials of u = (x - t(l)) / (t(l+1) - t(l)) in [0, 1): one
// division and ~20 multiply-adds instead of K(K+1)/2 divisions.  Data on an equidistant grid gives uniform knots
// everywhere except next to the two ends (FITPACK's interpolating spline: t(6) .. t(n-5) equidistant for k = 4),
// so nearly every evaluation takes this path; it agrees with the recurrence to the rounding noise of the knot
// positions (~1e-14 relative).
template <int K>
OT_HD void bspl_uniform(double u, double* h) {
    const double u2 = u * u, u3 = u2 * u;
    if (K == 4) {
        const double u4 = u2 * u2, v = 1.0 - u, v2 = v * v;
        h[0] = v2 * v2 * (1.0 / 24);
        h[1] = (((-4.0 * u + 12.0) * u - 6.0) * u - 12.0) * u * (1.0 / 24) + 11.0 / 24;
        h[2] = (((6.0 * u - 12.0) * u - 6.0) * u + 12.0) * u * (1.0 / 24) + 11.0 / 24;
        h[3] = ((((-4.0 * u + 4.0) * u + 6.0) * u + 4.0) * u + 1.0) * (1.0 / 24);
        h[4] = u4 * (1.0 / 24);
    } else {  // K == 3
        const double v = 1.0 - u;
        h[0] = v * v * v * (1.0 / 6);
        h[1] = ((3.0 * u - 6.0) * u2 + 4.0) * (1.0 / 6);
        h[2] = (((-3.0 * u + 3.0) * u + 3.0) * u + 1.0) * (1.0 / 6);
        h[3] = u3 * (1.0 / 6);
    }
}

// d/du of the uniform quartic basis (bspl_uniform<4>)
OT_HD void bspl_uniform4_deriv(double u, double* d) {
    const double v = 1.0 - u, u2 = u * u;
    d[0] = -(v * v * v) * (1.0 / 6);
    d[1] = (((-4.0 * u + 9.0) * u - 3.0) * u - 3.0) * (1.0 / 6);
    d[2] = (((2.0 * u - 3.0) * u - 1.0) * u + 1.0) * 0.5;
    d[3] = (((-4.0 * u + 3.0) * u + 3.0) * u + 1.0) * (1.0 / 6);
    d[4] = u2 * u * (1.0 / 6);
}

// basis values at x in knot interval l of the knot array t (n knots, uniform from index ulo to uhi, 0-based)
template <int K, class TP>
OT_HD void bspl_basis(TP t, int ulo, int uhi, double x, int l, double* h) {
    if (l - K >= ulo && l + K - 1 <= uhi) {
        const double tl = t[l - 1];
        bspl_uniform<K>((x - tl) / (t[l] - tl), h);
    } else {
        fpbspl<K>(t, x, l, h);
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

// The equidistant part of a knot array described without the array: knot i (0-based, ulo <= i <= uhi) sits at
// t0 + (i - ulo) * h; [lo, hi] = t(k1), t(nk1 + 1) is the range arguments are clamped to.  With it the knot interval and
// the local coordinate of a point come from arithmetic instead of the two dependent knot loads per dimension -- the
// feature level 2 kernels run two waves per SIMD, so every such round trip (~1 us under load) is exposed, and the
// Illinois search strings ~10 evaluations together.  The knot positions differ from the stored ones by their rounding
// (~1 ulp of t), like the equidistant basis polynomials above.
struct UniformKnots {
    double t0, h, inv_h, lo, hi;
    int ulo, uhi;
};

// FITPACK's interval l (t(l) <= x < t(l+1), 1-based) and u = (x - t(l)) / h; false near the ends of the knot array,
// where the knots are not equidistant (the caller takes the table path)
template <int K>
OT_HD bool uniform_cell(const UniformKnots& uk, double x, int& l, double& u) {
    const double fl = floor((x - uk.t0) * uk.inv_h);
    l = uk.ulo + (int)fl + 1;
    u = (x - (uk.t0 + fl * uk.h)) * uk.inv_h;
    return (l - K >= uk.ulo) && (l + K - 1 <= uk.uhi);  // false for NaN as well: (int)NaN = 0 fails the first test
}

// splev.f (der = 0, K = 4) and splder.f (der = 1: K = 3 on the same knots with the derivative coefficients);
// ext = 0: values outside the knot range are extrapolated from the end polynomials.
struct PatchCache;
template <int K, class TP>
OT_HD double spl1_eval(TP t, int n, TP c, double inv_h, double x, const UniformKnots* uk = nullptr,
                       PatchCache* pc = nullptr);

// Per-lane copy of one 5 x 5 coefficient patch in LDS (tracing kernel, hit search on spline surfaces).  The Illinois
// iteration evaluates the surface ~8 times per ray, and from the third evaluation on the points lie in the same knot
// cell or its neighbour; rays of a wave share nothing (random pairing of strata and ray indices), so every evaluation
// would gather its 25 coefficients through the texture addresser again -- which is what bounds the kernel (TA busy
// 85 %, VALU 33 %).  The lane keeps the patch it gathered last (200 B, entries interleaved by lane: conflict-free)
// and gathers again only when the cell changes.
struct PatchCache {
    double* slot;       // this lane's entry 0; entry e lives at slot[e * stride]
    int stride;         // lanes of the workgroup
    const double* key;  // address of the first coefficient of the cached patch, or null
};
// (Keeping the patch's knot cell in registers as well, to skip the interval search and its 8 knot loads for points in
// the same cell, was tried: the feature level 2 kernels are at 200-230 VGPRs already, the ten more spilled to scratch
// and the scene ran 18.0 instead of 7.1 ms.)

template <int K, class TP>
OT_HD double spl1_eval(TP t, int n, TP c, double inv_h, double x, const UniformKnots* uk, PatchCache* pc) {
    const int k1 = OT_SPL_K + 1;
    double h[K + 1];
    int l;
    double u;
    if (uk != nullptr && uniform_cell<OT_SPL_K>(*uk, x, l, u)) {
        bspl_uniform<K>(u, h);
    } else {
        l = spl_interval(t, n, k1, inv_h, x);
        bspl_basis<K>(t, OT_SPL_K + 1, n - OT_SPL_K - 2, x, l, h);
    }
    double sp = 0.0;
    if (pc != nullptr) {  // the lane's K + 1 coefficients in the first entries of its LDS patch
        const double* first = (const double*)(c + (l - k1));
        if (pc->key != first) {
#pragma unroll
            for (int j = 0; j <= K; j++) pc->slot[j * pc->stride] = c[l - k1 + j];
            pc->key = first;
        }
#pragma unroll
        for (int j = 0; j <= K; j++) sp = sp + pc->slot[j * pc->stride] * h[j];
        return sp;
    }
#pragma unroll
    for (int j = 0; j <= K; j++) sp = sp + c[l - k1 + j] * h[j];
    return sp;
}

// First derivative of the quartic spline at x from the lane's cached coefficients (the cell of the hit point), like
// spl2_grad_cached below; false outside the equidistant part or when the cache holds another cell.
template <class TP>
OT_HD bool spl1_grad_cached(TP c, double x, const UniformKnots* uk, PatchCache* pc, double& d) {
    constexpr int K = OT_SPL_K, k1 = K + 1;
    if (pc == nullptr || pc->key == nullptr || uk == nullptr) return false;
    int l;
    double u;
    if (!uniform_cell<K>(*uk, x, l, u)) return false;
    if (pc->key != (const double*)(c + (l - k1))) return false;
    double dh[k1];
    bspl_uniform4_deriv(u, dh);
    double sp = 0.0;
#pragma unroll
    for (int j = 0; j < k1; j++) sp = sp + pc->slot[j * pc->stride] * dh[j];
    d = sp * uk->inv_h;
    return true;
}

// bispeu.f / pardeu.f -> fpbisp.f for one point.  tx has nx knots of degree KX, ty ny knots of degree KY,
// c is (nx - KX - 1) x (ny - KY - 1) with y fastest.  Arguments are clamped to the knot range (no extrapolation).
// uk (value evaluation on the undifferentiated arrays only: KX = KY = OT_SPL_K): see UniformKnots.
template <int KX, int KY, class TP>
OT_HD double spl2_eval(TP tx, int nx, TP ty, int ny, TP c, double inv_h, double x, double y, PatchCache* pc = nullptr,
                       const UniformKnots* uk = nullptr) {
    const int kx1 = KX + 1, ky1 = KY + 1, nkx1 = nx - kx1, nky1 = ny - ky1;
    double ax = x, ay = y;
    double hx[KX + 1], hy[KY + 1];
    int lx, ly;
    bool fast = false;
    if (uk != nullptr) {
        ax = ax < uk->lo ? uk->lo : (ax > uk->hi ? uk->hi : ax);
        ay = ay < uk->lo ? uk->lo : (ay > uk->hi ? uk->hi : ay);
        double ux, uy;
        const bool fx = uniform_cell<KX>(*uk, ax, lx, ux), fy = uniform_cell<KY>(*uk, ay, ly, uy);
        fast = fx && fy;
        if (fast) {
            bspl_uniform<KX>(ux, hx);
            bspl_uniform<KY>(uy, hy);
        }
    }
    if (!fast) {
        if (ax < tx[kx1 - 1]) ax = tx[kx1 - 1];
        if (ax > tx[nkx1]) ax = tx[nkx1];
        if (ay < ty[ky1 - 1]) ay = ty[ky1 - 1];
        if (ay > ty[nky1]) ay = ty[nky1];
        lx = spl_interval(tx, nx, kx1, inv_h, ax);
        ly = spl_interval(ty, ny, ky1, inv_h, ay);
        // first / last equidistant knot (0-based) of each array: the derivative's array is the original minus its two
        // outer knots, i.e. shifted by one
        bspl_basis<KX>(tx, KX + 1, nx - KX - 2, ax, lx, hx);
        bspl_basis<KY>(ty, KY + 1, ny - KY - 2, ay, ly, hy);
    }
    double sp = 0.0;
    int l1 = (lx - kx1) * nky1 + (ly - ky1);
    if (pc != nullptr) {  // same sums in the same order, coefficients from the lane's LDS copy of the patch
        const double* first = (const double*)(c + l1);
        if (pc->key != first) {
#pragma unroll
            for (int i1 = 0; i1 <= KX; i1++)
#pragma unroll
                for (int j1 = 0; j1 <= KY; j1++) pc->slot[(i1 * (KY + 1) + j1) * pc->stride] = c[l1 + i1 * nky1 + j1];
            pc->key = first;
        }
        // Row sums with fused multiply-adds: sum_i hx_i (sum_j c_ij hy_j), 30 instructions instead of the 75 of fpbisp's
        // term-by-term order (sp += c hx hy).  The value moves by ~1e-16 relative -- three orders below what the
        // equidistant-knot arithmetic of this path already differs from the table path by (1e-13) -- and this branch
        // serves the hit search of the trace kernel only; the leaf operators (pc == nullptr) keep FITPACK's order.
#pragma unroll
        for (int i1 = 0; i1 <= KX; i1++) {
            double row = 0.0;
#pragma unroll
            for (int j1 = 0; j1 <= KY; j1++) row = __builtin_fma(pc->slot[(i1 * (KY + 1) + j1) * pc->stride], hy[j1], row);
            sp = __builtin_fma(row, hx[i1], sp);
        }
        return sp;
    }
#pragma unroll
    for (int i1 = 0; i1 <= KX; i1++) {
#pragma unroll
        for (int j1 = 0; j1 <= KY; j1++) sp = sp + c[l1 + j1] * hx[i1] * hy[j1];
        l1 += nky1;
    }
    return sp;
}

// Both first derivatives of the bi-quartic spline at (x, y) from the lane's cached coefficient patch -- the cell of the
// hit point, which the last value evaluation of the hit search left in LDS -- instead of two more 20-coefficient
// gathers from the derivative tables: dS/dx = sum c_ij B_i'(x) B_j(y).  FITPACK differentiates the coefficients and
// evaluates a spline of lower degree (splder / pardeu); on equidistant knots both are the same polynomial, evaluated
// in a different order (agreement ~1e-15 relative; normals never feed a hit mask).  Applies where both knot intervals
// lie in the equidistant part and the patch in LDS is the right one; returns false otherwise (table path).
template <class TP>
OT_HD bool spl2_grad_cached(TP t, int n, TP c, double inv_h, double x, double y, PatchCache* pc, double& sx, double& sy,
                            const UniformKnots* uk = nullptr) {
    constexpr int K = OT_SPL_K, k1 = K + 1;
    if (pc == nullptr || pc->key == nullptr) return false;
    const int nk1 = n - k1;
    if (uk != nullptr) {  // no knot loads: interval and local coordinate from the equidistant description
        const double cx = x < uk->lo ? uk->lo : (x > uk->hi ? uk->hi : x), cy = y < uk->lo ? uk->lo : (y > uk->hi ? uk->hi : y);
        int lx, ly;
        double ux, uy;
        const bool fx = uniform_cell<K>(*uk, cx, lx, ux), fy = uniform_cell<K>(*uk, cy, ly, uy);
        if (!(fx && fy)) return false;
        if (pc->key != (const double*)(c + ((lx - k1) * nk1 + (ly - k1)))) return false;
        double hx[k1], hy[k1], dx[k1], dy[k1];
        bspl_uniform<K>(ux, hx);
        bspl_uniform<K>(uy, hy);
        bspl_uniform4_deriv(ux, dx);
        bspl_uniform4_deriv(uy, dy);
        double gx = 0.0, gy = 0.0;
#pragma unroll
        for (int i = 0; i < k1; i++) {
            double rx = 0.0, ry = 0.0;
#pragma unroll
            for (int j = 0; j < k1; j++) {
                const double cij = pc->slot[(i * k1 + j) * pc->stride];
                rx = rx + cij * hy[j];
                ry = ry + cij * dy[j];
            }
            gx = gx + rx * dx[i];
            gy = gy + ry * hx[i];
        }
        sx = gx * uk->inv_h;
        sy = gy * uk->inv_h;
        return true;
    }
    double ax = x, ay = y;
    if (ax < t[k1 - 1]) ax = t[k1 - 1];
    if (ax > t[nk1]) ax = t[nk1];
    if (ay < t[k1 - 1]) ay = t[k1 - 1];
    if (ay > t[nk1]) ay = t[nk1];
    const int lx = spl_interval(t, n, k1, inv_h, ax), ly = spl_interval(t, n, k1, inv_h, ay);
    const int ulo = K + 1, uhi = n - K - 2;
    if (!(lx - K >= ulo && lx + K - 1 <= uhi && ly - K >= ulo && ly + K - 1 <= uhi)) return false;
    if (pc->key != (const double*)(c + ((lx - k1) * nk1 + (ly - k1)))) return false;
    const double tx0 = t[lx - 1], ty0 = t[ly - 1];
    const double ihx = 1.0 / (t[lx] - tx0), ihy = 1.0 / (t[ly] - ty0);
    const double ux = (ax - tx0) * ihx, uy = (ay - ty0) * ihy;
    double hx[k1], hy[k1], dx[k1], dy[k1];
    bspl_uniform<K>(ux, hx);
    bspl_uniform<K>(uy, hy);
    bspl_uniform4_deriv(ux, dx);
    bspl_uniform4_deriv(uy, dy);
    double gx = 0.0, gy = 0.0;
#pragma unroll
    for (int i = 0; i < k1; i++) {
        double rx = 0.0, ry = 0.0;  // row sums over y for this i
#pragma unroll
        for (int j = 0; j < k1; j++) {
            const double cij = pc->slot[(i * k1 + j) * pc->stride];
            rx = rx + cij * hy[j];
            ry = ry + cij * dy[j];
        }
        gx = gx + rx * dx[i];
        gy = gy + ry * hx[i];
    }
    sx = gx * ihx;
    sy = gy * ihy;
    return true;
}
