/*
 * oracle.c -- CPU restatement of optrace's sequential tracing hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the checker for the HIP path: only tests/, the
 * smoke test in __graft_entry__.py and bench.py's cpu_baseline leg may load it.  Nothing under
 * optrace_amd/ links, imports or calls it, and the product path has no CPU fallback.
 *
 * It is a scalar, one-ray-at-a-time restatement of the reference's whole-array NumPy code, in the
 * reference's operation order (left-to-right evaluation, no FMA contraction: build with
 * -ffp-contract=off), so that hit masks, section indices and counters are bit-identical and
 * positions/directions agree to rounding.  Every function cites the reference file:line it follows
 * (paths relative to the reference checkout, optrace v1.8.2).
 *
 * Pinning: tests/test_oracle_golden.py checks every function below against golden vectors produced
 * by the reference itself (tests/golden/generate_golden.py, run in the build container).
 *
 * Data layout: identical to include/optrace_amd.h (the header is shared for the plain-old-data
 * descriptors only; no code is shared with the HIP implementation).  All pointers are HOST memory.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/optrace_amd.h"

#define C_EPS OT_C_EPS
#define N_EPS OT_N_EPS_SURF

/* ------------------------------------------------------------------------------------------------
 * small vector helpers: optrace/tracer/misc.py
 * ---------------------------------------------------------------------------------------------- */

/* misc.rdot misc.py:94-118 */
static inline double rdot3(const double a[3], const double b[3]) {
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}

/* misc.cross misc.py:152-168 */
static inline void cross3(const double a[3], const double b[3], double n[3]) {
    n[0] = a[1] * b[2] - a[2] * b[1];
    n[1] = a[2] * b[0] - a[0] * b[2];
    n[2] = a[0] * b[1] - a[1] * b[0];
}

/* misc.normalize misc.py:136-150 (zero vectors become NaN) */
static inline void normalize3(double a[3]) {
    double l = sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
    a[0] = a[0] / l;
    a[1] = a[1] / l;
    a[2] = a[2] / l;
}

/* Surface._rotate_rc surface.py:427-434 */
static inline void rotate_rc(double x, double y, double alpha, double* xr, double* yr) {
    if (alpha != 0.0) {
        double ca = cos(alpha), sa = sin(alpha);
        *xr = x * ca - y * sa;
        *yr = x * sa + y * ca;
    } else {
        *xr = x;
        *yr = y;
    }
}

/* ------------------------------------------------------------------------------------------------
 * surfaces
 * ---------------------------------------------------------------------------------------------- */

static inline int surf_is_flat(const ot_surface* sf) { return sf->z_max == sf->z_min; } /* surface.py:47 */

/* FunctionSurface2D.mask function_surface_2d.py:158-191 with the callable restated as the bitmap the framework carries
 * (include/optrace_amd.h, OT_SURF_FLAG_MASK_TABLE): relative coordinates, rotated back by the surface's angle and
 * mirrored by its sign (2D), the radius (1D); the bit of the cell that holds the position. */
static int mask_func_cell(const ot_surface* sf, double dx, double dy) {
    int64_t n_t = sf->nknots, nc = n_t - OT_SPL_K - 1;
    int64_t at = (sf->kind == OT_SURF_DATA1D) ? 3 * n_t : n_t + nc * nc + 2 * (nc - 1) * nc;
    int64_t n = (int64_t)sf->tab[at], cell;
    const uint32_t* words = (const uint32_t*)(sf->tab + at + 1);
    if (sf->kind == OT_SURF_DATA1D) {
        int64_t i = (int64_t)(sqrt(dx * dx + dy * dy) * ((double)n / sf->r));
        cell = i < n ? i : n - 1;
    } else {
        double xr, yr;
        rotate_rc(dx, dy, -sf->angle, &xr, &yr);
        yr = sf->sign * yr;
        double scale = 0.5 * (double)n / sf->r;
        int64_t ix = (int64_t)floor((xr + sf->r) * scale), iy = (int64_t)floor((yr + sf->r) * scale);
        ix = ix < 0 ? 0 : (ix < n ? ix : n - 1);
        iy = iy < 0 ? 0 : (iy < n ? iy : n - 1);
        cell = iy * n + ix;
    }
    return (int)((words[cell >> 5] >> (cell & 31)) & 1u);
}

/* Surface.mask surface.py:235-245; RingSurface.mask ring_surface.py:123-133;
 * RectangularSurface.mask rectangular_surface.py:100-112; SlitSurface.mask slit_surface.py:89-102 */
int orc_mask1(const ot_surface* sf, double x, double y) {
    switch (sf->kind) {
        case OT_SURF_RING: {
            double dx = x - sf->pos[0], dy = y - sf->pos[1];
            double r2 = dx * dx + dy * dy;
            return (pow(sf->ri - N_EPS, 2.0) <= r2) && (r2 <= pow(sf->r + N_EPS, 2.0));
        }
        case OT_SURF_RECT:
        case OT_SURF_SLIT: {
            double xr, yr;
            rotate_rc(x - sf->pos[0], y - sf->pos[1], -sf->angle, &xr, &yr);
            double xs = -sf->dim[0] / 2, xe = sf->dim[0] / 2, ys = -sf->dim[1] / 2, ye = sf->dim[1] / 2;
            int outer = (xs - N_EPS <= xr) && (xr <= xe + N_EPS) && (ys - N_EPS <= yr) && (yr <= ye + N_EPS);
            if (sf->kind == OT_SURF_RECT) return outer;
            double xsi = -sf->dimi[0] / 2, xei = sf->dimi[0] / 2, ysi = -sf->dimi[1] / 2, yei = sf->dimi[1] / 2;
            int inside = (xsi + N_EPS <= xr) && (xr <= xei - N_EPS) && (ysi + N_EPS <= yr) && (yr <= yei - N_EPS);
            return outer && !inside;
        }
        default: { /* CIRCLE, CONIC, ASPHERE, TILTED, DATA1D, DATA2D */
            double dx = x - sf->pos[0], dy = y - sf->pos[1];
            int in = dx * dx + dy * dy <= pow(sf->r + N_EPS, 2.0);
            if (in && (sf->flags & OT_SURF_FLAG_MASK_TABLE)) in = mask_func_cell(sf, dx, dy);
            return in;
        }
    }
}

/* numpy.polyval (Horner over the full coefficient list) applied to AsphericSurface._np_coeff
 * aspheric_surface.py:104-113: [a_2n, 0, a_2n-2, 0, ..., a_2, 0, 0] */
static double asph_poly(const ot_surface* sf, double r) {
    double y = 0.0;
    for (int j = sf->ncoeff - 1; j >= 0; j--) {
        y = y * r + sf->coeff[j]; /* even power coefficient */
        y = y * r + 0.0;          /* odd power: zero coefficient */
    }
    y = y * r + 0.0; /* constant term */
    return y;
}

/* polyval(polyder(_np_coeff), r) aspheric_surface.py:79-80: derivative coefficients
 * [2n a_2n, 0, (2n-2) a_2n-2, 0, ..., 2 a_2, 0] */
static double asph_poly_deriv(const ot_surface* sf, double r) {
    double y = 0.0;
    for (int j = sf->ncoeff - 1; j >= 0; j--) {
        double c = sf->coeff[j] * (double)(2 * (j + 1)); /* np.polyder: coefficient * power */
        y = y * r + c;
        if (j > 0) y = y * r + 0.0;
    }
    y = y * r + 0.0; /* trailing zero (power 0 of the derivative) */
    return y;
}

/* ---- FITPACK B-spline evaluation as SciPy runs it for DataSurface1D/2D (data_surface_2d.py:72,101,126-135):
 * InterpolatedUnivariateSpline.__call__ -> splev.f / splder.f, RectBivariateSpline.__call__(grid=False) ->
 * bispeu.f / pardeu.f -> fpbisp.f, all on top of fpbspl.f.  FITPACK (netlib dierckx, as vendored by SciPy
 * 1.15.3) is not part of /root/reference; its published algorithms are restated here with 1-based indices
 * written as t[i - 1].  Table layout: include/optrace_amd.h (ot_surface.tab). */

/* fpbspl.f: the k+1 non-zero B-splines of degree k at x in knot interval l (t(l) <= x < t(l+1)) */
static void fpbspl(const double* t, int k, double x, int l, double* h) {
    double hh[8];
    h[0] = 1.0;
    for (int j = 1; j <= k; j++) {
        for (int i = 1; i <= j; i++) hh[i - 1] = h[i - 1];
        h[0] = 0.0;
        for (int i = 1; i <= j; i++) {
            int li = l + i, lj = li - j;
            if (t[li - 1] == t[lj - 1]) {
                h[i] = 0.0;
                continue;
            }
            double f = hh[i - 1] / (t[li - 1] - t[lj - 1]);
            h[i - 1] = h[i - 1] + f * (t[li - 1] - x);
            h[i] = f * (x - t[lj - 1]);
        }
    }
}

/* splev.f (der = 0: kk = k, coefficients c) / splder.f (der = 1: kk = k - 1, coefficients wrk); ext = 0 */
static double splev1(const double* t, int n, const double* c, int k, int kk, double arg) {
    int k1 = k + 1, k2 = k1 + 1, nk1 = n - k1;
    int l = k1, l1 = l + 1;
    while (!(arg >= t[l - 1] || l1 == k2)) {
        l1 = l;
        l = l - 1;
    }
    while (!(arg < t[l1 - 1] || l == nk1)) {
        l = l1;
        l1 = l + 1;
    }
    double h[8];
    fpbspl(t, kk, arg, l, h);
    double sp = 0.0;
    int ll = l - k1;
    for (int j = 1; j <= kk + 1; j++) {
        ll = ll + 1;
        sp = sp + c[ll - 1] * h[j - 1];
    }
    return sp;
}

/* fpbisp.f for a single point (mx = my = 1) */
static double fpbisp1(const double* tx, int nx, const double* ty, int ny, const double* c, int kx, int ky, double x,
                      double y) {
    double hx[8], hy[8];
    int kx1 = kx + 1, nkx1 = nx - kx1;
    double tb = tx[kx1 - 1], te = tx[nkx1];
    int l = kx1, l1 = l + 1;
    double arg = x;
    if (arg < tb) arg = tb;
    if (arg > te) arg = te;
    while (!(arg < tx[l1 - 1] || l == nkx1)) {
        l = l1;
        l1 = l + 1;
    }
    fpbspl(tx, kx, arg, l, hx);
    int lx = l - kx1;
    int ky1 = ky + 1, nky1 = ny - ky1;
    tb = ty[ky1 - 1];
    te = ty[nky1];
    l = ky1;
    l1 = l + 1;
    arg = y;
    if (arg < tb) arg = tb;
    if (arg > te) arg = te;
    while (!(arg < ty[l1 - 1] || l == nky1)) {
        l = l1;
        l1 = l + 1;
    }
    fpbspl(ty, ky, arg, l, hy);
    int ly = l - ky1;
    double sp = 0.0;
    int l1c = lx * nky1 + ly;
    for (int i1 = 1; i1 <= kx1; i1++) {
        int l2 = l1c;
        for (int j1 = 1; j1 <= ky1; j1++) {
            l2 = l2 + 1;
            sp = sp + c[l2 - 1] * hx[i1 - 1] * hy[j1 - 1];
        }
        l1c = l1c + nky1;
    }
    return sp;
}

/* DataSurface2D._call data_surface_2d.py:126-135 with optional derivative (dx, dy) / nu */
static double data_call(const ot_surface* sf, double x, double y, int dx, int dy) {
    const double* t = sf->tab;
    int n = sf->nknots, k = OT_SPL_K;
    if (sf->kind == OT_SURF_DATA1D) {
        double r = hypot(x, y);
        if (dx) return splev1(t, n, t + 2 * n, k, k - 1, r); /* nu = 1 */
        return splev1(t, n, t + n, k, k, r);
    }
    int nc = n - k - 1;
    const double* c = t + n;
    if (dx) return fpbisp1(t + 1, n - 2, t, n, c + nc * nc, k - 1, k, x, y);                 /* pardeu nux = 1 */
    if (dy) return fpbisp1(t, n, t + 1, n - 2, c + nc * nc + (nc - 1) * nc, k, k - 1, x, y); /* pardeu nuy = 1 */
    return fpbisp1(t, n, t, n, c, k, k, x, y);
}

/* Surface._values in coordinates relative to the centre:
 * ConicSurface._values conic_surface.py:57-68; AsphericSurface._asph aspheric_surface.py:51-65 through
 * FunctionSurface2D._values function_surface_2d.py:133-156 (1D branch, _sign = 1, _offset = 0) */
static double surf_values_rel(const ot_surface* sf, double x, double y) {
    if (sf->kind == OT_SURF_TILTED) { /* TiltedSurface._values tilted_surface.py:60-73 */
        double mx = -sf->normal[0] / sf->normal[2];
        double my = -sf->normal[1] / sf->normal[2];
        return x * mx + y * my;
    }
    if (sf->kind == OT_SURF_DATA1D) /* DataSurface2D._values data_surface_2d.py:137-151, rotational symmetry */
        return sf->sign * (data_call(sf, x, y, 0, 0) - sf->offset);
    if (sf->kind == OT_SURF_DATA2D) {
        double x_, y_;
        rotate_rc(x, y, -sf->angle, &x_, &y_);
        return sf->sign * (data_call(sf, x_, sf->sign * y_, 0, 0) - sf->offset);
    }
    double rho = 1 / sf->R, k = sf->k;
    if (sf->kind == OT_SURF_CONIC) {
        double r2 = x * x + y * y;
        return rho * r2 / (1 + sqrt(1 - (k + 1) * pow(rho, 2.0) * r2));
    }
    if (sf->kind == OT_SURF_ASPHERE) {
        double r = sqrt(x * x + y * y);
        double z = rho * (r * r) / (1 + sqrt(1 - (k + 1) * pow(rho, 2.0) * (r * r)));
        z += asph_poly(sf, r);
        return 1 * (z - 0.0);
    }
    return 0.0;
}

/* Surface.values surface.py:137-164 (absolute coordinates, radial edge continuation) */
double orc_values1(const ot_surface* sf, double x, double y) {
    if (surf_is_flat(sf)) return sf->z_max;
    if (orc_mask1(sf, x, y)) return sf->pos[2] + surf_values_rel(sf, x - sf->pos[0], y - sf->pos[1]);
    /* outside the mask the edge value is continued radially, surface.py:153-162 */
    double r = sf->r - N_EPS;
    if (sf->kind == OT_SURF_TILTED || sf->kind == OT_SURF_DATA2D) { /* rotational_symmetry == False */
        double phi = atan2(y - sf->pos[1], x - sf->pos[0]);
        return sf->pos[2] + surf_values_rel(sf, r * cos(phi), r * sin(phi));
    }
    return sf->pos[2] + surf_values_rel(sf, r, 0.0);
}

/* Surface.normals surface.py:247-256 (flat), ConicSurface.normals conic_surface.py:70-124,
 * FunctionSurface2D.normals function_surface_2d.py:202-251 with AsphericSurface._deriv
 * aspheric_surface.py:67-82 */
void orc_normals1(const ot_surface* sf, double x, double y, double n[3]) {
    n[0] = 0.0;
    n[1] = 0.0;
    n[2] = 1.0;
    if (sf->kind == OT_SURF_TILTED) { /* TiltedSurface.normals tilted_surface.py:75-89 */
        if (orc_mask1(sf, x, y))
            for (int c = 0; c < 3; c++) n[c] = sf->normal[c];
        return;
    }
    if (sf->kind == OT_SURF_DATA1D || sf->kind == OT_SURF_DATA2D) { /* DataSurface2D.normals :153-196 */
        if (!orc_mask1(sf, x, y)) return;
        double xm = x - sf->pos[0], ym = y - sf->pos[1];
        double nxn, nyn;
        if (sf->kind == OT_SURF_DATA2D) {
            double x_, y_, a, b;
            rotate_rc(xm, ym, -sf->angle, &x_, &y_);
            if (sf->flags & OT_SURF_FLAG_DERIV_UNROTATED) { /* function_surface_2d.py:235: deriv_func(xm, sign*ym) */
                x_ = xm;
                y_ = ym;
            }
            a = data_call(sf, x_, sf->sign * y_, 1, 0) * sf->sign;
            b = data_call(sf, x_, sf->sign * y_, 0, 1);
            rotate_rc(a, b, sf->angle, &nxn, &nyn);
        } else {
            double phi = atan2(ym, xm);
            double nr = sf->sign * data_call(sf, xm, ym, 1, 0);
            nxn = nr * cos(phi);
            nyn = nr * sin(phi);
        }
        n[0] = -nxn;
        n[1] = -nyn;
        n[2] = 1.0;
        normalize3(n);
        return;
    }
    if (sf->kind != OT_SURF_CONIC && sf->kind != OT_SURF_ASPHERE) return;
    if (sf->kind == OT_SURF_ASPHERE && surf_is_flat(sf)) return;

    int m = orc_mask1(sf, x, y);
    double x0 = sf->pos[0], y0 = sf->pos[1];
    double rho = 1 / sf->R;

    if (sf->kind == OT_SURF_CONIC) {
        if (sf->k == 0.0) {
            double n0 = -rho * (x - x0);
            double n1 = -rho * (y - y0);
            double n2 = sqrt(1 - pow(rho, 2.0) * ((x - x0) * (x - x0)) - pow(rho, 2.0) * ((y - y0) * (y - y0)));
            if (m) {
                n[0] = n0;
                n[1] = n1;
                n[2] = n2;
            }
            return;
        }
        double r = sqrt((x - x0) * (x - x0) + (y - y0) * (y - y0));
        double phi = atan2(y - y0, x - x0);
        double n_r = -rho * r / sqrt(1 - sf->k * pow(rho, 2.0) * (r * r));
        if (m) {
            n[0] = n_r * cos(phi);
            n[1] = n_r * sin(phi);
            n[2] = sqrt(1 - n_r * n_r);
        }
        return;
    }

    /* asphere */
    if (m) {
        double xm = x - x0, ym = y - y0;
        double phi = atan2(ym, xm);
        double rm = sqrt(xm * xm + ym * ym);
        double fr = rm * rho / sqrt(1 - (sf->k + 1) * pow(rho, 2.0) * (rm * rm));
        fr += asph_poly_deriv(sf, rm);
        double nr = 1 * fr;
        double nxn = nr * cos(phi), nyn = nr * sin(phi);
        n[0] = -nxn;
        n[1] = -nyn;
        n[2] = 1.0;
        normalize3(n);
    }
}

/* Surface._find_hit_handle_abnormal surface.py:436-479 */
static void handle_abnormal(const ot_surface* sf, const double p[3], const double s[3], double ph[3], int* is_hit) {
    double zs = orc_values1(sf, ph[0], ph[1]);
    int dev = fabs(ph[2] - zs) > C_EPS;
    int beh = p[2] > sf->z_max + N_EPS;
    int neg = ph[2] < p[2] - C_EPS;
    int bet = (neg || dev) && !beh;
    if (bet) {
        double tnm = (sf->z_max - p[2]) / s[2];
        for (int c = 0; c < 3; c++) ph[c] = p[c] + s[c] * tnm;
        *is_hit = 0;
    }
    if (beh) {
        for (int c = 0; c < 3; c++) ph[c] = p[c];
        *is_hit = 0;
    }
}

/* ConicSurface.find_hit conic_surface.py:126-203 */
static void find_hit_conic(const ot_surface* sf, const double p[3], const double s[3], double ph[3], int* is_hit) {
    double ox = p[0] - sf->pos[0], oy = p[1] - sf->pos[1], oz = p[2] - sf->pos[2];
    double sx = s[0], sy = s[1], sz = s[2];
    double k = sf->k, rho = 1 / sf->R;

    double A = (k != 0.0) ? 1 + k * (sz * sz) : 1.0;
    double B = sx * ox + sy * oy + sz * (oz * (k + 1) - 1 / rho);
    double C = oy * oy + ox * ox + oz * (oz * (k + 1) - 2 / rho);
    double D = sqrt(B * B - C * A);

    double t1 = (-B - D) / A;
    double t2 = (-B + D) / A;

    double z = p[2];
    double z1 = z + sz * t1;
    double z2 = z + sz * t2;
    double z_min = sf->z_min - N_EPS, z_max = sf->z_max + N_EPS;
    int c1 = (z_min <= z1) && (z1 <= z_max) && (z1 >= z);
    int c2 = (z_min <= z2) && (z2 <= z_max) && (z2 >= z) && (t2 < t1);
    double t = (c1 && !c2) ? t1 : t2;

    for (int c = 0; c < 3; c++) ph[c] = p[c] + s[c] * t;
    int hit = orc_mask1(sf, ph[0], ph[1]);

    if (A == 0 && B != 0) {
        t = -C / (2 * B);
        for (int c = 0; c < 3; c++) ph[c] = p[c] + s[c] * t;
        hit = orc_mask1(sf, ph[0], ph[1]);
    }

    int nh = !hit || !isfinite(D) || (A == 0 && B == 0) || (ph[2] < z_min) || (ph[2] > z_max);
    if (nh) {
        double tnh = (sf->z_max - p[2]) / s[2];
        for (int c = 0; c < 3; c++) ph[c] = p[c] + s[c] * tnh;
        hit = 0;
    }
    if (z > sf->z_max) {
        for (int c = 0; c < 3; c++) ph[c] = p[c];
        hit = 0;
    }
    *is_hit = hit;
}

/* Surface.find_hit surface.py:307-414: flat branch :319-327, Illinois regula falsi :329-414.
 * Returns 0, or -1 if the iteration timed out (reference raises TimeoutError, surface.py:403). */
static int find_hit_generic(const ot_surface* sf, const double p[3], const double s[3], double ph[3], int* is_hit,
                            int* ill);

/* dispatch: ConicSurface.find_hit, TiltedSurface.find_hit tilted_surface.py:91-123, Surface.find_hit */
int orc_find_hit1(const ot_surface* sf, const double p[3], const double s[3], double ph[3], int* is_hit, int* ill) {
    if (sf->kind == OT_SURF_TILTED && !surf_is_flat(sf)) {
        *ill = 0;
        double t_denom = rdot3(s, sf->normal);
        int nz = t_denom != 0;
        double d[3] = {sf->pos[0] - p[0], sf->pos[1] - p[1], sf->pos[2] - p[2]};
        double t = rdot3(d, sf->normal) / (nz ? t_denom : 1e-12);
        for (int c = 0; c < 3; c++) ph[c] = p[c] + s[c] * t;
        *is_hit = orc_mask1(sf, ph[0], ph[1]) && nz;
        int status = 0;
        if (!*is_hit) status = find_hit_generic(sf, p, s, ph, is_hit, ill); /* edge continued radially */
        handle_abnormal(sf, p, s, ph, is_hit);
        return status;
    }
    return find_hit_generic(sf, p, s, ph, is_hit, ill);
}

static int find_hit_generic(const ot_surface* sf, const double p[3], const double s[3], double ph[3], int* is_hit,
                            int* ill) {
    *ill = 0;
    if (sf->kind == OT_SURF_CONIC) {
        find_hit_conic(sf, p, s, ph, is_hit);
        return 0;
    }
    if (surf_is_flat(sf)) {
        double t = (sf->pos[2] - p[2]) / s[2];
        for (int c = 0; c < 3; c++) ph[c] = p[c] + s[c] * t;
        *is_hit = orc_mask1(sf, ph[0], ph[1]);
        handle_abnormal(sf, p, s, ph, is_hit);
        return 0;
    }

    double t1 = (sf->z_min - C_EPS / 10 - p[2]) / s[2];
    double t2 = (sf->z_max + C_EPS / 10 - p[2]) / s[2];
    if (t1 < 0) t1 = -C_EPS;

    double p1[3], p2[3];
    for (int c = 0; c < 3; c++) {
        p1[c] = p[c] + s[c] * t1;
        p2[c] = p[c] + s[c] * t2;
    }
    double f1 = p1[2] - orc_values1(sf, p1[0], p1[1]);
    double f2 = p2[2] - orc_values1(sf, p2[0], p2[1]);
    const double m = 0.5;

    int w = 1;
    if (!isfinite(t1) || !isfinite(t2)) w = 0;
    if ((t2 - t1) < C_EPS) w = 0;
    ph[0] = ph[1] = ph[2] = 0.0;
    if (!w) {
        for (int c = 0; c < 3; c++) ph[c] = p1[c];
    }
    *ill = f1 * f2 > 0;

    int it = 1, status = 0;
    while (w) {
        double ts = t1 - f1 / (f2 - f1) * (t2 - t1);
        double pl[3];
        for (int c = 0; c < 3; c++) pl[c] = p[c] + s[c] * ts;
        double fts = pl[2] - orc_values1(sf, pl[0], pl[1]);
        double prod = fts * f2;
        if (prod < 0) {
            t1 = t2;
            t2 = ts;
            f1 = f2;
            f2 = fts;
        } else if (prod > 0) {
            t2 = ts;
            f1 = m * f1;
            f2 = fts;
        } else if (prod == 0) {
            t1 = ts;
            t2 = ts;
            f1 = fts;
            f2 = fts;
        }
        if (fabs(t2 - t1) < C_EPS / 10) {
            for (int c = 0; c < 3; c++) ph[c] = pl[c];
            w = 0;
        }
        if (it == OT_MAX_HIT_ITER) { /* surface.py:403: raised inside the loop body */
            status = -1;
            break;
        }
        it++;
    }
    *is_hit = orc_mask1(sf, ph[0], ph[1]);
    handle_abnormal(sf, p, s, ph, is_hit);
    return status;
}

/* RingSurface.hurb_props ring_surface.py:88-121, SlitSurface.hurb_props slit_surface.py:65-87 */
void orc_hurb_props1(const ot_surface* sf, double x, double y, double* a_, double* b_, double b[3], int* inside) {
    if (sf->kind == OT_SURF_RING) {
        double dx = x - sf->pos[0], dy = y - sf->pos[1];
        double r = sqrt(dx * dx + dy * dy);
        double theta = atan2(dy, dx);
        double R = sf->ri;
        *inside = r < R;
        *b_ = R - r;
        *a_ = sqrt(*b_ * R);
        b[0] = cos(theta);
        b[1] = sin(theta);
        b[2] = 0.0;
    } else { /* SLIT */
        double x_, y_;
        rotate_rc(x - sf->pos[0], y - sf->pos[1], -sf->angle, &x_, &y_);
        *a_ = sf->dimi[1] / 2 - fabs(y_);
        *b_ = sf->dimi[0] / 2 - fabs(x_);
        *inside = (*a_ > 0) && (*b_ > 0);
        b[0] = cos(sf->angle);
        b[1] = sin(sf->angle);
        b[2] = 0.0;
    }
}

/* ------------------------------------------------------------------------------------------------
 * media and filters
 * ---------------------------------------------------------------------------------------------- */

/* numpy.interp (numpy/_core/src/multiarray/compiled_base.c arr_interp) as used by
 * Spectrum.__call__ "Data" spectrum.py:103-106 and observers.py:14-41 */
static double np_interp(double x, const double* xp, const double* fp, int64_t n, double left, double right) {
    if (isnan(x)) return x;
    if (x < xp[0]) return left;
    if (x > xp[n - 1]) return right;
    /* largest j with xp[j] <= x */
    int64_t lo = 0, hi = n - 1;
    while (hi - lo > 1) {
        int64_t mid = (lo + hi) / 2;
        if (xp[mid] <= x)
            lo = mid;
        else
            hi = mid;
    }
    int64_t j = (xp[hi] <= x) ? hi : lo;
    if (j == n - 1) return fp[j];
    if (xp[j] == x) return fp[j];
    double slope = (fp[j + 1] - fp[j]) / (xp[j + 1] - xp[j]);
    double res = slope * (x - xp[j]) + fp[j];
    if (isnan(res)) {
        res = slope * (x - xp[j + 1]) + fp[j + 1];
        if (isnan(res) && fp[j] == fp[j + 1]) res = fp[j];
    }
    return res;
}

/* RefractionIndex.__call__ refraction_index.py:62-169 for one wavelength (stored f32, upcast :70) */
double orc_refraction_index1(const ot_medium* md, const double* pool, float wl_f32) {
    double wl = (double)wl_f32;
    const double* c = md->c;
    double wl2 = (wl * 1e-3) * (wl * 1e-3);
    switch (md->model) {
        case OT_N_CONSTANT:
            return c[0];
        case OT_N_ABBE:
            return c[0] + c[1] / (wl2 - c[2]);
        case OT_N_CONRADY: {
            double l = wl * 1e-3;
            return c[0] + c[1] / l + c[2] / pow(l, 3.5);
        }
        case OT_N_CAUCHY:
            return c[0] + c[1] / wl2 + c[2] / (wl2 * wl2) + c[3] / pow(wl2, 3.0);
        case OT_N_SELLMEIER1:
            return sqrt(1 + c[0] * wl2 / (wl2 - c[1]) + c[2] * wl2 / (wl2 - c[3]) + c[4] * wl2 / (wl2 - c[5]));
        case OT_N_SELLMEIER2:
            return sqrt(1 + c[0] + c[1] * wl2 / (wl2 - pow(c[2], 2.0)) + c[3] / (wl2 - pow(c[4], 2.0)));
        case OT_N_SELLMEIER3:
            return sqrt(1 + c[0] * wl2 / (wl2 - c[1]) + c[2] * wl2 / (wl2 - c[3]) + c[4] * wl2 / (wl2 - c[5]) +
                        c[6] * wl2 / (wl2 - c[7]));
        case OT_N_SELLMEIER4:
            return sqrt(c[0] + c[1] * wl2 / (wl2 - c[2]) + c[3] * wl2 / (wl2 - c[4]));
        case OT_N_SELLMEIER5:
            return sqrt(1 + c[0] * wl2 / (wl2 - c[1]) + c[2] * wl2 / (wl2 - c[3]) + c[4] * wl2 / (wl2 - c[5]) +
                        c[6] * wl2 / (wl2 - c[7]) + c[8] * wl2 / (wl2 - c[9]));
        case OT_N_SCHOTT:
            return sqrt(c[0] + c[1] * wl2 + c[2] / wl2 + c[3] / (wl2 * wl2) + c[4] / pow(wl2, 3.0) +
                        c[5] / pow(wl2, 4.0));
        case OT_N_HERZBERGER: {
            double L = 1 / (wl2 - 0.028);
            return c[0] + c[1] * L + c[2] * (L * L) + c[3] * wl2 + c[4] * (wl2 * wl2) + c[5] * pow(wl2, 3.0);
        }
        case OT_N_HOO1:
            return sqrt(c[0] + c[1] / (wl2 - c[2]) - c[3] * wl2);
        case OT_N_HOO2:
            return sqrt(c[0] + c[1] * wl2 / (wl2 - c[2]) - c[3] * wl2);
        case OT_N_EXTENDED:
            return sqrt(c[0] + c[1] * wl2 + c[2] / wl2 + c[3] / (wl2 * wl2) + c[4] / pow(wl2, 3.0) +
                        c[5] / pow(wl2, 4.0) + c[6] / pow(wl2, 5.0) + c[7] / pow(wl2, 6.0));
        case OT_N_EXTENDED2:
            return sqrt(c[0] + c[1] * wl2 + c[2] / wl2 + c[3] / (wl2 * wl2) + c[4] / pow(wl2, 3.0) +
                        c[5] / pow(wl2, 4.0) + c[6] * (wl2 * wl2) + c[7] * pow(wl2, 3.0));
        case OT_N_EXTENDED3:
            return sqrt(c[0] + c[1] * wl2 + c[2] * (wl2 * wl2) + c[3] / wl2 + c[4] / (wl2 * wl2) +
                        c[5] / pow(wl2, 3.0) + c[6] * pow(wl2, 4.0) + c[7] * pow(wl2, 5.0) + c[8] / pow(wl2, 6.0));
        case OT_N_DATA: {
            const double* xp = pool + md->tab_off;
            return np_interp(wl, xp, xp + md->tab_len, md->tab_len, 0.0, 0.0);
        }
        case OT_N_LINES: {
            const double* xp = pool + md->tab_off;
            for (int j = 0; j < md->tab_len; j++)
                if (xp[j] == wl) return xp[md->tab_len + j];
            return NAN;
        }
    }
    return NAN;
}

/* Filter.__call__ filter.py:39 -> TransmissionSpectrum.__call__ transmission_spectrum.py:73-84
 * -> Spectrum.__call__ spectrum.py:81-120 */
double orc_filter1(const ot_filter* f, const double* pool, float wl_f32) {
    double wl = (double)wl_f32;
    double T;
    switch (f->type) {
        case OT_T_CONSTANT:
            T = f->val;
            break;
        case OT_T_DATA: {
            const double* xp = pool + f->tab_off;
            T = np_interp(wl, xp, xp + f->tab_len, f->tab_len, 0.0, 0.0);
            break;
        }
        case OT_T_RECTANGLE:
            T = (f->wl0 <= wl && wl <= f->wl1) ? f->val : 0.0;
            break;
        case OT_T_GAUSSIAN: {
            /* spectrum.py:113-115 uses the caller's array `wl` (float32 in the tracer), so NumPy
             * evaluates the whole expression in float32 */
            float d = wl_f32 - (float)f->mu;
            float q = -(d * d) / (float)(2 * pow(f->sig, 2.0));
            T = (double)((float)f->val * expf(q));
            break;
        }
        case OT_T_LINES: {
            const double* xp = pool + f->tab_off;
            T = NAN;
            for (int j = 0; j < f->tab_len; j++)
                if (xp[j] == wl) T = xp[f->tab_len + j];
            break;
        }
        default:
            T = NAN;
    }
    return f->inverse ? 1.0 - T : T;
}

/* ------------------------------------------------------------------------------------------------
 * tracing: Raytracer.trace / sub_trace raytracer.py:262-415
 * ---------------------------------------------------------------------------------------------- */

typedef struct {
    const ot_scene_desc* sc;
    const ot_rays* R;
    int64_t r; /* ray index */
    int64_t* msgs;
    float wl;
} rayctx;

#define P_(i, c) (cx->R->p[cx->r + cx->R->N * ((int64_t)(i) + (int64_t)cx->R->nt * (c))])
#define W_(i) (cx->R->w[cx->r + cx->R->N * (int64_t)(i)])
#define NS_(i) (cx->R->n[cx->r + cx->R->N * (int64_t)(i)])
#define POL_(i, c) (cx->R->pol[cx->r + cx->R->N * ((int64_t)(i) + (int64_t)cx->R->nt * (c))])
#define S_(c) (cx->R->s[cx->r + cx->R->N * (c)])
#define MSG_(info, sec) (cx->msgs[(info) * cx->R->nt + (sec)])

#define INV_SQRT2 (1 / sqrt(2.0)) /* 1/np.sqrt(2) raytracer.py:852,871 */

/* Raytracer.__compute_polarization raytracer.py:831-879 for one ray with hwh == True */
static void compute_polarization(rayctx* cx, const double s[3], const double s_[3], int i, double* A_ts, double* A_tp) {
    if (cx->sc->no_pol) {
        *A_ts = INV_SQRT2;
        *A_tp = INV_SQRT2;
        return;
    }
    int mask = (s[0] != s_[0]) || (s[1] != s_[1]) || (s[2] != s_[2]);
    double ps[3], pp[3], pp_[3], pol[3];
    cross3(s_, s, ps);
    normalize3(ps);
    cross3(ps, s, pp);
    for (int c = 0; c < 3; c++) pol[c] = (double)POL_(i, c);
    *A_ts = rdot3(ps, pol);
    *A_tp = rdot3(pp, pol);
    if (!mask) {
        *A_ts = INV_SQRT2;
        *A_tp = INV_SQRT2;
    }
    cross3(ps, s_, pp_);
    if (mask)
        for (int c = 0; c < 3; c++) POL_(i + 1, c) = (float)(ps[c] * *A_ts + pp_[c] * *A_tp);
}

/* Raytracer.__refraction raytracer.py:761-829 for one ray with hwh == True */
static void refraction(rayctx* cx, const ot_surface* sf, int i, double n1, double n2) {
    double n[3], s[3], s_[3];
    orc_normals1(sf, P_(i + 1, 0), P_(i + 1, 1), n);
    for (int c = 0; c < 3; c++) s[c] = S_(c);

    double ns = rdot3(n, s);
    double N = n1 / n2;
    double W = sqrt(1 - N * N * (1 - ns * ns));
    for (int c = 0; c < 3; c++) s_[c] = s[c] * N - n[c] * (N * ns - W);

    double A_ts, A_tp;
    compute_polarization(cx, s, s_, i, &A_ts, &A_tp);

    double cos_alpha = ns, cos_beta = W;
    double n1_cos_alpha = n1 * cos_alpha;
    double n2_cos_beta = n2 * cos_beta;
    double ts = 2 * n1_cos_alpha / (n1_cos_alpha + n2_cos_beta);
    double tp = 2 * n1_cos_alpha / (n2 * cos_alpha + n1 * cos_beta);
    double T = n2_cos_beta / n1_cos_alpha * ((A_ts * ts) * (A_ts * ts) + (A_tp * tp) * (A_tp * tp));

    if (!isfinite(W)) {
        T = 0;
        MSG_(OT_INFO_TIR, i) += 1;
    }
    W_(i + 1) = (float)((double)W_(i) * T);
    for (int c = 0; c < 3; c++) S_(c) = s_[c];
}

/* Raytracer.__refraction_ideal_lens raytracer.py:720-759 for one ray with hwh == True */
static void refraction_ideal(rayctx* cx, const ot_surface* sf, double D, int i) {
    double s0[3], s[3];
    for (int c = 0; c < 3; c++) s0[c] = S_(c);
    double f = 1000 / D;
    double fsz = f / s0[2];
    s[0] = s0[0] * fsz - (P_(i + 1, 0) - sf->pos[0]);
    s[1] = s0[1] * fsz - (P_(i + 1, 1) - sf->pos[1]);
    s[2] = f;
    normalize3(s);
    double sg = (f > 0) - (f < 0); /* np.sign(f) */
    for (int c = 0; c < 3; c++) s[c] = s[c] * sg;
    for (int c = 0; c < 3; c++) S_(c) = s[c];
    double A_ts, A_tp;
    compute_polarization(cx, s0, s, i, &A_ts, &A_tp);
}

/* Raytracer.__outline_intersection raytracer.py:666-718 for one ray of the mask `hw` */
static void outline_intersection(rayctx* cx, int i) {
    const double* o = cx->sc->outline;
    double x = P_(i + 1, 0), y = P_(i + 1, 1), z = P_(i + 1, 2);
    int inside = (o[0] < x) && (x < o[1]) && (o[2] < y) && (y < o[3]) && (o[4] < z) && (z < o[5]);
    if (inside) return;
    double t = NAN;
    for (int j = 0; j < 6; j++) {
        int c = j / 2;
        double T = (o[j] - P_(i, c)) / S_(c);
        if (T <= 0) T = NAN;
        if (!isnan(T) && (isnan(t) || T < t)) t = T; /* np.nanmin */
    }
    for (int c = 0; c < 3; c++) P_(i + 1, c) = P_(i, c) + S_(c) * t;
    W_(i + 1) = 0;
    MSG_(OT_INFO_OUTLINE_INTERSECTION, i) += 1;
}

/* Raytracer.__hurb raytracer.py:417-490 for one ray.  hwnh: ray has power and missed the aperture.
 * za, zb: standard-normal draws standing in for np.random.normal raytracer.py:468-469. */
static void hurb(rayctx* cx, const ot_surface* sf, int i, int hwnh, double za, double zb) {
    double a_, b_, b[3], a[3], s[3];
    int inside;
    orc_hurb_props1(sf, P_(i + 1, 0), P_(i + 1, 1), &a_, &b_, b, &inside);
    int hwnhi = hwnh && inside;
    a[0] = -b[1];
    a[1] = b[0];
    a[2] = b[2];
    for (int c = 0; c < 3; c++) s[c] = S_(c);

    double sa_ = rdot3(s, a), sb_ = rdot3(s, b);
    double cos_psi_a = sqrt(1 - sa_ * sa_);
    double cos_psi_b = sqrt(1 - sb_ * sb_);

    /* wl * 1e-9 is a float32 product in the reference (wl is the f32 wl_list slice) */
    float wlm = cx->wl * (float)1e-9;
    double k = 2 * M_PI * NS_(i) / (double)wlm;
    double hf = cx->sc->hurb_factor;
    double tan_sig_b = hf / (2 * b_ * cos_psi_b * 1e-3 * k);
    double tan_sig_a = hf / (2 * a_ * cos_psi_a * 1e-3 * k);
    double tan_tha = fabs(tan_sig_a) * za;
    double tan_thb = fabs(tan_sig_b) * zb;

    double sa[3], sb[3], sab[3];
    cross3(b, s, sa);
    normalize3(sa);
    cross3(s, sa, sb);
    for (int c = 0; c < 3; c++) sab[c] = s[c] + sa[c] * tan_tha + sb[c] * tan_thb;

    double s0[3] = {s[0], s[1], s[2]};
    if (hwnhi) {
        normalize3(sab);
        for (int c = 0; c < 3; c++) {
            s[c] = sab[c];
            S_(c) = s[c];
        }
    }
    if (s[2] < 0) { /* applies to every ray of the chunk, raytracer.py:484-486 */
        W_(i + 1) = 0;
        MSG_(OT_INFO_HURB_NEG_DIR, i + 1) += 1;
    }
    if (hwnhi) {
        double A_ts, A_tp;
        compute_polarization(cx, s0, s, i, &A_ts, &A_tp);
    }
}

/* one ray through all elements: sub_trace raytracer.py:297-397.  Returns 0 or -1 (hit timeout). */
static int trace_ray(rayctx* cx, const double* hurb_normals) {
    const ot_scene_desc* sc = cx->sc;
    const int pol = !sc->no_pol;
    int status = 0;
    int i = 0;
    int n1 = sc->n0;
    int hurb_idx = 0;
    const int64_t N = cx->R->N;
    cx->wl = cx->R->wl[cx->r];
    NS_(0) = orc_refraction_index1(&sc->media[n1], sc->table_pool, cx->wl);

    for (int en = 0; en < sc->n_elements; en++) {
        const ot_element* el = &sc->elements[en];
        for (int c = 0; c < 3; c++) P_(i + 1, c) = P_(i, c);
        W_(i + 1) = W_(i);
        int hw = W_(i) > 0;
        if (pol)
            for (int c = 0; c < 3; c++) POL_(i + 1, c) = POL_(i, c);

        double p[3], s[3], ph[3];
        int hit, ill;

        if (el->kind == OT_EL_LENS || el->kind == OT_EL_IDEAL_LENS) {
            const ot_surface* front = &sc->surfaces[el->front];
            for (int c = 0; c < 3; c++) {
                p[c] = P_(i, c);
                s[c] = S_(c);
            }
            if (orc_find_hit1(front, p, s, ph, &hit, &ill)) status = -1;
            if (hw)
                for (int c = 0; c < 3; c++) P_(i + 1, c) = ph[c];
            int hwh = hw && hit;
            if (hw && ill) MSG_(OT_INFO_ILL_COND, i + 1) += 1;
            if (hw && !hit) {
                W_(i + 1) = 0;
                MSG_(OT_INFO_ABSORB_MISSING, i + 1) += 1;
            }
            double n2_l = orc_refraction_index1(&sc->media[el->n_after], sc->table_pool, cx->wl);
            int hit_last;

            if (el->kind == OT_EL_LENS) {
                const ot_surface* back = &sc->surfaces[el->back];
                double n1_l = orc_refraction_index1(&sc->media[n1], sc->table_pool, cx->wl);
                double n_l = orc_refraction_index1(&sc->media[el->n_lens], sc->table_pool, cx->wl);
                if (hwh) refraction(cx, front, i, n1_l, n_l);
                if (hw && !hit) outline_intersection(cx, i);

                i += 1;
                for (int c = 0; c < 3; c++) P_(i + 1, c) = P_(i, c);
                W_(i + 1) = W_(i);
                NS_(i) = n_l;
                NS_(i + 1) = n2_l;
                if (pol)
                    for (int c = 0; c < 3; c++) POL_(i + 1, c) = POL_(i, c);

                hw = W_(i) > 0;
                for (int c = 0; c < 3; c++) {
                    p[c] = P_(i, c);
                    s[c] = S_(c);
                }
                if (orc_find_hit1(back, p, s, ph, &hit, &ill)) status = -1;
                if (hw)
                    for (int c = 0; c < 3; c++) P_(i + 1, c) = ph[c];
                if (hw && ill) MSG_(OT_INFO_ILL_COND, i + 1) += 1;
                if (hw && !hit) {
                    W_(i + 1) = 0;
                    for (int c = 0; c < 3; c++) P_(i + 1, c) = P_(i, c);
                    MSG_(OT_INFO_ABSORB_MISSING, i + 1) += 1;
                }
                if (hw && hit) refraction(cx, back, i, n_l, n2_l);
                hit_last = hit;
            } else {
                if (hwh) refraction_ideal(cx, front, el->D, i);
                NS_(i + 1) = n2_l;
                hit_last = hit;
            }
            if (hw && !hit_last) outline_intersection(cx, i);
            n1 = el->n_after;
        } else {
            const ot_surface* sf = &sc->surfaces[el->front];
            for (int c = 0; c < 3; c++) {
                p[c] = P_(i, c);
                s[c] = S_(c);
            }
            if (orc_find_hit1(sf, p, s, ph, &hit, &ill)) status = -1;
            if (hw)
                for (int c = 0; c < 3; c++) P_(i + 1, c) = ph[c];
            if (hw && ill) MSG_(OT_INFO_ILL_COND, i + 1) += 1;
            int hwh = hw && hit, hwnh = hw && !hit;

            if (el->kind == OT_EL_FILTER) {
                if (hwh) {
                    double T = orc_filter1(&sc->filters[el->filter], sc->table_pool, cx->wl);
                    W_(i + 1) = (float)((double)W_(i) * T);
                }
            } else {
                if (hwh) W_(i + 1) = 0;
                if (sc->use_hurb && el->hurb && en != sc->n_elements - 1) {
                    double za = hurb_normals ? hurb_normals[(2 * (int64_t)hurb_idx + 0) * N + cx->r] : 0.0;
                    double zb = hurb_normals ? hurb_normals[(2 * (int64_t)hurb_idx + 1) * N + cx->r] : 0.0;
                    hurb(cx, sf, i, hwnh, za, zb);
                    hurb_idx++;
                }
            }
            if (hwnh) outline_intersection(cx, i);
            NS_(i + 1) = NS_(i);
        }
        i += 1;
    }
    return status;
}

/* Raytracer.trace raytracer.py:262-415 with section 0 (p[:,0], s, w[:,0], wl, pol[:,0]) given.
 * msgs: int64[5*nt], ADDED to.  Returns 0, or -1 if any numeric hit search timed out. */
int orc_trace(const ot_scene_desc* sc, const ot_rays* R, const double* hurb_normals, int64_t* msgs) {
    int status = 0;
    for (int64_t r = 0; r < R->N; r++) {
        rayctx cx = {sc, R, r, msgs, 0.f};
        if (trace_ray(&cx, hurb_normals)) status = -1;
    }
    return status;
}

/* The same loop over rays split across host threads (rays are independent; the reference's own thread split is
 * ray_storage.py:147-171): used by bench.py's cpu_baseline leg only.  Counters are kept per thread and summed. */
int orc_trace_mt(const ot_scene_desc* sc, const ot_rays* R, const double* hurb_normals, int64_t* msgs, int n_threads) {
    int status = 0;
    const int n_msg = OT_N_INFOS * R->nt;
#pragma omp parallel num_threads(n_threads > 0 ? n_threads : 1) reduction(| : status)
    {
        int64_t* local = (int64_t*)calloc((size_t)n_msg, sizeof(int64_t));
#pragma omp for schedule(static)
        for (int64_t r = 0; r < R->N; r++) {
            rayctx cx = {sc, R, r, local, 0.f};
            if (trace_ray(&cx, hurb_normals)) status |= 1;
        }
#pragma omp critical
        for (int k = 0; k < n_msg; k++) msgs[k] += local[k];
        free(local);
    }
    return status ? -1 : 0;
}

/* ------------------------------------------------------------------------------------------------
 * array front ends of the leaf functions (for the known-answer tests)
 * ---------------------------------------------------------------------------------------------- */

int orc_surface_find_hit(const ot_surface* sf, int64_t n, const double* p, const double* s, double* p_hit,
                         uint8_t* is_hit, uint8_t* ill) {
    int status = 0;
    for (int64_t r = 0; r < n; r++) {
        double pp[3] = {p[r], p[r + n], p[r + 2 * n]}, ss[3] = {s[r], s[r + n], s[r + 2 * n]}, ph[3];
        int hit, il;
        if (orc_find_hit1(sf, pp, ss, ph, &hit, &il)) status = -1;
        for (int c = 0; c < 3; c++) p_hit[r + c * n] = ph[c];
        is_hit[r] = (uint8_t)hit;
        ill[r] = (uint8_t)il;
    }
    return status;
}

void orc_surface_normals(const ot_surface* sf, int64_t n, const double* x, const double* y, double* out) {
    for (int64_t r = 0; r < n; r++) {
        double nn[3];
        orc_normals1(sf, x[r], y[r], nn);
        for (int c = 0; c < 3; c++) out[r + c * n] = nn[c];
    }
}

void orc_surface_mask(const ot_surface* sf, int64_t n, const double* x, const double* y, uint8_t* out) {
    for (int64_t r = 0; r < n; r++) out[r] = (uint8_t)orc_mask1(sf, x[r], y[r]);
}

void orc_surface_values(const ot_surface* sf, int64_t n, const double* x, const double* y, double* out) {
    for (int64_t r = 0; r < n; r++) out[r] = orc_values1(sf, x[r], y[r]);
}

void orc_surface_hurb_props(const ot_surface* sf, int64_t n, const double* x, const double* y, double* a_,
                            double* b_, double* b, uint8_t* inside) {
    for (int64_t r = 0; r < n; r++) {
        double bb[3];
        int in;
        orc_hurb_props1(sf, x[r], y[r], &a_[r], &b_[r], bb, &in);
        for (int c = 0; c < 3; c++) b[r + c * n] = bb[c];
        inside[r] = (uint8_t)in;
    }
}

void orc_refraction_index(const ot_medium* md, const double* pool, int64_t n, const float* wl, double* out) {
    for (int64_t r = 0; r < n; r++) out[r] = orc_refraction_index1(md, pool, wl[r]);
}

void orc_filter(const ot_filter* f, const double* pool, int64_t n, const float* wl, double* out) {
    for (int64_t r = 0; r < n; r++) out[r] = orc_filter1(f, pool, wl[r]);
}

/* ------------------------------------------------------------------------------------------------
 * detector: Raytracer._hit_detector raytracer.py:881-1051
 * ---------------------------------------------------------------------------------------------- */

/* SphericalSurface.sphere_projection spherical_surface.py:36-97 (in place on one point) */
static void sphere_projection1(const ot_surface* sf, int projection, double p[3]) {
    if (projection == OT_PROJ_NONE || projection == OT_PROJ_ORTHOGRAPHIC) return;
    double x = p[0], y = p[1], z = p[2];
    double x0 = sf->pos[0], y0 = sf->pos[1], z0 = sf->pos[2];
    double zm = z0 + sf->R;
    double sgnR = (sf->R > 0) - (sf->R < 0);
    if (projection == OT_PROJ_EQUIDISTANT) {
        double r = sqrt((x - x0) * (x - x0) + (y - y0) * (y - y0));
        double theta = -sgnR * atan(r / (z - zm));
        double phi = atan2(y - y0, x - x0);
        p[0] = theta * cos(phi);
        p[1] = theta * sin(phi);
    } else if (projection == OT_PROJ_STEREOGRAPHIC) {
        double r = sqrt((x - x0) * (x - x0) + (y - y0) * (y - y0));
        double theta = M_PI / 2 - atan(r / (z - zm));
        double phi = atan2(y - y0, x - x0);
        r = -2 * sgnR * tan(M_PI / 4 - theta / 2);
        p[0] = r * cos(phi);
        p[1] = r * sin(phi);
    } else if (projection == OT_PROJ_EQUAL_AREA) {
        double x_ = (x - x0) / fabs(sf->R);
        double y_ = (y - y0) / fabs(sf->R);
        double z_ = (z - zm) / sf->R;
        p[0] = sqrt(2 / (1 - z_)) * x_;
        p[1] = sqrt(2 / (1 - z_)) * y_;
    }
}

/* direction of section k re-derived from stored positions, RayStorage.rays_by_mask
 * ray_storage.py:274-279 (normalised; the last section gives 0/0 = NaN) */
static void section_dir(const ot_rays* R, int64_t r, int k, double s[3]) {
    int k1 = (k < R->nt - 1) ? k + 1 : k;
    for (int c = 0; c < 3; c++)
        s[c] = R->p[r + R->N * ((int64_t)k1 + (int64_t)R->nt * c)] - R->p[r + R->N * ((int64_t)k + (int64_t)R->nt * c)];
    normalize3(s);
}

/* Raytracer._hit_detector raytracer.py:922-1051, per ray over [first, first+count).
 * ph (count,3) F-order, hw (count): weight of a valid hit, 0 otherwise; extent4: running
 * xmin,xmax,ymin,ymax over valid hits (after projection); ill_count ADDED to. */
int orc_detector_hits(const ot_rays* R, int64_t first, int64_t count, const ot_surface* det, int32_t projection,
                      double* ph_out, float* hw_out, double* extent4, int64_t* ill_count) {
    const int nt = R->nt;
    const int64_t N = R->N;
    int status = 0;
    double ext4, ext5;
    { /* Surface.extent surface.py:113-120 / RectangularSurface.extent: only z is needed */
        ext4 = det->z_min;
        ext5 = det->z_max;
    }
    for (int64_t q = 0; q < count; q++) {
        int64_t r = first + q;
        for (int c = 0; c < 3; c++) ph_out[q + c * count] = 0.0;
        hw_out[q] = 0.f;

        int all_b = 1, all_nb = 1, first_ge = -1;
        for (int j = 0; j < nt; j++) {
            double z = R->p[r + N * ((int64_t)j + (int64_t)nt * 2)];
            int bmin = z >= ext4, bmax = z >= ext5;
            if (!(bmin && bmax)) all_b = 0;
            if (!(!bmin && !bmax)) all_nb = 0;
            if (bmin && first_ge < 0) first_ge = j;
        }
        if (all_b || all_nb) continue; /* no_start | no_reach raytracer.py:933-935 */
        int k = (first_ge < 0 ? 0 : first_ge) - 1; /* np.argmax(bh_zmin) - 1, clipped at 0 */
        if (k < 0) k = 0;

        double p[3], s[3], ph[3] = {0, 0, 0};
        int ish = 0, any_ill = 0;
        for (int c = 0; c < 3; c++) p[c] = R->p[r + N * ((int64_t)k + (int64_t)nt * c)];
        section_dir(R, r, k, s);
        float w = R->w[r + N * (int64_t)k];

        for (;;) {
            k += 1;
            if (k >= nt) { /* raytracer.py:970-978 */
                w = 0;
                break;
            }
            int ill;
            if (orc_find_hit1(det, p, s, ph, &ish, &ill)) status = -1;
            any_ill |= ill;
            double p2z = R->p[r + N * ((int64_t)k + (int64_t)nt * 2)];
            int again = ph[2] > p2z + C_EPS; /* raytracer.py:985 */
            if (!again) break;
            for (int c = 0; c < 3; c++) p[c] = R->p[r + N * ((int64_t)k + (int64_t)nt * c)];
            section_dir(R, r, k, s);
            w = R->w[r + N * (int64_t)k];
        }
        if (any_ill) *ill_count += 1;
        if (ish && w > 0) {
            sphere_projection1(det, projection, ph);
            for (int c = 0; c < 3; c++) ph_out[q + c * count] = ph[c];
            hw_out[q] = w;
            if (extent4) {
                if (ph[0] < extent4[0]) extent4[0] = ph[0];
                if (ph[0] > extent4[1]) extent4[1] = ph[0];
                if (ph[1] < extent4[2]) extent4[2] = ph[1];
                if (ph[1] > extent4[3]) extent4[3] = ph[1];
            }
        }
    }
    return status;
}

/* ------------------------------------------------------------------------------------------------
 * rendering: RenderImage.render render_image.py:396-418
 * ---------------------------------------------------------------------------------------------- */

/* color.x/y/z_observer observers.py:14-41.  `table` = 471 rows x 4 columns (wl, x, y, z), row-major
 * (CIE 1931 2-degree standard observer, 360..830 nm in 1 nm steps). */
void orc_observers(const double* table, int64_t nrows, int64_t n, const float* wl, double* xyz) {
    double* xp = (double*)malloc(sizeof(double) * nrows * 4);
    for (int64_t j = 0; j < nrows; j++)
        for (int c = 0; c < 4; c++) xp[c * nrows + j] = table[j * 4 + c];
    for (int64_t r = 0; r < n; r++)
        for (int c = 0; c < 3; c++)
            xyz[r + c * n] = np_interp((double)wl[r], xp, xp + (c + 1) * nrows, nrows, 0.0, 0.0);
    free(xp);
}

/* misc.binning_indices_2d misc.py:59-91 for one position */
static inline void binning1(double x, double y, const double extent[4], int Nx, int Ny, int* xi, int* yi, int* outside) {
    double sx = extent[1] - extent[0], sy = extent[3] - extent[2];
    int32_t ix = (int32_t)floor(Nx / sx * (x - extent[0]));
    int32_t iy = (int32_t)floor(Ny / sy * (y - extent[2]));
    if (y == extent[3]) iy = Ny - 1;
    if (x == extent[1]) ix = Nx - 1;
    *outside = (ix < 0) || (iy < 0) || (iy >= Ny) || (ix >= Nx);
    if (*outside) {
        ix = 0;
        iy = 0;
    }
    *xi = ix;
    *yi = iy;
}

void orc_binning_indices_2d(int64_t n, const double* x, const double* y, const float* w, int Nx, int Ny,
                            const double extent[4], int32_t* xi, int32_t* yi, float* wm) {
    for (int64_t r = 0; r < n; r++) {
        int ix, iy, out;
        binning1(x[r], y[r], extent, Nx, Ny, &ix, &iy, &out);
        xi[r] = ix;
        yi[r] = iy;
        wm[r] = out ? 0.f : w[r];
    }
}

/* RenderImage.render render_image.py:396-418: hist (Ny, Nx, 4) f64 += w * [xbar, ybar, zbar, 1],
 * sequentially in ray order like np.add.at */
void orc_render_accumulate(const double* table, int64_t nrows, int64_t n, const double* px, const double* py,
                           const float* w, const float* wl, const double extent[4], int Nx, int Ny, double* hist) {
    double* xp = (double*)malloc(sizeof(double) * nrows * 4);
    for (int64_t j = 0; j < nrows; j++)
        for (int c = 0; c < 4; c++) xp[c * nrows + j] = table[j * 4 + c];
    for (int64_t r = 0; r < n; r++) {
        int ix, iy, out;
        binning1(px[r], py[r], extent, Nx, Ny, &ix, &iy, &out);
        double wm = out ? 0.0 : (double)w[r];
        double* h = hist + ((int64_t)iy * Nx + ix) * 4;
        for (int c = 0; c < 3; c++) h[c] += np_interp((double)wl[r], xp, xp + (c + 1) * nrows, nrows, 0.0, 0.0) * wm;
        h[3] += 1.0 * wm;
    }
    free(xp);
}
