// Per-ray device functions of the sequential tracer (gfx950, wave64).
//
// Everything here is written against the NumPy behaviour of the reference (file:line cited per function),
// NOT translated from it: one ray per lane, state in VGPRs, surface constants as wave-uniform scalars,
// IEEE f64 +,-,*,/ and sqrt in the reference's evaluation order (compile with -ffp-contract=off) so that hit
// masks and counters come out bit-identical.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

#include "ot_scene.hpp"

#define OT_DEV __device__ __forceinline__
#define OT_HD __host__ __device__ __forceinline__  // also used by the host when it tabulates discrete spectra

// Scene tables are read through the CONSTANT address space: with a wave-uniform index the backend then emits
// scalar loads (s_load_dwordx*) into SGPRs instead of per-lane flat loads into VGPRs.
#define OT_CONST __attribute__((address_space(4)))
template <class T>
OT_DEV const OT_CONST T* as_const(const T* p) {
    return (const OT_CONST T*)p;
}

struct V3 {
    double x, y, z;
};

OT_DEV double dot3(const V3& a, const V3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }  // misc.py:94

OT_DEV V3 cross3(const V3& a, const V3& b) {  // misc.py:152
    V3 n;
    n.x = a.y * b.z - a.z * b.y;
    n.y = a.z * b.x - a.x * b.z;
    n.z = a.x * b.y - a.y * b.x;
    return n;
}

// sqrt for the three roots per ray-surface of the tracing loop (discriminant, normal z, refraction W).
// Same rsq + Goldschmidt/Newton sequence the compiler emits for an IEEE f64 sqrt, minus its 2^+-256 range
// scaling (5 of 17 instructions): bit-identical for 2^-767 <= x < 2^1023, and x = 0, inf, NaN, x < 0 behave
// as sqrt does.  Arguments here are mm^2-scale or O(1) quantities.
OT_DEV double ot_sqrt(double x) {
    double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = 0.5 * y;
    double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    return (x == 0.0 || x == __builtin_inf()) ? x : g;
}

// n / d for the divisions of the tracing loop: reciprocal seed, two Newton steps, quotient, residual correction --
// the arithmetic core of the compiler's own f64 division, without its v_div_scale / v_div_fmas / v_div_fixup
// wrapping for extreme exponents and infinities (8 instead of 13 instructions; ot_rcp3 shares the reciprocal
// between quotients with one denominator).  Bit-identical to `/` for finite operands with exponents inside
// +-500 -- checked on the device by the library's own harness, ot_selftest_arith (csrc/ot_selftest.hpp), which
// tests/test_gpu_arith_exact.py drives with 16 operation x operand-class cases of 1.3e8 operand sets each -- which is where
// millimetre geometry and refractive indices live; a zero denominator gives NaN instead of +-inf, and every caller treats both as "no hit".
OT_DEV double ot_rcp3(double d) {
    double r = __builtin_amdgcn_rcp(d);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    return __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
}

OT_DEV double ot_div_r(double n, double d, double r) {  // r = ot_rcp3(d)
    double q = n * r;
    return __builtin_fma(__builtin_fma(-d, q, n), r, q);
}

// 1 / sqrt(x) for the generator's unit vectors (nothing the hit masks see): reciprocal-square-root seed and two Newton steps
// with fused multiply-adds, ~1 ulp -- 9 instructions where ot_rcp3(ot_sqrt(x)) takes 17.
OT_DEV double ot_rsqrt(double x) {
    double y = __builtin_amdgcn_rsq(x);
    double e = __builtin_fma(-(x * y), y, 1.0);
    y = __builtin_fma(0.5 * y, e, y);
    e = __builtin_fma(-(x * y), y, 1.0);
    return __builtin_fma(0.5 * y, e, y);
}

OT_DEV V3 normalize3(const V3& a) {  // misc.py:136 (zero vectors -> NaN); sqrt and `/` through their cores above:
    const double l = ot_sqrt(a.x * a.x + a.y * a.y + a.z * a.z);  // the same bits, 27 instead of 56 instructions
    const double il = ot_rcp3(l);
    V3 r = {ot_div_r(a.x, l, il), ot_div_r(a.y, l, il), ot_div_r(a.z, l, il)};
    return r;
}

OT_HD double ot_div(double n, double d) {
#if defined(__HIP_DEVICE_COMPILE__)
    return ot_div_r(n, d, ot_rcp3(d));
#else
    return n / d;  // host evaluation of the same tables (discrete-spectrum rows, edge values)
#endif
}

OT_DEV V3 along(const V3& p, const V3& s, double t) {
    V3 r = {p.x + s.x * t, p.y + s.y * t, p.z + s.z * t};
    return r;
}

#include "ot_spline.hpp"

// Hit levels: which surface kinds a piece of device code has to handle.  Kernels are compiled per level so that a scene
// only carries the code (registers, LDS, instruction cache) of what it contains:
//   OT_HIT_CLOSED    flat and conic surfaces: closed-form hit, no loop
//   OT_HIT_ILLINOIS  + aspheres and tilted surfaces: the numeric hit search (surface.py:329-414) on a closed-form sag
//   OT_HIT_SPLINE    + data / function surfaces: spline tables, coefficient patch cache, mask_func bitmaps
#define OT_HIT_CLOSED 0
#define OT_HIT_ILLINOIS 1
#define OT_HIT_SPLINE 2

// ---- spline surfaces: DataSurface2D._call / ._values / .normals data_surface_2d.py:126-196 -----------------
// table layout (include/optrace_amd.h): DATA1D t[n] | c[n] | dc[n];  DATA2D t[n] | c[(n-5)^2] | cx[(n-6)(n-5)] | cy
template <class SF>
OT_HD double data_values_rel(SF& sf, double x, double y, PatchCache* pc = nullptr) {
    const double* t = sf.tab;
    const int n = sf.nk;
    double v;
    // (ku_h == 0: no equidistant part -- an empty index range, so every point takes the table path; the pointer itself
    // stays a compile-time constant, which keeps the struct in registers)
    const UniformKnots ukv = {sf.ku_t0, sf.ku_h, sf.ku_inv_h, sf.ku_lo, sf.ku_hi,
                              (sf.ku_h > 0.0) ? OT_SPL_K + 1 : (1 << 30), n - OT_SPL_K - 2};
    const UniformKnots* const uk = &ukv;
    if (sf.kind == OT_SURF_DATA1D) {
        v = spl1_eval<OT_SPL_K>(t, n, t + n, sf.inv_h, hypot(x, y), uk, pc);
    } else {
        double xr = x, yr = y;
        if (sf.rot) {  // _rotate_rc(x, y, -angle)
            xr = x * sf.cna - y * sf.sna;
            yr = x * sf.sna + y * sf.cna;
        }
        v = spl2_eval<OT_SPL_K, OT_SPL_K>(t, n, t, n, t + n, sf.inv_h, xr, sf.sgn * yr, pc, uk);
    }
    return sf.sgn * (v - sf.offs);
}

// -(dz/dx), -(dz/dy), 1 normalised; x, y relative to the centre, inside the mask
template <class SF>
OT_HD void data_gradient(SF& sf, double x, double y, double& gx, double& gy, PatchCache* pc = nullptr) {
    const double* t = sf.tab;
    const int n = sf.nk;
    // (ku_h == 0: no equidistant part -- an empty index range, so every point takes the table path; the pointer itself
    // stays a compile-time constant, which keeps the struct in registers)
    const UniformKnots ukv = {sf.ku_t0, sf.ku_h, sf.ku_inv_h, sf.ku_lo, sf.ku_hi,
                              (sf.ku_h > 0.0) ? OT_SPL_K + 1 : (1 << 30), n - OT_SPL_K - 2};
    const UniformKnots* const uk = &ukv;
    if (sf.kind == OT_SURF_DATA1D) {
        const double r = hypot(x, y);
        double d1;
        if (!spl1_grad_cached(t + n, r, uk, pc, d1)) d1 = spl1_eval<OT_SPL_K - 1>(t, n, t + 2 * n, sf.inv_h, r, uk);
        const double nr = sf.sgn * d1;
        const double rr = sqrt(x * x + y * y);
        gx = nr * ((rr > 0.0) ? x / rr : 1.0);  // cos(arctan2(y, x))
        gy = nr * ((rr > 0.0) ? y / rr : 0.0);
        return;
    }
    double xr = x, yr = y;
    if (sf.rot && !sf.deriv_unrot) {  // deriv_unrot: OT_SURF_FLAG_DERIV_UNROTATED, function_surface_2d.py:235
        xr = x * sf.cna - y * sf.sna;
        yr = x * sf.sna + y * sf.cna;
    }
    const int nc = (n - OT_SPL_K - 1), ncx = (n - OT_SPL_K - 2) * nc;
    const double* c = t + n;
    double nxn, nyn;
    if (spl2_grad_cached(t, n, c, sf.inv_h, xr, sf.sgn * yr, pc, nxn, nyn, uk)) {  // from the patch of the hit search
        nxn = nxn * sf.sgn;
    } else {
        nxn = spl2_eval<OT_SPL_K - 1, OT_SPL_K>(t + 1, n - 2, t, n, c + nc * nc, sf.inv_h, xr, sf.sgn * yr) * sf.sgn;
        nyn = spl2_eval<OT_SPL_K, OT_SPL_K - 1>(t, n, t + 1, n - 2, c + nc * nc + ncx, sf.inv_h, xr, sf.sgn * yr);
    }
    if (sf.rot) {  // _rotate_rc(nxn, nyn, +angle)
        gx = nxn * sf.cpa - nyn * sf.spa;
        gy = nxn * sf.spa + nyn * sf.cpa;
    } else {
        gx = nxn;
        gy = nyn;
    }
}

// mask_func of a function surface as a bitmap (layout: include/optrace_amd.h, OT_SURF_FLAG_MASK_TABLE): the cell of
// (dx, dy) relative to the centre, taken back into the function's frame like the values (function_surface_2d.py:170-181)
template <class SF>
OT_HD bool mask_table(SF& sf, double dx, double dy) {
    const uint32_t* words = (const uint32_t*)(sf.tab + sf.mask_off);
    const int n = sf.mask_n;
    int64_t cell;
    if (sf.kind == OT_SURF_DATA1D) {
        const int i = (int)(sqrt(dx * dx + dy * dy) * sf.mask_scale);
        cell = i < n ? i : n - 1;
    } else {
        double xr = dx, yr = dy;
        if (sf.rot) {
            xr = dx * sf.cna - dy * sf.sna;
            yr = dx * sf.sna + dy * sf.cna;
        }
        yr = sf.sgn * yr;
        int ix = (int)floor((xr + sf.mask_r) * sf.mask_scale), iy = (int)floor((yr + sf.mask_r) * sf.mask_scale);
        ix = ix < 0 ? 0 : (ix < n ? ix : n - 1);
        iy = iy < 0 ? 0 : (iy < n ? iy : n - 1);
        cell = (int64_t)iy * n + ix;
    }
    return (words[cell >> 5] >> (uint32_t)(cell & 31)) & 1u;
}

// ---- masks: surface.py:235, ring_surface.py:123, rectangular_surface.py:100, slit_surface.py:89 ----------
// TABLES = false: call sites that cannot meet a spline surface (conic hit, kernels without numeric surfaces)
template <bool TABLES = true, class SF>
OT_DEV bool surf_mask(SF& sf, double x, double y) {
    if (sf.kind == OT_SURF_RECT || sf.kind == OT_SURF_SLIT) {
        double dx = x - sf.px, dy = y - sf.py;
        double xr = dx, yr = dy;
        if (sf.rot) {
            xr = dx * sf.cna - dy * sf.sna;
            yr = dx * sf.sna + dy * sf.cna;
        }
        bool outer = (sf.ox_lo <= xr) && (xr <= sf.ox_hi) && (sf.oy_lo <= yr) && (yr <= sf.oy_hi);
        if (sf.kind == OT_SURF_RECT) return outer;
        bool inner = (sf.ix_lo <= xr) && (xr <= sf.ix_hi) && (sf.iy_lo <= yr) && (yr <= sf.iy_hi);
        return outer && !inner;
    }
    double dx = x - sf.px, dy = y - sf.py;
    double r2 = dx * dx + dy * dy;
    bool in = r2 <= sf.r_eps2;
    if (sf.kind == OT_SURF_RING) in = in && (sf.ri_eps2 <= r2);
    if (TABLES && (sf.kind == OT_SURF_DATA1D || sf.kind == OT_SURF_DATA2D) && sf.mask_n) {
        if (in) in = mask_table(sf, dx, dy);
    }
    return in;
}

// numpy.polyval over AsphericSurface._np_coeff (aspheric_surface.py:104-113): Horner including the zero
// odd-order coefficients, i.e. y = (y*r + a)*r + 0 per even coefficient
template <class SF>
OT_DEV double asph_poly(SF& sf, double r) {
    double y = 0.0;
    for (int j = sf.ncoeff - 1; j >= 0; j--) {
        y = y * r + sf.coeff[j];
        y = y * r + 0.0;
    }
    y = y * r + 0.0;
    return y;
}

template <class SF>
OT_DEV double asph_poly_deriv(SF& sf, double r) {  // polyval(polyder(..)) aspheric_surface.py:79
    double y = 0.0;
    for (int j = sf.ncoeff - 1; j >= 0; j--) {
        y = y * r + sf.dcoeff[j];
        y = y * r + 0.0;
    }
    return y;
}

// Surface._values relative to the centre: conic_surface.py:57, aspheric_surface.py:51
template <int LEVEL = OT_HIT_SPLINE, class SF>
OT_DEV double surf_values_rel(SF& sf, double x, double y, PatchCache* pc = nullptr) {
    if (sf.kind == OT_SURF_TILTED) return x * sf.mx + y * sf.my;  // tilted_surface.py:60-73
    if (LEVEL >= OT_HIT_SPLINE && (sf.kind == OT_SURF_DATA1D || sf.kind == OT_SURF_DATA2D)) return data_values_rel(sf, x, y, pc);
    if (sf.kind == OT_SURF_CONIC) {
        double r2 = x * x + y * y;
        return ot_div(sf.rho * r2, 1 + sqrt(1 - sf.k1rho2 * r2));
    }
    double r = sqrt(x * x + y * y);
    double rr = r * r;
    double z = ot_div(sf.rho * rr, 1 + sqrt(1 - sf.k1rho2 * rr));
    z += asph_poly(sf, r);
    return z;
}

// Surface.values surface.py:137-164
template <int LEVEL = OT_HIT_SPLINE, class SF>
OT_DEV double surf_values(SF& sf, double x, double y, PatchCache* pc = nullptr) {
    if (sf.flat) return sf.z_max;
    if (surf_mask<(LEVEL >= OT_HIT_SPLINE)>(sf, x, y)) return sf.pz + surf_values_rel<LEVEL>(sf, x - sf.px, y - sf.py, pc);
    if (sf.kind == OT_SURF_TILTED || (LEVEL >= OT_HIT_SPLINE && sf.kind == OT_SURF_DATA2D)) {
        // no rotational symmetry: edge value along the direction of (x, y), surface.py:156-159;
        // cos / sin of arctan2 formed as dx / rr, dy / rr like in surf_normal
        double dx = x - sf.px, dy = y - sf.py;
        double rr = sqrt(dx * dx + dy * dy);
        double c = (rr > 0.0) ? dx / rr : 1.0, sn = (rr > 0.0) ? dy / rr : 0.0;
        return sf.pz + surf_values_rel<LEVEL>(sf, sf.r_edge * c, sf.r_edge * sn, pc);
    }
    return sf.edge_val;
}

// Surface.normals surface.py:247, ConicSurface.normals conic_surface.py:70-124,
// FunctionSurface2D.normals (1D branch) function_surface_2d.py:216-251 + AsphericSurface._deriv :67-82.
// cos(atan2(dy,dx)) and sin(atan2(dy,dx)) are formed as dx/r, dy/r: same value to 1-2 ulp without three
// transcendental calls per ray (normals never feed a mask directly; tolerance 1e-6).
template <bool INSIDE = false, int LEVEL = OT_HIT_SPLINE, class SF>
OT_DEV V3 surf_normal(SF& sf, double x, double y, PatchCache* pc = nullptr) {
    constexpr bool TABLES = LEVEL >= OT_HIT_SPLINE;
    V3 n = {0.0, 0.0, 1.0};
    if (sf.kind < OT_SURF_CONIC) return n;
    if (LEVEL == OT_HIT_CLOSED && sf.kind != OT_SURF_CONIC) return n;  // kernel variant for scenes of flat and conic surfaces
    if (sf.kind == OT_SURF_TILTED) {  // tilted_surface.py:75-89: constant, also when the plane happens to be flat
        if (!INSIDE && !surf_mask<TABLES>(sf, x, y)) return n;
        V3 m = {sf.nx, sf.ny, sf.nz};
        return m;
    }
    if (sf.flat) return n;
    if (!INSIDE && !surf_mask<TABLES>(sf, x, y)) return n;  // INSIDE: caller already knows mask(x, y) is true
    double dx = x - sf.px, dy = y - sf.py;
    if (sf.kind == OT_SURF_DATA1D || sf.kind == OT_SURF_DATA2D) {  // data_surface_2d.py:153-196
        if (LEVEL < OT_HIT_SPLINE) return n;  // (unreachable: such scenes run the spline level)
        double gx, gy;
        data_gradient(sf, dx, dy, gx, gy, pc);
        V3 m = {-gx, -gy, 1.0};
        return normalize3(m);
    }
    if (sf.kind == OT_SURF_CONIC) {
        if (sf.k == 0.0) {
            n.x = sf.nrho * dx;
            n.y = sf.nrho * dy;
            n.z = ot_sqrt(1 - sf.rho2 * (dx * dx) - sf.rho2 * (dy * dy));
            return n;
        }
        // n_r = -rho r / sqrt(1 - k rho^2 r^2), (n_x, n_y) = n_r (cos phi, sin phi), n_z = sqrt(1 - n_r^2)
        // (conic_surface.py:101-118) with cos phi = dx / r, sin phi = dy / r: the radius cancels, so
        // n_x = -rho dx / sqrt(1 - k rho^2 r^2), likewise n_y, and n_r^2 = n_x^2 + n_y^2 -- one square root, one
        // reciprocal and a third of the instructions less than going through r and n_r, the same values to 1-2 ulp
        // (the deviation class of dx / r for cos(atan2) itself, see above)
        const double r2 = dx * dx + dy * dy;
        const double q = ot_sqrt(1 - sf.krho2 * r2);
        const double iq = ot_rcp3(q);
        n.x = ot_div_r(sf.nrho * dx, q, iq);
        n.y = ot_div_r(sf.nrho * dy, q, iq);
        n.z = ot_sqrt(1 - (n.x * n.x + n.y * n.y));
        return n;
    }
    double rm = ot_sqrt(dx * dx + dy * dy);
    double fr = ot_div(rm * sf.rho, ot_sqrt(1 - sf.k1rho2 * (rm * rm)));
    fr += asph_poly_deriv(sf, rm);
    const double irm = ot_rcp3(rm);
    double c = (rm > 0.0) ? ot_div_r(dx, rm, irm) : 1.0;
    double s = (rm > 0.0) ? ot_div_r(dy, rm, irm) : 0.0;
    V3 m = {-(fr * c), -(fr * s), 1.0};
    return normalize3(m);
}

// Surface._find_hit_handle_abnormal surface.py:436-479
template <class SF>
OT_DEV void handle_abnormal_f(SF& sf, const V3& p, const V3& s, V3& ph, bool& hit, double f);
template <int LEVEL = OT_HIT_SPLINE, class SF>
OT_DEV void handle_abnormal(SF& sf, const V3& p, const V3& s, V3& ph, bool& hit, PatchCache* pc = nullptr) {
    double zs = surf_values<LEVEL>(sf, ph.x, ph.y, pc);
    handle_abnormal_f(sf, p, s, ph, hit, ph.z - zs);
}

// the same with f = ph.z - values(ph) known: the numeric hit search has just evaluated it at the point it returns
// (surface.py:369 computes exactly this difference for the point that becomes p_hit), one surface evaluation less
template <class SF>
OT_DEV void handle_abnormal_f(SF& sf, const V3& p, const V3& s, V3& ph, bool& hit, double f) {
    bool dev = fabs(f) > OT_C_EPS;
    bool beh = p.z > sf.z_beh;
    bool neg = ph.z < p.z - OT_C_EPS;
    if ((neg || dev) && !beh) {
        double tnm = (sf.z_max - p.z) / s.z;
        ph = along(p, s, tnm);
        hit = false;
    }
    if (beh) {
        ph = p;
        hit = false;
    }
}

// ConicSurface.find_hit conic_surface.py:126-203 (closed-form quadratic)
template <class SF>
OT_DEV void find_hit_conic(SF& sf, const V3& p, const V3& s, V3& ph, bool& hit) {
    double ox = p.x - sf.px, oy = p.y - sf.py, oz = p.z - sf.pz;
    const bool sphere = (sf.k == 0.0);  // wave-uniform: a scalar branch, not a select
    double ozk = oz * sf.k1;
    double B = s.x * ox + s.y * oy + s.z * (ozk - sf.inv_rho);
    double C = oy * oy + ox * ox + oz * (ozk - sf.two_inv_rho);
    double A = 1.0, D, t1, t2;
    if (sphere) {  // A = 1.0: C * 1.0 == C and x / 1.0 == x exactly, so both are skipped
        D = ot_sqrt(B * B - C);
        t1 = -B - D;
        t2 = -B + D;
    } else {
        A = 1 + sf.k * (s.z * s.z);
        D = ot_sqrt(B * B - C * A);
        const double iA = ot_rcp3(A);
        t1 = ot_div_r(-B - D, A, iA);
        t2 = ot_div_r(-B + D, A, iA);
    }
    double z = p.z;
    double z1 = z + s.z * t1;
    double z2 = z + s.z * t2;
    bool c1 = (sf.z_lo <= z1) && (z1 <= sf.z_hi) && (z1 >= z);
    bool c2 = (sf.z_lo <= z2) && (z2 <= sf.z_hi) && (z2 >= z) && (t2 < t1);
    double t = (c1 && !c2) ? t1 : t2;
    ph = along(p, s, t);
    hit = surf_mask<false>(sf, ph.x, ph.y);
    if (!sphere && A == 0 && B != 0) {
        t = -C / (2 * B);
        ph = along(p, s, t);
        hit = surf_mask<false>(sf, ph.x, ph.y);
    }
    bool nh = !hit || !isfinite(D) || (!sphere && A == 0 && B == 0) || (ph.z < sf.z_lo) || (ph.z > sf.z_hi);
    if (nh) {
        double tnh = (sf.z_max - p.z) / s.z;
        ph = along(p, s, tnh);
        hit = false;
    }
    if (z > sf.z_max) {
        ph = p;
        hit = false;
    }
}

// Surface.values of an AsphericSurface (surface.py:137-164 with aspheric_surface.py:51-65), the cost function of its hit
// search.  The generic surf_values spends most of an evaluation on what an asphere does not need: the mask switch
// over every surface kind, the edge continuation of non-symmetric surfaces, a dynamic loop that fetches one
// coefficient per scalar-memory round trip, IEEE sqrt with its range scaling.  Here: disc mask and sag share r^2, the
// two roots go through ot_sqrt (same bits), and the Horner steps are unrolled behind wave-uniform tests `j < ncoeff`, so
// the coefficients are scalar registers loaded once per surface instead of once per evaluation.
// polyval over [a_2n, 0, ..., a_2, 0, 0] is y = (y r + a) r per coefficient and one more `y r`; the reference's `+ 0`
// steps only turn -0 into +0, which no later sum can see.
// NC > 0: the surface has exactly NC coefficients (compile-time chain; find_hit picks the instance with one wave-uniform
// switch outside the search loop).  NC = 0: any count, one jump into the unrolled chain per evaluation.
template <int NC, class SF>
struct AsphereSag {
    SF& sf;
    OT_DEV double operator()(double x, double y) const {
        const double dx = x - sf.px, dy = y - sf.py;
        const double r2 = dx * dx + dy * dy;
        const double r = ot_sqrt(r2);
        const double rr = r * r;
        double z = ot_div(sf.rho * rr, 1 + ot_sqrt(1 - sf.k1rho2 * rr));
        double y_ = 0.0;
        if (NC > 0) {
            y_ = sf.coeff[NC - 1] * r;  // the first step 0 r + a is a itself (a NaN or infinite r ends as NaN either way)
#pragma unroll
            for (int j = NC - 2; j >= 0; j--) {
                y_ = y_ * r + sf.coeff[j];
                y_ = y_ * r;
            }
        } else {
            // Horner from the highest coefficient the surface has: one wave-uniform jump into the unrolled chain.  (The
            // empty asm statements keep the steps from being turned into twelve always-executed select pairs.)
#define OT_ASPH_STEP(J)               \
    case J + 1:                       \
        asm volatile("");             \
        y_ = y_ * r + sf.coeff[J];    \
        y_ = y_ * r;                  \
        [[fallthrough]];
            static_assert(OT_MAX_ASPH == 12, "AsphereSag unrolls twelve coefficients");
            switch (sf.ncoeff) {
                OT_ASPH_STEP(11) OT_ASPH_STEP(10) OT_ASPH_STEP(9) OT_ASPH_STEP(8) OT_ASPH_STEP(7) OT_ASPH_STEP(6)
                OT_ASPH_STEP(5) OT_ASPH_STEP(4) OT_ASPH_STEP(3) OT_ASPH_STEP(2) OT_ASPH_STEP(1) OT_ASPH_STEP(0)
                default: break;
            }
#undef OT_ASPH_STEP
        }
        y_ = y_ * r;
        z += y_;
        return (r2 <= sf.r_eps2) ? sf.pz + z : sf.edge_val;
    }
};

// The Illinois iteration of Surface.find_hit (surface.py:363-405) on f(t) = z_ray(t) - values(x(t), y(t)).  The
// reference shrinks its active set with boolean masks every iteration; here that is the wavefront's EXEC mask: the loop
// runs while the 64-bit ballot of unconverged lanes is non-zero (one scalar branch per iteration) and converged lanes
// idle, so a wave pays max(iterations) of its own 64 rays only.  The three cases of the update are selects inside one
// masked block (no nested divergence), and a converged lane keeps its parameter and cost value (t_out, f_out) instead
// of the point: the caller forms p + s t again, the same bits.
// Returns false if a lane hit the 200-iteration timeout (surface.py:403).
template <class EVAL>
OT_DEV bool illinois_search(const V3& p, const V3& s, double t1, double t2, double f1, double f2, bool w, EVAL&& values,
                            double& t_out, double& f_out) {
    bool ok = true;
    int it = 1;
    while (__ballot(w) != 0ull) {
        if (w) {
            const double ts = t1 - ot_div(f1, f2 - f1) * (t2 - t1);
            const double fts = (p.z + s.z * ts) - values(p.x + s.x * ts, p.y + s.y * ts);
            const double prod = fts * f2;
            const bool neg = prod < 0, pos = prod > 0, zero = prod == 0;  // none of them for a NaN
            const double t1n = neg ? t2 : (zero ? ts : t1);
            const double f1n = neg ? f2 : (pos ? 0.5 * f1 : (zero ? fts : f1));
            const bool any = neg || pos || zero;
            t1 = t1n;
            f1 = f1n;
            t2 = any ? ts : t2;
            f2 = any ? fts : f2;
            if (fabs(t2 - t1) < OT_C_EPS / 10) {
                t_out = ts;
                f_out = fts;
                w = false;
            }
        }
        if (it == OT_MAX_HIT_ITER) {  // every wave reaches this exit: the loop is bounded
            ok = !w;
            w = false;
        }
        it++;
    }
    return ok;
}

// the numeric branch of Surface.find_hit for an asphere with NC coefficients (0: any number); t1, t2 = the bracket
template <int NC, class SF>
OT_DEV bool find_hit_asphere(SF& sf, const V3& p, const V3& s, double t1, double t2, bool w0, V3& ph, bool& hit, bool& ill) {
    const AsphereSag<NC, SF> sag = {sf};
    const V3 p1 = along(p, s, t1), p2 = along(p, s, t2);
    const double f1 = p1.z - sag(p1.x, p1.y), f2 = p2.z - sag(p2.x, p2.y);
    ill = f1 * f2 > 0;
    double tf = t1, ff = f1;  // lanes without a search end at p1 (surface.py:353-354)
    const bool ok = illinois_search(p, s, t1, t2, f1, f2, w0, sag, tf, ff);
    ph = along(p, s, tf);
    const double dx = ph.x - sf.px, dy = ph.y - sf.py;
    hit = dx * dx + dy * dy <= sf.r_eps2;
    handle_abnormal_f(sf, p, s, ph, hit, ff);
    return ok;
}

// Surface.find_hit surface.py:307-414.  Numeric branch = Illinois regula falsi.  The reference shrinks its
// active set with boolean masks every iteration; on the GPU the same thing is the wavefront's EXEC mask: the
// loop below runs while the 64-bit ballot of unconverged lanes is non-zero (one scalar branch per iteration)
// and converged lanes idle, so a wave pays max(iterations) of its own 64 rays only.
// Returns false if a lane hit the 200-iteration timeout (surface.py:403).
template <int LEVEL = OT_HIT_SPLINE, class SF>
OT_DEV bool find_hit(SF& sf, const V3& p, const V3& s, V3& ph, bool& hit, bool& ill, PatchCache* pc = nullptr) {
    constexpr bool TABLES = LEVEL >= OT_HIT_SPLINE;
    ill = false;
    if (sf.kind == OT_SURF_CONIC) {
        find_hit_conic(sf, p, s, ph, hit);
        return true;
    }
    if (sf.flat) {
        double t = ot_div(sf.pz - p.z, s.z);
        ph = along(p, s, t);
        // a flat function surface may still carry a mask_func bitmap (the custom apertures of
        // docs/source/usage/surfaces.rst:360-369); the scene compiler sends such scenes to the spline level
        hit = surf_mask<TABLES>(sf, ph.x, ph.y);
        handle_abnormal<LEVEL>(sf, p, s, ph, hit);
        return true;
    }
    if (LEVEL == OT_HIT_CLOSED) {  // kernel variant without numeric surfaces: unreachable, keeps the Illinois loop out of it
        ph = p;
        hit = false;
        return true;
    }
    const double isz = ot_rcp3(s.z);
    double t1 = ot_div_r(sf.zt1 - p.z, s.z, isz);
    double t2 = ot_div_r(sf.zt2 - p.z, s.z, isz);
    if (t1 < 0) t1 = -OT_C_EPS;
    const bool w0 = isfinite(t1) && isfinite(t2) && !((t2 - t1) < OT_C_EPS);
    if (sf.kind == OT_SURF_ASPHERE) {  // its own copies of the search: see AsphereSag
        switch (sf.ncoeff) {
            case 1: return find_hit_asphere<1>(sf, p, s, t1, t2, w0, ph, hit, ill);
            case 2: return find_hit_asphere<2>(sf, p, s, t1, t2, w0, ph, hit, ill);
            case 3: return find_hit_asphere<3>(sf, p, s, t1, t2, w0, ph, hit, ill);
            case 4: return find_hit_asphere<4>(sf, p, s, t1, t2, w0, ph, hit, ill);
            default: return find_hit_asphere<0>(sf, p, s, t1, t2, w0, ph, hit, ill);
        }
    }
    // TiltedSurface.find_hit tilted_surface.py:91-123: closed-form plane hit first; rays that miss the disc go
    // through the generic search below (the edge is continued radially), the others sit it out
    bool pre = false;
    V3 ph_pre = {0.0, 0.0, 0.0};
    if (sf.kind == OT_SURF_TILTED) {
        double td = s.x * sf.nx + s.y * sf.ny + s.z * sf.nz;
        bool nz0 = td != 0;
        double t = ((sf.px - p.x) * sf.nx + (sf.py - p.y) * sf.ny + (sf.pz - p.z) * sf.nz) / (nz0 ? td : 1e-12);
        ph_pre = along(p, s, t);
        pre = surf_mask<TABLES>(sf, ph_pre.x, ph_pre.y) && nz0;
    }
    const V3 p1 = along(p, s, t1), p2 = along(p, s, t2);
    double f1 = 0.0, f2 = 0.0;
    auto values = [&](double x, double y) { return surf_values<LEVEL>(sf, x, y, pc); };
    if (!pre) {
        f1 = p1.z - values(p1.x, p1.y);
        f2 = p2.z - values(p2.x, p2.y);
    }
    ill = !pre && f1 * f2 > 0;
    double tf = t1, ff = f1;
    const bool ok = illinois_search(p, s, t1, t2, f1, f2, !pre && w0, values, tf, ff);
    ph = along(p, s, tf);
    if (!pre) {
        hit = surf_mask<TABLES>(sf, ph.x, ph.y);
        handle_abnormal_f(sf, p, s, ph, hit, ff);
    }
    if (sf.kind == OT_SURF_TILTED) {  // tilted_surface.py:119-120: abnormal handling once more, for all rays
        if (pre) {
            ph = ph_pre;
            hit = true;
        }
        handle_abnormal<LEVEL>(sf, p, s, ph, hit, pc);
    }
    return ok;
}

// RingSurface.hurb_props ring_surface.py:88-121, SlitSurface.hurb_props slit_surface.py:65-87
template <class SF>
OT_DEV void hurb_props(SF& sf, double x, double y, double& a_, double& b_, V3& b, bool& inside) {
    double dx = x - sf.px, dy = y - sf.py;
    if (sf.kind == OT_SURF_RING) {
        double r = sqrt(dx * dx + dy * dy);
        inside = r < sf.ri;
        b_ = sf.ri - r;
        a_ = sqrt(b_ * sf.ri);
        b.x = (r > 0.0) ? dx / r : 1.0;  // cos(atan2(dy, dx))
        b.y = (r > 0.0) ? dy / r : 0.0;  // sin(atan2(dy, dx))
        b.z = 0.0;
    } else {
        double xr = dx, yr = dy;
        if (sf.rot) {
            xr = dx * sf.cna - dy * sf.sna;
            yr = dx * sf.sna + dy * sf.cna;
        }
        a_ = sf.hdy - fabs(yr);
        b_ = sf.hdx - fabs(xr);
        inside = (a_ > 0) && (b_ > 0);
        b.x = sf.cpa;
        b.y = sf.spa;
        b.z = 0.0;
    }
}

// ---- media: RefractionIndex.__call__ refraction_index.py:62-169 ------------------------------------------
// numpy.interp on a sorted table (compiled_base.c arr_interp): binary search for the interval, exact value
// on a node, linear elsewhere; `left`/`right` = 0 outside (spectrum.py:106)
template <class PL>
OT_HD double interp_tab(double x, PL xp, PL fp, int n) {
    if (isnan(x)) return x;
    if (x < xp[0] || x > xp[n - 1]) return 0.0;
    int lo = 0, hi = n - 1;
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (xp[mid] <= x)
            lo = mid;
        else
            hi = mid;
    }
    int j = (xp[hi] <= x) ? hi : lo;
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

// TAB = false compiles the table models (per-lane global loads) out.  That matters far beyond the two cases:
// vmcnt retires in order, so ANY vector-memory load in the tracing loop makes the compiler wait for all section
// stores issued before it (and, through control-flow merges, before every write of the register the load might
// target).  Scenes without tabulated media/filters therefore run a loop that contains no VMEM load at all and
// never drains its store queue.
// The wavelength passes through an empty asm statement first: everything below depends on the ray's wavelength alone, so
// called from the step loop it is loop-invariant, and the optimiser hoists the wavelength-only subexpressions of EVERY
// model -- five pow() calls among them -- in front of the loop and evaluates them for every ray whatever the scene's
// media are (measured: 570 of the 1345 vector instructions per wave of C4's kernel before any surface was reached).
// Behind the barrier only the model the (wave-uniform) switch selects is evaluated, where it is needed.
// x**k for the whole-number exponents 3 .. 6 of the dispersion formulas (NumPy calls libm's pow for them,
// refraction_index.py:102-148).  On the device: repeated multiplication, within 3 ulp of pow -- the device library's pow
// is not correctly rounded either, and its code needs ~40 vector registers that every formula kernel then carried
// (122 instead of ~100 with polarisation).  The host (tables of discrete spectra) keeps libm's pow like the reference.
OT_HD double ot_powi(double x, int k) {
#if defined(__HIP_DEVICE_COMPILE__)
    const double x2 = x * x;
    switch (k) {
        case 3: return x2 * x;
        case 4: return x2 * x2;
        case 5: return x2 * x2 * x;
        default: return x2 * x2 * x2;
    }
#else
    return pow(x, (double)k);
#endif
}

OT_HD double ot_pow35(double x) {  // x**3.5 (Conrady): x^3 sqrt(x) on the device, see ot_powi
#if defined(__HIP_DEVICE_COMPILE__)
    return x * x * x * sqrt(x);
#else
    return pow(x, 3.5);
#endif
}

template <bool TAB = true, class MD, class PL>
OT_HD double medium_n(MD& md, PL pool, float wl32) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(wl32));
#endif
    double wl = (double)wl32;
    auto* c = md.c;
    double um = wl * 1e-3;
    double wl2 = um * um;
    switch (md.model) {
        case OT_N_CONSTANT: return c[0];
        case OT_N_ABBE: return c[0] + ot_div(c[1], wl2 - c[2]);
        case OT_N_CONRADY: return c[0] + c[1] / um + c[2] / ot_pow35(um);
        case OT_N_CAUCHY: return c[0] + ot_div(c[1], wl2) + ot_div(c[2], wl2 * wl2) + ot_div(c[3], ot_powi(wl2, 3));
        case OT_N_SELLMEIER1:
            return sqrt(1 + ot_div(c[0] * wl2, wl2 - c[1]) + ot_div(c[2] * wl2, wl2 - c[3]) + ot_div(c[4] * wl2, wl2 - c[5]));
        case OT_N_SELLMEIER2:
            return sqrt(1 + c[0] + c[1] * wl2 / (wl2 - c[2] * c[2]) + c[3] / (wl2 - c[4] * c[4]));
        case OT_N_SELLMEIER3:
            return sqrt(1 + c[0] * wl2 / (wl2 - c[1]) + c[2] * wl2 / (wl2 - c[3]) + c[4] * wl2 / (wl2 - c[5]) +
                        c[6] * wl2 / (wl2 - c[7]));
        case OT_N_SELLMEIER4: return sqrt(c[0] + c[1] * wl2 / (wl2 - c[2]) + c[3] * wl2 / (wl2 - c[4]));
        case OT_N_SELLMEIER5:
            return sqrt(1 + c[0] * wl2 / (wl2 - c[1]) + c[2] * wl2 / (wl2 - c[3]) + c[4] * wl2 / (wl2 - c[5]) +
                        c[6] * wl2 / (wl2 - c[7]) + c[8] * wl2 / (wl2 - c[9]));
        case OT_N_SCHOTT:
            return sqrt(c[0] + c[1] * wl2 + c[2] / wl2 + c[3] / (wl2 * wl2) + c[4] / ot_powi(wl2, 3) +
                        c[5] / ot_powi(wl2, 4));
        case OT_N_HERZBERGER: {
            double L = 1 / (wl2 - 0.028);
            return c[0] + c[1] * L + c[2] * (L * L) + c[3] * wl2 + c[4] * (wl2 * wl2) + c[5] * ot_powi(wl2, 3);
        }
        case OT_N_HOO1: return sqrt(c[0] + c[1] / (wl2 - c[2]) - c[3] * wl2);
        case OT_N_HOO2: return sqrt(c[0] + c[1] * wl2 / (wl2 - c[2]) - c[3] * wl2);
        case OT_N_EXTENDED:
            return sqrt(c[0] + c[1] * wl2 + c[2] / wl2 + c[3] / (wl2 * wl2) + c[4] / ot_powi(wl2, 3) +
                        c[5] / ot_powi(wl2, 4) + c[6] / ot_powi(wl2, 5) + c[7] / ot_powi(wl2, 6));
        case OT_N_EXTENDED2:
            return sqrt(c[0] + c[1] * wl2 + c[2] / wl2 + c[3] / (wl2 * wl2) + c[4] / ot_powi(wl2, 3) +
                        c[5] / ot_powi(wl2, 4) + c[6] * (wl2 * wl2) + c[7] * ot_powi(wl2, 3));
        case OT_N_EXTENDED3:
            return sqrt(c[0] + c[1] * wl2 + c[2] * (wl2 * wl2) + c[3] / wl2 + c[4] / (wl2 * wl2) +
                        c[5] / ot_powi(wl2, 3) + c[6] * ot_powi(wl2, 4) + c[7] * ot_powi(wl2, 5) + c[8] / ot_powi(wl2, 6));
        case OT_N_DATA: {
            if (!TAB) break;
            auto xp = pool + md.tab_off;
            return interp_tab(wl, xp, xp + md.tab_len, md.tab_len);
        }
        case OT_N_LINES: {
            if (!TAB) break;
            auto xp = pool + md.tab_off;
            double v = __builtin_nan("");
            for (int j = 0; j < md.tab_len; j++)
                if (xp[j] == wl) v = xp[md.tab_len + j];
            return v;
        }
    }
    return __builtin_nan("");
}

// Filter.__call__ filter.py:39 -> transmission_spectrum.py:73-84 -> spectrum.py:81-120
template <bool TAB = true, class FD, class PL>
OT_HD double filter_T(FD& f, PL pool, float wl32) {
    double wl = (double)wl32;
    double T;
    switch (f.type) {
        case OT_T_CONSTANT: T = f.val; break;
        case OT_T_DATA: {
            if (!TAB) { T = __builtin_nan(""); break; }
            auto xp = pool + f.tab_off;
            T = interp_tab(wl, xp, xp + f.tab_len, f.tab_len);
            break;
        }
        case OT_T_RECTANGLE: T = (f.wl0 <= wl && wl <= f.wl1) ? f.val : 0.0; break;
        case OT_T_GAUSSIAN: {  // float32 arithmetic, spectrum.py:113-115 with the tracer's f32 wavelengths
            float d = wl32 - f.mu32;
            float q = -(d * d) / f.den32;
            T = (double)(f.val32 * expf(q));
            break;
        }
        case OT_T_LINES: {
            T = __builtin_nan("");
            if (!TAB) break;
            auto xp = pool + f.tab_off;
            for (int j = 0; j < f.tab_len; j++)
                if (xp[j] == wl) T = xp[f.tab_len + j];
            break;
        }
        default: T = __builtin_nan("");
    }
    return f.inverse ? 1.0 - T : T;
}

// sin(pi t), cos(pi t) for |t| <= 4 (azimuths and polarisation angles of the generator): quadrant reduction
// k = rint(2 t), then the fdlibm kernels on |x| <= pi / 4 (errors < 1 ulp).  About 30 instructions against ~80 of the
// device library's sincospi with its general range reduction.
OT_DEV void sincospi_small(double t, double* sn, double* cs) {
    // explicit fused multiply-adds (not `#pragma clang fp contract`): every kernel that generates rays -- stored and
    // render-only forms, every feature level -- must form the SAME bits, whatever the code around the call looks like
    const double k = rint(2.0 * t);
    const double x = __builtin_fma(-0.5, k, t) * 3.141592653589793;  // exact subtraction, |x| <= pi / 4
    const double z = x * x;
    // __kernel_sin / __kernel_cos (fdlibm), argument reduced exactly so the tail terms vanish
    double ps = 1.58969099521155010221e-10;
    ps = __builtin_fma(z, ps, -2.50507602534068634195e-08);
    ps = __builtin_fma(z, ps, 2.75573137070700676789e-06);
    ps = __builtin_fma(z, ps, -1.98412698298579493134e-04);
    ps = __builtin_fma(z, ps, 8.33333333332248946124e-03);
    ps = __builtin_fma(z, ps, -1.66666666666666324348e-01);
    double pc = -1.13596475577881948265e-11;
    pc = __builtin_fma(z, pc, 2.08757232129817482790e-09);
    pc = __builtin_fma(z, pc, -2.75573143513906633035e-07);
    pc = __builtin_fma(z, pc, 2.48015872894767294178e-05);
    pc = __builtin_fma(z, pc, -1.38888888888741095749e-03);
    pc = __builtin_fma(z, pc, 4.16666666666666019037e-02);
    const double s0 = __builtin_fma(x * z, ps, x);
    const double c0 = __builtin_fma(z * z, pc, __builtin_fma(-0.5, z, 1.0));
    const int q = (int)k & 3;  // rotation by q quarter turns (two's complement & 3 is the quadrant for negative k too)
    const double s1 = (q & 1) ? c0 : s0, c1 = (q & 1) ? s0 : c0;
    *sn = (q & 2) ? -s1 : s1;
    *cs = ((q + 1) & 2) ? -c1 : c1;
}

// ---- counter-based RNG: Philox-4x32-10 (Salmon et al., SC'11) ---------------------------------------------
struct Philox {
    uint32_t c[4];
};

// ROUNDS = 10 is the standard generator; Philox4x32-7 is the smallest variant its authors report as passing BigCrush
// (Salmon et al., table 2) and serves where the numbers only dither positions inside strata (ray generation).
template <int ROUNDS = 10>
OT_DEV Philox philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < ROUNDS; r++) {
        uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
        uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
        uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0;
        c1 = n1;
        c2 = n2;
        c3 = n3;
        k0 += W0;
        k1 += W1;
    }
    Philox p = {{c0, c1, c2, c3}};
    return p;
}

// two uniforms in [0, 1) with 53 random bits each
OT_DEV void philox_u2(uint64_t seed, uint64_t idx, uint32_t stream, uint32_t sub, double& u0, double& u1) {
    Philox p = philox4x32((uint32_t)idx, (uint32_t)(idx >> 32), stream, sub, (uint32_t)seed, (uint32_t)(seed >> 32));
    uint64_t a = ((uint64_t)p.c[1] << 32) | p.c[0];
    uint64_t b = ((uint64_t)p.c[3] << 32) | p.c[2];
    u0 = (double)(a >> 11) * 0x1.0p-53;
    u1 = (double)(b >> 11) * 0x1.0p-53;
}

// two independent standard normals (Box-Muller on Philox uniforms)
OT_DEV void philox_normal2(uint64_t seed, uint64_t idx, uint32_t stream, uint32_t sub, double& z0, double& z1) {
    double u0, u1;
    philox_u2(seed, idx, stream, sub, u0, u1);
    double r = ot_sqrt(-2.0 * log(1.0 - u0));  // 1-u0 in (0, 1]
    double sn, cs;
    sincospi_small(2.0 * u1, &sn, &cs);  // angle 2 pi u1 without the general range reduction of sincos
    z0 = r * cs;
    z1 = r * sn;
}
