// The tracing megakernel: one launch walks every ray through all elements of the scene.
//
// Reference: Raytracer.trace / sub_trace (raytracer.py:262-415) runs ~25 whole-array NumPy passes per
// surface, each streaming the ray arrays through DRAM.  Here a ray's state (position, direction, weight,
// polarisation, wavelength, current index) lives in registers of one lane for the whole walk; HBM sees only
// the compulsory traffic -- the section stores RayStorage must contain afterwards (ray_storage.py:80-90),
// N*[(M+2)*48+28] bytes with polarisation -- as fully coalesced 512 B / 256 B wavefront stores in the
// reference's struct-of-arrays (Fortran) layout.  Scene constants are wave-uniform scalar loads.
#pragma once
#include "ot_device.hpp"
#include "ot_generate.hpp"

// Event counters (Raytracer._msgs, raytracer.py:289): wave-aggregated (one ballot, the section index is
// wave-uniform) into a per-workgroup LDS table; the persistent workgroup flushes its non-zero entries to the
// global table once at its end.  Global atomics per event would serialise on a handful of hot addresses:
// measured 5.4 ms -> 2.4 ms for the 10 M ray double-Gauss launch when they were removed from the loop.
OT_DEV void count_event(unsigned int* cnt, int nt, int info, int sec, bool cond) {
    unsigned long long m = __ballot(cond);
    if (m != 0ull) {
        int lane = __lane_id();
        if (lane == (int)__ffsll((long long)m) - 1) atomicAdd(&cnt[info * nt + sec], (unsigned int)__popcll(m));
    }
}

struct RayState {
    V3 p, s;
    float w, wl;
    float polx, poly, polz;
    double n_cur;
};

// Raytracer.__compute_polarization raytracer.py:831-879 (hwh == true for this lane).
// Everything in here only reaches float32-stored quantities (pol_list, and through A_ts/A_tp the weight), never
// a position, direction or mask, so it is evaluated with fused multiply-adds and one reciprocal square root
// instead of sqrt + three divisions: agreement with the reference is at the 1e-15 level before the float32
// rounding of the store (bar: 1e-6 relative).
// The reference's PROJECTION form (pol' = A_ts ps + A_tp pp') is kept on purpose: the float32-stored pol is only
// perpendicular to s to ~3e-8, and a rotation form (Rodrigues about ps; 25 % fewer instructions) carries that
// parallel component along instead of dropping it.  That flips the float32 rounding of pol' in ~17 % of the
// components, and since T is first order in pol the weights then drift by ~1e-7 per surface (measured 2.6e-6
// after 15 surfaces) -- outside the 1e-6 bar.
template <bool POL>
OT_DEV void compute_polarization(const V3& s, const V3& s_, const RayState& r, float& npx, float& npy, float& npz,
                                 double& A_ts, double& A_tp) {
#pragma clang fp contract(fast)
    const double inv_sqrt2 = 0.7071067811865475;  // 1/np.sqrt(2)
    if (!POL) {
        A_ts = inv_sqrt2;
        A_tp = inv_sqrt2;
        return;
    }
    bool mask = (s.x != s_.x) || (s.y != s_.y) || (s.z != s_.z);
    V3 ps = cross3(s_, s);
    double inv = rsqrt(ps.x * ps.x + ps.y * ps.y + ps.z * ps.z);
    ps.x *= inv;
    ps.y *= inv;
    ps.z *= inv;
    V3 pp = cross3(ps, s);
    V3 pol = {(double)r.polx, (double)r.poly, (double)r.polz};
    A_ts = dot3(ps, pol);
    A_tp = dot3(pp, pol);
    if (!mask) {
        A_ts = inv_sqrt2;
        A_tp = inv_sqrt2;
    }
    V3 pp_ = cross3(ps, s_);
    if (mask) {
        npx = (float)(ps.x * A_ts + pp_.x * A_tp);
        npy = (float)(ps.y * A_ts + pp_.y * A_tp);
        npz = (float)(ps.z * A_ts + pp_.z * A_tp);
    }
}

// 1/x to ~1 ulp: hardware reciprocal seed (v_rcp_f64, ~2^-26... relative) + two Newton steps; only used where the
// result ends up in a float32 (weights)
OT_DEV double fast_rcp(double x) {
    double y = __builtin_amdgcn_rcp(x);
    y = fma(fma(-x, y, 1.0), y, y);
    y = fma(fma(-x, y, 1.0), y, y);
    return y;
}

// Fresnel power transmission raytracer.py:813-819, algebraically regrouped to a single division:
//   T = n2cb/n1ca * ((A_ts*ts)^2 + (A_tp*tp)^2),  ts = 2 n1ca/d1,  tp = 2 n1ca/d2
//     = 4 n1ca n2cb (A_ts^2 d2^2 + A_tp^2 d1^2) / (d1 d2)^2
// (feeds only the float32 weight; see compute_polarization for the tolerance argument)
OT_DEV double fresnel_T(double n1, double n2, double ns, double W, double A_ts, double A_tp) {
#pragma clang fp contract(fast)
    double n1ca = n1 * ns, n2cb = n2 * W;
    double d1 = n1ca + n2cb;
    double d2 = n2 * ns + n1 * W;
    double a = A_ts * d2, b = A_tp * d1;
    double den = d1 * d2;
    double T = 4 * n1ca * n2cb * (a * a + b * b) * fast_rcp(den * den);
    if (n1ca == 0) T = __builtin_nan("");  // the reference divides by n1*cos(alpha)
    return T;
}

// __compute_polarization specialised for refraction, where s' = N s - q n lies in the plane of n and s.
// With m = n x s (|m| = sin(alpha)) the reference's basis vectors are, by (a x b) x c = b (a.c) - a (b.c):
//     ps  = +-m / |m|                      (the sign cancels in A_ts ps and in A_ts^2)
//     pp  = ps x s  = (ns s - n) / |m|
//     pp' = ps x s' = (W s - (s.s') n) / |m|,      n.s' = W,   s.s' = N - q ns
// so   A_ts = (m.pol)/|m|,  A_tp = (ns (s.pol) - n.pol)/|m|,  pol' = [(m.pol) m + A_tp|m| (W s - (s.s') n)] / |m|^2.
// No identity relies on pol being exactly perpendicular to s (the float32-stored pol is not; see above why that
// matters), so this agrees with the reference's projection to rounding, with one cross product instead of three
// and a reciprocal instead of a reciprocal square root (~66 instead of ~83 instructions).
template <bool POL>
OT_DEV void refraction_polarization(const V3& n, const V3& s, const V3& s_, double ns, double W, double N, double q,
                                    const RayState& r, float& npx, float& npy, float& npz, double& A_ts2, double& A_tp2) {
#pragma clang fp contract(fast)
    if (!POL) {  // (1/np.sqrt(2))**2 each
        A_ts2 = 0.5;
        A_tp2 = 0.5;
        return;
    }
    bool mask = (s.x != s_.x) || (s.y != s_.y) || (s.z != s_.z);
    V3 m = cross3(n, s);
    V3 pol = {(double)r.polx, (double)r.poly, (double)r.polz};
    double mm = dot3(m, m), mp = dot3(m, pol), sp = dot3(s, pol), np_ = dot3(n, pol);
    // n and s exactly parallel (a collimated beam along a face normal) while s' still differs from s by rounding:
    // the plane of incidence is undefined (m = 0, 1 / mm not finite).  The reference builds its basis from the
    // rounding noise s' x s and gets A_ts^2 + A_tp^2 = |pol|^2 with ts = tp, i.e. the normal-incidence transmission
    // and an unchanged pol -- which is the unchanged-direction branch.
    if (!(mm > 0)) mask = false;
    double inv = fast_rcp(mm);
    double tp = ns * sp - np_;   // A_tp |m|
    double ct = N - q * ns;      // s . s'
    A_ts2 = mp * mp * inv;
    A_tp2 = tp * tp * inv;
    if (!mask) {
        A_ts2 = 0.5;
        A_tp2 = 0.5;
    }
    if (mask) {
        V3 vec = {W * s.x - ct * n.x, W * s.y - ct * n.y, W * s.z - ct * n.z};
        npx = (float)((mp * m.x + tp * vec.x) * inv);
        npy = (float)((mp * m.y + tp * vec.y) * inv);
        npz = (float)((mp * m.z + tp * vec.z) * inv);
    }
}

// Fresnel power transmission from squared amplitudes (see fresnel_T)
OT_DEV double fresnel_T2(double n1, double n2, double ns, double W, double A_ts2, double A_tp2) {
#pragma clang fp contract(fast)
    double n1ca = n1 * ns, n2cb = n2 * W;
    double d1 = n1ca + n2cb;
    double d2 = n2 * ns + n1 * W;
    double den = d1 * d2;
    double T = 4 * n1ca * n2cb * (A_ts2 * (d2 * d2) + A_tp2 * (d1 * d1)) * fast_rcp(den * den);
    if (n1ca == 0) T = __builtin_nan("");  // the reference divides by n1*cos(alpha)
    return T;
}

// Raytracer.__refraction raytracer.py:761-829 for a lane that has power and hit the surface.
// The new direction s' is computed in the reference's exact operation order (it feeds the next hit mask).
// Returns true on total internal reflection.
template <bool POL, int LEVEL, class SF>
OT_DEV bool refract(SF& sf, RayState& r, const V3& pn, float& wn, float& npx, float& npy, float& npz,
                    double n1, double n2, double N, PatchCache* pc = nullptr) {  // N = n1 / n2 (raytracer.py:799)
    V3 n = surf_normal<true, LEVEL>(sf, pn.x, pn.y, pc);  // pn is a hit point: is_hit implies mask(pn) (surface.py:409)
    V3 s = r.s;
    double ns = dot3(n, s);
    double W = ot_sqrt(1 - N * N * (1 - ns * ns));
    double q = N * ns - W;
    V3 s_ = {s.x * N - n.x * q, s.y * N - n.y * q, s.z * N - n.z * q};

    double A_ts2, A_tp2;
    refraction_polarization<POL>(n, s, s_, ns, W, N, q, r, npx, npy, npz, A_ts2, A_tp2);
    double T = fresnel_T2(n1, n2, ns, W, A_ts2, A_tp2);
    bool tir = !isfinite(W);
    if (tir) T = 0;
    wn = (float)((double)r.w * T);
    r.s = s_;
    return tir;
}

// Raytracer.__refraction_ideal_lens raytracer.py:720-759
template <bool POL, class SF, class EL>
OT_DEV void refract_ideal(SF& sf, EL& el, RayState& r, const V3& pn, float& npx, float& npy,
                          float& npz) {
    V3 s0 = r.s;
    double fsz = el.f / s0.z;
    V3 s = {s0.x * fsz - (pn.x - sf.px), s0.y * fsz - (pn.y - sf.py), el.f};
    s = normalize3(s);
    s.x = s.x * el.fsign;
    s.y = s.y * el.fsign;
    s.z = s.z * el.fsign;
    r.s = s;
    double A_ts, A_tp;
    compute_polarization<POL>(s0, s, r, npx, npy, npz, A_ts, A_tp);
}

// Raytracer.__outline_intersection raytracer.py:666-718 for one lane of the mask; returns true if clipped
template <class OL>
OT_DEV bool outline_clip(OL o, const V3& p, const V3& s, V3& pn, float& wn) {
    bool inside = (o[0] < pn.x) && (pn.x < o[1]) && (o[2] < pn.y) && (pn.y < o[3]) && (o[4] < pn.z) && (pn.z < o[5]);
    if (inside) return false;
    double t = __builtin_nan("");
    double T;
    T = (o[0] - p.x) / s.x; if (T > 0 && !(t <= T)) t = T;
    T = (o[1] - p.x) / s.x; if (T > 0 && !(t <= T)) t = T;
    T = (o[2] - p.y) / s.y; if (T > 0 && !(t <= T)) t = T;
    T = (o[3] - p.y) / s.y; if (T > 0 && !(t <= T)) t = T;
    T = (o[4] - p.z) / s.z; if (T > 0 && !(t <= T)) t = T;
    T = (o[5] - p.z) / s.z; if (T > 0 && !(t <= T)) t = T;
    pn = along(p, s, t);
    wn = 0.f;
    return true;
}

// Raytracer.__hurb raytracer.py:417-490 for one lane.  Returns true if the (possibly bent) direction points
// in -z (absorbed + counted; applies to every ray of the bundle, alive or not, raytracer.py:484-486).
template <bool POL, class SC, class SF>
OT_DEV bool hurb_bend(SC& sc, SF& sf, RayState& r, const V3& pn, float& wn, float& npx,
                      float& npy, float& npz, bool hwnh, double za, double zb) {
    double a_, b_;
    V3 b;
    bool inside;
    hurb_props(sf, pn.x, pn.y, a_, b_, b, inside);
    bool bend = hwnh && inside;
    V3 a = {-b.y, b.x, b.z};
    V3 s = r.s;
    double sa_ = dot3(s, a), sb_ = dot3(s, b);
    double cos_psi_a = ot_sqrt(1 - sa_ * sa_);
    double cos_psi_b = ot_sqrt(1 - sb_ * sb_);
    float wlm = r.wl * (float)1e-9;  // float32 product in the reference (wl_list is float32)
    double k = ot_div(2 * M_PI * r.n_cur, (double)wlm);
    double tan_sig_b = ot_div(sc.hurb_factor, 2 * b_ * cos_psi_b * 1e-3 * k);
    double tan_sig_a = ot_div(sc.hurb_factor, 2 * a_ * cos_psi_a * 1e-3 * k);
    double tan_tha = fabs(tan_sig_a) * za;
    double tan_thb = fabs(tan_sig_b) * zb;
    V3 sa = normalize3(cross3(b, s));
    V3 sb = cross3(s, sa);
    V3 sab = {s.x + sa.x * tan_tha + sb.x * tan_thb, s.y + sa.y * tan_tha + sb.y * tan_thb,
              s.z + sa.z * tan_tha + sb.z * tan_thb};
    V3 s0 = s;
    if (bend) {
        s = normalize3(sab);
        r.s = s;
    }
    bool neg = s.z < 0;
    if (neg) wn = 0.f;
    if (bend) {
        double A_ts, A_tp;
        compute_polarization<POL>(s0, s, r, npx, npy, npz, A_ts, A_tp);
    }
    return neg;
}

// Stores with a wave-uniform 64-bit base in SGPRs and a 32-bit per-lane byte offset: `global_store ... v_off, v_data,
// s[base]`.  Written as inline assembly because the compiler otherwise keeps one 64-bit VGPR address per stored plane
// and advances each of them every section (9 v_lshl_add_u64 per section: 3.5 % of the loop's vector instructions).
// The base is copied to a scratch SGPR pair inside the statement: a base the register allocator reloads from a spill
// slot comes out of v_readlane_b32, and a vector-memory instruction that reads an SGPR written by the vector ALU less
// than five wait states earlier uses the OLD value (gfx9 hazard).  The compiler pads that hazard for its own
// instructions but cannot see into an asm statement; a scalar move in between is interlocked by the hardware on both
// sides.  (Found as stray final directions in 1 of ~3 runs of the kernel variants that spill SGPRs.)
#ifndef OT_STORE_HINT
// Cache-policy bits of the section stores.  Every section is written once and never read by this kernel: as non-temporal
// stores they stream past the caches instead of being allocated there (A/B on one box, tools/ab_bench.py,
// profiles/r2/store_hints.txt: kernel 1.671 -> 1.622 ms with " nt"; " sc0 sc1" alone 1.667; " sc0 sc1 nt" 1.619, shipped).
#define OT_STORE_HINT " sc0 sc1 nt"
#endif
// The statements clobber "memory": they write it, and without the clobber the compiler may move its own loads and stores
// of the same planes across them (the kernel variant that loads section 0 from the planes it then overwrites).
#ifndef OT_STORE_CLOBBER
#define OT_STORE_CLOBBER : "memory"
#endif
OT_DEV void store_f64(const void* base, uint32_t off, double v) {
    const void* b;
    asm volatile("s_mov_b64 %0, %3\n\tglobal_store_dwordx2 %1, %2, %0" OT_STORE_HINT : "=&s"(b) : "v"(off), "v"(v), "s"(base) OT_STORE_CLOBBER);
}
OT_DEV void store_f32(const void* base, uint32_t off, float v) {
    const void* b;
    asm volatile("s_mov_b64 %0, %3\n\tglobal_store_dword %1, %2, %0" OT_STORE_HINT : "=&s"(b) : "v"(off), "v"(v), "s"(base) OT_STORE_CLOBBER);
}

// One section of one ray.  The plane base addresses (array + N * plane) are wave-uniform and stay in SGPRs; the
// lane contributes the byte offsets of its ray in an f64 plane (`o8` = 8 * index) and in an f32 plane (`o4`),
// formed once per ray.  The offsets are 32 bits wide: a launch covers at most 2^28 rays (the host entry points
// split longer bundles and advance the base pointers).
template <bool POL>
OT_DEV void store_section(const ot_rays& R, uint32_t o8, uint32_t o4, int sec, const V3& p, float w, double n, float px,
                          float py, float pz) {
    // Measured on the bench scene, same box (profiles/r2/store_variants.txt): this form 1.676 ms, per-plane 64-bit
    // VGPR addresses 1.700 ms, a wave-tiled layout ([tile of 64 rays][section][component][lane] inside each array)
    // 1.682 ms -- so the reference's planar Fortran layout stays and host views need no re-ordering.  Without the
    // section stores the same kernel runs 1.238 ms at 2.39 GHz; with them the chip holds 1.97 GHz: compute and stores
    // do overlap, but together they run into the package power limit.
    const int64_t N = R.N;
    const int64_t nt = R.nt;
    store_f64(R.p + N * sec, o8, p.x);
    store_f64(R.p + N * (sec + nt), o8, p.y);
    store_f64(R.p + N * (sec + 2 * nt), o8, p.z);
    store_f32(R.w + N * sec, o4, w);
    store_f64(R.n + N * sec, o8, n);
    if (POL) {
        store_f32(R.pol + N * sec, o4, px);
        store_f32(R.pol + N * (sec + nt), o4, py);
        store_f32(R.pol + N * (sec + 2 * nt), o4, pz);
    }
}

// The same with running plane bases: eight wave-uniform 64-bit pointers that advance by one plane per section (two
// scalar adds each) instead of eight `array + N * (section + k * nt)` products per section.
struct PlaneBases {
    const double *px, *py, *pz, *n;
    const float *w, *qx, *qy, *qz;
    int64_t N;
};
template <bool POL>
OT_DEV PlaneBases plane_bases(const ot_rays& R) {
    const int64_t N = R.N, nt = R.nt;
    PlaneBases b = {R.p, R.p + N * nt, R.p + N * 2 * nt, R.n, R.w, nullptr, nullptr, nullptr, N};
    if (POL) {
        b.qx = R.pol;
        b.qy = R.pol + N * nt;
        b.qz = R.pol + N * 2 * nt;
    }
    return b;
}
template <bool POL>
OT_DEV void store_section_next(PlaneBases& b, uint32_t o8, uint32_t o4, const V3& p, float w, double n, float px, float py,
                               float pz) {
    store_f64(b.px, o8, p.x);
    store_f64(b.py, o8, p.y);
    store_f64(b.pz, o8, p.z);
    store_f32(b.w, o4, w);
    store_f64(b.n, o8, n);
    b.px += b.N, b.py += b.N, b.pz += b.N, b.n += b.N, b.w += b.N;
    if (POL) {
        store_f32(b.qx, o4, px);
        store_f32(b.qy, o4, py);
        store_f32(b.qz, o4, pz);
        b.qx += b.N, b.qy += b.N, b.qz += b.N;
    }
}
#ifndef OT_RUNNING_BASES
#define OT_RUNNING_BASES 1
#endif

// sub_trace raytracer.py:297-397 for one ray whose section 0 state is in `r`.
// The element list is flattened on the host into one STEP per tracing surface (lens front, lens back, ideal
// lens, filter, aperture), so the loop body contains a single copy of the hit search, the refraction and the
// outline clip: ~3x less code than walking elements (instruction cache) and lower register pressure.
// SPEC selects how wavelength-dependent quantities are obtained:
//   0  formulas only (no vector-memory load in the loop)      1  formulas + tabulated media / filters (global loads)
//   2  discrete spectrum: n, n1/n2 and filter T of every step were tabulated per line on the host and staged in
//      LDS (`ltab`); the lane only carries its line index.  Saves the IEEE division n1/n2 and the dispersion
//      formula per ray-surface and covers "Function" media exactly.
// `local` = index of the ray in this launch (R's pointers start at the launch's first ray), `ray` = its index in the
// whole bundle (random-number keys, injected HURB normals).
// TAIL (render-only chunks of iterative_render, trace_tail_kernel): no section is stored; `tail` receives the start of the
// LAST section (position and weight at section nt - 2), the caller takes its end from `r` -- all that a detector behind
// the last surface needs of a ray (raytracer.py:929-985).
struct TailState {
    V3 p;
    float w;
};
template <bool POL, int SPEC, int FEAT, bool TAIL = false, class SC>
OT_DEV bool trace_ray(SC& sc, const ot_rays& R, uint32_t local, uint64_t ray, RayState& r,
                      const double* __restrict__ hurb_normals, uint64_t seed, unsigned int* msgs, const double* ltab,
                      int lj, double* patch_lds, TailState* tail = nullptr) {
    constexpr bool TAB = (SPEC == 1);
    constexpr bool FULL = (FEAT & 1) != 0;  // HURB -- and, at hit level 0, ideal lenses and filters
    constexpr int LEVEL = FEAT / 2;         // hit level (ot_device.hpp): closed form / + Illinois search / + spline surfaces
    // Ideal lenses and filters cost 2-4 registers: the levels with a numeric hit search carry them always (their scenes
    // then need the 40-register HURB code only if they use HURB: the asphere test scene runs at 103 instead of 128
    // registers, the spline scene at three waves instead of two); level 0 keeps them behind the bit, so that the bench
    // kernel stays at 74.
    constexpr bool IDEAL_FILTER = FULL || LEVEL >= OT_HIT_ILLINOIS;
    constexpr bool NUMERIC = LEVEL >= OT_HIT_ILLINOIS;
    constexpr bool SPLINE = LEVEL >= OT_HIT_SPLINE;
    // lj = line index of this ray (SPEC == 2): found by the kernel by comparing the generated wavelength with the <= 8
    // entries of the line table in LDS (trace_kernel; seven LDS compares per ray, measured against carrying the index out of
    // generate_ray: no difference)
    const double* lrow = ltab + OT_MAX_LINES + lj;  // row 0 of this lane's column
    const uint32_t o8 = local * 8u, o4 = local * 4u;
    // numeric surfaces: this lane's coefficient patch in LDS (ot_spline.hpp::PatchCache)
    PatchCache pcache = {patch_lds + threadIdx.x, 256, nullptr};  // (spline-level launches always carry the buffer)
    PatchCache* const pc = SPLINE ? &pcache : nullptr;
    const int nt = sc.nt;
    const auto surfaces = as_const(sc.surfaces);
    const auto steps = as_const(sc.steps);
    const auto media = as_const(sc.media);
    const auto filters = as_const(sc.filters);
    const auto pool = as_const(sc.pool);
    bool ok = true;
    r.n_cur = (SPEC == 2) ? lrow[(3 * sc.n_steps) * OT_MAX_LINES] : medium_n<TAB>(media[sc.n0], pool, r.wl);
#if OT_RUNNING_BASES
    PlaneBases planes = {};
    if (!TAIL) {
        planes = plane_bases<POL>(R);
        store_section_next<POL>(planes, o8, o4, r.p, r.w, r.n_cur, r.polx, r.poly, r.polz);
    }
#else
    if (!TAIL) store_section<POL>(R, o8, o4, 0, r.p, r.w, r.n_cur, r.polx, r.poly, r.polz);
#endif

    for (int i = 0; i < sc.n_steps; i++) {  // i = index of the section the ray starts this step in
        auto& st = steps[i];
        if (TAIL && i + 1 == sc.n_steps) {  // the last section starts here
            tail->p = r.p;
            tail->w = r.w;
        }
        auto& sf = surfaces[st.surf];
        const int kind = st.kind;
        V3 pn = r.p;
        float wn = r.w;
        float npx = r.polx, npy = r.poly, npz = r.polz;
        const bool hw = r.w > 0;
        V3 ph;
        bool hit = false, ill = false;
        if (SPLINE) {  // another surface, another table: the cached patch and cell belong to the previous one
            pcache.key = nullptr;
        }
        if (hw) {
            ok &= find_hit<LEVEL>(sf, r.p, r.s, ph, hit, ill, pc);
            pn = ph;
        }
        if (NUMERIC) count_event(msgs, nt, OT_INFO_ILL_COND, i + 1, hw && ill);
        const bool hwh = hw && hit, hwnh = hw && !hit;
        bool tir = false, neg = false, clip = false;
        double n_next = r.n_cur;

        if (kind <= OT_STEP_IDEAL) {  // refracting surfaces: raytracer.py:314-370
            if (hwnh) {
                wn = 0.f;
                if (kind == OT_STEP_LENS_BACK) pn = r.p;  // absorbed at the front surface, raytracer.py:354
            }
            double Nq;
            if (SPEC == 2) {
                n_next = lrow[(3 * i + 0) * OT_MAX_LINES];
                Nq = lrow[(3 * i + 1) * OT_MAX_LINES];
            } else {
                n_next = medium_n<TAB>(media[st.n_next], pool, r.wl);
                Nq = ot_div(r.n_cur, n_next);
            }
            if (hwh) {
                if (IDEAL_FILTER && kind == OT_STEP_IDEAL)
                    refract_ideal<POL>(sf, st, r, pn, npx, npy, npz);
                else
                    tir = refract<POL, LEVEL>(sf, r, pn, wn, npx, npy, npz, r.n_cur, n_next, Nq, pc);
            }
        } else if (IDEAL_FILTER && kind == OT_STEP_FILTER) {  // raytracer.py:379-380
            if (hwh) {
                double T = (SPEC == 2) ? lrow[(3 * i + 2) * OT_MAX_LINES] : filter_T<TAB>(filters[st.filter], pool, r.wl);
                wn = (float)((double)r.w * T);
            }
        } else {  // aperture raytracer.py:381-386
            if (hwh) wn = 0.f;
            if (FULL && st.hurb) {
                double za, zb;
                if (TAB && hurb_normals) {
                    za = (hurb_normals + (2 * (int64_t)st.hurb_slot + 0) * R.N)[ray];
                    zb = (hurb_normals + (2 * (int64_t)st.hurb_slot + 1) * R.N)[ray];
                } else {
                    philox_normal2(seed, ray, 0x48555242u, (uint32_t)st.hurb_slot, za, zb);
                }
                neg = hurb_bend<POL>(sc, sf, r, pn, wn, npx, npy, npz, hwnh, za, zb);
            }
            if (FULL) count_event(msgs, nt, OT_INFO_HURB_NEG_DIR, i + 1, neg);
        }
        // rays that missed the surface may leave the outline box on their way (raytracer.py:338, 367, 389)
        if (hwnh) clip = outline_clip(sc.outline, r.p, r.s, pn, wn);
        // the rare events of a step are counted behind one wave-wide test (a surface that every ray of the wave hits
        // and passes has none)
        if (__ballot(hwnh || tir) != 0ull) {
            if (kind <= OT_STEP_IDEAL) {
                count_event(msgs, nt, OT_INFO_ABSORB_MISSING, i + 1, hwnh);
                count_event(msgs, nt, OT_INFO_TIR, i, tir);
            }
            count_event(msgs, nt, OT_INFO_OUTLINE_INTERSECTION, i, clip);
        }

        r.p = pn;
        r.w = wn;
        r.polx = npx;
        r.poly = npy;
        r.polz = npz;
        r.n_cur = n_next;
#ifdef OT_NO_SECTION_STORES  // experiment (tools/power_clock.py): the same arithmetic, only the last section is stored
        if (i + 1 == sc.n_steps)
#endif
#if OT_RUNNING_BASES
        if (!TAIL) store_section_next<POL>(planes, o8, o4, r.p, r.w, r.n_cur, r.polx, r.poly, r.polz);
#else
        if (!TAIL) store_section<POL>(R, o8, o4, i + 1, r.p, r.w, r.n_cur, r.polx, r.poly, r.polz);
#endif
    }
    if (!TAIL) {
        const int64_t N = R.N;
        store_f64(R.s, o8, r.s.x);
        store_f64(R.s + N, o8, r.s.y);
        store_f64(R.s + 2 * N, o8, r.s.z);
    }
    return ok;
}
