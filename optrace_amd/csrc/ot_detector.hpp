// Detector stage: hit search over the stored ray sections, sphere projection, and XYZ binning.
// Reference: Raytracer._hit_detector raytracer.py:881-1051, SphericalSurface.sphere_projection
// spherical_surface.py:36-97, RenderImage.render render_image.py:361-421 (+ misc.binning_indices_2d
// misc.py:59-91, color.x/y/z_observer observers.py:14-41).
#pragma once
#include <vector>
#include "ot_device.hpp"
#include "cie_observer_table.inc"

// ---- double min / max atomics (no native f64 min/max on global memory: CAS loop, one lane per wave) -------
OT_DEV void atomic_min_f64(double* addr, double v) {
    unsigned long long* a = (unsigned long long*)addr;
    unsigned long long old = *a;
    while (v < __longlong_as_double((long long)old)) {
        unsigned long long assumed = old;
        old = atomicCAS(a, assumed, (unsigned long long)__double_as_longlong(v));
        if (old == assumed) break;
    }
}

OT_DEV void atomic_max_f64(double* addr, double v) {
    unsigned long long* a = (unsigned long long*)addr;
    unsigned long long old = *a;
    while (v > __longlong_as_double((long long)old)) {
        unsigned long long assumed = old;
        old = atomicCAS(a, assumed, (unsigned long long)__double_as_longlong(v));
        if (old == assumed) break;
    }
}

// Order-preserving map double -> uint64 (and back): min / max of doubles become single hardware integer atomics
// (global_atomic_umin_x2 / umax_x2) instead of compare-and-swap loops on a plain, possibly stale, load.
OT_HD unsigned long long f64_to_ordered(double d) {
    unsigned long long u;
    __builtin_memcpy(&u, &d, 8);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}

OT_HD double ordered_to_f64(unsigned long long u) {
    u = (u >> 63) ? (u & 0x7fffffffffffffffull) : ~u;
    double d;
    __builtin_memcpy(&d, &u, 8);
    return d;
}

#define OT_EXT_SLOTS 64  // extent slot tables: workgroup b updates table b % 64 (four ordered values each)

__global__ void extent_init_kernel(unsigned long long* __restrict__ slots) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;  // 4 * OT_EXT_SLOTS threads per slot table set
    slots[i] = (i & 1) ? 0ull : ~0ull;  // max entries start at the smallest, min entries at the largest code
}

// extent4 = combine(extent4, all slot tables): x_min, x_max, y_min, y_max
__global__ void extent_final_kernel(const unsigned long long* __restrict__ slots, double* __restrict__ extent4) {
    const int c = threadIdx.x;
    if (c >= 4) return;
    double v = extent4[c];
    for (int k = 0; k < OT_EXT_SLOTS; k++) {
        const unsigned long long u = slots[4 * k + c];
        if (u == ((c & 1) ? 0ull : ~0ull)) continue;  // untouched
        const double d = ordered_to_f64(u);
        v = (c & 1) ? fmax(v, d) : fmin(v, d);
    }
    extent4[c] = v;
}

OT_DEV double wave_min(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o));
    return v;
}

OT_DEV double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
}

// Sum over the wave, delivered in lane 63 (row-wise DPP steps: no LDS traffic, unlike the shuffles above).
#define OT_DPP_ADD(x, CTRL, ROWS)                                                                  \
    do {                                                                                           \
        const int lo_ = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, ROWS, 0xF, false); \
        const int hi_ = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, ROWS, 0xF, false); \
        x += __hiloint2double(hi_, lo_);                                                           \
    } while (0)
OT_DEV double wave_sum_in_last_lane(double x) {
    OT_DPP_ADD(x, 0xB1, 0xF);   // quad_perm [1, 0, 3, 2]
    OT_DPP_ADD(x, 0x4E, 0xF);   // quad_perm [2, 3, 0, 1]
    OT_DPP_ADD(x, 0x141, 0xF);  // row_half_mirror
    OT_DPP_ADD(x, 0x140, 0xF);  // row_mirror: every lane holds its row's sum
    OT_DPP_ADD(x, 0x142, 0xA);  // row_bcast15 into rows 1 and 3
    OT_DPP_ADD(x, 0x143, 0xC);  // row_bcast31 into rows 2 and 3
    return x;
}

// Four values per lane, to be added under a key (a pixel): lanes of the wave that share a key are summed first and ONE lane
// adds for them.  Images of point-like objects -- the point spread functions this path serves -- send the 64 hits of a wave
// into one or two pixels: as LDS atomics on one address they are carried out one after the other, and the LDS pipe of the CU
// was what the binning of such images waited for (BASELINE config 2, 1e7 rays into five spots: 0.23 ms, 2.4 TB/s).  Up to
// three keys per call are treated this way, a key with fewer than eight lanes is not worth the 72 vector instructions of the
// four sums; whoever is left adds for himself.  All lanes of the wave must call (valid = false: nothing to add).
template <class ADD>
OT_DEV void wave_add4_by_key(bool valid, int key, double v0, double v1, double v2, double v3, ADD add) {
    const int lane = __lane_id();
    unsigned long long todo = __ballot(valid), self = 0ull;
    for (int round = 0; round < 3 && todo; round++) {
        const int leader = (int)__ffsll((long long)todo) - 1;
        const int k = __shfl(key, leader);
        const unsigned long long m = __ballot(valid && key == k) & todo;
        todo &= ~m;
        if (__popcll(m) < 8) {
            self |= m;
            continue;
        }
        const bool in = (m >> lane) & 1ull;
        const double a0 = wave_sum_in_last_lane(in ? v0 : 0.0), a1 = wave_sum_in_last_lane(in ? v1 : 0.0);
        const double a2 = wave_sum_in_last_lane(in ? v2 : 0.0), a3 = wave_sum_in_last_lane(in ? v3 : 0.0);
        if (lane == 63) add(k, a0, a1, a2, a3);
    }
    if (((todo | self) >> lane) & 1ull) add(key, v0, v1, v2, v3);
}

// SphericalSurface.sphere_projection spherical_surface.py:36-97
OT_DEV void sphere_project(double x0, double y0, double z0, double R, int projection, V3& p) {
    if (projection == OT_PROJ_NONE || projection == OT_PROJ_ORTHOGRAPHIC) return;
    double zm = z0 + R;
    double sgnR = (R > 0) - (R < 0);
    double dx = p.x - x0, dy = p.y - y0;
    if (projection == OT_PROJ_EQUAL_AREA) {
        double aR = fabs(R);
        double x_ = dx / aR, y_ = dy / aR, z_ = (p.z - zm) / R;
        double f = sqrt(2 / (1 - z_));
        p.x = f * x_;
        p.y = f * y_;
        return;
    }
    double r = sqrt(dx * dx + dy * dy);
    double c = (r > 0.0) ? dx / r : 1.0;  // cos(atan2(dy, dx))
    double s = (r > 0.0) ? dy / r : 0.0;  // sin(atan2(dy, dx))
    double q;
    if (projection == OT_PROJ_EQUIDISTANT) {
        q = -sgnR * atan(r / (p.z - zm));
    } else {
        double theta = M_PI / 2 - atan(r / (p.z - zm));
        q = -2 * sgnR * tan(M_PI / 4 - theta / 2);
    }
    p.x = q * c;
    p.y = q * s;
}

__global__ __launch_bounds__(256) void projection_kernel(double x0, double y0, double z0, double R, int projection,
                                                         int64_t n, const double* __restrict__ p, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    V3 v = {p[i], p[i + n], p[i + 2 * n]};
    sphere_project(x0, y0, z0, R, projection, v);
    out[i] = v.x;
    out[i + n] = v.y;
    out[i + 2 * n] = v.z;
}

// direction of section k re-derived from the stored positions (RayStorage.rays_by_mask ray_storage.py:274-279)
OT_DEV V3 section_dir(const ot_rays& R, int64_t r, int k) {
    const int64_t N = R.N, nt = R.nt;
    int k1 = (k < R.nt - 1) ? k + 1 : k;
    V3 d = {R.p[r + N * k1] - R.p[r + N * k], R.p[r + N * (k1 + nt)] - R.p[r + N * (k + nt)],
            R.p[r + N * (k1 + 2 * nt)] - R.p[r + N * (k + 2 * nt)]};
    return normalize3(d);
}

struct Crop {
    double x0, x1, y0, y1;
    int on;
};

// One detector of a launch.  Several positions of `iterative_render` (raytracer.py:1235-1267) are intersected in one
// pass over the ray sections: the sections are read once, every detector writes its own hits.
#define OT_DET_MAX 8
struct DetOne {
    SurfDev det;
    double Rcurv;
    Crop crop;
    double* ph;                     // (count, 3) F-order projected hits
    float* hw;                      // (count) weights, 0 = no valid hit
    unsigned long long* ext_slots;  // extent slot tables or null
    unsigned long long* ill;        // [2] ill-conditioned, timed out
    int projection;
    int xy_only;  // the z plane of ph is not wanted (binning and spectra use x, y only)
    // compact hit list (ot_detector_req.fill): valid hits only, gathered at the front of the list's pieces
    float* wl_out;
    unsigned int* fill;
    int piece_shift;
};

#define OT_HIT_PIECES_N 1024
// entries per piece of a compact hit list of capacity n: the power of two at or above n / 1024 (at least 1024), so that a
// ray's piece is a shift of its index; the last pieces of the 1024 stay empty
__host__ __device__ static inline int hit_piece_shift(int64_t n) {
    int s = 10;
    while (((int64_t)OT_HIT_PIECES_N << s) < n) s++;
    return s;
}
__host__ __device__ static inline int64_t hit_piece_len(int64_t n) { return (int64_t)1 << hit_piece_shift(n); }

// Raytracer._hit_detector raytracer.py:922-1051, one lane per ray of [first, first+count), n_det detectors.
// ill[0] += ill-conditioned rays, ill[1] += rays whose numeric hit search timed out.
// NUMERIC = false: detectors with a closed-form hit (flat, conic) -- without the Illinois loop and the spline code the
// kernel needs a third of the registers, and this kernel lives on loads in flight.
struct SectionPair {  // the last two sections of a ray and the weight of the last but one
    double zl, zq, xl, xq, yl, yq;
    float wq;
};

// direction of the last but one section, re-derived from the stored positions (ray_storage.py:274-279)
OT_DEV V3 pair_direction(const SectionPair& sp) {
    V3 d = {sp.xl - sp.xq, sp.yl - sp.yq, sp.zl - sp.zq};
    return normalize3(d);
}

// The usual place of a detector is behind the last surface: the ray's last section starts before it and ends behind it.
// That case, for a flat detector, settled from the prefetched pair of sections without the section search of
// detector_hit (same arithmetic, same results: the search would pick this section, raytracer.py:929-985); rays that
// end before the detector are settled too (no hit).  Returns false for a lane that needs the general search.
// CONIC: conic detectors as well (else flat ones only: the fused kernels, which keep the conic code out of their tile
// loops); CROP = false leaves the user extent to the caller (who projects first).
template <bool CONIC = false, bool CROP = true, class DET>
OT_DEV bool detector_hit_last(DET& D, int nt, bool active, const SectionPair& sp, const V3& sdir, V3& ph, float& w,
                              bool& valid) {
    const auto& det = D.det;
    valid = false;
    w = 0.f;
    ph = {0.0, 0.0, 0.0};
    if (!active) return true;
    const bool reaches = sp.zl >= det.z_min;
    if (!reaches && !(sp.zl >= det.z_max)) return true;  // np.all(~bh_zmin & ~bh_zmax): ends before the detector
    if (!((CONIC || det.flat) && nt >= 2 && reaches && !(sp.zq >= det.z_min))) return false;
    const V3 p = {sp.xq, sp.yq, sp.zq};
    bool ish, ill;
    find_hit<OT_HIT_CLOSED>(det, p, sdir, ph, ish, ill);  // (closed forms report no ill-conditioned hits)
    w = (ph.z > sp.zl + OT_C_EPS) ? 0.f : sp.wq;  // a hit behind the end of the ray is none (raytracer.py:985)
    valid = ish && (w > 0);
    if (CROP && D.crop.on) valid = valid && D.crop.x0 <= ph.x && ph.x <= D.crop.x1 && D.crop.y0 <= ph.y && ph.y <= D.crop.y1;
    return true;
}

// A storage of TWO sections (the tail storage of render-only traces, ot_trace_kernel.hpp::trace_tail_kernel): everything
// detector_hit can find out it finds in the prefetched pair -- its section search has one section to choose, its loop one
// round (raytracer.py:929-985 with nt = 2) -- so the hit is settled without a load and without the search code: the ray
// ends before the detector (np.all(~bh_zmin & ~bh_zmax)) or starts behind it (np.all(bh_zmin & bh_zmax)): no hit; otherwise
// the intersection with the one section, void if it lies behind the section's end.  Detectors with a closed-form hit (flat,
// conic), no sphere projection.  Same values as detector_hit<false, false> on such a storage.
template <class DET>
OT_DEV void detector_hit_pair(DET& D, bool active, const SectionPair& sp, const V3& sdir, V3& ph, float& w, bool& valid) {
    const auto& det = D.det;
    valid = false;
    w = 0.f;
    ph = {0.0, 0.0, 0.0};
    if (!active) return;
    const bool ends_before = !(sp.zl >= det.z_min) && !(sp.zl >= det.z_max);
    const bool starts_behind = sp.zl >= det.z_min && sp.zq >= det.z_min && sp.zq >= det.z_max;
    if (ends_before || starts_behind) return;
    const V3 p = {sp.xq, sp.yq, sp.zq};
    bool ish, ill;
    find_hit<OT_HIT_CLOSED>(det, p, sdir, ph, ish, ill);
    w = (ph.z > sp.zl + OT_C_EPS) ? 0.f : sp.wq;  // a hit behind the end of the ray is none (raytracer.py:985)
    valid = ish && (w > 0);
    if (D.crop.on) valid = valid && D.crop.x0 <= ph.x && ph.x <= D.crop.x1 && D.crop.y0 <= ph.y && ph.y <= D.crop.y1;
}

// The hit of one ray on one detector: section search, intersection, projection, user extent.
// -> valid, ph (projected), w; any_ill / timeout report the numeric hit search.
// PROJ = false leaves the sphere projections out (their atan / tan polynomials cost ~64 VGPRs of hoisted constants once
// the hit search sits inside a loop).
template <bool NUMERIC, bool PROJ = true, bool CROP = true, class DET>
OT_DEV void detector_hit(const ot_rays& R, int64_t r, bool active, DET& D, const SectionPair& sp, const V3& sdir, V3& ph,
                         float& w, bool& valid, bool& any_ill, bool& timeout) {
    const int64_t N = R.N;
    const int nt = R.nt;
    const double* __restrict__ zp = R.p + r + N * (2 * (int64_t)nt);
    const double* __restrict__ xp = R.p + r;
    const double* __restrict__ yp = R.p + r + N * (int64_t)nt;
    const int kq = nt >= 2 ? nt - 2 : 0;
    const double zl = sp.zl, zq = sp.zq, xq = sp.xq, yq = sp.yq;
    const float wq = sp.wq;
    const auto& det = D.det;
    ph = {0.0, 0.0, 0.0};
    w = 0.f;
    bool ish = false;
    any_ill = false;
    timeout = false;

    if (active) {
        // section search (raytracer.py:929-938): first section whose start lies at/behind the detector's z_min.
        // Along a traced ray z never decreases (s_z > 0 on every living section, dead rays keep their position),
        // so the reference's whole-row tests reduce to the two end sections and argmax(z >= z_min) to a binary
        // search: 2 + log2(nt) plane reads per ray instead of nt.
        // The planes are read from the far end: a detector behind the last surface (the usual place) is settled by
        // the last two, z[nt-2] < z_min <= z[nt-1].
        bool all_b = false;                                            // np.all(bh_zmin & bh_zmax): starts behind it
        const bool all_nb = !(zl >= det.z_min) && !(zl >= det.z_max);  // np.all(~bh_zmin & ~bh_zmax): ends before
        int first_ge = -1;
        if (!all_nb && zl >= det.z_min) {
            if (nt >= 2 && !(zq >= det.z_min)) {
                first_ge = nt - 1;  // z0 <= z[nt-2] < z_min: not all_b either
            } else {
                const double z0 = zp[0];
                all_b = z0 >= det.z_max && z0 >= det.z_min;
                if (z0 >= det.z_min) {
                    first_ge = 0;
                } else {
                    int lo = 0, hi = kq;  // z[lo] < z_min <= z[hi]
                    while (hi - lo > 1) {
                        int mid = (lo + hi) >> 1;
                        if (zp[N * (int64_t)mid] >= det.z_min)
                            hi = mid;
                        else
                            lo = mid;
                    }
                    first_ge = hi;
                }
            }
        } else if (!all_nb) {  // z_min > zl >= z_max cannot happen for a surface (z_max >= z_min); kept for exactness
            const double z0 = zp[0];
            all_b = z0 >= det.z_max && z0 >= det.z_min;
        }
        if (!(all_b || all_nb)) {
            int k = (first_ge < 0 ? 0 : first_ge) - 1;
            if (k < 0) k = 0;
            V3 p, s;
            if (k == kq && nt >= 2) {  // the prefetched pair of sections
                p = {xq, yq, zq};
                s = sdir;  // = pair_direction(sp): the same for every detector of a launch, formed once per ray
                w = wq;
            } else {
                p = {xp[N * (int64_t)k], yp[N * (int64_t)k], zp[N * (int64_t)k]};
                s = section_dir(R, r, k);
                w = R.w[r + N * k];
            }
            for (;;) {
                k += 1;
                if (k >= nt) {
                    w = 0.f;
                    break;
                }
                bool ill;
                if (!find_hit<(NUMERIC ? OT_HIT_SPLINE : OT_HIT_CLOSED)>(det, p, s, ph, ish, ill)) timeout = true;
                any_ill = any_ill || ill;
                double p2z = (k == nt - 1) ? zl : zp[N * (int64_t)k];
                if (!(ph.z > p2z + OT_C_EPS)) break;  // hit lies inside this section (raytracer.py:985)
                p.x = xp[N * (int64_t)k];
                p.y = yp[N * (int64_t)k];
                p.z = p2z;
                s = section_dir(R, r, k);
                w = R.w[r + N * k];
            }
        }
    }
    valid = active && ish && (w > 0);
    if (PROJ && valid) sphere_project(det.px, det.py, det.pz, D.Rcurv, D.projection, ph);
    // user extent: hits outside are dropped (raytracer.py:1036-1040)
    if (CROP && D.crop.on) valid = valid && D.crop.x0 <= ph.x && ph.x <= D.crop.x1 && D.crop.y0 <= ph.y && ph.y <= D.crop.y1;
}

// What becomes of a ray's hit: the entry of the (compact or dense) hit list, the counters, the extent (step one).
template <class DET>
OT_DEV void detector_emit(const ot_rays& R, int64_t q, int64_t r, bool active, int64_t count, DET& D, const V3& ph, float w,
                          bool valid, bool any_ill, bool timeout, double (*sext)[4][4], int di) {
    const int lane = __lane_id();
    if (D.fill) {
        // Compact list: the wave's valid hits go, in any order, to the next free entries of a piece.  Ranks inside the
        // wave from the ballot, the wave's base from ONE returning atomic on the piece's fill count; no barrier -- the
        // other waves of the SIMD cover the round trip.  Wave k fills piece k mod 1024: the ~7000 waves in flight at any
        // moment then spread their atomics over all 1024 counters.  (With contiguous pieces they all sat in the same two
        // or three pieces and the atomics on those counters serialised: 11.8 ms for the hit kernel of C4 instead of 2.7;
        // one atomic per workgroup behind an LDS count and three barriers: 3.8 ms.)  A piece receives at most
        // ceil(waves / 1024) * 64 <= piece_len hits; the list's capacity is 1024 * piece_len entries, whatever the count.
        const unsigned long long m = __ballot(valid);
        if (m) {
            const int64_t piece = (q >> 6) & (OT_HIT_PIECES_N - 1);
            unsigned int base = 0u;
            if (lane == (int)__ffsll((long long)m) - 1) base = atomicAdd(&D.fill[piece], (unsigned int)__popcll(m));
            base = (unsigned int)__shfl((int)base, (int)__ffsll((long long)m) - 1);
            base += (unsigned int)__popcll(m & ((1ull << lane) - 1ull));
            if (valid) {
                const int64_t i = (piece << D.piece_shift) + base;
                if (D.ph) {  // (weights and wavelengths alone serve the detector spectrum)
                    D.ph[i] = ph.x;
                    D.ph[i + ((int64_t)OT_HIT_PIECES_N << D.piece_shift)] = ph.y;  // the y plane follows the x plane's capacity
                }
                D.hw[i] = w;
                D.wl_out[i] = R.wl[r];
            }
        }
    } else if (active && D.ph) {  // (no hit list wanted: the extent-only pass of images with an automatic extent)
        D.ph[q] = valid ? ph.x : 0.0;
        D.ph[q + count] = valid ? ph.y : 0.0;
        if (!D.xy_only) D.ph[q + 2 * count] = valid ? ph.z : 0.0;
        D.hw[q] = valid ? w : 0.f;
    }
    // counters and extent: wave-level reduction, one atomic per wave
    unsigned long long m_ill = __ballot(any_ill), m_to = __ballot(timeout);
    if (lane == 0) {
        if (m_ill) atomicAdd(&D.ill[0], (unsigned long long)__popcll(m_ill));
        if (m_to) atomicAdd(&D.ill[1], (unsigned long long)__popcll(m_to));
    }
    if (D.ext_slots) {  // extent of the valid hits, step one: wave shuffle -> this detector's row of the LDS table
        const double inf = __builtin_inf();
        double e[4] = {wave_min(valid ? ph.x : inf), wave_max(valid ? ph.x : -inf), wave_min(valid ? ph.y : inf),
                       wave_max(valid ? ph.y : -inf)};
        const int wave = threadIdx.x >> 6;
        if (lane == 0)
            for (int c = 0; c < 4; c++) sext[di][wave][c] = e[c];
    }
}

template <bool NUMERIC, class DET>
OT_DEV void detector_one(const ot_rays& R, int64_t q, int64_t r, bool active, int64_t count, DET& D, const SectionPair& sp,
                         const V3& sdir, double (*sext)[4][4], int di) {
    V3 ph;
    float w = 0.f;
    bool valid = false, any_ill = false, timeout = false;
    // detector with a closed-form hit behind the last surface (the usual case): settled from the prefetched pair of
    // sections; the section search only if a lane of the wave needs it.  Sphere projection and user extent follow either
    // way (ONE copy of the projections' polynomials, outside of the search's loop: with them inside, every ray of a
    // spherical detector took the general path -- BASELINE config 3's retina, 0.92 ms for the hit kernel of 5e7 rays)
    bool settled = false;
    if (!NUMERIC) settled = detector_hit_last<true, false>(D, R.nt, active, sp, sdir, ph, w, valid);
    if (__ballot(!settled) != 0ull) {
        if (!settled) detector_hit<NUMERIC, false, false>(R, r, active, D, sp, sdir, ph, w, valid, any_ill, timeout);
    }
    if (valid) sphere_project(D.det.px, D.det.py, D.det.pz, D.Rcurv, D.projection, ph);
    // user extent: hits outside are dropped (raytracer.py:1036-1040)
    if (D.crop.on) valid = valid && D.crop.x0 <= ph.x && ph.x <= D.crop.x1 && D.crop.y0 <= ph.y && ph.y <= D.crop.y1;
    detector_emit(R, q, r, active, count, D, ph, w, valid, any_ill, timeout, sext, di);
}

// Extent, step two, once per workgroup behind ONE barrier for all detectors of the launch: the four waves' values of
// every (detector, bound) meet and go to one of the slot tables with an order-preserving integer atomic.  (A barrier pair
// per detector inside detector_one made the extent-only pass over six positions 6.2 ms for 6.7e7 rays; the hit search
// itself is 3.)
template <class DETS>
OT_DEV void extent_flush(DETS dets, int n_det, double (*sext)[4][4]) {
    __syncthreads();
    if ((int)threadIdx.x < 4 * n_det) {
        const int di = threadIdx.x >> 2, c = threadIdx.x & 3;
        unsigned long long* slots = dets[di].ext_slots;
        if (slots) {
            const double inf = __builtin_inf();
            double v = sext[di][0][c];
            for (int k = 1; k < 4; k++) v = (c & 1) ? fmax(v, sext[di][k][c]) : fmin(v, sext[di][k][c]);
            if (v == v && v != ((c & 1) ? -inf : inf)) {
                unsigned long long* dst = slots + 4 * (blockIdx.x % OT_EXT_SLOTS) + c;
                if (c & 1)
                    atomicMax(dst, f64_to_ordered(v));
                else
                    atomicMin(dst, f64_to_ordered(v));
            }
        }
    }
}

// The detector passes read every section value once: non-temporal loads (C4 image with a user extent 3.36-3.53 ->
// 3.17-3.26 ms, tools/ab_detector.py).  -DOT_LOADS_PLAIN restores ordinary loads.
#ifndef OT_LOADS_PLAIN
#define OT_STREAM_LOAD(p) __builtin_nontemporal_load(p)
#else
#define OT_STREAM_LOAD(p) (*(p))
#endif
OT_DEV SectionPair load_section_pair(const ot_rays& R, int64_t r, bool active) {
    // everything the usual case needs is requested at once (one memory round trip instead of three)
    SectionPair sp = {0, 0, 0, 0, 0, 0, 0.f};
    if (active) {
        const int64_t N = R.N;
        const int nt = R.nt;
        const double* __restrict__ zp = R.p + r + N * (2 * (int64_t)nt);
        const double* __restrict__ xp = R.p + r;
        const double* __restrict__ yp = R.p + r + N * (int64_t)nt;
        const int kq = nt >= 2 ? nt - 2 : 0;
        // (read once per detector pass: OT_STREAM_LOAD = non-temporal where the A/B favours it)
        sp.zl = OT_STREAM_LOAD(&zp[N * (int64_t)(nt - 1)]), sp.zq = OT_STREAM_LOAD(&zp[N * (int64_t)kq]);
        sp.xl = OT_STREAM_LOAD(&xp[N * (int64_t)(nt - 1)]), sp.xq = OT_STREAM_LOAD(&xp[N * (int64_t)kq]);
        sp.yl = OT_STREAM_LOAD(&yp[N * (int64_t)(nt - 1)]), sp.yq = OT_STREAM_LOAD(&yp[N * (int64_t)kq]);
        sp.wq = OT_STREAM_LOAD(&R.w[r + N * (int64_t)kq]);
    }
    return sp;
}

// one detector, its record in the kernel arguments (scalar registers)
template <bool NUMERIC>
__global__ __launch_bounds__(256) void detector_kernel(ot_rays R, int64_t first, int64_t count, DetOne D) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = q < count;
    const int64_t r = first + (active ? q : 0);
    __shared__ double sext[1][4][4];
    const SectionPair sp = load_section_pair(R, r, active);
    detector_one<NUMERIC>(R, q, r, active, count, D, sp, pair_direction(sp), sext, 0);
    if (D.ext_slots) extent_flush(&D, 1, sext);
}

// several detectors, records in device memory.  The detector loop is unrolled at compile time (NDET = 2, 4, 8; unused
// entries repeat the last record's outputs with a null hit list): as a run-time loop every lane constant of the body is
// hoisted and kept in registers, 157 instead of 77 VGPRs and half the waves in flight.
template <bool NUMERIC, int NDET>
__global__ __launch_bounds__(256) void detector_multi_kernel(ot_rays R, int64_t first, int64_t count,
                                                             const DetOne* __restrict__ dets, int n_det) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = q < count;
    const int64_t r = first + (active ? q : 0);
    __shared__ double sext[OT_DET_MAX][4][4];
    const SectionPair sp = load_section_pair(R, r, active);
    const V3 sdir = pair_direction(sp);
    if constexpr (NUMERIC) {  // the Illinois loop and the spline code do not unroll; rare as detectors (run-time loop)
        for (int di = 0; di < n_det; di++) detector_one<NUMERIC>(R, q, r, active, count, as_const(dets)[di], sp, sdir, sext, di);
    } else {
#pragma unroll
        for (int di = 0; di < NDET; di++)
            if (di < n_det) detector_one<NUMERIC>(R, q, r, active, count, as_const(dets)[di], sp, sdir, sext, di);
    }
    extent_flush(as_const(dets), n_det, sext);  // (a launch without extents pays one barrier)
}

// ---- rendering ---------------------------------------------------------------------------------------------
struct RenderArgs {
    double x0, x1, y0, y1;
    double fx, fy;  // Nx / sx, Ny / sy  (misc.py:75-76)
    int32_t Nx, Ny;
    double ws;      // every hit's weight times this, in f64, before it is added (1: plain; iterative_render: rays_step / N)
};

static const double* observer_table_device() {
    static thread_local const double* tab[64] = {nullptr};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    if (!tab[dev]) {
        // the 471 x 3 table, and behind it the same as (value, difference to the next row) pairs: 471 x 6 (observer_xyz_at6)
        std::vector<double> both((size_t)OT_OBS_N * 9);
        const double* src = (const double*)ot_observer_xyz;
        for (int i = 0; i < OT_OBS_N * 3; i++) both[i] = src[i];
        double* pairs = both.data() + (size_t)OT_OBS_N * 3;
        for (int j = 0; j < OT_OBS_N; j++)
            for (int c = 0; c < 3; c++) {
                pairs[6 * j + 2 * c] = src[3 * j + c];
                pairs[6 * j + 2 * c + 1] = (j + 1 < OT_OBS_N) ? (src[3 * (j + 1) + c] - src[3 * j + c]) / 1.0 : 0.0;  // observers.py:14-41
            }
        double* d = nullptr;
        if (hipMalloc((void**)&d, sizeof(double) * both.size()) != hipSuccess) return nullptr;
        if (hipMemcpy(d, both.data(), sizeof(double) * both.size(), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
        tab[dev] = d;
    }
    return tab[dev];
}

// RenderImage.render render_image.py:396-418.
//   * the 471x3 CIE observer table (11 KB) is staged in LDS once per workgroup: every lane indexes it with its
//     own wavelength;
//   * histogram updates are privatised per workgroup in an LDS hash table (pixel -> 4 x f64, ds_add_f64):
//     images of point-like objects put millions of hits into a few hundred pixels, and global f64 atomics to the
//     same few cache lines serialise in one L2 channel (measured: 90 ms for 10 M hits).  The grid is one
//     1024-thread workgroup per CU, so the final flush issues at most OT_HASH_N x 4 global atomics per CU;
//   * hits that find their hash neighbourhood occupied (wide images) fall back to global_atomic_add_f64, which
//     is the fast path for spread-out addresses.
#define OT_HASH_N 2048      // entries per workgroup (72 KB of LDS with the values)
#define OT_HASH_PROBES 4    // linear probes before falling back to a global atomic
#define OT_HASH_EMPTY (-1)

// misc.binning_indices_2d misc.py:75-89: pixel index iy * Nx + ix of a hit, -1 outside (weight 0 in the reference)
OT_DEV int hit_pixel(const RenderArgs& a, double x, double y, int32_t& ix, int32_t& iy) {
    ix = (int32_t)floor(a.fx * (x - a.x0));
    iy = (int32_t)floor(a.fy * (y - a.y0));
    if (y == a.y1) iy = a.Ny - 1;
    if (x == a.x1) ix = a.Nx - 1;
    if (ix < 0 || iy < 0 || iy >= a.Ny || ix >= a.Nx) return -1;
    return iy * a.Nx + ix;
}

// color.x/y/z_observer observers.py:14-41 = np.interp on the 1 nm CIE grid: the interval index is floor(wl - 360);
// obs: the 471 x 3 table (LDS copy)
OT_DEV void observer_xyz_at(const double* obs, double l, double& xo, double& yo, double& zo) {
    xo = yo = zo = 0.0;
    double u = l - OT_OBS_WL0;
    if (u >= 0.0 && u <= (double)(OT_OBS_N - 1)) {
        int j = (int)floor(u);
        if (j >= OT_OBS_N - 1) {
            xo = obs[3 * (OT_OBS_N - 1)];
            yo = obs[3 * (OT_OBS_N - 1) + 1];
            zo = obs[3 * (OT_OBS_N - 1) + 2];
        } else {
            double t = l - (OT_OBS_WL0 + (double)j);
            const double* f0 = &obs[3 * j];
            xo = (f0[3] - f0[0]) / 1.0 * t + f0[0];
            yo = (f0[4] - f0[1]) / 1.0 * t + f0[1];
            zo = (f0[5] - f0[2]) / 1.0 * t + f0[2];
        }
    }
}

// The same from the table of (value, difference) pairs (observer_table_device: 471 x 6 behind the 471 x 3), for the kernels
// that look a wavelength up per RECORD out of LDS: the three pairs of a row are 48 contiguous, 16-byte aligned bytes -- three
// 16-byte LDS reads instead of six 8-byte ones with a 24-byte row stride -- and the difference np.interp forms per call
// ((f[j + 1] - f[j]) / 1.0, the same f64 subtraction) comes ready: identical bits.
#define OT_OBS6_OFF (OT_OBS_N * 3)
OT_DEV void observer_xyz_at6(const double* obs6, double l, double& xo, double& yo, double& zo) {
    xo = yo = zo = 0.0;
    double u = l - OT_OBS_WL0;
    if (u >= 0.0 && u <= (double)(OT_OBS_N - 1)) {
        int j = (int)floor(u);
        double t = l - (OT_OBS_WL0 + (double)j);
        if (j >= OT_OBS_N - 1) {  // the last node itself
            j = OT_OBS_N - 1;
            t = 0.0;
        }
        const double2* r = (const double2*)(obs6 + 6 * j);
        const double2 a = r[0], b = r[1], c = r[2];
        xo = a.y * t + a.x;
        yo = b.y * t + b.x;
        zo = c.y * t + c.x;
    }
}

__global__ __launch_bounds__(1024) void render_kernel(int64_t n, const double* __restrict__ px, const double* __restrict__ py,
                                                      const float* __restrict__ w, const float* __restrict__ wl, RenderArgs a,
                                                      const double* __restrict__ table, double* __restrict__ hist,
                                                      const int* __restrict__ spread, const unsigned int* __restrict__ fill) {
    if (spread && spread[0]) return;  // the hits cover many pixels: the tile path bins them (ot_render_tiles.hpp)
    __shared__ double obs[OT_OBS_N * 3];
    __shared__ double hval[OT_HASH_N * 4];
    __shared__ int hkey[OT_HASH_N];
    for (int i = threadIdx.x; i < OT_OBS_N * 3; i += blockDim.x) obs[i] = table[i];
    for (int i = threadIdx.x; i < OT_HASH_N; i += blockDim.x) hkey[i] = OT_HASH_EMPTY;
    for (int i = threadIdx.x; i < OT_HASH_N * 4; i += blockDim.x) hval[i] = 0.0;
    __syncthreads();
    // every workgroup takes one contiguous piece of the ray range: rays of one source are neighbours in the
    // storage and land in the same part of the image, so the LDS table of a workgroup sees fewer distinct pixels
    // and fewer workgroups fight over the same global addresses than with an interleaved assignment
    // (compact lists, `fill`: the workgroup walks whole pieces, each up to its fill count)
    const int64_t plen = hit_piece_len(n);
    const int64_t chunk = fill ? plen : ((n + gridDim.x - 1) / gridDim.x + blockDim.x - 1) / blockDim.x * blockDim.x;
    for (int64_t c = blockIdx.x; fill ? c < OT_HIT_PIECES_N : c * chunk < n; c += fill ? gridDim.x : n) {
    int64_t i_end = ((c + 1) * chunk < n) ? (c + 1) * chunk : n;
    if (fill) i_end = c * chunk + (int64_t)fill[c];
    // (every lane of a wave takes part in every round: the sums over lanes with a common pixel need them all)
    auto add = [&](int pix, double a0, double a1, double a2, double a3) {
        // LDS hash insert: claim or match one of OT_HASH_PROBES consecutive entries
        unsigned int h = ((unsigned int)pix * 2654435761u) >> (32 - 11);  // OT_HASH_N = 2^11
        int slot = -1;
#pragma unroll
        for (int pr = 0; pr < OT_HASH_PROBES; pr++) {
            int sidx = (int)((h + pr) & (OT_HASH_N - 1));
            int k = hkey[sidx];
            if (k == OT_HASH_EMPTY) k = atomicCAS(&hkey[sidx], OT_HASH_EMPTY, pix);
            if (k == OT_HASH_EMPTY || k == pix) {
                slot = sidx;
                break;
            }
        }
        double* hv = (slot >= 0) ? &hval[slot * 4] : hist + (int64_t)pix * 4;
        unsafeAtomicAdd(hv + 0, a0);
        unsafeAtomicAdd(hv + 1, a1);
        unsafeAtomicAdd(hv + 2, a2);
        unsafeAtomicAdd(hv + 3, a3);
    };
    const int64_t i_first = c * chunk;
    for (int64_t base = i_first; base < i_end; base += blockDim.x) {
        const int64_t i = base + threadIdx.x;
        bool valid = i < i_end;
        float wi = valid ? w[i] : 0.f;
        valid = valid && (wi > 0.f || wi < 0.f);  // w == 0: adds nothing (and NaN weights are dropped)
        int32_t ix, iy;
        int pix = valid ? hit_pixel(a, px[i], py[i], ix, iy) : -1;
        valid = valid && pix >= 0;
        double xo = 0.0, yo = 0.0, zo = 0.0;
        if (valid) observer_xyz_at(obs, (double)wl[i], xo, yo, zo);
        const double wm = (double)wi * a.ws;
        wave_add4_by_key(valid, pix, xo * wm, yo * wm, zo * wm, 1.0 * wm, add);
    }
    }
    __syncthreads();
    for (int sidx = threadIdx.x; sidx < OT_HASH_N; sidx += blockDim.x) {
        int k = hkey[sidx];
        if (k != OT_HASH_EMPTY) {
            double* hg = hist + (int64_t)k * 4;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                double v = hval[sidx * 4 + c];
                if (v != 0.0) unsafeAtomicAdd(hg + c, v);
            }
        }
    }
}
