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

// wave-aggregated event counter: the section index is wave-uniform, so one atomic per wave and event
OT_DEV void count_event(unsigned long long* msgs, int nt, int info, int sec, bool cond) {
    unsigned long long m = __ballot(cond);
    if (m != 0ull) {
        int lane = __lane_id();
        if (lane == (int)__ffsll((long long)m) - 1) atomicAdd(&msgs[info * nt + sec], (unsigned long long)__popcll(m));
    }
}

struct RayState {
    V3 p, s;
    float w, wl;
    float polx, poly, polz;
    double n_cur;
};

// Raytracer.__compute_polarization raytracer.py:831-879 (hwh == true for this lane)
template <bool POL>
OT_DEV void compute_polarization(const V3& s, const V3& s_, const RayState& r, float& npx, float& npy, float& npz,
                                 double& A_ts, double& A_tp) {
    const double inv_sqrt2 = 1 / sqrt(2.0);
    if (!POL) {
        A_ts = inv_sqrt2;
        A_tp = inv_sqrt2;
        return;
    }
    bool mask = (s.x != s_.x) || (s.y != s_.y) || (s.z != s_.z);
    V3 ps = normalize3(cross3(s_, s));
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

// Raytracer.__refraction raytracer.py:761-829 for a lane that has power and hit the surface.
// Returns true on total internal reflection.
template <bool POL>
OT_DEV bool refract(const SurfDev& sf, RayState& r, const V3& pn, float& wn, float& npx, float& npy, float& npz,
                    double n1, double n2) {
    V3 n = surf_normal(sf, pn.x, pn.y);
    V3 s = r.s;
    double ns = dot3(n, s);
    double N = n1 / n2;
    double W = sqrt(1 - N * N * (1 - ns * ns));
    double q = N * ns - W;
    V3 s_ = {s.x * N - n.x * q, s.y * N - n.y * q, s.z * N - n.z * q};

    double A_ts, A_tp;
    compute_polarization<POL>(s, s_, r, npx, npy, npz, A_ts, A_tp);

    double n1ca = n1 * ns;
    double n2cb = n2 * W;
    double ts = 2 * n1ca / (n1ca + n2cb);
    double tp = 2 * n1ca / (n2 * ns + n1 * W);
    double a = A_ts * ts, b = A_tp * tp;
    double T = n2cb / n1ca * (a * a + b * b);
    bool tir = !isfinite(W);
    if (tir) T = 0;
    wn = (float)((double)r.w * T);
    r.s = s_;
    return tir;
}

// Raytracer.__refraction_ideal_lens raytracer.py:720-759
template <bool POL>
OT_DEV void refract_ideal(const SurfDev& sf, const ElemDev& el, RayState& r, const V3& pn, float& npx, float& npy,
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
OT_DEV bool outline_clip(const double* __restrict__ o, const V3& p, const V3& s, V3& pn, float& wn) {
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
template <bool POL>
OT_DEV bool hurb_bend(const SceneDev& sc, const SurfDev& sf, RayState& r, const V3& pn, float& wn, float& npx,
                      float& npy, float& npz, bool hwnh, double za, double zb) {
    double a_, b_;
    V3 b;
    bool inside;
    hurb_props(sf, pn.x, pn.y, a_, b_, b, inside);
    bool bend = hwnh && inside;
    V3 a = {-b.y, b.x, b.z};
    V3 s = r.s;
    double sa_ = dot3(s, a), sb_ = dot3(s, b);
    double cos_psi_a = sqrt(1 - sa_ * sa_);
    double cos_psi_b = sqrt(1 - sb_ * sb_);
    float wlm = r.wl * (float)1e-9;  // float32 product in the reference (wl_list is float32)
    double k = 2 * M_PI * r.n_cur / (double)wlm;
    double tan_sig_b = sc.hurb_factor / (2 * b_ * cos_psi_b * 1e-3 * k);
    double tan_sig_a = sc.hurb_factor / (2 * a_ * cos_psi_a * 1e-3 * k);
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

template <bool POL>
OT_DEV void store_section(const ot_rays& R, int64_t ray, int sec, const V3& p, float w, double n, float px, float py,
                          float pz) {
    const int64_t N = R.N;
    const int64_t nt = R.nt;
    R.p[ray + N * (sec)] = p.x;
    R.p[ray + N * (sec + nt)] = p.y;
    R.p[ray + N * (sec + 2 * nt)] = p.z;
    R.w[ray + N * sec] = w;
    R.n[ray + N * sec] = n;
    if (POL) {
        R.pol[ray + N * (sec)] = px;
        R.pol[ray + N * (sec + nt)] = py;
        R.pol[ray + N * (sec + 2 * nt)] = pz;
    }
}

// sub_trace raytracer.py:297-397 for one ray whose section 0 state is in `r`
template <bool POL>
OT_DEV bool trace_ray(const SceneDev& sc, const ot_rays& R, int64_t ray, RayState& r, const double* __restrict__ hurb_normals,
                      uint64_t seed, unsigned long long* msgs) {
    const int nt = sc.nt;
    bool ok = true;
    int i = 0;
    r.n_cur = medium_n(sc.media[sc.n0], sc.pool, r.wl);
    store_section<POL>(R, ray, 0, r.p, r.w, r.n_cur, r.polx, r.poly, r.polz);

    for (int en = 0; en < sc.n_elements; en++) {
        const ElemDev& el = sc.elements[en];
        const SurfDev& front = sc.surfaces[el.front];
        V3 pn = r.p;
        float wn = r.w;
        float npx = r.polx, npy = r.poly, npz = r.polz;
        bool hw = r.w > 0;
        V3 ph;
        bool hit = false, ill = false;

        if (el.kind == OT_EL_LENS || el.kind == OT_EL_IDEAL_LENS) {
            if (hw) {
                ok &= find_hit(front, r.p, r.s, ph, hit, ill);
                pn = ph;
                if (!hit) wn = 0.f;
            }
            count_event(msgs, nt, OT_INFO_ILL_COND, i + 1, hw && ill);
            count_event(msgs, nt, OT_INFO_ABSORB_MISSING, i + 1, hw && !hit);
            double n2_l = medium_n(sc.media[el.n_after], sc.pool, r.wl);

            if (el.kind == OT_EL_LENS) {
                const SurfDev& back = sc.surfaces[el.back];
                double n_l = medium_n(sc.media[el.n_lens], sc.pool, r.wl);
                bool tir = false, clip = false;
                V3 p_prev = r.p;
                if (hw && hit) tir = refract<POL>(front, r, pn, wn, npx, npy, npz, r.n_cur, n_l);
                if (hw && !hit) clip = outline_clip(sc.outline, p_prev, r.s, pn, wn);
                count_event(msgs, nt, OT_INFO_TIR, i, tir);
                count_event(msgs, nt, OT_INFO_OUTLINE_INTERSECTION, i, clip);

                i += 1;
                r.p = pn; r.w = wn; r.polx = npx; r.poly = npy; r.polz = npz; r.n_cur = n_l;
                store_section<POL>(R, ray, i, r.p, r.w, r.n_cur, r.polx, r.poly, r.polz);

                hw = r.w > 0;
                hit = false;
                ill = false;
                if (hw) {
                    ok &= find_hit(back, r.p, r.s, ph, hit, ill);
                    pn = ph;
                    if (!hit) {
                        wn = 0.f;
                        pn = r.p;  // absorbed at the front surface, raytracer.py:354
                    }
                }
                count_event(msgs, nt, OT_INFO_ILL_COND, i + 1, hw && ill);
                count_event(msgs, nt, OT_INFO_ABSORB_MISSING, i + 1, hw && !hit);
                tir = false;
                clip = false;
                p_prev = r.p;
                if (hw && hit) tir = refract<POL>(back, r, pn, wn, npx, npy, npz, n_l, n2_l);
                if (hw && !hit) clip = outline_clip(sc.outline, p_prev, r.s, pn, wn);
                count_event(msgs, nt, OT_INFO_TIR, i, tir);
                count_event(msgs, nt, OT_INFO_OUTLINE_INTERSECTION, i, clip);
            } else {
                bool clip = false;
                V3 p_prev = r.p;
                V3 s_prev = r.s;
                if (hw && hit) refract_ideal<POL>(front, el, r, pn, npx, npy, npz);
                if (hw && !hit) clip = outline_clip(sc.outline, p_prev, s_prev, pn, wn);
                count_event(msgs, nt, OT_INFO_OUTLINE_INTERSECTION, i, clip);
            }
            r.n_cur = n2_l;
        } else {
            if (hw) {
                ok &= find_hit(front, r.p, r.s, ph, hit, ill);
                pn = ph;
            }
            count_event(msgs, nt, OT_INFO_ILL_COND, i + 1, hw && ill);
            bool hwh = hw && hit, hwnh = hw && !hit;
            bool neg = false;
            if (el.kind == OT_EL_FILTER) {
                if (hwh) wn = (float)((double)r.w * filter_T(sc.filters[el.filter], sc.pool, r.wl));
            } else {
                if (hwh) wn = 0.f;
                if (el.hurb) {
                    double za, zb;
                    if (hurb_normals) {
                        za = hurb_normals[(2 * (int64_t)el.hurb_slot + 0) * R.N + ray];
                        zb = hurb_normals[(2 * (int64_t)el.hurb_slot + 1) * R.N + ray];
                    } else {
                        philox_normal2(seed, (uint64_t)ray, 0x48555242u, (uint32_t)el.hurb_slot, za, zb);
                    }
                    neg = hurb_bend<POL>(sc, front, r, pn, wn, npx, npy, npz, hwnh, za, zb);
                }
            }
            count_event(msgs, nt, OT_INFO_HURB_NEG_DIR, i + 1, neg);
            bool clip = false;
            if (hwnh) clip = outline_clip(sc.outline, r.p, r.s, pn, wn);
            count_event(msgs, nt, OT_INFO_OUTLINE_INTERSECTION, i, clip);
        }
        i += 1;
        r.p = pn; r.w = wn; r.polx = npx; r.poly = npy; r.polz = npz;
        store_section<POL>(R, ray, i, r.p, r.w, r.n_cur, r.polx, r.poly, r.polz);
    }
    const int64_t N = R.N;
    R.s[ray] = r.s.x;
    R.s[ray + N] = r.s.y;
    R.s[ray + 2 * N] = r.s.z;
    return ok;
}
