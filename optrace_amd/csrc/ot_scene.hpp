// Device-side scene tables of the tracing kernels (gfx950).
//
// `ot_scene_create` turns the plain descriptors of include/optrace_amd.h into these "compiled" records:
// every per-surface scalar sub-expression of the reference's NumPy code (e.g. (r + N_EPS)**2, 1/rho,
// (k+1)*rho**2) is evaluated ONCE on the host with the same libm the reference uses, so that the per-ray
// device code only contains IEEE +,-,*,/ and sqrt in the reference's operation order.  The records are
// read through wave-uniform (scalar, SGPR) loads: every lane of a wavefront is at the same element of the
// same scene, so a scalar broadcast beats an LDS copy (no LDS traffic, no bank conflicts, no barrier).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/optrace_amd.h"

struct SurfDev {
    int32_t kind;
    int32_t ncoeff;
    int32_t flat;  // z_max == z_min (surface.py:47)
    int32_t rot;   // RECT/SLIT: angle != 0
    double px, py, pz;
    // masks
    double r_eps2;   // (r + N_EPS)**2           surface.py:245
    double ri_eps2;  // (ri - N_EPS)**2          ring_surface.py:133
    double ox_lo, ox_hi, oy_lo, oy_hi;  // rect outer bounds incl. eps   rectangular_surface.py:112
    double ix_lo, ix_hi, iy_lo, iy_hi;  // slit inner bounds incl. eps   slit_surface.py:100
    double cna, sna;                    // cos(-angle), sin(-angle)     surface.py:432
    // conic / asphere
    double k, k1, rho, nrho, rho2, k1rho2, krho2, inv_rho, two_inv_rho;
    double z_min, z_max;      // Surface.z_min / z_max
    double z_lo, z_hi;        // z_min - N_EPS, z_max + N_EPS          conic_surface.py:162
    double z_beh;             // z_max + N_EPS                          surface.py:460
    double zt1, zt2;          // z_min - C_EPS/10, z_max + C_EPS/10     surface.py:331-332
    double edge_val;          // value outside the mask                 surface.py:162
    double r_edge;            // r - N_EPS                              surface.py:153
    // hurb
    double ri;                // ring inner radius
    double hdx, hdy;          // dimi[0]/2, dimi[1]/2
    double cpa, spa;          // cos(angle), sin(angle)                 slit_surface.py:83-84
    double coeff[OT_MAX_ASPH];   // a2, a4, ...
    double dcoeff[OT_MAX_ASPH];  // a_j * (2j+2): np.polyder coefficients
    // tilted plane: unit normal and the slopes -n_x/n_z, -n_y/n_z      tilted_surface.py:69-72
    double nx, ny, nz, mx, my;
    // spline surfaces (DATA1D / DATA2D): sign, centre offset, knots per unit length (interval guess),
    // device pointer to the tables laid out as include/optrace_amd.h describes, knots per dimension
    double sgn, offs, inv_h;
    // the equidistant part of the knots (ot_spline.hpp::UniformKnots): first such knot, spacing and its inverse, clamp range
    double ku_t0, ku_h, ku_inv_h, ku_lo, ku_hi;
    const double* tab;
    int32_t nk, deriv_unrot;
    // mask bitmap of a function surface (OT_SURF_FLAG_MASK_TABLE): offset of the words in `tab` (in doubles), cells per
    // dimension (0 = no bitmap), radius and cells per unit length
    int64_t mask_off;
    int32_t mask_n, mask_pad;
    double mask_r, mask_scale;
};

// One step per tracing surface, in ray order (the element list of raytracer.py:492-508 flattened).
#define OT_STEP_LENS_FRONT 0  // raytracer.py:316-338  medium behind = lens material
#define OT_STEP_LENS_BACK 1   // raytracer.py:340-367  medium behind = n2 or n0
#define OT_STEP_IDEAL 2       // raytracer.py:360-363
#define OT_STEP_FILTER 3      // raytracer.py:379-380
#define OT_STEP_APERTURE 4    // raytracer.py:381-386

struct StepDev {
    int32_t kind, surf, n_next, filter, hurb, hurb_slot, _pad0, _pad1;
    double f;      // ideal lens: 1000 / D          raytracer.py:748
    double fsign;  // ideal lens: np.sign(f)        raytracer.py:755
};

struct FilterDev {
    int32_t type, inverse, tab_len, _pad;
    int64_t tab_off;
    double val, wl0, wl1;
    float mu32, den32, val32;  // Gaussian evaluated in float32: mu, 2*sig**2, val
    float _pad2;
};

struct SceneDev {
    double outline[6];
    int32_t n_surfaces, n_steps, n_media, n_filters;
    int32_t n0, no_pol, use_hurb, nt;
    int32_t n_hurb, _pad;
    double hurb_factor;
    const SurfDev* surfaces;
    const StepDev* steps;
    const ot_medium* media;
    const FilterDev* filters;
    const double* pool;
    int64_t pool_len;
    // discrete-spectrum tables (n_lines > 0): lines[OT_MAX_LINES] as float32 values, then per step i the rows
    // (n_next, n1/n2, filter T) and finally the ambient row n0: doubles at line_tab[row * OT_MAX_LINES + j]
    int32_t n_lines, _pad2;
    const double* line_tab;
};

struct ot_scene {
    SceneDev h;        // host copy of the header (pointers are device pointers)
    SceneDev* d;       // device copy of the header
    void* blob;        // one device allocation holding all tables
    int device;
    unsigned int* cnt_slots;  // OT_CNT_SLOTS x (5*nt+1) counter slot tables (zero between launches)
    bool needs_full;   // the "full" bit of the kernel variants: HURB present; at hit level 0 also ideal lenses / filters
    int hit_level;     // OT_HIT_CLOSED / _ILLINOIS (aspheres, tilted) / _SPLINE (data surfaces, mask bitmaps), ot_device.hpp
    bool needs_tables; // some medium / filter is tabulated (DATA / LINES): per-lane global loads in the loop
    // synchronous entry points (ot_generate_and_trace_host): the counter reduction writes straight into this pinned,
    // device-mapped host buffer (5*nt + 1 words), so a trace costs one wait and no device-to-host copy
    unsigned long long* pin_msgs = nullptr;
    // optional kernel timing (ot_scene_set_timing): events on the launch stream right around the trace kernel
    bool timing = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool ev_valid = false;
};

// ---- sources -------------------------------------------------------------------------------------------
// Start hint for an inverse-CDF search: the range [x0, x0 + K / scale] of the cumulative table is cut into K
// buckets, g[b] = largest index whose cumulative value is <= the lower edge of bucket b.  The search walks from
// there (both directions, so the result never depends on the hint): ~1 load instead of log2(n) dependent ones.
struct CdfGuide {
    const int32_t* g;
    int32_t K, _pad;
    double x0, scale;
};

struct SourceDev {
    int32_t shape, divergence, div_2d, orientation, polarization, spectrum, img_w, img_h;
    double pos[3];
    double r, ri, dim[2];
    double ca, sa;          // cos/sin of the rectangle / line rotation
    double div_sin;         // sin(radians(div_angle))          ray_source.py:303
    double div_rad;         // radians(div_angle)
    double div_axis;        // radians(div_axis_angle)
    double s[3], conv_pos[3];
    double pol_angle, pol_cos, pol_sin;  // constant polarisation angle and its cos / sin
    double axis_cos, axis_sin;          // cos / sin of div_axis (2-D divergence)
    double px_w, px_h;                  // image sources: pixel size dim / (img_w, img_h)
    double inv_img_w;                   // 1 / img_w (the pixel's row without an integer division)
    double wl, wl0, wl1, mu, sig;
    double gauss_xl, gauss_xr;  // truncated-normal cdf bounds  light_spectrum.py:117-118
    double power;
    const double* spec_tab; int64_t n_spec;
    const double* pol_tab;  int64_t n_pol;
    const double* div_tab;  int64_t n_div;
    // image sources (ot_generate.hpp::pixel_pick): one 32-byte record per pixel {cumulative pdf F_j, cumulative primary
    // mix r / (r+g+b), (r+g) / (r+g+b), 0}, and per bucket b of [0, F_total) the number of pixels whose bucket lies
    // before b (pick_lo[0 .. pick_K], buckets by the device's own expression (int)(X * pick_scale)): a pixel is found
    // with two memory round trips, the second one already carrying its colours
    const double* pix_rec;
    const int32_t* pick_lo;
    int32_t pick_K, _pad3;
    double pick_scale, pix_total;
    // wavelength of an sRGB primary as a function of the uniform variable: OT_PRIM_M + 1 samples of the inverse cumulative
    // spectrum per primary (3 tables), linear in between -- one round trip, no search
    const double* prim_inv;
    CdfGuide g_spec, g_pol, g_div;
    // the continuous inverse-CDF tables once more as (F_j, x_j) pairs: the two nodes an interpolation needs sit in
    // one 32-byte load, which also decides whether the search has to step (see inv_cdf_linear)
    const double *spec_pairs, *pol_pairs, *div_pairs;
    const double* s_or; int64_t n_or;  // OR_ARRAY: caller-owned base orientations x[n_or] | y[n_or] | z[n_or]
    // the same base orientation for every ray (constant orientation; point source converging onto conv_pos): it is in
    // `s` and its divergence frame in fx, fy, all evaluated once on the host instead of once per ray
    int32_t frame_uniform, _pad2;
    double fx[3], fy[3];
};

struct ot_sources {
    SourceDev* d;
    int32_t n;
    void* blob;
    int device;
    int64_t* n_or;  // host, per source: rays a range of this source must hold (OR_ARRAY with an array), else -1
    double* power;  // host, per source
    bool has_image = false;  // some source is an image (those scenes take the formula kernels, see generate_ray<IMAGES>)
    struct RangeCache* rcache = nullptr;  // the last range list seen by make_ranges and what was derived from it
};

#define OT_PRIM_N 5000    // wavelengths the reference tabulates the primaries on (srgb.py:528)
#define OT_PRIM_M 65536   // buckets of the inverse tables (SourceDev::prim_inv)
