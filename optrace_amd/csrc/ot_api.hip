// C-ABI of the MI355X tracing core (include/optrace_amd.h): host entry points + HIP kernels for gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -munsafe-fp-atomics -fPIC -shared (see Makefile).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <algorithm>
#include <memory>
#include <thread>
#include <vector>

#include "ot_detector.hpp"
#include "ot_detector_fused.hpp"
#include "ot_device.hpp"
#include "ot_focus.hpp"
#include "ot_generate.hpp"
#include "ot_image.hpp"
#include "ot_render_tiles.hpp"
#include "ot_scene.hpp"
#include "ot_scratch.hpp"
#include "ot_selftest.hpp"
#include "ot_spectrum.hpp"
#include "ot_trace.hpp"
#include "ot_trace_kernel.hpp"

// ---------------------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------------------
static thread_local std::string g_err;

static int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return fail(OT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));                 \
    } while (0)

extern "C" int ot_abi_version(void) { return OT_ABI_VERSION; }
extern "C" const char* ot_last_error(void) { return g_err.c_str(); }
extern "C" int ot_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return OT_ERR_NO_DEVICE;
    return n;
}

static int require_device() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(OT_ERR_NO_DEVICE, "no HIP device available; this library has no CPU fallback");
    return OT_OK;
}

// ---------------------------------------------------------------------------------------------------------
// scene compilation (host)
// ---------------------------------------------------------------------------------------------------------
static double conic_sag(double rho, double k1rho2, double r2) { return rho * r2 / (1 + std::sqrt(1 - k1rho2 * r2)); }

static int64_t spline_table_len(const ot_surface& s) {
    const int64_t n = s.nknots, nc = n - OT_SPL_K - 1;
    if (s.kind == OT_SURF_DATA1D) return 3 * n;
    if (s.kind == OT_SURF_DATA2D) return n + nc * nc + 2 * (nc - 1) * nc;
    return 0;
}

// cells per dimension of the mask bitmap behind the spline tables (OT_SURF_FLAG_MASK_TABLE), 0 without one, -1 if
// the count stored there is not a whole number in 1 .. 2^15 (2D) or 1 .. 2^20 (1D)
static int64_t mask_table_cells(const ot_surface& s) {
    if (!(s.flags & OT_SURF_FLAG_MASK_TABLE)) return 0;
    const int64_t at = spline_table_len(s);
    if (!s.tab || s.tab_len <= at) return -1;
    const double n = s.tab[at];
    if (!(n >= 1.0 && n <= (s.kind == OT_SURF_DATA1D ? 1048576.0 : 32768.0)) || n != std::floor(n)) return -1;
    return (int64_t)n;
}

static int64_t surface_table_len(const ot_surface& s) {
    int64_t len = spline_table_len(s);
    const int64_t n = mask_table_cells(s);
    if (n > 0) len += 1 + ((s.kind == OT_SURF_DATA1D ? n : n * n) + 63) / 64;
    return len;
}

static int compile_surface(const ot_surface& s, SurfDev& d) {
    std::memset(&d, 0, sizeof(d));
    const double NE = OT_N_EPS_SURF;
    d.kind = s.kind;
    d.ncoeff = s.ncoeff;
    d.flat = (s.z_max == s.z_min);
    d.px = s.pos[0];
    d.py = s.pos[1];
    d.pz = s.pos[2];
    d.z_min = s.z_min;
    d.z_max = s.z_max;
    d.z_lo = s.z_min - NE;
    d.z_hi = s.z_max + NE;
    d.z_beh = s.z_max + NE;
    d.zt1 = s.z_min - OT_C_EPS / 10;
    d.zt2 = s.z_max + OT_C_EPS / 10;
    d.edge_val = s.z_max;
    d.r_edge = s.r - NE;
    switch (s.kind) {
        case OT_SURF_CIRCLE:
            d.r_eps2 = std::pow(s.r + NE, 2.0);
            break;
        case OT_SURF_RING:
            d.r_eps2 = std::pow(s.r + NE, 2.0);
            d.ri_eps2 = std::pow(s.ri - NE, 2.0);
            d.ri = s.ri;
            break;
        case OT_SURF_RECT:
        case OT_SURF_SLIT: {
            d.rot = (s.angle != 0.0);
            d.cna = std::cos(-s.angle);
            d.sna = std::sin(-s.angle);
            d.cpa = std::cos(s.angle);
            d.spa = std::sin(s.angle);
            double xs = -s.dim[0] / 2, xe = s.dim[0] / 2, ys = -s.dim[1] / 2, ye = s.dim[1] / 2;
            d.ox_lo = xs - NE;
            d.ox_hi = xe + NE;
            d.oy_lo = ys - NE;
            d.oy_hi = ye + NE;
            if (s.kind == OT_SURF_SLIT) {
                double xsi = -s.dimi[0] / 2, xei = s.dimi[0] / 2, ysi = -s.dimi[1] / 2, yei = s.dimi[1] / 2;
                d.ix_lo = xsi + NE;
                d.ix_hi = xei - NE;
                d.iy_lo = ysi + NE;
                d.iy_hi = yei - NE;
                d.hdx = s.dimi[0] / 2;
                d.hdy = s.dimi[1] / 2;
            }
            break;
        }
        case OT_SURF_CONIC:
        case OT_SURF_ASPHERE: {
            if (s.R == 0.0 || !std::isfinite(s.R)) return fail(OT_ERR_INVALID, "surface: R must be finite and non-zero");
            if (s.kind == OT_SURF_ASPHERE && (s.ncoeff < 1 || s.ncoeff > OT_MAX_ASPH))
                return fail(OT_ERR_UNSUPPORTED, "asphere: ncoeff out of range");
            d.r_eps2 = std::pow(s.r + NE, 2.0);
            d.k = s.k;
            d.k1 = s.k + 1;
            d.rho = 1 / s.R;
            d.nrho = -d.rho;
            d.rho2 = std::pow(d.rho, 2.0);
            d.k1rho2 = (s.k + 1) * d.rho2;
            d.krho2 = s.k * d.rho2;
            d.inv_rho = 1 / d.rho;
            d.two_inv_rho = 2 / d.rho;
            for (int j = 0; j < s.ncoeff && j < OT_MAX_ASPH; j++) {
                d.coeff[j] = s.coeff[j];
                d.dcoeff[j] = s.coeff[j] * (double)(2 * (j + 1));
            }
            // Surface.values outside the mask: pos_z + _values(r - N_EPS, 0) (surface.py:153-162)
            if (!d.flat) {
                double re = s.r - NE;
                double v;
                if (s.kind == OT_SURF_CONIC) {
                    v = conic_sag(d.rho, d.k1rho2, re * re + 0.0 * 0.0);
                } else {
                    double r = std::sqrt(re * re + 0.0 * 0.0);
                    v = d.rho * (r * r) / (1 + std::sqrt(1 - d.k1rho2 * (r * r)));
                    double y = 0.0;
                    for (int j = s.ncoeff - 1; j >= 0; j--) {
                        y = y * r + s.coeff[j];
                        y = y * r + 0.0;
                    }
                    y = y * r + 0.0;
                    v += y;
                }
                d.edge_val = s.pos[2] + v;
            }
            break;
        }
        case OT_SURF_TILTED: {
            const double* nv = s.normal;
            if (!(nv[2] > 0.0) || !std::isfinite(nv[0]) || !std::isfinite(nv[1]))
                return fail(OT_ERR_INVALID, "tilted surface: normal[2] must be above 0");
            d.r_eps2 = std::pow(s.r + NE, 2.0);
            d.nx = nv[0];
            d.ny = nv[1];
            d.nz = nv[2];
            d.mx = -nv[0] / nv[2];  // tilted_surface.py:69-70
            d.my = -nv[1] / nv[2];
            if (!d.flat) d.edge_val = s.pos[2] + ((s.r - NE) * d.mx + 0.0 * d.my);
            break;
        }
        case OT_SURF_DATA1D:
        case OT_SURF_DATA2D: {
            const int n = s.nknots, nc = n - OT_SPL_K - 1;
            if (!s.tab || nc < OT_SPL_K + 1) return fail(OT_ERR_INVALID, "data surface: spline tables missing or too short");
            if (mask_table_cells(s) < 0) return fail(OT_ERR_INVALID, "data surface: mask table header missing or out of range");
            if (s.tab_len != surface_table_len(s)) return fail(OT_ERR_INVALID, "data surface: tab_len does not match nknots");
            if (!(s.sign == 1.0 || s.sign == -1.0)) return fail(OT_ERR_INVALID, "data surface: sign must be +1 or -1");
            d.r_eps2 = std::pow(s.r + NE, 2.0);
            d.sgn = s.sign;
            d.offs = s.offset;
            d.nk = n;
            d.deriv_unrot = (s.flags & OT_SURF_FLAG_DERIV_UNROTATED) ? 1 : 0;
            d.mask_n = (int32_t)mask_table_cells(s);
            if (d.mask_n) {
                d.mask_off = spline_table_len(s) + 1;
                d.mask_r = s.r;
                d.mask_scale = (s.kind == OT_SURF_DATA1D ? (double)d.mask_n : 0.5 * (double)d.mask_n) / s.r;
            }
            const double span = s.tab[nc] - s.tab[OT_SPL_K];  // t(nk1 + 1) - t(k1)
            if (!(span > 0.0)) return fail(OT_ERR_INVALID, "data surface: knots must increase");
            d.inv_h = (double)(nc - OT_SPL_K - 1 > 0 ? nc - OT_SPL_K - 1 : 1) / span;
            {   // equidistant part of the knots: indices K + 1 .. n - K - 2 (ot_spline.hpp::bspl_basis)
                const int ulo = OT_SPL_K + 1, uhi = n - OT_SPL_K - 2;
                d.ku_lo = s.tab[OT_SPL_K];  // t(k1), t(nk1 + 1): the range arguments are clamped to
                d.ku_hi = s.tab[nc];
                d.ku_t0 = d.ku_h = d.ku_inv_h = 0.0;  // ku_h == 0: no equidistant part (table path everywhere)
                if (uhi - ulo >= 2 * OT_SPL_K) {
                    const double h = (s.tab[uhi] - s.tab[ulo]) / (double)(uhi - ulo);
                    bool uniform = h > 0.0;
                    for (int i = ulo; i <= uhi && uniform; i++)
                        uniform = std::fabs(s.tab[i] - (s.tab[ulo] + (i - ulo) * h)) <= 1e-9 * h;
                    if (uniform) {
                        d.ku_t0 = s.tab[ulo];
                        d.ku_h = h;
                        d.ku_inv_h = 1.0 / h;
                    }
                }
            }
            if (s.kind == OT_SURF_DATA2D) {
                d.rot = (s.angle != 0.0);
                d.cna = std::cos(-s.angle);
                d.sna = std::sin(-s.angle);
                d.cpa = std::cos(s.angle);
                d.spa = std::sin(s.angle);
            }
            d.tab = s.tab;  // host pointer for the edge value below; the caller swaps in the device copy
            if (!d.flat) d.edge_val = s.pos[2] + data_values_rel(d, s.r - NE, 0.0);
            break;
        }
        default:
            return fail(OT_ERR_INVALID, "surface: unknown kind");
    }
    return OT_OK;
}

// A compiled surface for the leaf entry points: spline tables (if any) are uploaded for the duration of the call.
struct LeafSurface {
    SurfDev d;
    double* dev_tab = nullptr;
    hipStream_t st = nullptr;
    int init(const ot_surface* surf, hipStream_t stream) {
        st = stream;
        if (int rc = compile_surface(*surf, d)) return rc;
        if (d.tab) {
            HIP_TRY(hipMalloc((void**)&dev_tab, sizeof(double) * surf->tab_len));
            HIP_TRY(hipMemcpyAsync(dev_tab, surf->tab, sizeof(double) * surf->tab_len, hipMemcpyHostToDevice, st));
            d.tab = dev_tab;
        }
        return OT_OK;
    }
    ~LeafSurface() {
        if (dev_tab) {
            (void)hipStreamSynchronize(st);  // kernels of this call still read the tables
            (void)hipFree(dev_tab);
        }
    }
};

static size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

extern "C" int ot_scene_create(const ot_scene_desc* desc, ot_scene** out) {
    if (!desc || !out) return fail(OT_ERR_INVALID, "ot_scene_create: null argument");
    if (int rc = require_device()) return rc;
    if (desc->n_elements < 1 || desc->n_surfaces < 1 || desc->n_media < 1)
        return fail(OT_ERR_INVALID, "scene needs at least one element, surface and medium");
    if (desc->n0 < 0 || desc->n0 >= desc->n_media) return fail(OT_ERR_INVALID, "scene: n0 out of range");

    std::vector<SurfDev> surfs(desc->n_surfaces);
    for (int i = 0; i < desc->n_surfaces; i++)
        if (int rc = compile_surface(desc->surfaces[i], surfs[i])) return rc;

    // flatten elements into one step per tracing surface
    std::vector<StepDev> steps;
    int n_hurb = 0;
    for (int i = 0; i < desc->n_elements; i++) {
        const ot_element& e = desc->elements[i];
        if (e.front < 0 || e.front >= desc->n_surfaces) return fail(OT_ERR_INVALID, "element: front surface out of range");
        StepDev d;
        std::memset(&d, 0, sizeof(d));
        d.surf = e.front;
        d.n_next = -1;
        d.filter = -1;
        d.hurb_slot = -1;
        switch (e.kind) {
            case OT_EL_LENS: {
                if (e.back < 0 || e.back >= desc->n_surfaces) return fail(OT_ERR_INVALID, "lens: back surface out of range");
                if (e.n_lens < 0 || e.n_lens >= desc->n_media || e.n_after < 0 || e.n_after >= desc->n_media)
                    return fail(OT_ERR_INVALID, "lens: medium out of range");
                d.kind = OT_STEP_LENS_FRONT;
                d.n_next = e.n_lens;
                steps.push_back(d);
                d.kind = OT_STEP_LENS_BACK;
                d.surf = e.back;
                d.n_next = e.n_after;
                steps.push_back(d);
                break;
            }
            case OT_EL_IDEAL_LENS:
                if (e.n_after < 0 || e.n_after >= desc->n_media) return fail(OT_ERR_INVALID, "ideal lens: medium out of range");
                if (e.D == 0.0) return fail(OT_ERR_INVALID, "ideal lens: optical power must be non-zero");
                d.kind = OT_STEP_IDEAL;
                d.n_next = e.n_after;
                d.f = 1000 / e.D;
                d.fsign = (d.f > 0) - (d.f < 0);
                steps.push_back(d);
                break;
            case OT_EL_FILTER:
                if (e.filter < 0 || e.filter >= desc->n_filters) return fail(OT_ERR_INVALID, "filter index out of range");
                d.kind = OT_STEP_FILTER;
                d.filter = e.filter;
                steps.push_back(d);
                break;
            case OT_EL_APERTURE:
                d.kind = OT_STEP_APERTURE;
                d.hurb = (desc->use_hurb && e.hurb && i != desc->n_elements - 1) ? 1 : 0;  // raytracer.py:385
                if (d.hurb && surfs[e.front].kind != OT_SURF_RING && surfs[e.front].kind != OT_SURF_SLIT)
                    return fail(OT_ERR_UNSUPPORTED, "HURB is only defined for ring and slit apertures (raytracer.py:548-552)");
                d.hurb_slot = d.hurb ? n_hurb++ : -1;
                steps.push_back(d);
                break;
            default:
                return fail(OT_ERR_INVALID, "element: unknown kind");
        }
    }
    bool needs_hurb = false, needs_ideal_filter = false;  // anything beyond lens surfaces and plain apertures
    int hit_level = OT_HIT_CLOSED;
    for (const StepDev& d : steps) {
        needs_ideal_filter |= d.kind == OT_STEP_IDEAL || d.kind == OT_STEP_FILTER;
        needs_hurb |= d.hurb != 0;
        const SurfDev& sf = surfs[d.surf];
        int lv = OT_HIT_CLOSED;
        if (sf.kind == OT_SURF_DATA1D || sf.kind == OT_SURF_DATA2D) {
            // flat data surfaces have a closed-form hit -- unless a mask_func bitmap has to be consulted
            if (!sf.flat || sf.mask_n != 0) lv = OT_HIT_SPLINE;
        } else if (sf.kind >= OT_SURF_ASPHERE && !sf.flat) {
            lv = OT_HIT_ILLINOIS;
        }
        hit_level = std::max(hit_level, lv);
    }
    const int nt = (int)steps.size() + 1;  // sections = tracing surfaces + 2, the end aperture being a step

    std::vector<FilterDev> filts(desc->n_filters > 0 ? desc->n_filters : 1);
    for (int i = 0; i < desc->n_filters; i++) {
        const ot_filter& f = desc->filters[i];
        FilterDev& d = filts[i];
        std::memset(&d, 0, sizeof(d));
        d.type = f.type;
        d.inverse = f.inverse;
        d.tab_len = f.tab_len;
        d.tab_off = f.tab_off;
        d.val = f.val;
        d.wl0 = f.wl0;
        d.wl1 = f.wl1;
        d.mu32 = (float)f.mu;
        d.den32 = (float)(2 * std::pow(f.sig, 2.0));
        d.val32 = (float)f.val;
        if ((f.type == OT_T_DATA || f.type == OT_T_LINES) &&
            (f.tab_off < 0 || f.tab_off + 2 * (int64_t)f.tab_len > desc->table_pool_len))
            return fail(OT_ERR_INVALID, "filter table outside the pool");
    }
    bool needs_tables = false;
    for (int i = 0; i < desc->n_filters; i++)
        needs_tables |= (desc->filters[i].type == OT_T_DATA || desc->filters[i].type == OT_T_LINES);
    for (int i = 0; i < desc->n_media; i++) {
        const ot_medium& m = desc->media[i];
        needs_tables |= (m.model == OT_N_DATA || m.model == OT_N_LINES);
        if ((m.model == OT_N_DATA || m.model == OT_N_LINES) &&
            (m.tab_off < 0 || m.tab_off + 2 * (int64_t)m.tab_len > desc->table_pool_len))
            return fail(OT_ERR_INVALID, "medium table outside the pool");
    }

    // discrete spectra: tabulate n(lambda), n1/n2 and filter T per step and line (IEEE arithmetic on the host, the
    // same expressions the device evaluates; raytracer.py:305, 327-332, 799, 380)
    std::vector<double> line_tab;
    int n_lines = 0;
    if (desc->n_lines > 0 && desc->n_lines <= OT_MAX_LINES && desc->lines) {
        n_lines = desc->n_lines;
        const int rows = 3 * (int)steps.size() + 2;
        line_tab.assign((size_t)rows * OT_MAX_LINES, 0.0);
        for (int j = 0; j < n_lines; j++) {
            const float wl32 = (float)desc->lines[j];
            line_tab[j] = (double)wl32;
            double n_cur = medium_n(desc->media[desc->n0], desc->table_pool, wl32);
            line_tab[(size_t)(1 + 3 * steps.size()) * OT_MAX_LINES + j] = n_cur;  // ambient row
            for (size_t i = 0; i < steps.size(); i++) {
                const StepDev& d = steps[i];
                double n_next = n_cur, Nq = 1.0, T = 1.0;
                if (d.kind <= OT_STEP_IDEAL) {
                    n_next = medium_n(desc->media[d.n_next], desc->table_pool, wl32);
                    Nq = n_cur / n_next;
                } else if (d.kind == OT_STEP_FILTER) {
                    T = filter_T(filts[d.filter], desc->table_pool, wl32);
                }
                line_tab[(size_t)(1 + 3 * i + 0) * OT_MAX_LINES + j] = n_next;
                line_tab[(size_t)(1 + 3 * i + 1) * OT_MAX_LINES + j] = Nq;
                line_tab[(size_t)(1 + 3 * i + 2) * OT_MAX_LINES + j] = T;
                n_cur = n_next;
            }
        }
    }

    // one device blob: header | surfaces | elements | media | filters | pool
    size_t o_hdr = 0;
    size_t o_surf = align_up(o_hdr + sizeof(SceneDev));
    size_t o_elem = align_up(o_surf + sizeof(SurfDev) * surfs.size());
    size_t o_med = align_up(o_elem + sizeof(StepDev) * steps.size());
    size_t o_flt = align_up(o_med + sizeof(ot_medium) * desc->n_media);
    size_t o_pool = align_up(o_flt + sizeof(FilterDev) * filts.size());
    size_t pool_n = desc->table_pool_len > 0 ? (size_t)desc->table_pool_len : 1;
    size_t o_lines = align_up(o_pool + sizeof(double) * pool_n);
    size_t o_cnt = align_up(o_lines + sizeof(double) * (line_tab.size() + 1));
    size_t o_stab = align_up(o_cnt + sizeof(unsigned int) * (size_t)OT_CNT_SLOTS * (OT_N_INFOS * nt + 1));
    std::vector<size_t> stab_off(surfs.size(), 0);  // spline tables of data surfaces
    size_t total = o_stab;
    for (size_t i = 0; i < surfs.size(); i++) {
        if (!surfs[i].tab) continue;
        stab_off[i] = total;
        total = align_up(total + sizeof(double) * (size_t)desc->surfaces[i].tab_len);
    }
    total = align_up(total + 1);

    std::vector<char> host(total, 0);
    char* blob = nullptr;
    HIP_TRY(hipMalloc((void**)&blob, total));

    SceneDev h;
    std::memset(&h, 0, sizeof(h));
    std::memcpy(h.outline, desc->outline, sizeof(h.outline));
    h.n_surfaces = desc->n_surfaces;
    h.n_steps = (int32_t)steps.size();
    h.n_media = desc->n_media;
    h.n_filters = desc->n_filters;
    h.n0 = desc->n0;
    h.no_pol = desc->no_pol;
    h.use_hurb = desc->use_hurb;
    h.nt = nt;
    h.n_hurb = n_hurb;
    h.hurb_factor = desc->hurb_factor;
    h.surfaces = (const SurfDev*)(blob + o_surf);
    h.steps = (const StepDev*)(blob + o_elem);
    h.media = (const ot_medium*)(blob + o_med);
    h.filters = (const FilterDev*)(blob + o_flt);
    h.pool = (const double*)(blob + o_pool);
    h.pool_len = desc->table_pool_len;
    h.n_lines = n_lines;
    h.line_tab = (const double*)(blob + o_lines);

    for (size_t i = 0; i < surfs.size(); i++) {
        if (!surfs[i].tab) continue;
        std::memcpy(host.data() + stab_off[i], desc->surfaces[i].tab, sizeof(double) * (size_t)desc->surfaces[i].tab_len);
        surfs[i].tab = (const double*)(blob + stab_off[i]);
    }
    std::memcpy(host.data() + o_hdr, &h, sizeof(h));
    std::memcpy(host.data() + o_surf, surfs.data(), sizeof(SurfDev) * surfs.size());
    std::memcpy(host.data() + o_elem, steps.data(), sizeof(StepDev) * steps.size());
    std::memcpy(host.data() + o_med, desc->media, sizeof(ot_medium) * desc->n_media);
    std::memcpy(host.data() + o_flt, filts.data(), sizeof(FilterDev) * filts.size());
    if (desc->table_pool_len > 0)
        std::memcpy(host.data() + o_pool, desc->table_pool, sizeof(double) * desc->table_pool_len);
    if (!line_tab.empty()) std::memcpy(host.data() + o_lines, line_tab.data(), sizeof(double) * line_tab.size());
    hipError_t e = hipMemcpy(blob, host.data(), total, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(blob);
        return fail(OT_ERR_HIP, std::string("scene upload: ") + hipGetErrorString(e));
    }

    ot_scene* sc = new ot_scene;
    sc->h = h;
    sc->needs_tables = needs_tables;
    sc->cnt_slots = (unsigned int*)(blob + o_cnt);
    // the "full" bit of the kernel variants: HURB; at hit level 0 ideal lenses and filters as well (trace_ray)
    sc->needs_full = needs_hurb || (needs_ideal_filter && hit_level == OT_HIT_CLOSED);
    sc->hit_level = hit_level;
    sc->d = (SceneDev*)(blob + o_hdr);
    sc->blob = blob;
    (void)hipGetDevice(&sc->device);
    *out = sc;
    return OT_OK;
}

extern "C" void ot_scene_destroy(ot_scene* sc) {
    if (!sc) return;
    (void)hipFree(sc->blob);
    if (sc->pin_msgs) (void)hipHostFree(sc->pin_msgs);
    if (sc->ev0) (void)hipEventDestroy(sc->ev0);
    if (sc->ev1) (void)hipEventDestroy(sc->ev1);
    delete sc;
}

// Kernel timing for measurements (bench.py): with timing on, every trace launch is bracketed by two events on its
// own stream, recorded right before and right after the tracing kernel (the counter reduction stays outside).
extern "C" int ot_scene_set_timing(ot_scene* sc, int32_t on) {
    if (!sc) return fail(OT_ERR_INVALID, "ot_scene_set_timing: null scene");
    if (on && !sc->ev0) {
        HIP_TRY(hipEventCreate(&sc->ev0));
        HIP_TRY(hipEventCreate(&sc->ev1));
    }
    sc->timing = on != 0;
    sc->ev_valid = false;
    return OT_OK;
}

extern "C" int ot_scene_last_trace_ms(const ot_scene* sc, double* ms) {
    if (!sc || !ms) return fail(OT_ERR_INVALID, "ot_scene_last_trace_ms: null argument");
    if (!sc->ev_valid) return fail(OT_ERR_INVALID, "no timed trace launch on this scene (ot_scene_set_timing)");
    HIP_TRY(hipEventSynchronize(sc->ev1));
    float t = 0.f;
    HIP_TRY(hipEventElapsedTime(&t, sc->ev0, sc->ev1));
    *ms = (double)t;
    return OT_OK;
}

extern "C" int ot_scene_sections(const ot_scene* sc) { return sc ? sc->h.nt : OT_ERR_INVALID; }

// ---------------------------------------------------------------------------------------------------------
// sources (host)
// ---------------------------------------------------------------------------------------------------------
static double gauss_peak1(double x, double mu, double sig) {  // color/srgb.py:447-457
    return 1 / (sig * std::sqrt(2 * M_PI)) * std::exp(-0.5 / (sig * sig) * (x - mu) * (x - mu));
}

static double srgb_primary(int c, double wl) {  // color/srgb.py:469-509
    if (wl < 380. || wl > 780.) return 0.0;
    switch (c) {
        case 0: return 75.1660756583 * 0.951190393 * (gauss_peak1(wl, 639.854491, 30.0) + 0.0500907584 * gauss_peak1(wl, 418.905848, 80.6220465));
        case 1: return 83.4999222966 * 1 * gauss_peak1(wl, 539.13108974, 33.31164968);
        default: return 47.99521746361 * 1.16364585503 * (gauss_peak1(wl, 454.833119, 20.1460206) + 0.184484176 * gauss_peak1(wl, 459.658190, 71.0927568));
    }
}

static double srgb_to_linear(double v) {  // color/srgb.py:30-47
    double a = 0.055, av = std::fabs(v);
    if (av <= 0.04045) return 1 / 12.92 * v;
    double sg = (v > 0) - (v < 0);
    return sg * std::pow(1 / (1 + a) * (av + a), 2.4);
}

extern "C" int ot_sources_create(const ot_source* sources, int32_t n_sources, ot_sources** out) {
    if (!sources || !out || n_sources < 1) return fail(OT_ERR_INVALID, "ot_sources_create: bad argument");
    if (int rc = require_device()) return rc;

    std::vector<SourceDev> devs(n_sources);
    std::vector<double> tabs;                 // all tables, offsets resolved after upload
    std::vector<std::vector<size_t>> offs(n_sources, std::vector<size_t>(9, (size_t)-1));
    auto push_pairs = [&](const double* tab, size_t n) {  // x[n] | F[n]  ->  (F_j, x_j) pairs
        size_t o = tabs.size();
        for (size_t j = 0; j < n; j++) {
            tabs.push_back(tab[n + j]);
            tabs.push_back(tab[j]);
        }
        return o;
    };
    auto push = [&](const double* p, size_t n) {
        size_t o = tabs.size();
        tabs.insert(tabs.end(), p, p + n);
        return o;
    };
    // inverse-CDF start hints (CdfGuide): int32 tables behind the double tables in the same blob
    std::vector<int32_t> guides;
    struct GuideRef { size_t off; CdfGuide* dst; };
    std::vector<GuideRef> grefs;
    auto add_guide = [&](const double* F, size_t n, double x0, CdfGuide* dst) {
        const double x1 = F[n - 1];
        size_t K = 16;
        while (K < 4 * n && K < ((size_t)1 << 22)) K <<= 1;  // ~4 buckets per table node
        dst->K = (int32_t)K;
        dst->x0 = x0;
        dst->scale = (x1 > x0) ? (double)K / (x1 - x0) : 0.0;
        grefs.push_back({guides.size(), dst});
        size_t j = 0;
        for (size_t b = 0; b < K; b++) {
            const double xb = x0 + (double)b / (dst->scale > 0 ? dst->scale : 1.0);
            while (j + 1 < n && F[j + 1] <= xb) j++;
            guides.push_back((int32_t)j);
        }
    };
    struct PickRef { size_t off; int src; };
    std::vector<PickRef> pick_refs;
    bool any_rgb = false;
    for (int i = 0; i < n_sources; i++) {
        const ot_source& s = sources[i];
        SourceDev& d = devs[i];
        std::memset(&d, 0, sizeof(d));
        d.shape = s.shape; d.divergence = s.divergence; d.div_2d = s.div_2d; d.orientation = s.orientation;
        d.polarization = s.polarization; d.spectrum = s.spectrum; d.img_w = s.img_w; d.img_h = s.img_h;
        std::memcpy(d.pos, s.pos, sizeof(d.pos));
        d.r = s.r; d.ri = s.ri; d.dim[0] = s.dim[0]; d.dim[1] = s.dim[1];
        d.ca = (s.angle != 0.0) ? std::cos(s.angle) : 1.0;
        d.sa = (s.angle != 0.0) ? std::sin(s.angle) : 0.0;
        d.div_rad = s.div_angle * (M_PI / 180.0);
        d.div_sin = std::sin(d.div_rad);
        d.div_axis = s.div_axis_angle * (M_PI / 180.0);
        std::memcpy(d.s, s.s, sizeof(d.s));
        std::memcpy(d.conv_pos, s.conv_pos, sizeof(d.conv_pos));
        if (s.orientation == OT_OR_CONSTANT || (s.orientation == OT_OR_CONVERGING && s.shape == OT_SRC_POINT)) {
            if (s.orientation == OT_OR_CONVERGING) {  // misc.normalize(conv_pos - p) with p = pos (ray_source.py:269)
                const double dx = s.conv_pos[0] - s.pos[0], dy = s.conv_pos[1] - s.pos[1], dz = s.conv_pos[2] - s.pos[2];
                const double l = std::sqrt(dx * dx + dy * dy + dz * dz);
                d.s[0] = dx / l; d.s[1] = dy / l; d.s[2] = dz / l;
            }
            const double fa = 1.0 / std::sqrt(1 - d.s[0] * d.s[0]);  // ray_source.py:339-341
            d.fy[0] = 0.0; d.fy[1] = -d.s[2] * fa; d.fy[2] = d.s[1] * fa;
            d.fx[0] = d.s[1] * d.fy[2] - d.s[2] * d.fy[1];
            d.fx[1] = d.s[2] * d.fy[0] - d.s[0] * d.fy[2];
            d.fx[2] = d.s[0] * d.fy[1] - d.s[1] * d.fy[0];
            d.frame_uniform = 1;
        }
        d.pol_angle = s.pol_angle;
        d.pol_cos = std::cos(s.pol_angle);
        d.pol_sin = std::sin(s.pol_angle);
        d.axis_cos = std::cos(d.div_axis);
        d.axis_sin = std::sin(d.div_axis);
        d.px_w = (s.img_w > 0) ? s.dim[0] / (double)s.img_w : 0.0;  // ray_source.py:252-253
        d.px_h = (s.img_h > 0) ? s.dim[1] / (double)s.img_h : 0.0;
        d.inv_img_w = (s.img_w > 0) ? 1.0 / (double)s.img_w : 0.0;
        d.wl = s.wl; d.wl0 = s.wl0; d.wl1 = s.wl1; d.mu = s.mu; d.sig = s.sig;
        d.power = s.power;
        if (s.spectrum == OT_SPEC_GAUSSIAN) {  // light_spectrum.py:117-118
            d.gauss_xl = (1 + std::erf((s.wl0 - s.mu) / (std::sqrt(2.0) * s.sig))) / 2;
            d.gauss_xr = (1 + std::erf((s.wl1 - s.mu) / (std::sqrt(2.0) * s.sig))) / 2;
        }
        if (s.shape < OT_SRC_POINT || s.shape > OT_SRC_IMAGE_GRAY) return fail(OT_ERR_INVALID, "source: unknown shape");
        if (s.orientation < OT_OR_CONSTANT || s.orientation > OT_OR_ARRAY) return fail(OT_ERR_INVALID, "source: unknown orientation");
        if (s.orientation == OT_OR_ARRAY && s.s_or) {
            if (s.n_or < 1) return fail(OT_ERR_INVALID, "source: orientation array without a length");
            d.s_or = s.s_or;
            d.n_or = s.n_or;
        }
        bool needs_spec = s.shape != OT_SRC_IMAGE_RGB && (s.spectrum == OT_SPEC_LINES || s.spectrum == OT_SPEC_TABLE);
        if (needs_spec) {
            if (!s.spec_tab || s.n_spec < 1) return fail(OT_ERR_INVALID, "source: spectrum table missing");
            offs[i][0] = push(s.spec_tab, 2 * (size_t)s.n_spec);
            d.n_spec = s.n_spec;
            const double* F = s.spec_tab + s.n_spec;
            add_guide(F, (size_t)s.n_spec, s.spectrum == OT_SPEC_LINES ? 0.0 : F[0], &d.g_spec);
            if (s.spectrum != OT_SPEC_LINES) offs[i][6] = push_pairs(s.spec_tab, (size_t)s.n_spec);
        }
        if (s.polarization == OT_POL_LIST || s.polarization == OT_POL_TABLE) {
            if (!s.pol_tab || s.n_pol < 1) return fail(OT_ERR_INVALID, "source: polarisation table missing");
            offs[i][1] = push(s.pol_tab, 2 * (size_t)s.n_pol);
            d.n_pol = s.n_pol;
            const double* F = s.pol_tab + s.n_pol;
            add_guide(F, (size_t)s.n_pol, s.polarization == OT_POL_LIST ? 0.0 : F[0], &d.g_pol);
            if (s.polarization != OT_POL_LIST) offs[i][7] = push_pairs(s.pol_tab, (size_t)s.n_pol);
        }
        if (s.divergence == OT_DIV_TABLE) {
            if (!s.div_tab || s.n_div < 2) return fail(OT_ERR_INVALID, "source: divergence table missing");
            offs[i][2] = push(s.div_tab, 2 * (size_t)s.n_div);
            d.n_div = s.n_div;
            add_guide(s.div_tab + s.n_div, (size_t)s.n_div, s.div_tab[s.n_div], &d.g_div);
            offs[i][8] = push_pairs(s.div_tab, (size_t)s.n_div);
        }
        if (s.shape == OT_SRC_IMAGE_RGB || s.shape == OT_SRC_IMAGE_GRAY) {
            size_t npx = (size_t)s.img_w * (size_t)s.img_h;
            if (!s.img_pdf || npx < 1) return fail(OT_ERR_INVALID, "image source: pixel pdf missing");
            if (s.shape == OT_SRC_IMAGE_RGB && !s.img_rgb) return fail(OT_ERR_INVALID, "RGB image source: pixel colours missing");
            std::vector<double> rec(4 * npx, 0.0);
            double acc = 0.0;  // np.cumsum of the pixel pdf (random.py:133 on f_ = f[f > 0]; zero-weight pixels
            const double fr = 0.885651229244, fb = 0.775993481741;  // srgb.py:24-26
            for (size_t j = 0; j < npx; j++) {  // keep the running sum and can never be selected by "next")
                acc += s.img_pdf[j];
                rec[4 * j] = acc;
                if (s.shape == OT_SRC_IMAGE_RGB) {  // color.random_wavelengths_from_srgb srgb.py:522-541
                    double r = srgb_to_linear(s.img_rgb[3 * j]) * fr;
                    double g = srgb_to_linear(s.img_rgb[3 * j + 1]);
                    double b = srgb_to_linear(s.img_rgb[3 * j + 2]) * fb;
                    double c0 = r, c1 = r + g, c2 = r + g + b;
                    double den = (c2 != 0.0) ? c2 : 1.0;
                    rec[4 * j + 1] = c0 / den;
                    rec[4 * j + 2] = c1 / den;
                }
            }
            if (tabs.size() & 1) tabs.push_back(0.0);  // PixRec is read with 16-byte loads
            offs[i][3] = push(rec.data(), rec.size());
            // bucket table of the pixel pick: K ~ 4 buckets per pixel; pick_lo[b] = pixels whose own bucket lies before b.
            // The bucket of a value is the DEVICE's expression (pixel_bucket, monotone in X), so for X in bucket b every
            // pixel before pick_lo[b] has F < X and every pixel from pick_lo[b + 1] on has F > X.
            size_t K = 16;
            while (K < 4 * npx && K < ((size_t)1 << 22)) K <<= 1;
            d.pick_K = (int32_t)K;
            d.pix_total = acc;
            d.pick_scale = (acc > 0.0) ? (double)K / acc : 0.0;
            pick_refs.push_back({guides.size(), i});
            size_t j = 0;
            for (size_t b = 0; b <= K; b++) {
                while (j < npx && (size_t)pixel_bucket(rec[4 * j], d.pick_scale, (int)K) < b) j++;
                guides.push_back((int32_t)j);
            }
            any_rgb = any_rgb || s.shape == OT_SRC_IMAGE_RGB;
        }
    }
    size_t prim_off = (size_t)-1;
    if (any_rgb) {
        // The three primaries over wavelengths(5000) (srgb.py:528, 549-551): cumulative trapezoid F_j, and the inverse
        // x(F) the reference interpolates linearly between its nodes (random.py:150-157) sampled at OT_PRIM_M + 1
        // equidistant values of the uniform variable.  Between two samples the device interpolates linearly as well:
        // exact where no node lies between them, elsewhere off by less than the spacing of the reference's own
        // wavelength grid (0.08 nm) except in the few buckets of the far tails (1.5e-5 of the rays each).
        std::vector<double> inv(3 * (size_t)(OT_PRIM_M + 1));
        std::vector<double> x(OT_PRIM_N), F(OT_PRIM_N);
        for (int c = 0; c < 3; c++) {
            double prev = 0.0;
            for (int j = 0; j < OT_PRIM_N; j++) {
                x[j] = 380.0 + (780.0 - 380.0) * (double)j / (double)(OT_PRIM_N - 1);
                double f = srgb_primary(c, x[j]);
                F[j] = (j == 0) ? 0.0 : F[j - 1] + (f + prev) / 2;
                prev = f;
            }
            double* o = inv.data() + (size_t)c * (OT_PRIM_M + 1);
            int lo = 0;
            for (int m = 0; m <= OT_PRIM_M; m++) {
                const double X = F[0] + ((double)m / (double)OT_PRIM_M) * (F[OT_PRIM_N - 1] - F[0]);
                while (lo < OT_PRIM_N - 2 && F[lo + 1] <= X) lo++;
                const double dF = F[lo + 1] - F[lo];
                o[m] = (dF > 0) ? x[lo] + (X - F[lo]) / dF * (x[lo + 1] - x[lo]) : x[lo];
            }
        }
        prim_off = push(inv.data(), inv.size());
    }

    size_t o_tab = align_up(sizeof(SourceDev) * n_sources);
    size_t o_guide = align_up(o_tab + sizeof(double) * (tabs.size() + 1));
    size_t total = align_up(o_guide + sizeof(int32_t) * (guides.size() + 1));
    char* blob = nullptr;
    HIP_TRY(hipMalloc((void**)&blob, total));
    const double* dtab = (const double*)(blob + o_tab);
    for (const GuideRef& r : grefs) r.dst->g = (const int32_t*)(blob + o_guide) + r.off;
    for (const PickRef& r : pick_refs) devs[r.src].pick_lo = (const int32_t*)(blob + o_guide) + r.off;
    for (int i = 0; i < n_sources; i++) {
        SourceDev& d = devs[i];
        if (offs[i][0] != (size_t)-1) d.spec_tab = dtab + offs[i][0];
        if (offs[i][1] != (size_t)-1) d.pol_tab = dtab + offs[i][1];
        if (offs[i][2] != (size_t)-1) d.div_tab = dtab + offs[i][2];
        if (offs[i][3] != (size_t)-1) d.pix_rec = dtab + offs[i][3];
        if (prim_off != (size_t)-1) d.prim_inv = dtab + prim_off;
        if (offs[i][6] != (size_t)-1) d.spec_pairs = dtab + offs[i][6];
        if (offs[i][7] != (size_t)-1) d.pol_pairs = dtab + offs[i][7];
        if (offs[i][8] != (size_t)-1) d.div_pairs = dtab + offs[i][8];
    }
    std::vector<char> host(total, 0);
    std::memcpy(host.data(), devs.data(), sizeof(SourceDev) * n_sources);
    if (!tabs.empty()) std::memcpy(host.data() + o_tab, tabs.data(), sizeof(double) * tabs.size());
    if (!guides.empty()) std::memcpy(host.data() + o_guide, guides.data(), sizeof(int32_t) * guides.size());
    hipError_t e = hipMemcpy(blob, host.data(), total, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(blob);
        return fail(OT_ERR_HIP, std::string("source upload: ") + hipGetErrorString(e));
    }
    ot_sources* so = new ot_sources;
    so->d = (SourceDev*)blob;
    so->n = n_sources;
    so->blob = blob;
    so->has_image = false;
    for (int i = 0; i < n_sources; i++) so->has_image = so->has_image || sources[i].shape >= OT_SRC_IMAGE_RGB;
    so->n_or = new int64_t[n_sources];
    so->power = new double[n_sources];
    for (int i = 0; i < n_sources; i++) {
        so->n_or[i] = devs[i].s_or ? devs[i].n_or : -1;
        so->power[i] = devs[i].power;
    }
    (void)hipGetDevice(&so->device);
    *out = so;
    return OT_OK;
}

static void drop_range_cache(ot_sources* s);

extern "C" void ot_sources_destroy(ot_sources* s) {
    if (!s) return;
    drop_range_cache(s);
    (void)hipFree(s->blob);
    delete[] s->n_or;
    delete[] s->power;
    delete s;
}

// ---------------------------------------------------------------------------------------------------------
// kernels

// sums the slot tables into the caller's int64 counters (ADD) and clears them for the next launch:
// one workgroup per counter, one lane per 4 slots, wave shuffle + LDS reduction
template <bool ACCUM>  // ACCUM: add to the caller's counters; otherwise overwrite them (pinned host buffer: no read over PCIe)
__global__ __launch_bounds__(256) void reduce_counters_kernel(unsigned int* __restrict__ slots, int n_cnt,
                                                              unsigned long long* __restrict__ msgs) {
    __shared__ unsigned long long part[4];
    const int k = blockIdx.x;
    unsigned long long sum = 0;
    for (int sidx = threadIdx.x; sidx < OT_CNT_SLOTS; sidx += blockDim.x) {
        unsigned int v = slots[(size_t)sidx * n_cnt + k];
        if (v) {
            sum += v;
            slots[(size_t)sidx * n_cnt + k] = 0u;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = part[0] + part[1] + part[2] + part[3];
        if (!ACCUM) {
            msgs[k] = (k == n_cnt - 1) ? (t ? 1ull : 0ull) : t;
        } else if (t) {
            if (k == n_cnt - 1) msgs[k] = 1ull; else msgs[k] += t;
        }
    }
}

// RaySource.create_rays only: writes section 0 (ot_rays_generate)
template <bool POL>
__global__ __launch_bounds__(256) void generate_kernel(ot_rays R, const SourceDev* __restrict__ sources, RangeArgs rg,
                                                       uint64_t seed) {
    const int64_t ray = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (ray >= R.N) return;
    NewRay nr;
    if (!generate_lane(rg, sources, ray, seed, !POL, nr)) return;
    const int64_t N = R.N, nt = R.nt;
    R.p[ray] = nr.p.x;
    R.p[ray + N * nt] = nr.p.y;
    R.p[ray + N * 2 * nt] = nr.p.z;
    R.s[ray] = nr.s.x;
    R.s[ray + N] = nr.s.y;
    R.s[ray + 2 * N] = nr.s.z;
    R.w[ray] = nr.w;
    R.wl[ray] = nr.wl;
    if (POL) {
        R.pol[ray] = (float)nr.polx;
        R.pol[ray + N * nt] = (float)nr.poly;
        R.pol[ray + N * 2 * nt] = (float)nr.polz;
    }
}

// ---- leaf kernels (one lane per element) -------------------------------------------------------------------
__global__ __launch_bounds__(256) void find_hit_kernel(SurfDev sf, int64_t n, const double* __restrict__ p,
                                                       const double* __restrict__ s, double* __restrict__ ph_out,
                                                       uint8_t* __restrict__ hit_out, uint8_t* __restrict__ ill_out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    V3 pp = {p[i], p[i + n], p[i + 2 * n]}, ss = {s[i], s[i + n], s[i + 2 * n]}, ph;
    bool hit, ill;
    bool ok = find_hit(sf, pp, ss, ph, hit, ill);
    ph_out[i] = ph.x;
    ph_out[i + n] = ph.y;
    ph_out[i + 2 * n] = ph.z;
    hit_out[i] = hit;
    ill_out[i] = (uint8_t)((ill ? 1 : 0) | (ok ? 0 : 2));
}

__global__ __launch_bounds__(256) void normals_kernel(SurfDev sf, int64_t n, const double* __restrict__ x,
                                                      const double* __restrict__ y, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    V3 nn = surf_normal(sf, x[i], y[i]);
    out[i] = nn.x;
    out[i + n] = nn.y;
    out[i + 2 * n] = nn.z;
}

__global__ __launch_bounds__(256) void mask_kernel(SurfDev sf, int64_t n, const double* __restrict__ x,
                                                   const double* __restrict__ y, uint8_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = surf_mask(sf, x[i], y[i]);
}

__global__ __launch_bounds__(256) void values_kernel(SurfDev sf, int64_t n, const double* __restrict__ x,
                                                     const double* __restrict__ y, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = surf_values(sf, x[i], y[i]);
}

__global__ __launch_bounds__(256) void hurb_props_kernel(SurfDev sf, int64_t n, const double* __restrict__ x,
                                                         const double* __restrict__ y, double* __restrict__ a_,
                                                         double* __restrict__ b_, double* __restrict__ b,
                                                         uint8_t* __restrict__ inside) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double a, bb;
    V3 bv;
    bool in;
    hurb_props(sf, x[i], y[i], a, bb, bv, in);
    a_[i] = a;
    b_[i] = bb;
    b[i] = bv.x;
    b[i + n] = bv.y;
    b[i + 2 * n] = bv.z;
    inside[i] = in;
}

__global__ __launch_bounds__(256) void refraction_index_kernel(ot_medium md, const double* __restrict__ pool, int64_t n,
                                                               const float* __restrict__ wl, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = medium_n(md, pool, wl[i]);
}

// ---------------------------------------------------------------------------------------------------------
// launch helpers
// ---------------------------------------------------------------------------------------------------------
static inline dim3 grid_for(int64_t n) { return dim3((unsigned)((n + 255) / 256)); }

// What make_ranges derived from the last range list of a source table.  Chunked rendering and repeated traces
// pass the same list again and again: the argument block is reused, and for long lists so is the device copy of
// the records -- no allocation, no upload and no stream synchronisation on the launch path.
struct RangeCache {
    std::vector<ot_source_range> key;
    int64_t N = -1;
    RangeArgs rg;
    RangeRec* ext = nullptr;  // device records (n > OT_MAX_RANGES), owned by the cache
};

static void drop_range_cache(ot_sources* s) {
    if (!s->rcache) return;
    if (s->rcache->ext) (void)hipFree(s->rcache->ext);  // hipFree waits for work that may still read the records
    delete s->rcache;
    s->rcache = nullptr;
}

// Fills the kernel argument block; with more than OT_MAX_RANGES ranges the records go to device memory (kept in
// the source table's cache until a different list arrives).
static int make_ranges(const ot_source_range* ranges, int32_t n_ranges, const ot_sources* src_c, int64_t N,
                       const RangeArgs** out) {
    ot_sources* src = const_cast<ot_sources*>(src_c);
    if (!ranges || n_ranges < 1) return fail(OT_ERR_INVALID, "at least one source range is needed");
    if (RangeCache* c = src->rcache) {
        if (c->N == N && (int32_t)c->key.size() == n_ranges &&
            std::memcmp(c->key.data(), ranges, sizeof(ot_source_range) * (size_t)n_ranges) == 0) {
            *out = &c->rg;
            return OT_OK;
        }
    }
    RangeArgs rg;
    rg.ext = nullptr;
    rg.n = n_ranges;
    const bool big = n_ranges > OT_MAX_RANGES;
    std::vector<RangeRec> recs(big ? n_ranges : 0);
    // the kernel finds a wave's range by bisection: records sorted by their first ray, gap-free (empty ranges first
    // among equal starts)
    std::vector<int> order(n_ranges);
    for (int k = 0; k < n_ranges; k++) order[k] = k;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
        return ranges[a].first != ranges[b].first ? ranges[a].first < ranges[b].first : ranges[a].count < ranges[b].count;
    });
    int64_t covered = 0;
    for (int q = 0; q < n_ranges; q++) {
        const int k = order[q];
        if (ranges[k].source < 0 || ranges[k].source >= src->n) return fail(OT_ERR_INVALID, "range: source out of range");
        if (ranges[k].first < 0 || ranges[k].count < 0 || ranges[k].first + ranges[k].count > N)
            return fail(OT_ERR_INVALID, "range outside the ray storage");
        if (ranges[k].count > 0xffffffffll) return fail(OT_ERR_UNSUPPORTED, "more than 2^32 rays in one source range");
        if (src->n_or[ranges[k].source] >= 0 && src->n_or[ranges[k].source] != ranges[k].count)
            return fail(OT_ERR_INVALID, "range: ray count differs from the length of the source's orientation array");
        if (ranges[k].first != covered) return fail(OT_ERR_INVALID, "source ranges must cover all N rays exactly once");
        const uint64_t cnt = (uint64_t)ranges[k].count;
        uint32_t n2 = (uint32_t)std::sqrt((double)cnt);
        while ((uint64_t)n2 * n2 > cnt) n2--;
        while ((uint64_t)(n2 + 1) * (n2 + 1) <= cnt) n2++;
        const double inv_n = cnt ? 1.0 / (double)cnt : 0.0, inv_n2 = n2 ? 1.0 / (double)n2 : 0.0;
        const float w = (float)(ranges[k].ray_power > 0 ? ranges[k].ray_power
                                                        : (cnt ? src->power[ranges[k].source] / (double)cnt : 0.0));
        if (big) {
            recs[q] = {ranges[k].first, ranges[k].count, ranges[k].source, n2, inv_n, inv_n2, w};
        } else {
            rg.w[q] = w;
            rg.source[q] = ranges[k].source;
            rg.first[q] = ranges[k].first;
            rg.count[q] = ranges[k].count;
            rg.n2[q] = n2;
            rg.inv_n[q] = inv_n;
            rg.inv_n2[q] = inv_n2;
        }
        covered += ranges[k].count;
    }
    // (covered <= N by the checks above; rays behind the last range -- the padding of a storage whose plane stride N is
    // larger than its ray count -- are not generated and not traced)
    RangeRec* d = nullptr;
    if (big) {
        HIP_TRY(hipMalloc((void**)&d, sizeof(RangeRec) * recs.size()));
        hipError_t e = hipMemcpy(d, recs.data(), sizeof(RangeRec) * recs.size(), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            (void)hipFree(d);
            return fail(OT_ERR_HIP, std::string("range upload: ") + hipGetErrorString(e));
        }
        rg.ext = d;
    }
    drop_range_cache(src);  // the previous list (its device records are no longer needed by any new launch)
    RangeCache* c = new RangeCache;
    c->key.assign(ranges, ranges + n_ranges);
    c->N = N;
    c->rg = rg;
    c->ext = d;
    src->rcache = c;
    *out = &c->rg;
    return OT_OK;
}

static int check_rays(const ot_rays* r, bool need_pol) {
    if (!r || r->N < 0 || r->nt < 1) return fail(OT_ERR_INVALID, "bad ray storage");
    if (!r->p || !r->s || !r->w || !r->n || !r->wl) return fail(OT_ERR_INVALID, "ray storage: null buffer");
    if (need_pol && !r->pol) return fail(OT_ERR_INVALID, "ray storage: pol buffer missing although polarisation is on");
    return OT_OK;
}

extern "C" int ot_rays_generate(const ot_sources* src, const ot_source_range* ranges, int32_t n_ranges, uint64_t seed,
                                int32_t no_pol, const ot_rays* rays, void* stream) {
    if (!src) return fail(OT_ERR_INVALID, "ot_rays_generate: null sources");
    if (int rc = check_rays(rays, !no_pol)) return rc;
    const RangeArgs* rgp = nullptr;
    if (int rc = make_ranges(ranges, n_ranges, src, rays->N, &rgp)) return rc;
    const RangeArgs& rg = *rgp;
    hipStream_t st = (hipStream_t)stream;
    if (rays->N > 0) {
        if (no_pol)
            hipLaunchKernelGGL(generate_kernel<false>, grid_for(rays->N), dim3(256), 0, st, *rays, src->d, rg, seed);
        else
            hipLaunchKernelGGL(generate_kernel<true>, grid_for(rays->N), dim3(256), 0, st, *rays, src->d, rg, seed);
    }
    HIP_TRY(hipGetLastError());
    return OT_OK;
}

// msgs: device counters the launch ADDS to, or nullptr: the counters of this launch alone go to the scene's pinned
// host buffer (created on first use)
// tail: render-only launch (trace_tail_kernel) of n_tail rays -- `rays` is not used then
static int launch_trace(const ot_scene* sc_c, const ot_sources* src, const RangeArgs* rg, const ot_rays* rays,
                        const double* hurb_normals, uint64_t seed, int64_t* msgs, void* stream, const TailOut* tail = nullptr,
                        int64_t n_tail = 0) {
    ot_scene* sc = const_cast<ot_scene*>(sc_c);
    if (!sc) return fail(OT_ERR_INVALID, "ot_trace: null argument");
    bool pol = !sc->h.no_pol;
    ot_rays tail_rays = {};
    if (tail) {
        tail_rays.N = n_tail;
        rays = &tail_rays;
    } else {
        if (int rc = check_rays(rays, pol)) return rc;
        if (rays->nt != sc->h.nt) return fail(OT_ERR_INVALID, "ray storage has " + std::to_string(rays->nt) +
                                                                 " sections, the scene needs " + std::to_string(sc->h.nt));
    }
    const int n_cnt = OT_N_INFOS * sc->h.nt + 1;
    if (!msgs && !sc->pin_msgs) {
        HIP_TRY(hipHostMalloc((void**)&sc->pin_msgs, sizeof(unsigned long long) * (size_t)n_cnt,
                              hipHostMallocMapped | hipHostMallocCoherent));
        std::memset(sc->pin_msgs, 0, sizeof(unsigned long long) * (size_t)n_cnt);
    }
    if (rays->N == 0) {
        if (!msgs) std::memset(sc->pin_msgs, 0, sizeof(unsigned long long) * (size_t)n_cnt);
        return OT_OK;
    }
    hipStream_t st = (hipStream_t)stream;
    RangeArgs none;
    none.n = 0;
    none.ext = nullptr;
    const RangeArgs& r = rg ? *rg : none;
    const SourceDev* sd = src ? src->d : nullptr;
    unsigned long long* m = (unsigned long long*)msgs;
    if (!msgs) HIP_TRY(hipHostGetDevicePointer((void**)&m, sc->pin_msgs, 0));
    // kernel variant: polarisation x on-device generation x spectrum handling x feature set
    const bool tab = sc->needs_tables || hurb_normals != nullptr;
    const int feat = OT_FEAT(sc->hit_level, sc->needs_full);
    // discrete-spectrum kernels: generated rays only, and no image source (their variant of the generator has none)
    bool lines = src != nullptr && sc->h.n_lines > 0 && hurb_normals == nullptr && !src->has_image;
    // dynamic LDS: the counter table, and with discrete spectra the per-line tables (3 rows per step).  Very long
    // stacks do not fit the 64 KB a kernel gets without asking: the formula kernels (SPEC 0 / 1) trace those.
    const size_t lds_cnt = sizeof(unsigned int) * (size_t)n_cnt + 8;
    const size_t lds_lines = sizeof(double) * (size_t)(3 * sc->h.n_steps + 2) * OT_MAX_LINES;
    // spline surfaces: a 5 x 5 coefficient patch per lane (ot_spline.hpp::PatchCache)
    const size_t lds_patch = (sc->hit_level == OT_HIT_SPLINE) ? 256 * 25 * sizeof(double) + 16 : 0;
    if (lds_cnt + lds_patch > 65000)
        return fail(OT_ERR_UNSUPPORTED, lds_patch ? "ot_trace: more than ~650 tracing surfaces in a scene with spline surfaces"
                                                  : "ot_trace: more than ~3000 tracing surfaces in one scene");
    if (lines && lds_cnt + lds_lines + lds_patch > 65000) lines = false;
    const size_t lds = ((lds_cnt + (lines ? lds_lines : 0) + 7) / 8) * 8 + lds_patch;
    unsigned int* slots = sc->cnt_slots;
    if (sc->timing) HIP_TRY(hipEventRecord(sc->ev0, st));
    // lanes address their ray with 32-bit byte offsets: launches of at most 2^28 rays, base pointers advanced
    const int64_t chunk = 1ll << 28;
    for (int64_t base = 0; base < rays->N; base += chunk) {
        TraceLaunch L;
        L.count = (uint32_t)std::min<int64_t>(chunk, rays->N - base);
        L.grid = grid_for(L.count);
        L.lds = lds;
        L.st = st;
        L.sc = sc->d;
        L.part = *rays;
        if (!tail) {
            L.part.p += base; L.part.s += base; L.part.w += base; L.part.n += base; L.part.wl += base;
            if (L.part.pol) L.part.pol += base;
        }
        L.sd = sd;
        L.rg = &r;
        L.hurb_normals = hurb_normals;
        L.seed = seed;
        L.slots = slots;
        L.base = base;
        L.pol = pol;
        L.gen = src != nullptr;
        L.spec = (src && lines) ? 2 : (tab ? 1 : 0);
        if (tail) {  // the render-only form of the same feature level
            switch (feat) {
                case OT_FEAT(OT_HIT_CLOSED, 0): launch_trace_tail_feat<OT_FEAT(OT_HIT_CLOSED, 0)>(L, *tail); break;
                case OT_FEAT(OT_HIT_CLOSED, 1): launch_trace_tail_feat<OT_FEAT(OT_HIT_CLOSED, 1)>(L, *tail); break;
                case OT_FEAT(OT_HIT_ILLINOIS, 0): launch_trace_tail_feat<OT_FEAT(OT_HIT_ILLINOIS, 0)>(L, *tail); break;
                case OT_FEAT(OT_HIT_ILLINOIS, 1): launch_trace_tail_feat<OT_FEAT(OT_HIT_ILLINOIS, 1)>(L, *tail); break;
                case OT_FEAT(OT_HIT_SPLINE, 0): launch_trace_tail_feat<OT_FEAT(OT_HIT_SPLINE, 0)>(L, *tail); break;
                default: launch_trace_tail_feat<OT_FEAT(OT_HIT_SPLINE, 1)>(L, *tail); break;
            }
            continue;
        }
        switch (feat) {
            case OT_FEAT(OT_HIT_CLOSED, 0): launch_trace_feat<OT_FEAT(OT_HIT_CLOSED, 0)>(L); break;
            case OT_FEAT(OT_HIT_CLOSED, 1): launch_trace_feat<OT_FEAT(OT_HIT_CLOSED, 1)>(L); break;
            case OT_FEAT(OT_HIT_ILLINOIS, 0): launch_trace_feat<OT_FEAT(OT_HIT_ILLINOIS, 0)>(L); break;
            case OT_FEAT(OT_HIT_ILLINOIS, 1): launch_trace_feat<OT_FEAT(OT_HIT_ILLINOIS, 1)>(L); break;
            case OT_FEAT(OT_HIT_SPLINE, 0): launch_trace_feat<OT_FEAT(OT_HIT_SPLINE, 0)>(L); break;
            default: launch_trace_feat<OT_FEAT(OT_HIT_SPLINE, 1)>(L); break;
        }
    }
    HIP_TRY(hipGetLastError());
    if (sc->timing) {
        HIP_TRY(hipEventRecord(sc->ev1, st));
        sc->ev_valid = true;
    }
    if (msgs)
        hipLaunchKernelGGL(reduce_counters_kernel<true>, dim3(n_cnt), dim3(256), 0, st, slots, n_cnt, m);
    else
        hipLaunchKernelGGL(reduce_counters_kernel<false>, dim3(n_cnt), dim3(256), 0, st, slots, n_cnt, m);
    HIP_TRY(hipGetLastError());
    return OT_OK;
}

extern "C" int ot_trace(const ot_scene* scene, const ot_rays* rays, const double* hurb_normals, uint64_t seed,
                        int64_t* msgs, void* stream) {
    if (!msgs) return fail(OT_ERR_INVALID, "ot_trace: null argument");
    return launch_trace(scene, nullptr, nullptr, rays, hurb_normals, seed, msgs, stream);
}

extern "C" int ot_generate_and_trace(const ot_scene* scene, const ot_sources* src, const ot_source_range* ranges,
                                     int32_t n_ranges, uint64_t seed, const ot_rays* rays, int64_t* msgs, void* stream) {
    if (!src || !rays || !msgs) return fail(OT_ERR_INVALID, "ot_generate_and_trace: null argument");
    const RangeArgs* rg = nullptr;
    if (int rc = make_ranges(ranges, n_ranges, src, rays->N, &rg)) return rc;
    return launch_trace(scene, src, rg, rays, nullptr, seed, msgs, stream);
}

// The whole of Raytracer.trace in one synchronous call: launch, wait, counters of this launch in host memory.
// The counter reduction writes into a pinned host buffer of the scene, so the wait for the stream is the only
// synchronisation and nothing is copied back.
extern "C" int ot_generate_and_trace_host(const ot_scene* scene, const ot_sources* src, const ot_source_range* ranges,
                                          int32_t n_ranges, uint64_t seed, const ot_rays* rays, int64_t* msgs_host,
                                          void* stream) {
    if (!scene || !src || !rays || !msgs_host) return fail(OT_ERR_INVALID, "ot_generate_and_trace_host: null argument");
    const RangeArgs* rg = nullptr;
    if (int rc = make_ranges(ranges, n_ranges, src, rays->N, &rg)) return rc;
    if (int rc = launch_trace(scene, src, rg, rays, nullptr, seed, nullptr, stream)) return rc;
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    std::memcpy(msgs_host, scene->pin_msgs, sizeof(int64_t) * (size_t)(OT_N_INFOS * scene->h.nt + 1));
    return OT_OK;
}

// One workgroup per piece: rows in use = ceil(max fill / 64); the slots of this piece between its fill and the end of the
// last row in use get weight 0 and finite positions.  Workgroup 0 reports result2 = {slots in use, living rays}.
__global__ __launch_bounds__(256) void tail_seal_kernel(TailOut T, long long* __restrict__ result2) {
    __shared__ unsigned int s_max[256];
    __shared__ unsigned long long s_sum[256];
    unsigned int mx = 0;
    unsigned long long sum = 0;
    for (int k = threadIdx.x; k < OT_TAIL_PIECES; k += 256) {
        const unsigned int f = T.fill[k];
        mx = f > mx ? f : mx;
        sum += f;
    }
    s_max[threadIdx.x] = mx;
    s_sum[threadIdx.x] = sum;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            s_max[threadIdx.x] = s_max[threadIdx.x] > s_max[threadIdx.x + o] ? s_max[threadIdx.x] : s_max[threadIdx.x + o];
            s_sum[threadIdx.x] += s_sum[threadIdx.x + o];
        }
        __syncthreads();
    }
    const unsigned int rows = (s_max[0] + 63u) >> 6;
    const unsigned int piece = blockIdx.x;
    const int64_t N = T.cap;
    for (unsigned int q = T.fill[piece] + threadIdx.x; q < rows * 64u; q += 256) {
        const int64_t slot = (((int64_t)(q >> 6) * OT_TAIL_PIECES + piece) << 6) + (q & 63u);
        for (int c = 0; c < 6; c++) T.p[slot + c * N] = 0.0;
        T.w[slot] = 0.f;
        T.w[N + slot] = 0.f;
        T.wl[slot] = 0.f;
    }
    if (piece == 0 && threadIdx.x == 0) {
        result2[0] = (long long)rows * 64 * OT_TAIL_PIECES;
        result2[1] = (long long)s_sum[0];
    }
}

// Render-only chunk of Raytracer.iterative_render (raytracer.py:1235-1267: only the last chunk's rays are kept): n_rays
// rays are generated and traced without storing a section; the last section of every ray that is alive behind the last
// surface goes to the compact two-section storage `tail` (ot_trace_kernel.hpp::trace_tail_kernel).  Synchronous like
// ot_generate_and_trace_host.
extern "C" int64_t ot_tail_capacity(int64_t n_rays) {
    if (n_rays < 0) return 0;
    const int64_t waves = (n_rays + 63) / 64;
    return 65536 * std::max<int64_t>(1, (waves + OT_TAIL_PIECES - 1) / OT_TAIL_PIECES);
}

extern "C" int ot_scene_tail_supported(const ot_scene* scene) { return scene ? 1 : 0; }  // (every feature level has the form)

extern "C" int ot_generate_and_trace_tail(const ot_scene* scene, const ot_sources* src, const ot_source_range* ranges,
                                          int32_t n_ranges, uint64_t seed, int64_t n_rays, const ot_rays* tail,
                                          uint32_t* fill, int64_t* result2, int64_t* msgs_host, void* stream) {
    if (!scene || !src || !tail || !fill || !result2 || !msgs_host)
        return fail(OT_ERR_INVALID, "ot_generate_and_trace_tail: null argument");
    if (n_rays < 1) return fail(OT_ERR_INVALID, "ot_generate_and_trace_tail: no rays");
    if (tail->nt != 2 || !tail->p || !tail->w || !tail->wl)
        return fail(OT_ERR_INVALID, "ot_generate_and_trace_tail: the tail storage has two sections and needs p, w and wl");
    if (tail->N < ot_tail_capacity(n_rays) || tail->N % 65536)
        return fail(OT_ERR_INVALID, "ot_generate_and_trace_tail: tail storage smaller than ot_tail_capacity(n_rays)");
    const RangeArgs* rg = nullptr;
    if (int rc = make_ranges(ranges, n_ranges, src, n_rays, &rg)) return rc;
    hipStream_t st = (hipStream_t)stream;
    TailOut T;
    T.p = tail->p;
    T.w = tail->w;
    T.wl = tail->wl;
    T.fill = fill;
    T.cap = tail->N;
    HIP_TRY(hipMemsetAsync(fill, 0, sizeof(uint32_t) * OT_TAIL_PIECES, st));
    if (int rc = launch_trace(scene, src, rg, nullptr, nullptr, seed, nullptr, stream, &T, n_rays)) return rc;
    hipLaunchKernelGGL(tail_seal_kernel, dim3(OT_TAIL_PIECES), dim3(256), 0, st, T, (long long*)result2);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));
    std::memcpy(msgs_host, scene->pin_msgs, sizeof(int64_t) * (size_t)(OT_N_INFOS * scene->h.nt + 1));
    return OT_OK;
}

// The living rays of a STORED chunk join a tail storage: `iterative_render` leaves the rays of its last chunk in the tracer
// (raytracer.py:1235-1267), so that chunk goes through the ray storage -- but its binning need not be a pass of its own (for
// 2^20 rays the fixed costs of the tile chain are most of it: 0.3-0.4 ms, a seventh of a rank's time when 2e8 rays are sharded
// over eight GPUs): the last sections of its living rays are appended to the tail of the render-only chunk before it, weights
// scaled to that chunk's rays (x chunk rays / tail rays, in f64, rounded once), and the two are binned together.
// Wave k of the range continues the round robin over the pieces behind the `waves_before` waves that filled the tail.
__global__ __launch_bounds__(256) void tail_append_kernel(ot_rays R, int64_t first, int64_t count, TailOut T, int64_t waves_before,
                                                          double weight_scale) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool have = q < count;
    const int64_t r = first + (have ? q : 0), N = R.N;
    const int nt = R.nt, kq = nt - 2;
    const float w = have ? R.w[r + N * kq] : 0.f;
    const bool alive = w > 0.f;
    const unsigned long long m = __ballot(alive);
    if (!m) return;
    const unsigned int n_alive = (unsigned int)__popcll(m);
    const unsigned int rank = __builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u));
    const unsigned int piece = (unsigned int)((waves_before + (q >> 6)) & (OT_TAIL_PIECES - 1));
    unsigned int q0 = 0;
    if (rank == 0 && alive) q0 = atomicAdd(&T.fill[piece], n_alive);
    q0 = __shfl(q0, __ffsll((long long)m) - 1);
    if (alive) {
        const unsigned int qs = q0 + rank;
        const int64_t slot = (((int64_t)(qs >> 6) * OT_TAIL_PIECES + piece) << 6) + (qs & 63u);
        const int64_t C = T.cap;
        double* px = T.p + slot;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            px[(2 * c) * C] = R.p[r + N * (kq + (int64_t)nt * c)];
            px[(2 * c + 1) * C] = R.p[r + N * (kq + 1 + (int64_t)nt * c)];
        }
        T.w[slot] = (float)((double)w * weight_scale);
        T.w[C + slot] = 0.f;
        T.wl[slot] = R.wl[r];
    }
}

extern "C" int ot_tail_append(const ot_rays* rays, int64_t first, int64_t count, double weight_scale, int64_t rays_before,
                              const ot_rays* tail, uint32_t* fill, int64_t* result2, void* stream) {
    if (!rays || !tail || !fill || !result2) return fail(OT_ERR_INVALID, "ot_tail_append: null argument");
    if (!rays->p || !rays->w || !rays->wl || rays->nt < 2) return fail(OT_ERR_INVALID, "ot_tail_append: the ray storage needs p, w, wl and two sections");
    if (first < 0 || count < 0 || first + count > rays->N || rays_before < 0) return fail(OT_ERR_INVALID, "ot_tail_append: range outside the storage");
    if (tail->nt != 2 || !tail->p || !tail->w || !tail->wl) return fail(OT_ERR_INVALID, "ot_tail_append: the tail storage has two sections and needs p, w and wl");
    if (!(weight_scale > 0.0) || !std::isfinite(weight_scale)) return fail(OT_ERR_INVALID, "ot_tail_append: weight_scale must be positive");
    const int64_t waves_before = (rays_before + 63) / 64, waves = (count + 63) / 64;
    if (tail->N % 65536 || tail->N < 65536 * std::max<int64_t>(1, (waves_before + waves + OT_TAIL_PIECES - 1) / OT_TAIL_PIECES))
        return fail(OT_ERR_INVALID, "ot_tail_append: tail storage smaller than ot_tail_capacity(rays_before + count + 64)");
    if (int rc = require_device()) return rc;
    hipStream_t st = (hipStream_t)stream;
    TailOut T;
    T.p = tail->p;
    T.w = tail->w;
    T.wl = tail->wl;
    T.fill = fill;
    T.cap = tail->N;
    if (count) hipLaunchKernelGGL(tail_append_kernel, grid_for(count), dim3(256), 0, st, *rays, first, count, T, waves_before, weight_scale);
    hipLaunchKernelGGL(tail_seal_kernel, dim3(OT_TAIL_PIECES), dim3(256), 0, st, T, (long long*)result2);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));
    return OT_OK;
}

// ---- leaf entry points -------------------------------------------------------------------------------------
extern "C" int ot_surface_find_hit(const ot_surface* surf, int64_t n, const double* p, const double* s, double* p_hit,
                                   uint8_t* is_hit, uint8_t* ill, void* stream) {
    if (!surf || n < 0 || (n && (!p || !s || !p_hit || !is_hit || !ill))) return fail(OT_ERR_INVALID, "ot_surface_find_hit: bad argument");
    if (int rc = require_device()) return rc;
    LeafSurface ls;
    if (int rc = ls.init(surf, (hipStream_t)stream)) return rc;
    const SurfDev& d = ls.d;
    if (n == 0) return OT_OK;
    hipLaunchKernelGGL(find_hit_kernel, grid_for(n), dim3(256), 0, (hipStream_t)stream, d, n, p, s, p_hit, is_hit, ill);
    HIP_TRY(hipGetLastError());
    return OT_OK;
}

extern "C" int ot_surface_normals(const ot_surface* surf, int64_t n, const double* x, const double* y, double* normals,
                                  void* stream) {
    if (!surf || n < 0 || (n && (!x || !y || !normals))) return fail(OT_ERR_INVALID, "ot_surface_normals: bad argument");
    if (int rc = require_device()) return rc;
    LeafSurface ls;
    if (int rc = ls.init(surf, (hipStream_t)stream)) return rc;
    const SurfDev& d = ls.d;
    if (n == 0) return OT_OK;
    hipLaunchKernelGGL(normals_kernel, grid_for(n), dim3(256), 0, (hipStream_t)stream, d, n, x, y, normals);
    HIP_TRY(hipGetLastError());
    return OT_OK;
}

extern "C" int ot_surface_mask(const ot_surface* surf, int64_t n, const double* x, const double* y, uint8_t* mask,
                               void* stream) {
    if (!surf || n < 0 || (n && (!x || !y || !mask))) return fail(OT_ERR_INVALID, "ot_surface_mask: bad argument");
    if (int rc = require_device()) return rc;
    LeafSurface ls;
    if (int rc = ls.init(surf, (hipStream_t)stream)) return rc;
    const SurfDev& d = ls.d;
    if (n == 0) return OT_OK;
    hipLaunchKernelGGL(mask_kernel, grid_for(n), dim3(256), 0, (hipStream_t)stream, d, n, x, y, mask);
    HIP_TRY(hipGetLastError());
    return OT_OK;
}

extern "C" int ot_surface_values(const ot_surface* surf, int64_t n, const double* x, const double* y, double* z,
                                 void* stream) {
    if (!surf || n < 0 || (n && (!x || !y || !z))) return fail(OT_ERR_INVALID, "ot_surface_values: bad argument");
    if (int rc = require_device()) return rc;
    LeafSurface ls;
    if (int rc = ls.init(surf, (hipStream_t)stream)) return rc;
    const SurfDev& d = ls.d;
    if (n == 0) return OT_OK;
    hipLaunchKernelGGL(values_kernel, grid_for(n), dim3(256), 0, (hipStream_t)stream, d, n, x, y, z);
    HIP_TRY(hipGetLastError());
    return OT_OK;
}

extern "C" int ot_surface_hurb_props(const ot_surface* surf, int64_t n, const double* x, const double* y, double* a_,
                                     double* b_, double* b, uint8_t* inside, void* stream) {
    if (!surf || n < 0 || (n && (!x || !y || !a_ || !b_ || !b || !inside)))
        return fail(OT_ERR_INVALID, "ot_surface_hurb_props: bad argument");
    if (surf->kind != OT_SURF_RING && surf->kind != OT_SURF_SLIT)
        return fail(OT_ERR_UNSUPPORTED, "hurb_props is defined for ring and slit surfaces only");
    if (int rc = require_device()) return rc;
    LeafSurface ls;
    if (int rc = ls.init(surf, (hipStream_t)stream)) return rc;
    const SurfDev& d = ls.d;
    if (n == 0) return OT_OK;
    hipLaunchKernelGGL(hurb_props_kernel, grid_for(n), dim3(256), 0, (hipStream_t)stream, d, n, x, y, a_, b_, b, inside);
    HIP_TRY(hipGetLastError());
    return OT_OK;
}

extern "C" int ot_refraction_index(const ot_medium* medium, const double* table_pool, int64_t table_pool_len, int64_t n,
                                   const float* wl, double* out, void* stream) {
    if (!medium || n < 0 || (n && (!wl || !out))) return fail(OT_ERR_INVALID, "ot_refraction_index: bad argument");
    if ((medium->model == OT_N_DATA || medium->model == OT_N_LINES) &&
        (!table_pool || medium->tab_off < 0 || medium->tab_off + 2 * (int64_t)medium->tab_len > table_pool_len))
        return fail(OT_ERR_INVALID, "ot_refraction_index: table outside the pool");
    if (int rc = require_device()) return rc;
    if (n == 0) return OT_OK;
    hipLaunchKernelGGL(refraction_index_kernel, grid_for(n), dim3(256), 0, (hipStream_t)stream, *medium, table_pool, n, wl, out);
    HIP_TRY(hipGetLastError());
    return OT_OK;
}

// ---- detector + render -------------------------------------------------------------------------------------
enum { OT_WS_RENDER = 0, OT_WS_FUSED = 1, OT_WS_FUSED_HITS = 2, OT_WS_AUTO = 3, OT_WS_DET = 4 };
static ot_scratch::Lease workspace(int purpose, size_t bytes, hipStream_t st);

extern "C" int ot_detector_hits_multi(const ot_rays* rays, int64_t first, int64_t count, const ot_detector_req* reqs,
                                      int32_t n_reqs, void* stream) {
    if (!rays || !reqs || n_reqs < 1) return fail(OT_ERR_INVALID, "ot_detector_hits: null argument");
    if (n_reqs > OT_DET_MAX) return fail(OT_ERR_INVALID, "ot_detector_hits_multi: at most 8 detectors per call");
    if (!rays->p || !rays->w) return fail(OT_ERR_INVALID, "ot_detector_hits: ray storage has null buffers");
    if (first < 0 || count < 0 || first + count > rays->N) return fail(OT_ERR_INVALID, "ot_detector_hits: range outside the storage");
    for (int k = 0; k < n_reqs; k++) {
        const ot_detector_req& q = reqs[k];
        if (!q.detector || !q.ill_count) return fail(OT_ERR_INVALID, "ot_detector_hits: null argument");
        // ph and hw both NULL: extent-only request (no hit list is written)
        // compact lists may go without positions (weights and wavelengths: the detector spectrum)
        if (q.fill && (!q.hw || !q.wl_out || !q.xy_only || !rays->wl))
            return fail(OT_ERR_INVALID, "ot_detector_hits: a compact hit list needs hw, wl_out and xy_only");
        if (!q.fill && (!q.ph || !q.hw) && (q.ph || q.hw || !q.extent4))
            return fail(OT_ERR_INVALID, "ot_detector_hits: ph and hw may only be NULL together, and only with extent4");
        if (q.projection < OT_PROJ_NONE || q.projection > OT_PROJ_STEREOGRAPHIC) return fail(OT_ERR_INVALID, "unknown projection");
    }
    if (int rc = require_device()) return rc;
    hipStream_t st = (hipStream_t)stream;
    std::vector<LeafSurface> ls(n_reqs);
    std::vector<DetOne> host(n_reqs);
    bool numeric = false;
    int n_ext = 0;
    for (int k = 0; k < n_reqs; k++) {
        if (int rc = ls[k].init(reqs[k].detector, st)) return rc;
        n_ext += reqs[k].extent4 != nullptr;
    }
    if (count == 0) return OT_OK;
    // scratch: the detector records, then the extent slot tables -- a few KB from the kept pool (it used to come from the
    // stream-ordered pool; that cost 0.2 ms per call, and 7-58 ms whenever the driver was still busy with memory a large free
    // had returned to it, profiles/r3/readback_after_free.txt)
    const size_t o_slots = align_up(sizeof(DetOne) * OT_DET_MAX);
    const size_t total = o_slots + sizeof(unsigned long long) * 4 * OT_EXT_SLOTS * (size_t)OT_DET_MAX;
    const ot_scratch::Lease lease = workspace(OT_WS_DET, total, st);
    if (!lease) return fail(OT_ERR_HIP, "ot_detector_hits: no scratch memory");
    char* scratch = lease.p();
    unsigned long long* slots = (unsigned long long*)(scratch + o_slots);
    int e = 0;
    for (int k = 0; k < n_reqs; k++) {
        DetOne& d = host[k];
        std::memset(&d, 0, sizeof(d));
        d.det = ls[k].d;
        d.Rcurv = reqs[k].detector->R;
        if (reqs[k].crop4) d.crop = {reqs[k].crop4[0], reqs[k].crop4[1], reqs[k].crop4[2], reqs[k].crop4[3], 1};
        d.ph = reqs[k].ph;
        d.hw = reqs[k].hw;
        d.ill = (unsigned long long*)reqs[k].ill_count;
        d.projection = reqs[k].projection;
        d.xy_only = reqs[k].xy_only != 0;
        d.wl_out = reqs[k].wl_out;
        d.fill = reqs[k].fill;
        d.piece_shift = hit_piece_shift(count);
        if (reqs[k].extent4) d.ext_slots = slots + (size_t)4 * OT_EXT_SLOTS * e++;
        numeric = numeric || !(d.det.kind == OT_SURF_CONIC || d.det.flat);
    }
    hipError_t err = hipSuccess;
    if (n_reqs > 1) err = hipMemcpyAsync(scratch, host.data(), sizeof(DetOne) * n_reqs, hipMemcpyHostToDevice, st);
    if (err == hipSuccess) {
        if (n_ext) hipLaunchKernelGGL(extent_init_kernel, dim3(n_ext), dim3(4 * OT_EXT_SLOTS), 0, st, slots);
        if (n_reqs == 1) {  // the record travels in the kernel arguments (scalar registers)
            if (numeric)
                hipLaunchKernelGGL(detector_kernel<true>, grid_for(count), dim3(256), 0, st, *rays, first, count, host[0]);
            else
                hipLaunchKernelGGL(detector_kernel<false>, grid_for(count), dim3(256), 0, st, *rays, first, count, host[0]);
        } else {
            const dim3 g = grid_for(count), b(256);
            const DetOne* dd = (const DetOne*)scratch;
#define OT_LAUNCH_DET(NUM, ND) hipLaunchKernelGGL((detector_multi_kernel<NUM, ND>), g, b, 0, st, *rays, first, count, dd, n_reqs)
            if (numeric) {
                OT_LAUNCH_DET(true, 8);
            } else {
                if (n_reqs <= 2) OT_LAUNCH_DET(false, 2);
                else if (n_reqs <= 4) OT_LAUNCH_DET(false, 4);
                else OT_LAUNCH_DET(false, 8);
            }
#undef OT_LAUNCH_DET
        }
        e = 0;
        for (int k = 0; k < n_reqs; k++)
            if (reqs[k].extent4)
                hipLaunchKernelGGL(extent_final_kernel, dim3(1), dim3(64), 0, st, slots + (size_t)4 * OT_EXT_SLOTS * e++, reqs[k].extent4);
        err = hipGetLastError();
    }
    HIP_TRY(err);
    return OT_OK;
}

extern "C" int64_t ot_hit_piece_len(int64_t count) { return hit_piece_len(count); }

extern "C" int ot_detector_hits(const ot_rays* rays, int64_t first, int64_t count, const ot_surface* detector,
                                int32_t projection, const double* crop4, double* ph, float* hw, double* extent4,
                                int64_t* ill_count, void* stream) {
    ot_detector_req q;
    q.wl_out = nullptr;
    q.fill = nullptr;
    q.detector = detector;
    q.projection = projection;
    q.xy_only = 0;
    q.crop4 = crop4;
    q.ph = ph;
    q.hw = hw;
    q.extent4 = extent4;
    q.ill_count = ill_count;
    return ot_detector_hits_multi(rays, first, count, &q, 1, stream);
}

extern "C" int ot_sphere_projection(const ot_surface* surf, int32_t projection, int64_t n, const double* p, double* out,
                                    void* stream) {
    if (!surf || n < 0 || (n && (!p || !out))) return fail(OT_ERR_INVALID, "ot_sphere_projection: bad argument");
    if (surf->kind != OT_SURF_CONIC || surf->k != 0.0) return fail(OT_ERR_INVALID, "sphere projection needs a spherical surface");
    if (projection < OT_PROJ_NONE || projection > OT_PROJ_STEREOGRAPHIC) return fail(OT_ERR_INVALID, "unknown projection");
    if (int rc = require_device()) return rc;
    if (n == 0) return OT_OK;
    hipLaunchKernelGGL(projection_kernel, grid_for(n), dim3(256), 0, (hipStream_t)stream, surf->pos[0], surf->pos[1],
                       surf->pos[2], surf->R, projection, n, p, out);
    HIP_TRY(hipGetLastError());
    return OT_OK;
}

// Scratch of the binning paths (hit records, slabs: up to ~25 B per ray and image): ot_scratch.hpp.  One pool per process,
// blocks keyed by (device, stream, purpose), LEASED for the launches of a call (an automatic image keeps its lease from begin
// to finish / cancel), kept between calls, capped (OT_SCRATCH_CAP_GB in the environment or ot_scratch_set_cap; 64 GB of the
// 288), idle blocks freed least recently used first.  torch's allocator, which owns the ray storage, cannot see this memory:
// ot_scratch_trim() hands the idle part back on request.
static void* ws_alloc(size_t bytes) {
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess || !p) {
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
}
static void ws_release(void* p) { (void)hipFree(p); }  // (waits for the work that may still use p)
static void ws_sync() { (void)hipDeviceSynchronize(); }
static ot_scratch::Pool& scratch_pool() {
    static ot_scratch::Pool pool({ws_alloc, ws_release, ws_sync}, [] {
        const char* v = std::getenv("OT_SCRATCH_CAP_GB");
        const double gb = v ? std::atof(v) : 64.0;
        return (size_t)((gb > 0.0 ? gb : 64.0) * 1e9);
    }());
    return pool;
}

// -> lease on a block of at least `bytes` for this device, stream and purpose; empty when out of memory (the callers fall
// back to paths without scratch)
static ot_scratch::Lease workspace(int purpose, size_t bytes, hipStream_t st) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return {};
    return ot_scratch::Lease(scratch_pool(), dev, purpose, (void*)st, bytes);
}

extern "C" int ot_scratch_trim(void) {
    if (int rc = require_device()) return rc;
    scratch_pool().trim();
    return OT_OK;
}

extern "C" int ot_scratch_set_cap(int64_t bytes) {
    if (bytes < 0) return fail(OT_ERR_INVALID, "ot_scratch_set_cap: negative cap");
    scratch_pool().set_cap((size_t)bytes);
    return OT_OK;
}

extern "C" int ot_scratch_stats(int64_t* kept_bytes, int32_t* blocks, int32_t* leased) {
    size_t kept = 0;
    int nb = 0, busy = 0;
    scratch_pool().stats(&kept, &nb, &busy);
    if (kept_bytes) *kept_bytes = (int64_t)kept;
    if (blocks) *blocks = nb;
    if (leased) *leased = busy;
    return OT_OK;
}

#define OT_TILE_MIN_HITS (1ll << 21)  // shorter lists: the direct kernel alone

static int render_accumulate(int64_t n, const unsigned int* fill, const double* px, const double* py, const float* w,
                             const float* wl, const double extent[4], int32_t Nx, int32_t Ny, double* hist, void* stream,
                             double weight_scale = 1.0);

extern "C" int ot_render_accumulate(int64_t n, const double* px, const double* py, const float* w, const float* wl,
                                    const double extent[4], int32_t Nx, int32_t Ny, double* hist, void* stream) {
    return render_accumulate(n, nullptr, px, py, w, wl, extent, Nx, Ny, hist, stream);
}

extern "C" int ot_render_accumulate_compact(int64_t n, const uint32_t* fill, const double* px, const double* py,
                                            const float* w, const float* wl, const double extent[4], int32_t Nx,
                                            int32_t Ny, double* hist, void* stream) {
    if (!fill) return fail(OT_ERR_INVALID, "ot_render_accumulate_compact: fill counts missing");
    return render_accumulate(n, fill, px, py, w, wl, extent, Nx, Ny, hist, stream);
}

static int render_accumulate(int64_t n, const unsigned int* fill, const double* px, const double* py, const float* w,
                             const float* wl, const double extent[4], int32_t Nx, int32_t Ny, double* hist, void* stream,
                             double weight_scale) {
    if (n < 0 || !extent || !hist || Nx < 1 || Ny < 1 || (n && (!px || !py || !w || !wl)))
        return fail(OT_ERR_INVALID, "ot_render_accumulate: bad argument");
    if (int rc = require_device()) return rc;
    if (n == 0) return OT_OK;
    RenderArgs a;
    a.x0 = extent[0];
    a.y0 = extent[2];
    a.x1 = extent[1];
    a.y1 = extent[3];
    a.fx = (double)Nx / (extent[1] - extent[0]);  // Nx / s[0]  misc.py:75
    a.fy = (double)Ny / (extent[3] - extent[2]);
    a.Nx = Nx;
    a.Ny = Ny;
    a.ws = weight_scale;
    const double* table = observer_table_device();
    if (!table) return fail(OT_ERR_HIP, "could not upload the CIE observer table");
    // one 1024-thread workgroup per CU (grid-stride): LDS-privatised histogram, see render_kernel
    int dev = 0, cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
        cus = prop.multiProcessorCount;
    int64_t blocks = (n + 1023) / 1024;
    if (blocks > cus) blocks = cus;
    hipStream_t st = (hipStream_t)stream;
    // OT_RENDER_PATH = direct | tiles pins the path (tests, profiling); default: by list length and probe
    const char* pin = std::getenv("OT_RENDER_PATH");
    const bool pin_direct = pin && !std::strcmp(pin, "direct"), pin_tiles = pin && !std::strcmp(pin, "tiles");
    if (pin_direct || (n < OT_TILE_MIN_HITS && !pin_tiles)) {
        hipLaunchKernelGGL(render_kernel, dim3((unsigned)blocks), dim3(1024), 0, st, n, px, py, w, wl, a, table, hist, (const int*)nullptr, fill);
        HIP_TRY(hipGetLastError());
        return OT_OK;
    }
    // long lists: a probe decides on the device whether the direct kernel or the tile path bins them
    // (ot_render_tiles.hpp); both are enqueued, the one that is not needed returns at once
    TileArgs t;
    t.a = a;
    t.tx = (Nx + OT_TILE_W - 1) / OT_TILE_W;
    t.ty = (Ny + OT_TILE_W - 1) / OT_TILE_W;
    t.K = t.tx * t.ty;
    if (t.K > OT_TILE_MAX) {  // no image of RenderImage is this large; stay on the direct path
        hipLaunchKernelGGL(render_kernel, dim3((unsigned)blocks), dim3(1024), 0, st, n, px, py, w, wl, a, table, hist, (const int*)nullptr, fill);
        HIP_TRY(hipGetLastError());
        return OT_OK;
    }
    t.n = n;
    t.piece = fill ? hit_piece_len(n) : ((n + OT_TILE_PIECES - 1) / OT_TILE_PIECES + 1023) / 1024 * 1024;
    t.chunk = ((n + 1023) / 1024 + 1023) / 1024 * 1024;
    if (t.chunk < 16384) t.chunk = 16384;
    t.max_chunks = (int32_t)(n / t.chunk + t.K + 1);
    size_t off = 0;
    auto carve = [&](size_t bytes) {
        size_t o = off;
        off = (off + bytes + 255) / 256 * 256;
        return o;
    };
    const size_t o_spread = carve(sizeof(int));
    const size_t o_counts = carve(sizeof(unsigned int) * OT_TILE_PIECES * (size_t)t.K);
    const size_t o_tot = carve(sizeof(unsigned long long) * t.K);
    const size_t o_starts = carve(sizeof(unsigned long long) * (t.K + 1));
    const size_t o_cstart = carve(sizeof(int) * (t.K + 1));
    const size_t o_rec = carve(sizeof(TileRec) * (size_t)n);
    const size_t o_slabs = carve(sizeof(double) * OT_TILE_PX * 4 * (size_t)t.max_chunks);
    const ot_scratch::Lease lease = workspace(OT_WS_RENDER, off, st);
    char* ws = lease.p();
    if (!ws) {
        // no room for the hit records (12 B per hit): the direct kernel needs no scratch
        hipLaunchKernelGGL(render_kernel, dim3((unsigned)blocks), dim3(1024), 0, st, n, px, py, w, wl, a, table, hist, (const int*)nullptr, fill);
        HIP_TRY(hipGetLastError());
        return OT_OK;
    }
    TileWork wk;
    wk.fill = fill;
    wk.spread = (int*)(ws + o_spread);
    wk.counts = (unsigned int*)(ws + o_counts);
    wk.tot = (unsigned long long*)(ws + o_tot);
    wk.starts = (unsigned long long*)(ws + o_starts);
    wk.chunk_start = (int*)(ws + o_cstart);
    wk.rec = (TileRec*)(ws + o_rec);
    wk.slabs = (double*)(ws + o_slabs);
    const int lds_probe = OT_TILE_PROBE_SET * (int)sizeof(int);
    const int lds_accum = (OT_TILE_PX * 4 + OT_OBS_N * 6) * (int)sizeof(double);  // tile + (value, difference) observer table
    static thread_local bool lds_set[64] = {false};
    if (dev >= 0 && dev < 64 && !lds_set[dev]) {
        HIP_TRY(hipFuncSetAttribute((const void*)tile_probe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_probe));
        HIP_TRY(hipFuncSetAttribute((const void*)tile_accum_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_accum));
        lds_set[dev] = true;
    }
    if (pin_tiles)
        HIP_TRY(hipMemsetAsync(wk.spread, 1, sizeof(int), st));
    else
        hipLaunchKernelGGL(tile_probe_kernel, dim3(1), dim3(1024), lds_probe, st, t, px, py, w, wk.spread, fill);
    hipLaunchKernelGGL(render_kernel, dim3((unsigned)blocks), dim3(1024), 0, st, n, px, py, w, wl, a, table, hist, (const int*)wk.spread, fill);
    hipLaunchKernelGGL(tile_count_kernel, dim3(OT_TILE_PIECES), dim3(1024), 0, st, t, px, py, w, wk);
    hipLaunchKernelGGL(tile_cursor_kernel, dim3((unsigned)((t.K + 3) / 4)), dim3(256), 0, st, t, wk, wk.tot);
    hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, st, t, wk, (const unsigned long long*)wk.tot);
    hipLaunchKernelGGL(tile_scatter_kernel, dim3(OT_TILE_PIECES), dim3(1024), 0, st, t, px, py, w, wl, wk);
    hipLaunchKernelGGL(tile_accum_kernel, dim3((unsigned)t.max_chunks), dim3(1024), lds_accum, st, t, table, wk);
    hipLaunchKernelGGL(tile_reduce_kernel, dim3(OT_TILE_PX / 256, (unsigned)t.K), dim3(256), 0, st, t, wk, hist);
    HIP_TRY(hipGetLastError());
    return OT_OK;
}

// ---- detector image in one pass (ot_detector_fused.hpp) ------------------------------------------------------
// a small record to device memory through the kernel arguments (no staging copy, nothing for the host to wait for)
template <class T>
__global__ void put_kernel(T v, T* dst) {
    if (threadIdx.x == 0) *dst = v;
}

struct FuseIndexAll {
    FuseIndex v[OT_DET_MAX];
};

static int cu_count();

// one detector, an image of few tiles: the tile kernel with line buffers (OT_TILE_LINEBUF=0 in the environment: the plain one)
static bool fuse_use_linebuf(int K) {
    const char* v = std::getenv("OT_TILE_LINEBUF");
    return K <= OT_LB_MAXK && !(v && v[0] == '0');
}

// the one-detector tile kernels stage their records in LDS: with many tiles more than the 64 KB a kernel gets unasked
static int fuse_tiles_allow_lds(int dev) {
    static thread_local bool done[64] = {false};
    if (dev < 0 || dev >= 64 || done[dev]) return OT_OK;
    const int most = 96 * 1024;
    HIP_TRY(hipFuncSetAttribute((const void*)fuse_tiles_kernel<false, 1, 1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, most));
    HIP_TRY(hipFuncSetAttribute((const void*)fuse_tiles_kernel<false, 1, 2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, most));
    HIP_TRY(hipFuncSetAttribute((const void*)fuse_tiles_kernel<false, 1, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, most));
    HIP_TRY(hipFuncSetAttribute((const void*)fuse_tiles_kernel<false, 1, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, most));
    const int lb = (int)fuse_lb_lds(OT_LB_MAXK);
    HIP_TRY(hipFuncSetAttribute((const void*)fuse_tiles_lb_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lb));
    HIP_TRY(hipFuncSetAttribute((const void*)fuse_tiles_lb_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lb));
    done[dev] = true;
    return OT_OK;
}

extern "C" int ot_detector_images(const ot_rays* rays, int64_t first, int64_t count, const ot_detector_image_req* reqs,
                                  int32_t n_reqs, void* stream) {
    if (!rays || !reqs || n_reqs < 1) return fail(OT_ERR_INVALID, "ot_detector_images: null argument");
    if (n_reqs > OT_DET_MAX) return fail(OT_ERR_INVALID, "ot_detector_images: at most 8 detectors per call");
    if (!rays->p || !rays->w || !rays->wl) return fail(OT_ERR_INVALID, "ot_detector_images: ray storage has null buffers");
    if (first < 0 || count < 0 || first + count > rays->N) return fail(OT_ERR_INVALID, "ot_detector_images: range outside the storage");
    for (int k = 0; k < n_reqs; k++) {
        const ot_detector_image_req& q = reqs[k];
        if (!q.detector || !q.hist || !q.ill_count || q.Nx < 1 || q.Ny < 1) return fail(OT_ERR_INVALID, "ot_detector_images: bad request");
        if (!(q.extent[1] > q.extent[0]) || !(q.extent[3] > q.extent[2])) return fail(OT_ERR_INVALID, "ot_detector_images: empty image extent");
        if ((int64_t)q.Nx * q.Ny > (1ll << 27)) return fail(OT_ERR_INVALID, "ot_detector_images: image too large");
        if (!std::isfinite(q.weight_scale)) return fail(OT_ERR_INVALID, "ot_detector_images: weight_scale is not finite");
        if (q.projection < OT_PROJ_NONE || q.projection > OT_PROJ_STEREOGRAPHIC) return fail(OT_ERR_INVALID, "unknown projection");
    }
    if (int rc = require_device()) return rc;
    hipStream_t st = (hipStream_t)stream;
    std::vector<LeafSurface> ls(n_reqs);
    for (int k = 0; k < n_reqs; k++)
        if (int rc = ls[k].init(reqs[k].detector, st)) return rc;
    if (count == 0) return OT_OK;
    // Detectors that need the numeric hit search (aspheric, tilted, spline surfaces) or a sphere projection with
    // transcendentals take the two-step chain: with the Illinois loop and the projection polynomials inside, the fused
    // kernels need every vector register there is and lose to hit search + binning (C3, 5e7 rays: 3.1 against 2.1 ms).
    for (int k = 0; k < n_reqs; k++) {
        const ot_detector_image_req& q = reqs[k];
        const bool closed = q.detector->kind <= OT_SURF_CONIC || q.detector->z_min == q.detector->z_max;
        const bool plain = q.projection == OT_PROJ_NONE || q.projection == OT_PROJ_ORTHOGRAPHIC;
        if (closed && plain) continue;
        // this request alone through ot_detector_hits + ot_render_accumulate, the others through the fused kernels
        const size_t o_hw = align_up(sizeof(double) * 2 * (size_t)count);
        const ot_scratch::Lease hits = workspace(OT_WS_FUSED_HITS, o_hw + sizeof(float) * (size_t)count, st);
        char* tmp = hits.p();
        if (!tmp) return fail(OT_ERR_HIP, "ot_detector_images: no memory for the hit list");
        ot_detector_req dq;
        dq.detector = q.detector;
        dq.projection = q.projection;
        dq.xy_only = 1;
        dq.crop4 = q.crop4;
        dq.ph = (double*)tmp;
        dq.hw = (float*)(tmp + o_hw);
        dq.extent4 = nullptr;
        dq.wl_out = nullptr;
        dq.fill = nullptr;
        dq.ill_count = q.ill_count;
        int rc = ot_detector_hits_multi(rays, first, count, &dq, 1, stream);
        if (!rc) rc = render_accumulate(count, nullptr, dq.ph, dq.ph + count, dq.hw, rays->wl + first, q.extent, q.Nx, q.Ny, q.hist, stream,
                                        q.weight_scale);
        if (rc) return rc;
        std::vector<ot_detector_image_req> rest;
        for (int j = 0; j < n_reqs; j++)
            if (j != k) rest.push_back(reqs[j]);
        return rest.empty() ? OT_OK : ot_detector_images(rays, first, count, rest.data(), (int32_t)rest.size(), stream);
    }
    const double* table = observer_table_device();
    if (!table) return fail(OT_ERR_HIP, "could not upload the CIE observer table");
    int dev = 0;
    (void)hipGetDevice(&dev);
    const int cus = cu_count();
    // a tile-kernel workgroup keeps 20 B of LDS per (detector, tile): more tiles than fit -> two calls
    int KT_all = 0;
    for (int k = 0; k < n_reqs; k++)
        KT_all += ((reqs[k].Nx + OT_TILE_W - 1) / OT_TILE_W) * ((reqs[k].Ny + OT_TILE_W - 1) / OT_TILE_W);
    if (n_reqs > 1 && KT_all > OT_FUSE_LDS_ENTRIES) {
        const int h = n_reqs / 2;
        if (int rc = ot_detector_images(rays, first, count, reqs, h, stream)) return rc;
        return ot_detector_images(rays, first, count, reqs + h, n_reqs - h, stream);
    }
    // OT_RENDER_PATH = direct | tiles pins the binning path (tests, profiling); default: by ray count and probe
    const char* pin = std::getenv("OT_RENDER_PATH");
    const bool pin_direct = pin && !std::strcmp(pin, "direct"), pin_tiles = pin && !std::strcmp(pin, "tiles");
    // (the threshold counts the hits a call may bin: rays x detectors -- the last, short chunk of an iterative render with six
    // positions then stays on the tile path instead of 6e6 global atomic quadruples)
    const bool want_tiles = !pin_direct && (pin_tiles || count * (int64_t)n_reqs >= OT_TILE_MIN_HITS);
    if (count >= (1ll << 31)) return fail(OT_ERR_UNSUPPORTED, "ot_detector_images: at most 2^31 - 1 rays per call");
    // tile kernel: two rays per thread and sub-block where the images have at most 1024 tiles (10-bit tile numbers)
    bool small_k = n_reqs == 1;
    for (int k = 0; k < n_reqs; k++)
        small_k = small_k && ((reqs[k].Nx + OT_TILE_W - 1) / OT_TILE_W) * ((reqs[k].Ny + OT_TILE_W - 1) / OT_TILE_W) <= 1024;
    // one detector with a closed-form hit and an image of few tiles: the kernel with line buffers, one 1024-thread workgroup per CU
    // (every request that reaches this point has a closed-form hit and no sphere projection)
    const bool linebuf = n_reqs == 1 && fuse_use_linebuf(((reqs[0].Nx + OT_TILE_W - 1) / OT_TILE_W) * ((reqs[0].Ny + OT_TILE_W - 1) / OT_TILE_W));
    const int64_t brt = linebuf ? OT_LB_BR * OT_LB_RPT : OT_FUSE_BR * (small_k ? 2 : 1);
    const unsigned n_wg = (unsigned)std::min<int64_t>((linebuf ? 1 : OT_FUSE_WG_PER_CU) * (int64_t)cus, (count + brt - 1) / brt);
    const int64_t piece = ((count + n_wg - 1) / n_wg + brt - 1) / brt * brt;

    std::vector<FuseOne> host(n_reqs);
    bool numeric = false, general = false;  // general: numeric hit search or a sphere projection somewhere
    int KT = 0, Kmax = 1;
    uint32_t capmax = 1;
    size_t off = 0;
    auto carve = [&](size_t bytes) {
        size_t o = off;
        off = (off + bytes + 255) / 256 * 256;
        return o;
    };
    const size_t o_dets = carve(sizeof(FuseOne) * n_reqs);
    const size_t o_flags = carve(sizeof(int) * 4 * n_reqs);  // per detector: spread, -, overflow, pad
    const size_t o_pcnt = carve(sizeof(int) * 2 * n_reqs);   // probe: distinct pixels, workgroups done
    const size_t o_pset = carve(sizeof(int) * OT_TILE_PROBE_SET * (size_t)n_reqs);  // probe: pixel sets
    std::vector<size_t> o_ctile(n_reqs), o_cfill(n_reqs), o_rec(n_reqs);
    for (int k = 0; k < n_reqs; k++) {
        FuseOne& f = host[k];
        std::memset(&f, 0, sizeof(f));
        const ot_detector_image_req& q = reqs[k];
        f.det = ls[k].d;
        f.Rcurv = q.detector->R;
        if (q.crop4) f.crop = {q.crop4[0], q.crop4[1], q.crop4[2], q.crop4[3], 1};
        f.projection = q.projection;
        f.a.x0 = q.extent[0];
        f.a.x1 = q.extent[1];
        f.a.y0 = q.extent[2];
        f.a.y1 = q.extent[3];
        f.a.fx = (double)q.Nx / (q.extent[1] - q.extent[0]);  // Nx / s[0]  misc.py:75
        f.a.fy = (double)q.Ny / (q.extent[3] - q.extent[2]);
        f.a.Nx = q.Nx;
        f.a.Ny = q.Ny;
        f.a.ws = q.weight_scale;
        f.tx = (q.Nx + OT_TILE_W - 1) / OT_TILE_W;
        f.K = f.tx * ((q.Ny + OT_TILE_W - 1) / OT_TILE_W);
        f.ill = (unsigned long long*)q.ill_count;
        f.hist = q.hist;
        f.tiles_ok = want_tiles && f.K <= OT_TILE_MAX && f.K <= OT_FUSE_LDS_ENTRIES;
        f.koff = KT;
        if (f.tiles_ok) {
            KT += f.K;
            // a workgroup hands out chunks of its own part of the pool; every (workgroup, tile) pair leaves at most one
            // chunk partly filled
            f.per_wg = (uint32_t)((piece + OT_FUSE_CH - 1) / OT_FUSE_CH + f.K + 2);
            f.cap = (uint32_t)std::min<int64_t>((int64_t)f.per_wg * n_wg, 0xffffffffll / OT_FUSE_CH - 1);  // record numbers: 32 bits
            o_ctile[k] = carve(sizeof(uint32_t) * f.cap);
            o_cfill[k] = carve(sizeof(uint32_t) * f.cap);
            o_rec[k] = carve(sizeof(TileRec) * (size_t)f.cap * OT_FUSE_CH);
            Kmax = std::max(Kmax, f.K);
            capmax = std::max(capmax, f.cap);
        }
        numeric = numeric || !(f.det.kind == OT_SURF_CONIC || f.det.flat);
        general = general || !(f.det.kind == OT_SURF_CONIC || f.det.flat) || (q.projection != OT_PROJ_NONE && q.projection != OT_PROJ_ORTHOGRAPHIC);
    }
    // second pass (chunks grouped by tile, accumulation, reduction): every detector its own index and slabs, so that one launch
    // per step serves them all (six positions of an iterative render: 400 accumulation workgroups each, 1.6 rounds over 256 CUs
    // when launched one after the other)
    const int n_idx = KT ? n_reqs : 0;
    const size_t o_ixs = carve(sizeof(FuseIndexAll));
    const size_t o_tn = carve(sizeof(unsigned int) * Kmax * (size_t)n_idx);
    const size_t o_ts = carve(sizeof(unsigned int) * (Kmax + 1) * (size_t)n_idx);
    const size_t o_list = carve(sizeof(unsigned int) * (size_t)capmax * n_idx);
    const size_t o_ws = carve(sizeof(unsigned int) * (Kmax + 1) * (size_t)n_idx);
    // one slab per accumulation workgroup: a tile with n chunks takes ceil(n / OT_FUSE_CPW) of them
    const unsigned n_slabs = (unsigned)((capmax + OT_FUSE_CPW - 1) / OT_FUSE_CPW) + (unsigned)Kmax;
    const size_t o_slabs = carve(sizeof(double) * OT_TILE_PX * 4 * (size_t)n_slabs * n_idx);
    ot_scratch::Lease lease = workspace(OT_WS_FUSED, off, st);
    char* ws = lease.p();
    if (!ws) {
        if (!KT) return fail(OT_ERR_HIP, "ot_detector_images: no scratch memory");
        // no room for the records: bin directly (needs the flags and the detector table only)
        KT = 0;
        for (auto& f : host) f.tiles_ok = 0;
        off = o_flags + sizeof(int) * 4 * n_reqs + 256;
        lease = workspace(OT_WS_FUSED, off, st);
        ws = lease.p();
        if (!ws) return fail(OT_ERR_HIP, "ot_detector_images: no scratch memory");
    }
    int* flags = (int*)(ws + o_flags);
    for (int k = 0; k < n_reqs; k++) {
        FuseOne& f = host[k];
        f.spread = flags + 4 * k;
        f.overflow = flags + 4 * k + 2;
        if (f.tiles_ok) {
            f.chunk_tile = (uint32_t*)(ws + o_ctile[k]);
            f.chunk_fill = (uint32_t*)(ws + o_cfill[k]);
            f.rec = (TileRec*)(ws + o_rec[k]);
        }
    }
    hipError_t err = hipMemsetAsync(flags, 0, sizeof(int) * 4 * n_reqs, st);
    if (err == hipSuccess) err = hipMemcpyAsync(ws + o_dets, host.data(), sizeof(FuseOne) * n_reqs, hipMemcpyHostToDevice, st);
    const FuseOne* dd = (const FuseOne*)(ws + o_dets);
    static thread_local bool lds_set[64] = {false};
    const int lds_accum = (OT_TILE_PX * 4 + OT_OBS_N * 6) * (int)sizeof(double);  // tile + (value, difference) observer table
    if (err == hipSuccess && dev >= 0 && dev < 64 && !lds_set[dev]) {
        HIP_TRY(hipFuncSetAttribute((const void*)fuse_accum_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_accum));
        HIP_TRY(hipFuncSetAttribute((const void*)fuse_accum_multi_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_accum));
        lds_set[dev] = true;
    }
    if (int rc = fuse_tiles_allow_lds(dev)) return rc;
    if (err == hipSuccess) {
        if (KT) {
            if (pin_tiles) {  // spread = 1 for every detector with a pool
                std::vector<int> hf(4 * n_reqs, 0);
                for (int k = 0; k < n_reqs; k++) hf[4 * k] = host[k].tiles_ok;
                err = hipMemcpyAsync(flags, hf.data(), sizeof(int) * 4 * n_reqs, hipMemcpyHostToDevice, st);
                (void)hipStreamSynchronize(st);  // hf goes out of scope
            } else {
                int* pcnt = (int*)(ws + o_pcnt);
                int* pset = (int*)(ws + o_pset);
                err = hipMemsetAsync(pcnt, 0, sizeof(int) * 2 * n_reqs, st);
                if (err == hipSuccess) err = hipMemsetAsync(pset, 0xff, sizeof(int) * OT_TILE_PROBE_SET * (size_t)n_reqs, st);
                const dim3 pg(n_reqs, OT_TILE_PROBE / OT_FUSE_PROBE_WG);
                if (numeric)
                    hipLaunchKernelGGL(fuse_probe_kernel<true>, pg, dim3(OT_FUSE_PROBE_WG), 0, st, *rays, first, count, dd, pset, pcnt);
                else
                    hipLaunchKernelGGL(fuse_probe_kernel<false>, pg, dim3(OT_FUSE_PROBE_WG), 0, st, *rays, first, count, dd, pset, pcnt);
            }
        }
        const unsigned blocks = (unsigned)std::min<int64_t>(cus, (count + 1023) / 1024);
        const size_t lds_tiles = fuse_tiles_lds(KT, general ? 8 : (n_reqs == 1 ? 1 : 2), small_k ? 2 : 1, false);
        ot_rays part = *rays;  // the tile kernel addresses its rays with 32 bits from the start of the range
        part.p += first;
        part.w += first;
        part.wl += first;
#define OT_LAUNCH_FUSE(NUM, ND, RP)                                                                                       \
    do {                                                                                                                  \
        hipLaunchKernelGGL((fuse_direct_kernel<NUM, ND>), dim3(blocks), dim3(1024), 0, st, *rays, first, count, dd, n_reqs, table); \
        if (KT)                                                                                                           \
            hipLaunchKernelGGL((fuse_tiles_kernel<NUM, ND, RP>), dim3(n_wg), dim3(OT_FUSE_BR), lds_tiles, st, part,        \
                               (uint32_t)count, dd, n_reqs, KT, (uint32_t)piece);                                          \
    } while (0)
        if (general) {
            OT_LAUNCH_FUSE(true, 8, 1);
        } else if (linebuf) {
            hipLaunchKernelGGL((fuse_direct_kernel<false, 1>), dim3(blocks), dim3(1024), 0, st, *rays, first, count, dd, n_reqs, table);
            if (KT)
                hipLaunchKernelGGL(fuse_tiles_lb_kernel<false>, dim3(n_wg), dim3(OT_LB_BR), fuse_lb_lds(KT), st, part, (uint32_t)count,
                                   dd, KT, (uint32_t)piece);
        } else if (n_reqs == 1) {
            if (small_k) OT_LAUNCH_FUSE(false, 1, 2); else OT_LAUNCH_FUSE(false, 1, 1);
        } else if (rays->nt == 2) {  // two sections (a tail storage): every hit from the prefetched pair, no section search
#define OT_LAUNCH_FUSE_PAIR(ND)                                                                                           \
    do {                                                                                                                  \
        hipLaunchKernelGGL((fuse_direct_kernel<false, ND>), dim3(blocks), dim3(1024), 0, st, *rays, first, count, dd, n_reqs, table); \
        if (KT)                                                                                                           \
            hipLaunchKernelGGL((fuse_tiles_kernel<false, ND, 1, false, true>), dim3(n_wg), dim3(OT_FUSE_BR), lds_tiles, st, part, \
                               (uint32_t)count, dd, n_reqs, KT, (uint32_t)piece);                                          \
    } while (0)
            if (n_reqs <= 2) OT_LAUNCH_FUSE_PAIR(2); else if (n_reqs <= 4) OT_LAUNCH_FUSE_PAIR(4); else OT_LAUNCH_FUSE_PAIR(8);
#undef OT_LAUNCH_FUSE_PAIR
        } else if (n_reqs <= 2) {
            OT_LAUNCH_FUSE(false, 2, 1);
        } else if (n_reqs <= 4) {
            OT_LAUNCH_FUSE(false, 4, 1);
        } else {
            OT_LAUNCH_FUSE(false, 8, 1);
        }
#undef OT_LAUNCH_FUSE
        err = hipGetLastError();
        // tile path, all detectors per launch: chunks grouped by tile, LDS accumulation, slabs summed into the images
        if (err == hipSuccess && KT) {
            FuseIndexAll ixs;
            std::memset(&ixs, 0, sizeof(ixs));
            for (int k = 0; k < n_reqs; k++) {
                FuseIndex& ix = ixs.v[k];
                ix.tile_n = (unsigned int*)(ws + o_tn) + (size_t)Kmax * k;
                ix.tstart = (unsigned int*)(ws + o_ts) + (size_t)(Kmax + 1) * k;
                ix.wstart = (unsigned int*)(ws + o_ws) + (size_t)(Kmax + 1) * k;
                ix.n_slabs = n_slabs;
                ix.list = (unsigned int*)(ws + o_list) + (size_t)capmax * k;
                ix.slabs = (double*)(ws + o_slabs) + (size_t)OT_TILE_PX * 4 * n_slabs * k;
            }
            err = hipMemsetAsync(ws + o_tn, 0, sizeof(unsigned int) * Kmax * (size_t)n_reqs, st);
            if (err == hipSuccess) {
                hipLaunchKernelGGL(put_kernel<FuseIndexAll>, dim3(1), dim3(64), 0, st, ixs, (FuseIndexAll*)(ws + o_ixs));
                const FuseIndex* dix = (const FuseIndex*)(ws + o_ixs);
                const unsigned gc = (capmax + 1024 * OT_FUSE_IDX_PER - 1) / (1024 * OT_FUSE_IDX_PER);
                const unsigned nd = (unsigned)n_reqs;
                hipLaunchKernelGGL(fuse_chunk_hist_multi_kernel, dim3(gc, 1, nd), dim3(1024), 0, st, dd, dix);
                hipLaunchKernelGGL(fuse_chunk_scan_multi_kernel, dim3(1, 1, nd), dim3(1024), 0, st, dd, dix);
                hipLaunchKernelGGL(fuse_chunk_place_multi_kernel, dim3(gc, 1, nd), dim3(1024), 0, st, dd, dix);
                hipLaunchKernelGGL(fuse_accum_multi_kernel, dim3(std::min<unsigned>(n_slabs, (unsigned)cus), 1, nd), dim3(1024), lds_accum, st, dd, dix, table);
                hipLaunchKernelGGL(fuse_reduce_multi_kernel, dim3(OT_TILE_PX / 256, (unsigned)Kmax, nd), dim3(256), 0, st, dd, dix);
                err = hipGetLastError();
            }
        }
    }
    HIP_TRY(err);
    return OT_OK;
}

// ---- detector image with an automatic extent in one pass (ot_detector_fused.hpp, last section) ---------------------
// Scratch layout of OT_WS_AUTO: the head (detector record, flags, extent slots) is shared by the sample pass and the image.
struct AutoHead {
    size_t o_dets, o_flags, o_slots, end;
    AutoHead() {
        o_dets = 0;
        o_flags = align_up(sizeof(FuseOne));
        o_slots = o_flags + 256;
        end = o_slots + align_up(sizeof(unsigned long long) * 4 * OT_EXT_SLOTS);
    }
};

struct ot_auto_image {
    FuseOne f;  // host copy; the image grid (a, hist) is filled in by finish
    ot_scratch::Lease lease;  // the records: leased until finish / cancel (neither reused nor trimmed in between)
    char* ws;
    size_t o_tn, o_ts, o_list, o_ws, o_slabs;
    unsigned n_slabs;
    hipStream_t st;
    int dev;
};

static int auto_detector(const char* who, const ot_rays* rays, int64_t first, int64_t count, const ot_surface* detector,
                         int32_t projection) {
    if (!rays || !detector) return fail(OT_ERR_INVALID, std::string(who) + ": null argument");
    if (!rays->p || !rays->w || !rays->wl) return fail(OT_ERR_INVALID, std::string(who) + ": ray storage has null buffers");
    if (first < 0 || count < 1 || first + count > rays->N) return fail(OT_ERR_INVALID, std::string(who) + ": range outside the storage");
    if (count >= (1ll << 31)) return fail(OT_ERR_UNSUPPORTED, std::string(who) + ": at most 2^31 - 1 rays per call");
    const bool closed = detector->kind <= OT_SURF_CONIC || detector->z_min == detector->z_max;
    const bool plain = projection == OT_PROJ_NONE || projection == OT_PROJ_ORTHOGRAPHIC;
    if (!closed || !plain)
        return fail(OT_ERR_UNSUPPORTED, std::string(who) + ": detectors with a numeric hit search or a sphere projection take ot_detector_hits_multi");
    return OT_OK;
}

static void auto_fill_detector(FuseOne& f, const LeafSurface& ls, const ot_surface* detector, int32_t projection) {
    std::memset(&f, 0, sizeof(f));
    f.det = ls.d;
    f.Rcurv = detector->R;
    f.projection = projection;
}

extern "C" int ot_detector_extent_sample(const ot_rays* rays, int64_t first, int64_t count, const ot_surface* detector,
                                         int32_t projection, int32_t stride, double* extent4, void* stream) {
    if (int rc = auto_detector("ot_detector_extent_sample", rays, first, count, detector, projection)) return rc;
    if (!extent4 || stride < 1) return fail(OT_ERR_INVALID, "ot_detector_extent_sample: bad argument");
    if (int rc = require_device()) return rc;
    hipStream_t st = (hipStream_t)stream;
    LeafSurface ls;
    if (int rc = ls.init(detector, st)) return rc;
    const AutoHead h;
    const ot_scratch::Lease lease = workspace(OT_WS_AUTO, h.end, st);
    char* ws = lease.p();
    if (!ws) return fail(OT_ERR_HIP, "ot_detector_extent_sample: no scratch memory");
    FuseOne f;
    auto_fill_detector(f, ls, detector, projection);
    f.g.ext_slots = (unsigned long long*)(ws + h.o_slots);
    hipLaunchKernelGGL(put_kernel<FuseOne>, dim3(1), dim3(64), 0, st, f, (FuseOne*)(ws + h.o_dets));
    hipLaunchKernelGGL(extent_init_kernel, dim3(1), dim3(4 * OT_EXT_SLOTS), 0, st, f.g.ext_slots);
    ot_rays part = *rays;
    part.p += first;
    part.w += first;
    part.wl += first;
    const int64_t waves = (count + 64ll * stride - 1) / (64ll * stride);
    hipLaunchKernelGGL(spec_sample_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, part, (uint32_t)count,
                       (const FuseOne*)(ws + h.o_dets), (uint32_t)stride);
    hipLaunchKernelGGL(spec_result_kernel, dim3(1), dim3(64), 0, st, (const unsigned long long*)f.g.ext_slots,
                       (const unsigned int*)nullptr, 0u, extent4);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));  // the caller reads extent4 next (and the detector's tables may go)
    return OT_OK;
}

extern "C" int ot_detector_image_auto_begin(const ot_rays* rays, int64_t first, int64_t count, const ot_surface* detector,
                                            int32_t projection, const double origin[2], const double tile[2],
                                            const int32_t tiles[2], double* result6, ot_auto_image** out, void* stream) {
    if (out) *out = nullptr;
    if (int rc = auto_detector("ot_detector_image_auto_begin", rays, first, count, detector, projection)) return rc;
    if (!origin || !tile || !tiles || !result6 || !out) return fail(OT_ERR_INVALID, "ot_detector_image_auto_begin: null argument");
    if (!(tile[0] > 0.0) || !(tile[1] > 0.0) || tiles[0] < 1 || tiles[1] < 1 || !std::isfinite(origin[0]) || !std::isfinite(origin[1]))
        return fail(OT_ERR_INVALID, "ot_detector_image_auto_begin: bad tile grid");
    const int64_t K = (int64_t)tiles[0] * tiles[1];
    if (K > OT_TILE_MAX || K > OT_FUSE_LDS_ENTRIES) return fail(OT_ERR_UNSUPPORTED, "ot_detector_image_auto_begin: more than 2048 tiles");
    if (int rc = require_device()) return rc;
    hipStream_t st = (hipStream_t)stream;
    LeafSurface ls;
    if (int rc = ls.init(detector, st)) return rc;
    const int cus = cu_count();
    const bool small_k = K <= 1024;  // two rays per thread and sub-block (10-bit tile numbers), as in ot_detector_images
    const bool linebuf = fuse_use_linebuf((int)K);  // few tiles: line buffers, one 1024-thread workgroup per CU
    const int64_t brt = linebuf ? OT_LB_BR * OT_LB_RPT : OT_FUSE_BR * (small_k ? 2 : 1);
    const unsigned n_wg = (unsigned)std::min<int64_t>((linebuf ? 1 : OT_FUSE_WG_PER_CU) * (int64_t)cus, (count + brt - 1) / brt);
    const int64_t piece = ((count + n_wg - 1) / n_wg + brt - 1) / brt * brt;

    std::unique_ptr<ot_auto_image> im(new ot_auto_image);
    FuseOne& f = im->f;
    auto_fill_detector(f, ls, detector, projection);
    f.tx = tiles[0];
    f.K = (int32_t)K;
    f.tiles_ok = 1;
    f.koff = 0;
    f.per_wg = (uint32_t)((piece + OT_FUSE_CH - 1) / OT_FUSE_CH + f.K + 2);
    f.cap = (uint32_t)std::min<int64_t>((int64_t)f.per_wg * n_wg, 0xffffffffll / OT_FUSE_CH - 1);  // record numbers: 32 bits
    f.g.X0 = origin[0];
    f.g.Y0 = origin[1];
    f.g.tw = tile[0];
    f.g.th = tile[1];
    f.g.itw = 1.0 / tile[0];
    f.g.ith = 1.0 / tile[1];
    f.g.tx = tiles[0];
    f.g.ty = tiles[1];
    f.g.esc_cap = (unsigned int)std::max<int64_t>(1ll << 18, count / 64);

    const AutoHead h;
    size_t off = h.end;
    auto carve = [&](size_t bytes) {
        size_t o = off;
        off = (off + bytes + 255) / 256 * 256;
        return o;
    };
    const size_t o_ctile = carve(sizeof(uint32_t) * f.cap);
    const size_t o_cfill = carve(sizeof(uint32_t) * f.cap);
    const size_t o_rec = carve(sizeof(SpecRec) * (size_t)f.cap * OT_FUSE_CH);
    const size_t o_esc = carve(sizeof(SpecRec) * (size_t)f.g.esc_cap);
    im->o_tn = carve(sizeof(unsigned int) * f.K);
    im->o_ts = carve(sizeof(unsigned int) * (f.K + 1));
    im->o_list = carve(sizeof(unsigned int) * f.cap);
    im->o_ws = carve(sizeof(unsigned int) * (f.K + 1));
    im->n_slabs = (unsigned)((f.cap + OT_FUSE_CPW - 1) / OT_FUSE_CPW) + (unsigned)f.K;
    im->o_slabs = carve(sizeof(double) * OT_TILE_PX * 4 * (size_t)im->n_slabs);
    im->lease = workspace(OT_WS_AUTO, off, st);
    char* ws = im->lease.p();
    if (!ws) return fail(OT_ERR_UNSUPPORTED, "ot_detector_image_auto_begin: no memory for the records (take the hit-list path)");
    im->ws = ws;
    im->st = st;
    (void)hipGetDevice(&im->dev);
    if (int rc = fuse_tiles_allow_lds(im->dev)) return rc;
    int* flags = (int*)(ws + h.o_flags);
    f.spread = flags;
    f.overflow = flags + 2;
    f.g.esc_n = (unsigned int*)(flags + 3);
    f.g.ext_slots = (unsigned long long*)(ws + h.o_slots);
    f.chunk_tile = (uint32_t*)(ws + o_ctile);
    f.chunk_fill = (uint32_t*)(ws + o_cfill);
    f.rec = (TileRec*)(ws + o_rec);
    f.g.esc = (SpecRec*)(ws + o_esc);
    hipLaunchKernelGGL(put_kernel<int4>, dim3(1), dim3(64), 0, st, make_int4(1, 0, 0, 0), (int4*)flags);  // spread = 1: tiles always
    hipLaunchKernelGGL(put_kernel<FuseOne>, dim3(1), dim3(64), 0, st, f, (FuseOne*)(ws + h.o_dets));
    hipLaunchKernelGGL(extent_init_kernel, dim3(1), dim3(4 * OT_EXT_SLOTS), 0, st, f.g.ext_slots);
    ot_rays part = *rays;  // the tile kernel addresses its rays with 32 bits from the start of the range
    part.p += first;
    part.w += first;
    part.wl += first;
    const FuseOne* dd = (const FuseOne*)(ws + h.o_dets);
    const size_t lds_tiles = fuse_tiles_lds(f.K, 1, small_k ? 2 : 1, true);
    if (linebuf)
        hipLaunchKernelGGL(fuse_tiles_lb_kernel<true>, dim3(n_wg), dim3(OT_LB_BR), fuse_lb_lds(f.K), st, part, (uint32_t)count, dd, f.K,
                           (uint32_t)piece);
    else if (small_k)
        hipLaunchKernelGGL((fuse_tiles_kernel<false, 1, 2, true>), dim3(n_wg), dim3(OT_FUSE_BR), lds_tiles, st, part, (uint32_t)count,
                           dd, 1, f.K, (uint32_t)piece);
    else
        hipLaunchKernelGGL((fuse_tiles_kernel<false, 1, 1, true>), dim3(n_wg), dim3(OT_FUSE_BR), lds_tiles, st, part, (uint32_t)count,
                           dd, 1, f.K, (uint32_t)piece);
    hipLaunchKernelGGL(spec_result_kernel, dim3(1), dim3(64), 0, st, (const unsigned long long*)f.g.ext_slots,
                       (const unsigned int*)f.g.esc_n, f.g.esc_cap, result6);
    HIP_TRY(hipGetLastError());
    // the caller needs the extent before it can go on: wait here (also: the detector's tables may go)
    HIP_TRY(hipStreamSynchronize(st));
    *out = im.release();
    return OT_OK;
}

extern "C" void ot_detector_image_auto_cancel(ot_auto_image* im) { delete im; }

extern "C" int ot_detector_image_auto_finish(ot_auto_image* im_raw, const double extent[4], int32_t Nx, int32_t Ny,
                                             double* hist, void* stream) {
    std::unique_ptr<ot_auto_image> im(im_raw);  // freed whatever happens
    if (!im || !extent || !hist || Nx < 1 || Ny < 1) return fail(OT_ERR_INVALID, "ot_detector_image_auto_finish: bad argument");
    if (!(extent[1] > extent[0]) || !(extent[3] > extent[2])) return fail(OT_ERR_INVALID, "ot_detector_image_auto_finish: empty image extent");
    if ((int64_t)Nx * Ny > (1ll << 27)) return fail(OT_ERR_INVALID, "ot_detector_image_auto_finish: image too large");
    hipStream_t st = (hipStream_t)stream;
    if (st != im->st) return fail(OT_ERR_INVALID, "ot_detector_image_auto_finish: not the stream of ot_detector_image_auto_begin");
    // (the scratch block of begin is still ours: the handle holds its lease)
    const double* table = observer_table_device();
    if (!table) return fail(OT_ERR_HIP, "could not upload the CIE observer table");
    FuseOne& f = im->f;
    f.a.x0 = extent[0];
    f.a.x1 = extent[1];
    f.a.y0 = extent[2];
    f.a.y1 = extent[3];
    f.a.fx = (double)Nx / (extent[1] - extent[0]);  // Nx / s[0]  misc.py:75
    f.a.fy = (double)Ny / (extent[3] - extent[2]);
    f.a.Nx = Nx;
    f.a.Ny = Ny;
    f.a.ws = 1.0;
    f.hist = hist;
    char* ws = im->ws;
    FuseIndex ix;
    ix.tile_n = (unsigned int*)(ws + im->o_tn);
    ix.tstart = (unsigned int*)(ws + im->o_ts);
    ix.wstart = (unsigned int*)(ws + im->o_ws);
    ix.n_slabs = im->n_slabs;
    ix.list = (unsigned int*)(ws + im->o_list);
    ix.slabs = (double*)(ws + im->o_slabs);
    const int lds_accum = (OT_TILE_PX * 4 + OT_OBS_N * 6) * (int)sizeof(double);  // tile + (value, difference) observer table
    static thread_local bool lds_set[64] = {false};
    if (im->dev >= 0 && im->dev < 64 && !lds_set[im->dev]) {
        HIP_TRY(hipFuncSetAttribute((const void*)spec_accum_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_accum));
        lds_set[im->dev] = true;
    }
    HIP_TRY(hipMemsetAsync(ix.tile_n, 0, sizeof(unsigned int) * f.K, st));
    const unsigned gc = (f.cap + 1024 * OT_FUSE_IDX_PER - 1) / (1024 * OT_FUSE_IDX_PER);
    hipLaunchKernelGGL(fuse_chunk_hist_kernel, dim3(gc), dim3(1024), 0, st, f, ix);
    hipLaunchKernelGGL(fuse_chunk_scan_kernel, dim3(1), dim3(1024), 0, st, f, ix);
    hipLaunchKernelGGL(fuse_chunk_place_kernel, dim3(gc), dim3(1024), 0, st, f, ix);
    hipLaunchKernelGGL(spec_accum_kernel, dim3(std::min<unsigned>(ix.n_slabs, (unsigned)cu_count())), dim3(1024), lds_accum, st, f, ix, table);
    hipLaunchKernelGGL(spec_reduce_kernel, dim3(OT_TILE_PX / 256, (unsigned)f.K), dim3(256), 0, st, f, ix);
    hipLaunchKernelGGL(spec_escaped_kernel, dim3(64), dim3(256), 0, st, f, table);
    HIP_TRY(hipGetLastError());
    return OT_OK;
}

// ---- image conversion ----------------------------------------------------------------------------------------
extern "C" int ot_image_convert(const double* hist, int32_t Nx, int32_t Ny, int32_t fact, int32_t mode, double apx,
                                double K, double L_th, double chroma_scale, double* out, double* workspace, void* stream) {
    if (!hist || !out || !workspace || Nx < 1 || Ny < 1 || fact < 1 || Nx % fact || Ny % fact)
        return fail(OT_ERR_INVALID, "ot_image_convert: bad argument");
    const int flags = mode & (OT_IMG_FLAG_NO_NORMALIZE | OT_IMG_FLAG_NO_CLIP);
    mode &= ~(OT_IMG_FLAG_NO_NORMALIZE | OT_IMG_FLAG_NO_CLIP);
    if (mode < OT_IMG_IRRADIANCE || mode > OT_IMG_SATURATION) return fail(OT_ERR_INVALID, "ot_image_convert: unknown mode");
    if (int rc = require_device()) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int64_t npx = (int64_t)(Nx / fact) * (Ny / fact);
    double* img = workspace;            // (ny, nx, 4) down-binned working copy
    double* red = workspace + npx * 4;  // OT_RED_N reduction slots
    dim3 grid = grid_for(npx), block(256);
    hipLaunchKernelGGL(img_downbin_kernel, grid, block, 0, st, hist, Nx, Ny, fact, img);
    const double inf = INFINITY;
    double init[OT_RED_N] = {-inf, -inf, 0.0, -inf, 0.0, inf, -inf, 0.0};
    HIP_TRY(hipMemcpyAsync(red, init, sizeof(init), hipMemcpyHostToDevice, st));
    if (mode != OT_IMG_IRRADIANCE && mode != OT_IMG_ILLUMINANCE)
        hipLaunchKernelGGL(img_reduce1_kernel, grid, block, 0, st, img, npx, red);
    if (mode == OT_IMG_SRGB_ABSOLUTE || mode == OT_IMG_SRGB_PERCEPTUAL) {
        double h[OT_RED_N];
        HIP_TRY(hipMemcpyAsync(h, red, sizeof(h), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        const bool any_inv = h[OT_RED_ANY_INV] != 0.0;
        const bool cs_given = !std::isnan(chroma_scale);
        int intent = 0;  // srgb.py:318-319: nothing out of gamut and no fixed chroma scale -> plain conversion
        int use_ones = 0;
        double cs = 1.0;
        if (any_inv || cs_given) {
            if (mode == OT_IMG_SRGB_ABSOLUTE) {
                intent = 1;
            } else {
                intent = 2;
                hipLaunchKernelGGL(img_reduce2_kernel, grid, block, 0, st, img, npx, L_th, red);
                HIP_TRY(hipMemcpyAsync(h, red, sizeof(h), hipMemcpyDeviceToHost, st));
                HIP_TRY(hipStreamSynchronize(st));
                use_ones = h[OT_RED_ANY_GAMUT] == 0.0;
                double crmin = (use_ones || !std::isfinite(h[OT_RED_CRMIN])) ? 1.0 : h[OT_RED_CRMIN];
                double f = std::sqrt(crmin);
                f = f < 0.32 ? 0.32 : (f > 1.0 ? 1.0 : f);  // srgb.py:252
                cs = cs_given ? chroma_scale : f;
            }
        }
        hipLaunchKernelGGL(img_correct_kernel, grid, block, 0, st, img, npx, intent, cs, use_ones, red);
    }
    hipLaunchKernelGGL(img_final_kernel, grid, block, 0, st, img, npx, mode | flags, apx, K, red, out);
    HIP_TRY(hipGetLastError());
    return OT_OK;
}

extern "C" int ot_image_convolve(const double* in, int32_t Nx, int32_t Ny, const double* psf, int32_t ps, double* out,
                                 void* stream) {
    if (!in || !psf || !out || Nx < 1 || Ny < 1 || ps < 0 || in == out) return fail(OT_ERR_INVALID, "ot_image_convolve: bad argument");
    const size_t lds = sizeof(double) * (size_t)(2 * ps + 1) * (2 * ps + 1);
    if (lds > 150 * 1024) return fail(OT_ERR_UNSUPPORTED, "ot_image_convolve: kernel larger than 137 x 137 taps");
    if (int rc = require_device()) return rc;
    if (lds > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute((const void*)img_convolve_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(img_convolve_kernel, grid_for((int64_t)Nx * Ny), dim3(256), lds, (hipStream_t)stream, in, Nx, Ny, psf, ps, out);
    HIP_TRY(hipGetLastError());
    return OT_OK;
}

// ---- spectrum rendering ---------------------------------------------------------------------------------------
static int cu_count() {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
        return prop.multiProcessorCount;
    return 256;
}

static int spectrum_range(int64_t n, const unsigned int* fill, const float* wl, const float* w, double* range2, int64_t* count,
                          void* stream);
static int spectrum_histogram(int64_t n, const unsigned int* fill, const float* wl, const float* w, const float* edges,
                              int32_t nbins, double* hist, void* stream);

extern "C" int ot_spectrum_range(int64_t n, const float* wl, const float* w, double* range2, int64_t* count, void* stream) {
    return spectrum_range(n, nullptr, wl, w, range2, count, stream);
}
extern "C" int ot_spectrum_histogram(int64_t n, const float* wl, const float* w, const float* edges, int32_t nbins,
                                     double* hist, void* stream) {
    return spectrum_histogram(n, nullptr, wl, w, edges, nbins, hist, stream);
}
extern "C" int ot_spectrum_range_compact(int64_t n, const uint32_t* fill, const float* wl, const float* w, double* range2,
                                         int64_t* count, void* stream) {
    if (!fill) return fail(OT_ERR_INVALID, "ot_spectrum_range_compact: fill counts missing");
    return spectrum_range(n, fill, wl, w, range2, count, stream);
}
extern "C" int ot_spectrum_histogram_compact(int64_t n, const uint32_t* fill, const float* wl, const float* w,
                                             const float* edges, int32_t nbins, double* hist, void* stream) {
    if (!fill) return fail(OT_ERR_INVALID, "ot_spectrum_histogram_compact: fill counts missing");
    return spectrum_histogram(n, fill, wl, w, edges, nbins, hist, stream);
}

static int spectrum_range(int64_t n, const unsigned int* fill, const float* wl, const float* w, double* range2, int64_t* count,
                          void* stream) {
    if (n < 0 || !range2 || !count || (n && (!wl || !w))) return fail(OT_ERR_INVALID, "ot_spectrum_range: bad argument");
    if (int rc = require_device()) return rc;
    hipStream_t st = (hipStream_t)stream;
    const double init[2] = {INFINITY, -INFINITY};
    HIP_TRY(hipMemcpyAsync(range2, init, sizeof(init), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(count, 0, sizeof(int64_t), st));
    if (n == 0) return OT_OK;
    int64_t blocks = (n + 255) / 256;
    const int64_t cap = (int64_t)cu_count() * 8;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(spectrum_stats_kernel, dim3((unsigned)blocks), dim3(256), 0, st, n, wl, w, range2,
                       (unsigned long long*)count, fill);
    HIP_TRY(hipGetLastError());
    return OT_OK;
}

static int spectrum_histogram(int64_t n, const unsigned int* fill, const float* wl, const float* w, const float* edges,
                              int32_t nbins, double* hist, void* stream) {
    if (n < 0 || !edges || !hist || nbins < 1 || (n && (!wl || !w)))
        return fail(OT_ERR_INVALID, "ot_spectrum_histogram: bad argument");
    if (int rc = require_device()) return rc;
    if (n == 0) return OT_OK;
    hipStream_t st = (hipStream_t)stream;
    // sums (f64) + edges (f32) per workgroup; 64 KiB keeps two workgroups of LDS per CU free for other work
    const size_t lds = (size_t)nbins * sizeof(double) + ((size_t)nbins + 2) * sizeof(float);
    const int lds_bins = lds <= 64 * 1024 ? nbins : 0;
    int64_t blocks = (n + 1023) / 1024;
    const int64_t cap = cu_count();
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(spectrum_hist_kernel, dim3((unsigned)blocks), dim3(1024), lds_bins ? lds : 0, st, n, wl, w, edges,
                       nbins, lds_bins, hist, fill);
    HIP_TRY(hipGetLastError());
    return OT_OK;
}

// ---- focus search -------------------------------------------------------------------------------------------
static unsigned stream_blocks(int64_t n, int threads, int per_cu) {
    int64_t blocks = (n + threads - 1) / threads;
    const int64_t cap = (int64_t)cu_count() * per_cu;
    if (blocks > cap) blocks = cap;
    return (unsigned)(blocks < 1 ? 1 : blocks);
}

extern "C" int ot_focus_prepare(const ot_rays* rays, int64_t first, int64_t count, double z, double* pasb, float* w,
                                int64_t* n_use, void* stream) {
    if (!rays || !rays->p || !rays->w || first < 0 || count < 0 || first + count > rays->N || rays->nt < 2 || !n_use ||
        (count && (!pasb || !w)))
        return fail(OT_ERR_INVALID, "ot_focus_prepare: bad argument");
    if (int rc = require_device()) return rc;
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemsetAsync(n_use, 0, sizeof(int64_t), st));
    if (count == 0) return OT_OK;
    hipLaunchKernelGGL(focus_prepare_kernel, grid_for(count), dim3(256), 0, st, *rays, first, count, z, pasb, w,
                       (unsigned long long*)n_use);
    HIP_TRY(hipGetLastError());
    return OT_OK;
}

extern "C" int ot_focus_cost(int64_t count, const double* pasb, const float* w, int32_t mode, const double* z, int32_t nz,
                             int32_t n_px, double* workspace, double* cost, void* stream) {
    if (count < 1 || !pasb || !w || !z || nz < 1 || !workspace || !cost || mode < OT_FOCUS_RMS ||
        mode > OT_FOCUS_CENTER_SHARPNESS || (mode != OT_FOCUS_RMS && n_px < 2))
        return fail(OT_ERR_INVALID, "ot_focus_cost: bad argument");
    if (int rc = require_device()) return rc;
    hipStream_t st = (hipStream_t)stream;
    double* img = workspace + OT_FOCUS_WS;
    const int64_t np2 = (int64_t)n_px * n_px;
    const unsigned gs = stream_blocks(count, 256, 8), gb = stream_blocks(count, 1024, 1);
    for (int i = 0; i < nz; i++) {
        hipLaunchKernelGGL(focus_init_kernel, dim3(1), dim3(64), 0, st, workspace);
        hipLaunchKernelGGL(focus_stats_kernel, dim3(gs), dim3(256), 0, st, count, pasb, w, z[i], workspace);
        if (mode == OT_FOCUS_RMS) {
            hipLaunchKernelGGL(focus_var_kernel, dim3(gs), dim3(256), 0, st, count, pasb, w, z[i], workspace);
        } else {
            HIP_TRY(hipMemsetAsync(img, 0, sizeof(double) * np2, st));
            hipLaunchKernelGGL(focus_bin_kernel, dim3(gb), dim3(1024), 0, st, count, pasb, w, z[i], workspace, n_px, img);
            hipLaunchKernelGGL(focus_image1_kernel, grid_for(np2), dim3(256), 0, st, img, n_px, mode, workspace);
            if (mode == OT_FOCUS_IRR_VAR)
                hipLaunchKernelGGL(focus_image2_kernel, grid_for(np2), dim3(256), 0, st, img, n_px, workspace);
        }
        hipLaunchKernelGGL(focus_finalize_kernel, dim3(1), dim3(64), 0, st, mode, n_px, workspace, cost + i);
    }
    HIP_TRY(hipGetLastError());
    return OT_OK;
}

extern "C" int ot_focus_moments(int64_t count, const double* pasb, const float* w, double b0, double b1, double* sums,
                                void* stream) {
    if (count < 1 || !pasb || !w || !sums || !(b1 > b0)) return fail(OT_ERR_INVALID, "ot_focus_moments: bad argument");
    if (int rc = require_device()) return rc;
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemsetAsync(sums, 0, sizeof(double) * 16, st));
    const unsigned gs = stream_blocks(count, 256, 8);
    hipLaunchKernelGGL(focus_moments1_kernel, dim3(gs), dim3(256), 0, st, count, pasb, w, sums);
    hipLaunchKernelGGL(focus_moments2_kernel, dim3(gs), dim3(256), 0, st, count, pasb, w, b0, b1, sums);
    HIP_TRY(hipGetLastError());
    return OT_OK;
}

// ---------------------------------------------------------------------------------------------------------
// diagnostics: exactness of the division / square-root cores (ot_selftest.hpp)
// ---------------------------------------------------------------------------------------------------------
extern "C" int ot_selftest_arith(int32_t op, int32_t operand_class, int64_t n, uint64_t seed, int64_t* mismatches,
                                 double* first_bad4, void* stream) {
    if (!mismatches || !first_bad4 || n < 0) return fail(OT_ERR_INVALID, "ot_selftest_arith: bad argument");
    if (op < OT_ST_DIV || op > OT_ST_DIV_SHARED || operand_class < OT_CLS_WIDE || operand_class > OT_CLS_COSINE)
        return fail(OT_ERR_INVALID, "ot_selftest_arith: unknown operation or operand class");
    if (int rc = require_device()) return rc;
    hipStream_t st = (hipStream_t)stream;
    char* scratch = nullptr;
    HIP_TRY(hipMalloc((void**)&scratch, 64));
    hipError_t e = hipMemsetAsync(scratch, 0, 64, st);
    unsigned long long host[2] = {0, 0};
    double bad[4] = {0, 0, 0, 0};
    if (e == hipSuccess && n > 0) {
        hipLaunchKernelGGL(selftest_arith_kernel, dim3((unsigned)(cu_count() * 8)), dim3(256), 0, st, (int)op,
                           (int)operand_class, n, seed, (unsigned long long*)scratch, (double*)(scratch + 16));
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(host, scratch, 16, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(bad, scratch + 16, 32, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(scratch);
    HIP_TRY(e);
    *mismatches = (int64_t)host[0];
    std::memcpy(first_bad4, bad, sizeof(bad));
    return OT_OK;
}

extern "C" int ot_selftest_eval(int32_t op, int64_t n, const double* a, const double* b, const double* c, double* core_out,
                                double* ieee_out, void* stream) {
    if (n < 0 || (n && (!a || !core_out || !ieee_out))) return fail(OT_ERR_INVALID, "ot_selftest_eval: bad argument");
    if (op < OT_ST_DIV || op > OT_ST_DIV_SHARED) return fail(OT_ERR_INVALID, "ot_selftest_eval: unknown operation");
    if (int rc = require_device()) return rc;
    if (n == 0) return OT_OK;
    hipLaunchKernelGGL(selftest_eval_kernel, grid_for(n), dim3(256), 0, (hipStream_t)stream, (int)op, n, a, b, c, core_out,
                       ieee_out);
    HIP_TRY(hipGetLastError());
    return OT_OK;
}
