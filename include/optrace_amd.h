/*
 * optrace_amd.h -- C-ABI of the MI355X-native sequential ray-tracing core.
 *
 * The upstream reference (drocheam/optrace, pure NumPy) has no FFI boundary of its own
 * (SURVEY.md section 8b); the drop-in boundary is the Python surface of
 * optrace/tracer/__init__.py:3-63.  This header is the C-ABI that our Python mirror of that
 * surface (package optrace_amd) binds with ctypes.  Every entry point names the reference
 * function(s) it replaces as file:line relative to the reference checkout.
 *
 * Conventions
 *   - All ray buffers are DEVICE pointers (HIP), caller-owned, struct-of-arrays.  Multi-component
 *     quantities use the reference's Fortran-ordered layout (ray_storage.py:80-90): element
 *     (ray r, section i, component c) of an (N, nt, 3) array lives at  r + N*(i + nt*c).
 *   - dtypes follow ray_storage.py:77-90: positions/directions/refractive indices f64,
 *     weights / wavelengths / polarisation f32.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls are asynchronous
 *     with respect to the host unless stated otherwise.
 *   - Return value: 0 on success, negative OT_ERR_* otherwise; ot_last_error() gives the text.
 *   - Nothing here allocates or frees caller buffers.  Handles (ot_scene) own small device tables only.
 */
#ifndef OPTRACE_AMD_H
#define OPTRACE_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OT_ABI_VERSION 9

/* ---- status codes -------------------------------------------------------------------------- */
#define OT_OK 0
#define OT_ERR_INVALID (-1)     /* bad argument / descriptor                                     */
#define OT_ERR_HIP (-2)         /* HIP runtime error (text in ot_last_error)                     */
#define OT_ERR_UNSUPPORTED (-3) /* feature not representable on the device                       */
#define OT_ERR_NO_DEVICE (-4)   /* no HIP device available                                       */

/* ---- numeric contract (reference class constants) ------------------------------------------ */
#define OT_C_EPS 1e-6        /* Surface.C_EPS   surface.py:17        */
#define OT_N_EPS_SURF 1e-10  /* Surface.N_EPS   surface.py:20        */
#define OT_N_EPS_TRACE 1e-11 /* Raytracer.N_EPS raytracer.py:30      */
#define OT_MAX_HIT_ITER 200  /* regula-falsi timeout surface.py:403  */

/* counter rows, Raytracer.INFOS raytracer.py:43-48 */
#define OT_INFO_ABSORB_MISSING 0
#define OT_INFO_TIR 1
#define OT_INFO_ILL_COND 2
#define OT_INFO_OUTLINE_INTERSECTION 3
#define OT_INFO_HURB_NEG_DIR 4
#define OT_N_INFOS 5

/* ---- surfaces (optrace/tracer/geometry/surface/) -------------------------------------------- */
#define OT_SURF_CIRCLE 0  /* CircularSurface    circular_surface.py:9     flat disc               */
#define OT_SURF_RING 1    /* RingSurface        ring_surface.py:10        flat annulus            */
#define OT_SURF_RECT 2    /* RectangularSurface rectangular_surface.py:10 flat, rotated by angle  */
#define OT_SURF_SLIT 3    /* SlitSurface        slit_surface.py:12        rect minus inner rect   */
#define OT_SURF_CONIC 4   /* ConicSurface / SphericalSurface (k = 0) conic_surface.py:10          */
#define OT_SURF_ASPHERE 5 /* AsphericSurface    aspheric_surface.py:9     conic + even polynomial */
#define OT_SURF_TILTED 6  /* TiltedSurface      tilted_surface.py:10      tilted plane inside a disc */
#define OT_SURF_DATA1D 7  /* DataSurface1D / FunctionSurface1D  data_surface_1d.py:6, function_surface_1d.py:8:
                             radial profile z(r) as a FITPACK B-spline (knots + coefficients in `tab`)      */
#define OT_SURF_DATA2D 8  /* DataSurface2D / FunctionSurface2D  data_surface_2d.py:10, function_surface_2d.py:12:
                             z(x, y) as a tensor-product FITPACK B-spline                                   */

#define OT_MAX_ASPH 12 /* even-order coefficients a2 .. a24 */
#define OT_MAX_LINES 8 /* discrete wavelengths tabulated per scene */

typedef struct ot_surface {
    int32_t kind;   /* OT_SURF_*                                                                 */
    int32_t ncoeff; /* ASPHERE: number of even polynomial coefficients                           */
    double pos[3];  /* centre position                                                           */
    double r;       /* outer radius (CIRCLE/RING/CONIC/ASPHERE); RECT/SLIT keep the reference's 1 */
    double ri;      /* RING: inner radius                                                        */
    double dim[2];  /* RECT/SLIT: outer side lengths                                             */
    double dimi[2]; /* SLIT: inner (open) side lengths                                           */
    double angle;   /* RECT/SLIT: rotation about z [rad] (RectangularSurface._angle)             */
    double R;       /* CONIC/ASPHERE: vertex radius of curvature                                 */
    double k;       /* CONIC/ASPHERE: conic constant                                             */
    double z_min;   /* absolute z range of the surface (Surface.z_min / z_max)                   */
    double z_max;
    double coeff[OT_MAX_ASPH]; /* ASPHERE: a2, a4, ... [mm^-1, mm^-3, ...]                         */
    double normal[3]; /* TILTED: unit normal with normal[2] > 0 (TiltedSurface.normal)            */
    double sign;      /* DATA1D/2D: +1, or -1 after flip() (DataSurface2D._sign)                   */
    double offset;    /* DATA1D/2D: spline value at the centre, removed from every value (._offset) */
    /* DATA1D/2D: spline tables (host pointer, caller-owned, copied by the callee).  k = OT_SPL_K = 4.
     *   DATA1D: t[nknots] | c[nknots] | dc[nknots]: knots, B-spline coefficients (padded with zeros to nknots,
     *           as FITPACK/SciPy store them) and the coefficients of the first derivative (order k-1 on the
     *           knots t[1..nknots-2]).  scipy InterpolatedUnivariateSpline(k=4) of the mirrored profile.
     *   DATA2D: t[nknots] (same knots in x and y) | c[(nknots-5)^2] | cx[(nknots-6)(nknots-5)] |
     *           cy[(nknots-5)(nknots-6)]: RectBivariateSpline(kx=ky=4) coefficients, row-major with y
     *           fastest (FITPACK order), and those of d/dx and d/dy.  `angle` = DataSurface2D._angle.     */
    const double* tab;
    int64_t tab_len;
    int32_t nknots;
    int32_t flags;    /* OT_SURF_FLAG_* */
} ot_surface;
/* FunctionSurface2D.normals with a user deriv_func evaluates it at the UNROTATED relative coordinates and only
 * rotates the resulting gradient (function_surface_2d.py:229-244); without deriv_func normals come from central
 * differences of the (rotated) surface.  The flag selects the first behaviour so results match the reference. */
#define OT_SURF_FLAG_DERIV_UNROTATED 1
/* FunctionSurface mask_func (function_surface_2d.py:158-191): a Python callable cannot run per ray, so the mask travels
 * as a bitmap sampled at cell centres in the function's own frame (before rotate() and flip(), like the spline).
 * With this flag `tab` continues behind the spline tables with
 *     n (as a double) | ceil(cells / 64) doubles whose bytes are little-endian uint32 words, bit (i & 31) of word i >> 5
 * DATA1D: cells = n over the radius [0, r], cell i covers [i, i + 1) * r / n;
 * DATA2D: cells = n * n over [-r, r]^2, index iy * n + ix, cell (ix, iy) covers [-r + ix h, -r + (ix + 1) h) with
 *         h = 2 r / n in x and likewise in y.
 * A position is on the surface if it is inside r (+ N_EPS) and its cell's bit is set; positions closer to the mask's
 * edge than one cell can therefore differ from the callable.  tab_len counts these doubles too. */
#define OT_SURF_FLAG_MASK_TABLE 2
#define OT_SPL_K 4

/* ---- media: RefractionIndex.__call__ refraction_index.py:62-169 ----------------------------- */
#define OT_N_CONSTANT 0    /* c[0] = n                                                            */
#define OT_N_ABBE 1        /* c = {A, B, d}: n = A + B/(wl2 - d)  (A, B solved on the host exactly
                              as refraction_index.py:85-100 does, including its float32 line table) */
#define OT_N_CAUCHY 2      /* :105 */
#define OT_N_CONRADY 3     /* :101 */
#define OT_N_SELLMEIER1 4  /* :108 */
#define OT_N_SELLMEIER2 5  /* :111 */
#define OT_N_SELLMEIER3 6  /* :114 */
#define OT_N_SELLMEIER4 7  /* :117 */
#define OT_N_SELLMEIER5 8  /* :120 */
#define OT_N_SCHOTT 9      /* :124 */
#define OT_N_HERZBERGER 10 /* :127 */
#define OT_N_HOO1 11       /* "Handbook of Optics 1" :131 */
#define OT_N_HOO2 12       /* "Handbook of Optics 2" :134 */
#define OT_N_EXTENDED 13   /* :137 */
#define OT_N_EXTENDED2 14  /* :141 */
#define OT_N_EXTENDED3 15  /* :145 */
#define OT_N_DATA 16       /* np.interp over an equally spaced table (spectrum.py:103-106), table
                              = tab_len (wl, n) pairs at table_pool[tab_off ...]: first the tab_len
                              wavelengths, then the tab_len values.  Also carries host-tabulated
                              "Function" media.                                                     */
#define OT_N_LINES 17      /* exact per-line values for discrete spectra: n = value[j] where
                              wl == line[j] (same table layout as DATA); used for "Function" media
                              under Monochromatic / Lines sources                                   */

typedef struct ot_medium {
    int32_t model; /* OT_N_* */
    int32_t tab_len;
    int64_t tab_off;
    double c[10];
} ot_medium;

/* ---- filters: TransmissionSpectrum.__call__ transmission_spectrum.py:73-84 -------------------- */
#define OT_T_CONSTANT 0  /* spectrum.py:99   */
#define OT_T_DATA 1      /* spectrum.py:103  np.interp, 0 outside                                  */
#define OT_T_RECTANGLE 2 /* spectrum.py:108  */
#define OT_T_GAUSSIAN 3  /* spectrum.py:113  evaluated in float32 like the reference does for f32 wl */
#define OT_T_LINES 4     /* host-tabulated per-line values ("Function" spectra, discrete sources)   */

typedef struct ot_filter {
    int32_t type;    /* OT_T_* */
    int32_t inverse; /* TransmissionSpectrum.inverse: T -> 1 - T */
    int32_t tab_len;
    int32_t _pad;
    int64_t tab_off;
    double val, wl0, wl1, mu, sig;
} ot_filter;

/* ---- tracing elements: Raytracer.__tracing_elements raytracer.py:492-508 ----------------------- */
#define OT_EL_LENS 0       /* Lens        raytracer.py:314-370 (two refracting surfaces)            */
#define OT_EL_IDEAL_LENS 1 /* IdealLens   raytracer.py:360-363, 720-759                             */
#define OT_EL_FILTER 2     /* Filter      raytracer.py:372-391                                      */
#define OT_EL_APERTURE 3   /* Aperture    raytracer.py:372-391 (the invisible end aperture as well) */

typedef struct ot_element {
    int32_t kind;    /* OT_EL_*                                                                    */
    int32_t front;   /* surface index                                                              */
    int32_t back;    /* surface index (LENS) or -1                                                 */
    int32_t n_lens;  /* LENS: medium index of the lens material                                    */
    int32_t n_after; /* LENS / IDEAL_LENS: medium index behind the lens (n2 or n0, raytracer.py:326) */
    int32_t filter;  /* FILTER: filter index                                                       */
    int32_t hurb;    /* APERTURE: 1 = apply HURB edge bending here (raytracer.py:385)              */
    int32_t _pad;
    double D;        /* IDEAL_LENS: optical power [dpt]                                            */
} ot_element;

typedef struct ot_scene_desc {
    double outline[6]; /* Raytracer.outline [x0,x1,y0,y1,z0,z1]                                      */
    int32_t n_surfaces, n_elements, n_media, n_filters;
    const ot_surface* surfaces;
    const ot_element* elements; /* z-sorted; the last one must be the end aperture (raytracer.py:501) */
    const ot_medium* media;
    const ot_filter* filters;
    const double* table_pool; /* shared table storage for DATA/LINES media and filters               */
    int64_t table_pool_len;
    int32_t n0;       /* ambient medium index (Raytracer.n0)                                         */
    int32_t no_pol;   /* Raytracer.no_pol                                                            */
    int32_t use_hurb; /* Raytracer.use_hurb                                                          */
    int32_t n_lines;  /* > 0: every source has a discrete spectrum; `lines` = the distinct wavelengths       */
    double hurb_factor; /* Raytracer.HURB_FACTOR raytracer.py:33                                     */
    const double* lines; /* n_lines float32 wavelength values (as doubles) or NULL.  With 1..OT_MAX_LINES lines
                            ot_generate_and_trace tabulates n(lambda), n1/n2 and filter transmissions per line
                            once on the host and the kernel reads them from LDS instead of evaluating them per ray */
} ot_scene_desc;

typedef struct ot_scene ot_scene; /* opaque: device copy of the tables */

/* ---- ray sources: RaySource.create_rays ray_source.py:204-437 --------------------------------- */
#define OT_SRC_POINT 0  /* Point  point.py:62                                                       */
#define OT_SRC_LINE 1   /* Line   line.py:81                                                        */
#define OT_SRC_CIRCLE 2 /* CircularSurface.random_positions circular_surface.py:33                  */
#define OT_SRC_RING 3   /* RingSurface.random_positions     ring_surface.py:135                     */
#define OT_SRC_RECT 4   /* RectangularSurface.random_positions rectangular_surface.py:142           */
#define OT_SRC_IMAGE_RGB 5  /* RGBImage: pixel pdf + sRGB primaries ray_source.py:233-258           */
#define OT_SRC_IMAGE_GRAY 6 /* GrayscaleImage: pixel pdf, spectrum from `spectrum`                  */

#define OT_DIV_NONE 0
#define OT_DIV_LAMBERTIAN 1 /* ray_source.py:300-309 */
#define OT_DIV_ISOTROPIC 2  /* ray_source.py:311-318 */
#define OT_DIV_TABLE 3      /* "Function": host-tabulated pdf(theta), ray_source.py:320-333           */

#define OT_OR_CONSTANT 0   /* ray_source.py:266 */
#define OT_OR_CONVERGING 1 /* ray_source.py:269 */
#define OT_OR_ARRAY 2      /* "Function" ray_source.py:272-274: base orientations evaluated by the caller  */

#define OT_POL_CONSTANT 0 /* also "x" (0) and "y" (pi/2), ray_source.py:366-376 */
#define OT_POL_UNIFORM 1  /* :378 */
#define OT_POL_LIST 2     /* "List" and "xy": discrete angles with probabilities :372, :381-385       */
#define OT_POL_TABLE 3    /* "Function": host-tabulated pdf(angle) :387-392                          */

#define OT_SPEC_MONO 0     /* light_spectrum.py:91  */
#define OT_SPEC_UNIFORM 1  /* Constant / Rectangle: uniform in [wl0, wl1] :94-98                     */
#define OT_SPEC_LINES 2    /* :100-103 discrete inverse CDF                                          */
#define OT_SPEC_GAUSSIAN 3 /* :110-122 truncated normal via erfinv                                   */
#define OT_SPEC_TABLE 4    /* Data / Blackbody / Function / Histogram: linear inverse CDF :105,:129   */

typedef struct ot_source {
    int32_t shape;       /* OT_SRC_*                                                                */
    int32_t divergence;  /* OT_DIV_*                                                                */
    int32_t div_2d;      /* RaySource.div_2d                                                        */
    int32_t orientation; /* OT_OR_*                                                                 */
    int32_t polarization; /* OT_POL_*                                                               */
    int32_t spectrum;    /* OT_SPEC_*                                                               */
    int32_t img_w, img_h; /* IMAGE_*: pixel counts                                                  */
    double pos[3];
    double r, ri;        /* LINE: half length r; CIRCLE/RING radii                                  */
    double dim[2];       /* RECT / IMAGE side lengths                                               */
    double angle;        /* RECT rotation [rad]; LINE angle [rad]                                   */
    double div_angle;    /* [deg]                                                                   */
    double div_axis_angle; /* [deg]                                                                 */
    double s[3];         /* unit orientation (CONSTANT)                                             */
    double conv_pos[3];  /* CONVERGING                                                              */
    double pol_angle;    /* [rad] for POL_CONSTANT                                                  */
    double wl, wl0, wl1, mu, sig;
    double power;        /* power of the source; a range without ray_power gives each of its rays
                          * power / count (ray_storage.py:160, ray_source.py:220)                   */
    /* tables (device copies made by ot_sources_create); layouts:
     *   spec_tab : LINES: n_spec lines (as the reference's float32 values upcast) then n_spec weights;
     *              TABLE: n_spec wavelengths then n_spec pdf values
     *   pol_tab  : LIST: n_pol angles [rad] then n_pol probabilities; TABLE: n_pol angles then pdf
     *   div_tab  : TABLE: n_div thetas [rad] then n_div pdf values
     *   img_pdf  : IMAGE_*: img_w*img_h relative pixel powers (RaySource._pIf)
     *   img_rgb  : IMAGE_RGB: img_w*img_h*3 sRGB values in [0,1], row-major (y, x, c)              */
    const double* spec_tab; int64_t n_spec;
    const double* pol_tab;  int64_t n_pol;
    const double* div_tab;  int64_t n_div;
    const double* img_pdf;
    const double* img_rgb;
    /* OR_ARRAY: DEVICE pointer (caller-owned, NOT copied; must stay valid while the table is used) to the base
     * orientation of each ray of this source, x[n_or] | y[n_or] | z[n_or]; every range of this source must
     * hold exactly n_or rays, ray j of the range reads element j.  With a NULL pointer the source emits along
     * `s`: start positions do not depend on the orientation, so the caller runs ot_rays_generate once with NULL,
     * evaluates its orientation function at the positions, and generates again with the result.             */
    const double* s_or; int64_t n_or;
} ot_source;

/* A contiguous block of rays generated from one source: rays [first, first+count) of the launch (the blocks of a
 * launch follow each other without gaps from ray 0 on; slots of the storage behind the last block stay unused).  A block
 * is one stratification domain, as one thread's share is in the reference (ray_storage.py:147-166); a source
 * may be cut into several.  Blocks whose count is a power of two are the cheapest to generate (their stratum
 * permutation needs no rejection step). */
typedef struct ot_source_range {
    int32_t source;  /* index into the source table                                                 */
    int32_t _pad;
    int64_t first;   /* first ray index inside this launch                                          */
    int64_t count;   /* number of rays (stratification domain, ray_storage.py:160-163)              */
    double ray_power; /* power of each ray of the block (`power/N` ray_source.py:220, stored f32);
                       * <= 0: the source's `power` / count                                         */
} ot_source_range;

typedef struct ot_sources ot_sources; /* opaque: device copy of sources + tables */

/* ---- ray storage: RayStorage ray_storage.py:35-90 --------------------------------------------- */
/* N is the number of ray slots AND the plane stride: element (ray r, section i, component c) of p lives at
 * r + N * (i + nt * c).  A caller may make N larger than its ray count so that every plane starts on a 128-byte line
 * (a multiple of 32 rays; with planes off the lines the tracing kernel runs 30-50 % slower): the source ranges of
 * ot_generate_and_trace / ot_rays_generate then cover the first rays only and the slots behind are neither generated
 * nor traced; ot_trace (rays handed in) walks all N slots, so the unused ones must hold weight 0 and a finite
 * direction; every detector / spectrum / focus entry point takes its own first / count. */
typedef struct ot_rays {
    int64_t N;   /* ray slots = plane stride (>= rays of the launch, see above)                     */
    int32_t nt;  /* sections per ray = n tracing surfaces + 2 (raytracer.py:278)                   */
    int32_t _pad;
    double* p;   /* (N, nt, 3) f64 F-order  p_list                                                 */
    double* s;   /* (N, 3)     f64 F-order  s0_list: in = initial, out = final direction           */
    float* w;    /* (N, nt)    f32 F-order  w_list                                                 */
    double* n;   /* (N, nt)    f64 F-order  n_list                                                 */
    float* wl;   /* (N,)       f32          wl_list                                                */
    float* pol;  /* (N, nt, 3) f32 F-order  pol_list; NULL when no_pol                             */
} ot_rays;

/* ---- library --------------------------------------------------------------------------------- */
int ot_abi_version(void);
const char* ot_last_error(void);
/* number of HIP devices visible; negative on error */
int ot_device_count(void);

/* ---- scene ------------------------------------------------------------------------------------ */
/* Validates and uploads the scene tables (replaces nothing 1:1 -- the reference walks Python objects,
 * raytracer.py:274, 492-508).  Synchronous. */
int ot_scene_create(const ot_scene_desc* desc, ot_scene** out);
void ot_scene_destroy(ot_scene* scene);
/* nt = number of tracing surfaces + 2 (raytracer.py:278) */
int ot_scene_sections(const ot_scene* scene);

/* ---- sources ---------------------------------------------------------------------------------- */
int ot_sources_create(const ot_source* sources, int32_t n_sources, ot_sources** out);
void ot_sources_destroy(ot_sources* src);

/* RaySource.create_rays (ray_source.py:204) for every range, writing section 0 of `rays`
 * (p[:,0], s, w[:,0], wl, pol[:,0]) the way RayStorage.thread_rays does (ray_storage.py:156-166).
 * Device RNG: counter-based (Philox-4x32-10) keyed by (seed, range, ray); stratified grids use a
 * keyed bijective index permutation instead of the reference's shuffle (random.py:39-43). */
int ot_rays_generate(const ot_sources* src, const ot_source_range* ranges, int32_t n_ranges,
                     uint64_t seed, int32_t no_pol, const ot_rays* rays, void* stream);

/* ---- tracing ---------------------------------------------------------------------------------- */
/* Raytracer.trace / sub_trace (raytracer.py:262-415) for rays whose section 0 is already in `rays`
 * (injected or produced by ot_rays_generate).  Fills sections 1..nt-1 of p/w/n/pol, n[:,0], the
 * final directions in rays->s and ADDS the per-section counters to msgs (device, int64[5*nt + 1]:
 * the counters row-major (info, section), then one word that is set non-zero if a numeric hit search
 * ran into the 200-iteration timeout, where the reference raises TimeoutError surface.py:403).  hurb_normals: optional device array (2*n_hurb_elements*N f64:
 * for the j-th HURB aperture, N standard-normal draws for the a axis then N for the b axis,
 * replacing np.random.normal raytracer.py:468-469); NULL = device RNG keyed by `seed`. */
int ot_trace(const ot_scene* scene, const ot_rays* rays, const double* hurb_normals, uint64_t seed,
             int64_t* msgs, void* stream);

/* Fused generation + trace: same result as ot_rays_generate followed by ot_trace, but the freshly
 * generated ray never makes a round trip through HBM (one launch, ray state in registers). */
int ot_generate_and_trace(const ot_scene* scene, const ot_sources* src,
                          const ot_source_range* ranges, int32_t n_ranges, uint64_t seed,
                          const ot_rays* rays, int64_t* msgs, void* stream);

/* Raytracer.trace (raytracer.py:262-415) as ONE synchronous call: ot_generate_and_trace, then the wait for
 * `stream`; `msgs_host` (HOST memory, int64[5*nt + 1], layout as above) receives the counters of this
 * launch alone (the reference builds a fresh msgs array per trace, raytracer.py:289).  The device writes them
 * into a pinned buffer of the scene, so there is no device-to-host copy and no second synchronisation. */
int ot_generate_and_trace_host(const ot_scene* scene, const ot_sources* src,
                               const ot_source_range* ranges, int32_t n_ranges, uint64_t seed,
                               const ot_rays* rays, int64_t* msgs_host, void* stream);

/* Render-only chunk of Raytracer.iterative_render (raytracer.py:1235-1267: the images of all chunks are summed, only
 * the LAST chunk's rays stay in the tracer): n_rays rays are generated and traced as by ot_generate_and_trace_host,
 * but no section is stored.  Every ray that is still alive behind the last surface leaves its last section -- the
 * positions at sections nt-2 and nt-1, the weight at nt-2, the wavelength: what a detector behind the last surface
 * reads of a ray (raytracer.py:929-985) -- in `tail`, a ray storage with TWO sections (tail->nt == 2; p (N, 2, 3), w
 * (N, 2): section 1 is written as 0, every ray ends absorbed --, wl (N); s, n, pol unused, may be NULL) of
 * N = tail->N >= ot_tail_capacity(n_rays) slots, N a multiple of 65536.  The living rays are gathered wave by wave
 * into 1024 interleaved pieces (fill: device uint32[1024], scratch of the call); result2 (device-visible HOST memory,
 * int64[2]) receives the number of leading slots in use (a multiple of 65536; slots in it without a ray carry weight
 * 0) and the number of living rays.  The detector entry points (ot_detector_images, ot_detector_hits_multi,
 * ot_detector_extent_sample, ot_detector_image_auto_*) take `tail` with count = result2[0] like any other storage;
 * they give the images of the stored path for every detector that lies behind the last tracing surface.  The order
 * of the rays is not the order of generation (no per-source ranges).  msgs_host as for
 * ot_generate_and_trace_host.  Every scene has the form (ot_scene_tail_supported: 1 for a valid handle; kept for callers
 * written against ABI 7, where scenes with a numeric hit search had none). */
int64_t ot_tail_capacity(int64_t n_rays);
int ot_scene_tail_supported(const ot_scene* scene);
int ot_generate_and_trace_tail(const ot_scene* scene, const ot_sources* src, const ot_source_range* ranges,
                               int32_t n_ranges, uint64_t seed, int64_t n_rays, const ot_rays* tail, uint32_t* fill,
                               int64_t* result2, int64_t* msgs_host, void* stream);

/* The living rays of a stored chunk join the tail of the render-only chunk traced before it (ABI 9): `iterative_render`
 * keeps the rays of its LAST chunk (raytracer.py:1235-1267), which therefore goes through the ray storage -- its binning need
 * not be a pass of its own.  For the rays [first, first + count) of `rays` that are alive in their last section
 * (w[nt - 2] > 0), positions nt - 2 and nt - 1, the wavelength and the weight times weight_scale (formed in f64, rounded to
 * f32 once; the ratio chunk rays / tail rays makes the chunk's rays carry the power of the tail's) are appended to `tail`
 * behind what ot_generate_and_trace_tail(n_rays = rays_before) has written there -- same `fill`, untouched in between --,
 * then the tail is sealed again: result2 as above, for both together.  tail->N >= ot_tail_capacity(rays_before + count + 64).
 * Synchronous. */
int ot_tail_append(const ot_rays* rays, int64_t first, int64_t count, double weight_scale, int64_t rays_before,
                   const ot_rays* tail, uint32_t* fill, int64_t* result2, void* stream);

/* Measurement aid (the reference times `RT.trace` with perf_counter, tests/benchmark.py:81-86): with timing
 * on, every tracing launch of this scene records one HIP event right before and one right after the tracing
 * kernel on the launch stream; ot_scene_last_trace_ms waits for the later one and returns the kernel's
 * duration in milliseconds. */
int ot_scene_set_timing(ot_scene* scene, int32_t on);
int ot_scene_last_trace_ms(const ot_scene* scene, double* ms);

/* ---- leaf operators (public Surface / RefractionIndex methods) ---------------------------------- */
/* Surface.find_hit (surface.py:307, conic_surface.py:126): p, s are (n,3) F-order device arrays;
 * outputs p_hit (n,3) F-order, is_hit (n) uint8, ill (n) uint8: bit 0 = ill-conditioned bracket
 * (always 0 for analytic surfaces), bit 1 = the iteration timed out (surface.py:403). */
int ot_surface_find_hit(const ot_surface* surf, int64_t n, const double* p, const double* s,
                        double* p_hit, uint8_t* is_hit, uint8_t* ill, void* stream);
/* Surface.normals (surface.py:247, conic_surface.py:70, function_surface_2d.py:202): out (n,3) F-order */
int ot_surface_normals(const ot_surface* surf, int64_t n, const double* x, const double* y,
                       double* normals, void* stream);
/* Surface.mask (surface.py:235, ring_surface.py:123, rectangular_surface.py:100, slit_surface.py:89) */
int ot_surface_mask(const ot_surface* surf, int64_t n, const double* x, const double* y,
                    uint8_t* mask, void* stream);
/* Surface.values (surface.py:137) */
int ot_surface_values(const ot_surface* surf, int64_t n, const double* x, const double* y,
                      double* z, void* stream);
/* RingSurface.hurb_props / SlitSurface.hurb_props (ring_surface.py:88, slit_surface.py:65):
 * a_, b_ (n), b (n,3) F-order, inside (n) uint8 */
int ot_surface_hurb_props(const ot_surface* surf, int64_t n, const double* x, const double* y,
                          double* a_, double* b_, double* b, uint8_t* inside, void* stream);
/* RefractionIndex.__call__ (refraction_index.py:62): wl f32 device array as stored in RayStorage */
int ot_refraction_index(const ot_medium* medium, const double* table_pool, int64_t table_pool_len,
                        int64_t n, const float* wl, double* out, void* stream);

/* ---- detector --------------------------------------------------------------------------------- */
#define OT_PROJ_NONE 0          /* flat detector or projection_method=None                           */
#define OT_PROJ_EQUIDISTANT 1   /* spherical_surface.py:64                                          */
#define OT_PROJ_ORTHOGRAPHIC 2  /* :50                                                              */
#define OT_PROJ_EQUAL_AREA 3    /* :87                                                              */
#define OT_PROJ_STEREOGRAPHIC 4 /* :75                                                              */

/* Raytracer._hit_detector (raytracer.py:881-1051) over rays [first, first+count): for every ray
 * the section straddling the detector is searched, the detector surface is intersected with the
 * direction re-derived from stored positions (ray_storage.py:274-279), hits beyond the section end
 * are retried on the next section, and the optional sphere projection is applied.
 * Outputs are dense per-ray arrays (count entries): ph (count,3) F-order projected hit, hw f32 weight
 * (0 = no valid hit; the reference drops those rows, raytracer.py:1023), and ill_count (device
 * int64[2], ADDED to: [0] ill-conditioned rays, [1] rays whose numeric hit search timed out).  extent4 (device f64[4], may be NULL): running xmin,xmax,ymin,ymax of valid hits
 * (raytracer.py:1044-1046), must be initialised by the caller to +inf,-inf,+inf,-inf.
 * crop4 (HOST f64[4] xmin,xmax,ymin,ymax, may be NULL): a user extent; hits outside it are dropped like hits
 * without weight (raytracer.py:1036-1040). */
int ot_detector_hits(const ot_rays* rays, int64_t first, int64_t count, const ot_surface* detector,
                     int32_t projection, const double* crop4, double* ph, float* hw, double* extent4,
                     int64_t* ill_count, void* stream);

/* The same for up to 8 detectors (or positions of one detector) in one pass over the ray sections: the sections are
 * read once, every request gets its own outputs.  Serves Raytracer.iterative_render with a list of detector positions
 * (raytracer.py:1235-1267). */
typedef struct ot_detector_req {
    const ot_surface* detector;
    int32_t projection;   /* OT_PROJ_*                                              */
    int32_t xy_only;      /* 1: ph is (count,2), the z plane is not written         */
    const double* crop4;  /* HOST f64[4] user extent or NULL                        */
    double* ph;           /* device (count,3) F-order, (count,2) with xy_only; ph and hw both NULL (extent4 given):  */
    float* hw;            /* device (count)      extent-only request, nothing but extent4 and ill_count is written   */
    double* extent4;      /* device f64[4] or NULL, initialised by the caller       */
    int64_t* ill_count;   /* device int64[2], added to                              */
    /* Compact hit list (fill != NULL; needs xy_only; ph may be NULL = no positions): only the VALID hits are written, into a list of OT_HIT_PIECES
     * pieces of L = ot_hit_piece_len(count) entries each -- CAPACITY = OT_HIT_PIECES * L entries, which the caller
     * allocates: ph = x plane [CAPACITY] then y plane [CAPACITY], hw [CAPACITY], wl_out [CAPACITY] (the hit's
     * wavelength: the list no longer lines up with the rays).  Piece k holds fill[k] hits at its front, [k * L,
     * k * L + fill[k]); which piece a hit lands in and its place there are arbitrary.  fill (device
     * uint32[OT_HIT_PIECES]) must be zero on entry.  Consumed by ot_render_accumulate_compact. */
    float* wl_out;        /* device (CAPACITY) or NULL                              */
    uint32_t* fill;       /* device uint32[OT_HIT_PIECES] or NULL = dense list      */
} ot_detector_req;
#define OT_HIT_PIECES 1024
int64_t ot_hit_piece_len(int64_t count); /* entries per piece of a compact hit list of capacity count: the power of
                                          * two at or above count / 1024, at least 1024 (the last pieces stay empty) */
int ot_detector_hits_multi(const ot_rays* rays, int64_t first, int64_t count, const ot_detector_req* reqs,
                           int32_t n_reqs, void* stream);

/* Raytracer.detector_image with a known extent (raytracer.py:1053-1098: _hit_detector + RenderImage.render), and
 * the per-chunk, per-position body of iterative_render after its first chunk (raytracer.py:1244-1267), in one pass
 * over the ray sections for up to 8 detectors (or positions of one detector): hit search, projection, user extent,
 * misc.binning_indices_2d and the XYZW histogram update without the hit positions ever being written to memory.
 * Per request: hist (Ny, Nx, 4) f64 device, ADDED to; extent = image extent after RenderImage.__fix_extent;
 * crop4 = the user extent hits are restricted to (HOST f64[4]) or NULL; ill_count as in ot_detector_hits.
 * Same sums as ot_detector_hits_multi + ot_render_accumulate, in another order.
 * Environment (tests, profiling): OT_RENDER_PATH = direct | tiles pins the binning path (default: a probe of 4096 rays
 * decides); OT_TILE_LINEBUF = 0 takes the plain tile kernel where the one with per-tile line buffers in LDS would run
 * (one detector, image of at most 361 tiles of 64 x 64 pixels; also for ot_detector_image_auto_begin). */
typedef struct ot_detector_image_req {
    const ot_surface* detector;
    int32_t projection;   /* OT_PROJ_*                                              */
    int32_t Nx, Ny;       /* pixel counts (render_image.py:383-387)                 */
    int32_t _pad;
    const double* crop4;  /* HOST f64[4] or NULL                                    */
    double extent[4];     /* image extent [x0, x1, y0, y1]                          */
    double* hist;         /* device (Ny, Nx, 4) f64                                 */
    int64_t* ill_count;   /* device int64[2], added to                              */
    double weight_scale;  /* every hit's weight times this (f64) before it is added: 1 for a plain image; the chunks of
                           * an iterative render pass rays_step / N (raytracer.py:1257) and bin straight into one image */
} ot_detector_image_req;
int ot_detector_images(const ot_rays* rays, int64_t first, int64_t count, const ot_detector_image_req* reqs,
                       int32_t n_reqs, void* stream);

/* Detector image with an AUTOMATIC extent (Raytracer.detector_image(extent=None), raytracer.py:1042-1049 + 1053-1098) in
 * one pass over the ray sections, for detectors with a closed-form hit (flat, conic / spherical) without a sphere
 * projection -- others: OT_ERR_UNSUPPORTED, use ot_detector_hits_multi (compact list) + ot_render_accumulate_compact.
 * The image extent is the bounding box of the hits, known only after the last one; the calls below bin the hits on a
 * provisional tile grid first and into the final pixel grid afterwards (csrc/ot_detector_fused.hpp, last section):
 *
 *   ot_detector_extent_sample      extent4 (x0, x1, y0, y1; +-inf without a hit) of the hits of every stride-th wave of 64
 *                                  rays.  It lies inside the extent of all hits.  The caller lays a grid of tiles[0] x
 *                                  tiles[1] <= 2048 tiles of tile[0] x tile[1] mm with its corner at origin over it (plus
 *                                  a margin), such that one tile covers at most 61 x 61 pixels of the final image.
 *   ot_detector_image_auto_begin   hit search, records (x, y, w, wl: 24 B per valid hit) by tile, result6 = the exact
 *                                  extent of all valid hits (as ot_detector_req.extent4 reports it), result6[4] = hits
 *                                  outside the grid (kept in a list of result6[5] entries: more than that -> cancel and
 *                                  take the other path).  OT_ERR_UNSUPPORTED also where the records (24 B per ray,
 *                                  worst case) do not fit the device: the hit-list path needs less.
 *   ot_detector_image_auto_finish  extent = the image extent after RenderImage.__fix_extent, Nx, Ny its pixel counts,
 *                                  hist (Ny, Nx, 4) f64 device, ADDED to.  Same pixels and sums as ot_detector_hits_multi +
 *                                  ot_render_accumulate (sums in another order).  Frees the handle, also on failure.
 *   ot_detector_image_auto_cancel  frees the handle without an image.
 *
 * extent4 / result6: device-visible host memory (pinned, mapped) or device memory (every entry is written by a kernel).
 * Both calls that report wait for the stream before they return.  The records of an image live in a scratch block the
 * handle LEASES from begin until finish / cancel: automatic images may be open side by side (also on one stream, each
 * gets a block of its own), ot_scratch_trim leaves an open image alone.  finish must be given the stream of begin. */
typedef struct ot_auto_image ot_auto_image;
int ot_detector_extent_sample(const ot_rays* rays, int64_t first, int64_t count, const ot_surface* detector,
                              int32_t projection, int32_t stride, double* extent4, void* stream);
int ot_detector_image_auto_begin(const ot_rays* rays, int64_t first, int64_t count, const ot_surface* detector,
                                 int32_t projection, const double origin[2], const double tile[2], const int32_t tiles[2],
                                 double* result6, ot_auto_image** out, void* stream);
int ot_detector_image_auto_finish(ot_auto_image* image, const double extent[4], int32_t Nx, int32_t Ny, double* hist,
                                  void* stream);
void ot_detector_image_auto_cancel(ot_auto_image* image);

/* The binning paths keep their scratch (up to ~25 B per ray and image) between calls: blocks keyed by device, stream
 * and purpose, grown on demand (calls on one stream run in order, so a block serves call after call), not tied to the
 * thread that created them.  A call leases its block until its last launch is enqueued; a block on lease is neither
 * handed to another caller nor freed.  The pool keeps at most a cap of bytes (64 GB; OT_SCRATCH_CAP_GB in the
 * environment, or ot_scratch_set_cap): beyond it, and when an allocation fails, idle blocks go least recently used
 * first.  ot_scratch_trim waits for the device and returns every IDLE block to the driver (for a caller whose own
 * allocator needs the room); ot_scratch_stats reports bytes kept, blocks and blocks on lease.  All three are
 * thread-safe. */
int ot_scratch_trim(void);
int ot_scratch_set_cap(int64_t bytes);
int ot_scratch_stats(int64_t* kept_bytes, int32_t* blocks, int32_t* leased);

/* SphericalSurface.sphere_projection (spherical_surface.py:36-97): p (n,3) F-order -> out (n,3) F-order */
int ot_sphere_projection(const ot_surface* surf, int32_t projection, int64_t n, const double* p, double* out,
                         void* stream);

/* RenderImage.render inner part (render_image.py:396-418 + misc.binning_indices_2d misc.py:59-91 +
 * color.x/y/z_observer observers.py:14-41): bins n hits into hist (Ny, Nx, 4) f64 device array,
 * ADDING w*[xbar(wl), ybar(wl), zbar(wl), 1].  extent = [x0,x1,y0,y1] after RenderImage.__fix_extent.
 * Rays with w == 0 are skipped (they add nothing in the reference either).  Lists of 2^21 hits or more that
 * spread over many pixels are binned tile by tile in LDS (stream-ordered scratch of ~11 B per hit from the
 * device's default memory pool, which is set to keep its memory); the environment variable OT_RENDER_PATH =
 * direct | tiles pins the path. */
int ot_render_accumulate(int64_t n, const double* px, const double* py, const float* w,
                         const float* wl, const double extent[4], int32_t Nx, int32_t Ny,
                         double* hist, void* stream);
/* The same for the compact hit list of a bundle of n rays (see ot_detector_req.fill; px, py, w, wl hold
 * OT_HIT_PIECES * ot_hit_piece_len(n) entries): piece k contributes its first fill[k] entries.  With an automatic extent the hit search writes a third of the bytes (valid hits only) and the binning
 * reads only those. */
int ot_render_accumulate_compact(int64_t n, const uint32_t* fill, const double* px, const double* py, const float* w,
                                 const float* wl, const double extent[4], int32_t Nx, int32_t Ny, double* hist,
                                 void* stream);

/* ---- image conversion (next row) -------------------------------------------------- */
#define OT_IMG_IRRADIANCE 0       /* render_image.py:180 */
#define OT_IMG_ILLUMINANCE 1      /* :184 */
#define OT_IMG_SRGB_ABSOLUTE 2    /* :189  color.xyz_to_srgb srgb.py:379, rendering intent "Absolute"   */
#define OT_IMG_SRGB_PERCEPTUAL 3  /* :189  rendering intent "Perceptual" (L_th, chroma_scale)           */
#define OT_IMG_OUTSIDE_GAMUT 4    /* :197  color.outside_srgb_gamut srgb.py:84                          */
#define OT_IMG_LIGHTNESS 5        /* :202  CIELUV L   luv.py xyz_to_luv                                 */
#define OT_IMG_HUE 6              /* :206  luv_hue                                                      */
#define OT_IMG_CHROMA 7           /* :211  luv_chroma                                                   */
#define OT_IMG_SATURATION 8       /* :216  luv_saturation                                               */

/* added to an sRGB mode: color.xyz_to_srgb(normalize=False) / (clip=False), srgb.py:379-407 (used by convolve()) */
#define OT_IMG_FLAG_NO_NORMALIZE 0x100
#define OT_IMG_FLAG_NO_CLIP 0x200

/* RenderImage.get (render_image.py:131-222): converts the (Ny, Nx, 4) float64 XYZW histogram into a display
 * quantity.  fact joins fact x fact bins first (the reference's cv2.resize INTER_AREA, :174; must divide Nx and
 * Ny).  apx = area of one ORIGINAL pixel, K = luminous efficacy.  chroma_scale = NaN selects the automatic value.
 * out: (Ny/fact, Nx/fact, 3) float64 for the sRGB modes, (Ny/fact, Nx/fact) float64 otherwise.
 * workspace: device scratch of 4*(Nx/fact)*(Ny/fact) + 8 doubles.  Synchronises the stream for the sRGB modes
 * (image-wide decisions of srgb.py:318, 209, 250 are taken on the host). */
int ot_image_convert(const double* hist, int32_t Nx, int32_t Ny, int32_t fact, int32_t mode, double apx, double K,
                     double L_th, double chroma_scale, double* out, double* workspace, void* stream);

/* RenderImage._apply_rayleigh_filter (render_image.py:257-296): out = "same"-size 2-D convolution of each of the
 * 4 channels of `in` (Ny, Nx, 4) with the (2*ps+1)^2 kernel `psf` (device, row-major), zero padded, negative
 * results clamped to 0.  in and out must not alias.  next row, SURVEY 8f rank 2. */
int ot_image_convolve(const double* in, int32_t Nx, int32_t Ny, const double* psf, int32_t ps, double* out, void* stream);

/* LightSpectrum.render (spectrum/light_spectrum.py:41-79), the histogram behind Raytracer.detector_spectrum
 * (raytracer.py:1100-1132) and Raytracer.source_spectrum (raytracer.py:1307-1328).  Rays come dense: weight 0 =
 * not selected (as ot_detector_hits leaves them).  next row, SURVEY 8f rank 3.
 * ot_spectrum_range: range2[0..1] (device) = min / max wavelength of the rays with w > 0 (+inf / -inf if none),
 *   count[0] (device) = number of such rays (np.count_nonzero(w), light_spectrum.py:60).
 * ot_spectrum_histogram: hist[nbins] (device, f64, accumulated into: zero it first) += weights per bin of the
 *   float32 edges[nbins + 1] (device; np.linspace(wl0, wl1, nbins + 1) as float32).  Bin search as NumPy's
 *   uniform-bin path in float32 (numpy/lib/_histograms_impl.py), last bin closed on the right. */
int ot_spectrum_range(int64_t n, const float* wl, const float* w, double* range2, int64_t* count, void* stream);
int ot_spectrum_histogram(int64_t n, const float* wl, const float* w, const float* edges, int32_t nbins, double* hist,
                          void* stream);
/* The same over a compact hit list of a bundle of n rays (ot_detector_req.fill; positions are not needed: such a request may
 * leave ph NULL): piece k contributes its first fill[k] entries of wl and w. */
int ot_spectrum_range_compact(int64_t n, const uint32_t* fill, const float* wl, const float* w, double* range2,
                              int64_t* count, void* stream);
int ot_spectrum_histogram_compact(int64_t n, const uint32_t* fill, const float* wl, const float* w, const float* edges,
                                  int32_t nbins, double* hist, void* stream);

/* Raytracer.focus_search (raytracer.py:1463-1640).  next row, SURVEY 8f rank 3.
 * ot_focus_prepare: for rays [first, first + count) pick the section crossing z (pos = argmax(z < p_z) - 1,
 *   raytracer.py:1552-1560) and write the hit line ph(z') = pa + sb * z' (raytracer.py:1580-1583):
 *   pasb[4 * count] (device) = pa_x | pa_y | sb_x | sb_y, w[count] (device) = section weight, -1 for rays
 *   without such a section; n_use[0] (device) = number of rays kept.
 * ot_focus_cost: cost function __focus_search_cost_function (raytracer.py:1354-1418) of the kept rays at the nz
 *   positions z[] (host array), mode OT_FOCUS_*; cost[nz] (device).  n_px = image side for the image methods
 *   (100 * int(1 + sqrt(N) / 1500), made odd), workspace (device) >= OT_FOCUS_WS + n_px * n_px doubles.
 * ot_focus_moments: sums[16] (device) for __focus_rms_spot_direct_solution (raytracer.py:1420-1460), the mean
 *   position and the RMS cost curve: [0..4] = sum w, w pa_x, w pa_y, w sb_x, w sb_y; [5] = sum w^2 (dtx^2 + dty^2);
 *   [6] = sum w^2 (dtx dx + dty dy) for the bounds b0 < b1; [7] = sum w^2; [8..10] = sum w x0'^2, w x0' sbx', w sbx'^2
 *   with x0' = pa_x + sb_x z0 - mean, sbx' = sb_x - mean, z0 = (b0 + b1) / 2, [11..13] the same in y: the weighted
 *   variance at any z is ([8] + 2 (z - z0) [9] + (z - z0)^2 [10]) / ([0] - [7] / [0]) (np.cov with aweights). */
#define OT_FOCUS_RMS 0
#define OT_FOCUS_IRR_VAR 1
#define OT_FOCUS_SHARPNESS 2
#define OT_FOCUS_CENTER_SHARPNESS 3
#define OT_FOCUS_WS 16
int ot_focus_prepare(const ot_rays* rays, int64_t first, int64_t count, double z, double* pasb, float* w, int64_t* n_use,
                     void* stream);
int ot_focus_cost(int64_t count, const double* pasb, const float* w, int32_t mode, const double* z, int32_t nz,
                  int32_t n_px, double* workspace, double* cost, void* stream);
int ot_focus_moments(int64_t count, const double* pasb, const float* w, double b0, double b1, double* sums, void* stream);

/* ---- diagnostics ------------------------------------------------------------------------------------------
 * The tracing loop issues f64 division and square root as their bare cores (reciprocal / reciprocal-square-root seed +
 * the refinement steps of the IEEE sequence, without its range scaling and special-value fix-up; csrc/ot_device.hpp).
 * Bit-exact hit masks need those cores to return the bits of IEEE `/` and sqrt on every operand the path produces.
 * ot_selftest_arith draws n random operand sets of a class ON THE DEVICE, evaluates core and IEEE operator side by side
 * and counts sets whose result bits differ (two NaNs count as equal); first_bad4 (HOST f64[4]) receives the operands
 * a, b, c and the core's first result of one differing set.  mismatches is a HOST int64.
 *   op:    0 ot_div(a, b) | 1 ot_sqrt(|a|) | 2 normalize3(a, b, c) | 3 two quotients a / c, b / c sharing one reciprocal
 *   class: 0 uniform mantissas, exponents +-500 (sqrt: 2^-760 .. 2^1020, normalize3: +-250) | 1 mm geometry 2^-20 .. 2^14
 *          | 2 refractive indices [1, 2.5] | 3 direction cosines near 0 and near 1
 * ot_selftest_eval evaluates both on caller-supplied DEVICE operands (special values); outputs are (n, 3) F-order. */
int ot_selftest_arith(int32_t op, int32_t operand_class, int64_t n, uint64_t seed, int64_t* mismatches,
                      double* first_bad4, void* stream);
int ot_selftest_eval(int32_t op, int64_t n, const double* a, const double* b, const double* c, double* core_out,
                     double* ieee_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OPTRACE_AMD_H */
