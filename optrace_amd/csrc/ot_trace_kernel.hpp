// The tracing kernel and its launch plumbing, shared by the translation units that instantiate its variants.
//
// One translation unit per feature level (ot_trace_f<level>.hip) instantiates the 10 variants of that level
// (polarisation x generation x spectrum handling); ot_api.hip only calls launch_trace_feat<level>().  The split is a
// build-time matter (the variants compile in parallel, one level rebuilds alone) and changes nothing in the kernels.
#pragma once
#include "ot_trace.hpp"

#define OT_CNT_SLOTS 1024  // counter slot tables per scene (see trace_kernel)

// ---------------------------------------------------------------------------------------------------------
#define OT_MAX_RANGES 64
#ifndef OT_TRACE_MIN_WAVES
#define OT_TRACE_MIN_WAVES 1
#endif
// Feature level 0 kernels (flat and conic surfaces only): five waves per SIMD asked for.  The continuous-spectrum
// variants (C3, C4) need 90-92 registers since the dispersion formulas no longer pull in the device library's pow
// (ot_device.hpp::ot_powi; 109-122 before, i.e. four waves or spills): no scratch in any of them.
#ifndef OT_TRACE_MIN_WAVES_F0
#define OT_TRACE_MIN_WAVES_F0 5
#endif
// waves per SIMD asked of the register allocator, per feature level (see OT_FEAT below)
// (discrete-spectrum kernels of level 0 -- the bench scene -- fit six: asked for explicitly, the allocator otherwise
// spreads out over whatever five waves leave it)
#ifndef OT_TRACE_MIN_WAVES_F0_LINES
#define OT_TRACE_MIN_WAVES_F0_LINES 6
#endif
// level 1 (ideal lenses, filters, HURB on closed-form surfaces) fits four waves without a spill once asked (127-128
// registers instead of 129-131); level 3 (the same on top of the asphere search, 140-148) spills 4-10 values at four
// waves and is still faster there: the asphere test scene 1.56 -> 1.47 ms at 1e7 rays (two runs each on one box)
#ifndef OT_TRACE_MIN_WAVES_F1
#define OT_TRACE_MIN_WAVES_F1 4
#endif
#ifndef OT_TRACE_MIN_WAVES_F3
#define OT_TRACE_MIN_WAVES_F3 4
#endif
// spline level without HURB: 165-169 registers, asked to stay inside the 168 of three waves (the patch buffer in LDS,
// 51 KB per workgroup, allows three workgroups per CU as well)
#ifndef OT_TRACE_MIN_WAVES_F4
#define OT_TRACE_MIN_WAVES_F4 3
#endif
#define OT_TRACE_WAVES(FEAT, SPEC, POL)                                                                    \
    ((FEAT) == 0 ? ((SPEC) == 2 ? OT_TRACE_MIN_WAVES_F0_LINES : OT_TRACE_MIN_WAVES_F0)                     \
                 : ((FEAT) == 1 ? OT_TRACE_MIN_WAVES_F1                                                    \
                                : ((FEAT) == 3 ? OT_TRACE_MIN_WAVES_F3 : ((FEAT) == 4 ? OT_TRACE_MIN_WAVES_F4 : OT_TRACE_MIN_WAVES))))

struct RangeRec {  // one source range in device memory (scenes with more than OT_MAX_RANGES ranges)
    int64_t first, count;
    int32_t source;
    uint32_t n2;
    double inv_n, inv_n2;
    float w;
};

struct RangeArgs {
    int32_t n;
    int32_t source[OT_MAX_RANGES];
    int64_t first[OT_MAX_RANGES];
    int64_t count[OT_MAX_RANGES];
    // per-range constants of the stratified samplers, evaluated once on the host: 1 / count,
    // floor(sqrt(count)) and its reciprocal (random.py:23-31, 62)
    uint32_t n2[OT_MAX_RANGES];
    double inv_n[OT_MAX_RANGES];
    double inv_n2[OT_MAX_RANGES];
    float w[OT_MAX_RANGES];  // power of each ray of the range
    const RangeRec* ext;  // n > OT_MAX_RANGES: n records sorted by `first`, contiguous; the arrays above are unused
};

OT_DEV bool locate_range(const RangeArgs& rg, int64_t ray, GenCtx& g, int& k, int& src) {
    if (rg.ext) {  // many ranges: binary search for the last record starting at or before this ray
        int lo = 0, hi = rg.n - 1;
        while (lo < hi) {
            int mid = (lo + hi + 1) >> 1;
            if (rg.ext[mid].first <= ray)
                lo = mid;
            else
                hi = mid - 1;
        }
        const RangeRec rr = rg.ext[lo];
        if (ray < rr.first || ray >= rr.first + rr.count) return false;
        g.j = (uint32_t)(ray - rr.first);
        g.n = (uint32_t)rr.count;
        g.n2 = rr.n2;
        g.inv_n = rr.inv_n;
        g.inv_n2 = rr.inv_n2;
        g.w = rr.w;
        k = lo;
        src = rr.source;
        return true;
    }
    for (int q = 0; q < rg.n; q++) {
        if (ray >= rg.first[q] && ray < rg.first[q] + rg.count[q]) {
            g.j = (uint32_t)(ray - rg.first[q]);
            g.n = (uint32_t)rg.count[q];
            g.n2 = rg.n2[q];
            g.inv_n = rg.inv_n[q];
            g.inv_n2 = rg.inv_n2[q];
            g.w = rg.w[q];
            k = q;
            src = rg.source[q];
            return true;
        }
    }
    return false;
}

// Generates the ray of every lane that has one.  Source ranges span millions of rays, so a wavefront nearly
// always lies inside one range: then the range index, its ray count and its source record are WAVE-UNIFORM
// (broadcast from the first lane) -- the source is read with scalar loads, the switches over shape / spectrum /
// divergence / polarisation are scalar branches, permutation keys come from the scalar ALU.  A wave that
// straddles a range boundary takes the per-lane path.  (A loop over the ranges of a wave instead of the two
// paths made the register allocator give up: 256 VGPRs.)
template <bool IMAGES = true>
OT_DEV bool generate_lane(const RangeArgs& rg, const SourceDev* __restrict__ sources, int64_t ray, uint64_t seed,
                          bool no_pol, NewRay& nr) {
    GenCtx g;
    g.seed = seed;
    g.gidx = (uint64_t)ray;
    // the range of the wave's first ray, found on the scalar unit: the records are sorted and contiguous
    // (make_ranges), so it is the last one that starts at or before that ray
    const uint64_t ray0 = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)((uint64_t)ray >> 32)) << 32) |
                          (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ray);
    int lo = 0, hi = rg.n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        const int64_t f = rg.ext ? as_const(rg.ext)[mid].first : rg.first[mid];
        if ((uint64_t)f <= ray0)
            lo = mid;
        else
            hi = mid - 1;
    }
    int64_t first0, count0;
    int src0;
    if (rg.ext) {
        const auto& rr = as_const(rg.ext)[lo];
        first0 = rr.first; count0 = rr.count; src0 = rr.source;
        g.n2 = rr.n2; g.inv_n = rr.inv_n; g.inv_n2 = rr.inv_n2; g.w = rr.w;
    } else {
        first0 = rg.first[lo]; count0 = rg.count[lo]; src0 = rg.source[lo];
        g.n2 = rg.n2[lo]; g.inv_n = rg.inv_n[lo]; g.inv_n2 = rg.inv_n2[lo]; g.w = rg.w[lo];
    }
    const bool inside = ray >= first0 && ray < first0 + count0;
    if (__ballot(inside) == __ballot(true)) {  // one range in this wave
        g.j = (uint32_t)(ray - first0);
        g.n = (uint32_t)count0;
        g.range = (uint32_t)lo;
        const auto& S = as_const(sources)[src0];
        fill_dither_for<IMAGES>(g, S);
        nr = generate_ray<IMAGES>(S, g, no_pol);
        return true;
    }
    int k = -1, src = 0;
    const bool have = locate_range(rg, ray, g, k, src);
    if (have) {
        g.range = (uint32_t)k;
        fill_dither_for<IMAGES>(g, sources[src]);
        nr = generate_ray<IMAGES>(sources[src], g, no_pol);
    }
    return have;
}

// Raytracer.trace: optional on-the-fly generation, then all steps.  One ray per lane, 256-thread workgroups
// (4 wave64); template switches select a kernel that only contains what the scene needs:
//   POL  polarisation tracked          GEN  rays generated in registers (no section-0 round trip)
//   SPEC 0 dispersion formulas only   1 + tabulated media / filters or injected HURB normals (per-lane global loads inside the
//        loop)   2 discrete spectra: n, n1 / n2 and filter transmissions per line, staged in LDS
//   FEAT = OT_FEAT(hit level, full), six levels (below): hit level 0 flat and conic surfaces, 1 + aspheres and tilted
//        surfaces (Illinois search on closed-form sags), 2 + spline surfaces; full = HURB (at hit level 0 also ideal lenses and
//        filters).  Registers / waves per SIMD of every variant: profiles/r3/resource_usage.txt (level 0: 60-92 / 5-8, the
//        bench kernel <true, true, 2, 0> 74 / 6; level 5: 205-212 / 2)
// Event counters go wave -> LDS (per workgroup) -> one of OT_CNT_SLOTS global slot tables (blockIdx % slots) ->
// reduce_counters_kernel, so that no two workgroups hammer the same address (see count_event).
// The launch covers the rays [ray_base, ray_base + count) of the bundle; R's pointers are advanced to ray_base by
// the host (R.N stays the plane stride), so lanes address their ray with a 32-bit offset (count <= 2^28).
template <bool POL, bool GEN, int SPEC, int FEAT>
__global__ __launch_bounds__(256, OT_TRACE_WAVES(FEAT, SPEC, POL)) void trace_kernel(const SceneDev* __restrict__ scp, ot_rays R,
                                                    const SourceDev* __restrict__ sources, RangeArgs rg,
                                                    const double* __restrict__ hurb_normals, uint64_t seed,
                                                    unsigned int* __restrict__ slots, int64_t ray_base,
                                                    uint32_t count) {
    extern __shared__ double lds[];  // [discrete-spectrum table (SPEC == 2)] [event counters + timeout flag]
    auto& sc = *as_const(scp);
    const int n_tab = (SPEC == 2) ? (3 * sc.n_steps + 2) * OT_MAX_LINES : 0;
    double* ltab = lds;
    unsigned int* cnt = (unsigned int*)(lds + n_tab);  // (OT_N_INFOS x nt) counters + 1 flag
    const int n_cnt = OT_N_INFOS * sc.nt + 1;
    for (int k = threadIdx.x; k < n_cnt; k += blockDim.x) cnt[k] = 0u;
    if (SPEC == 2)
        for (int k = threadIdx.x; k < n_tab; k += blockDim.x) ltab[k] = sc.line_tab[k];
    __syncthreads();

    const uint32_t local = blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t ray = ray_base + (int64_t)local;
    bool have = local < count;
    RayState r;
    if (have) {
        if (GEN) {
            NewRay nr;
            have = generate_lane<(SPEC != 2)>(rg, sources, ray, seed, !POL, nr);  // no image sources with SPEC 2 (launch_trace)
            if (have) {
                r.p = nr.p;
                r.s = nr.s;
                r.w = nr.w;
                r.wl = nr.wl;
                r.polx = (float)nr.polx;
                r.poly = (float)nr.poly;
                r.polz = (float)nr.polz;
                __builtin_nontemporal_store(r.wl, &R.wl[local]);  // written once, streamed (see OT_STORE_HINT)
            }
        } else {
            const int64_t N = R.N, nt = R.nt;
            r.p.x = R.p[local];
            r.p.y = R.p[local + N * nt];
            r.p.z = R.p[local + N * 2 * nt];
            r.s.x = R.s[local];
            r.s.y = R.s[local + N];
            r.s.z = R.s[local + 2 * N];
            r.w = R.w[local];
            r.wl = R.wl[local];
            r.polx = r.poly = r.polz = 0.f;
            if (POL) {
                r.polx = R.pol[local];
                r.poly = R.pol[local + N * nt];
                r.polz = R.pol[local + N * 2 * nt];
            }
        }
    }
    // RaySource.create_rays raises if any generated direction has s_z <= 0 (ray_source.py:353).  Reported as
    // counter (HURB_NEG_DIR, section 0), a cell no tracing event can touch; the host turns it into that error.
    if (GEN) count_event(cnt, sc.nt, OT_INFO_HURB_NEG_DIR, 0, have && !(r.s.z > 0));
    if (have) {
        int lj = 0;  // discrete spectra: which line this ray carries
        if (SPEC == 2) {
            for (int j = 1; j < sc.n_lines; j++)
                if ((float)ltab[j] == r.wl) lj = j;
        }
        // spline level: 25 doubles per lane behind the counters for the spline patch cache
        double* patch = (FEAT / 2 >= OT_HIT_SPLINE) ? lds + n_tab + (n_cnt + 2) / 2 : nullptr;
        bool ok = trace_ray<POL, SPEC, FEAT>(sc, R, local, (uint64_t)ray, r, hurb_normals, seed, cnt, ltab, lj, patch);
        if (!ok) cnt[n_cnt - 1] = 1u;  // numeric hit search timed out (surface.py:403)
    }
    __syncthreads();
    unsigned int* slot = slots + (size_t)(blockIdx.x % OT_CNT_SLOTS) * n_cnt;
    for (int k = threadIdx.x; k < n_cnt; k += blockDim.x)
        if (cnt[k]) atomicAdd(&slot[k], cnt[k]);
}

// ---- render-only tracing: the living rays' LAST SECTION instead of every section -------------------------------------
// `iterative_render` keeps the rays of its last chunk only (raytracer.py:1235-1267): every chunk before it exists to be
// binned.  A detector behind the last surface sees a ray through its last section alone -- positions at sections nt - 2 and
// nt - 1, the weight at nt - 2, the wavelength (raytracer.py:929-985) -- and only if the ray is still alive there.  This kernel
// is trace_kernel with on-device generation and without ANY section store; at the end every wave writes that 56-byte record
// for its living rays into a compact two-section ray storage (`TailOut`: the layout of ot_rays with nt = 2), which the
// detector kernels read like any other storage.  C4 (2 surfaces, no_pol): 20 B per traced ray instead of 172 B written, and
// the detector passes read 20 B instead of 56 B per traced ray.
// Compaction as for the compact hit lists (ot_api.hip::ot_detector_hits_multi): wave k takes `cnt` slots of piece k mod 1024
// with ONE returning atomic on that piece's fill count (waves in flight spread over all counters).  The pieces are
// interleaved in rows of 64 entries -- slot q of piece p lives at ((q >> 6) * 1024 + p) * 64 + (q & 63) -- and fill at the same
// rate, so the storage is dense up to the fullest piece's row; tail_seal_kernel clears the ragged end (weight 0) and reports
// the number of slots in use.
#define OT_TAIL_PIECES 1024
struct TailOut {
    double* p;           // (cap, 2, 3) f64, F order like ot_rays.p with nt = 2
    float* w;            // (cap, 2) f32; section 1 = 0
    float* wl;           // (cap) f32
    unsigned int* fill;  // [OT_TAIL_PIECES] slots taken per piece
    int64_t cap;         // plane stride = 65536 * rows
};

template <bool POL, int SPEC, int FEAT>
__global__ __launch_bounds__(256, OT_TRACE_WAVES(FEAT, SPEC, POL)) void trace_tail_kernel(
    const SceneDev* __restrict__ scp, TailOut T, const SourceDev* __restrict__ sources, RangeArgs rg, uint64_t seed,
    unsigned int* __restrict__ slots, int64_t ray_base, uint32_t count) {
    extern __shared__ double lds[];  // as in trace_kernel
    auto& sc = *as_const(scp);
    const int n_tab = (SPEC == 2) ? (3 * sc.n_steps + 2) * OT_MAX_LINES : 0;
    double* ltab = lds;
    unsigned int* cnt = (unsigned int*)(lds + n_tab);
    const int n_cnt = OT_N_INFOS * sc.nt + 1;
    for (int k = threadIdx.x; k < n_cnt; k += blockDim.x) cnt[k] = 0u;
    if (SPEC == 2)
        for (int k = threadIdx.x; k < n_tab; k += blockDim.x) ltab[k] = sc.line_tab[k];
    __syncthreads();

    const uint32_t local = blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t ray = ray_base + (int64_t)local;
    bool have = local < count;
    RayState r;
    TailState tail;
    tail.w = 0.f;
    tail.p = {0.0, 0.0, 0.0};
    r.p = {0.0, 0.0, 0.0};
    r.wl = 0.f;
    if (have) {
        NewRay nr;
        have = generate_lane<(SPEC != 2)>(rg, sources, ray, seed, !POL, nr);
        if (have) {
            r.p = nr.p;
            r.s = nr.s;
            r.w = nr.w;
            r.wl = nr.wl;
            r.polx = (float)nr.polx;
            r.poly = (float)nr.poly;
            r.polz = (float)nr.polz;
        }
    }
    count_event(cnt, sc.nt, OT_INFO_HURB_NEG_DIR, 0, have && !(r.s.z > 0));
    if (have) {
        int lj = 0;
        if (SPEC == 2) {
            for (int j = 1; j < sc.n_lines; j++)
                if ((float)ltab[j] == r.wl) lj = j;
        }
        const ot_rays none = {};
        // spline level: 25 doubles per lane behind the counters for the spline patch cache (as in trace_kernel)
        double* patch = (FEAT / 2 >= OT_HIT_SPLINE) ? lds + n_tab + (n_cnt + 2) / 2 : nullptr;
        bool ok = trace_ray<POL, SPEC, FEAT, true>(sc, none, local, (uint64_t)ray, r, (const double*)nullptr, seed, cnt, ltab, lj,
                                                   patch, &tail);
        if (!ok) cnt[n_cnt - 1] = 1u;
    }
    // the living rays of this wave, compacted
    const bool alive = have && tail.w > 0.f;
    const unsigned long long m = __ballot(alive);
    if (m) {
        const unsigned int n_alive = (unsigned int)__popcll(m);
        const unsigned int rank = __builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u));
        const uint64_t wave = ((uint64_t)ray_base >> 6) + (local >> 6);
        const unsigned int piece = (unsigned int)(wave & (OT_TAIL_PIECES - 1));
        unsigned int q0 = 0;
        if (rank == 0 && alive) q0 = atomicAdd(&T.fill[piece], n_alive);  // (the first living lane)
        q0 = __shfl(q0, __ffsll((long long)m) - 1);
        if (alive) {
            const unsigned int q = q0 + rank;
            const int64_t slot = (((int64_t)(q >> 6) * OT_TAIL_PIECES + piece) << 6) + (q & 63u);
            const int64_t N = T.cap;
            double* px = T.p + slot;  // element (ray, section, component) at ray + N * (section + 2 * component)
            px[0] = tail.p.x;
            px[N] = r.p.x;
            px[2 * N] = tail.p.y;
            px[3 * N] = r.p.y;
            px[4 * N] = tail.p.z;
            px[5 * N] = r.p.z;
            T.w[slot] = tail.w;
            T.w[N + slot] = 0.f;  // every ray ends absorbed (the end aperture): the weight of its last section
            T.wl[slot] = r.wl;
        }
    }
    __syncthreads();
    unsigned int* slot_tab = slots + (size_t)(blockIdx.x % OT_CNT_SLOTS) * n_cnt;
    for (int k = threadIdx.x; k < n_cnt; k += blockDim.x)
        if (cnt[k]) atomicAdd(&slot_tab[k], cnt[k]);
}

// Feature levels of the kernel variants.  Bit 0 ("full"): HURB, and at hit level 0 ideal lenses and filters as well (the
// higher hit levels always carry those two, see trace_ray).  Upper part = hit level:
//   OT_HIT_CLOSED   flat and conic surfaces (closed-form hit)
//   OT_HIT_ILLINOIS + aspheres and tilted surfaces: the numeric hit search on closed-form sag functions
//   OT_HIT_SPLINE   + data / function surfaces: spline tables, per-lane coefficient patch in LDS, mask bitmaps
#define OT_FEAT(hit, full) (2 * (hit) + ((full) ? 1 : 0))
#define OT_FEAT_HIT(feat) ((feat) / 2)
#define OT_FEAT_FULL(feat) (((feat)&1) != 0)

// everything a launch of one variant needs
struct TraceLaunch {
    dim3 grid;
    size_t lds;
    hipStream_t st;
    const SceneDev* sc;
    ot_rays part;
    const SourceDev* sd;
    const RangeArgs* rg;
    const double* hurb_normals;
    uint64_t seed;
    unsigned int* slots;
    int64_t base;
    uint32_t count;
    bool pol, gen;
    int spec;
};

// defined in ot_trace_f<FEAT>.hip through OT_DEFINE_TRACE_LAUNCHER
template <int FEAT>
void launch_trace_feat(const TraceLaunch& L);
// render-only variants (generation on the device, no injected HURB normals): defined in ot_trace_t<FEAT>.hip
template <int FEAT>
void launch_trace_tail_feat(const TraceLaunch& L, const TailOut& T);

#define OT_DEFINE_TRACE_TAIL_LAUNCHER(FEAT)                                                                             \
    template <>                                                                                                         \
    void launch_trace_tail_feat<FEAT>(const TraceLaunch& L, const TailOut& T) {                                         \
        const dim3 block(256);                                                                                          \
        auto go = [&](auto kern) {                                                                                      \
            hipLaunchKernelGGL(kern, L.grid, block, L.lds, L.st, L.sc, T, L.sd, *L.rg, L.seed, L.slots, L.base, L.count); \
        };                                                                                                              \
        if (L.spec == 2) { if (L.pol) go(trace_tail_kernel<true, 2, FEAT>); else go(trace_tail_kernel<false, 2, FEAT>); } \
        else if (L.spec == 1) { if (L.pol) go(trace_tail_kernel<true, 1, FEAT>); else go(trace_tail_kernel<false, 1, FEAT>); } \
        else { if (L.pol) go(trace_tail_kernel<true, 0, FEAT>); else go(trace_tail_kernel<false, 0, FEAT>); }           \
    }

#define OT_DEFINE_TRACE_LAUNCHER(FEAT)                                                                                  \
    template <>                                                                                                         \
    void launch_trace_feat<FEAT>(const TraceLaunch& L) {                                                                \
        const dim3 block(256);                                                                                          \
        auto go = [&](auto kern) {                                                                                      \
            hipLaunchKernelGGL(kern, L.grid, block, L.lds, L.st, L.sc, L.part, L.sd, *L.rg, L.hurb_normals, L.seed,     \
                               L.slots, L.base, L.count);                                                               \
        };                                                                                                              \
        if (L.gen) {                                                                                                    \
            if (L.spec == 2) { if (L.pol) go(trace_kernel<true, true, 2, FEAT>); else go(trace_kernel<false, true, 2, FEAT>); } \
            else if (L.spec == 1) { if (L.pol) go(trace_kernel<true, true, 1, FEAT>); else go(trace_kernel<false, true, 1, FEAT>); } \
            else { if (L.pol) go(trace_kernel<true, true, 0, FEAT>); else go(trace_kernel<false, true, 0, FEAT>); }     \
        } else {                                                                                                        \
            if (L.spec == 1) { if (L.pol) go(trace_kernel<true, false, 1, FEAT>); else go(trace_kernel<false, false, 1, FEAT>); } \
            else { if (L.pol) go(trace_kernel<true, false, 0, FEAT>); else go(trace_kernel<false, false, 0, FEAT>); }   \
        }                                                                                                               \
    }
