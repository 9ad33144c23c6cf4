// Detector image in one pass over the ray sections: Raytracer._hit_detector (raytracer.py:881-1051) and
// RenderImage.render (render_image.py:361-421) fused for images whose extent is known before the hit search (a
// user extent, or every chunk of an iterative render after the first, raytracer.py:1262).
//
// The unfused chain writes every hit position to HBM (20-28 B per ray and detector), and the binning reads it back
// twice (tile_count, tile_scatter): 140 B per ray for one detector.  Here the hit kernel turns a hit straight into
// its pixel, so the positions never exist in memory:
//
//   probe    one workgroup per detector intersects 4096 sample rays: few distinct pixels -> direct binning
//   direct   hit + pixel + LDS-hash privatised f64 adds (point-like images), sections read once for all detectors
//   tiles    hit + pixel + 12-byte record (w, wl, pixel in tile) appended to a chunk list of the record's image
//            tile; a workgroup keeps one open chunk per tile and takes new ones from its own part of a pool (an
//            LDS counter), so records of a tile end up in 3 KB runs without any counting pre-pass  reads 56, writes 12 B
//   index    the chunks are grouped by tile (counting sort over ~count / 256 chunk numbers)
//   accum    per tile: ds_add_f64 of its chunks' records into the LDS tile, slabs, reduce        reads 12 B
//
// 80 B per ray and detector instead of 140, and 56 + 24 * n for n detector positions instead of 52 + 88 * n.
#pragma once
#include <type_traits>
#include "ot_render_tiles.hpp"

#define OT_FUSE_CH 256       // records per chunk (3 KB)
#ifndef OT_FUSE_BR
#define OT_FUSE_BR 512       // rays per sub-block = threads per workgroup of the tile kernel
#endif
#ifndef OT_FUSE_WG_PER_CU
#define OT_FUSE_WG_PER_CU 2  // ... and its persistent workgroups per CU (A/B: 1024 threads x 1, profiles/r4/tile_kernel_wg_ab.txt)
#endif
#ifndef OT_FUSE_CPW
#define OT_FUSE_CPW 256      // chunks per accumulation workgroup (and slab): 64 records per thread
#endif
#define OT_FUSE_NONE 0xffffffffu
#define OT_FUSE_LDS_ENTRIES 2400  // (detector, tile) entries a tile-kernel workgroup can keep (20 B each)

// Automatic extent in one pass ("speculative grid", see the section at the end of this file): tiles of a provisional
// grid laid over the extent of a sample of the rays; records carry the hit position itself.
struct SpecRec {
    double x, y;
    float w, wl;
};
struct SpecGrid {
    double X0, Y0;    // lower left corner of tile (0, 0)
    double tw, th;    // tile width / height (mm), and their reciprocals
    double itw, ith;
    int32_t tx, ty;   // tiles along x / y
    unsigned long long* ext_slots;  // OT_EXT_SLOTS x 4 ordered values: extent of the valid hits
    SpecRec* esc;                   // hits outside the grid (binned with global atomics)
    unsigned int* esc_n;            // [1] hits that asked for a place in esc
    unsigned int esc_cap;
};

struct FuseOne {  // one detector of a fused launch
    SurfDev det;
    double Rcurv;
    Crop crop;
    int projection;
    RenderArgs a;
    int32_t tx, K;   // tiles along x, in all
    int32_t koff;    // first entry of this detector in the workgroup's LDS arrays
    int32_t tiles_ok;  // 0: this detector can only be binned directly (no pool)
    unsigned long long* ill;  // [2] ill-conditioned, timed out
    int* spread;              // [1] probe verdict: 1 = tile path
    double* hist;             // (Ny, Nx, 4), added to
    // chunk pool of the tile path
    uint32_t* chunk_tile;  // [cap]
    uint32_t* chunk_fill;  // [cap]
    TileRec* rec;          // [cap * OT_FUSE_CH]
    uint32_t cap;          // = workgroups of the tile kernel * per_wg
    uint32_t per_wg;       // every workgroup hands out chunks of its own part of the pool: no global atomics
    int* overflow;         // [1] set if a part ran dry (cannot happen with the size the host computes)
    SpecGrid g;            // SPECX kernels only
};

template <class FT>
OT_DEV int fuse_pixel(FT& F, const V3& ph, int32_t& ix, int32_t& iy) {
    RenderArgs a;  // (field by field: F lives in the constant address space)
    a.x0 = F.a.x0;
    a.x1 = F.a.x1;
    a.y0 = F.a.y0;
    a.y1 = F.a.y1;
    a.fx = F.a.fx;
    a.fy = F.a.fy;
    a.Nx = F.a.Nx;
    a.Ny = F.a.Ny;
    a.ws = 1.0;
    return hit_pixel(a, ph.x, ph.y, ix, iy);
}

// ---- probe: distinct pixels among the hits of 4096 sample rays ---------------------------------------------
// grid (detectors, OT_TILE_PROBE / 256): one sample ray per thread, the pixel set and the count per detector in global
// memory (pset: OT_TILE_PROBE_SET ints preset to -1, pcnt: {distinct, workgroups done} preset to 0); the last workgroup
// of a detector writes the verdict.  (One 1024-thread workgroup per detector walking its 4096 samples in four rounds on a
// single CU took 70 us in front of every image.)
#define OT_FUSE_PROBE_WG 256
template <bool NUMERIC>
__global__ __launch_bounds__(OT_FUSE_PROBE_WG) void fuse_probe_kernel(ot_rays R, int64_t first, int64_t count,
                                                                      const FuseOne* __restrict__ dets, int* __restrict__ pset_all,
                                                                      int* __restrict__ pcnt_all) {
    const auto& F = as_const(dets)[blockIdx.x];
    int* pset = pset_all + (size_t)blockIdx.x * OT_TILE_PROBE_SET;
    int* pcnt = pcnt_all + 2 * blockIdx.x;
    const int64_t S = count < OT_TILE_PROBE ? count : OT_TILE_PROBE;
    const int64_t stride = count / S;
    const int64_t k = (int64_t)blockIdx.y * blockDim.x + threadIdx.x;
    const bool in_sample = k < S;
    const int64_t r = first + (in_sample ? k : 0) * stride;
    const SectionPair sp = load_section_pair(R, r, in_sample);
    V3 ph;
    float w;
    bool valid, ill, to;
    detector_hit<NUMERIC>(R, r, in_sample, F, sp, pair_direction(sp), ph, w, valid, ill, to);
    int32_t ix, iy;
    const int pix = valid ? fuse_pixel(F, ph, ix, iy) : -1;
    bool is_new = false;
    if (pix >= 0) {
        unsigned int h = ((unsigned int)pix * 2654435761u) >> (32 - 13);  // OT_TILE_PROBE_SET = 2^13
        for (int pr = 0; pr < OT_TILE_PROBE_SET; pr++) {  // the set is twice as large as the sample: always ends
            const int sidx = (int)((h + pr) & (OT_TILE_PROBE_SET - 1));
            const int k0 = atomicCAS(&pset[sidx], -1, pix);
            if (k0 == -1) {
                is_new = true;
                break;
            }
            if (k0 == pix) break;
        }
    }
    const unsigned long long m_new = __ballot(is_new);
    if (__lane_id() == 0 && m_new) atomicAdd(&pcnt[0], (int)__popcll(m_new));
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        if (atomicAdd(&pcnt[1], 1) == (int)gridDim.y - 1) {  // every other workgroup's adds are behind its fence
            const int distinct = atomicAdd(&pcnt[0], 0);
            F.spread[0] = (distinct > OT_TILE_DISTINCT) && F.tiles_ok;
        }
    }
}

// wave-aggregated report of the numeric hit search
template <class FT>
OT_DEV void fuse_count_ill(FT& F, bool any_ill, bool timeout) {
    const unsigned long long m_ill = __ballot(any_ill), m_to = __ballot(timeout);
    if (__lane_id() == 0) {
        if (m_ill) atomicAdd(&F.ill[0], (unsigned long long)__popcll(m_ill));
        if (m_to) atomicAdd(&F.ill[1], (unsigned long long)__popcll(m_to));
    }
}

// ---- direct path: detectors whose hits fall into few pixels -------------------------------------------------
// render_kernel (ot_detector.hpp) with the hit search in front; the LDS hash is shared by the detectors of the launch
// (key = pixel * 8 + detector).
// GENERAL = false: flat / conic detectors without a sphere projection (the usual case) -- a kernel without the Illinois
// loop, the spline code and the projection polynomials.
template <bool GENERAL, int NDET>
__global__ __launch_bounds__(1024) void fuse_direct_kernel(ot_rays R, int64_t first, int64_t count,
                                                           const FuseOne* __restrict__ dets, int n_det,
                                                           const double* __restrict__ table) {
    bool any = false;
    for (int d = 0; d < n_det; d++) any = any || !as_const(dets)[d].spread[0];
    if (!any) return;
    __shared__ double obs[OT_OBS_N * 3];
    __shared__ double hval[OT_HASH_N * 4];
    __shared__ int hkey[OT_HASH_N];
    for (int i = threadIdx.x; i < OT_OBS_N * 3; i += blockDim.x) obs[i] = table[i];
    for (int i = threadIdx.x; i < OT_HASH_N; i += blockDim.x) hkey[i] = OT_HASH_EMPTY;
    for (int i = threadIdx.x; i < OT_HASH_N * 4; i += blockDim.x) hval[i] = 0.0;
    __syncthreads();
    const int64_t chunk = ((count + gridDim.x - 1) / gridDim.x + blockDim.x - 1) / blockDim.x * blockDim.x;
    const int64_t i0 = (int64_t)blockIdx.x * chunk;
    const int64_t i_end = (i0 + chunk < count) ? i0 + chunk : count;
    // (like the tile kernels: the sections of round i + 1 are requested before round i is worked on)
    SectionPair sp_n;
    float wl_n;
    bool act_n;
    auto request = [&](int64_t s) {
        const int64_t q = s + threadIdx.x;
        act_n = q < i_end;
        const int64_t r = first + (act_n ? q : 0);
        sp_n = load_section_pair(R, r, act_n);
        wl_n = act_n ? OT_STREAM_LOAD(&R.wl[r]) : 0.f;
    };
    request(i0);
    for (int64_t s = i0; s < i_end; s += blockDim.x) {  // whole workgroup iterates together (ballots below)
        const int64_t q = s + threadIdx.x;
        const bool active = act_n;
        const int64_t r = first + (active ? q : 0);
        const SectionPair sp = sp_n;
        const double wl = (double)wl_n;
        request(s + blockDim.x);
        const V3 sdir = pair_direction(sp);
        double xo = 0.0, yo = 0.0, zo = 0.0;
        bool have_obs = false;
#pragma unroll
        for (int d = 0; d < NDET; d++) {
            if (d >= n_det) continue;
            const auto& F = as_const(dets)[d];
            if (F.spread[0]) continue;
            V3 ph;
            float w = 0.f;
            bool valid = false, ill = false, to = false;
            // flat detector behind the last surface (the usual case): settled from the prefetched pair; the section search
            // only if a lane of the wave needs it
            bool settled = false;
            if (!GENERAL) settled = detector_hit_last(F, R.nt, active, sp, sdir, ph, w, valid);
            if (GENERAL || __ballot(!settled) != 0ull) {
                if (!settled) detector_hit<GENERAL, GENERAL>(R, r, active, F, sp, sdir, ph, w, valid, ill, to);
            }
            if (GENERAL) fuse_count_ill(F, ill, to);
            int32_t ix, iy;
            const int pix = valid ? fuse_pixel(F, ph, ix, iy) : -1;
            valid = valid && pix >= 0;
            if (valid && !have_obs) {
                observer_xyz_at(obs, wl, xo, yo, zo);
                have_obs = true;
            }
            const double wm = (double)w * F.a.ws;
            // (all lanes of the wave: hits of a wave that share a pixel are summed before they meet the LDS, ot_detector.hpp)
            wave_add4_by_key(valid, pix * OT_DET_MAX + d, xo * wm, yo * wm, zo * wm, 1.0 * wm,
                             [&](int key, double a0, double a1, double a2, double a3) {
                unsigned int h = ((unsigned int)key * 2654435761u) >> (32 - 11);  // OT_HASH_N = 2^11
                int slot = -1;
#pragma unroll
                for (int pr = 0; pr < OT_HASH_PROBES; pr++) {
                    const int sidx = (int)((h + pr) & (OT_HASH_N - 1));
                    int k = hkey[sidx];
                    if (k == OT_HASH_EMPTY) k = atomicCAS(&hkey[sidx], OT_HASH_EMPTY, key);
                    if (k == OT_HASH_EMPTY || k == key) {
                        slot = sidx;
                        break;
                    }
                }
                double* hv = (slot >= 0) ? &hval[slot * 4] : F.hist + (int64_t)(key / OT_DET_MAX) * 4;
                unsafeAtomicAdd(hv + 0, a0);
                unsafeAtomicAdd(hv + 1, a1);
                unsafeAtomicAdd(hv + 2, a2);
                unsafeAtomicAdd(hv + 3, a3);
            });
        }
    }
    __syncthreads();
    for (int sidx = threadIdx.x; sidx < OT_HASH_N; sidx += blockDim.x) {
        const int k = hkey[sidx];
        if (k != OT_HASH_EMPTY) {
            double* hg = dets[k % OT_DET_MAX].hist + (int64_t)(k / OT_DET_MAX) * 4;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const double v = hval[sidx * 4 + c];
                if (v != 0.0) unsafeAtomicAdd(hg + c, v);
            }
        }
    }
}

// ---- tile path, pass 1: hits -> records in per-tile chunk lists ------------------------------------------------
// LDS per (detector, tile) entry e: cnt[2] (records of the current / the previous sub-block), fill (records in the
// open chunk), cur (open chunk), nb (first chunk taken for the sub-block in flight).
//
// The kernel lives on loads in flight (56 B per ray, nothing else to do while they travel), so
//   * the sections of sub-block i + 1 are requested before sub-block i is processed, and
//   * the two workgroup barriers per sub-block are bare `s_barrier`s behind a wait for the LDS operations only:
//     __syncthreads() would also wait for every outstanding vector-memory operation, i.e. for the prefetch.
// Only LDS state is shared inside the workgroup; the record stores need no ordering among its waves.
OT_DEV void fuse_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// the open chunk of entry e takes the c records counted for it: fill / cur move on (nb = chunks taken for them)
OT_DEV void fuse_advance(unsigned int* fill, unsigned int* cur, const unsigned int* nb, int e, unsigned int c) {
    const unsigned int f = fill[e] + c;
    if (f > OT_FUSE_CH) {
        const unsigned int n_new = (f - 1) / OT_FUSE_CH;
        cur[e] = nb[e] + n_new - 1;
        fill[e] = f - n_new * OT_FUSE_CH;  // 1 .. CH
    } else {
        fill[e] = f;
    }
}

// extent of a workgroup's valid hits -> one of the slot tables (ordered integer atomics, as extent_flush does)
OT_DEV void spec_extent_flush(unsigned long long* slots, const double ext[4]) {
    __shared__ double sx[16][4];  // up to 1024 threads
    const double e[4] = {wave_min(ext[0]), wave_max(ext[1]), wave_min(ext[2]), wave_max(ext[3])};
    const int wave = threadIdx.x >> 6, n_waves = (blockDim.x + 63) >> 6;
    if (__lane_id() == 0)
        for (int c = 0; c < 4; c++) sx[wave][c] = e[c];
    __syncthreads();
    if (threadIdx.x < 4) {
        const int c = threadIdx.x;
        const double inf = __builtin_inf();
        double v = sx[0][c];
        for (int k = 1; k < n_waves; k++) v = (c & 1) ? fmax(v, sx[k][c]) : fmin(v, sx[k][c]);
        if (v == v && v != ((c & 1) ? -inf : inf)) {
            unsigned long long* dst = slots + 4 * (blockIdx.x % OT_EXT_SLOTS) + c;
            if (c & 1)
                atomicMax(dst, f64_to_ordered(v));
            else
                atomicMin(dst, f64_to_ordered(v));
        }
    }
}

// RPT rays per thread and sub-block (2 where the detectors' images have at most 1024 tiles: more loads in flight per
// barrier); R's pointers are advanced to the first ray of the range by the host, rays are addressed with 32 bits.
// LDS of the tile kernel: 6 KT + 10 counters, then (one detector) the staging area of a sub-block's records
__host__ __device__ static inline size_t fuse_stage_offset(int KT) {
    return (sizeof(unsigned int) * (6 * (size_t)KT + OT_DET_MAX + 2) + 15) / 16 * 16;
}
__host__ __device__ static inline size_t fuse_tiles_lds(int KT, int n_det, int rpt, bool specx) {
    const size_t counters = sizeof(unsigned int) * (5 * (size_t)(KT > 1 ? KT : 1) + OT_DET_MAX);
#ifndef OT_FUSE_SORT
    return counters;
#endif
    if (n_det != 1) return counters;
    return fuse_stage_offset(KT) + (size_t)OT_FUSE_BR * rpt * (specx ? 28 : 16);
}

// SPECX: the speculative-grid form (one detector, closed-form hit): the tile comes from SpecGrid instead of the image's
// pixel grid, the record is a SpecRec, and the extent of the valid hits is gathered on the way.
// PAIR: the storage has two sections (tail storage): detector_hit_pair settles every ray from the prefetched pair, the
// section search is not compiled in (with eight detectors unrolled it was most of the kernel: 11 700 instructions, 157 scalar
// registers spilled into vector lanes).
template <bool GENERAL, int NDET, int RPT, bool SPECX = false, bool PAIR = false>
__global__ __launch_bounds__(OT_FUSE_BR) void fuse_tiles_kernel(ot_rays R, uint32_t count, const FuseOne* __restrict__ dets,
                                                                int n_det, int KT, uint32_t piece) {
    // the probe's verdicts, read ONCE: a load of spread[0] inside the loop below is a vector-memory operation like the
    // prefetch of the next sub-block's sections, and the wait for it (vmcnt(0)) was a wait for that prefetch as well
    unsigned int spread_mask = 0u;
    for (int d = 0; d < n_det; d++)
        if (as_const(dets)[d].spread[0]) spread_mask |= 1u << d;
    spread_mask = (unsigned int)__builtin_amdgcn_readfirstlane((int)spread_mask);
    if (!spread_mask) return;
    constexpr uint32_t BRT = OT_FUSE_BR * RPT;        // rays per sub-block
    constexpr int TB = (RPT == 1) ? 11 : 10;          // bits of the tile number in a record key
    // SORT (one detector, -DOT_FUSE_SORT): the records of a sub-block are gathered tile by tile in LDS before they are stored,
    // so that neighbouring lanes write neighbouring records of one chunk instead of 64 different chunks per store
    // instruction.  Built because the record stores cost as much as everything else in this kernel together (C5, 9.3e7
    // records of 24 B: 2.10 ms; 1.05 ms without the stores, 1.62 ms with the same stores aimed at 400 KB per workgroup);
    // measured: no gain (C5 image 3.17 = 3.17 ms, C4 with a known extent 3.4 against 3.2 ms) -- a tile receives ~3 records
    // per sub-block, the runs stay shorter than a cache line.  Off by default.
#ifdef OT_FUSE_SORT
    constexpr bool SORT = (NDET == 1);
#else
    constexpr bool SORT = false;
#endif
    extern __shared__ unsigned int fl[];
    unsigned int* cnt = fl;  // [2][KT]
    unsigned int* fill = fl + 2 * KT;
    unsigned int* cur = fl + 3 * KT;
    unsigned int* nb = fl + 4 * KT;
    unsigned int* next = fl + 5 * KT;  // [n_det] next free chunk of this workgroup's part of each detector's pool
    unsigned int* sbase = next + OT_DET_MAX;  // SORT [KT]: first staging slot of a tile's records of this sub-block
    unsigned int* stop = sbase + KT;          // SORT [2]: staging slots taken (by sub-block parity)
    char* stg = (char*)fl + fuse_stage_offset(KT);
    double* sx = (double*)stg;                // SPECX: [BRT] x, [BRT] y
    double* sy = sx + BRT;
    float* sw = (float*)(SPECX ? stg + 16 * BRT : stg);  // [BRT] weight, [BRT] wavelength
    float* swl = sw + BRT;
    unsigned int* sdst = (unsigned int*)(swl + BRT);     // [BRT] record number in the pool, OT_FUSE_NONE: dropped
    unsigned int* spx = sdst + BRT;                      // plain records: [BRT] pixel in the tile
    if (threadIdx.x < (unsigned)n_det) next[threadIdx.x] = blockIdx.x * as_const(dets)[threadIdx.x].per_wg;
    if (SORT && threadIdx.x < 2) stop[threadIdx.x] = 0u;
    for (int e = threadIdx.x; e < KT; e += blockDim.x) {
        cnt[e] = 0u;
        cnt[KT + e] = 0u;
        fill[e] = OT_FUSE_CH;  // "full": the first record of a tile takes a chunk
        cur[e] = OT_FUSE_NONE;
        nb[e] = 0u;
    }
    __syncthreads();
    const uint32_t i0 = blockIdx.x * piece;
    const uint32_t i1 = (i0 + piece < count) ? i0 + piece : count;
    // The sections of sub-block i + 1 are requested at the top of round i.  (Tried: loads without a condition around them,
    // so that the compiler can count the operations in flight instead of waiting for all of them, and the wait for the
    // prefetch moved behind the second barrier -- no measurable difference in an interleaved A/B: the kernel is bound by its
    // scattered record stores, see below.)
    SectionPair sp_n[RPT];
    float wl_n[RPT];
    bool act_n[RPT];
    auto request = [&](uint32_t base) {
#pragma unroll
        for (int j = 0; j < RPT; j++) {
            const uint32_t q = base + j * OT_FUSE_BR + threadIdx.x;
            act_n[j] = q < i1;  // (count < 2^31: no wrap)
            sp_n[j] = load_section_pair(R, (int64_t)(act_n[j] ? q : 0u), act_n[j]);
            wl_n[j] = act_n[j] ? OT_STREAM_LOAD(&R.wl[q]) : 0.f;
        }
    };
    request(i0);
    const double inf = __builtin_inf();
    double ext[4] = {inf, -inf, inf, -inf};  // SPECX: x_min, x_max, y_min, y_max of this lane's valid hits
    int par = 0;
    for (uint32_t s = i0; s < i1; s += BRT, par ^= 1) {
        SectionPair sp[RPT];
        float wl[RPT];
        bool act[RPT];
#pragma unroll
        for (int j = 0; j < RPT; j++) {
            sp[j] = sp_n[j];
            wl[j] = wl_n[j];
            act[j] = act_n[j];
        }
        request(s + BRT);

        // The detector records are read through a pointer the optimiser cannot see through, once per sub-block:
        // otherwise lane constants derived from them are hoisted out of this loop and kept in vector registers.
        const FuseOne* dl = dets;
        asm volatile("" : "+s"(dl));
        unsigned int* cnt_a = cnt + par * KT;        // this sub-block
        unsigned int* cnt_b = cnt + (par ^ 1) * KT;  // the previous one (its records are written, fill / cur pending)
        // phase 1: hits, pixel, rank inside the tile's share of this sub-block
        float wk[RPT][NDET];
        unsigned int key[RPT][NDET];  // rank << (12 + TB) | tile << 12 | pixel in tile
        double hx[RPT], hy[RPT];      // SPECX: the hit itself
#pragma unroll
        for (int j = 0; j < RPT; j++) {
            const uint32_t q = s + j * OT_FUSE_BR + threadIdx.x;
            const int64_t r = (int64_t)(act[j] ? q : 0u);
            const V3 sdir = pair_direction(sp[j]);
#pragma unroll
            for (int d = 0; d < NDET; d++) {
                wk[j][d] = 0.f;
                key[j][d] = 0u;
                if (d >= n_det) continue;
                const auto& F = as_const(dl)[d];
                if (!((spread_mask >> d) & 1u)) continue;
                V3 ph;
                float w;
                bool valid = false, ill = false, to = false;
                // flat detector behind the last surface (the usual case): no section search; the wave takes the general
                // path only if one of its lanes needs it
                if constexpr (PAIR) {
                    detector_hit_pair(F, act[j], sp[j], sdir, ph, w, valid);
                } else {
                    bool settled = false;
                    if (!GENERAL) settled = detector_hit_last(F, R.nt, act[j], sp[j], sdir, ph, w, valid);
                    if (GENERAL || __ballot(!settled) != 0ull) {
                        if (!settled) detector_hit<GENERAL, GENERAL>(R, r, act[j], F, sp[j], sdir, ph, w, valid, ill, to);
                    }
                    if (GENERAL) fuse_count_ill(F, ill, to);
                }
                if (!valid) continue;
                unsigned int local, tile;
                if constexpr (SPECX) {
                    ext[0] = fmin(ext[0], ph.x), ext[1] = fmax(ext[1], ph.x);
                    ext[2] = fmin(ext[2], ph.y), ext[3] = fmax(ext[3], ph.y);
                    const double u = (ph.x - F.g.X0) * F.g.itw, v = (ph.y - F.g.Y0) * F.g.ith;
                    if (!(u >= 0.0 && u < (double)F.g.tx && v >= 0.0 && v < (double)F.g.ty)) {
                        // outside the provisional grid (the tail of a distribution the sample did not reach): kept aside
                        const unsigned int k = atomicAdd(F.g.esc_n, 1u);
                        if (k < F.g.esc_cap) F.g.esc[k] = {ph.x, ph.y, w, wl[j]};
                        continue;
                    }
                    local = 0u;
                    tile = (unsigned int)((int)v * F.g.tx + (int)u);
                    hx[j] = ph.x, hy[j] = ph.y;
                } else {
                    int32_t ix, iy;
                    if (fuse_pixel(F, ph, ix, iy) < 0) continue;
                    local = (unsigned int)(((iy & (OT_TILE_W - 1)) << 6) | (ix & (OT_TILE_W - 1)));
                    tile = (unsigned int)((iy >> 6) * F.tx + (ix >> 6));
                }
                const unsigned int rank = atomicAdd(&cnt_a[F.koff + (int)tile], 1u);
                wk[j][d] = w;
                key[j][d] = (rank << (12 + TB)) | (tile << 12) | local;
            }
        }
        fuse_lds_barrier();
        // phase 2: the previous sub-block's counts move the open chunks on; tiles whose open chunk overflows with this
        // sub-block's records take new chunks from the workgroup's part of their detector's pool
        if (SORT && threadIdx.x == 0) stop[par ^ 1] = 0u;  // (last read before this round's first barrier)
        for (int e = threadIdx.x; e < KT; e += blockDim.x) {
            const unsigned int cb = cnt_b[e];
            if (cb) {
                fuse_advance(fill, cur, nb, e, cb);
                cnt_b[e] = 0u;
            }
            const unsigned int c = cnt_a[e];
            if (!c) continue;
            if (SORT) sbase[e] = atomicAdd(&stop[par], c);  // the tile's records stand together, tiles in any order
            const unsigned int f = fill[e] + c;
            if (f <= OT_FUSE_CH) continue;
            int d = 0;
            while (d + 1 < n_det && e >= as_const(dl)[d + 1].koff) d++;
            const auto& F = as_const(dl)[d];
            const unsigned int n_new = (f - 1) / OT_FUSE_CH;  // ceil((f - CH) / CH)
            const unsigned int base = atomicAdd(&next[d], n_new);
            nb[e] = base;
            if (base + n_new > (blockIdx.x + 1) * F.per_wg) {
                F.overflow[0] = 1;
                nb[e] = F.cap;  // records of these chunks are dropped (phase 3 checks the chunk number)
            } else {
                for (unsigned int m = 0; m < n_new; m++) {
                    F.chunk_tile[base + m] = (unsigned int)(e - F.koff);
                    F.chunk_fill[base + m] = OT_FUSE_CH;  // every chunk but a tile's last one ends up full
                }
            }
        }
        fuse_lds_barrier();
        // phase 3: one 12-byte store per hit.  (The next phase 2 changes fill / cur / nb only behind the next barrier,
        // which every wave reaches after these reads.)  SORT: into the staging slots first.
#pragma unroll
        for (int j = 0; j < RPT; j++) {
#pragma unroll
            for (int d = 0; d < NDET; d++) {
                if (d >= n_det) continue;
                if (!(wk[j][d] > 0.f)) continue;
                const auto& F = as_const(dl)[d];
                const unsigned int local = key[j][d] & 0xfffu, tile = (key[j][d] >> 12) & ((1u << TB) - 1u),
                                   rank = key[j][d] >> (12 + TB);
                const int e = F.koff + (int)tile;
                unsigned int dest = fill[e] + rank, chunk;
                if (dest < OT_FUSE_CH) {
                    chunk = cur[e];
                } else {
                    dest -= OT_FUSE_CH;
                    chunk = nb[e] + dest / OT_FUSE_CH;
                    dest %= OT_FUSE_CH;
                }
                if constexpr (SORT) {
                    const unsigned int t = sbase[e] + rank;
                    sdst[t] = (chunk < F.cap) ? chunk * OT_FUSE_CH + dest : OT_FUSE_NONE;  // (pool < 2^32 records: host)
                    sw[t] = wk[j][d];
                    swl[t] = wl[j];
                    if constexpr (SPECX) {
                        sx[t] = hx[j];
                        sy[t] = hy[j];
                    } else {
                        spx[t] = local;
                    }
                } else if (chunk < F.cap) {
                    if constexpr (SPECX) {
                        SpecRec* dst = (SpecRec*)F.rec + ((size_t)chunk * OT_FUSE_CH + dest);
                        SpecRec rec = {hx[j], hy[j], wk[j][d], wl[j]};
                        *dst = rec;
                    } else {
                        TileRec rec = {wk[j][d], wl[j], local};
                        F.rec[(size_t)chunk * OT_FUSE_CH + dest] = rec;
                    }
                }
            }
        }
        if constexpr (SORT) {
            // phase 4: the staged records in slot order -- lanes next to each other write records next to each other
            fuse_lds_barrier();
            const auto& F = as_const(dl)[0];
            const unsigned int n_rec = stop[par];
            for (unsigned int t = threadIdx.x; t < n_rec; t += OT_FUSE_BR) {
                unsigned int dsti = sdst[t];
                if (dsti == OT_FUSE_NONE) continue;

                if constexpr (SPECX) {
                    SpecRec rec = {sx[t], sy[t], sw[t], swl[t]};
                    ((SpecRec*)F.rec)[dsti] = rec;
                } else {
                    TileRec rec = {sw[t], swl[t], spx[t]};
                    F.rec[dsti] = rec;
                }
            }
        }
    }
    __syncthreads();
    // the last sub-block's counts, then the open chunks: the only ones that are not full
    unsigned int* cnt_l = cnt + (par ^ 1) * KT;
    for (int e = threadIdx.x; e < KT; e += blockDim.x) {
        const unsigned int c = cnt_l[e];
        if (c) fuse_advance(fill, cur, nb, e, c);
        if (cur[e] == OT_FUSE_NONE) continue;
        int d = 0;
        while (d + 1 < n_det && e >= as_const(dets)[d + 1].koff) d++;
        const auto& F = as_const(dets)[d];
        if (cur[e] < F.cap) F.chunk_fill[cur[e]] = fill[e];
    }
    // chunks of this workgroup's parts that were never handed out
    for (int d = 0; d < n_det; d++) {
        const auto& F = as_const(dets)[d];
        if (!((spread_mask >> d) & 1u)) continue;
        const unsigned int end = (blockIdx.x + 1) * F.per_wg;
        for (unsigned int c = min(next[d], end) + threadIdx.x; c < end; c += blockDim.x) F.chunk_tile[c] = OT_FUSE_NONE;
    }
    if constexpr (SPECX) spec_extent_flush(as_const(dets)[0].g.ext_slots, ext);
}

// ---- tile path, pass 1 with line buffers (one detector, images of up to OT_LB_MAXK tiles) ---------------------------
// The record stores of fuse_tiles_kernel are half of its time (profiles/r3/tile_kernel_store_cost.txt): every store
// instruction touches 64 cache lines with 12 or 24 bytes each, the partial lines leave the L2 before the next sub-block
// adds to them, and the memory side pays for each visit.  Here a workgroup keeps, for every tile, the records of the
// chunk segment that is being filled (384 B: 16 or 32 records) in LDS and writes a segment when it is complete: 384
// contiguous bytes, three whole cache lines.  State per tile: fill / cur / nb as in fuse_tiles_kernel, and lo = the
// position in the open chunk from which records stand in LDS (positions below are in memory).  Per sub-block and tile with
// c new records at positions [fill, fill + c), seg_end = the end of the segment that holds position fill:
//   * positions below seg_end go to the LDS segment (slot = position mod segment length);
//   * if fill + c reaches seg_end the segment [lo, seg_end) is written out -- a "job", carried out by the threads that have
//     no tile to look after during phase 2 of the NEXT sub-block, so the round keeps its two barriers --; records beyond
//     seg_end bypass the buffer (straight to memory, as in fuse_tiles_kernel) and the buffer is empty afterwards (lo =
//     the new fill): a tile that receives more than a segment per sub-block (point-like images) simply streams.
// One 1024-thread workgroup per CU (the buffers of 361 tiles are 139 KB), two rays per thread and sub-block.
#define OT_LB_BR 1024
#ifndef OT_LB_RPT
#define OT_LB_RPT 2
#endif
#define OT_LB_MAXK 361
#define OT_LB_SEG_BYTES 384

__host__ __device__ static inline size_t fuse_lb_buf_offset(int K) { return (sizeof(unsigned int) * (10 * (size_t)K + 8) + 15) / 16 * 16; }
__host__ __device__ static inline size_t fuse_lb_lds(int K) { return fuse_lb_buf_offset(K) + (size_t)K * OT_LB_SEG_BYTES; }

template <bool SPECX>
__global__ __launch_bounds__(OT_LB_BR) void fuse_tiles_lb_kernel(ot_rays R, uint32_t count, const FuseOne* __restrict__ dets,
                                                                 int K, uint32_t piece) {
    using REC = typename std::conditional<SPECX, SpecRec, TileRec>::type;
    constexpr unsigned int SEG = OT_LB_SEG_BYTES / sizeof(REC);  // records per segment: 16 (24 B) or 32 (12 B)
    constexpr int RPT = OT_LB_RPT;
    constexpr uint32_t BRT = OT_LB_BR * RPT;
    static_assert(OT_FUSE_CH % SEG == 0, "segments must not straddle chunks");
    const auto& F = as_const(dets)[0];
    if (!F.spread[0]) return;
    extern __shared__ unsigned int fl[];
    unsigned int* cnt = fl;  // [2][K]
    unsigned int* fill = fl + 2 * K;
    unsigned int* cur = fl + 3 * K;
    unsigned int* nb = fl + 4 * K;
    unsigned int* lo = fl + 5 * K;
    unsigned int* job = fl + 6 * K;    // [2][K] by sub-block parity: tile | first slot << 11 | records << 22
    unsigned int* jdst = fl + 8 * K;   // [2][K] record number of the job's first record
    unsigned int* next = fl + 10 * K;  // [1] next free chunk of this workgroup's part of the pool
    unsigned int* jn = next + 1;       // [2] jobs of a sub-block (by parity)
    REC* buf = (REC*)((char*)fl + fuse_lb_buf_offset(K));  // [K][SEG]
    if (threadIdx.x == 0) {
        next[0] = blockIdx.x * F.per_wg;
        jn[0] = jn[1] = 0u;
    }
    for (int e = threadIdx.x; e < K; e += blockDim.x) {
        cnt[e] = 0u;
        cnt[K + e] = 0u;
        fill[e] = OT_FUSE_CH;  // "full": the first record of a tile takes a chunk
        cur[e] = OT_FUSE_NONE;
        nb[e] = 0u;
        lo[e] = OT_FUSE_CH;    // nothing in the buffer
    }
    __syncthreads();
    const uint32_t i0 = blockIdx.x * piece;
    const uint32_t i1 = (i0 + piece < count) ? i0 + piece : count;
    // position p of a tile's record stream (relative to its open chunk; >= CH: in the chunks taken this round) -> record number
    auto place = [&](int e, unsigned int p) -> unsigned int {
        unsigned int chunk;
        if (p < OT_FUSE_CH) {
            chunk = cur[e];
        } else {
            p -= OT_FUSE_CH;
            chunk = nb[e] + p / OT_FUSE_CH;
            p %= OT_FUSE_CH;
        }
        return chunk < F.cap ? chunk * OT_FUSE_CH + p : OT_FUSE_NONE;  // (pool < 2^32 records: host)
    };
    // the previous sub-block's c records join the open chunk: fill / cur move on, and lo with them
    auto advance = [&](int e, unsigned int c) {
        const unsigned int f_old = fill[e];
        const bool completed = f_old + c >= (f_old / SEG + 1) * SEG;  // its segment was written out, the rest bypassed the buffer
        fuse_advance(fill, cur, nb, e, c);
        const unsigned int l = lo[e];
        lo[e] = completed ? fill[e] : (l >= OT_FUSE_CH ? l - OT_FUSE_CH : l);
    };
    // the jobs of one sub-block: SEG lanes per job, neighbouring lanes write neighbouring records -- whole cache lines
    auto run_jobs = [&](int parity, unsigned int first_thread) {
        if (threadIdx.x < first_thread) return;
        const unsigned int n_jobs = jn[parity];
        const unsigned int t = threadIdx.x - first_thread, groups = (OT_LB_BR - first_thread) / SEG;
        const unsigned int g = t / SEG, r = t % SEG;
        if (g >= groups) return;
        for (unsigned int j = g; j < n_jobs; j += groups) {
            const unsigned int jb = job[parity * K + j], dst = jdst[parity * K + j];
            const unsigned int e = jb & 0x7ffu, slot = (jb >> 11) & 0x7ffu, n = jb >> 22;
            if (r < n && dst != OT_FUSE_NONE) ((REC*)F.rec)[dst + r] = buf[(size_t)e * SEG + slot + r];
        }
    };
    SectionPair sp_n[RPT];
    float wl_n[RPT];
    bool act_n[RPT];
    auto request = [&](uint32_t base) {
#pragma unroll
        for (int j = 0; j < RPT; j++) {
            const uint32_t q = base + j * OT_LB_BR + threadIdx.x;
            act_n[j] = q < i1;
            sp_n[j] = load_section_pair(R, (int64_t)(act_n[j] ? q : 0u), act_n[j]);
            wl_n[j] = act_n[j] ? OT_STREAM_LOAD(&R.wl[q]) : 0.f;
        }
    };
    request(i0);
    const double inf = __builtin_inf();
    double ext[4] = {inf, -inf, inf, -inf};
    const unsigned int k_threads = ((unsigned int)K + 63u) & ~63u;  // phase 2 occupies the first waves, the jobs the others
    int par = 0;
    for (uint32_t s = i0; s < i1; s += BRT, par ^= 1) {
        SectionPair sp[RPT];
        float wl[RPT];
        bool act[RPT];
#pragma unroll
        for (int j = 0; j < RPT; j++) {
            sp[j] = sp_n[j];
            wl[j] = wl_n[j];
            act[j] = act_n[j];
        }
        request(s + BRT);
        unsigned int* cnt_a = cnt + par * K;
        unsigned int* cnt_b = cnt + (par ^ 1) * K;
        // phase 1: hit, tile, rank inside the tile's share of this sub-block
        float wk[RPT];
        unsigned int tile[RPT], rank[RPT], local[RPT];
        double hx[RPT], hy[RPT];
#pragma unroll
        for (int j = 0; j < RPT; j++) {
            wk[j] = 0.f;
            tile[j] = rank[j] = local[j] = 0u;
            hx[j] = hy[j] = 0.0;
            const uint32_t q = s + j * OT_LB_BR + threadIdx.x;
            const int64_t r = (int64_t)(act[j] ? q : 0u);
            const V3 sdir = pair_direction(sp[j]);
            V3 ph;
            float w;
            bool valid = false, ill = false, to = false;
            const bool settled = detector_hit_last(F, R.nt, act[j], sp[j], sdir, ph, w, valid);
            if (__ballot(!settled) != 0ull) {
                if (!settled) detector_hit<false, false>(R, r, act[j], F, sp[j], sdir, ph, w, valid, ill, to);
            }
            if (!valid) continue;
            if constexpr (SPECX) {
                ext[0] = fmin(ext[0], ph.x), ext[1] = fmax(ext[1], ph.x);
                ext[2] = fmin(ext[2], ph.y), ext[3] = fmax(ext[3], ph.y);
                const double u = (ph.x - F.g.X0) * F.g.itw, v = (ph.y - F.g.Y0) * F.g.ith;
                if (!(u >= 0.0 && u < (double)F.g.tx && v >= 0.0 && v < (double)F.g.ty)) {
                    const unsigned int k = atomicAdd(F.g.esc_n, 1u);
                    if (k < F.g.esc_cap) F.g.esc[k] = {ph.x, ph.y, w, wl[j]};
                    continue;
                }
                tile[j] = (unsigned int)((int)v * F.g.tx + (int)u);
                hx[j] = ph.x, hy[j] = ph.y;
            } else {
                int32_t ix, iy;
                if (fuse_pixel(F, ph, ix, iy) < 0) continue;
                local[j] = (unsigned int)(((iy & (OT_TILE_W - 1)) << 6) | (ix & (OT_TILE_W - 1)));
                tile[j] = (unsigned int)((iy >> 6) * F.tx + (ix >> 6));
            }
            rank[j] = atomicAdd(&cnt_a[tile[j]], 1u);
            wk[j] = w;
        }
        fuse_lds_barrier();
        // phase 2, one thread per tile: the previous sub-block's counts move the open chunks on; chunks for this sub-block's
        // records; a job for every segment they complete.  Meanwhile the other threads carry out the previous sub-block's jobs
        // (their destinations were settled when they were made; the slots they read are written again only behind the barrier).
        run_jobs(par ^ 1, k_threads);
        for (int e = threadIdx.x; e < K; e += blockDim.x) {
            const unsigned int cb = cnt_b[e];
            if (cb) {
                advance(e, cb);
                cnt_b[e] = 0u;
            }
            const unsigned int c = cnt_a[e];
            if (!c) continue;
            const unsigned int f0 = fill[e], f = f0 + c;
            if (f > OT_FUSE_CH) {
                const unsigned int n_new = (f - 1) / OT_FUSE_CH;
                const unsigned int base = atomicAdd(&next[0], n_new);
                nb[e] = base;
                if (base + n_new > (blockIdx.x + 1) * F.per_wg) {
                    F.overflow[0] = 1;
                    nb[e] = F.cap;  // records of these chunks are dropped (place() checks the chunk number)
                } else {
                    for (unsigned int m = 0; m < n_new; m++) {
                        F.chunk_tile[base + m] = (unsigned int)e;
                        F.chunk_fill[base + m] = OT_FUSE_CH;  // every chunk but a tile's last one ends up full
                    }
                }
            }
            const unsigned int seg_end = (f0 / SEG + 1) * SEG;
            if (f >= seg_end) {
                const unsigned int l = lo[e] < f0 ? lo[e] : f0;  // (an empty buffer has lo = f0)
                const unsigned int j = atomicAdd(&jn[par], 1u);
                job[par * K + j] = (unsigned int)e | ((l % SEG) << 11) | ((seg_end - l) << 22);
                jdst[par * K + j] = place(e, l);  // the segment lies in one chunk
            }
        }
        fuse_lds_barrier();
        if (threadIdx.x == 0) jn[par ^ 1] = 0u;  // its jobs are done; next written behind the next round's first barrier
        // phase 3: a record below its tile's segment end goes to the LDS segment, the others straight to memory
#pragma unroll
        for (int j = 0; j < RPT; j++) {
            if (!(wk[j] > 0.f)) continue;
            const int e = (int)tile[j];
            const unsigned int f0 = fill[e], p = f0 + rank[j];
            REC rec;
            if constexpr (SPECX) rec = {hx[j], hy[j], wk[j], wl[j]}; else rec = {wk[j], wl[j], local[j]};
            if (p < (f0 / SEG + 1) * SEG) {
                buf[(size_t)e * SEG + p % SEG] = rec;
            } else {
                const unsigned int dst = place(e, p);
                if (dst != OT_FUSE_NONE) ((REC*)F.rec)[dst] = rec;
            }
        }
    }
    __syncthreads();
    // the last sub-block's jobs and counts; what still stands in the buffers; the open chunks: the only ones that are not full
    run_jobs(par ^ 1, 0u);
    unsigned int* cnt_l = cnt + (par ^ 1) * K;
    __syncthreads();
    for (int e = threadIdx.x; e < K; e += blockDim.x) {
        const unsigned int c = cnt_l[e];
        if (c) advance(e, c);
    }
    __syncthreads();
    {
        const unsigned int g = threadIdx.x / SEG, r = threadIdx.x % SEG;
        for (int e = (int)g; e < K; e += OT_LB_BR / SEG) {
            const unsigned int l = lo[e], f = fill[e];
            if (cur[e] == OT_FUSE_NONE || l >= f) continue;
            const unsigned int p = l + r;  // (l and f lie in one segment)
            if (p < f && cur[e] < F.cap) ((REC*)F.rec)[cur[e] * OT_FUSE_CH + p] = buf[(size_t)e * SEG + p % SEG];
        }
    }
    for (int e = threadIdx.x; e < K; e += blockDim.x) {
        if (cur[e] == OT_FUSE_NONE) continue;
        if (cur[e] < F.cap) F.chunk_fill[cur[e]] = fill[e];
    }
    // chunks of this workgroup's part that were never handed out
    {
        const unsigned int end = (blockIdx.x + 1) * F.per_wg;
        for (unsigned int c = min(next[0], end) + threadIdx.x; c < end; c += blockDim.x) F.chunk_tile[c] = OT_FUSE_NONE;
    }
    if constexpr (SPECX) spec_extent_flush(F.g.ext_slots, ext);
}

// ---- tile path, pass 2: chunks grouped by tile ------------------------------------------------------------------
struct FuseIndex {
    unsigned int* tile_n;   // [K] chunks of each tile, later the placement cursor
    unsigned int* tstart;   // [K + 1]
    unsigned int* wstart;   // [K + 1] first accumulation workgroup (= slab) of each tile: ceil(chunks / OT_FUSE_CPW) each
    unsigned int* list;     // [cap] chunk numbers grouped by tile
    double* slabs;          // [n_slabs][TILE_PX * 4]
    unsigned int n_slabs;   // workgroups of the accumulation launch: cap / OT_FUSE_CPW + K bounds the number needed
};

// chunks per tile: LDS histogram per workgroup (16 chunk numbers per thread), one global add per tile and workgroup
#define OT_FUSE_IDX_PER 16
OT_DEV void fuse_chunk_hist_body(const FuseOne& F, const FuseIndex& ix, const unsigned int bx, const unsigned int by) {
    if (!F.spread[0]) return;
    const unsigned int n = F.cap;
    const unsigned int c0 = bx * (1024 * OT_FUSE_IDX_PER);
    if (c0 >= n) return;
    __shared__ unsigned int h[OT_TILE_MAX];
    for (int i = threadIdx.x; i < F.K; i += blockDim.x) h[i] = 0u;
    __syncthreads();
    for (int k = 0; k < OT_FUSE_IDX_PER; k++) {
        const unsigned int c = c0 + k * 1024 + threadIdx.x;
        if (c < n) {
            const unsigned int t = F.chunk_tile[c];
            if (t != OT_FUSE_NONE) atomicAdd(&h[t], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < F.K; i += blockDim.x)
        if (h[i]) atomicAdd(&ix.tile_n[i], h[i]);
}
__global__ __launch_bounds__(1024) void fuse_chunk_hist_kernel(FuseOne F, FuseIndex ix) {
    fuse_chunk_hist_body(F, ix, blockIdx.x, blockIdx.y);
}
// several detectors in one launch (blockIdx.z), their records in device memory
__global__ __launch_bounds__(1024) void fuse_chunk_hist_multi_kernel(const FuseOne* __restrict__ dets, const FuseIndex* __restrict__ ixs) {
    fuse_chunk_hist_body(dets[blockIdx.z], ixs[blockIdx.z], blockIdx.x, blockIdx.y);
}

OT_DEV void fuse_chunk_scan_body(const FuseOne& F, const FuseIndex& ix, const unsigned int bx, const unsigned int by) {
    if (!F.spread[0]) return;
    // exclusive scans of the chunk counts (tstart) and of the workgroups they take (wstart) over K <= 2048 tiles: two
    // tiles per thread, wave scans by shuffles, the 16 wave totals through LDS
    __shared__ unsigned int wsum[2][16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int i0 = 2 * t, i1 = 2 * t + 1;
    const unsigned int n0 = i0 < F.K ? ix.tile_n[i0] : 0u, n1 = i1 < F.K ? ix.tile_n[i1] : 0u;
    const unsigned int g0 = (n0 + OT_FUSE_CPW - 1) / OT_FUSE_CPW, g1 = (n1 + OT_FUSE_CPW - 1) / OT_FUSE_CPW;
    unsigned int a = n0 + n1, w = g0 + g1;  // inclusive scans over the threads
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned int ua = __shfl_up(a, o), uw = __shfl_up(w, o);
        if (lane >= o) a += ua, w += uw;
    }
    if (lane == 63) wsum[0][wave] = a, wsum[1][wave] = w;
    __syncthreads();
    unsigned int base_a = 0, base_w = 0;
    for (int k = 0; k < wave; k++) base_a += wsum[0][k], base_w += wsum[1][k];
    const unsigned int ea = base_a + a - (n0 + n1), ew = base_w + w - (g0 + g1);  // exclusive, at tile i0
    if (i0 < F.K) ix.tstart[i0] = ea, ix.wstart[i0] = ew;
    if (i1 < F.K) ix.tstart[i1] = ea + n0, ix.wstart[i1] = ew + g0;
    if (i0 == F.K - 1 || i1 == F.K - 1) ix.tstart[F.K] = base_a + a, ix.wstart[F.K] = base_w + w;  // the totals
    __syncthreads();  // (tile_n is read above by its own thread only; the barrier keeps the kernel's phases apart)
    if (i0 < F.K) ix.tile_n[i0] = 0u;  // becomes the placement cursor
    if (i1 < F.K) ix.tile_n[i1] = 0u;
}
__global__ __launch_bounds__(1024) void fuse_chunk_scan_kernel(FuseOne F, FuseIndex ix) {
    fuse_chunk_scan_body(F, ix, blockIdx.x, blockIdx.y);
}
// several detectors in one launch (blockIdx.z), their records in device memory
__global__ __launch_bounds__(1024) void fuse_chunk_scan_multi_kernel(const FuseOne* __restrict__ dets, const FuseIndex* __restrict__ ixs) {
    fuse_chunk_scan_body(dets[blockIdx.z], ixs[blockIdx.z], blockIdx.x, blockIdx.y);
}

// the workgroup's chunks of a tile get consecutive places behind one global reservation per tile
OT_DEV void fuse_chunk_place_body(const FuseOne& F, const FuseIndex& ix, const unsigned int bx, const unsigned int by) {
    if (!F.spread[0]) return;
    const unsigned int n = F.cap;
    const unsigned int c0 = bx * (1024 * OT_FUSE_IDX_PER);
    if (c0 >= n) return;
    __shared__ unsigned int h[OT_TILE_MAX];
    for (int i = threadIdx.x; i < F.K; i += blockDim.x) h[i] = 0u;
    __syncthreads();
    unsigned int rank[OT_FUSE_IDX_PER], tl[OT_FUSE_IDX_PER];
#pragma unroll
    for (int k = 0; k < OT_FUSE_IDX_PER; k++) {
        const unsigned int c = c0 + k * 1024 + threadIdx.x;
        tl[k] = OT_FUSE_NONE;
        rank[k] = 0u;
        if (c < n) {
            tl[k] = F.chunk_tile[c];
            if (tl[k] != OT_FUSE_NONE) rank[k] = atomicAdd(&h[tl[k]], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < F.K; i += blockDim.x)
        if (h[i]) h[i] = ix.tstart[i] + atomicAdd(&ix.tile_n[i], h[i]);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < OT_FUSE_IDX_PER; k++)
        if (tl[k] != OT_FUSE_NONE) ix.list[h[tl[k]] + rank[k]] = c0 + k * 1024 + threadIdx.x;
}
__global__ __launch_bounds__(1024) void fuse_chunk_place_kernel(FuseOne F, FuseIndex ix) {
    fuse_chunk_place_body(F, ix, blockIdx.x, blockIdx.y);
}
// several detectors in one launch (blockIdx.z), their records in device memory
__global__ __launch_bounds__(1024) void fuse_chunk_place_multi_kernel(const FuseOne* __restrict__ dets, const FuseIndex* __restrict__ ixs) {
    fuse_chunk_place_body(dets[blockIdx.z], ixs[blockIdx.z], blockIdx.x, blockIdx.y);
}

// Accumulation, one workgroup per OT_FUSE_CPW chunks of a tile (wstart): the hits of an image are rarely spread evenly
// -- C4's picture covers a fifth of the detector, 42 of 225 tiles hold every record -- and a fixed number of workgroups
// per tile left most CUs idle behind the few heavy tiles (1.7 of 4 waves per SIMD resident on average).  Workgroup b
// finds its tile by bisection of wstart in LDS, adds its chunks into an LDS tile and writes slab b.
OT_DEV void fuse_accum_body(const FuseOne& F, const FuseIndex& ix, const double* __restrict__ table, const unsigned int bx, const unsigned int stride) {
    if (!F.spread[0]) return;
    extern __shared__ double lds[];  // [TILE_PX * 4 tile] [471 * 6 observer table: (value, difference) pairs]; the tile part first holds wstart
    double* tile = lds;
    double* obs = lds + OT_TILE_PX * 4;
    unsigned int* ws = (unsigned int*)lds;
    const int K = F.K;
    // The grid holds as many workgroups as can be resident (one per CU: the tile takes the LDS); workgroup bx takes the slabs
    // bx, bx + stride, ...  (One workgroup per slab of the WORST case -- every chunk of the pool in use -- meant thousands of
    // launches that found nothing to do: 0.1-0.15 ms per pass of C4, and most of the short last chunk's 0.17 ms.)
    const unsigned int total = ix.wstart[K];
    if (bx >= total) return;
    for (int i = threadIdx.x; i < OT_OBS_N * 6; i += blockDim.x) obs[i] = table[OT_OBS6_OFF + i];
    for (unsigned int b = bx; b < total; b += stride) {
        for (int i = threadIdx.x; i <= K; i += blockDim.x) ws[i] = ix.wstart[i];
        __syncthreads();
        int lo = 0, hi = K;  // ws[lo] <= b < ws[hi]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (ws[mid] <= b) lo = mid; else hi = mid;
        }
        const int t = lo;
        const unsigned int part = b - ws[t];
        __syncthreads();  // everyone has read ws: the tile may be cleared
        const unsigned int c0 = ix.tstart[t], n_t = ix.tstart[t + 1] - c0;
        const unsigned int j_begin = part * OT_FUSE_CPW;
        const unsigned int j_end = (j_begin + OT_FUSE_CPW < n_t) ? j_begin + OT_FUSE_CPW : n_t;
        for (int i = threadIdx.x; i < OT_TILE_PX * 4; i += blockDim.x) tile[i] = 0.0;
        __shared__ unsigned int s_chunk[OT_FUSE_CPW], s_fill[OT_FUSE_CPW];
        for (unsigned int i = threadIdx.x; i < j_end - j_begin; i += blockDim.x) {
            const unsigned int c = ix.list[c0 + j_begin + i];
            s_chunk[i] = c;
            s_fill[i] = F.chunk_fill[c];
        }
        __syncthreads();
        // The workgroup's chunk numbers and fill counts go to LDS first (list -> fill -> record used to be three dependent
        // round trips per round), and the records of round i + 1 are requested before those of round i are added.
        constexpr int PER = 1024 / OT_FUSE_CH;  // chunks a workgroup handles at once
        constexpr int DEPTH = 4;                // chunk rounds per stage; two stages in flight per thread
        const int g = threadIdx.x / OT_FUSE_CH, slot = threadIdx.x % OT_FUSE_CH;
        const int n_c = (int)(j_end - j_begin);
        const TileRec* __restrict__ recs = F.rec;
        auto load = [&](int first, TileRec* rec, bool* ok) {
    #pragma unroll
            for (int k = 0; k < DEPTH; k++) {
                const int i = first + k * PER, ic = i < n_c ? i : 0;
                ok[k] = i < n_c && (unsigned int)slot < s_fill[ic];
                if (ok[k]) rec[k] = recs[(size_t)s_chunk[ic] * OT_FUSE_CH + slot];
            }
        };
        auto add = [&](const TileRec* rec, const bool* ok) {
    #pragma unroll
            for (int k = 0; k < DEPTH; k++) {
                if (!ok[k]) continue;
                const double wm = (double)rec[k].w * F.a.ws;
                double xo, yo, zo;
                observer_xyz_at6(obs, (double)rec[k].wl, xo, yo, zo);
                // plane-major tile [channel][pixel]: the lanes of one add then spread over 16 bank pairs; pixel-major
                // (4 doubles per pixel) would leave them 4 and make every add a 16-way bank conflict
                double* hv = tile + (int)rec[k].px;
                unsafeAtomicAdd(hv + 0 * OT_TILE_PX, xo * wm);
                unsafeAtomicAdd(hv + 1 * OT_TILE_PX, yo * wm);
                unsafeAtomicAdd(hv + 2 * OT_TILE_PX, zo * wm);
                unsafeAtomicAdd(hv + 3 * OT_TILE_PX, 1.0 * wm);
            }
        };
        TileRec ra[DEPTH], rb[DEPTH];
        bool oa[DEPTH], ob[DEPTH];
        load(g, ra, oa);
        for (int i0 = g; i0 < n_c; i0 += 2 * PER * DEPTH) {
            load(i0 + PER * DEPTH, rb, ob);
            add(ra, oa);
            load(i0 + 2 * PER * DEPTH, ra, oa);
            add(rb, ob);
        }
        __syncthreads();
        double* slab = ix.slabs + (size_t)b * (OT_TILE_PX * 4);
        for (int i = threadIdx.x; i < OT_TILE_PX * 4; i += blockDim.x) slab[i] = tile[i];
        __syncthreads();  // the slab is out: the tile part holds wstart again in the next round
    }
}
__global__ __launch_bounds__(1024) void fuse_accum_kernel(FuseOne F, FuseIndex ix, const double* __restrict__ table) {
    fuse_accum_body(F, ix, table, blockIdx.x, gridDim.x);
}
// several detectors in one launch (blockIdx.z), their records in device memory
__global__ __launch_bounds__(1024) void fuse_accum_multi_kernel(const FuseOne* __restrict__ dets, const FuseIndex* __restrict__ ixs, const double* __restrict__ table) {
    fuse_accum_body(dets[blockIdx.z], ixs[blockIdx.z], table, blockIdx.x, gridDim.x);
}

// grid (16, K): thread = one pixel of tile blockIdx.y, all four planes
OT_DEV void fuse_reduce_body(const FuseOne& F, const FuseIndex& ix, const unsigned int bx, const unsigned int by) {
    if (!F.spread[0]) return;
    const int tl = by;
    if (tl >= F.K) return;  // (a launch for several detectors has the tile count of the largest image)
    const unsigned int s_first = ix.wstart[tl], s_end = ix.wstart[tl + 1];
    if (s_end == s_first) return;
    const int local = bx * blockDim.x + threadIdx.x;
    const int px = (tl % F.tx) * OT_TILE_W + (local & (OT_TILE_W - 1));
    const int py = (tl / F.tx) * OT_TILE_W + (local >> 6);
    if (px >= F.a.Nx || py >= F.a.Ny) return;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    for (unsigned int s = s_first; s < s_end; s++) {
        const double* sl = ix.slabs + (size_t)s * (OT_TILE_PX * 4) + local;  // plane-major slab
        s0 += sl[0 * OT_TILE_PX];
        s1 += sl[1 * OT_TILE_PX];
        s2 += sl[2 * OT_TILE_PX];
        s3 += sl[3 * OT_TILE_PX];
    }
    double* hg = F.hist + ((int64_t)py * F.a.Nx + px) * 4;
    hg[0] += s0;
    hg[1] += s1;
    hg[2] += s2;
    hg[3] += s3;
}
__global__ __launch_bounds__(256) void fuse_reduce_kernel(FuseOne F, FuseIndex ix) {
    fuse_reduce_body(F, ix, blockIdx.x, blockIdx.y);
}
// several detectors in one launch (blockIdx.z), their records in device memory
__global__ __launch_bounds__(256) void fuse_reduce_multi_kernel(const FuseOne* __restrict__ dets, const FuseIndex* __restrict__ ixs) {
    fuse_reduce_body(dets[blockIdx.z], ixs[blockIdx.z], blockIdx.x, blockIdx.y);
}

// ---- automatic extent in one pass over the ray sections ---------------------------------------------------------
// Raytracer.detector_image(extent=None) (raytracer.py:1042-1049): the image extent is the bounding box of the hits, so
// the pixel a hit falls into is known only after the last hit.  The chain "hit list, then binning" writes every valid
// hit (24 B) and reads the list three more times; an extent-only pass in front of the fused kernels reads the sections
// twice.  Here the sections are read ONCE:
//
//   sample   extent E0 of the hits of every 128th wave of rays (~1 % of the bytes).  E0 lies inside the final extent E,
//            so E's pixels are at least as large as those of E0's own image (up to the ratio snap of
//            RenderImage._pixel_counts, which the host anticipates)
//   tiles    fuse_tiles_kernel<SPECX>: hit -> tile of a PROVISIONAL grid (tiles of ~60 E0-pixels, laid over E0 plus a
//            margin) -> 24-byte record (x, y, w, wl) appended to the tile's chunk list; the exact extent E of all hits
//            is gathered on the way; hits outside the grid go to a short list of their own         reads 56, writes 24 B
//   (host)   E -> RenderImage.__fix_extent, pixel counts, histogram
//   accum    per tile: a provisional tile covers at most ~62 x 62 pixels of the final grid, so its records are binned
//            with the exact rule (hit_pixel on the f64 position) into a 64 x 64 LDS window whose origin is the pixel
//            of the tile's corner; windows of neighbouring tiles overlap, the reduction adds them with atomics.  A
//            record outside its window (cannot happen while the host's check of the tile span holds) and the hits of
//            the escape list are added straight to the image.                                      reads 24 B
//
// 104 B per ray instead of 136; same pixels and sums as the chain (the sums in another order).
OT_DEV void spec_origin(const FuseOne& F, int t, int& ox, int& oy) {
    const int tcx = t % F.g.tx, tcy = t / F.g.tx;
    // one pixel of slack: membership in a tile was decided by floor((x - X0) / tw), its corner here is X0 + tcx * tw
    ox = (int)floor(F.a.fx * (F.g.X0 + (double)tcx * F.g.tw - F.a.x0)) - 1;
    oy = (int)floor(F.a.fy * (F.g.Y0 + (double)tcy * F.g.th - F.a.y0)) - 1;
}

// extent of the hits of a sample of the rays: wave k of the launch takes rays [64 k stride, 64 k stride + 64)
__global__ __launch_bounds__(256) void spec_sample_kernel(ot_rays R, uint32_t count, const FuseOne* __restrict__ dets,
                                                          uint32_t stride) {
    const auto& F = as_const(dets)[0];
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t q = (uint64_t)(t >> 6) * 64u * stride + (t & 63u);
    const bool active = q < count;
    const SectionPair sp = load_section_pair(R, (int64_t)(active ? q : 0u), active);
    const V3 sdir = pair_direction(sp);
    V3 ph;
    float w;
    bool valid, ill = false, to = false;
    const bool settled = detector_hit_last(F, R.nt, active, sp, sdir, ph, w, valid);
    if (__ballot(!settled) != 0ull) {
        if (!settled) detector_hit<false, false>(R, (int64_t)(active ? q : 0u), active, F, sp, sdir, ph, w, valid, ill, to);
    }
    const double inf = __builtin_inf();
    const double ext[4] = {valid ? ph.x : inf, valid ? ph.x : -inf, valid ? ph.y : inf, valid ? ph.y : -inf};
    spec_extent_flush(F.g.ext_slots, ext);
}

// out[0..3] = extent of the slot tables (+-inf where no hit); with esc_n: out[4] = hits that asked for the escape list
__global__ __launch_bounds__(64) void spec_result_kernel(const unsigned long long* __restrict__ slots,
                                                         const unsigned int* __restrict__ esc_n, unsigned int esc_cap,
                                                         double* __restrict__ out) {
    const int k = threadIdx.x;  // one slot table per lane (OT_EXT_SLOTS = 64), then a wave reduction
    const double inf = __builtin_inf();
    double e[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const unsigned long long u = slots[4 * k + c];
        const bool untouched = u == ((c & 1) ? 0ull : ~0ull);
        e[c] = untouched ? ((c & 1) ? -inf : inf) : ordered_to_f64(u);
    }
    e[0] = wave_min(e[0]), e[1] = wave_max(e[1]), e[2] = wave_min(e[2]), e[3] = wave_max(e[3]);
    if (k < 4) out[k] = e[k];
    if (k == 4 && esc_n) out[4] = (double)esc_n[0];
    if (k == 5 && esc_n) out[5] = (double)esc_cap;  // (on the device: result6 may be device memory)
}

// fuse_accum_kernel for SpecRec chunks: LDS window of tile t = final pixels [ox, ox + 64) x [oy, oy + 64)
__global__ __launch_bounds__(1024) void spec_accum_kernel(FuseOne F, FuseIndex ix, const double* __restrict__ table) {
    extern __shared__ double lds[];  // [TILE_PX * 4 window] [471 * 6 observer table: (value, difference) pairs]; the window part first holds wstart
    double* tile = lds;
    double* obs = lds + OT_TILE_PX * 4;
    unsigned int* ws = (unsigned int*)lds;
    const int K = F.K;
    const unsigned int total = ix.wstart[K];  // (resident workgroups take the slabs in turn, see fuse_accum_body)
    if (blockIdx.x >= total) return;
    for (int i = threadIdx.x; i < OT_OBS_N * 6; i += blockDim.x) obs[i] = table[OT_OBS6_OFF + i];
    for (unsigned int b = blockIdx.x; b < total; b += gridDim.x) {
        for (int i = threadIdx.x; i <= K; i += blockDim.x) ws[i] = ix.wstart[i];
        __syncthreads();
        int lo = 0, hi = K;  // ws[lo] <= b < ws[hi]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (ws[mid] <= b) lo = mid; else hi = mid;
        }
        const int t = lo;
        const unsigned int part = b - ws[t];
        __syncthreads();  // everyone has read ws: the window may be cleared
        const unsigned int c0 = ix.tstart[t], n_t = ix.tstart[t + 1] - c0;
        const unsigned int j_begin = part * OT_FUSE_CPW;
        const unsigned int j_end = (j_begin + OT_FUSE_CPW < n_t) ? j_begin + OT_FUSE_CPW : n_t;
        for (int i = threadIdx.x; i < OT_TILE_PX * 4; i += blockDim.x) tile[i] = 0.0;
        __shared__ unsigned int s_chunk[OT_FUSE_CPW], s_fill[OT_FUSE_CPW];
        for (unsigned int i = threadIdx.x; i < j_end - j_begin; i += blockDim.x) {
            const unsigned int c = ix.list[c0 + j_begin + i];
            s_chunk[i] = c;
            s_fill[i] = F.chunk_fill[c];
        }
        __syncthreads();
        int ox, oy;
        spec_origin(F, t, ox, oy);
        const SpecRec* __restrict__ recs = (const SpecRec*)F.rec;
        constexpr int PER = 1024 / OT_FUSE_CH;
        constexpr int DEPTH = 4;  // two stages of DEPTH records in flight per thread, see fuse_accum_kernel
        const int g = threadIdx.x / OT_FUSE_CH, slot = threadIdx.x % OT_FUSE_CH;
        const int n_c = (int)(j_end - j_begin);
        auto load = [&](int first, SpecRec* rec, bool* ok) {
    #pragma unroll
            for (int k = 0; k < DEPTH; k++) {
                const int i = first + k * PER, ic = i < n_c ? i : 0;
                ok[k] = i < n_c && (unsigned int)slot < s_fill[ic];
                if (ok[k]) rec[k] = recs[(size_t)s_chunk[ic] * OT_FUSE_CH + slot];
            }
        };
        auto add = [&](const SpecRec* rec, const bool* ok) {
    #pragma unroll
            for (int k = 0; k < DEPTH; k++) {
                if (!ok[k]) continue;
                int32_t px, py;
                const int pix = hit_pixel(F.a, rec[k].x, rec[k].y, px, py);
                if (pix < 0) continue;
                const double wm = (double)rec[k].w * F.a.ws;
                double xo, yo, zo;
                observer_xyz_at6(obs, (double)rec[k].wl, xo, yo, zo);
                const int lx = px - ox, ly = py - oy;
                if ((unsigned)lx < (unsigned)OT_TILE_W && (unsigned)ly < (unsigned)OT_TILE_W) {
                    double* hv = tile + ((ly << 6) | lx);  // plane-major window, see fuse_accum_kernel
                    unsafeAtomicAdd(hv + 0 * OT_TILE_PX, xo * wm);
                    unsafeAtomicAdd(hv + 1 * OT_TILE_PX, yo * wm);
                    unsafeAtomicAdd(hv + 2 * OT_TILE_PX, zo * wm);
                    unsafeAtomicAdd(hv + 3 * OT_TILE_PX, 1.0 * wm);
                } else {
                    double* hg = F.hist + (int64_t)pix * 4;
                    unsafeAtomicAdd(hg + 0, xo * wm);
                    unsafeAtomicAdd(hg + 1, yo * wm);
                    unsafeAtomicAdd(hg + 2, zo * wm);
                    unsafeAtomicAdd(hg + 3, 1.0 * wm);
                }
            }
        };
        SpecRec ra[DEPTH], rb[DEPTH];
        bool oa[DEPTH], ob[DEPTH];
        load(g, ra, oa);
        for (int i0 = g; i0 < n_c; i0 += 2 * PER * DEPTH) {
            load(i0 + PER * DEPTH, rb, ob);
            add(ra, oa);
            load(i0 + 2 * PER * DEPTH, ra, oa);
            add(rb, ob);
        }
        __syncthreads();
        double* slab = ix.slabs + (size_t)b * (OT_TILE_PX * 4);
        for (int i = threadIdx.x; i < OT_TILE_PX * 4; i += blockDim.x) slab[i] = tile[i];
        __syncthreads();  // the slab is out: the window part holds wstart again in the next round
    }
}

// grid (16, K): thread = one pixel of the window of tile blockIdx.y, all four planes; windows overlap -> atomics
__global__ __launch_bounds__(256) void spec_reduce_kernel(FuseOne F, FuseIndex ix) {
    const int tl = blockIdx.y;
    const unsigned int s_first = ix.wstart[tl], s_end = ix.wstart[tl + 1];
    if (s_end == s_first) return;
    const int local = blockIdx.x * blockDim.x + threadIdx.x;
    int ox, oy;
    spec_origin(F, tl, ox, oy);
    const int px = ox + (local & (OT_TILE_W - 1)), py = oy + (local >> 6);
    if (px < 0 || py < 0 || px >= F.a.Nx || py >= F.a.Ny) return;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    for (unsigned int s = s_first; s < s_end; s++) {
        const double* sl = ix.slabs + (size_t)s * (OT_TILE_PX * 4) + local;
        s0 += sl[0 * OT_TILE_PX];
        s1 += sl[1 * OT_TILE_PX];
        s2 += sl[2 * OT_TILE_PX];
        s3 += sl[3 * OT_TILE_PX];
    }
    if (s3 == 0.0) return;  // (weights are positive: no hit in this pixel)
    double* hg = F.hist + ((int64_t)py * F.a.Nx + px) * 4;
    unsafeAtomicAdd(hg + 0, s0);
    unsafeAtomicAdd(hg + 1, s1);
    unsafeAtomicAdd(hg + 2, s2);
    unsafeAtomicAdd(hg + 3, s3);
}

// the escape list: a few hits far outside the sample's extent, straight into the image
__global__ __launch_bounds__(256) void spec_escaped_kernel(FuseOne F, const double* __restrict__ table) {
    const unsigned int n = min(F.g.esc_n[0], F.g.esc_cap);
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const SpecRec r = F.g.esc[i];
        int32_t px, py;
        const int pix = hit_pixel(F.a, r.x, r.y, px, py);
        if (pix < 0) continue;
        double xo, yo, zo;
        observer_xyz_at(table, (double)r.wl, xo, yo, zo);
        const double wm = (double)r.w * F.a.ws;
        double* hg = F.hist + (int64_t)pix * 4;
        unsafeAtomicAdd(hg + 0, xo * wm);
        unsafeAtomicAdd(hg + 1, yo * wm);
        unsafeAtomicAdd(hg + 2, zo * wm);
        unsafeAtomicAdd(hg + 3, 1.0 * wm);
    }
}
