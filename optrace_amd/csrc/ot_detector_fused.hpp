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
#include "ot_render_tiles.hpp"

#define OT_FUSE_CH 256       // records per chunk (3 KB)
#define OT_FUSE_BR 512       // rays per sub-block = threads per workgroup of the tile kernel
#define OT_FUSE_CPW 256      // chunks per accumulation workgroup (and slab): 64 records per thread
#define OT_FUSE_NONE 0xffffffffu
#define OT_FUSE_LDS_ENTRIES 2400  // (detector, tile) entries a tile-kernel workgroup can keep (20 B each)

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
    return hit_pixel(a, ph.x, ph.y, ix, iy);
}

// ---- probe: distinct pixels among the hits of 4096 sample rays ---------------------------------------------
template <bool NUMERIC>
__global__ __launch_bounds__(1024) void fuse_probe_kernel(ot_rays R, int64_t first, int64_t count,
                                                          const FuseOne* __restrict__ dets) {
    extern __shared__ int pset[];  // OT_TILE_PROBE_SET keys
    __shared__ int distinct;
    const auto& F = as_const(dets)[blockIdx.x];
    for (int i = threadIdx.x; i < OT_TILE_PROBE_SET; i += blockDim.x) pset[i] = -1;
    if (threadIdx.x == 0) distinct = 0;
    __syncthreads();
    const int64_t S = count < OT_TILE_PROBE ? count : OT_TILE_PROBE;
    const int64_t stride = count / S;
    for (int64_t k = threadIdx.x; k < S; k += blockDim.x) {
        const int64_t r = first + k * stride;
        const SectionPair sp = load_section_pair(R, r, true);
        V3 ph;
        float w;
        bool valid, ill, to;
        detector_hit<NUMERIC>(R, r, true, F, sp, pair_direction(sp), ph, w, valid, ill, to);
        if (!valid) continue;
        int32_t ix, iy;
        const int pix = fuse_pixel(F, ph, ix, iy);
        if (pix < 0) continue;
        unsigned int h = ((unsigned int)pix * 2654435761u) >> (32 - 13);  // OT_TILE_PROBE_SET = 2^13
        for (int pr = 0; pr < OT_TILE_PROBE_SET; pr++) {  // the set is twice as large as the sample: always ends
            const int sidx = (int)((h + pr) & (OT_TILE_PROBE_SET - 1));
            int k0 = pset[sidx];
            if (k0 == -1) {
                k0 = atomicCAS(&pset[sidx], -1, pix);
                if (k0 == -1) {
                    atomicAdd(&distinct, 1);
                    break;
                }
            }
            if (k0 == pix) break;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) F.spread[0] = (distinct > OT_TILE_DISTINCT) && F.tiles_ok;
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
    for (int64_t s = i0; s < i_end; s += blockDim.x) {  // whole workgroup iterates together (ballots below)
        const int64_t q = s + threadIdx.x;
        const bool active = q < i_end;
        const int64_t r = first + (active ? q : 0);
        const SectionPair sp = load_section_pair(R, r, active);
        const double wl = active ? (double)R.wl[r] : 0.0;
        const V3 sdir = pair_direction(sp);
        double xo = 0.0, yo = 0.0, zo = 0.0;
        bool have_obs = false;
#pragma unroll
        for (int d = 0; d < NDET; d++) {
            if (d >= n_det) continue;
            const auto& F = as_const(dets)[d];
            if (F.spread[0]) continue;
            V3 ph;
            float w;
            bool valid, ill, to;
            detector_hit<GENERAL, GENERAL>(R, r, active, F, sp, sdir, ph, w, valid, ill, to);
            if (GENERAL) fuse_count_ill(F, ill, to);
            if (!valid) continue;
            int32_t ix, iy;
            const int pix = fuse_pixel(F, ph, ix, iy);
            if (pix < 0) continue;
            if (!have_obs) {
                observer_xyz_at(obs, wl, xo, yo, zo);
                have_obs = true;
            }
            const double wm = (double)w;
            const int key = pix * OT_DET_MAX + d;
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
            double* hv = (slot >= 0) ? &hval[slot * 4] : F.hist + (int64_t)pix * 4;
            unsafeAtomicAdd(hv + 0, xo * wm);
            unsafeAtomicAdd(hv + 1, yo * wm);
            unsafeAtomicAdd(hv + 2, zo * wm);
            unsafeAtomicAdd(hv + 3, 1.0 * wm);
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

// RPT rays per thread and sub-block (2 where the detectors' images have at most 1024 tiles: more loads in flight per
// barrier); R's pointers are advanced to the first ray of the range by the host, rays are addressed with 32 bits.
template <bool GENERAL, int NDET, int RPT>
__global__ __launch_bounds__(OT_FUSE_BR) void fuse_tiles_kernel(ot_rays R, uint32_t count, const FuseOne* __restrict__ dets,
                                                                int n_det, int KT, uint32_t piece) {
    bool any = false;
    for (int d = 0; d < n_det; d++) any = any || as_const(dets)[d].spread[0];
    if (!any) return;
    constexpr uint32_t BRT = OT_FUSE_BR * RPT;        // rays per sub-block
    constexpr int TB = (RPT == 1) ? 11 : 10;          // bits of the tile number in a record key
    extern __shared__ unsigned int fl[];
    unsigned int* cnt = fl;  // [2][KT]
    unsigned int* fill = fl + 2 * KT;
    unsigned int* cur = fl + 3 * KT;
    unsigned int* nb = fl + 4 * KT;
    unsigned int* next = fl + 5 * KT;  // [n_det] next free chunk of this workgroup's part of each detector's pool
    if (threadIdx.x < (unsigned)n_det) next[threadIdx.x] = blockIdx.x * as_const(dets)[threadIdx.x].per_wg;
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
    // sections of the first sub-block
    SectionPair sp_n[RPT];
    float wl_n[RPT];
    bool act_n[RPT];
#pragma unroll
    for (int j = 0; j < RPT; j++) {
        const uint32_t q = i0 + j * OT_FUSE_BR + threadIdx.x;
        act_n[j] = q < i1;
        sp_n[j] = load_section_pair(R, (int64_t)(act_n[j] ? q : 0u), act_n[j]);
        wl_n[j] = act_n[j] ? OT_STREAM_LOAD(&R.wl[q]) : 0.f;
    }
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
            // request the next sub-block's sections now: they travel while this one is processed
            const uint32_t q = s + BRT + j * OT_FUSE_BR + threadIdx.x;
            act_n[j] = q < i1;  // (count < 2^31: no wrap)
            sp_n[j] = load_section_pair(R, (int64_t)(act_n[j] ? q : 0u), act_n[j]);
            wl_n[j] = act_n[j] ? OT_STREAM_LOAD(&R.wl[q]) : 0.f;
        }

        // The detector records are read through a pointer the optimiser cannot see through, once per sub-block:
        // otherwise lane constants derived from them are hoisted out of this loop and kept in vector registers.
        const FuseOne* dl = dets;
        asm volatile("" : "+s"(dl));
        unsigned int* cnt_a = cnt + par * KT;        // this sub-block
        unsigned int* cnt_b = cnt + (par ^ 1) * KT;  // the previous one (its records are written, fill / cur pending)
        // phase 1: hits, pixel, rank inside the tile's share of this sub-block
        float wk[RPT][NDET];
        unsigned int key[RPT][NDET];  // rank << (12 + TB) | tile << 12 | pixel in tile
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
                if (!F.spread[0]) continue;
                V3 ph;
                float w;
                bool valid, ill = false, to = false;
                // flat detector behind the last surface (the usual case): no section search; the wave takes the general
                // path only if one of its lanes needs it
                bool settled = false;
                if (!GENERAL) settled = detector_hit_last(F, R.nt, act[j], sp[j], sdir, ph, w, valid);
                if (GENERAL || __ballot(!settled) != 0ull) {
                    if (!settled) detector_hit<GENERAL, GENERAL>(R, r, act[j], F, sp[j], sdir, ph, w, valid, ill, to);
                }
                if (GENERAL) fuse_count_ill(F, ill, to);
                if (!valid) continue;
                int32_t ix, iy;
                if (fuse_pixel(F, ph, ix, iy) < 0) continue;
                const unsigned int local = (unsigned int)(((iy & (OT_TILE_W - 1)) << 6) | (ix & (OT_TILE_W - 1)));
                const unsigned int tile = (unsigned int)((iy >> 6) * F.tx + (ix >> 6));
                const unsigned int rank = atomicAdd(&cnt_a[F.koff + (int)tile], 1u);
                wk[j][d] = w;
                key[j][d] = (rank << (12 + TB)) | (tile << 12) | local;
            }
        }
        fuse_lds_barrier();
        // phase 2: the previous sub-block's counts move the open chunks on; tiles whose open chunk overflows with this
        // sub-block's records take new chunks from the workgroup's part of their detector's pool
        for (int e = threadIdx.x; e < KT; e += blockDim.x) {
            const unsigned int cb = cnt_b[e];
            if (cb) {
                fuse_advance(fill, cur, nb, e, cb);
                cnt_b[e] = 0u;
            }
            const unsigned int c = cnt_a[e];
            if (!c) continue;
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
        // which every wave reaches after these reads.)
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
                if (chunk < F.cap) {
                    TileRec rec = {wk[j][d], wl[j], local};
                    F.rec[(size_t)chunk * OT_FUSE_CH + dest] = rec;
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
        if (!F.spread[0]) continue;
        const unsigned int end = (blockIdx.x + 1) * F.per_wg;
        for (unsigned int c = min(next[d], end) + threadIdx.x; c < end; c += blockDim.x) F.chunk_tile[c] = OT_FUSE_NONE;
    }
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
__global__ __launch_bounds__(1024) void fuse_chunk_hist_kernel(FuseOne F, FuseIndex ix) {
    if (!F.spread[0]) return;
    const unsigned int n = F.cap;
    const unsigned int c0 = blockIdx.x * (1024 * OT_FUSE_IDX_PER);
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

__global__ __launch_bounds__(1024) void fuse_chunk_scan_kernel(FuseOne F, FuseIndex ix) {
    if (!F.spread[0]) return;
    __shared__ unsigned int n_s[OT_TILE_MAX];
    for (int i = threadIdx.x; i < F.K; i += blockDim.x) n_s[i] = ix.tile_n[i];
    __syncthreads();
    if (threadIdx.x == 0) {  // K <= 2048 entries in LDS: a serial scan costs a few microseconds
        unsigned int acc = 0, wg = 0;
        for (int i = 0; i < F.K; i++) {
            ix.tstart[i] = acc;
            ix.wstart[i] = wg;
            acc += n_s[i];
            wg += (n_s[i] + OT_FUSE_CPW - 1) / OT_FUSE_CPW;
        }
        ix.tstart[F.K] = acc;
        ix.wstart[F.K] = wg;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < F.K; i += blockDim.x) ix.tile_n[i] = 0u;  // becomes the placement cursor
}

// the workgroup's chunks of a tile get consecutive places behind one global reservation per tile
__global__ __launch_bounds__(1024) void fuse_chunk_place_kernel(FuseOne F, FuseIndex ix) {
    if (!F.spread[0]) return;
    const unsigned int n = F.cap;
    const unsigned int c0 = blockIdx.x * (1024 * OT_FUSE_IDX_PER);
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

// Accumulation, one workgroup per OT_FUSE_CPW chunks of a tile (wstart): the hits of an image are rarely spread evenly
// -- C4's picture covers a fifth of the detector, 42 of 225 tiles hold every record -- and a fixed number of workgroups
// per tile left most CUs idle behind the few heavy tiles (1.7 of 4 waves per SIMD resident on average).  Workgroup b
// finds its tile by bisection of wstart in LDS, adds its chunks into an LDS tile and writes slab b.
__global__ __launch_bounds__(1024) void fuse_accum_kernel(FuseOne F, FuseIndex ix, const double* __restrict__ table) {
    if (!F.spread[0]) return;
    extern __shared__ double lds[];  // [TILE_PX * 4 tile] [471 * 3 observer table]; the tile part first holds wstart
    double* tile = lds;
    double* obs = lds + OT_TILE_PX * 4;
    unsigned int* ws = (unsigned int*)lds;
    const int K = F.K;
    for (int i = threadIdx.x; i <= K; i += blockDim.x) ws[i] = ix.wstart[i];
    __syncthreads();
    const unsigned int b = blockIdx.x;
    if (b >= ws[K]) return;
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
    for (int i = threadIdx.x; i < OT_OBS_N * 3; i += blockDim.x) obs[i] = table[i];
    __syncthreads();
    constexpr int PER = 1024 / OT_FUSE_CH;  // chunks a workgroup handles at once
    constexpr int DEPTH = 8;                // chunk rounds in flight per thread (list -> fill -> record are dependent loads)
    const int g = threadIdx.x / OT_FUSE_CH, slot = threadIdx.x % OT_FUSE_CH;
    for (unsigned int j0 = j_begin + (unsigned int)g; j0 < j_end; j0 += PER * DEPTH) {
        TileRec rec[DEPTH];
        bool ok[DEPTH];
#pragma unroll
        for (int k = 0; k < DEPTH; k++) {
            const unsigned int j = j0 + k * PER;
            ok[k] = j < j_end;
            if (ok[k]) {
                const unsigned int c = ix.list[c0 + j];
                ok[k] = (unsigned int)slot < F.chunk_fill[c];
                if (ok[k]) rec[k] = F.rec[(size_t)c * OT_FUSE_CH + slot];
            }
        }
#pragma unroll
        for (int k = 0; k < DEPTH; k++) {
            if (!ok[k]) continue;
            const double wm = (double)rec[k].w;
            double xo, yo, zo;
            observer_xyz_at(obs, (double)rec[k].wl, xo, yo, zo);
            // plane-major tile [channel][pixel]: the lanes of one add then spread over 16 bank pairs; pixel-major
            // (4 doubles per pixel) would leave them 4 and make every add a 16-way bank conflict
            double* hv = tile + (int)rec[k].px;
            unsafeAtomicAdd(hv + 0 * OT_TILE_PX, xo * wm);
            unsafeAtomicAdd(hv + 1 * OT_TILE_PX, yo * wm);
            unsafeAtomicAdd(hv + 2 * OT_TILE_PX, zo * wm);
            unsafeAtomicAdd(hv + 3 * OT_TILE_PX, 1.0 * wm);
        }
    }
    __syncthreads();
    double* slab = ix.slabs + (size_t)b * (OT_TILE_PX * 4);
    for (int i = threadIdx.x; i < OT_TILE_PX * 4; i += blockDim.x) slab[i] = tile[i];
}

// grid (16, K): thread = one pixel of tile blockIdx.y, all four planes
__global__ __launch_bounds__(256) void fuse_reduce_kernel(FuseOne F, FuseIndex ix) {
    if (!F.spread[0]) return;
    const int tl = blockIdx.y;
    const unsigned int s_first = ix.wstart[tl], s_end = ix.wstart[tl + 1];
    if (s_end == s_first) return;
    const int local = blockIdx.x * blockDim.x + threadIdx.x;
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
