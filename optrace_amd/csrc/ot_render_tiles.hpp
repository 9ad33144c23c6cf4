// RenderImage.render for hits that spread over many pixels (render_image.py:396-418, misc.py:59-91).
//
// The direct path (render_kernel, ot_detector.hpp) privatises a few hundred warm pixels per workgroup in LDS and
// sends everything else to global f64 atomics.  An image of an extended object puts each of its ~10^6 pixels a few
// hundred times into the list, in no order: every hit becomes four memory-side atomics, and those run at 2e10/s
// (spread out) down to 2e9/s (a diffraction band of some 10^4 warm pixels) -- 12-17 ms for 1-2e8 hits, more than the
// trace that produced them.  Here the image is cut into 64 x 64 pixel tiles; one tile's four planes are 128 KB and
// fit the 160 KB of LDS of a CU:
//
//   probe    one workgroup looks at 4096 hits spread over the list; few distinct pixels -> the direct path runs
//            and the kernels below return at once (no host round trip)
//   count    hits per tile of every workgroup's contiguous piece of the list            reads x, y, w      20 B/hit
//   cursor   every piece's write cursor inside each tile's part of the list (one wave per tile)
//   scan     tile starts, work list of (tile, chunk of hits)
//   scatter  (w, wl, pixel inside the tile) of every hit to its tile's part of the list reads 24, writes 12 B/hit
//   accum    one workgroup per chunk: ds_add_f64 into the LDS tile, tile written to its own slab, no atomics
//   reduce   slabs of a tile summed into the image, every pixel owned by one thread
//
// 68 B of traffic per hit instead of 4 contended atomics.  Sums are the same values added in another order.
#pragma once
#include "ot_detector.hpp"

#define OT_TILE_W 64
#define OT_TILE_PX (OT_TILE_W * OT_TILE_W)
#define OT_TILE_MAX 2048           // tiles of the largest image (945 x 4725 -> 15 x 74)
#define OT_TILE_PIECES 1024        // workgroups of count / scatter: contiguous pieces of the hit list
#define OT_TILE_PROBE 4096         // hits the probe looks at
#define OT_TILE_PROBE_SET 8192     // hash set of the probe (ints, 32 KB)
#define OT_TILE_DISTINCT 1024      // more distinct pixels than this among the probed hits: tile path
#define OT_TILE_DISTINCT_COMPACT 128  // the same for compact hit lists (see tile_probe_kernel)

struct TileArgs {
    RenderArgs a;
    int32_t tx, ty, K;   // tiles along x, along y, in all
    int64_t n;           // hits
    int64_t piece;       // hits per piece (count / scatter)
    int64_t chunk;       // hits per chunk (accum)
    int32_t max_chunks;  // grid of accum, slabs allocated
};

// one hit in its tile's part of the list: a single 12-byte store per hit (three separate arrays cost three partial
// cache-line writes per hit, and the scatter is bound by exactly those)
struct TileRec {
    float w, wl;
    uint32_t px;  // pixel inside the tile
};

// everything the kernels exchange, carved out of one allocation
struct TileWork {
    const unsigned int* fill;    // compact hit list: entries of each piece that are filled (null: dense list)
    int* spread;                 // [1] probe verdict
    unsigned int* counts;        // [PIECES][K] hits of piece g in tile t, later the piece's write cursor
    unsigned long long* tot;     // [K] hits of each tile
    unsigned long long* starts;  // [K + 1] first record of each tile
    int* chunk_start;            // [K + 1] first chunk of each tile
    TileRec* rec;                // [n] hits sorted by tile
    double* slabs;               // [max_chunks][TILE_PX * 4]
};

OT_DEV int tile_of(const TileArgs& t, int32_t ix, int32_t iy, int& local) {
    local = ((iy & (OT_TILE_W - 1)) << 6) | (ix & (OT_TILE_W - 1));
    return (iy >> 6) * t.tx + (ix >> 6);
}

__global__ __launch_bounds__(1024) void tile_probe_kernel(TileArgs t, const double* __restrict__ px, const double* __restrict__ py,
                                                          const float* __restrict__ w, int* __restrict__ spread,
                                                          const unsigned int* __restrict__ fill) {
    extern __shared__ int pset[];  // OT_TILE_PROBE_SET keys
    __shared__ int distinct;
    for (int i = threadIdx.x; i < OT_TILE_PROBE_SET; i += blockDim.x) pset[i] = -1;
    if (threadIdx.x == 0) distinct = 0;
    __syncthreads();
    const int64_t S = t.n < OT_TILE_PROBE ? t.n : OT_TILE_PROBE;
    const int64_t stride = t.n / S;
    for (int64_t k = threadIdx.x; k < S; k += blockDim.x) {
        int64_t i = k * stride;
        if (fill) {  // compact list: sample k from piece k mod 1024, spread over what the piece holds
            const int64_t pc = k % OT_TILE_PIECES;
            const unsigned int f = fill[pc];
            if (!f) continue;
            i = pc * t.piece + (int64_t)(((unsigned long long)(k / OT_TILE_PIECES) * 2654435761ull) % f);
        }
        const float wi = w[i];
        if (!(wi > 0.f || wi < 0.f)) continue;
        int32_t ix, iy;
        const int pix = hit_pixel(t.a, px[i], py[i], ix, iy);
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
    // Compact lists scatter neighbouring rays over all pieces, so a workgroup of the direct kernel no longer sees one
    // corner of the image: its LDS table overflows earlier and a wrong verdict is expensive (C5's diffraction band, near
    // the dense threshold: 47 instead of 3.8 ms when the probe said "direct").  Only clearly point-like images stay direct.
    if (threadIdx.x == 0) spread[0] = distinct > (fill ? OT_TILE_DISTINCT_COMPACT : OT_TILE_DISTINCT);
}

__global__ __launch_bounds__(1024) void tile_count_kernel(TileArgs t, const double* __restrict__ px, const double* __restrict__ py,
                                                          const float* __restrict__ w, TileWork wk) {
    if (!wk.spread[0]) return;
    __shared__ unsigned int cnt[OT_TILE_MAX];
    for (int i = threadIdx.x; i < t.K; i += blockDim.x) cnt[i] = 0u;
    __syncthreads();
    const int64_t i0 = (int64_t)blockIdx.x * t.piece;
    int64_t i1 = (i0 + t.piece < t.n) ? i0 + t.piece : t.n;
    if (wk.fill) i1 = i0 + (int64_t)wk.fill[blockIdx.x];
    for (int64_t i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
        const float wi = w[i];
        if (!(wi > 0.f || wi < 0.f)) continue;
        int32_t ix, iy;
        if (hit_pixel(t.a, px[i], py[i], ix, iy) < 0) continue;
        int local;
        atomicAdd(&cnt[tile_of(t, ix, iy, local)], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < t.K; i += blockDim.x) wk.counts[(size_t)blockIdx.x * t.K + i] = cnt[i];
}

// one wave per tile: the counts of the 1024 pieces (16 per lane) become each piece's write cursor relative to the tile
// start (32 bits: a tile holds fewer than 2^32 hits because the list does); tot[tile] = hits of the tile
__global__ __launch_bounds__(256) void tile_cursor_kernel(TileArgs t, TileWork wk, unsigned long long* __restrict__ tot) {
    if (!wk.spread[0]) return;
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= t.K) return;
    const int lane = threadIdx.x & 63;
    constexpr int PER = OT_TILE_PIECES / 64;
    unsigned int c[PER], sum = 0;
#pragma unroll
    for (int k = 0; k < PER; k++) {
        c[k] = wk.counts[(size_t)(lane * PER + k) * t.K + tile];
        sum += c[k];
    }
    unsigned int incl = sum;  // inclusive scan over the lanes
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned int v = __shfl_up(incl, d);
        if (lane >= d) incl += v;
    }
    unsigned int run = incl - sum;
#pragma unroll
    for (int k = 0; k < PER; k++) {
        wk.counts[(size_t)(lane * PER + k) * t.K + tile] = run;
        run += c[k];
    }
    if (lane == 63) tot[tile] = incl;
}

// one workgroup: tile starts and the work list of the accumulation (chunks of t.chunk hits of one tile)
__global__ __launch_bounds__(1024) void tile_scan_kernel(TileArgs t, TileWork wk, const unsigned long long* __restrict__ tot) {
    if (!wk.spread[0]) return;
    __shared__ unsigned long long acc_s[OT_TILE_MAX];
    __shared__ int nch_s[OT_TILE_MAX];
    for (int i = threadIdx.x; i < t.K; i += blockDim.x) {
        acc_s[i] = tot[i];
        nch_s[i] = (int)((tot[i] + (unsigned long long)t.chunk - 1) / (unsigned long long)t.chunk);
    }
    __syncthreads();
    if (threadIdx.x == 0) {  // K <= 2048 entries in LDS: a serial scan costs a few microseconds
        unsigned long long acc = 0;
        int cacc = 0;
        for (int i = 0; i < t.K; i++) {
            const unsigned long long n_i = acc_s[i];
            const int c_i = nch_s[i];
            wk.starts[i] = acc;
            wk.chunk_start[i] = cacc;
            acc += n_i;
            cacc += c_i;
        }
        wk.starts[t.K] = acc;
        wk.chunk_start[t.K] = cacc;
    }
}

__global__ __launch_bounds__(1024) void tile_scatter_kernel(TileArgs t, const double* __restrict__ px, const double* __restrict__ py,
                                                            const float* __restrict__ w, const float* __restrict__ wl, TileWork wk) {
    if (!wk.spread[0]) return;
    __shared__ unsigned int cur[OT_TILE_MAX];
    for (int i = threadIdx.x; i < t.K; i += blockDim.x) cur[i] = wk.counts[(size_t)blockIdx.x * t.K + i];
    __syncthreads();
    const int64_t i0 = (int64_t)blockIdx.x * t.piece;
    int64_t i1 = (i0 + t.piece < t.n) ? i0 + t.piece : t.n;
    if (wk.fill) i1 = i0 + (int64_t)wk.fill[blockIdx.x];
    for (int64_t i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
        const float wi = w[i];
        if (!(wi > 0.f || wi < 0.f)) continue;
        int32_t ix, iy;
        if (hit_pixel(t.a, px[i], py[i], ix, iy) < 0) continue;
        int local;
        const int tl = tile_of(t, ix, iy, local);
        const unsigned long long pos = wk.starts[tl] + atomicAdd(&cur[tl], 1u);
        TileRec rec = {wi, wl[i], (uint32_t)local};
        wk.rec[pos] = rec;
    }
}

__global__ __launch_bounds__(1024) void tile_accum_kernel(TileArgs t, const double* __restrict__ table, TileWork wk) {
    if (!wk.spread[0]) return;
    const int c = blockIdx.x;
    if (c >= wk.chunk_start[t.K]) return;
    extern __shared__ double lds[];  // [TILE_PX * 4 tile] [471 * 6 observer table: (value, difference) pairs]
    double* tile = lds;
    double* obs = lds + OT_TILE_PX * 4;
    for (int i = threadIdx.x; i < OT_TILE_PX * 4; i += blockDim.x) tile[i] = 0.0;
    for (int i = threadIdx.x; i < OT_OBS_N * 6; i += blockDim.x) obs[i] = table[OT_OBS6_OFF + i];
    // the tile of this chunk: last tile whose first chunk is <= c
    int lo = 0, hi = t.K - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (wk.chunk_start[mid] <= c)
            lo = mid;
        else
            hi = mid - 1;
    }
    const unsigned long long r0 = wk.starts[lo] + (unsigned long long)(c - wk.chunk_start[lo]) * (unsigned long long)t.chunk;
    unsigned long long r1 = r0 + (unsigned long long)t.chunk;
    if (r1 > wk.starts[lo + 1]) r1 = wk.starts[lo + 1];
    __syncthreads();
    for (unsigned long long r = r0 + threadIdx.x; r < r1; r += blockDim.x) {
        const TileRec rec = wk.rec[r];
        const double wm = (double)rec.w * t.a.ws;
        double xo, yo, zo;
        observer_xyz_at6(obs, (double)rec.wl, xo, yo, zo);
        // plane-major tile [channel][pixel]: the lanes of one add spread over 16 bank pairs (with 4 doubles per pixel
        // they share 4, a 16-way bank conflict on every add)
        double* hv = tile + (int)rec.px;
        unsafeAtomicAdd(hv + 0 * OT_TILE_PX, xo * wm);
        unsafeAtomicAdd(hv + 1 * OT_TILE_PX, yo * wm);
        unsafeAtomicAdd(hv + 2 * OT_TILE_PX, zo * wm);
        unsafeAtomicAdd(hv + 3 * OT_TILE_PX, 1.0 * wm);
    }
    __syncthreads();
    double* slab = wk.slabs + (size_t)c * (OT_TILE_PX * 4);
    for (int i = threadIdx.x; i < OT_TILE_PX * 4; i += blockDim.x) slab[i] = tile[i];
}

// grid (16, K): thread = one pixel of tile blockIdx.y, all four planes
__global__ __launch_bounds__(256) void tile_reduce_kernel(TileArgs t, TileWork wk, double* __restrict__ hist) {
    if (!wk.spread[0]) return;
    const int tl = blockIdx.y;
    const int c0 = wk.chunk_start[tl], c1 = wk.chunk_start[tl + 1];
    if (c0 == c1) return;
    const int local = blockIdx.x * blockDim.x + threadIdx.x;
    const int ix = (tl % t.tx) * OT_TILE_W + (local & (OT_TILE_W - 1));
    const int iy = (tl / t.tx) * OT_TILE_W + (local >> 6);
    if (ix >= t.a.Nx || iy >= t.a.Ny) return;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    for (int c = c0; c < c1; c++) {
        const double* sl = wk.slabs + (size_t)c * (OT_TILE_PX * 4) + local;  // plane-major slab
        s0 += sl[0 * OT_TILE_PX];
        s1 += sl[1 * OT_TILE_PX];
        s2 += sl[2 * OT_TILE_PX];
        s3 += sl[3 * OT_TILE_PX];
    }
    double* hg = hist + ((int64_t)iy * t.a.Nx + ix) * 4;
    hg[0] += s0;
    hg[1] += s1;
    hg[2] += s2;
    hg[3] += s3;
}
