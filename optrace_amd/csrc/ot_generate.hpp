// On-device ray generation: RaySource.create_rays (ray_source.py:204-437) with the sampling helpers of
// optrace/tracer/random.py, redesigned for a counter-based GPU RNG.
//
// The reference builds whole arrays (stratified grid + dither, then a Fisher-Yates shuffle, random.py:32-43)
// with a sequential SFC64 stream.  Neither the stream nor the shuffle can be reproduced in parallel, so the
// device version is stateless per ray: ray j of a source range of n rays takes stratum perm_K(j), where perm_K
// is a keyed bijection of [0, n) (cycle-walking hash permutation), and its dither comes from Philox-4x32-10
// keyed by (seed, global ray index).  Every independently shuffled quantity of the reference
// (position, divergence, wavelength, polarisation, ...) gets its own permutation key, which gives the same
// joint distribution: stratified marginals, independent pairing.  Parity is therefore statistical
// (tests/test_sources_gpu.py against fingerprints of the reference) -- SURVEY.md section 7 "RNG".
#pragma once
#include "ot_device.hpp"

enum : uint32_t {
    ST_POS = 1, ST_PIX_JITTER = 2, ST_DIV = 3, ST_DIV_ALPHA = 4, ST_WL = 5, ST_POL = 6, ST_RGB_CHOICE = 7,
    ST_RGB_WL = 8, ST_PIXEL = 9
};

// Keyed bijection of [0, l): hash rounds on the enclosing power of two with cycle walking
// (construction of A. Kensler, "Correlated Multi-Jittered Sampling", Pixar TM 13-01, 2013).
OT_DEV uint32_t permute_hash(uint32_t i, uint32_t w, uint32_t key) {  // bijection of [0, w], w = 2^k - 1
    i ^= key;
    i *= 0xe170893du;
    i ^= key >> 16;
    i ^= (i & w) >> 4;
    i ^= key >> 8;
    i *= 0x0929eb3fu;
    i ^= key >> 23;
    i ^= (i & w) >> 1;
    i *= 1u | key >> 27;
    i *= 0x6935fa69u;
    i ^= (i & w) >> 11;
    i *= 0x74dcb303u;
    i ^= (i & w) >> 2;
    i *= 0x9e501cc3u;
    i ^= (i & w) >> 2;
    i *= 0xc860a3dfu;
    i &= w;
    i ^= i >> 5;
    return i;
}

// Blocks of 2^20 strata and more (where the rays are): three multiply / xor-shift rounds whose shifts follow the width k
// of the domain -- half and a third of it, so that every round folds the high bits, which a multiplication modulo 2^k
// leaves poorly mixed, onto the low ones.  16 instead of 34 instructions.  On these domain sizes its pairings are
// indistinguishable from the hash above in chi-square tests of pairs and triples of streams, of a stream against the ray
// index and against itself at lags 1, 64 and 4096, and for neighbouring keys (tools/experiments/perm_quality.py,
// profiles/r1/perm_quality.txt); below 2^20 it is measurably weaker, so small blocks keep the longer hash.
OT_DEV uint32_t permute_pow2_large(uint32_t i, uint32_t w, uint32_t key) {
    const int k = 32 - __builtin_clz(w);  // wave-uniform, like w and key: the constants below live on the scalar unit
    const int sA = (k + 1) >> 1, sB = (k + 2) / 3;
    const uint32_t m2 = (0x0929eb3fu ^ ((key >> 7) << 1)) | 1u;
    i ^= key & w;
    i = (i * 0xe170893du) & w;
    i ^= i >> sA;
    i = (i * m2) & w;
    i ^= i >> sB;
    i ^= (key >> 13) & w;
    i = (i * 0x6935fa69u) & w;
    i ^= i >> sA;
    return (i + ((key >> 3) & w)) & w;
}

OT_DEV uint32_t permute_index(uint32_t i, uint32_t l, uint32_t key) {
    if (l <= 1) return 0;
    uint32_t w = l - 1;
    if ((l & w) == 0) {  // power of two (what the host cuts long ranges into): no cycle walking, no divergence
        if (l >= (1u << 20)) return permute_pow2_large(i, w, key);
        i = permute_hash(i, w, key);
        return (i + (key & w)) & w;
    }
    w |= w >> 1;
    w |= w >> 2;
    w |= w >> 4;
    w |= w >> 8;
    w |= w >> 16;
    // cycle walking: a wave iterates until its slowest lane is back inside [0, l) -- with l just above half the
    // power of two that is 4-6 rounds for 64 lanes, although a single lane needs 1.3 on average
    do {
        i = permute_hash(i, w, key);
    } while (i >= l);
    // rotate by a key-dependent offset in [0, l): w < 2 l, so one conditional subtraction each replaces `% l`
    uint32_t rot = key & w;
    if (rot >= l) rot -= l;
    i += rot;
    if (i >= l) i -= l;
    return i;
}

// permutation key of one (seed, range, stream): a 32-bit finaliser-style mix (wave-uniform, a few SALU ops)
OT_DEV uint32_t stream_key(uint64_t seed, uint32_t range, uint32_t stream) {
    uint32_t h = (uint32_t)seed ^ ((uint32_t)(seed >> 32) * 0x9E3779B1u) ^ (range * 0x85EBCA77u) ^ (stream * 0xC2B2AE3Du);
    h ^= h >> 16;
    h *= 0x7feb352du;
    h ^= h >> 15;
    h *= 0x846ca68bu;
    h ^= h >> 16;
    return h;
}

struct GenCtx {
    uint64_t seed;
    uint64_t gidx;   // global ray index in the launch (Philox counter)
    uint32_t j;      // index inside the source range
    uint32_t n;      // rays in the source range (stratification domain)
    uint32_t range;  // range id (permutation keys differ per range)
    uint32_t n2;     // floor(sqrt(n)): side of the jittered grid of the 2-D samplers
    double inv_n, inv_n2;  // 1 / n, 1 / n2 (host)
    float w;         // power of each ray of the range (host)
    // dither values in [0, 1): the raw words of one or two Philox blocks, turned into doubles where they are used
    // (`dither`): twelve ready-made doubles occupied 24 vector registers from the top of the generator to the last use
    uint32_t ra[4], rb[4];
    bool image, has_b;  // wave-uniform: eight 16-bit dithers out of block A (image sources); block B was drawn
};

// dither value of slot k (see dither_slot): 32 bits of block A (slots 0-3) or B (4-7); image sources cut block A into
// eight 16-bit values for the slots 0-3 and 8-11.  Slots nobody filled read 0.5.
OT_DEV double dither(const GenCtx& g, int k) {
    if (k >= 4 && k < 8) return g.has_b ? ((double)g.rb[k - 4] + 0.5) * 0x1.0p-32 : 0.5;
    if (g.image) {
        const int j = (k < 4) ? k : k - 4;  // 0 .. 7: word j / 2, low or high half
        const uint32_t wv = g.ra[j >> 1];
        return ((double)((j & 1) ? (wv >> 16) : (wv & 0xffffu)) + 0.5) * 0x1.0p-16;
    }
    return (k < 4) ? ((double)g.ra[k] + 0.5) * 0x1.0p-32 : 0.5;
}

// stratum index + dither in one fused multiply-add: (k + u) with u = bits * 2^-16 or 2^-32 in [0, 1) (the dither only places
// the sample inside its stratum: no centring, two instructions less per sample than `(double)k + dither(g, slot)`)
OT_DEV double stratum_plus_dither(const GenCtx& g, int slot, double k) {
    if (slot >= 4 && slot < 8) return g.has_b ? __builtin_fma((double)g.rb[slot - 4], 0x1.0p-32, k) : k + 0.5;
    if (g.image) {
        const int j = (slot < 4) ? slot : slot - 4;
        const uint32_t wv = g.ra[j >> 1];
        return __builtin_fma((double)((j & 1) ? (wv >> 16) : (wv & 0xffffu)), 0x1.0p-16, k);
    }
    return (slot < 4) ? __builtin_fma((double)g.ra[slot], 0x1.0p-32, k) : k + 0.5;
}

// stream -> which dither value(s) it uses (consecutive pairs for the 2-D samplers).  The slots are grouped so
// that a source only pays for the Philox blocks it needs: block A (u0-u3) serves every source -- point sources
// with direction, wavelength and polarisation need nothing else --, block B (u4-u7) extended emitters and 2-D
// divergence; image sources take eight 16-bit dithers (u0-u3, u8-u11) out of block A alone (fill_dither).
OT_DEV int dither_slot(uint32_t stream) {
    switch (stream) {
        case ST_DIV: return 0;         // u0, u1
        case ST_WL: return 2;
        case ST_RGB_WL: return 2;      // RGB images draw the wavelength from a primary instead of a spectrum
        case ST_POL: return 3;
        case ST_POS: return 4;         // u4, u5
        case ST_DIV_ALPHA: return 6;
        case ST_PIXEL: return 8;
        case ST_PIX_JITTER: return 9;  // u9, u10
        default: return 11;            // ST_RGB_CHOICE
    }
}

OT_DEV void fill_dither(GenCtx& g, bool need_b, bool image) {
    const uint32_t i0 = (uint32_t)g.gidx, i1 = (uint32_t)(g.gidx >> 32), k0 = (uint32_t)g.seed, k1 = (uint32_t)(g.seed >> 32);
    // image sources draw eight dithers (direction x 2, wavelength, polarisation, pixel, in-pixel x 2, primary): the one
    // block serves them all with 16 bits each -- a dither only places the ray inside its stratum, whose width is
    // 1 / (rays of the range) of the sampled interval already
    const Philox a = philox4x32<7>(i0, i1, 0x67656e31u, 0, k0, k1);
    g.image = image;
    g.has_b = need_b;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        g.ra[k] = a.c[k];
        g.rb[k] = 0u;
    }
    if (need_b) {
        const Philox b = philox4x32<7>(i0, i1, 0x67656e32u, 1, k0, k1);
#pragma unroll
        for (int k = 0; k < 4; k++) g.rb[k] = b.c[k];
    }
}

// which Philox blocks a source needs (see dither_slot)
template <bool IMAGES = true, class SRC>
OT_DEV void fill_dither_for(GenCtx& g, SRC& src) {
    const bool image = IMAGES && src.shape >= OT_SRC_IMAGE_RGB;
    const bool extended = src.shape != OT_SRC_POINT && !image;
    fill_dither(g, extended || (src.divergence != OT_DIV_NONE && src.div_2d), image);
}

// random.stratified_interval_sampling random.py:48-67: one of n strata of [a, b), uniformly dithered
OT_DEV double strat_interval(const GenCtx& g, uint32_t stream, double a, double b) {
    uint32_t k = permute_index(g.j, g.n, stream_key(g.seed, g.range, stream));
    double dba = (b - a) * g.inv_n;
    return a + stratum_plus_dither(g, dither_slot(stream), (double)k) * dba;
}

// the same for a stratum index the caller already has (a second stream tied to another one's permutation, see generate_ray)
OT_DEV double strat_interval_at(const GenCtx& g, uint32_t k, uint32_t stream, double a, double b) {
    double dba = (b - a) * g.inv_n;
    return a + stratum_plus_dither(g, dither_slot(stream), (double)k) * dba;
}

// random.stratified_rectangle_sampling random.py:8-45: floor(sqrt(n))^2 jittered grid cells, the remaining
// n - N2^2 samples uniform over the rectangle
OT_DEV void strat_rect(const GenCtx& g, uint32_t stream, double a, double b, double c, double d, double& x, double& y) {
    uint32_t k = permute_index(g.j, g.n, stream_key(g.seed, g.range, stream));
    const int slot = dither_slot(stream);
    if ((g.n & (g.n - 1u)) == 0u && g.n >= 4u) {
        // A power-of-two block of 2^m rays (what the host cuts long ranges into): a jittered grid of 2^ceil(m/2) x 2^floor(m/2)
        // cells takes EVERY sample (the reference's floor(sqrt n)^2 square cells leave n - N2^2 samples unstratified) and a
        // sample finds its cell with a shift and a mask instead of a division with fix-ups.  (wave-uniform branch)
        const int m = 31 - __builtin_clz(g.n), mx = (m + 1) >> 1, my = m >> 1;
        const uint32_t ix = k & ((1u << mx) - 1u), iy = k >> mx;
        const double sx = __builtin_ldexp(b - a, -mx), sy = __builtin_ldexp(d - c, -my);  // (scalar operands: per wave)
        x = a + stratum_plus_dither(g, slot, (double)ix) * sx;
        y = c + stratum_plus_dither(g, slot + 1, (double)iy) * sy;
        return;
    }
    double u0 = dither(g, slot), u1 = dither(g, slot + 1);
    const uint32_t N2 = g.n2;
    if (k < N2 * N2) {
        // k / N2 through the host's 1 / N2 with an exact fix-up instead of an integer division
        uint32_t iy = (uint32_t)(((double)k + 0.5) * g.inv_n2);
        if (iy * N2 > k) iy--;
        if ((iy + 1) * N2 <= k) iy++;
        uint32_t ix = k - iy * N2;
        x = a + ((double)ix + u0) * ((b - a) * g.inv_n2);
        y = c + ((double)iy + u1) * ((d - c) * g.inv_n2);
    } else {
        x = a + u0 * (b - a);
        y = c + u1 * (d - c);
    }
}

// random.stratified_ring_sampling random.py:70-110: Shirley's equal-area square->disc map, then disc->annulus.
// polar == false: cartesian (x, y); polar == true: (|r|, theta / pi) with theta shifted by -pi for negative r.
OT_DEV void strat_ring(const GenCtx& g, uint32_t stream, double ri, double r, bool polar, double& o0, double& o1) {
    double x, y;
    strat_rect(g, stream, -r, r, -r, r, x, y);
    double x2 = x * x, y2 = y * y;
    double r_ = 0.0, th = 0.0;  // th = theta / pi: sincospi needs no range reduction against an inexact pi
    if (x2 > y2) {
        r_ = x;
        th = 0.25 * ot_div(y, x);
    } else if (y2 > 0) {
        r_ = y;
        th = 0.5 - 0.25 * ot_div(x, y);
    }
    if (ri != 0.0) {
        double q = ot_div(ri, r);
        double m = ot_sqrt(ri * ri + r_ * r_ * (1 - q * q));
        r_ = (r_ < 0) ? -m : m;
    }
    if (!polar) {
        double sn, cs;
        sincospi_small(th, &sn, &cs);
        o0 = r_ * cs;
        o1 = r_ * sn;
    } else {  // (|r|, theta / pi) with theta shifted by -pi for negative r
        if (r_ < 0) th -= 1.0;
        o0 = fabs(r_);
        o1 = th;
    }
}

template <class GD>
OT_DEV int guide_start(GD& G, double X) {
    double t = (X - G.x0) * G.scale;
    int b = (t > 0.0) ? ((t < (double)(G.K - 1)) ? (int)t : G.K - 1) : 0;
    return G.g[b];
}

// random.inverse_transform_sampling random.py:113-159, kind="discrete": first entry whose cumulative weight
// reaches X (scipy interp1d kind="next"); tab = n values then n cumulative weights.  Returns the index.
template <class GD>
OT_DEV int cdf_index_discrete(const double* __restrict__ F, int n, double X, GD& G) {
    int lo = guide_start(G, X);  // smallest j with F[j] >= X (n - 1 if there is none)
    while (lo < n - 1 && F[lo] < X) lo++;
    while (lo > 0 && F[lo - 1] >= X) lo--;
    return lo;
}

template <class GD>
OT_DEV double inv_cdf_discrete(const double* __restrict__ tab, int n, double X, GD& G) {
    return tab[cdf_index_discrete(tab + n, n, X, G)];
}

// kind="continuous": linear interpolation of the inverse cumulative-trapezoid table (random.py:150-157).
// pairs = (F_0, x_0, F_1, x_1, ...): the start hint gives a node at or just before X; its pair and the next one
// come in with one 32-byte load, stepping is rare with four hint buckets per table node.
template <class GD>
OT_DEV double inv_cdf_linear(const double* __restrict__ pairs, int n, double X, GD& G) {
    if (n < 2) return pairs[1];
    int lo = guide_start(G, X);  // largest j <= n - 2 with F[j] <= X
    if (lo > n - 2) lo = n - 2;
    double F0 = pairs[2 * lo], x0 = pairs[2 * lo + 1], F1 = pairs[2 * lo + 2], x1 = pairs[2 * lo + 3];
    while (lo < n - 2 && F1 <= X) {
        lo++;
        F0 = F1;
        x0 = x1;
        F1 = pairs[2 * lo + 2];
        x1 = pairs[2 * lo + 3];
    }
    while (lo > 0 && F0 > X) {
        lo--;
        F1 = F0;
        x1 = x0;
        F0 = pairs[2 * lo];
        x0 = pairs[2 * lo + 1];
    }
    double dF = F1 - F0;
    if (!(dF > 0)) return x0;
    return x0 + ot_div(X - F0, dF) * (x1 - x0);
}

struct NewRay {
    V3 p, s;
    double polx, poly, polz;
    float w, wl;
};

// ---- image sources: pixel by inverse CDF in two memory round trips ------------------------------------------------
// bucket of a value of the cumulative pixel pdf; the host builds SourceDev::pick_lo with this very expression
OT_HD int pixel_bucket(double X, double scale, int K) {
    const double t = X * scale;
    return (t > 0.0) ? ((t < (double)(K - 1)) ? (int)t : K - 1) : 0;
}

struct alignas(16) PixRec {  // SourceDev::pix_rec
    double F, c_r, c_rg, pad;
};

// random.inverse_transform_sampling random.py:113-159, kind="discrete", for the pixel of an image source: the first
// pixel whose cumulative weight reaches X (scipy interp1d kind="next").  Round trip 1 fetches the bucket's pixel range
// [lo, hi]; round trip 2 the records of pixels lo and lo + 1 -- cumulative weight AND colours, so whichever of the two
// it is (four buckets per pixel: nearly always) nothing more has to be fetched.  (Before: bucket hint, then a walk over
// the cumulative table in dependent loads, then the colours of the pixel found: four and more round trips, and the
// few-surface kernels are bound by exactly this latency chain -- profiles/r2/sq_configs.txt.)
struct PixelPick {
    double X;
    int lo, hi;
};

template <class SRC>
OT_DEV PixelPick pixel_pick_issue(SRC& src, const GenCtx& g, double X) {
    PixelPick pk;
    pk.X = X;
    const int b = pixel_bucket(X, src.pick_scale, src.pick_K);
    pk.lo = src.pick_lo[b];
    pk.hi = src.pick_lo[b + 1];
    return pk;
}

template <class SRC>
OT_DEV uint32_t pixel_pick_finish(SRC& src, const PixelPick& pk, int npx, const PixRec& r0, const PixRec& r1, PixRec& rec) {
    int P = pk.lo;
    rec = r0;
    const int hi = pk.hi < npx - 1 ? pk.hi : npx - 1;
    if (P < hi && rec.F < pk.X) {
        P++;
        rec = r1;
        const PixRec* recs = (const PixRec*)src.pix_rec;
        while (P < hi && rec.F < pk.X) {  // more than one pixel inside one bucket: dark pixels
            P++;
            rec = recs[P];
        }
    }
    return (uint32_t)P;
}

// RaySource.create_rays ray_source.py:204-437 for ray j of a range of n rays of source `src`.
// Image sources need three dependent memory round trips (pixel range, pixel records, wavelength of the primary); the
// function is laid out so that everything that does not depend on them -- in-pixel jitter, the divergence sample, the
// polarisation angle, later the frames and the direction -- is computed while they are in flight.
// IMAGES = false: a variant without the image sources (the discrete-spectrum kernels: the host sends scenes with an
// image source to the formula kernels, and the bench kernel keeps the 76 registers it needs without that path).
template <bool IMAGES = true, class SRC>
OT_DEV NewRay generate_ray(SRC& src, const GenCtx& g, bool no_pol) {
    NewRay o;
    o.w = g.w;  // power / N, ray_source.py:220
    const bool image = IMAGES && src.shape >= OT_SRC_IMAGE_RGB;
    const uint32_t npx = image ? (uint32_t)src.img_w * (uint32_t)src.img_h : 0u;

    // ---- image sources, round trip 1: the pixel range of this ray's stratified uniform variable (ray_source.py:243-245)
    PixelPick pk = {0.0, 0, 0};
    uint32_t k_pixel = 0u;  // the stratum of the pixel variable (RGB images tie the primary's variable to it, below)
    if (image) {
        k_pixel = permute_index(g.j, g.n, stream_key(g.seed, g.range, ST_PIXEL));
        if (npx > 1) pk = pixel_pick_issue(src, g, strat_interval_at(g, k_pixel, ST_PIXEL, 0.0, src.pix_total));
    }

    // ---- wavelength (light_spectrum.py:81-138) ----
    double wl = 0.0;
    if (!IMAGES || src.shape != OT_SRC_IMAGE_RGB) {
        switch (src.spectrum) {
            case OT_SPEC_MONO: wl = (double)(float)src.wl; break;
            case OT_SPEC_UNIFORM: wl = strat_interval(g, ST_WL, src.wl0, src.wl1); break;
            case OT_SPEC_LINES: {
                const double* F = src.spec_tab + src.n_spec;
                double X = strat_interval(g, ST_WL, 0.0, F[src.n_spec - 1]);
                wl = inv_cdf_discrete(src.spec_tab, (int)src.n_spec, X, src.g_spec);
                break;
            }
            case OT_SPEC_GAUSSIAN: {
                double X = strat_interval(g, ST_WL, src.gauss_xl, src.gauss_xr);
                wl = src.mu + 1.4142135623730951 * src.sig * erfinv(2 * X - 1);
                break;
            }
            default: {
                const double* F = src.spec_tab + src.n_spec;
                double X = strat_interval(g, ST_WL, F[0], F[src.n_spec - 1]);
                wl = inv_cdf_linear(src.spec_pairs, (int)src.n_spec, X, src.g_spec);
            }
        }
    }

    // ---- image sources: in-pixel position and the choice of the primary (no memory involved) ----
    double rx = 0.0, ry = 0.0, choice = 0.0;
    if (image) {
        // Position inside the pixel: uniform.  The reference draws it from a grid stratified over ALL rays of the source
        // and shuffled independently of the pixel choice (ray_source.py:247), so the rays that share a pixel carry a
        // random subset of that grid -- indistinguishable from independent uniform values.  The dithers serve directly;
        // a permutation and the grid arithmetic (~45 vector instructions per ray) bought nothing.
        rx = dither(g, dither_slot(ST_PIX_JITTER));
        ry = dither(g, dither_slot(ST_PIX_JITTER) + 1);
        if (src.shape == OT_SRC_IMAGE_RGB) {
            // The variable that chooses the primary (and, rescaled, places the wavelength inside it): stratified over the
            // rays of the range like the pixel variable.  Its stratum is the pixel stratum times an odd constant modulo the
            // block size -- a rank-1 lattice pairing (Korobov): the rays that share a pixel have consecutive pixel strata,
            // and consecutive multiples of 0.618.. n modulo n are spread evenly over [0, n), so every pixel sees the primaries
            // and their spectra with LESS noise than under an independent shuffle (the reference stratifies the wavelengths
            // over the rays of each primary for the same purpose, srgb.py:549-551) -- and the second keyed permutation (16
            // instructions) becomes a multiplication.  Ragged blocks (no power of two) keep their own permutation.
            if ((g.n & (g.n - 1u)) == 0u && g.n >= 2u) {
                const int m = 31 - __builtin_clz(g.n);                  // n = 2^m
                const uint32_t mult = (0x9E3779B1u >> (32 - m)) | 1u;   // ~ 0.618 n, odd: a bijection of [0, n) (scalar unit)
                choice = strat_interval_at(g, (k_pixel * mult + stream_key(g.seed, g.range, ST_RGB_CHOICE)) & (g.n - 1u),
                                           ST_RGB_CHOICE, 0.0, 1.0);
            } else
                choice = strat_interval(g, ST_RGB_CHOICE, 0.0, 1.0);
        }
    }
    // ---- image sources, round trip 2: the records of the first two pixels of the range ----
    PixRec r0 = {0.0, 0.0, 0.0, 0.0}, r1 = r0;
    if (image && npx > 1) {
        const PixRec* recs = (const PixRec*)src.pix_rec;
        r0 = recs[pk.lo];
        r1 = recs[pk.lo + 1 < (int)npx ? pk.lo + 1 : pk.lo];
    } else if (image) {
        r0 = ((const PixRec*)src.pix_rec)[0];
    }

    // ---- divergence sample (ray_source.py:290-351): sin / cos of the polar angle theta and of the azimuth alpha; where
    // the reference goes through asin / acos and back (ray_source.py:303-320, 343-351) the pair is formed algebraically
    double st = 0.0, ct = 1.0, sa = 0.0, ca = 1.0;
    auto divergence_sample = [&]() {
        if (src.divergence == OT_DIV_NONE) return;
        if (src.div_2d) {
            double X = strat_interval(g, ST_DIV_ALPHA, 0.0, 2.0);
            const double sgn = (X <= 1.0) ? 1.0 : -1.0;  // alpha = div_axis or div_axis + pi
            ca = sgn * src.axis_cos;
            sa = sgn * src.axis_sin;
            double theta;
            switch (src.divergence) {
                case OT_DIV_LAMBERTIAN: {
                    st = strat_interval(g, ST_DIV, 0.0, src.div_sin);  // theta = asin(st)
                    ct = ot_sqrt(1 - st * st);
                    break;
                }
                case OT_DIV_ISOTROPIC:
                    theta = strat_interval(g, ST_DIV, 0.0, src.div_rad);
                    sincos(theta, &st, &ct);
                    break;
                default: {
                    const double* F = src.div_tab + src.n_div;
                    double X2 = strat_interval(g, ST_DIV, F[0], F[src.n_div - 1]);
                    theta = inv_cdf_linear(src.div_pairs, (int)src.n_div, X2, src.g_div);
                    sincos(theta, &st, &ct);
                }
            }
        } else {
            double r, alpha_pi;
            strat_ring(g, ST_DIV, 0.0, src.div_sin, true, r, alpha_pi);
            sincospi_small(alpha_pi, &sa, &ca);
            switch (src.divergence) {
                case OT_DIV_LAMBERTIAN:  // theta = asin(r)
                    st = r;
                    ct = ot_sqrt(1 - r * r);
                    break;
                case OT_DIV_ISOTROPIC:  // theta = acos(1 - r^2)
                    ct = 1 - r * r;
                    st = r * ot_sqrt(2 - r * r);
                    break;
                default: {
                    const double* F = src.div_tab + src.n_div;
                    double X0 = r * r / (src.div_sin * src.div_sin);
                    double theta = inv_cdf_linear(src.div_pairs, (int)src.n_div, F[0] + X0 * (F[src.n_div - 1] - F[0]), src.g_div);
                    sincos(theta, &st, &ct);
                }
            }
        }
    };

    // ---- polarisation angle (ray_source.py:359-385) ----
    double psn = 0.0, pcs = 1.0;
    auto polarization_angle = [&]() {
        if (no_pol) return;
        switch (src.polarization) {
            case OT_POL_CONSTANT:
                pcs = src.pol_cos;
                psn = src.pol_sin;
                break;
            case OT_POL_UNIFORM: sincospi_small(strat_interval(g, ST_POL, 0.0, 2.0), &psn, &pcs); break;  // angle in [0, 2 pi)
            case OT_POL_LIST: {
                const double* F = src.pol_tab + src.n_pol;
                double ang = inv_cdf_discrete(src.pol_tab, (int)src.n_pol, strat_interval(g, ST_POL, 0.0, F[src.n_pol - 1]), src.g_pol);
                sincos(ang, &psn, &pcs);
                break;
            }
            default: {
                const double* F = src.pol_tab + src.n_pol;
                double ang = inv_cdf_linear(src.pol_pairs, (int)src.n_pol, strat_interval(g, ST_POL, F[0], F[src.n_pol - 1]), src.g_pol);
                sincos(ang, &psn, &pcs);
            }
        }
    };
    // Image sources sample both while their pixel records are on the way; every other source where the values are used
    // (the samples would only occupy registers in between: 86 instead of 76 for the point sources of the bench scene).
    if (image) {
        divergence_sample();
        polarization_angle();
    }

    // ---- start position (ray_source.py:229-255) ----
    V3 p = {src.pos[0], src.pos[1], src.pos[2]};
    double wl_x0 = 0.0, wl_x1 = 0.0, wl_f = 0.0;  // RGB images: two samples of the primary's inverse table and the weight
    switch (src.shape) {
        case OT_SRC_POINT: break;
        case OT_SRC_LINE: {
            double t = strat_interval(g, ST_POS, -src.r, src.r);
            p.x += src.ca * t;
            p.y += src.sa * t;
            break;
        }
        case OT_SRC_CIRCLE:
        case OT_SRC_RING: {
            double x, y;
            strat_ring(g, ST_POS, src.shape == OT_SRC_RING ? src.ri : 0.0, src.r, false, x, y);
            p.x += x;
            p.y += y;
            break;
        }
        case OT_SRC_RECT: {
            double x, y;
            strat_rect(g, ST_POS, -src.dim[0] / 2, src.dim[0] / 2, -src.dim[1] / 2, src.dim[1] / 2, x, y);
            p.x += x * src.ca - y * src.sa;
            p.y += x * src.sa + y * src.ca;
            break;
        }
        default: {  // image sources: pixel by inverse CDF, uniform inside the pixel
            if (!IMAGES) break;
            PixRec rec = r0;
            uint32_t P = 0;
            if (npx > 1) P = pixel_pick_finish(src, pk, (int)npx, r0, r1, rec);
            if (src.shape == OT_SRC_IMAGE_RGB) {  // color.random_wavelengths_from_srgb srgb.py:513-553
                const double c_r = rec.c_r, c_rg = rec.c_rg;
                const int prim = (choice < c_r) ? 0 : ((choice > c_rg) ? 2 : 1);
                // The reference stratifies the wavelengths of each primary over exactly the rays that got that primary
                // (srgb.py:549-551), which keeps the colour noise of an image far below 1 / sqrt(rays per primary).  The
                // stratified choice variable carries that for free: inside the sub-interval that selected the primary it is
                // itself a stratified uniform variable over that primary's rays (exactly so for equal pixel colours).
                const double lo = (prim == 0) ? 0.0 : ((prim == 1) ? c_r : c_rg);
                const double hi = (prim == 0) ? c_r : ((prim == 1) ? c_rg : 1.0);
                const double t = (hi > lo) ? ot_div(choice - lo, hi - lo) : 0.5;
                // round trip 3: the inverse cumulative spectrum of the primary around t (SourceDev::prim_inv)
                const double tm = t * (double)OT_PRIM_M;
                int m = (int)tm;
                m = m < 0 ? 0 : (m > OT_PRIM_M - 1 ? OT_PRIM_M - 1 : m);
                const double* inv = src.prim_inv + (size_t)prim * (OT_PRIM_M + 1) + m;
                wl_x0 = inv[0];
                wl_x1 = inv[1];
                wl_f = tm - (double)m;
            }
            // row = P / img_w through the host's reciprocal with an exact fix-up instead of an integer division
            uint32_t PY = (uint32_t)(((double)P + 0.5) * src.inv_img_w);
            if (PY * (uint32_t)src.img_w > P) PY--;
            if ((PY + 1) * (uint32_t)src.img_w <= P) PY++;
            const uint32_t PX = P - PY * (uint32_t)src.img_w;
            double xs = src.pos[0] - src.dim[0] / 2, ys = src.pos[1] - src.dim[1] / 2;
            p.x = src.px_w * ((double)PX + rx) + xs;
            p.y = src.px_h * ((double)PY + ry) + ys;
        }
    }
    o.p = p;

    // ---- orientation (ray_source.py:264-277) ----
    V3 s_or;
    V3 fy_conv = {0.0, 0.0, 0.0};
    bool have_fy = false;  // (wave-uniform: converging orientation)
    if (src.frame_uniform) {  // one base orientation for the whole source (host): constant, or a point source converging
        s_or.x = src.s[0];
        s_or.y = src.s[1];
        s_or.z = src.s[2];
    } else if (src.orientation == OT_OR_CONVERGING) {
        // s_or = d / |d|, and its divergence frame straight from d: sy = [1, 0, 0] x s_or / sqrt(1 - s_or_x^2) =
        // (0, -d_z, d_y) / sqrt(d_y^2 + d_z^2) -- two reciprocal square roots instead of two square roots and two reciprocals
        const V3 d = {src.conv_pos[0] - p.x, src.conv_pos[1] - p.y, src.conv_pos[2] - p.z};
        const double m2 = d.y * d.y + d.z * d.z;
        const double il = ot_rsqrt(d.x * d.x + m2);
        s_or.x = d.x * il;
        s_or.y = d.y * il;
        s_or.z = d.z * il;
        const double im = ot_rsqrt(m2);
        fy_conv = {0.0, -d.z * im, d.y * im};
        have_fy = true;
    } else if (src.orientation == OT_OR_ARRAY && src.s_or) {  // or_func(x, y) evaluated by the caller (:272-274)
        s_or.x = src.s_or[g.j];
        s_or.y = src.s_or[g.j + src.n_or];
        s_or.z = src.s_or[g.j + 2 * src.n_or];
    } else {
        s_or.x = src.s[0];
        s_or.y = src.s[1];
        s_or.z = src.s[2];
    }

    // ---- direction: the divergence sample in the frame around s_or (ray_source.py:339-351) ----
    V3 s = s_or;
    if (!image) divergence_sample();
    if (src.divergence != OT_DIV_NONE) {
        V3 sx, sy;  // sy = [1, 0, 0] x s_or, sx = s_or x sy
        if (src.frame_uniform) {
            sx = {src.fx[0], src.fx[1], src.fx[2]};
            sy = {src.fy[0], src.fy[1], src.fy[2]};
        } else if (have_fy) {
            sy = fy_conv;
            sx = cross3(s_or, sy);
        } else {
            double fa = ot_rsqrt(1 - s_or.x * s_or.x);
            sy = {0.0, -s_or.z * fa, s_or.y * fa};
            sx = cross3(s_or, sy);
        }
        s.x = ct * s_or.x + st * (ca * sx.x + sa * sy.x);
        s.y = ct * s_or.y + st * (ca * sx.y + sa * sy.y);
        s.z = ct * s_or.z + st * (ca * sx.z + sa * sy.z);
    }
    o.s = s;

    // ---- polarisation vector (ray_source.py:387-433) ----
    o.polx = o.poly = o.polz = 0.0;
    if (!image) polarization_angle();
    if (!no_pol) {
        double px = pcs, py = psn, pz = 0.0;
        if (s.z != 1) {
            double fa = ot_rcp3(ot_sqrt(1 - s.z * s.z) + 1e-16);
            V3 ps = {s.y * fa, -s.x * fa, 0.0};
            double A_ts = ps.x * px + ps.y * py;
            double A_tp = ps.y * px - ps.x * py;
            V3 pp_ = cross3(ps, s);
            px = ps.x * A_ts + pp_.x * A_tp;
            py = ps.y * A_ts + pp_.y * A_tp;
            pz = ps.z * A_ts + pp_.z * A_tp;
        }
        o.polx = px;
        o.poly = py;
        o.polz = pz;
    }
    if (IMAGES && src.shape == OT_SRC_IMAGE_RGB) wl = wl_x0 + wl_f * (wl_x1 - wl_x0);
    o.wl = (float)wl;
    return o;
}
