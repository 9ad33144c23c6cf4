// Exactness harness for the division / square-root cores of the tracing loop (ot_device.hpp: ot_div, ot_rcp3 + ot_div_r,
// ot_sqrt, normalize3).  Bit-exact hit masks rest on these cores returning the bits of IEEE `/` and sqrt for every
// operand the path can produce; this kernel checks that claim on the device itself: random operands of a chosen class go
// through the core and through the compiler's own IEEE sequence (the same translation unit, -ffp-contract=off), the
// result bits are compared, mismatches are counted and the first one is kept.
// Driven by tests/test_gpu_arith_exact.py through ot_selftest_arith / ot_selftest_eval.
#pragma once
#include "ot_device.hpp"

#define OT_ST_DIV 0        // ot_div(a, b)                      against a / b
#define OT_ST_SQRT 1       // ot_sqrt(|a|)                      against sqrt(|a|)
#define OT_ST_NORMALIZE 2  // normalize3(a, b, c)               against (a, b, c) / sqrt(a a + b b + c c), component-wise
#define OT_ST_DIV_SHARED 3 // ot_div_r(a, c, r), ot_div_r(b, c, r) with r = ot_rcp3(c)   against a / c, b / c

#define OT_CLS_WIDE 0      // uniform mantissa, exponent uniform in +-500 (sqrt: -760 .. 1020; normalize3: +-250), random sign
#define OT_CLS_GEOMETRY 1  // millimetre geometry: magnitude 2^-20 .. 2^14 (log-uniform), random sign
#define OT_CLS_INDEX 2     // refractive indices: uniform in [1, 2.5]
#define OT_CLS_COSINE 3    // direction cosines near 0 and near 1: 2^-k m  or  1 - 2^-k m / 2,  k = 0 .. 60, m in [1, 2)

OT_DEV double st_make(uint64_t bits, int e_lo, int e_hi, bool sign_rnd) {  // mantissa from bits, exponent in [e_lo, e_hi]
    const uint64_t mant = bits & 0xfffffffffffffull;
    const uint32_t r = (uint32_t)(bits >> 52);  // 12 more random bits
    const int span = e_hi - e_lo + 1;
    const int e = e_lo + (int)((r & 0x7ffu) * (uint32_t)span >> 11);
    const uint64_t sign = sign_rnd ? ((uint64_t)(r >> 11) & 1ull) << 63 : 0ull;
    const uint64_t u = sign | ((uint64_t)(e + 1023) << 52) | mant;
    return __longlong_as_double((long long)u);
}

OT_DEV double st_operand(uint64_t bits, int cls, int op) {
    switch (cls) {
        case OT_CLS_WIDE:
            if (op == OT_ST_SQRT) return st_make(bits, -760, 1020, false);
            if (op == OT_ST_NORMALIZE) return st_make(bits, -250, 250, true);
            return st_make(bits, -500, 500, true);
        case OT_CLS_GEOMETRY: return st_make(bits, -20, 14, op != OT_ST_SQRT);
        case OT_CLS_INDEX: {
            const double u = (double)(bits >> 11) * 0x1.0p-53;
            return 1.0 + 1.5 * u;
        }
        default: {  // OT_CLS_COSINE
            const double m = st_make(bits, 0, 0, false);  // [1, 2)
            const int k = (int)((bits >> 52) & 0x3f);
            const double small = ldexp(m, -(k < 61 ? k : 60));
            return ((bits >> 58) & 1) ? 1.0 - 0.5 * small : small;
        }
    }
}

OT_DEV bool st_same(double a, double b) {
    return __double_as_longlong(a) == __double_as_longlong(b) || (a != a && b != b);
}

// core and IEEE result of one operand set; n_out results each
OT_DEV int st_eval(int op, double a, double b, double c, double* core, double* ieee) {
    switch (op) {
        case OT_ST_DIV:
            core[0] = ot_div(a, b);
            ieee[0] = a / b;
            return 1;
        case OT_ST_SQRT:
            core[0] = ot_sqrt(fabs(a));
            ieee[0] = sqrt(fabs(a));
            return 1;
        case OT_ST_NORMALIZE: {
            const V3 v = {a, b, c};
            const V3 n = normalize3(v);
            const double l = sqrt(a * a + b * b + c * c);  // misc.py:136
            core[0] = n.x, core[1] = n.y, core[2] = n.z;
            ieee[0] = a / l, ieee[1] = b / l, ieee[2] = c / l;
            return 3;
        }
        default: {
            const double r = ot_rcp3(c);
            core[0] = ot_div_r(a, c, r), core[1] = ot_div_r(b, c, r);
            ieee[0] = a / c, ieee[1] = b / c;
            return 2;
        }
    }
}

// out[0] += mismatching operand sets; the first one (lowest index is not guaranteed, any one) goes to bad[0..3] = a, b, c, core
__global__ __launch_bounds__(256) void selftest_arith_kernel(int op, int cls, int64_t n, uint64_t seed,
                                                             unsigned long long* __restrict__ out, double* __restrict__ bad) {
    unsigned long long mism = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        // one Philox block = two 64-bit words; a second block for the third operand
        Philox p = philox4x32((uint32_t)i, (uint32_t)(i >> 32), 0x53454c46u, (uint32_t)op, (uint32_t)seed, (uint32_t)(seed >> 32));
        const uint64_t w0 = ((uint64_t)p.c[1] << 32) | p.c[0], w1 = ((uint64_t)p.c[3] << 32) | p.c[2];
        double a = st_operand(w0, cls, op), b = st_operand(w1, cls, op), c = 1.0;
        if (op >= OT_ST_NORMALIZE) {
            Philox q = philox4x32((uint32_t)i, (uint32_t)(i >> 32), 0x53454c47u, (uint32_t)op, (uint32_t)seed, (uint32_t)(seed >> 32));
            c = st_operand(((uint64_t)q.c[1] << 32) | q.c[0], cls, op);
        }
        double core[3], ieee[3];
        const int k = st_eval(op, a, b, c, core, ieee);
        bool ok = true;
        for (int j = 0; j < k; j++) ok = ok && st_same(core[j], ieee[j]);
        if (!ok) {
            if (mism == 0 && atomicAdd(&out[1], 1ull) == 0ull) {
                bad[0] = a, bad[1] = b, bad[2] = c, bad[3] = core[0];
            }
            mism++;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mism += __shfl_xor(mism, o);
    if ((threadIdx.x & 63) == 0 && mism) atomicAdd(&out[0], mism);
}

// the cores and the IEEE operators on caller-supplied operands (special values: zeros, infinities, subnormals, NaN)
__global__ __launch_bounds__(256) void selftest_eval_kernel(int op, int64_t n, const double* __restrict__ a,
                                                            const double* __restrict__ b, const double* __restrict__ c,
                                                            double* __restrict__ core_out, double* __restrict__ ieee_out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double core[3] = {0, 0, 0}, ieee[3] = {0, 0, 0};
    st_eval(op, a[i], b ? b[i] : 1.0, c ? c[i] : 1.0, core, ieee);
    for (int j = 0; j < 3; j++) {
        core_out[i + j * n] = core[j];
        ieee_out[i + j * n] = ieee[j];
    }
}
