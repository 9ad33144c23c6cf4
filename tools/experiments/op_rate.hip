// Experiment: issue rate of a few VALU operations on gfx950 (independent chains, one wave64 per SIMD x 8 waves).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
#define ITER 4096
template <int OP>
__global__ __launch_bounds__(256) void k(unsigned* out, unsigned seed) {
    unsigned a = threadIdx.x + seed, b = a * 3u + 1u, c = a ^ 0x9e3779b9u, d = a + 77u;
    double fa = a * 1e-9 + 1.0, fb = fa + 0.5, fc = fa + 0.25, fd = fa + 0.125;
    for (int i = 0; i < ITER; i++) {
        if (OP == 1) { a = __umulhi(a, 0xD2511F53u) ^ 1u; b = __umulhi(b, 0xCD9E8D57u) ^ 3u; c = __umulhi(c, 0x9E3779B9u) ^ 5u; d = __umulhi(d, 0xBB67AE85u) ^ 7u; }
        if (OP == 2) { a = (a ^ (a >> 7)) + 1u; b = (b ^ (b >> 5)) + 3u; c = (c ^ (c >> 9)) + 5u; d = (d ^ (d >> 3)) + 7u; }
        if (OP == 3) { fa = fa * 1.0000001 + 0.5; fb = fb * 0.9999999 + 0.25; fc = fc * 1.0000002 + 0.125; fd = fd * 0.9999998 + 0.0625; }
        if (OP == 4) { fa = __builtin_amdgcn_rsq(fa) + 1.5; fb = __builtin_amdgcn_rsq(fb) + 1.25; fc = __builtin_amdgcn_rsq(fc) + 1.125; fd = __builtin_amdgcn_rsq(fd) + 1.0625; }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a ^ b ^ c ^ d ^ (unsigned)(fa + fb + fc + fd);
}
int main() {
    unsigned* out; int blocks = 256 * 8;  // 8 workgroups of 4 waves per CU
    CHECK(hipMalloc(&out, blocks * 256 * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto f, double ops_per_iter) {
        f(); CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0)); for (int i = 0; i < 5; i++) f(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
        double waves = blocks * 4.0, instr = waves * ITER * ops_per_iter;  // wave-instructions
        printf("%-34s %.3f ms  %.2f G wave-instr/s  (%.2f per SIMD-ns)\n", name, ms, instr / ms / 1e6, instr / ms / 1e6 / 1024);
    };
    run("v_mul_hi_u32 + xor (8 ops/iter)", [&] { hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, 1u); }, 8);
    run("xor-shift-add (12 ops/iter)", [&] { hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, out, 1u); }, 12);
    run("f64 mul + add (8 ops/iter)", [&] { hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, out, 1u); }, 8);
    run("v_rsq_f64 + add (8 ops/iter)", [&] { hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, out, 1u); }, 8);
    return 0;
}
