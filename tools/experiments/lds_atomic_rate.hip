// Experiment: rate of LDS atomic adds on gfx950 -- what bounds the accumulation kernels of the detector stage (four
// ds_add_f64 per hit into a 64 x 64 x 4 tile).  1024-thread workgroups, one per CU and two per CU; random addresses in a
// 128 KB tile; f64, f32, u32, u64 and plain ds_write_b64 for reference.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
#define ITER 2048
#define TILE 16384  // doubles (128 KB)
template <int OP>
__global__ __launch_bounds__(1024) void k(double* out, unsigned seed) {
    extern __shared__ double tile[];
    for (int i = threadIdx.x; i < TILE; i += 1024) tile[i] = 0.0;
    __syncthreads();
    unsigned s = (threadIdx.x + 1024u * blockIdx.x) * 2654435761u + seed;
    float* t32 = (float*)tile;
    unsigned* u32 = (unsigned*)tile;
    unsigned long long* u64 = (unsigned long long*)tile;
    for (int i = 0; i < ITER; i++) {
        s = s * 1664525u + 1013904223u;
        const unsigned px = (s >> 12) & (TILE / 4 - 1);  // a pixel: four consecutive doubles
        if (OP == 0) { __hip_atomic_fetch_add(&tile[4 * px + 0], 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); __hip_atomic_fetch_add(&tile[4 * px + 1], 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                       __hip_atomic_fetch_add(&tile[4 * px + 2], 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); __hip_atomic_fetch_add(&tile[4 * px + 3], 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
        if (OP == 1) { for (int c = 0; c < 4; c++) __hip_atomic_fetch_add(&t32[4 * px + c], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
        if (OP == 2) { for (int c = 0; c < 4; c++) __hip_atomic_fetch_add(&u32[4 * px + c], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
        if (OP == 3) { for (int c = 0; c < 4; c++) __hip_atomic_fetch_add(&u64[4 * px + c], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
        if (OP == 4) { for (int c = 0; c < 4; c++) tile[4 * px + c] = (double)i; }
        if (OP == 5) { for (int c = 0; c < 4; c++) __hip_atomic_fetch_add(&tile[px + c * (TILE / 4)], 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }  // planar tile
    }
    __syncthreads();
    double acc = 0;
    for (int i = threadIdx.x; i < TILE; i += 1024) acc += tile[i];
    out[blockIdx.x * 1024 + threadIdx.x] = acc;
}
int main() {
    double* out; const int blocks = 256 * 4;
    CHECK(hipMalloc(&out, blocks * 1024 * 8));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto kern) {
        CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, TILE * 8));
        auto f = [&] { hipLaunchKernelGGL(kern, dim3(blocks), dim3(1024), TILE * 8, 0, out, 1u); };
        f(); CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0)); for (int i = 0; i < 3; i++) f(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
        const double ops = (double)blocks * 1024 * ITER * 4;  // lane-operations
        printf("%-40s %.3f ms  %.1f G lane-ops/s  (%.2f per CU-clock at 2.4 GHz)\n", name, ms, ops / ms / 1e6, ops / ms / 1e6 / 256 / 2.4);
    };
    run("ds_add_f64, pixel-interleaved (shipped)", k<0>);
    run("ds_add_f64, planar tile", k<5>);
    run("ds_add_f32", k<1>);
    run("ds_add_u32", k<2>);
    run("ds_add_u64", k<3>);
    run("ds_write_b64", k<4>);
    return 0;
}
