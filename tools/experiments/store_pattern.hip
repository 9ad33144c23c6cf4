// Experiment: HBM store throughput of the RayStorage write pattern (planar F-order) vs a wave-tiled layout.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int WORK>
__global__ __launch_bounds__(256) void planar(double* p, double* n, float* w, float* pol, long N, int nt) {
    long ray = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (ray >= N) return;
    double a = (double)ray * 1e-9, b = 1.0 + a, c = 2.0 - a;
    for (int sec = 0; sec < nt; sec++) {
#pragma unroll 1
        for (int k = 0; k < WORK; k++) { a = a * 1.0000001 + b; b = b * 0.9999999 + c; c = c * 1.0000002 - a; }
        p[ray + N * sec] = a; p[ray + N * (sec + nt)] = b; p[ray + N * (sec + 2L * nt)] = c;
        n[ray + N * sec] = a + b; w[ray + N * sec] = (float)c;
        pol[ray + N * sec] = (float)a; pol[ray + N * (sec + nt)] = (float)b; pol[ray + N * (sec + 2L * nt)] = (float)c;
    }
}

// planar with non-temporal (streaming) stores: written once, never read by this kernel
template <int WORK>
__global__ __launch_bounds__(256) void planar_nt(double* p, double* n, float* w, float* pol, long N, int nt) {
    long ray = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (ray >= N) return;
    double a = (double)ray * 1e-9, b = 1.0 + a, c = 2.0 - a;
    for (int sec = 0; sec < nt; sec++) {
#pragma unroll 1
        for (int k = 0; k < WORK; k++) { a = a * 1.0000001 + b; b = b * 0.9999999 + c; c = c * 1.0000002 - a; }
        __builtin_nontemporal_store(a, &p[ray + N * sec]);
        __builtin_nontemporal_store(b, &p[ray + N * (sec + nt)]);
        __builtin_nontemporal_store(c, &p[ray + N * (sec + 2L * nt)]);
        __builtin_nontemporal_store(a + b, &n[ray + N * sec]);
        __builtin_nontemporal_store((float)c, &w[ray + N * sec]);
        __builtin_nontemporal_store((float)a, &pol[ray + N * sec]);
        __builtin_nontemporal_store((float)b, &pol[ray + N * (sec + nt)]);
        __builtin_nontemporal_store((float)c, &pol[ray + N * (sec + 2L * nt)]);
    }
}

// tile = 64 rays x nt sections x (4 f64 + 4 f32) = 64*nt*48 bytes, contiguous per wave
template <int WORK>
__global__ __launch_bounds__(256) void tiled(char* buf, long N, int nt) {
    long ray = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (ray >= N) return;
    long tile = ray >> 6; int lane = ray & 63;
    char* base = buf + tile * (long)nt * 64 * 48;
    double a = (double)ray * 1e-9, b = 1.0 + a, c = 2.0 - a;
    for (int sec = 0; sec < nt; sec++) {
#pragma unroll 1
        for (int k = 0; k < WORK; k++) { a = a * 1.0000001 + b; b = b * 0.9999999 + c; c = c * 1.0000002 - a; }
        double* d = (double*)(base + (long)sec * 64 * 48);
        d[lane] = a; d[64 + lane] = b; d[128 + lane] = c; d[192 + lane] = a + b;
        float* f = (float*)(d + 256);
        f[lane] = (float)c; f[64 + lane] = (float)a; f[128 + lane] = (float)b; f[192 + lane] = (float)c;
    }
}

// per-array tiles: each array keeps its own allocation, inside it a wave's 64 rays x nt sections are contiguous
// (p: [tile][sec][comp][lane], n / w: [tile][sec][lane], pol: [tile][sec][comp][lane])
template <int WORK>
__global__ __launch_bounds__(256) void semi(double* p, double* n, float* w, float* pol, long N, int nt) {
    long ray = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (ray >= N) return;
    long tile = ray >> 6; int lane = ray & 63;
    double* pt = p + tile * nt * 192; double* nn = n + tile * nt * 64; float* wt = w + tile * nt * 64; float* po = pol + tile * nt * 192;
    double a = (double)ray * 1e-9, b = 1.0 + a, c = 2.0 - a;
    for (int sec = 0; sec < nt; sec++) {
#pragma unroll 1
        for (int k = 0; k < WORK; k++) { a = a * 1.0000001 + b; b = b * 0.9999999 + c; c = c * 1.0000002 - a; }
        double* d = pt + sec * 192;
        d[lane] = a; d[64 + lane] = b; d[128 + lane] = c;
        nn[sec * 64 + lane] = a + b; wt[sec * 64 + lane] = (float)c;
        float* f = po + sec * 192;
        f[lane] = (float)a; f[64 + lane] = (float)b; f[128 + lane] = (float)c;
    }
}

int main() {
    long N = 10000000; int nt = 17;
    double *p, *n; float *w, *pol; char* buf;
    CHECK(hipMalloc(&p, (N + 64) * nt * 24)); CHECK(hipMalloc(&n, (N + 64) * nt * 8)); CHECK(hipMalloc(&w, (N + 64) * nt * 4)); CHECK(hipMalloc(&pol, (N + 64) * nt * 12));
    CHECK(hipMalloc(&buf, (N + 64) * nt * 48));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 g((N + 255) / 256), b(256);
    auto run = [&](const char* name, auto f) {
        f(); CHECK(hipDeviceSynchronize());
        hipEventRecord(e0); for (int i = 0; i < 5; i++) f(); hipEventRecord(e1); CHECK(hipEventSynchronize(e1));
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        printf("%-28s %.3f ms  %.0f GB/s\n", name, ms, N * nt * 48.0 / ms / 1e6);
    };
    run("planar work=0", [&] { hipLaunchKernelGGL(planar<0>, g, b, 0, 0, p, n, w, pol, N, nt); });
    run("planar non-temporal work=0", [&] { hipLaunchKernelGGL(planar_nt<0>, g, b, 0, 0, p, n, w, pol, N, nt); });
    run("planar non-temporal work=60", [&] { hipLaunchKernelGGL(planar_nt<60>, g, b, 0, 0, p, n, w, pol, N, nt); });
    run("planar non-temporal work=150", [&] { hipLaunchKernelGGL(planar_nt<150>, g, b, 0, 0, p, n, w, pol, N, nt); });
    run("tiled  work=0", [&] { hipLaunchKernelGGL(tiled<0>, g, b, 0, 0, buf, N, nt); });
    run("semi   work=0", [&] { hipLaunchKernelGGL(semi<0>, g, b, 0, 0, p, n, w, pol, N, nt); });
    run("semi   work=60", [&] { hipLaunchKernelGGL(semi<60>, g, b, 0, 0, p, n, w, pol, N, nt); });
    run("semi   work=150", [&] { hipLaunchKernelGGL(semi<150>, g, b, 0, 0, p, n, w, pol, N, nt); });
    run("planar work=60", [&] { hipLaunchKernelGGL(planar<60>, g, b, 0, 0, p, n, w, pol, N, nt); });
    run("tiled  work=60", [&] { hipLaunchKernelGGL(tiled<60>, g, b, 0, 0, buf, N, nt); });
    run("planar work=150", [&] { hipLaunchKernelGGL(planar<150>, g, b, 0, 0, p, n, w, pol, N, nt); });
    run("tiled  work=150", [&] { hipLaunchKernelGGL(tiled<150>, g, b, 0, 0, buf, N, nt); });
    return 0;
}
