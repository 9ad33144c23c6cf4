#include "../../optrace_amd/csrc/ot_trace.hpp"
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double* in, double* out, int n) {
    int i = blockIdx.x * 256 + threadIdx.x; if (i >= n) return;
    const double* x = in + 9 * i;
    V3 s = {x[0], x[1], x[2]}, s_ = {x[3], x[4], x[5]};
    RayState r; r.polx = (float)x[6]; r.poly = (float)x[7]; r.polz = (float)x[8];
    float a = r.polx, b = r.poly, c = r.polz; double A, B;
    compute_polarization<true>(s, s_, r, a, b, c, A, B);
    double* o = out + 5 * i; o[0] = a; o[1] = b; o[2] = c; o[3] = A; o[4] = B;
}
static void nrm(double* v) { double l = std::sqrt(v[0]*v[0]+v[1]*v[1]+v[2]*v[2]); v[0]/=l; v[1]/=l; v[2]/=l; }
static void cross(const double* a, const double* b, double* o) { o[0]=a[1]*b[2]-a[2]*b[1]; o[1]=a[2]*b[0]-a[0]*b[2]; o[2]=a[0]*b[1]-a[1]*b[0]; }
int main() {
    int n = 100000; std::vector<double> in(9*n), out(5*n);
    srand(1);
    auto rnd = [] { return rand() / (double)RAND_MAX * 2 - 1; };
    for (int i = 0; i < n; i++) {
        double* x = &in[9*i];
        double s[3] = {rnd()*0.4, rnd()*0.4, 1.0}; nrm(s);
        double nn[3] = {rnd()*0.6, rnd()*0.6, 1.0}; nrm(nn);
        double N = 1/1.6, ns = nn[0]*s[0]+nn[1]*s[1]+nn[2]*s[2], W = std::sqrt(1-N*N*(1-ns*ns)), q = N*ns-W;
        double s_[3] = {s[0]*N-nn[0]*q, s[1]*N-nn[1]*q, s[2]*N-nn[2]*q};
        double t[3] = {rnd(), rnd(), rnd()}, p[3]; cross(s, t, p); nrm(p);
        for (int c = 0; c < 3; c++) { x[c] = s[c]; x[3+c] = s_[c]; x[6+c] = (double)(float)p[c]; }
    }
    double *din, *dout; hipMalloc(&din, in.size()*8); hipMalloc(&dout, out.size()*8);
    hipMemcpy(din, in.data(), in.size()*8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3((n+255)/256), dim3(256), 0, 0, din, dout, n);
    hipMemcpy(out.data(), dout, out.size()*8, hipMemcpyDeviceToHost);
    double ea = 0, eb = 0, ep = 0, dn_gpu = 0, dn_ref = 0; int nflip = 0;
    for (int i = 0; i < n; i++) {
        const double* x = &in[9*i]; const double *s = x, *s_ = x+3, *pol = x+6;
        double ps[3], pp[3], pp_[3]; cross(s_, s, ps); nrm(ps); cross(ps, s, pp); cross(ps, s_, pp_);
        double Ats = ps[0]*pol[0]+ps[1]*pol[1]+ps[2]*pol[2], Atp = pp[0]*pol[0]+pp[1]*pol[1]+pp[2]*pol[2];
        const double* o = &out[5*i];
        ea = fmax(ea, fabs(o[3] - Ats*Ats)); eb = fmax(eb, fabs(o[4] - Atp*Atp));
        double ng = 0, nr = 0, np_ = 0;
        for (int c = 0; c < 3; c++) { double rf = (double)(float)(ps[c]*Ats + pp_[c]*Atp); ep = fmax(ep, fabs(o[c] - rf)); nflip += (o[c] != rf); ng += o[c]*o[c]; nr += rf*rf; np_ += pol[c]*pol[c]; }
        dn_gpu += ng - np_; dn_ref += nr - np_;
    }
    printf("max err A_ts2 %.3e  A_tp2 %.3e  pol' %.3e flips %d mean dnorm2 gpu %.3e ref %.3e\n", ea, eb, ep, nflip, dn_gpu/n, dn_ref/n);
    return 0;
}
