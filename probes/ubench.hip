// micro-benchmarks: dependent-chain latency and independent-stream throughput of the instructions the DP kernels use.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITERS 2000
template <int MODE> __global__ void k(double *out, long long *cyc, double a0, double b0) {
    __shared__ __attribute__((aligned(32))) double tab[64];
    if (threadIdx.x < 64) tab[threadIdx.x] = 0.001 * threadIdx.x;
    __syncthreads();
    double a = a0 + threadIdx.x * 1e-9, b = b0, c = b0 * 0.5, d = b0 * 0.25;
    double e = a * 1.1, f = a * 1.2, g = a * 1.3;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (MODE == 0) a = fma(a, b, c);                       // dependent fma chain
            if (MODE == 1) { a = fma(a, b, c); e = fma(e, b, c); f = fma(f, b, c); g = fma(g, b, c); }  // 4 independent
            if (MODE == 2) a = a + b;                              // dependent add
            if (MODE == 3) a = __builtin_fmax(a, b) + d;           // max + add dependent
            if (MODE == 4) a = (a > b) ? a + d : a - d;            // cmp + cndmask + add
            if (MODE == 5) { int idx = ((int) __double2loint(a)) & 3; a += tab[4 * idx]; }   // dependent LDS read b64
            if (MODE == 6) { int idx = ((int) __double2loint(a)) & 3; double4 q = *(const double4 *) &tab[4 * idx]; a += q.x + q.w; }
            if (MODE == 7) { int lo = __double2loint(a), hi = __double2hiint(a); lo = __builtin_amdgcn_update_dpp(0, lo, 0x13C, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x13C, 0xF, 0xF, false); a = __hiloint2double(hi, lo) + d; }
            if (MODE == 8) { a = a * b; }                          // dependent mul
            if (MODE == 9) { float x = (float) a; x = fmaf(x, 1.0001f, 0.5f); a = (double) x; }  // cvt round trip + f32 fma
            if (MODE == 10) { a = a + b; e = e + b; f = f + b; g = g + b; }
            if (MODE == 11) { int ia = __double2loint(a); ia = ia * 3 + 1; ia ^= (ia >> 3); a = __hiloint2double(__double2hiint(a), ia); } // int chain
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + e + f + g;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE> void run(const char *name, int ops_per_j, int blocks, int threads) {
    double *out; long long *cyc;
    hipMalloc(&out, sizeof(double) * blocks * threads); hipMalloc(&cyc, sizeof(long long) * blocks);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.0, 1.0000001);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.0, 1.0000001);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    double per = (double) c / (ITERS * 8.0);
    printf("%-34s blocks=%4d thr=%4d  memtime-ticks/iter=%7.2f  (%.2f per op)  wall=%.3f ms -> %.2f ns/iter\n", name, blocks, threads, per, per / ops_per_j, ms, ms * 1e6 / (ITERS * 8.0));
    hipFree(out); hipFree(cyc);
}
int main() {
    run<0>("dep fma_f64", 1, 1, 64);
    run<1>("4 indep fma_f64", 4, 1, 64);
    run<1>("4 indep fma_f64 x4 waves/SIMD", 4, 1024, 1024);
    run<2>("dep add_f64", 1, 1, 64);
    run<10>("4 indep add_f64", 4, 1, 64);
    run<10>("4 indep add_f64 x4 waves/SIMD", 4, 1024, 1024);
    run<8>("dep mul_f64", 1, 1, 64);
    run<3>("dep max+add", 2, 1, 64);
    run<4>("dep cmp+cndmask+add/sub", 1, 1, 64);
    run<5>("dep LDS read b64 (+cvt/and/add)", 1, 1, 64);
    run<6>("dep LDS read b128x2 (+adds)", 1, 1, 64);
    run<7>("dep dpp ror x2 + add", 1, 1, 64);
    run<9>("dep cvt f64->f32, fma32, cvt back", 1, 1, 64);
    run<11>("dep int mul/xor chain", 1, 1, 64);
    run<11>("int chain x4 waves/SIMD", 1, 1024, 1024);
    return 0;
}
