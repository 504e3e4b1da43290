// probe: write throughput of many sequential per-wave streams (the forward sweep's store pattern), no compute
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
// mode 0: 43 lanes x 8 B per row, rows packed (344 B apart); mode 1: 64 lanes x 16 B per 2 rows (1 KB slots);
// mode 2: like 0 but rows 512 B apart; mode 3: 43 lanes x 8 B, two rows per iteration issued back to back
// mode 3/4: the forward kernel's shape -- request next row's input (small L2-resident table), compute on the current
// one, store; mode 4 skips the store.  Shows what the in-order vmcnt costs when loads and stores are mixed.
__global__ __launch_bounds__(256) void k2(double *base, const double *tab, long long stride_bytes, int rows, int mode, int spin) {
    int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    int lane = threadIdx.x & 63;
    char *p = (char *) base + (long long) wave * stride_bytes;
    const double *t = tab + (wave & 255) * 4096;
    double v = lane;
    double nxt = t[lane];
    for (int r = 0; r < rows; r++) {
        double cur = nxt;
        nxt = t[((r + 1) * 3 + lane) & 4095];
        v += cur;
        for (int s = 0; s < spin; s++) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(v));
        if (mode == 3 && lane < 43) *(double *) (p + (long long) r * 344 + lane * 8) = v;
    }
    if (mode == 4 && v == 1.2345) *(double *) p = v;
}
__global__ __launch_bounds__(256) void k(double *base, long long stride_bytes, int rows, int mode, int spin) {
    int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    int lane = threadIdx.x & 63;
    char *p = (char *) base + (long long) wave * stride_bytes;
    double v = lane;
    for (int r = 0; r < rows; r++) {
        for (int s = 0; s < spin; s++) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(v));
        if (mode == 0) {
            if (lane < 43) *(double *) (p + (long long) r * 344 + lane * 8) = v;
        } else if (mode == 1) {
            if ((r & 1) == 1) *(double2 *) (p + (long long) (r >> 1) * 1024 + lane * 16) = make_double2(v, v);
        } else if (mode == 2) {
            if (lane < 43) *(double *) (p + (long long) r * 512 + lane * 8) = v;
        }
    }
}
int main() {
    const int rows = 8300;
    for (int waves : {2048, 8192}) {
        long long stride = (long long) rows * 512 + 4096;
        double *buf;
        CK(hipMalloc(&buf, (size_t) stride * waves));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int spin : {0, 40}) for (int mode = 0; mode < 3; mode++) {
            for (int rep = 0; rep < 2; rep++) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(k, dim3(waves / 4), dim3(256), 0, 0, buf, stride, rows, mode, spin);
                CK(hipEventRecord(e1));
                CK(hipDeviceSynchronize());
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                double bytes = mode == 1 ? (double) waves * (rows / 2) * 1024 : (double) waves * rows * 344;
                if (rep) printf("waves %5d spin %2d mode %d: %7.3f ms  %6.2f TB/s payload, %.1f ns per row per wave\n", waves, spin, mode, ms, bytes / ms / 1e9, ms * 1e6 / rows);
            }
        }
        double *tab; CK(hipMalloc(&tab, 8 * 4096 * 256)); CK(hipMemset(tab, 0, 8 * 4096 * 256));
        for (int spin : {10, 40}) for (int mode = 3; mode < 5; mode++) {
            for (int rep = 0; rep < 2; rep++) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(k2, dim3(waves / 4), dim3(256), 0, 0, buf, tab, stride, rows, mode, spin);
                CK(hipEventRecord(e1));
                CK(hipDeviceSynchronize());
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep) printf("waves %5d spin %2d mode %d (load+compute%s): %7.3f ms, %.1f ns per row per wave\n", waves, spin, mode, mode == 3 ? "+store" : "", ms, ms * 1e6 / rows);
            }
        }
        CK(hipFree(tab));
        CK(hipFree(buf));
    }
    return 0;
}
