// probe: does a device-to-host copy on one stream overlap a long kernel queued on another stream?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
__global__ void spin(long long cycles, int *sink) {
    long long t0 = clock64();
    while (clock64() - t0 < cycles) { }
    if (sink && threadIdx.x == 12345) sink[0] = 1;
}
struct Hidden { int *p; };
__global__ void spin_hidden(long long cycles, Hidden hsink) {
    long long t0 = clock64();
    while (clock64() - t0 < cycles) { }
    if (hsink.p && threadIdx.x == 12345) hsink.p[0] = 1;
}
__global__ void kcopy(const uint4 *src, uint4 *dst, size_t n) {
    size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x;
    size_t stride = (size_t) gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = src[i];
}
int main() {
    size_t bytes = 54u << 20;
    char *d = nullptr, *h = nullptr;
    CK(hipMalloc(&d, bytes));
    CK(hipHostMalloc(&h, bytes, hipHostMallocDefault));
    CK(hipMemset(d, 1, bytes));
    hipStream_t a, b;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    hipEvent_t e0, e1, c0, c1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&c0)); CK(hipEventCreate(&c1));
    for (int variant = 0; variant < 4; variant++) {
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0, a));
            // grid that fills the chip (variant 2: only a quarter of the chip)
            int blocks = variant == 2 ? 64 : 256 * 8;
            hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, a, 10000000ll, (int *) nullptr);  // ~5 ms at 100 MHz clock64? measured below
            CK(hipEventRecord(e1, a));
            CK(hipEventRecord(c0, b));
            if (variant == 0 || variant == 2) CK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, b));
            if (variant == 1) CK(hipMemcpyDtoHAsync(h, (hipDeviceptr_t) d, bytes, b));
            if (variant == 3) hipLaunchKernelGGL(kcopy, dim3(64), dim3(256), 0, b, (const uint4 *) d, (uint4 *) h, bytes / 16);
            CK(hipEventRecord(c1, b));
            CK(hipDeviceSynchronize());
            float k = 0, cs = 0, ce = 0;
            CK(hipEventElapsedTime(&k, e0, e1));
            CK(hipEventElapsedTime(&cs, e0, c0));
            CK(hipEventElapsedTime(&ce, e0, c1));
            printf("variant %d rep %d: kernel %.3f ms; copy window %.3f -> %.3f ms\n", variant, rep, k, cs, ce);
        }
    }
    for (int variant = 4; variant < 6; variant++) {
        // the spinning kernel is handed the copy's source buffer: as a plain pointer argument (4) or inside a struct (5)
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0, a));
            Hidden hs; hs.p = (int *) d;
            if (variant == 4) hipLaunchKernelGGL(spin, dim3(2048), dim3(256), 0, a, 10000000ll, (int *) d);
            else hipLaunchKernelGGL(spin_hidden, dim3(2048), dim3(256), 0, a, 10000000ll, hs);
            CK(hipEventRecord(e1, a));
            CK(hipEventRecord(c0, b));
            CK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, b));
            CK(hipEventRecord(c1, b));
            CK(hipDeviceSynchronize());
            float k = 0, cs = 0, ce = 0;
            CK(hipEventElapsedTime(&k, e0, e1));
            CK(hipEventElapsedTime(&cs, e0, c0));
            CK(hipEventElapsedTime(&ce, e0, c1));
            printf("variant %d rep %d: kernel %.3f ms; copy window %.3f -> %.3f ms\n", variant, rep, k, cs, ce);
        }
    }
    // copy alone
    CK(hipEventRecord(c0, b));
    CK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, b));
    CK(hipEventRecord(c1, b));
    CK(hipDeviceSynchronize());
    float t = 0;
    CK(hipEventElapsedTime(&t, c0, c1));
    printf("copy alone: %.3f ms (%.1f GB/s)\n", t, bytes / t / 1e6);
    CK(hipEventRecord(c0, b));
    hipLaunchKernelGGL(kcopy, dim3(64), dim3(256), 0, b, (const uint4 *) d, (uint4 *) h, bytes / 16);
    CK(hipEventRecord(c1, b));
    CK(hipDeviceSynchronize());
    CK(hipEventElapsedTime(&t, c0, c1));
    printf("kernel copy alone (64 blocks): %.3f ms (%.1f GB/s)\n", t, bytes / t / 1e6);
    return 0;
}
