// probe: can a SIMD issue a VALU instruction of one wave and a SALU / second VALU instruction of another wave in the
// same 4-cycle slot, or is instruction issue the shared limit?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
#define REP16(x) x x x x x x x x x x x x x x x x
// mode bits per wave role: role A = waves 0..3 of the block, role B = waves 4..7
// kind: 0 idle, 1 VALU f64 fma (independent x4), 2 SALU adds, 3 VALU f32 (independent), 4 LDS reads
__device__ __forceinline__ void run_kind(int kind, int iters, double *sink, int *isink) {
    if (kind == 1) {
        double a = threadIdx.x, b = 1.0001, c = 0.5, d = 2.0, e = 3.0;
        for (int i = 0; i < iters; i++) {
            REP16(asm volatile("v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4" : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b));)
        }
        sink[threadIdx.x] = a + c + d + e;
    } else if (kind == 2) {
        int s = iters, t = 1;
        for (int i = 0; i < iters; i++) {
            REP16(asm volatile("s_add_i32 %0, %0, %1\n s_xor_b32 %1, %1, %0\n s_add_i32 %0, %0, %1\n s_xor_b32 %1, %1, %0" : "+s"(s), "+s"(t) : : "scc");)
        }
        if (threadIdx.x == 0) isink[0] = s + t;
    } else if (kind == 5) {
        double a = threadIdx.x, b = 1.0001, c = 0.5, d = 2.0, e = 3.0;
        int s = iters, t = 1;
        for (int i = 0; i < iters; i++) {
            REP16(asm volatile("v_fma_f64 %0, %0, %4, %4\n s_add_i32 %5, %5, %6\n v_fma_f64 %1, %1, %4, %4\n s_xor_b32 %6, %6, %5\n v_fma_f64 %2, %2, %4, %4\n s_add_i32 %5, %5, %6\n v_fma_f64 %3, %3, %4, %4\n s_xor_b32 %6, %6, %5" : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b), "s"(s), "s"(t) : "scc");)
        }
        sink[threadIdx.x] = a + c + d + e;
    } else if (kind == 3) {
        float a = threadIdx.x, b = 1.0001f, c = 0.5f, d = 2.0f, e = 3.0f;
        for (int i = 0; i < iters; i++) {
            REP16(asm volatile("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4" : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b));)
        }
        sink[threadIdx.x] = a + c + d + e;
    }
}
__global__ __launch_bounds__(1024) void k(int kindA, int kindB, int iters, double *sink, int *isink) {
    int w = threadIdx.x >> 6;
    int role_b = (w & 4) ? 1 : 0;  // waves 0-3 role A, 4-7 role B, 8-11 A, 12-15 B
    run_kind(role_b ? kindB : kindA, iters, sink + blockIdx.x * 1024, isink);
}
int main() {
    double *sink; int *isink;
    CK(hipMalloc(&sink, 8 * 1024 * 256)); CK(hipMalloc(&isink, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20000;  // x 64 instructions
    struct { int a, b, threads; const char *name; } cases[] = {
        {1, 0, 512, "1 wave/SIMD: f64 fma"}, {2, 0, 512, "1 wave/SIMD: salu"}, {3, 0, 512, "1 wave/SIMD: f32 fma"},
        {1, 1, 512, "2 waves/SIMD: f64 + f64"}, {1, 2, 512, "2 waves/SIMD: f64 + salu"}, {2, 2, 512, "2 waves/SIMD: salu + salu"},
        {3, 3, 512, "2 waves/SIMD: f32 + f32"}, {5, 0, 512, "1 wave/SIMD: f64 and salu interleaved (128 instr)"},
        {5, 5, 512, "2 waves/SIMD: interleaved x2"}, {5, 5, 1024, "4 waves/SIMD: interleaved x4"}, {1, 1, 1024, "4 waves/SIMD: f64 x4"}, {1, 2, 1024, "4 waves/SIMD: 2 f64 + 2 salu"},
    };
    for (auto &c : cases) {
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k, dim3(256), dim3(c.threads), 0, 0, c.a, c.b, iters, sink, isink);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("%-34s %8.3f ms  -> %.2f ns per instruction per wave\n", c.name, ms, ms * 1e6 / (iters * 64.0));
        }
    }
    return 0;
}
