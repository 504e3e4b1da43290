#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int *out) {
    int lane = threadIdx.x;
    int v = lane + 100;
    int a = __builtin_amdgcn_update_dpp(-1, v, 0x13C, 0xF, 0xF, false);  // wave_ror:1
    int b = __builtin_amdgcn_update_dpp(-1, v, 0x134, 0xF, 0xF, false);  // wave_rol:1
    int c = __builtin_amdgcn_update_dpp(-1, v, 0x138, 0xF, 0xF, false);  // wave_shr:1
    int d = __builtin_amdgcn_update_dpp(-1, v, 0x130, 0xF, 0xF, false);  // wave_shl:1
    out[lane] = a; out[64 + lane] = b; out[128 + lane] = c; out[192 + lane] = d;
}
int main() {
    int *d; hipMalloc(&d, 256 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    int h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char *n[4] = {"wave_ror1", "wave_rol1", "wave_shr1", "wave_shl1"};
    for (int r = 0; r < 4; r++) { printf("%s:", n[r]); for (int i = 0; i < 64; i += 1) if (i<4||i>59||(i>14&&i<18)||(i>30&&i<34)) printf(" [%d]=%d", i, h[r*64+i]); printf("\n"); }
    return 0;
}
