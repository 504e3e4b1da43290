// probe: v_mov_b32_dpp wave_shr:1 / wave_shl:1 on gfx950 -- lanes without a source keep the destination's old value
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int *o) {
    int lane = threadIdx.x;
    int v = lane * 3, old = 1000 + lane;
    o[lane] = __builtin_amdgcn_update_dpp(old, v, 0x138, 0xF, 0xF, false);       // wave_shr:1
    o[64 + lane] = __builtin_amdgcn_update_dpp(old, v, 0x130, 0xF, 0xF, false);  // wave_shl:1
}
int main() {
    int *d, h[128];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++) {
        int e_shr = l == 0 ? 1000 : (l - 1) * 3, e_shl = l == 63 ? 1063 : (l + 1) * 3;
        if (h[l] != e_shr || h[64 + l] != e_shl) { bad++; printf("lane %d: shr %d (want %d) shl %d (want %d)\n", l, h[l], e_shr, h[64 + l], e_shl); }
    }
    printf(bad ? "DPP SHIFT PROBE FAILED\n" : "dpp wave_shr/wave_shl ok\n");
    return bad != 0;
}
