// Does the device-to-host rate into a page-locked block depend on what else is page-locked?  Blocks of 8, 5, 2 GB are taken one
// after the other (all alive), 2 GB are copied into each; then the first is freed and the others are measured again.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <time.h>
static double now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static double rate(void *h, void *d, size_t bytes, size_t piece) {
    double best = 0;
    hipStream_t s; hipStreamCreate(&s);
    for (int r = 0; r < 3; r++) {
        double a = now();
        for (size_t o = 0; o < bytes; o += piece)
            hipMemcpyAsync((char *) h + o, (char *) d + o, piece < bytes - o ? piece : bytes - o, hipMemcpyDeviceToHost, s);
        hipStreamSynchronize(s);
        double g = bytes / (now() - a) / 1e9;
        if (g > best) best = g;
    }
    hipStreamDestroy(s);
    return best;
}
int main() {
    const size_t G = (size_t) 1 << 30;
    void *d = nullptr;
    if (hipMalloc(&d, 2 * G) != hipSuccess) return 1;
    hipMemset(d, 1, 2 * G);
    size_t sizes[3] = {8 * G, 5 * G, 2 * G};
    void *h[3];
    for (int i = 0; i < 3; i++) {
        double t0 = now();
        if (hipHostMalloc(&h[i], sizes[i], hipHostMallocDefault) != hipSuccess) { printf("alloc %d failed\n", i); return 1; }
        printf("block %d (%zu GB) pinned in %.0f ms; 2 GB into its start: %.1f GB/s in one copy, %.1f GB/s in 445 MB pieces; into its end: %.1f GB/s\n", i, sizes[i] / G,
               (now() - t0) * 1e3, rate(h[i], d, 2 * G, 2 * G), rate(h[i], d, 2 * G, (size_t) 445 << 20),
               rate((char *) h[i] + sizes[i] - 2 * G, d, 2 * G, 2 * G));
    }
    hipHostFree(h[0]);
    for (int i = 1; i < 3; i++) printf("after freeing block 0: block %d: %.1f GB/s\n", i, rate(h[i], d, 2 * G, 2 * G));
    return 0;
}
