// Does the NUMA node a page-locked block lands on decide the device-to-host rate?  For every node of the host: the calling thread
// is bound to the node's CPUs, hipHostMalloc takes 2 GB, a device buffer is copied into it five times.
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/pinned_numa_probe probes/pinned_numa_probe.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include <sys/syscall.h>
static double now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static int node_cpus(int node, cpu_set_t *set) {
    char path[128], buf[4096];
    snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    if (!fgets(buf, sizeof(buf), f)) { fclose(f); return -1; }
    fclose(f);
    CPU_ZERO(set);
    int n = 0;
    for (char *tok = strtok(buf, ",\n"); tok; tok = strtok(NULL, ",\n")) {
        int a, b;
        if (sscanf(tok, "%d-%d", &a, &b) == 2) { for (int c = a; c <= b; c++) { CPU_SET(c, set); n++; } }
        else if (sscanf(tok, "%d", &a) == 1) { CPU_SET(a, set); n++; }
    }
    return n;
}
int main() {
    const size_t bytes = (size_t) 2 << 30;
    void *d = nullptr;
    if (hipMalloc(&d, bytes) != hipSuccess) return 1;
    hipMemset(d, 1, bytes);
    char pci[64] = "";
    hipDeviceGetPCIBusId(pci, sizeof(pci), 0);
    char path[160];
    for (char *p = pci; *p; p++) if (*p >= 'A' && *p <= 'F') *p += 32;
    snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", pci);
    FILE *f = fopen(path, "r");
    int gnode = -2;
    if (f) { if (fscanf(f, "%d", &gnode) != 1) gnode = -2; fclose(f); }
    printf("device 0 at %s, numa_node %d\n", pci, gnode);
    cpu_set_t old;
    sched_getaffinity(0, sizeof(old), &old);
    for (int node = 0; node < 8; node++) {
        cpu_set_t set;
        int n = node_cpus(node, &set);
        if (n <= 0) break;
        cpu_set_t both;
        CPU_AND(&both, &set, &old);
        if (CPU_COUNT(&both) == 0) { printf("node %d: none of its %d CPUs allowed\n", node, n); continue; }
        sched_setaffinity(0, sizeof(both), &both);
        void *h = nullptr;
        double t0 = now();
        if (hipHostMalloc(&h, bytes, hipHostMallocDefault) != hipSuccess) { printf("node %d: hipHostMalloc failed\n", node); continue; }
        double t1 = now();
        double best = 0;
        for (int r = 0; r < 5; r++) {
            double a = now();
            hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost);
            double b = now();
            double gbs = bytes / (b - a) / 1e9;
            if (gbs > best) best = gbs;
        }
        printf("node %d (%d CPUs, %d allowed): hipHostMalloc %.0f ms, device-to-host best of 5: %.1f GB/s\n", node, n, CPU_COUNT(&both), (t1 - t0) * 1e3, best);
        hipHostFree(h);
    }
    sched_setaffinity(0, sizeof(old), &old);
    // ... and with the MEMORY bound to a node (set_mempolicy MPOL_BIND), whatever CPU the thread runs on
    for (int node = 0; node < 2; node++) {
        unsigned long mask = 1ul << node;
        if (syscall(SYS_set_mempolicy, 2 /* MPOL_BIND */, &mask, 64ul) != 0) { printf("set_mempolicy(node %d) failed\n", node); continue; }
        void *h = nullptr;
        if (hipHostMalloc(&h, bytes, hipHostMallocDefault) != hipSuccess) { printf("memory on node %d: hipHostMalloc failed\n", node); continue; }
        syscall(SYS_set_mempolicy, 0 /* MPOL_DEFAULT */, nullptr, 0ul);
        double best = 0;
        for (int r = 0; r < 5; r++) {
            double a = now();
            hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost);
            double g = bytes / (now() - a) / 1e9;
            if (g > best) best = g;
        }
        printf("memory bound to node %d: device-to-host best of 5: %.1f GB/s\n", node, best);
        hipHostFree(h);
    }
    return 0;
}
