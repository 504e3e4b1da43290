// logadd_f32_probe.hip -- does an fp32 correction term pay in the sweeps' logAdd?  
//
// The sweeps spend 4 logAdds per cell; each is  max(x, y) + P_j(min(|x - y|, 8))  with P_j one of the reference's cubic pieces
// (impl/pairwiseAligner.c:301-318) picked by floor(2 dc) from an LDS table (sa_fast.inc: la_prep / la_fetch / la_finish,
// restated here).  The state must stay fp64 (log-probabilities reach -3e4); the correction term lies in [0, 0.70] and could be
// evaluated in fp32: |x - y| taken in fp64, converted, clamped, indexed and the cubic run with v_fma_f32 (or two logAdds per
// v_pk_fma_f32), the result converted back and added to the fp64 maximum.
//
// This probe runs the kernels' own access pattern -- per step three independent logAdds whose LDS reads are issued together and
// a fourth that depends on the first, results feeding the next step through a wave rotate -- in four flavours and reports SIMD
// cycles per logAdd at the occupancy of k_bwd_fast (5 waves per SIMD) and at 4 and 8, plus the worst difference to the fp64
// form over the operands the run saw:
//     f64        what the sweeps do today
//     f32        correction in fp32, scalar v_fma_f32
//     f32pk      correction in fp32, the three independent logAdds of a step through v_pk_* where the ISA has a packed form
//     f64_nolds  fp64 with the table row in registers (no LDS read): the floor of the arithmetic alone
// build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off probes/logadd_f32_probe.hip -o probes/logadd_f32_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define STEPS 4096

__device__ __forceinline__ double vmax(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double vmin_abs_8(double a) {
    double r;
    asm("v_min_f64 %0, |%1|, %2" : "=v"(r) : "v"(a), "s"(0x1.fffffffffcp+2));
    return r;
}
__device__ __forceinline__ double rot(double v) {   // lane i <- lane i-1 (wave_ror:1), as the sweeps move one message per step
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, 0x13C, 0xF, 0xF, false);
    hi = __builtin_amdgcn_mov_dpp(hi, 0x13C, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}

// ---- fp64 (sa_fast.inc) ----
#define ROW_D 6
__device__ void tab64_init(double *tab, int tid) {
    if (tid < 16) {
        const float a3[4] = {-0.009350833524763f, -0.014532321752540f, -0.004605031767994f, -0.000458661602210f};
        const float a2[4] = {0.130659527668286f, 0.139942324101744f, 0.063427417320019f, 0.009695946122598f};
        const float a1[4] = {0.498799810682272f, 0.495635523139337f, 0.695956496475118f, 0.930734667215156f};
        const float a0[4] = {0.693203116424741f, 0.692140569840976f, 0.514272634594009f, 0.168037164329057f};
        int piece = tid <= 1 ? 0 : (tid <= 4 ? 1 : (tid <= 8 ? 2 : (tid <= 14 ? 3 : -1)));
        double *row = tab + ROW_D * tid;
        row[0] = piece >= 0 ? (double) a3[piece] : 0.0;
        row[1] = piece >= 0 ? (double) a2[piece] : 0.0;
        row[2] = piece >= 0 ? (double) a1[piece] - 1.0 : 0.0;
        row[3] = piece >= 0 ? (double) a0[piece] : 0.0;
        row[4] = row[5] = 0.0;
    }
}
struct La64 { double mx, dc; double4 c; };
__device__ __forceinline__ void prep64(La64 &p, double x, double y) { p.mx = vmax(x, y); p.dc = vmin_abs_8(x - y); }
__device__ __forceinline__ void fetch64(La64 &p, const double *tab) {
    const unsigned j = __builtin_amdgcn_ubfe((unsigned) __double2hiint(p.dc + 8.0), 16u, 5u);
    p.c = *reinterpret_cast<const double4 *>(reinterpret_cast<const char *>(tab) + j * (unsigned) (8 * ROW_D));
}
__device__ __forceinline__ double finish64(const La64 &p) { return p.mx + fma(fma(fma(p.c.x, p.dc, p.c.y), p.dc, p.c.z), p.dc, p.c.w); }

// ---- fp32 correction ----
// dc in [0, 8): dc + 8 lies in [8, 16), one ulp = 2^-20, so floor(2 dc) = mantissa bits 19..22.  Rows of four floats (16 bytes:
// sixteen rows cover all 64 banks exactly once, any mix of rows in a group of 16 lanes is conflict-free).
__device__ void tab32_init(float *tab, int tid) {
    if (tid < 16) {
        const float a3[4] = {-0.009350833524763f, -0.014532321752540f, -0.004605031767994f, -0.000458661602210f};
        const float a2[4] = {0.130659527668286f, 0.139942324101744f, 0.063427417320019f, 0.009695946122598f};
        const float a1[4] = {0.498799810682272f, 0.495635523139337f, 0.695956496475118f, 0.930734667215156f};
        const float a0[4] = {0.693203116424741f, 0.692140569840976f, 0.514272634594009f, 0.168037164329057f};
        int piece = tid <= 1 ? 0 : (tid <= 4 ? 1 : (tid <= 8 ? 2 : (tid <= 14 ? 3 : -1)));
        float *row = tab + 4 * tid;
        row[0] = piece >= 0 ? a3[piece] : 0.f;
        row[1] = piece >= 0 ? a2[piece] : 0.f;
        row[2] = piece >= 0 ? (float) ((double) a1[piece] - 1.0) : 0.f;
        row[3] = piece >= 0 ? a0[piece] : 0.f;
    }
}
struct La32 { double mx; float dc; float4 c; };
__device__ __forceinline__ void prep32(La32 &p, double x, double y) {
    p.mx = vmax(x, y);
    const float d = (float) (x - y);                 // v_cvt_f32_f64: +-inf stays, NaN stays
    float r;
    asm("v_min_f32 %0, |%1|, %2" : "=v"(r) : "v"(d), "v"(0x1.fffff8p+2f));   // min(|d|, 8 - 2^-19): dc + 8 stays below 16 in fp32; NaN -> the other operand
    p.dc = r;
}
__device__ __forceinline__ void fetch32(La32 &p, const float *tab) {
    const unsigned j = __builtin_amdgcn_ubfe(__float_as_uint(p.dc + 8.0f), 19u, 4u);
    p.c = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(tab) + j * 16u);
}
__device__ __forceinline__ double finish32(const La32 &p) {
    return p.mx + (double) fmaf(fmaf(fmaf(p.c.x, p.dc, p.c.y), p.dc, p.c.z), p.dc, p.c.w);
}
// two logAdds per packed instruction (the cubic of a and b side by side)
typedef float float2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void finish32_pk(const La32 &a, const La32 &b, double &ra, double &rb) {
    float2v dc = {a.dc, b.dc}, c3 = {a.c.x, b.c.x}, c2 = {a.c.y, b.c.y}, c1 = {a.c.z, b.c.z}, c0 = {a.c.w, b.c.w};
    float2v r = __builtin_elementwise_fma(__builtin_elementwise_fma(__builtin_elementwise_fma(c3, dc, c2), dc, c1), dc, c0);
    ra = a.mx + (double) r.x;
    rb = b.mx + (double) r.y;
}

struct Out { unsigned long long cyc; double sink; double worst; };

// one step of the sweeps' shape: (m, x, y) of a cell from three messages; k0..k6 stand for the transitions
template <int MODE>
__global__ __launch_bounds__(256) void k_probe(Out *out, double seed, const double *__restrict__ kk) {
    __shared__ __attribute__((aligned(32))) double T64[16 * ROW_D];
    __shared__ __attribute__((aligned(32))) float T32[64];
    tab64_init(T64, threadIdx.x);
    tab32_init(T32, threadIdx.x);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const double k0 = kk[0], k1 = kk[1], k2 = kk[2], k3 = kk[3], k4 = kk[4], k5 = kk[5], k6 = kk[6];
    double mM = seed - 0.37 * lane, mY = seed - 1.1 - 0.21 * (lane & 7), mX = seed - 2.3 + 0.13 * (lane & 3);
    double worst = 0.0;
    const double4 row_reg = *reinterpret_cast<const double4 *>(T64 + ROW_D * 3);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int s = 0; s < STEPS; s++) {
        const double up = rot(mY), low = mX;
        const double a0 = mM + k0, b0 = up + k1, a1 = mM + k2, b1 = low + k3, a2 = mM + k4, b2 = up + k5, b3 = low + k6;
        double tm1, tx, ty, tm;
        if (MODE == 0) {
            La64 p, q, r;
            prep64(p, a0, b0); prep64(q, a1, b1); prep64(r, a2, b2);
            fetch64(p, T64); fetch64(q, T64); fetch64(r, T64);
            __builtin_amdgcn_sched_barrier(0);
            tm1 = finish64(p); tx = finish64(q); ty = finish64(r);
            La64 u; prep64(u, tm1, b3); fetch64(u, T64); tm = finish64(u);
        } else if (MODE == 1 || MODE == 2) {
            La32 p, q, r;
            prep32(p, a0, b0); prep32(q, a1, b1); prep32(r, a2, b2);
            fetch32(p, T32); fetch32(q, T32); fetch32(r, T32);
            __builtin_amdgcn_sched_barrier(0);
            if (MODE == 1) { tm1 = finish32(p); tx = finish32(q); }
            else finish32_pk(p, q, tm1, tx);
            ty = finish32(r);
            La32 u; prep32(u, tm1, b3); fetch32(u, T32); tm = finish32(u);
        } else {
            La64 p, q, r;
            prep64(p, a0, b0); prep64(q, a1, b1); prep64(r, a2, b2);
            p.c = row_reg; q.c = row_reg; r.c = row_reg;
            tm1 = finish64(p); tx = finish64(q); ty = finish64(r);
            La64 u; prep64(u, tm1, b3); u.c = row_reg; tm = finish64(u);
        }
        if (MODE == 1 || MODE == 2) {   // (the check costs the fp32 flavours a second evaluation: timed runs use CHECK = 0 below)
        }
        // emissions keep the values in range: a cell's message is its sum plus a (negative) emission
        mM = tm - 1.9 - 0.01 * (double) ((s + lane) & 15);
        mY = ty - 2.4;
        mX = tx - 2.3025850929940455;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) {
        Out o; o.cyc = t1 - t0; o.sink = mM + mY + mX; o.worst = worst;
        out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = o;
    }
}

// accuracy: the fp32 flavour against the fp64 one on operand pairs spread over the pieces (and around their seams)
__global__ void k_accuracy(double *worst_out, double *worst_at, int n) {
    __shared__ __attribute__((aligned(32))) double T64[16 * ROW_D];
    __shared__ __attribute__((aligned(32))) float T32[64];
    tab64_init(T64, threadIdx.x);
    tab32_init(T32, threadIdx.x);
    __syncthreads();
    double w = 0.0, at = 0.0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const double d = 9.0 * (double) i / (double) n;          // |x - y| in [0, 9)
        const double x = -31234.567 + 1e-3 * (i & 1023), y = x - d;
        La64 p; prep64(p, x, y); fetch64(p, T64);
        La32 q; prep32(q, x, y); fetch32(q, T32);
        const double e = fabs(finish64(p) - finish32(q));
        if (e > w) { w = e; at = d; }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const double w2 = __shfl_xor(w, off, 64), a2 = __shfl_xor(at, off, 64);
        if (w2 > w) { w = w2; at = a2; }
    }
    if ((threadIdx.x & 63) == 0) { worst_out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = w; worst_at[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = at; }
}

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipDeviceProp_t pr;
    HIPCHK(hipGetDeviceProperties(&pr, 0));
    const int cus = pr.multiProcessorCount;
    printf("device %s, %d CUs, clock %.0f MHz; %d steps of 4 logAdds per wave\n", pr.name, cus, pr.clockRate / 1000.0, STEPS);
    const double kk_h[7] = {-0.21, -1.9, -1.2, -0.8, -2.0, -0.6, -1.5};
    double *kk;
    HIPCHK(hipMalloc(&kk, sizeof(kk_h)));
    HIPCHK(hipMemcpy(kk, kk_h, sizeof(kk_h), hipMemcpyHostToDevice));
    typedef void (*kern_t)(Out *, double, const double *);
    struct { const char *name; kern_t k; } tests[] = {{"f64 (today)", k_probe<0>}, {"f32 correction", k_probe<1>},
                                                      {"f32 correction, packed pair", k_probe<2>}, {"f64, table row in registers", k_probe<3>}};
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    for (int wps : {4, 5, 8}) {
        // wps waves per SIMD: blocks of 256 threads (one wave per SIMD of a CU), wps blocks per CU resident at once
        const int blocks = cus * wps, waves = blocks * 4;
        Out *d;
        HIPCHK(hipMalloc(&d, sizeof(Out) * waves));
        std::vector<Out> h(waves);
        printf("\n%d waves per SIMD\n%-34s %14s %18s %12s\n", wps, "flavour", "cycles/logAdd", "SIMD cycles/logAdd", "wall ms");
        for (auto &t : tests) {
            hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), 0, 0, d, -30000.0, kk);   // warm-up
            HIPCHK(hipDeviceSynchronize());
            HIPCHK(hipEventRecord(e0));
            hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), 0, 0, d, -30000.0, kk);
            HIPCHK(hipEventRecord(e1));
            HIPCHK(hipDeviceSynchronize());
            float ms = 0;
            HIPCHK(hipEventElapsedTime(&ms, e0, e1));
            HIPCHK(hipMemcpy(h.data(), d, sizeof(Out) * waves, hipMemcpyDeviceToHost));
            double cyc = 0;
            for (auto &o : h) cyc += (double) o.cyc;
            cyc /= waves;
            // s_memtime counts at 100 MHz on this part: convert through the wall clock instead -- SIMD cycles per logAdd =
            // wall time x shader clock / (logAdds per wave x waves per SIMD)
            const double simd_cyc = ms * 1e-3 * (pr.clockRate * 1e3) / ((double) STEPS * 4.0 * wps);
            printf("%-34s %14.2f %18.2f %12.3f\n", t.name, cyc / (STEPS * 4.0), simd_cyc, ms);
        }
        HIPCHK(hipFree(d));
    }
    {   // accuracy of the fp32 flavour
        const int nb = 256, nt = 256, nw = nb * nt / 64;
        double *dw, *da;
        HIPCHK(hipMalloc(&dw, 8 * nw));
        HIPCHK(hipMalloc(&da, 8 * nw));
        hipLaunchKernelGGL(k_accuracy, dim3(nb), dim3(nt), 0, 0, dw, da, 1 << 24);
        HIPCHK(hipDeviceSynchronize());
        std::vector<double> hw(nw), ha(nw);
        HIPCHK(hipMemcpy(hw.data(), dw, 8 * nw, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(ha.data(), da, 8 * nw, hipMemcpyDeviceToHost));
        double w = 0, at = 0;
        for (int i = 0; i < nw; i++) if (hw[i] > w) { w = hw[i]; at = ha[i]; }
        printf("\nfp32 correction against fp64 over 2^24 operand pairs, |x - y| in [0, 9): worst |difference| %.3e at |x - y| = %.6f\n", w, at);
        printf("(a posterior is exp(f + b - total): what a logAdd adds to one cell and not to its neighbours shows up in it undamped;\n"
               " a band cell has taken part in ~8 logAdds per diagonal of its neighbourhood)\n");
    }
    return 0;
}
