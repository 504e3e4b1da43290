// valu_rate_probe.hip -- issue cost of the instruction kinds the sweep kernels are made of (gfx950).
// Each test runs WPS waves per SIMD (all CUs), every wave executes REP x 8 independent instructions of one kind (or a mix)
// from inline asm; the shader clock (s_memtime) around the loop gives cycles per wave-instruction PER SIMD:
//   cycles * 1 / (REP * 8 * WPS).  4.0 = one instruction per issue slot of the SIMD (64 lanes on 16).
// build: hipcc -O3 --offload-arch=gfx950 probes/valu_rate_probe.hip -o probes/valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define REP 512

#define BODY8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)

struct Out { unsigned long long cyc; double sink; };

#define KERNEL(name, DECL, ASM8, SINK)                                                        \
    __global__ __launch_bounds__(256) void name(Out *out, double seed, int iseed) {          \
        DECL;                                                                                 \
        __syncthreads();                                                                      \
        const unsigned long long t0 = __builtin_readcyclecounter();                          \
        for (int r = 0; r < REP; r++) { ASM8; }                                               \
        const unsigned long long t1 = __builtin_readcyclecounter();                          \
        if ((threadIdx.x & 63) == 0) {                                                        \
            Out o; o.cyc = t1 - t0; o.sink = SINK;                                            \
            out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = o;                            \
        }                                                                                     \
    }

#define DECL_D double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7, b = seed * 0.5, c = seed * 0.25
#define SINK_D (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7)
#define DECL_I int a0 = iseed, a1 = iseed + 1, a2 = iseed + 2, a3 = iseed + 3, a4 = iseed + 4, a5 = iseed + 5, a6 = iseed + 6, a7 = iseed + 7, b = iseed * 3, c = iseed * 5
#define SINK_I ((double) (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7))

#define A8(INS) asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c))

#define I_FMA(i) "v_fma_f64 %" #i ", %" #i ", %8, %9\n"
#define I_FMAC(i) "v_fmac_f64_e32 %" #i ", %8, %9\n"
#define I_ADD(i) "v_add_f64 %" #i ", %" #i ", %8\n"
#define I_ADDABS(i) "v_add_f64 %" #i ", |%" #i "|, -%8\n"
#define I_MUL(i) "v_mul_f64 %" #i ", %" #i ", %8\n"
#define I_MIN(i) "v_min_f64 %" #i ", %" #i ", %8\n"
#define I_MAX(i) "v_max_f64 %" #i ", %" #i ", %8\n"
#define I_CMP(i) "v_cmp_lt_f64_e32 vcc, %" #i ", %8\n"
#define I_MOV64(i) "v_mov_b64 %" #i ", %8\n"
#define I_ADDU(i) "v_add_u32_e32 %" #i ", %" #i ", %8\n"
#define I_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 3, %8\n"
#define I_AND(i) "v_and_b32_e32 %" #i ", %" #i ", %8\n"
#define I_CNDMASK(i) "v_cndmask_b32_e32 %" #i ", %" #i ", %8, vcc\n"
#define I_DPP(i) "v_mov_b32_dpp %" #i ", %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %9\n"
#define I_SNOP(i) "s_nop 0\n"
#define I_SALU(i) "s_add_u32 s" #i ", s" #i ", 1\n"
#define I_FMA_SALU(i) "v_fma_f64 %" #i ", %" #i ", %8, %9\n s_add_u32 s" #i ", s" #i ", 1\n"
#define I_FMA_SALU2(i) "v_fma_f64 %" #i ", %" #i ", %8, %9\n s_add_u32 s" #i ", s" #i ", 1\n s_and_b32 s1" #i ", s1" #i ", 7\n"
#define I_ADD_SGPR(i) "v_add_f64 %" #i ", %" #i ", s[20:21]\n"
#define I_READLANE(i) "v_readlane_b32 s" #i ", %" #i ", 3\n"
#define I_MIN_ADD(i) "v_min_f64 %" #i ", %" #i ", %8\n v_add_f64 %" #i ", %" #i ", %9\n"
#define I_LDS(i) "ds_read_b64 %" #i ", %8\n"
#define I_LDS128(i) "ds_read_b64 %" #i ", %8\n"
#define I_FMA_LDS(i) "v_fma_f64 %" #i ", %" #i ", %9, %9\n ds_read_b64 %" #i ", %8\n"
#define I_WAIT(i) "s_waitcnt lgkmcnt(0)\n"
#define I_FMA_WAIT(i) "v_fma_f64 %" #i ", %" #i ", %8, %9\n s_waitcnt lgkmcnt(0)\n"

#define I_CND64(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, s[10:11]\n"
#define I_CND_FMA(i) "v_cndmask_b32_e32 %" #i ", %" #i ", %8, vcc\n v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define I_FMA32(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define I_CMP_CND(i) "v_cmp_lt_i32_e32 vcc, %" #i ", %8\n s_nop 1\n v_cndmask_b32_e32 %" #i ", %" #i ", %8, vcc\n"
#define I_CND_SAME(i) "v_cndmask_b32_e32 %" #i ", %8, %9, vcc\n"
#define I_BFI(i) "v_bfi_b32 %" #i ", %8, %" #i ", %9\n"
#define I_MAXSEL(i) "v_max_f64 %" #i ", %" #i ", %8\n"
#define I_CMP64(i) "v_cmp_lt_f64_e64 s[10:11], %" #i ", %8\n"
#define I_CMPI(i) "v_cmp_lt_i32_e32 vcc, %" #i ", %8\n"
#define I_LSHLADD64(i) "v_lshl_add_u64 %" #i ", %" #i ", 3, %8\n"
#define I_PKADD(i) "v_pk_add_f32 %" #i ", %" #i ", %8\n"
#define I_CND2_FMA(i) "v_cndmask_b32_e32 %" #i ", %" #i ", %8, vcc\n v_cndmask_b32_e32 %" #i ", %" #i ", %9, vcc\n v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define I_CND2NOP(i) "v_cndmask_b32_e32 %" #i ", %" #i ", %8, vcc\n s_nop 0\n v_cndmask_b32_e32 %" #i ", %" #i ", %9, vcc\n s_nop 0\n"
#define I_CND64VCC(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, vcc\n"
#define I_CND2_64VCC_FMA(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, vcc\n v_cndmask_b32_e64 %" #i ", %" #i ", %9, vcc\n v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define I_CND2_S_FMA(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, s[10:11]\n v_cndmask_b32_e64 %" #i ", %" #i ", %9, s[10:11]\n v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define I_ADDC(i) "v_addc_co_u32_e32 %" #i ", vcc, %" #i ", %8, vcc\n"
#define I_CNDSDWA(i) "v_cndmask_b32_sdwa %" #i ", %" #i ", %8, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n"
#define CLOB_S : "s0", "s1", "s2", "s3", "s4", "s5", "s6", "s7", "s10", "s11", "s12", "s13", "s14", "s15", "s16", "s17", "vcc", "scc"
#define A8S(INS) asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) CLOB_S)

KERNEL(k_fma, DECL_D, A8(I_FMA), SINK_D)
KERNEL(k_fmac, DECL_D, A8(I_FMAC), SINK_D)
KERNEL(k_add, DECL_D, A8(I_ADD), SINK_D)
KERNEL(k_addabs, DECL_D, A8(I_ADDABS), SINK_D)
KERNEL(k_mul, DECL_D, A8(I_MUL), SINK_D)
KERNEL(k_min, DECL_D, A8(I_MIN), SINK_D)
KERNEL(k_max, DECL_D, A8(I_MAX), SINK_D)
KERNEL(k_cmp, DECL_D, A8S(I_CMP), SINK_D)
KERNEL(k_mov64, DECL_D, A8(I_MOV64), SINK_D)
KERNEL(k_addu, DECL_I, A8(I_ADDU), SINK_I)
KERNEL(k_lshladd, DECL_I, A8(I_LSHLADD), SINK_I)
KERNEL(k_and, DECL_I, A8(I_AND), SINK_I)
KERNEL(k_cndmask, DECL_I, A8S(I_CNDMASK), SINK_I)
KERNEL(k_cnd64, DECL_I, A8S(I_CND64), SINK_I)
KERNEL(k_cnd_same, DECL_I, A8S(I_CND_SAME), SINK_I)
KERNEL(k_bfi, DECL_I, A8(I_BFI), SINK_I)
KERNEL(k_cmpi, DECL_I, A8S(I_CMPI), SINK_I)
KERNEL(k_cnd_fma, DECL_I, A8S(I_CND_FMA), SINK_I)
KERNEL(k_fma32, DECL_I, A8(I_FMA32), SINK_I)
KERNEL(k_cmp_cnd, DECL_I, A8S(I_CMP_CND), SINK_I)
KERNEL(k_cmp64, DECL_D, A8S(I_CMP64), SINK_D)
KERNEL(k_lshladd64, DECL_D, A8(I_LSHLADD64), SINK_D)
KERNEL(k_pkadd, DECL_D, A8(I_PKADD), SINK_D)
KERNEL(k_cnd2_fma, DECL_I, A8S(I_CND2_FMA), SINK_I)
KERNEL(k_cnd2nop, DECL_I, A8S(I_CND2NOP), SINK_I)
KERNEL(k_cnd64vcc, DECL_I, A8S(I_CND64VCC), SINK_I)
KERNEL(k_cnd2_64vcc_fma, DECL_I, A8S(I_CND2_64VCC_FMA), SINK_I)
KERNEL(k_cnd2_s_fma, DECL_I, A8S(I_CND2_S_FMA), SINK_I)
KERNEL(k_cndsdwa, DECL_I, A8S(I_CNDSDWA), SINK_I)
KERNEL(k_dpp, DECL_I, A8(I_DPP), SINK_I)
KERNEL(k_add3, DECL_I, A8(I_ADD3), SINK_I)
KERNEL(k_snop, DECL_I, A8(I_SNOP), SINK_I)
KERNEL(k_salu, DECL_I, A8S(I_SALU), SINK_I)
KERNEL(k_fma_salu, DECL_D, A8S(I_FMA_SALU), SINK_D)
KERNEL(k_fma_salu2, DECL_D, A8S(I_FMA_SALU2), SINK_D)
KERNEL(k_add_sgpr, DECL_D, A8(I_ADD_SGPR), SINK_D)
KERNEL(k_readlane, DECL_I, A8S(I_READLANE), SINK_I)
KERNEL(k_min_add, DECL_D, A8(I_MIN_ADD), SINK_D)
KERNEL(k_fma_wait, DECL_D, A8(I_FMA_WAIT), SINK_D)

// LDS reads: address operand is an int (LDS byte address), destinations doubles
__global__ __launch_bounds__(256) void k_lds(Out *out, double seed, int iseed) {
    __shared__ double tab[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) tab[i] = seed + i;
    __syncthreads();
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
    const unsigned b = (unsigned) (size_t) tab + (threadIdx.x & 63) * 8u + (unsigned) iseed;
    const double c = seed;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < REP; r++) {
        asm volatile(I_LDS(0) I_LDS(1) I_LDS(2) I_LDS(3) I_LDS(4) I_LDS(5) I_LDS(6) I_LDS(7) "s_waitcnt lgkmcnt(0)\n"
                     : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(b), "v"(c));
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) { Out o; o.cyc = t1 - t0; o.sink = SINK_D; out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = o; }
}
// one LDS read (b128) per two FMAs, as in logAdd: does the LDS instruction take a VALU slot?
__global__ __launch_bounds__(256) void k_fma_lds(Out *out, double seed, int iseed) {
    __shared__ double tab[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) tab[i] = seed + i;
    __syncthreads();
    double a0 = seed, a1 = seed, a2 = seed, a3 = seed, a4 = seed, a5 = seed, a6 = seed, a7 = seed;
    double l0 = 0, l1 = 0, l2 = 0, l3 = 0;
    const unsigned b = (unsigned) (size_t) tab + (threadIdx.x & 63) * 8u + (unsigned) iseed;
    const double c = seed;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < REP; r++) {
        asm volatile("v_fma_f64 %0, %0, %13, %13\n ds_read_b64 %8, %12\n v_fma_f64 %1, %1, %13, %13\n"
                     "v_fma_f64 %2, %2, %13, %13\n ds_read_b64 %9, %12 offset:64\n v_fma_f64 %3, %3, %13, %13\n"
                     "v_fma_f64 %4, %4, %13, %13\n ds_read_b64 %10, %12 offset:128\n v_fma_f64 %5, %5, %13, %13\n"
                     "v_fma_f64 %6, %6, %13, %13\n ds_read_b64 %11, %12 offset:192\n v_fma_f64 %7, %7, %13, %13\n"
                     "s_waitcnt lgkmcnt(0)\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=v"(l0), "=v"(l1), "=v"(l2), "=v"(l3)
                     : "v"(b), "v"(c));
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) { Out o; o.cyc = t1 - t0; o.sink = SINK_D + l0 + l1 + l2 + l3; out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = o; }
}

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef void (*kern_t)(Out *, double, int);
struct Test { const char *name; kern_t k; int per_iter; };

int main(int argc, char **argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipDeviceProp_t pr;
    HIPCHK(hipGetDeviceProperties(&pr, 0));
    const int cus = pr.multiProcessorCount;
    printf("device %s, %d CUs, clock %.0f MHz\n", pr.name, cus, pr.clockRate / 1000.0);
    Test tests[] = {
        {"v_fma_f64", k_fma, 8}, {"v_fmac_f64_e32", k_fmac, 8}, {"v_add_f64", k_add, 8}, {"v_add_f64 |a|,-b", k_addabs, 8},
        {"v_add_f64 v,s", k_add_sgpr, 8}, {"v_mul_f64", k_mul, 8}, {"v_min_f64", k_min, 8}, {"v_max_f64", k_max, 8},
        {"v_cmp_lt_f64", k_cmp, 8}, {"v_mov_b64", k_mov64, 8}, {"v_add_u32", k_addu, 8}, {"v_lshl_add_u32", k_lshladd, 8},
        {"v_and_b32", k_and, 8}, {"v_cndmask_b32 a,a,b,vcc", k_cndmask, 8}, {"v_cndmask_b32 a,b,c,vcc", k_cnd_same, 8},
        {"v_cndmask_b32_e64 s[10:11]", k_cnd64, 8}, {"v_bfi_b32", k_bfi, 8}, {"2 cnd_e32 vcc + fma_f32 (triples)", k_cnd2_fma, 24},
        {"cnd_e32, s_nop, cnd_e32, s_nop", k_cnd2nop, 32}, {"v_cndmask_b32_e64 vcc", k_cnd64vcc, 8},
        {"2 cnd_e64 vcc + fma_f32 (triples)", k_cnd2_64vcc_fma, 24}, {"2 cnd_e64 sgpr + fma_f32 (triples)", k_cnd2_s_fma, 24},
        {"v_cndmask_b32_sdwa vcc", k_cndsdwa, 8}, {"v_cmp_lt_i32 vcc", k_cmpi, 8},
        {"v_fma_f32", k_fma32, 8}, {"cndmask + fma_f32 (pairs)", k_cnd_fma, 16}, {"cmp + s_nop 1 + cndmask (triples)", k_cmp_cnd, 24},
        {"v_cmp_lt_f64_e64 sgpr", k_cmp64, 8}, {"v_lshl_add_u64", k_lshladd64, 8}, {"v_pk_add_f32", k_pkadd, 8}, {"v_mov_b32 dpp wave_shr", k_dpp, 8}, {"v_add3_u32", k_add3, 8},
        {"v_readlane_b32", k_readlane, 8}, {"s_nop 0", k_snop, 8}, {"s_add_u32", k_salu, 8},
        {"fma + s_add (pairs)", k_fma_salu, 16}, {"fma + 2 salu (triples)", k_fma_salu2, 24}, {"min + dependent add (pairs)", k_min_add, 16},
        {"fma + s_waitcnt (pairs)", k_fma_wait, 16}, {"ds_read_b64 x8 + wait", k_lds, 9}, {"8 fma + 4 ds_read_b64 + wait", k_fma_lds, 13},
    };
    hipEvent_t ev0, ev1;
    HIPCHK(hipEventCreate(&ev0));
    HIPCHK(hipEventCreate(&ev1));
    for (int wps : {1, 2, 4, 8}) {
        const int waves = cus * 4 * wps, blocks = waves / 4;
        Out *d;
        HIPCHK(hipMalloc(&d, sizeof(Out) * waves));
        std::vector<Out> h(waves);
        printf("\n%d wave(s) per SIMD\n%-32s %12s %12s %12s\n", wps, "instruction", "cyc/instr", "cyc/instr/SIMD", "wall-clock");
        for (const Test &t : tests) {
            hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), 0, 0, d, 1.0, 0);   // warm
            HIPCHK(hipEventRecord(ev0, 0));
            hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), 0, 0, d, 1.0, 0);
            HIPCHK(hipEventRecord(ev1, 0));
            HIPCHK(hipDeviceSynchronize());
            float ms = 0;
            HIPCHK(hipEventElapsedTime(&ms, ev0, ev1));
            HIPCHK(hipMemcpy(h.data(), d, sizeof(Out) * waves, hipMemcpyDeviceToHost));
            double sum = 0;
            for (int i = 0; i < waves; i++) sum += (double) h[i].cyc;
            const double per_wave = sum / waves / ((double) REP * t.per_iter);
            // the same from the wall clock of the launch (includes launch overhead), at the reported shader clock
            const double wall = (double) ms * 1e-3 * pr.clockRate * 1e3 / ((double) REP * t.per_iter * wps);
            printf("%-32s %12.2f %12.2f %12.2f\n", t.name, per_wave, per_wave / wps, wall);
        }
        HIPCHK(hipFree(d));
    }
    return 0;
}
