// valu_rate2.hip -- cycles per wave64 instruction per SIMD on gfx950 for the instructions the BoxScene row loops are made
// of (companion of valu_rate.hip; diagnostic only).   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rate2.hip -o tools/micro/build/valu_rate2
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP 64
__device__ unsigned long long g_ticks[128];
#define KERNEL(NAME, ASM)                                                                                  \
    __global__ __launch_bounds__(256) void NAME(float *out, int iters, float seed) {                       \
        float a[8];                                                                                        \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) a[i] = seed + (float)threadIdx.x * 1e-3f + (float)i; \
        float b = seed * 0.5f, c = seed * 0.25f;                                                           \
        asm volatile("" : "+v"(b), "+v"(c));                                                               \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();  \
        for (int it = 0; it < iters; ++it) {                                                               \
            _Pragma("unroll") for (int r = 0; r < REP / 8; ++r) {                                          \
                _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c) : "vcc"); \
            }                                                                                              \
        }                                                                                                  \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();  \
        float s = 0;                                                                                       \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) s += a[i];                                           \
        out[blockIdx.x * 256 + threadIdx.x] = s;                                                           \
        if (threadIdx.x == 0 && blockIdx.x < 64) { g_ticks[2 * blockIdx.x] = t1 - t0; g_ticks[2 * blockIdx.x + 1] = r1 - r0; } \
    }

// g_ticks: shader-clock ticks (s_memtime) and 100 MHz ticks (s_memrealtime) a wave spent in its loop: the true cycle count,
// whatever clock the chip held (the figures "at 2.40 GHz" convert wall time with the nominal clock and read ~10 % high)

KERNEL(k_fma, "v_fma_f32 %0, %0, %1, %2")
KERNEL(k_add, "v_add_f32 %0, %0, %1")
KERNEL(k_sub_abs, "v_sub_f32 %0, |%0|, %1")
KERNEL(k_max, "v_max_f32 %0, %0, %1")
KERNEL(k_med3, "v_med3_f32 %0, %0, %1, %2")
KERNEL(k_cmp, "v_cmp_gt_f32 vcc, %0, %1")
KERNEL(k_cmp_s, "v_cmp_gt_f32 s[20:21], %0, %1")
KERNEL(k_and, "v_and_b32 %0, %0, %1")
KERNEL(k_and_or, "v_and_or_b32 %0, %0, %1, %2")
KERNEL(k_ashr, "v_ashrrev_i32 %0, 31, %0")
KERNEL(k_lshl, "v_lshlrev_b32 %0, 3, %0")
KERNEL(k_addu, "v_add_u32 %0, %0, %1")
KERNEL(k_bfe, "v_bfe_u32 %0, %0, 3, 8")
KERNEL(k_rcp, "v_rcp_f32 %0, %0")
KERNEL(k_mov, "v_mov_b32 %0, %1")
KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL(k_fract, "v_fract_f32 %0, %0")
KERNEL(k_perm, "v_perm_b32 %0, %0, %1, %2")
KERNEL(k_cvt, "v_cvt_u32_f32 %0, %0")
KERNEL(k_rndne, "v_rndne_f32 %0, %0")
KERNEL(k_mul_legacy, "v_mul_legacy_f32 %0, %0, %1")
KERNEL(k_fmac, "v_fmac_f32 %0, %1, %2")
KERNEL(k_mad_u24, "v_mad_u32_u24 %0, %0, %1, %2")
KERNEL(k_bfi, "v_bfi_b32 %0, %0, %1, %2")
KERNEL(k_xor, "v_xor_b32 %0, %0, %1")
KERNEL(k_div_scale, "v_div_scale_f32 %0, vcc, %0, %1, %2")
KERNEL(k_div_fmas, "v_div_fmas_f32 %0, %0, %1, %2")
KERNEL(k_div_fixup, "v_div_fixup_f32 %0, %0, %1, %2")
KERNEL(k_sqrt, "v_sqrt_f32 %0, %0")
KERNEL(k_rsq, "v_rsq_f32 %0, %0")
KERNEL(k_cmp_class, "v_cmp_class_f32 vcc, %0, %1")
KERNEL(k_max3, "v_max3_f32 %0, %0, %1, %2")
KERNEL(k_readfirstlane, "v_readfirstlane_b32 s20, %0")

typedef void (*kern_t)(float *, int, float);
static void run(const char *name, kern_t fn, int blocks_per_cu) {
    int cus = 0, clk = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    const int blocks = cus * blocks_per_cu, iters = 3000;
    float *out;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), 0, 0, out, 100, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double inst_per_simd = (double)blocks_per_cu * iters * REP;
    unsigned long long ticks[128];
    hipMemcpyFromSymbol(ticks, HIP_SYMBOL(g_ticks), sizeof(ticks));
    double cyc = 0, real = 0;
    for (int i = 0; i < 64; ++i) { cyc += (double)ticks[2 * i]; real += (double)ticks[2 * i + 1]; }
    printf("%-16s waves/SIMD %d  %.3f ms -> %.2f cycles per wave-instruction per SIMD at the nominal %.2f GHz; in-kernel: %.2f cycles, clock %.2f GHz\n", name,
           blocks_per_cu, ms, ms * 1e6 / inst_per_simd * clk * 1e-6, clk * 1e-6, cyc / 64.0 / inst_per_simd, cyc / real * 0.1);
    hipFree(out);
}
#define RUN(NAME) run(#NAME, NAME, w)
int main() {
    for (int w : {4}) {
        RUN(k_fma); RUN(k_add); RUN(k_sub_abs); RUN(k_max); RUN(k_med3); RUN(k_max3); RUN(k_cmp); RUN(k_cmp_s); RUN(k_cmp_class); RUN(k_and);
        RUN(k_and_or); RUN(k_ashr); RUN(k_lshl); RUN(k_addu); RUN(k_bfe); RUN(k_bfi); RUN(k_xor); RUN(k_mov); RUN(k_cndmask); RUN(k_fract);
        RUN(k_perm); RUN(k_cvt); RUN(k_rndne); RUN(k_mul_legacy); RUN(k_fmac); RUN(k_mad_u24); RUN(k_rcp); RUN(k_rsq); RUN(k_sqrt);
        RUN(k_div_scale); RUN(k_div_fmas); RUN(k_div_fixup); RUN(k_readfirstlane);
    }
    return 0;
}
