// valu_rate3.hip -- is the "half rate" of v_max / v_cmp / v_cndmask / v_perm ... (valu_rate2.hip: twice the WALL time of v_add /
// v_fma streams) an issue cost, or something else?  Every wave stamps its loop with the shader clock (s_memtime) and the 100 MHz
// real-time counter and records where it ran (HW_ID: XCC, SE, CU, SIMD); the host prints, per instruction: the kernel's span
// (first start to last end, real time), the mean and the spread of a wave's own loop time, waves per SIMD as placed, and the
// cycles per wave-instruction per SIMD computed three ways.   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rate3.hip -o tools/micro/build/valu_rate3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <map>
#include <vector>

#define REP 64
struct Rec { unsigned long long t0, t1, r0, r1; unsigned hw, xcc; };
#define KERNEL(NAME, ASM)                                                                                  \
    __global__ __launch_bounds__(64) void NAME(Rec *rec, int iters, float seed) {                          \
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
        if (s == 12345.678f) rec[0].hw = 1;                                                                \
        if (threadIdx.x == 0) {                                                                            \
            Rec q; q.t0 = t0; q.t1 = t1; q.r0 = r0; q.r1 = r1;                                             \
            q.hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));        /* HW_REG_HW_ID */          \
            q.xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));      /* HW_REG_XCC_ID */         \
            rec[blockIdx.x] = q;                                                                           \
        }                                                                                                  \
    }

KERNEL(k_fma, "v_fma_f32 %0, %0, %1, %2")
KERNEL(k_add, "v_add_f32 %0, %0, %1")
KERNEL(k_max, "v_max_f32 %0, %0, %1")
KERNEL(k_cmp, "v_cmp_gt_f32 vcc, %0, %1")
KERNEL(k_and, "v_and_b32 %0, %0, %1")
KERNEL(k_lshl, "v_lshlrev_b32 %0, 3, %0")
KERNEL(k_fract, "v_fract_f32 %0, %0")
KERNEL(k_perm, "v_perm_b32 %0, %0, %1, %2")
KERNEL(k_cvt, "v_cvt_u32_f32 %0, %0")
KERNEL(k_rsq, "v_rsq_f32 %0, %0")
KERNEL(k_mov, "v_mov_b32 %0, %1")
KERNEL(k_cndmask_s, "v_cndmask_b32 %0, %0, %1, s[20:21]")

typedef void (*kern_t)(Rec *, int, float);
static void run(const char *name, kern_t fn, int waves_per_simd) {
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int blocks = cus * 4 * waves_per_simd, iters = 3000;        // one wave a block
    Rec *rec;
    hipMalloc(&rec, (size_t)blocks * sizeof(Rec));
    hipLaunchKernelGGL(fn, dim3(blocks), dim3(64), 0, 0, rec, 100, 1.0f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(fn, dim3(blocks), dim3(64), 0, 0, rec, iters, 1.0f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<Rec> h(blocks);
    hipMemcpy(h.data(), rec, (size_t)blocks * sizeof(Rec), hipMemcpyDeviceToHost);
    unsigned long long rmin = ~0ull, rmax = 0;
    double dur_real = 0, dur_cyc = 0, dmin = 1e30, dmax = 0;
    std::map<unsigned long long, int> per_simd;
    for (const Rec &q : h) {
        rmin = std::min(rmin, q.r0);
        rmax = std::max(rmax, q.r1);
        const double d = (double)(q.r1 - q.r0) * 0.01;          // us
        dur_real += d;
        dmin = std::min(dmin, d);
        dmax = std::max(dmax, d);
        dur_cyc += (double)(q.t1 - q.t0);
        // HW_ID: wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ... ; key = (xcc, se, sh, cu, simd)
        const unsigned long long key = ((unsigned long long)(q.xcc & 15) << 32) | (q.hw & 0xff30u);
        per_simd[key]++;
    }
    int wmin = 1 << 30, wmax = 0;
    for (auto &kv : per_simd) { wmin = std::min(wmin, kv.second); wmax = std::max(wmax, kv.second); }
    const double span_us = (double)(rmax - rmin) * 0.01;
    const double per_wave_inst = (double)iters * REP;
    printf("%-12s %d waves/SIMD asked: event %.1f us, span %.1f us; a wave's loop: mean %.1f us (min %.1f, max %.1f), %.0f cycles -> clock %.2f GHz; "
           "SIMDs used %zu, waves on a SIMD %d..%d\n"
           "             cycles per wave-instruction per SIMD: from the span %.2f, from a wave's own loop (x waves asked) %.2f\n",
           name, waves_per_simd, ms * 1e3, span_us, dur_real / blocks, dmin, dmax, dur_cyc / blocks, dur_cyc / dur_real * 1e-3 / 1.0,
           per_simd.size(), wmin, wmax,
           span_us * (dur_cyc / dur_real) / (per_wave_inst * waves_per_simd), dur_cyc / blocks / (per_wave_inst * waves_per_simd));
    hipFree(rec);
}
#define RUN(NAME, W) run(#NAME, NAME, W)
int main() {
    for (int w : {1, 4}) {
        RUN(k_fma, w); RUN(k_add, w); RUN(k_max, w); RUN(k_cmp, w); RUN(k_and, w); RUN(k_lshl, w); RUN(k_fract, w); RUN(k_perm, w); RUN(k_cvt, w);
        RUN(k_mov, w); RUN(k_cndmask_s, w); RUN(k_rsq, w);
    }
    return 0;
}
