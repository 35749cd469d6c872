// valu_rate.hip -- how many cycles does one SIMD of gfx950 spend per wave64 VALU instruction?  (diagnostic for the
// `roofline.valu` figure of bench.py)   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rate.hip -o /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP 64
template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed) {
    float a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = seed + (float)threadIdx.x * 1e-3f + (float)i;
    const float b = seed * 0.5f, c = seed * 0.25f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
            if (KIND == 0) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            } else if (KIND == 1) {
#pragma unroll
                for (int i = 0; i < 8; i += 2) {
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(*reinterpret_cast<double *>(&a[i])) : "v"(*reinterpret_cast<const double *>(&a[(i + 2) & 7])), "v"(*reinterpret_cast<const double *>(&a[(i + 4) & 7])));
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(*reinterpret_cast<double *>(&a[i])) : "v"(*reinterpret_cast<const double *>(&a[(i + 2) & 7])), "v"(*reinterpret_cast<const double *>(&a[(i + 4) & 7])));
                }
            } else if (KIND == 2) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i]));
            } else if (KIND == 3) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            } else if (KIND == 4) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));
            } else if (KIND == 5) {
                const unsigned long long m = __builtin_amdgcn_ballot_w64(seed > 0.5f);
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "s"(m));
            } else if (KIND == 6) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
            } else if (KIND == 7) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
            } else if (KIND == 8) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_fract_f32 %0, %0" : "+v"(a[i]));
            } else if (KIND == 9) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_cvt_u32_f32 %0, %0" : "+v"(a[i]));
            } else if (KIND == 10) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            } else if (KIND == 11) {
                int sg;
#pragma unroll
                for (int i = 0; i < 8; ++i) { asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(sg) : "v"(a[i])); asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i]) : "s"(sg)); }
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND>
void run(const char *name, int blocks_per_cu) {
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    int clk = 0;
    hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    const int blocks = cus * blocks_per_cu, iters = 4000;
    float *out;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 100, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions per SIMD: each block = 4 waves = one per SIMD; blocks_per_cu waves per SIMD
    const double inst_per_simd = (double)blocks_per_cu * iters * REP;
    printf("%-14s waves/SIMD %d  %.3f ms  -> %.2f ns per wave-instruction per SIMD = %.2f cycles at %.2f GHz (reported clock)\n", name, blocks_per_cu, ms,
           ms * 1e6 / inst_per_simd, ms * 1e6 / inst_per_simd * clk * 1e-6, clk * 1e-6);
    hipFree(out);
}

int main() {
    for (int w : {2, 6}) {
        run<0>("v_fma_f32", w);
        run<1>("v_pk_fma_f32", w);
        run<2>("v_rsq_f32", w);
        run<3>("v_mul_f32", w);
        run<4>("cndmask vcc", w);
        run<5>("cndmask sgpr", w);
        run<6>("cmp+cndmask", w);
        run<7>("v_sqrt_f32", w);
        run<8>("v_fract_f32", w);
        run<9>("v_cvt_u32_f32", w);
        run<10>("v_perm_b32", w);
        run<11>("readlane+add", w);
        run<12>("v_div_fixup", w);
    }
    return 0;
}
