// HBM write rate of the BoxScene tile kernel's store pattern against plainer ones (stand-alone microbenchmark):
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/store_rate tools/micro/store_rate.hip && /tmp/store_rate
// 160 frames of 1920 x 1080 pixels; every variant writes each byte of the buffer once per launch.
//   tile4    one wave per 64 x 64-pixel tile, a row of 64 dwords (256 bytes) per store instruction -- box_tile_kernel, RGBX8
//   tile8    one wave per 128 x 64-pixel tile, two pixels a lane: 512-byte pieces
//   tile12   one wave per 64 x 64-pixel tile of 12-byte pixels: 768-byte pieces -- box_tile_kernel, three fp32 channels
//   linear4  every wave writes consecutive 256-byte pieces of a contiguous 16 KB: the same size of store, no stride
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));

__global__ __launch_bounds__(64) void tile4(uint32_t *fb, int width, int height, uint32_t v) {
    const int lane = threadIdx.x;
    const int x = blockIdx.x * 64 + lane;
    uint32_t *p = fb + ((size_t)blockIdx.z * height + (size_t)blockIdx.y * 64) * width + x;
    for (int r = 0; r < 64; ++r) {
        if (blockIdx.y * 64 + r < height) p[(size_t)r * width] = v + r;
    }
}
__global__ __launch_bounds__(64) void tile8(uint32_t *fb, int width, int height, uint32_t v) {
    const int lane = threadIdx.x;
    const int x = blockIdx.x * 128 + 2 * lane;
    uint32_t *p = fb + ((size_t)blockIdx.z * height + (size_t)blockIdx.y * 64) * width + x;
    for (int r = 0; r < 64; ++r) {
        if (blockIdx.y * 64 + r < height) *reinterpret_cast<u32x2 *>(p + (size_t)r * width) = u32x2{v + r, v};
    }
}
__global__ __launch_bounds__(64) void tile12(uint32_t *fb, int width, int height, uint32_t v) {
    const int lane = threadIdx.x;
    const int x = blockIdx.x * 64 + lane;
    uint32_t *p = fb + (((size_t)blockIdx.z * height + (size_t)blockIdx.y * 64) * width + x) * 3;
    for (int r = 0; r < 64; ++r) {
        if (blockIdx.y * 64 + r < height) *reinterpret_cast<u32x3 *>(p + (size_t)r * width * 3) = u32x3{v + r, v, v};
    }
}
__global__ __launch_bounds__(64) void linear4(uint32_t *fb, size_t n, uint32_t v) {
    const size_t base = ((size_t)blockIdx.x * 64) * 64 + threadIdx.x;          // 64 pieces of 64 dwords
    for (int r = 0; r < 64; ++r) {
        const size_t i = base + (size_t)r * 64;
        if (i < n) fb[i] = v + r;
    }
}

int main() {
    const int W = 1920, H = 1080, F = 160;
    const size_t px = (size_t)W * H * F;
    uint32_t *fb;
    CHECK(hipMalloc(&fb, px * 12));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int reps = 20;
    for (int variant = 0; variant < 4; ++variant) {
        float best = 1e30f;
        for (int pass = 0; pass < 3; ++pass) {
            CHECK(hipEventRecord(e0));
            for (int k = 0; k < reps; ++k) {
                if (variant == 0) hipLaunchKernelGGL(tile4, dim3(W / 64, (H + 63) / 64, F), dim3(64), 0, 0, fb, W, H, (uint32_t)k);
                if (variant == 1) hipLaunchKernelGGL(tile8, dim3(W / 128, (H + 63) / 64, F), dim3(64), 0, 0, fb, W, H, (uint32_t)k);
                if (variant == 2) hipLaunchKernelGGL(tile12, dim3(W / 64, (H + 63) / 64, F), dim3(64), 0, 0, fb, W, H, (uint32_t)k);
                if (variant == 3) hipLaunchKernelGGL(linear4, dim3((unsigned)((px + 4095) / 4096)), dim3(64), 0, 0, fb, px, (uint32_t)k);
            }
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms / reps < best) best = ms / reps;
        }
        const char *names[] = {"tile4   (256-byte pieces, 64 x 64 tiles)", "tile8   (512-byte pieces, 128 x 64 tiles)", "tile12  (768-byte pieces, 64 x 64 tiles, 12-byte pixels)",
                               "linear4 (256-byte pieces, contiguous)"};
        const double bytes = (double)px * (variant == 2 ? 12 : 4);
        printf("%-58s %8.1f us  %6.2f TB/s\n", names[variant], best * 1e3, bytes / (best * 1e-3) / 1e12);
    }
    return 0;
}
