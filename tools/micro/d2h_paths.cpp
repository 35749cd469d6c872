// d2h_paths.cpp -- what it costs to bring one finished frame from HBM into the CALLER's (pageable) buffer, the last step of
// the drop-in call nt_render (reference: the `dest` of obj_BlockingRenderer_render, src/render.cpp:853-909, is a Python buffer).
//   hipcc -O2 tools/micro/d2h_paths.cpp -o tools/micro/build/d2h_paths -pthread && tools/micro/build/d2h_paths [bytes]
// Stand-alone microbenchmark, not part of the product.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); return 1; } } while (0)

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void copy_threads(char *dst, const char *src, size_t n, int threads) {
    if (threads <= 1) { memcpy(dst, src, n); return; }
    std::vector<std::thread> th;
    const size_t chunk = (n / threads + 4095) & ~(size_t)4095;
    for (int t = 0; t < threads; ++t) {
        const size_t b = (size_t)t * chunk;
        if (b >= n) break;
        const size_t len = n - b < chunk ? n - b : chunk;
        th.emplace_back([=] { memcpy(dst + b, src + b, len); });
    }
    for (auto &t : th) t.join();
}

int main(int argc, char **argv) {
    const size_t n = argc > 1 ? (size_t)atoll(argv[1]) : (size_t)1920 * 1080 * 4;
    const int reps = 20;
    void *dev = nullptr, *pinned = nullptr;
    CK(hipMalloc(&dev, n));
    CK(hipMemset(dev, 0x5a, n));
    CK(hipHostMalloc(&pinned, n, hipHostMallocDefault));
    char *page = (char *)malloc(n + 4096);
    memset(page, 1, n);
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    auto timeit = [&](const char *what, auto fn) {
        fn();
        double best = 1e30, sum = 0;
        for (int i = 0; i < reps; ++i) {
            const double t0 = now_us();
            fn();
            const double t = now_us() - t0;
            best = t < best ? t : best;
            sum += t;
        }
        printf("%-64s best %8.1f us  mean %8.1f us  (%.1f GB/s best)\n", what, best, sum / reps, n / best / 1e3);
    };
    printf("%zu bytes\n", n);
    timeit("D2H async -> pinned + sync", [&] { (void)hipMemcpyAsync(pinned, dev, n, hipMemcpyDeviceToHost, st); (void)hipStreamSynchronize(st); });
    timeit("D2H async -> pageable + sync (the runtime's own staging)", [&] { (void)hipMemcpyAsync(page, dev, n, hipMemcpyDeviceToHost, st); (void)hipStreamSynchronize(st); });
    timeit("memcpy pinned -> pageable, 1 thread", [&] { copy_threads(page, (const char *)pinned, n, 1); });
    timeit("memcpy pinned -> pageable, 4 threads (spawned per call)", [&] { copy_threads(page, (const char *)pinned, n, 4); });
    timeit("memcpy pinned -> pageable, 8 threads (spawned per call)", [&] { copy_threads(page, (const char *)pinned, n, 8); });
    timeit("hipHostRegister + hipHostUnregister (pageable, touched)", [&] { (void)hipHostRegister(page, n, hipHostRegisterDefault); (void)hipHostUnregister(page); });
    timeit("register + D2H async -> registered + sync + unregister", [&] {
        (void)hipHostRegister(page, n, hipHostRegisterDefault);
        (void)hipMemcpyAsync(page, dev, n, hipMemcpyDeviceToHost, st);
        (void)hipStreamSynchronize(st);
        (void)hipHostUnregister(page);
    });
    CK(hipHostRegister(page, n, hipHostRegisterDefault));
    timeit("D2H async -> registered (kept registered) + sync", [&] { (void)hipMemcpyAsync(page, dev, n, hipMemcpyDeviceToHost, st); (void)hipStreamSynchronize(st); });
    hipPointerAttribute_t at;
    const double t0 = now_us();
    const hipError_t pe = hipPointerGetAttributes(&at, page);
    printf("hipPointerGetAttributes(registered): %s type %d, %.1f us\n", hipGetErrorString(pe), pe == hipSuccess ? (int)at.type : -1, now_us() - t0);
    CK(hipHostUnregister(page));
    const double t1 = now_us();
    const hipError_t pe2 = hipPointerGetAttributes(&at, page);
    printf("hipPointerGetAttributes(plain malloc): %s type %d, %.1f us\n", hipGetErrorString(pe2), pe2 == hipSuccess ? (int)at.type : -1, now_us() - t1);
    // chunked: D2H of chunk k+1 into pinned while chunk k is copied out by the host (double buffer, 1 MiB chunks)
    timeit("pipelined: 1 MiB chunks D2H -> pinned, host memcpy behind it", [&] {
        const size_t chunk = 1 << 20;
        hipEvent_t ev[2];
        (void)hipEventCreateWithFlags(&ev[0], hipEventDisableTiming);
        (void)hipEventCreateWithFlags(&ev[1], hipEventDisableTiming);
        const size_t nch = (n + chunk - 1) / chunk;
        for (size_t k = 0; k < nch + 1; ++k) {
            if (k < nch) {
                const size_t b = k * chunk, len = n - b < chunk ? n - b : chunk;
                (void)hipMemcpyAsync((char *)pinned + b, (char *)dev + b, len, hipMemcpyDeviceToHost, st);
                (void)hipEventRecord(ev[k & 1], st);
            }
            if (k > 0) {
                const size_t b = (k - 1) * chunk, len = n - b < chunk ? n - b : chunk;
                (void)hipEventSynchronize(ev[(k - 1) & 1]);
                memcpy(page + b, (char *)pinned + b, len);
            }
        }
        (void)hipEventDestroy(ev[0]);
        (void)hipEventDestroy(ev[1]);
    });
    return 0;
}
