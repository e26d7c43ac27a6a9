// Host<->device copy strategies for a pageable 32 MiB buffer (dev aid):
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/ubench_h2d tools/ubench_h2d.cpp -lpthread && /tmp/ubench_h2d
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void par_memcpy(char* dst, const char* src, size_t n, int threads) {
    if (threads <= 1) { memcpy(dst, src, n); return; }
    std::vector<std::thread> th;
    size_t per = (n / threads + 4095) & ~size_t(4095);
    for (int t = 0; t < threads; ++t) {
        size_t o = per * t; if (o >= n) break;
        size_t len = o + per <= n ? per : n - o;
        th.emplace_back([=] { memcpy(dst + o, src + o, len); });
    }
    for (auto& t : th) t.join();
}

int main() {
    const size_t N = 32u << 20;
    char* host = (char*)malloc(N); memset(host, 1, N);
    char* back = (char*)malloc(N); memset(back, 0, N);
    void* dev; CK(hipMalloc(&dev, N));
    hipStream_t s; CK(hipStreamCreate(&s));
    const size_t CH = 4u << 20;
    char* pin[2]; hipEvent_t ev[2];
    for (int i = 0; i < 2; ++i) { CK(hipHostMalloc((void**)&pin[i], CH)); CK(hipEventCreate(&ev[i])); }
    for (int rep = 0; rep < 3; ++rep) {
        double t = now(); CK(hipMemcpy(dev, host, N, hipMemcpyHostToDevice)); double a = now() - t;
        t = now(); CK(hipMemcpy(back, dev, N, hipMemcpyDeviceToHost)); double b = now() - t;
        printf("plain pageable: H2D %.2f ms (%.1f GB/s)  D2H %.2f ms (%.1f GB/s)\n", a * 1e3, N / a / 1e9, b * 1e3, N / b / 1e9);
    }
    for (int rep = 0; rep < 3; ++rep) {
        double t = now(); CK(hipHostRegister(host, N, hipHostRegisterDefault)); double r = now() - t;
        t = now(); CK(hipMemcpy(dev, host, N, hipMemcpyHostToDevice)); double a = now() - t;
        t = now(); CK(hipHostUnregister(host)); double u = now() - t;
        printf("register %.2f ms + H2D %.2f ms + unregister %.2f ms = %.2f ms\n", r * 1e3, a * 1e3, u * 1e3, (r + a + u) * 1e3);
    }
    for (int threads : {1, 2, 4, 8}) {
        for (int rep = 0; rep < 2; ++rep) {
            double t = now();
            int k = 0;
            for (size_t o = 0; o < N; o += CH, ++k) {
                int i = k & 1;
                if (k >= 2) CK(hipEventSynchronize(ev[i]));
                par_memcpy(pin[i], host + o, CH, threads);
                CK(hipMemcpyAsync((char*)dev + o, pin[i], CH, hipMemcpyHostToDevice, s));
                CK(hipEventRecord(ev[i], s));
            }
            CK(hipStreamSynchronize(s));
            double a = now() - t;
            t = now();
            k = 0;
            // D2H: DMA chunk k+1 while the CPU copies chunk k out of the pinned buffer
            CK(hipMemcpyAsync(pin[0], dev, CH, hipMemcpyDeviceToHost, s)); CK(hipEventRecord(ev[0], s));
            for (size_t o = 0; o < N; o += CH, ++k) {
                int i = k & 1;
                if (o + CH < N) { CK(hipMemcpyAsync(pin[1 - i], (char*)dev + o + CH, CH, hipMemcpyDeviceToHost, s)); CK(hipEventRecord(ev[1 - i], s)); }
                CK(hipEventSynchronize(ev[i]));
                par_memcpy(back + o, pin[i], CH, threads);
            }
            double b = now() - t;
            printf("staged 4 MiB chunks, %d copy threads: H2D %.2f ms (%.1f GB/s)  D2H %.2f ms (%.1f GB/s)\n", threads, a * 1e3, N / a / 1e9, b * 1e3, N / b / 1e9);
        }
    }
    printf("check %d\n", memcmp(host, back, N));
    return 0;
}
