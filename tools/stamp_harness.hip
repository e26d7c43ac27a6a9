// Diagnostic: per-phase cycle stamps of the product's row kernel k_freq<float,4096,1,16,FM_PHASE,U16> (one row per launch, like a lane; zero tables).
#define SSFM_STAMPS 1
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
__device__ unsigned long long* g_stamp_buf;
#define SSFM_STAMP_FFT(i) SSFM_STAMP(i)
#include "../opticomlib_amd/csrc/ssfm_kernels.hpp"
using namespace ssfm;
int main(int argc, char** argv) {
    const int rows = argc > 1 ? atoi(argv[1]) : 1;
    const int N1 = 256, N2 = 4096; const long long N = (long long)N1 * N2;
    cf32 *F, *tab, *tw2; unsigned long long* st;
    hipMalloc(&F, sizeof(cf32) * N * rows); hipMalloc(&tab, sizeof(cf32) * N); hipMalloc(&tw2, sizeof(cf32) * 65536);
    hipMalloc(&st, 8 * 16 * 256 * rows);
    hipMemset(F, 0, sizeof(cf32) * N * rows); hipMemset(tab, 0, sizeof(cf32) * N); hipMemset(tw2, 0, sizeof(cf32) * 65536);
    hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf), &st, sizeof(st));
    FreqArgs<float> a; a.F = F; a.tab = tab; a.tw2 = tw2; a.st = nullptr; a.h = 0.1f; a.amp = 1.f / N; a.step = 0; a.inv_n = 1.f / N; a.N1 = N1; a.rows = rows; a.u16 = 1;
    const size_t lds = ((size_t)row_lds_elems(N2, 16) + fft_tw_lds_entries(N2, 16)) * sizeof(cf32);
    for (int it = 0; it < 5; ++it) hipLaunchKernelGGL((k_freq<float, 4096, 1, 16, FM_PHASE, true>), dim3(256 * rows), dim3(256), lds, 0, SSFM_FREQ_KERNEL_ARGS(a));
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(16 * 256 * rows);
    hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
    const char* names[] = {"entry->loads issued", "loads issued->all landed", "landed->fwd FFT done", "table multiply", "inverse FFT", "store issue+drain"};
    unsigned long long tmin = ~0ull, tmax = 0;
    for (int b = 0; b < 256 * rows; ++b) { tmin = std::min(tmin, h[b * 16]); tmax = std::max(tmax, h[b * 16 + 6]); }
    printf("rows=%d: kernel span (first entry -> last end) %llu cycles (s_memtime, 100 MHz? see below)\n", rows, tmax - tmin);
    for (int k = 0; k < 6; ++k) {
        std::vector<unsigned long long> d;
        for (int b = 0; b < 256 * rows; ++b) d.push_back(h[b * 16 + k + 1] - h[b * 16 + k]);
        std::sort(d.begin(), d.end());
        printf("  %-28s median %6llu  p10 %6llu  p90 %6llu\n", names[k], d[d.size() / 2], d[d.size() / 10], d[d.size() * 9 / 10]);
    }
    const char* fn[] = {"s0 entry->pre-bfly", "s0 butterflies", "s0 write+barrier (to s1 entry)", "s1 LDS read+twiddle", "s1 butterflies", "s1 write+barrier", "s2 LDS read+twiddle", "s2 butterflies"};
    for (int k = 0; k < 8; ++k) {
        std::vector<unsigned long long> d;
        for (int b = 0; b < 256 * rows; ++b) d.push_back(h[b * 16 + 8 + k] - h[b * 16 + 7 + k]);
        std::sort(d.begin(), d.end());
        printf("    fwd FFT %-32s median %6llu\n", fn[k], d[d.size() / 2]);
    }
    std::vector<unsigned long long> e, tot;
    for (int b = 0; b < 256 * rows; ++b) { e.push_back(h[b * 16] - tmin); tot.push_back(h[b * 16 + 6] - h[b * 16]); }
    std::sort(e.begin(), e.end()); std::sort(tot.begin(), tot.end());
    printf("  block start skew: median %llu max %llu ; block lifetime median %llu max %llu\n", e[e.size() / 2], e.back(), tot[tot.size() / 2], tot.back());
    return 0;
}
