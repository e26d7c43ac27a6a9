#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
template <typename V, int K>
__global__ void k(const V* __restrict__ src, float* out, unsigned long long* st, int stride) {
    const V* p = src + (size_t)blockIdx.x * stride * K + threadIdx.x;
    V v[K];
    unsigned long long t0, t1, t2;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll
    for (int i = 0; i < K; ++i) v[i] = p[(size_t)i * stride];
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2)::"memory");
    float s = 0;
#pragma unroll
    for (int i = 0; i < K; ++i) { const float* f = (const float*)&v[i]; for (int q = 0; q < (int)(sizeof(V) / 4); ++q) s += f[q]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { st[blockIdx.x * 2] = t1 - t0; st[blockIdx.x * 2 + 1] = t2 - t0; }
}
template <typename V, int K> void run(const char* name, int threads) {
    const int blocks = 256, stride = threads;   // each i: one coalesced row of `threads` vectors
    V* src; float* out; unsigned long long* st;
    hipMalloc(&src, sizeof(V) * (size_t)blocks * stride * K); hipMalloc(&out, 4 * blocks * threads); hipMalloc(&st, 16 * blocks);
    hipMemset(src, 0, sizeof(V) * (size_t)blocks * stride * K);
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL((k<V, K>), dim3(blocks), dim3(threads), 0, 0, src, out, st, stride);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 2); hipMemcpy(h.data(), st, 16 * blocks, hipMemcpyDeviceToHost);
    std::vector<unsigned long long> a, b; for (int i = 0; i < blocks; ++i) { a.push_back(h[2 * i]); b.push_back(h[2 * i + 1]); }
    std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
    const double kb = sizeof(V) * (double)threads * K / 1024.0;
    printf("%-10s K=%2d threads=%4d (%5.1f KiB/WG): issue %5llu cyc (%.1f cyc/instr), landed %5llu cyc -> %.1f B/clk/CU\n", name, K, threads, kb, a[blocks / 2], (double)a[blocks / 2] / K, b[blocks / 2], kb * 1024 / b[blocks / 2]);
    hipFree(src); hipFree(out); hipFree(st);
}
int main() {
    run<float, 32>("dword", 256); run<float2, 32>("dwordx2", 256); run<float4, 16>("dwordx4", 256); run<float4, 32>("dwordx4", 256);
    run<float2, 48>("dwordx2", 256); run<float2, 24>("dwordx2", 512); run<float4, 24>("dwordx4", 256);
    return 0;
}
