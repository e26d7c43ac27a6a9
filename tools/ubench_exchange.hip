// ubench_exchange.hip -- what one radix-16 exchange of the workgroup FFT costs by mechanism (dev aid).
// Every lane holds 16 complex64 values (32 dwords); the exchange of a radix-16 stage inside a 16-lane row is a
// 16 x 16 transpose: register t of lane l <-> register l of lane t.
//   lds      16 ds_write_b64 + 16 ds_read_b64 through a padded LDS tile (what wgfft.hpp does; wave-local here, no barrier)
//   swizzle  four butterfly steps (lane xor 1, 2, 4, 8): per step 8 register pairs, each 2 ds_swizzle_b32 + 6 v_cndmask
//            (gfx950 has no lane-xor DPP mode; ds_swizzle uses the LDS crossbar, no LDS storage)
//   bperm    the same butterflies with ds_bpermute_b32 (__shfl_xor)
// Prints cycles per exchange per wave (s_memtime, median over workgroups) with 1, 2 and 4 waves per SIMD busy.
//   hipcc --offload-arch=gfx950 -O3 -o build/ubench_exchange tools/ubench_exchange.hip && ./build/ubench_exchange
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float cf32 __attribute__((ext_vector_type(2)));

template <int K> __device__ __forceinline__ float xor_swz(float x) {
    return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(x), 0x1f | (K << 10)));
}
template <int K, int MODE> __device__ __forceinline__ void butterfly(cf32 (&v)[16], int lane) {
    const bool hi = (lane & K) != 0;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        if (t & K) continue;
        const cf32 a = v[t], b = v[t | K];
        cf32 send, x;
        send.x = hi ? a.x : b.x; send.y = hi ? a.y : b.y;
        if (MODE == 1) { x.x = xor_swz<K>(send.x); x.y = xor_swz<K>(send.y); }
        else { x.x = __shfl_xor(send.x, K, 64); x.y = __shfl_xor(send.y, K, 64); }
        v[t].x = hi ? x.x : a.x; v[t].y = hi ? x.y : a.y;
        v[t | K].x = hi ? b.x : x.x; v[t | K].y = hi ? b.y : x.y;
    }
}
template <int MODE> __global__ void k(float* out, unsigned long long* cyc, int iters) {
    __shared__ cf32 tile[16][64 * 17];          // per wave: 64 lanes x 16 values, one pad element per 16
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, row = lane >> 4, l = lane & 15;
    cf32 v[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) { v[t].x = (float)(threadIdx.x * 16 + t); v[t].y = -v[t].x; }
    cf32* T = tile[wv];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int t = 0; t < 16; ++t) T[row * (16 * 17) + l * 17 + t] = v[t];      // value (lane l, register t)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int t = 0; t < 16; ++t) v[t] = T[row * (16 * 17) + t * 17 + l];      // register t <- (lane t, register l)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        } else {
            butterfly<1, MODE>(v, lane); butterfly<2, MODE>(v, lane); butterfly<4, MODE>(v, lane); butterfly<8, MODE>(v, lane);
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) v[t].x += 1.0f;                                   // (keeps the iterations dependent)
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int t = 0; t < 16; ++t) s += v[t].x * (t + 1) + v[t].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE> void run(const char* name, int threads, float* d, unsigned long long* dc, std::vector<float>& ref) {
    const int iters = 2000, blocks = 256;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, dc, 10);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, dc, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> c(blocks); hipMemcpy(c.data(), dc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
    std::sort(c.begin(), c.end());
    std::vector<float> h((size_t)blocks * threads); hipMemcpy(h.data(), d, sizeof(float) * h.size(), hipMemcpyDeviceToHost);
    bool same = true;
    if (ref.empty()) ref = h; else for (size_t i = 0; i < h.size(); ++i) same = same && h[i] == ref[i];
    printf("%-8s %4d threads/workgroup (%d waves/SIMD): %7.1f cycles per 16x16 complex exchange per wave (s_memtime), %6.3f us per exchange wall; result %s\n",
           name, threads, threads / 256, (double)c[blocks / 2] / iters, ms * 1e3 / iters, same ? "identical to lds" : "DIFFERS");
}
int main() {
    float* d; unsigned long long* dc; hipMalloc(&d, sizeof(float) * 256 * 1024); hipMalloc(&dc, sizeof(unsigned long long) * 256);
    for (int threads : {256, 512, 1024}) {
        std::vector<float> ref;
        run<0>("lds", threads, d, dc, ref);
        run<1>("swizzle", threads, d, dc, ref);
        run<2>("bperm", threads, d, dc, ref);
    }
    return 0;
}
