// barrier_probe.hip -- cost of a data-less grid barrier over 512 co-resident workgroups (2 per CU) on MI355X, for the question
// "can k_time<MID> of an ADAPTIVE step wait for the global max |A|^2 inside the kernel instead of ending (END) and starting
// again (BEGIN)?".  Variants: 0 = one counter, every workgroup polls it; 1 = 64 counters packed in 256 B, a wavefront polls all;
// 2 = 64 counters on lines of their own; 3 = as 2 plus an atomicMax on a slot before the arrival (the real sequence).
// Build: hipcc --offload-arch=gfx950 -O3 -o barrier_probe barrier_probe.hip ;  ./barrier_probe [iterations]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Ctl {
    unsigned one[32];
    unsigned packed[64];
    unsigned padded[64][32];
    unsigned long long slots[64];
    unsigned error;
};
__device__ __forceinline__ unsigned ld_sc1(const unsigned* p) {
    unsigned v;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int VAR>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_bar(Ctl* c, int iters, unsigned long long* out) {
    const int tid = threadIdx.x;
    const unsigned per_slot = gridDim.x / 64;
    __shared__ int ok;
    unsigned long long acc = 0;
    for (int it = 1; it <= iters; ++it) {
        __syncthreads();
        if (tid < 64) {
            bool good = false;
            if (VAR == 0) {
                if (tid == 0) __hip_atomic_fetch_add(&c->one[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (int spin = 0; spin < 1000000; ++spin) {
                    const unsigned v = ld_sc1(&c->one[0]);
                    if (v >= (unsigned)it * gridDim.x) { good = true; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            } else {
                unsigned* cnt = VAR == 1 ? &c->packed[0] : &c->padded[0][0];
                const int stride = VAR == 1 ? 1 : 32;
                if (tid == 0) {
                    if (VAR == 3) atomicMax(&c->slots[blockIdx.x % 64], (unsigned long long)(it * 1000 + blockIdx.x));
                    __hip_atomic_fetch_add(&cnt[(blockIdx.x % 64) * stride], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                }
                for (int spin = 0; spin < 1000000; ++spin) {
                    const unsigned v = ld_sc1(&cnt[tid * stride]);
                    if (__all(v >= (unsigned)it * per_slot)) { good = true; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (VAR == 3) {
                    unsigned long long m = c->slots[tid];
                    asm volatile("" ::: "memory");
                    acc += m;
                }
            }
            if (tid == 0) { ok = good; if (!good) atomicExch(&c->error, 1u); }
        }
        __syncthreads();
        if (!ok) return;
    }
    if (tid == 0 && blockIdx.x == 0) out[0] = acc;
}
template <int VAR> int run(Ctl* c, unsigned long long* out, int iters, int grid) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipMemset(c, 0, sizeof(Ctl)));
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_bar<VAR>, dim3(grid), dim3(256), 0, 0, c, iters, out);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
        unsigned err = 0; CHECK(hipMemcpy(&err, &c->error, 4, hipMemcpyDeviceToHost));
        printf("variant %d, %d workgroups: %.2f us per barrier (error flag %u)\n", VAR, grid, ms * 1e3 / iters, err);
    }
    return 0;
}
int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 500;
    Ctl* c; unsigned long long* out;
    CHECK(hipMalloc(&c, sizeof(Ctl))); CHECK(hipMalloc(&out, 8));
    for (int grid : {512, 256, 64}) {
        if (run<0>(c, out, iters, grid)) return 1;
        if (run<1>(c, out, iters, grid)) return 1;
        if (run<2>(c, out, iters, grid)) return 1;
        if (run<3>(c, out, iters, grid)) return 1;
    }
    return 0;
}
