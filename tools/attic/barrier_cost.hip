// What does a stream that sits BLOCKED on an event cost the other streams' launches?  (round 5: the capture run's unexplained 1.2-1.4 ms.)
// Two "lanes" (high-priority streams) run chains of short memory-bound kernels, as the propagator's lanes do; a third stream meanwhile
//   (a) does nothing,  (b) holds a hipStreamWaitEvent on an event that completes only after the chains (recorded behind a long spin kernel on a fourth
//   stream),  (c) the same with a normal-priority third stream,  (d) holds a spinning one-wavefront kernel instead of a barrier packet.
// hipcc --offload-arch=gfx950 -O2 -o barrier_cost tools/attic/barrier_cost.hip && ./barrier_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_touch(float4* p, long long n) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i < n) { float4 v = p[i]; v.x += 1.f; p[i] = v; }
}
__global__ void k_spin(long long ticks) {                          // 100 MHz clock
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
__global__ void k_spin_flag(const unsigned* flag, long long ticks) {
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u && wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
__global__ void k_set(unsigned* flag, unsigned v) { __hip_atomic_store(flag, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

int main(int argc, char** argv) {
    const long long spin_ticks = argc > 1 ? atoll(argv[1]) * 100000ll : 4000000ll;       // argv[1]: ms the far event is away (default 40)
    int lo = 0, hi = 0;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    hipStream_t lane[2], third_hi, third_lo, fourth;
    for (auto& s : lane) CK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi));
    CK(hipStreamCreateWithPriority(&third_hi, hipStreamNonBlocking, hi));
    CK(hipStreamCreateWithFlags(&third_lo, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&fourth, hipStreamNonBlocking));
    const long long n = 1 << 20;                                   // 16 MiB per lane: a ~7 us kernel
    float4* buf[2];
    for (auto& b : buf) { CK(hipMalloc(&b, sizeof(float4) * n)); CK(hipMemset(b, 0, sizeof(float4) * n)); }
    unsigned* flag; CK(hipMalloc(&flag, 64)); CK(hipMemset(flag, 0, 64));
    hipEvent_t e0, e1, far, join;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreateWithFlags(&far, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    const int launches = 2000;
    auto chains = [&](float* ms) -> int {
        CK(hipEventRecord(e0, lane[0]));
        CK(hipStreamWaitEvent(lane[1], e0, 0));
        for (int i = 0; i < launches; ++i)
            for (int g = 0; g < 2; ++g) hipLaunchKernelGGL(k_touch, dim3((unsigned)(n / 256)), dim3(256), 0, lane[g], buf[g], n);
        CK(hipEventRecord(join, lane[1]));
        CK(hipStreamWaitEvent(lane[0], join, 0));
        CK(hipEventRecord(e1, lane[0]));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(ms, e0, e1));
        return 0;
    };
    const char* names[5] = {"third stream idle", "third stream (high priority) blocked on a far event", "third stream (normal priority) blocked on a far event",
                            "third stream holds a spinning wavefront", "two more streams each blocked on a far event"};
    std::vector<float> res[5];
    float ms = 0.f;
    if (chains(&ms)) return 1;                                     // warm-up
    for (int round = 0; round < 5; ++round)
        for (int mode = 0; mode < 5; ++mode) {
            CK(hipDeviceSynchronize());
            CK(hipMemset(flag, 0, 4));
            if (mode == 1 || mode == 2 || mode == 4) {
                hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, fourth, spin_ticks);
                CK(hipEventRecord(far, fourth));
                CK(hipStreamWaitEvent(mode == 2 ? third_lo : third_hi, far, 0));
                hipLaunchKernelGGL(k_set, dim3(1), dim3(64), 0, mode == 2 ? third_lo : third_hi, flag + 1, 1u);    // (something behind the barrier)
                if (mode == 4) { CK(hipStreamWaitEvent(third_lo, far, 0)); hipLaunchKernelGGL(k_set, dim3(1), dim3(64), 0, third_lo, flag + 2, 1u); }
            } else if (mode == 3) {
                hipLaunchKernelGGL(k_spin_flag, dim3(1), dim3(64), 0, third_hi, (const unsigned*)flag, 4000000ll);
            }
            if (chains(&ms)) return 1;
            if (mode == 3) hipLaunchKernelGGL(k_set, dim3(1), dim3(64), 0, fourth, flag, 1u);
            res[mode].push_back(ms);
        }
    CK(hipDeviceSynchronize());
    std::printf("# the far event is %lld ms away.  Two lanes x %d launches of a 16 MiB read-modify-write kernel; ms for the pair of chains (min / median of 5 interleaved rounds), us per launch\n", spin_ticks / 100000, launches);
    for (int mode = 0; mode < 5; ++mode) {
        std::sort(res[mode].begin(), res[mode].end());
        std::printf("%-58s %7.2f / %7.2f ms   %5.2f us per launch\n", names[mode], res[mode][0], res[mode][2], res[mode][0] * 1e3 / launches);
    }
    return 0;
}
