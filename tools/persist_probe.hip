// persist_probe.hip -- how fast is the data path of a PERSISTENT split-step kernel on MI355X?
//
// 2 x 256 workgroups (one per polarisation row and tile, two per CU) loop over phases
//     F: load row i (32 KiB contiguous) -> "compute" -> store row i          | barrier over the 256 workgroups of the row
//     T: load tile i (256 rows x 128 B)  -> "compute" -> store tile i        | barrier
// with write-through (sc1) 16-byte stores, sc1 16-byte loads and a counter barrier per polarisation
// (MI355X_MICROARCH.md, "Hand-offs measured with sc1 loads").  Every phase adds 1 to every element, so a stale
// read shows up in the final values.  Build: hipcc --offload-arch=gfx950 -O3 -o persist_probe persist_probe.hip
//   ./persist_probe [iterations] [compute_cycles] [wgs_per_pol]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int N1 = 256, N2 = 4096, THREADS = 256;

__device__ __forceinline__ f4 ld_sc1(const f4* p) {
    f4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void st_sc1(f4* p, f4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ unsigned ld_flag(const unsigned* p) {
    unsigned v;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

struct Ctl {
    unsigned cnt[2][2][32];     // [pol][which barrier][pad to separate lines]
    unsigned error;
};

// all waves: stores drained; lane 0 of the workgroup: arrive, then wait for `target` arrivals
__device__ __forceinline__ bool barrier_pol(unsigned* cnt, unsigned target, unsigned* err) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __shared__ int ok;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int good = 0;
        for (int spin = 0; spin < 2000000; ++spin) {
            if (ld_flag(cnt) >= target) { good = 1; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        if (!good) atomicExch(err, 1u);
        ok = good;
    }
    __syncthreads();
    return ok != 0;
}

__device__ __forceinline__ void spin_cycles(long long cycles) {
    const long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < cycles) {}
}

template <bool PLAIN>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_probe(f4* Y, Ctl* ctl, int iters, int compute_cycles, int wgs_per_pol, int nobar, int nodata) {
    const int pol = blockIdx.x / wgs_per_pol, i = blockIdx.x % wgs_per_pol;
    f4* Yp = Y + (size_t)pol * N1 * N2 / 2;
    const int tid = threadIdx.x;
    const int units = (N1 * N2 / 2) / wgs_per_pol / THREADS;          // 16-byte units per thread per phase (8 at 256 workgroups)
    f4 v[16];
    unsigned epoch = 0;
    for (int it = 0; it < iters; ++it) {
        // ---- F: row i, contiguous
        if (!nodata) {
            f4* row = Yp + (size_t)i * (N1 * N2 / 2 / wgs_per_pol);
#pragma unroll
            for (int g = 0; g < 16; ++g) if (g < units) v[g] = PLAIN ? row[g * THREADS + tid] : ld_sc1(row + g * THREADS + tid);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int g = 0; g < 16; ++g) if (g < units) v[g] += 1.0f;
            if (compute_cycles) spin_cycles(compute_cycles);
#pragma unroll
            for (int g = 0; g < 16; ++g) if (g < units) { if (PLAIN) row[g * THREADS + tid] = v[g]; else st_sc1(row + g * THREADS + tid, v[g]); }
        }
        ++epoch;
        if (!nobar && !barrier_pol(&ctl->cnt[pol][0][0], epoch * wgs_per_pol, &ctl->error)) return;
        // ---- T: tile i = 8 units (128 B) of every row (k1 = j + 16 (2g + h)), lane = h*32 + (j&3)*8 + c8
        if (!nodata) {
            const int lane = tid & 63, c8 = lane & 7, h = lane >> 5, j = ((tid >> 6) << 2) | ((lane >> 3) & 3);
            const int tiles = N2 / 16;                                // 256 tiles of 8 units per row
            const int per_wg = tiles / wgs_per_pol;                   // 1 at 256 workgroups
            for (int tt = 0; tt < per_wg; ++tt) {
                f4* base = Yp + (size_t)(j + 16 * h) * (N2 / 2) + (size_t)(i * per_wg + tt) * 8 + c8;
#pragma unroll
                for (int g = 0; g < 8; ++g) v[g] = PLAIN ? base[(size_t)g * 32 * (N2 / 2)] : ld_sc1(base + (size_t)g * 32 * (N2 / 2));
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int g = 0; g < 8; ++g) v[g] += 1.0f;
                if (compute_cycles) spin_cycles(compute_cycles);
#pragma unroll
                for (int g = 0; g < 8; ++g) { if (PLAIN) base[(size_t)g * 32 * (N2 / 2)] = v[g]; else st_sc1(base + (size_t)g * 32 * (N2 / 2), v[g]); }
            }
        }
        if (!nobar && !barrier_pol(&ctl->cnt[pol][1][0], epoch * wgs_per_pol, &ctl->error)) return;
    }
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 200;
    const int cyc = argc > 2 ? atoi(argv[2]) : 0;
    const int wgs = argc > 3 ? atoi(argv[3]) : 256;
    const size_t nunits = (size_t)2 * N1 * N2 / 2;
    f4* Y; Ctl* ctl;
    CHECK(hipMalloc(&Y, nunits * sizeof(f4)));
    CHECK(hipMalloc(&ctl, sizeof(Ctl)));
    int nb = 0;
    CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_probe<false>, THREADS, 0));
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    printf("occupancy %d blocks/CU, %d CUs; grid %d\n", nb, prop.multiProcessorCount, 2 * wgs);
    if ((long long)nb * prop.multiProcessorCount < 2 * wgs) { printf("grid does not fit\n"); return 1; }
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int mode = argc > 4 ? atoi(argv[4]) : 0;       // 1: no data phase, 2: plain loads / stores (results may be stale), 4: no barrier (results wrong)
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipMemset(Y, 0, nunits * sizeof(f4)));
        CHECK(hipMemset(ctl, 0, sizeof(Ctl)));
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        if (mode & 2) hipLaunchKernelGGL(k_probe<true>, dim3(2 * wgs), dim3(THREADS), 0, 0, Y, ctl, iters, cyc, wgs, (mode & 4) != 0, (mode & 1) != 0);
        else hipLaunchKernelGGL(k_probe<false>, dim3(2 * wgs), dim3(THREADS), 0, 0, Y, ctl, iters, cyc, wgs, (mode & 4) != 0, (mode & 1) != 0);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        Ctl h; CHECK(hipMemcpy(&h, ctl, sizeof(h), hipMemcpyDeviceToHost));
        std::vector<float> out(nunits * 4);
        CHECK(hipMemcpy(out.data(), Y, nunits * sizeof(f4), hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (float x : out) bad += (x != (float)(2 * iters));
        printf("mode %d (1 no data, 2 plain ld/st, 4 no barrier): %d iterations, compute %d cycles/phase: %.2f us per phase; error flag %u; wrong values %zu of %zu\n",
               mode, iters, cyc, ms * 1e3 / iters / 2, h.error, bad, out.size());
    }
    // the same data movement as separate launches of plain loads / stores (one launch per phase pair is not possible: one per iteration = F and T
    // without any barrier is WRONG data-wise but shows the launch-bound figure is not the comparison; instead time 2 launches per iteration)
    return 0;
}
