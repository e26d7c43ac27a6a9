// xcd_barrier_probe.hip -- what does a grid barrier cost when all its workgroups sit on ONE XCD (one L2) and talk through that L2,
// against workgroups spread over the eight XCDs that must meet in memory?  (Question behind a one-launch engine for 2^14 ... 2^16-sample
// plans, whose whole working set fits one XCD's 4 MiB L2.)
//   local : grid = 8 G workgroups, those not on XCC 0 leave at once; arrival = L2 atomic (no sc1), poll = the same atomic adding 0; data: plain stores
//           (the L1 writes through to the L2) and, behind an L1 invalidate (buffer_inv sc0), plain loads
//   spread: grid = G workgroups over all XCDs; arrival = agent-scope relaxed atomic, poll = sc1 load
// Each barrier is followed by a data check: every workgroup writes a 1 KiB block (plain stores / sc1 stores), after the barrier reads the
// block of the next workgroup (sc0 / sc1 loads) and counts values that are not this iteration's.
// Build: hipcc --offload-arch=gfx950 -O3 -o xcd_barrier_probe xcd_barrier_probe.hip ; ./xcd_barrier_probe [iterations]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Ctl {
    unsigned ticket, pad0[31];
    unsigned arrive[2][32];
    unsigned bad, pad1[31];
    unsigned on_xcc0, pad2[31];
    unsigned timeout;
};
__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 15u;
}
// LOCAL data loads, variant V: 0 = load with sc0; 1 = buffer_inv sc0, then a plain load; 2 = buffer_inv sc1, then a plain load; 3 = load with sc0 and nt
template <bool LOCAL, int V = 0> __device__ __forceinline__ unsigned ld_u32(const unsigned* p) {
    unsigned v;
    if (!LOCAL) asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (V == 0) asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (V == 1) asm volatile("buffer_inv sc0\n\tglobal_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (V == 2) asm volatile("buffer_inv sc1\n\tglobal_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else asm volatile("global_load_dword %0, %1, off sc0 nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
// an atomic add executed in the XCD's L2, returning the old value
__device__ __forceinline__ unsigned l2_add(unsigned* p, unsigned v) {
    unsigned old;
    asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(old) : "v"(p), "v"(v) : "memory");
    return old;
}
template <bool LOCAL> __device__ __forceinline__ void st_u32(unsigned* p, unsigned v) {
    if (LOCAL) asm volatile("global_store_dword %0, %1, off" : : "v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dword %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
}
template <bool LOCAL, int V>
__global__ __launch_bounds__(256) void k_bar(Ctl* c, unsigned* data, int G, int iters) {
    const int tid = threadIdx.x;
    __shared__ int s_me, s_ok;
    if (LOCAL && xcc_id() != 0) return;
    if (tid == 0) {
        s_me = LOCAL ? (int)l2_add(&c->ticket, 1u) : (int)blockIdx.x;
        if (LOCAL) __hip_atomic_fetch_add(&c->on_xcc0, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const int me = s_me;
    if (me >= G) return;
    unsigned bad = 0;
    for (int it = 1; it <= iters; ++it) {
        unsigned* const buf = data + (size_t)(it & 1) * 64 * 256;          // (two buffers in turn: a fast workgroup's next round must not overwrite what a slow one still reads)
        st_u32<LOCAL>(buf + (size_t)me * 256 + tid, (unsigned)it * 1024u + (unsigned)me);
        asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
        __syncthreads();
        if (tid == 0) {
            unsigned* cnt = &c->arrive[0][0];
            if (LOCAL) l2_add(cnt, 1u);
            else __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int good = 0;
            const long long t0 = wall_clock64();
            for (;;) {
                const unsigned seen = LOCAL ? l2_add(cnt, 0u) : ld_u32<false>(cnt);
                if (seen >= (unsigned)it * (unsigned)G) { good = 1; break; }
                if (wall_clock64() - t0 > 2000000) break;           // 20 ms
            }
            s_ok = good;
            if (!good) c->timeout = 1;
        }
        __syncthreads();
        if (!s_ok) return;
        const int nb = (me + 1) % G;
        const unsigned got = ld_u32<LOCAL, V>(buf + (size_t)nb * 256 + tid);
        bad += got != (unsigned)it * 1024u + (unsigned)nb;
        __syncthreads();
    }
    if (bad) atomicAdd(&c->bad, bad);
}
template <bool LOCAL, int V> int run(Ctl* c, unsigned* data, int G, int iters) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipMemset(c, 0, sizeof(Ctl)));
        CHECK(hipMemset(data, 0, 2 * 64 * 1024 * 4));
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k_bar<LOCAL, V>), dim3(LOCAL ? 8 * G : G), dim3(256), 0, 0, c, data, G, iters);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
        Ctl h; CHECK(hipMemcpy(&h, c, sizeof(Ctl), hipMemcpyDeviceToHost));
        printf("%s (loads %d), %2d workgroups: %.2f us per (store + barrier + load) round; stale values %u, timeouts %u%s\n", LOCAL ? "one XCD " : "all XCDs", V, G,
               ms * 1e3 / iters, h.bad, h.timeout, LOCAL ? (h.on_xcc0 == (unsigned)G ? "; placement as expected" : "; PLACEMENT differs") : "");
    }
    return 0;
}
int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    Ctl* c; unsigned* data;
    CHECK(hipMalloc(&c, sizeof(Ctl))); CHECK(hipMalloc(&data, 2 * 64 * 1024 * 4));
    for (int G : {16, 32}) {
        if (run<true, 0>(c, data, G, iters)) return 1;
        if (run<true, 1>(c, data, G, iters)) return 1;
        if (run<true, 2>(c, data, G, iters)) return 1;
        if (run<true, 3>(c, data, G, iters)) return 1;
        if (run<false, 0>(c, data, G, iters)) return 1;
    }
    return 0;
}
