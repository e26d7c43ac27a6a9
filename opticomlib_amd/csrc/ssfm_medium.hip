// ssfm_medium.hip -- the single-launch fixed-step engine of plans of 2^14 ... 2^17 samples (ssfm_kernels.hpp k_medium), in a
// translation unit of its own (its four shapes x two operator-table kinds are a minute of compile time).
#include <hip/hip_runtime.h>

#include <mutex>

#include "ssfm_kernels.hpp"
#include "ssfm_medium.hpp"

namespace ssfm {
namespace {
template <int N1, int N2, int FMODE>
hipError_t launch_shape(int nblk, int xccs, hipStream_t s, const MediumArgs<float>& a) {
    using T = float;
    constexpr int E = 8, C = 16, ROWS = N1 * C / N2;
    constexpr size_t lds_t = (fft_nstages(N1, E) > 1 ? (size_t)N1 * C * sizeof(cx<T>) : 0) + (size_t)E * C * sizeof(cx<T>)
                           + (size_t)fft_tw_lds_entries(N1, E) * sizeof(cx<T>);
    constexpr size_t lds_f = (fft_nstages(N2, E) > 1 ? (size_t)ROWS * row_lds_elems(N2, E) * sizeof(cx<T>) : 0)
                           + (size_t)fft_tw_lds_entries(N2, E) * sizeof(cx<T>);
    constexpr size_t lds = lds_t > lds_f ? lds_t : lds_f;
    static hipError_t attr = lds <= 48 * 1024 ? hipSuccess
        : hipFuncSetAttribute(reinterpret_cast<const void*>(k_medium<T, N1, N2, E, FMODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (attr != hipSuccess) return attr;
    hipLaunchKernelGGL((k_medium<T, N1, N2, E, FMODE>), dim3(xccs * nblk), dim3(N1 * C / E), lds, s, a);
    return hipGetLastError();
}
template <int FMODE> hipError_t launch_mode(int N1, int N2, int nblk, int xccs, hipStream_t s, const MediumArgs<float>& a) {
    if (N1 == 64 && N2 == 64) return launch_shape<64, 64, FMODE>(nblk, xccs, s, a);
    if (N1 == 64 && N2 == 128) return launch_shape<64, 128, FMODE>(nblk, xccs, s, a);
    if (N1 == 128 && N2 == 128) return launch_shape<128, 128, FMODE>(nblk, xccs, s, a);
    if (N1 == 128 && N2 == 256) return launch_shape<128, 256, FMODE>(nblk, xccs, s, a);
    if (N1 == 256 && N2 == 256) return launch_shape<256, 256, FMODE>(nblk, xccs, s, a);
    if (N1 == 256 && N2 == 512) return launch_shape<256, 512, FMODE>(nblk, xccs, s, a);
    return hipErrorInvalidValue;
}
}  // namespace

namespace {
template <int N1, int N2>
hipError_t launch_adapt_shape(int nblk, int xccs, hipStream_t s, const MediumAdaptArgs<float>& a) {
    using T = float;
    constexpr int E = 8, C = 16, ROWS = N1 * C / N2;
    constexpr size_t lds_t = (fft_nstages(N1, E) > 1 ? (size_t)N1 * C * sizeof(cx<T>) : 0) + (size_t)E * C * sizeof(cx<T>)
                           + (size_t)fft_tw_lds_entries(N1, E) * sizeof(cx<T>);
    constexpr size_t lds_f = (fft_nstages(N2, E) > 1 ? (size_t)ROWS * row_lds_elems(N2, E) * sizeof(cx<T>) : 0)
                           + (size_t)fft_tw_lds_entries(N2, E) * sizeof(cx<T>);
    constexpr size_t lds = lds_t > lds_f ? lds_t : lds_f;
    static hipError_t attr = lds <= 48 * 1024 ? hipSuccess
        : hipFuncSetAttribute(reinterpret_cast<const void*>(k_medium_adapt<T, N1, N2, E>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (attr != hipSuccess) return attr;
    hipLaunchKernelGGL((k_medium_adapt<T, N1, N2, E>), dim3(xccs * nblk), dim3(N1 * C / E), lds, s, a);
    return hipGetLastError();
}
}  // namespace
hipError_t launch_medium_adapt(int N1, int N2, int nblk, int xccs, hipStream_t s, const MediumAdaptArgs<float>& a) {
    if (N1 == 64 && N2 == 64) return launch_adapt_shape<64, 64>(nblk, xccs, s, a);
    if (N1 == 64 && N2 == 128) return launch_adapt_shape<64, 128>(nblk, xccs, s, a);
    if (N1 == 128 && N2 == 128) return launch_adapt_shape<128, 128>(nblk, xccs, s, a);
    if (N1 == 128 && N2 == 256) return launch_adapt_shape<128, 256>(nblk, xccs, s, a);
    if (N1 == 256 && N2 == 256) return launch_adapt_shape<256, 256>(nblk, xccs, s, a);
    if (N1 == 256 && N2 == 512) return launch_adapt_shape<256, 512>(nblk, xccs, s, a);
    return hipErrorInvalidValue;
}

namespace {
template <int N1, int N2>
hipError_t launch_chirp_shape(int nblk, int xccs, hipStream_t s, const MediumChirpArgs<float>& a) {
    using T = float;
    constexpr int E = 8, C = 16, ROWS = N1 * C / N2;
    constexpr size_t lds_t = (fft_nstages(N1, E) > 1 ? (size_t)N1 * C * sizeof(cx<T>) : 0) + (size_t)E * C * sizeof(cx<T>)
                           + (size_t)fft_tw_lds_entries(N1, E) * sizeof(cx<T>);
    constexpr size_t lds_f = (fft_nstages(N2, E) > 1 ? (size_t)ROWS * row_lds_elems(N2, E) * sizeof(cx<T>) : 0)
                           + (size_t)fft_tw_lds_entries(N2, E) * sizeof(cx<T>);
    constexpr size_t lds = lds_t > lds_f ? lds_t : lds_f;
    static hipError_t attr = lds <= 48 * 1024 ? hipSuccess
        : hipFuncSetAttribute(reinterpret_cast<const void*>(k_medium_chirp<T, N1, N2, E>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (attr != hipSuccess) return attr;
    hipLaunchKernelGGL((k_medium_chirp<T, N1, N2, E>), dim3(xccs * nblk), dim3(N1 * C / E), lds, s, a);
    return hipGetLastError();
}
}  // namespace
hipError_t launch_medium_chirp(int N1, int N2, int nblk, int xccs, hipStream_t s, const MediumChirpArgs<float>& a) {
    if (N1 == 64 && N2 == 64) return launch_chirp_shape<64, 64>(nblk, xccs, s, a);
    if (N1 == 64 && N2 == 128) return launch_chirp_shape<64, 128>(nblk, xccs, s, a);
    if (N1 == 128 && N2 == 128) return launch_chirp_shape<128, 128>(nblk, xccs, s, a);
    if (N1 == 128 && N2 == 256) return launch_chirp_shape<128, 256>(nblk, xccs, s, a);
    if (N1 == 256 && N2 == 256) return launch_chirp_shape<256, 256>(nblk, xccs, s, a);
    if (N1 == 256 && N2 == 512) return launch_chirp_shape<256, 512>(nblk, xccs, s, a);
    return hipErrorInvalidValue;
}

namespace {
template <int N1, int N2>
hipError_t launch_chirp_adapt_shape(int nblk, int xccs, hipStream_t s, const MediumChirpAdaptArgs<float>& a) {
    using T = float;
    constexpr int E = 8, C = 16, ROWS = N1 * C / N2;
    constexpr size_t lds_t = (fft_nstages(N1, E) > 1 ? (size_t)N1 * C * sizeof(cx<T>) : 0) + (size_t)E * C * sizeof(cx<T>)
                           + (size_t)fft_tw_lds_entries(N1, E) * sizeof(cx<T>);
    constexpr size_t lds_f = (fft_nstages(N2, E) > 1 ? (size_t)ROWS * row_lds_elems(N2, E) * sizeof(cx<T>) : 0)
                           + (size_t)fft_tw_lds_entries(N2, E) * sizeof(cx<T>);
    constexpr size_t lds = lds_t > lds_f ? lds_t : lds_f;
    static hipError_t attr = lds <= 48 * 1024 ? hipSuccess
        : hipFuncSetAttribute(reinterpret_cast<const void*>(k_medium_chirp_adapt<T, N1, N2, E>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (attr != hipSuccess) return attr;
    hipLaunchKernelGGL((k_medium_chirp_adapt<T, N1, N2, E>), dim3(xccs * nblk), dim3(N1 * C / E), lds, s, a);
    return hipGetLastError();
}
}  // namespace
hipError_t launch_medium_chirp_adapt(int N1, int N2, int nblk, int xccs, hipStream_t s, const MediumChirpAdaptArgs<float>& a) {
    if (N1 == 64 && N2 == 64) return launch_chirp_adapt_shape<64, 64>(nblk, xccs, s, a);
    if (N1 == 64 && N2 == 128) return launch_chirp_adapt_shape<64, 128>(nblk, xccs, s, a);
    if (N1 == 128 && N2 == 128) return launch_chirp_adapt_shape<128, 128>(nblk, xccs, s, a);
    if (N1 == 128 && N2 == 256) return launch_chirp_adapt_shape<128, 256>(nblk, xccs, s, a);
    if (N1 == 256 && N2 == 256) return launch_chirp_adapt_shape<256, 256>(nblk, xccs, s, a);
    if (N1 == 256 && N2 == 512) return launch_chirp_adapt_shape<256, 512>(nblk, xccs, s, a);
    return hipErrorInvalidValue;
}

bool medium_shape(int N1, int N2) { return (N1 == 64 && (N2 == 64 || N2 == 128)) || (N1 == 128 && (N2 == 128 || N2 == 256)) || (N1 == 256 && (N2 == 256 || N2 == 512)); }

hipError_t launch_medium(int N1, int N2, bool phase_tables, int nblk, int xccs, hipStream_t s, const MediumArgs<float>& a) {
    return phase_tables ? launch_mode<FM_PHASE>(N1, N2, nblk, xccs, s, a) : launch_mode<FM_TABLE>(N1, N2, nblk, xccs, s, a);
}

namespace {
__global__ void k_xcc_probe(unsigned* mask) {
    if (threadIdx.x == 0) atomicOr(mask, 1u << xcc_id());
}
}  // namespace
unsigned xcc_mask(int device) {
    static std::mutex mu;
    static unsigned cache[64] = {0};
    if (device < 0 || device >= 64) return 0u;
    std::lock_guard<std::mutex> lock(mu);
    if (cache[device]) return cache[device];
    int cur = 0;
    unsigned* d = nullptr;
    unsigned h = 0u;
    hipStream_t s = nullptr;            // (a stream of its own: nothing of this library runs on the legacy stream)
    if (hipGetDevice(&cur) != hipSuccess || cur != device || hipMalloc(&d, sizeof(unsigned)) != hipSuccess) return 0u;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) == hipSuccess) {
        if (hipMemsetAsync(d, 0, sizeof(unsigned), s) == hipSuccess) {
            hipLaunchKernelGGL(k_xcc_probe, dim3(256), dim3(64), 0, s, d);
            if (hipMemcpyAsync(&h, d, sizeof(unsigned), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) h = 0u;
        }
        (void)hipStreamDestroy(s);
    }
    (void)hipFree(d);
    if (h == 0u) return 0u;             // (not cached: a later plan may try again)
    return cache[device] = h;
}
}  // namespace ssfm
