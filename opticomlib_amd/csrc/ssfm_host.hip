// ssfm_host.hip -- host side of the C ABI declared in include/ssfm_amd.h: plan, buffers,
// operator tables, launch sequences.  Pure HIP runtime; no torch types anywhere.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <memory>
#include <cmath>
#include <cstdarg>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "ssfm_amd.h"
#include "ssfm_common.hpp"
#include "ssfm_kernels.hpp"
#include "ssfm_medium.hpp"
#include "ssfm_split.hpp"

using namespace ssfm;

namespace {

constexpr int kLog2Min = 8, kLog2Max = 24;      // (above 2^22: split plans, ssfm_split.hpp)
constexpr int kLog2MaxDirect = 22;                // the two-kernel engine's own range: N1 <= 512 column points x N2 <= 8192 row points
// C, columns per k_time tile: one 128-byte line per row segment in either precision.  (complex128 used 16 as well
// at first: 512-thread workgroups, one per CU; with 8 the two precisions have the same workgroup shape, two per CU.)
template <typename T> constexpr int cols_per_tile() { return sizeof(T) == 8 ? 8 : 16; }
constexpr int kMaxTables = 4;

// rows per k_freq workgroup: at least 64 threads where the row count allows (N1 >= 16 rows per batch entry)
constexpr int freq_rows(int n2, int e) { return (n2 / e) >= 64 ? 1 : ((64 * e / n2) > 16 ? 16 : (64 * e / n2)); }

template <typename K> hipError_t allow_lds(K kernel, size_t bytes) {
    if (bytes <= 48 * 1024) return hipSuccess;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

// ----------------------------------------------------------------------------- launchers
// E = points per thread: 16 (default) or 8 (twice the waves, half the registers, one more LDS exchange).
template <typename T, int MODE, int N1, int E>
hipError_t launch_time_n1(dim3 grid, hipStream_t s, const TimeArgs<T>& a) {
    constexpr int C = cols_per_tile<T>();
    constexpr bool U16 = u16_layout<T>(N1, C, E);          // the field layout between the kernels (ssfm_kernels.hpp "U16")
    if constexpr (!U16 && MODE > TM_END && MODE != TM_MID_A && MODE != TM_MID_L) return hipErrorInvalidValue;      // (tile-private time-domain modes: U16 plans only)
    else if constexpr (U16 && MODE == TM_MID_L) return hipErrorInvalidValue;                    // (as a launch of its own: plans in the plain layout)
    else if constexpr (MODE == TM_MID_A && N1 != 128 && N1 != 256) return hipErrorInvalidValue;   // (the fused adaptive kernel: plans of 2^14 ... 2^18 samples)
    else {
    constexpr size_t lds = (fft_nstages(N1, E) > 1 ? (size_t)N1 * C * sizeof(cx<T>) : 0) + (size_t)E * C * sizeof(cx<T>)
                         + (size_t)fft_tw_lds_entries(N1, E) * sizeof(cx<T>);
    if constexpr (MODE == TM_BEGIN || MODE == TM_MID || MODE == TM_END) {
        if (a.scal != nullptr) {             // a capture run with the scalar log: the instantiation that keeps it (ssfm_kernels.hpp time_body, LOG)
            static hipError_t attr_log = allow_lds(k_time<T, N1, C, E, MODE, U16, true>, lds);
            if (attr_log != hipSuccess) return attr_log;
            hipLaunchKernelGGL((k_time<T, N1, C, E, MODE, U16, true>), grid, dim3(N1 * C / E), lds, s, SSFM_TIME_KERNEL_ARGS(a));
            return hipGetLastError();
        }
    }
    static hipError_t attr = allow_lds(k_time<T, N1, C, E, MODE, U16>, lds);
    if (attr != hipSuccess) return attr;
    hipLaunchKernelGGL((k_time<T, N1, C, E, MODE, U16>), grid, dim3(N1 * C / E), lds, s, SSFM_TIME_KERNEL_ARGS(a));
    return hipGetLastError();
    }
}
template <typename T, int MODE, int E>
hipError_t launch_time_e(int N1, int batch, hipStream_t s, const TimeArgs<T>& a) {
    const dim3 grid((unsigned)(a.N2 / cols_per_tile<T>()) * batch);
    switch (N1) {
        case 16:  return launch_time_n1<T, MODE, 16, E>(grid, s, a);
        case 32:  return launch_time_n1<T, MODE, 32, E>(grid, s, a);
        case 64:  return launch_time_n1<T, MODE, 64, E>(grid, s, a);
        case 128: return launch_time_n1<T, MODE, 128, E>(grid, s, a);
        case 256: return launch_time_n1<T, MODE, 256, E>(grid, s, a);
        case 512: if constexpr (E == 16) return launch_time_n1<T, MODE, 512, E>(grid, s, a); else break;
    }
    return hipErrorInvalidValue;
}
template <typename T, int MODE>
hipError_t launch_time(int N1, int batch, hipStream_t s, TimeArgs<T> a, int E) {
    a.rows = batch;
    return E == 8 ? launch_time_e<T, MODE, 8>(N1, batch, s, a) : launch_time_e<T, MODE, 16>(N1, batch, s, a);
}

template <typename T, int MODE, int N2, int E, bool U16>
hipError_t launch_freq_u(int nrows, hipStream_t s, const FreqArgs<T>& a) {
    constexpr int ROWS = freq_rows(N2, E);
    constexpr size_t lds = (fft_nstages(N2, E) > 1 ? (size_t)ROWS * row_lds_elems(N2, E) * sizeof(cx<T>) : 0)
                         + (size_t)fft_tw_lds_entries(N2, E) * sizeof(cx<T>);
    if constexpr (MODE == FM_FWD_ONLY || MODE == FM_INV_ONLY) {
        static hipError_t attr_h = allow_lds(k_freq_half<T, N2, ROWS, E, MODE, U16>, lds);
        if (attr_h != hipSuccess) return attr_h;
        hipLaunchKernelGGL((k_freq_half<T, N2, ROWS, E, MODE, U16>), dim3(nrows / ROWS), dim3(ROWS * N2 / E), lds, s, SSFM_FREQ_KERNEL_ARGS(a));
        return hipGetLastError();
    } else {
    static hipError_t attr = allow_lds(k_freq<T, N2, ROWS, E, MODE, U16>, lds);
    if (attr != hipSuccess) return attr;
    hipLaunchKernelGGL((k_freq<T, N2, ROWS, E, MODE, U16>), dim3(nrows / ROWS), dim3(ROWS * N2 / E), lds, s, SSFM_FREQ_KERNEL_ARGS(a));
    return hipGetLastError();
    }
}
template <typename T, int MODE, int N2, int E>
hipError_t launch_freq_n2(int nrows, hipStream_t s, const FreqArgs<T>& a) {
    // the layout is the plan's (it follows from k_time's tile shape); complex128 plans never use it, and the
    // forward-only transform writes the plain transposed spectrum for its consumers
    if constexpr (sizeof(T) == 4)
        if (a.u16) return launch_freq_u<T, MODE, N2, E, true>(nrows, s, a);
    if constexpr ((MODE == FM_PHASE && sizeof(T) == 4) || MODE == FM_FLY_IM) return hipErrorInvalidValue;      // (complex64 phase / imaginary-part tables: plans in the unit layout only)
    else return launch_freq_u<T, MODE, N2, E, false>(nrows, s, a);
}
template <typename T, int MODE, int E>
hipError_t launch_freq_e(int N2, int nrows, hipStream_t s, const FreqArgs<T>& a) {
    if constexpr (MODE == FM_INV_ONLY) {          // (the inverse half of a split plan's row pass: its sub-sequences are 2^20 ... 2^22 samples long)
        switch (N2) {
            case 4096: return launch_freq_n2<T, MODE, 4096, E>(nrows, s, a);
            case 8192: if constexpr (E == 16) return launch_freq_n2<T, MODE, 8192, E>(nrows, s, a); else break;
        }
        return hipErrorInvalidValue;
    } else
    switch (N2) {
        case 16:   return launch_freq_n2<T, MODE, 16, E>(nrows, s, a);
        case 32:   return launch_freq_n2<T, MODE, 32, E>(nrows, s, a);
        case 64:   return launch_freq_n2<T, MODE, 64, E>(nrows, s, a);
        case 128:  return launch_freq_n2<T, MODE, 128, E>(nrows, s, a);
        case 256:  return launch_freq_n2<T, MODE, 256, E>(nrows, s, a);
        case 512:  return launch_freq_n2<T, MODE, 512, E>(nrows, s, a);
        case 1024: return launch_freq_n2<T, MODE, 1024, E>(nrows, s, a);
        case 2048: return launch_freq_n2<T, MODE, 2048, E>(nrows, s, a);
        case 4096: return launch_freq_n2<T, MODE, 4096, E>(nrows, s, a);
        case 8192: if constexpr (E == 16) return launch_freq_n2<T, MODE, 8192, E>(nrows, s, a); else break;
    }
    return hipErrorInvalidValue;
}
template <typename T, int MODE>
hipError_t launch_freq(int N2, int nrows, hipStream_t s, FreqArgs<T> a, int E) {
    a.rows = nrows / a.N1;
    return E == 8 ? launch_freq_e<T, MODE, 8>(N2, nrows, s, a) : launch_freq_e<T, MODE, 16>(N2, nrows, s, a);
}

// The passes of a chirp-z line whose field is stored as complex64 on a complex128 plan (ssfm_kernels.hpp time_body / freq_body, H): the shapes such plans have
// from 2^18 points up -- 256 column points x 8 per thread, 512 x 16; rows of 1024 ... 8192 points x 16 per thread.
template <int MODE, int N1, int E>
hipError_t launch_time_h_n1(dim3 grid, hipStream_t s, const TimeArgs<double>& a) {
    constexpr int C = cols_per_tile<double>();
    constexpr size_t lds = (fft_nstages(N1, E) > 1 ? (size_t)N1 * C * sizeof(cx<double>) : 0) + (size_t)E * C * sizeof(cx<double>)
                         + (size_t)fft_tw_lds_entries(N1, E) * sizeof(cx<double>);
    static hipError_t attr = allow_lds(k_time_h<double, N1, C, E, MODE>, lds);
    if (attr != hipSuccess) return attr;
    hipLaunchKernelGGL((k_time_h<double, N1, C, E, MODE>), grid, dim3(N1 * C / E), lds, s, SSFM_TIME_KERNEL_ARGS(a));
    return hipGetLastError();
}
constexpr bool line_half_shape(int N1, int N2, int E, int Ef) { return ((N1 == 256 && E == 8) || (N1 == 512 && E == 16)) && Ef == 16 && N2 >= 1024 && N2 <= 8192; }
template <int MODE>
hipError_t launch_time_h(int N1, int batch, hipStream_t s, TimeArgs<double> a, int E) {
    a.rows = batch;
    const dim3 grid((unsigned)(a.N2 / cols_per_tile<double>()) * batch);
    if (N1 == 256 && E == 8) return launch_time_h_n1<MODE, 256, 8>(grid, s, a);
    if (N1 == 512 && E == 16) return launch_time_h_n1<MODE, 512, 16>(grid, s, a);
    return hipErrorInvalidValue;
}
template <int N2>
hipError_t launch_freq_h_n2(int nrows, hipStream_t s, const FreqArgs<double>& a, const int* done) {
    constexpr int E = 16, ROWS = freq_rows(N2, E);
    constexpr size_t lds = (fft_nstages(N2, E) > 1 ? (size_t)ROWS * row_lds_elems(N2, E) * sizeof(cx<double>) : 0) + (size_t)fft_tw_lds_entries(N2, E) * sizeof(cx<double>);
    static hipError_t attr = allow_lds(k_freq_h<double, N2, ROWS, E, FM_TABLE>, lds);
    if (attr != hipSuccess) return attr;
#if SSFM_TRACE
    hipLaunchKernelGGL((k_freq_h<double, N2, ROWS, E, FM_TABLE>), dim3(nrows / ROWS), dim3(ROWS * N2 / E), lds, s, a.F, a.tab, a.tw2, a.st, a.h, a.amp, a.inv_n, a.step, a.N1, a.rows, a.u16, done,
                       a.trace, a.trace_slot);
#else
    hipLaunchKernelGGL((k_freq_h<double, N2, ROWS, E, FM_TABLE>), dim3(nrows / ROWS), dim3(ROWS * N2 / E), lds, s, a.F, a.tab, a.tw2, a.st, a.h, a.amp, a.inv_n, a.step, a.N1, a.rows, a.u16, done);
#endif
    return hipGetLastError();
}
inline hipError_t launch_freq_h(int N2, int nrows, hipStream_t s, FreqArgs<double> a, const int* done = nullptr) {
    a.rows = nrows / a.N1;
    switch (N2) {
        case 1024: return launch_freq_h_n2<1024>(nrows, s, a, done);
        case 2048: return launch_freq_h_n2<2048>(nrows, s, a, done);
        case 4096: return launch_freq_h_n2<4096>(nrows, s, a, done);
        case 8192: return launch_freq_h_n2<8192>(nrows, s, a, done);
    }
    return hipErrorInvalidValue;
}

template <typename T, int MODE, int R>
hipError_t launch_split_mid_r(hipStream_t s, const SplitArgs<T>& a) {
    constexpr int V = split_positions<T, R>();
    const long long units = (long long)a.N1 * a.N2 / V;
    hipLaunchKernelGGL((k_split_mid<T, R, MODE>), dim3((unsigned)((units + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}
template <typename T, int MODE>
hipError_t launch_split_mid(int R, hipStream_t s, const SplitArgs<T>& a) {
    switch (R) {
        case 2:  return launch_split_mid_r<T, MODE, 2>(s, a);
        case 4:  return launch_split_mid_r<T, MODE, 4>(s, a);
        case 8:  return launch_split_mid_r<T, MODE, 8>(s, a);
        case 16: return launch_split_mid_r<T, MODE, 16>(s, a);
    }
    return hipErrorInvalidValue;
}
template <typename T, bool TO_SUB>
hipError_t launch_split_shuffle(int R, hipStream_t s, cx<T>* sub, cx<T>* nat, long long M, int rows) {
    const dim3 grid((unsigned)((M * rows + 255) / 256));
    switch (R) {
        case 2:  hipLaunchKernelGGL((k_split_shuffle<T, 2, TO_SUB>), grid, dim3(256), 0, s, sub, nat, M, rows); break;
        case 4:  hipLaunchKernelGGL((k_split_shuffle<T, 4, TO_SUB>), grid, dim3(256), 0, s, sub, nat, M, rows); break;
        case 8:  hipLaunchKernelGGL((k_split_shuffle<T, 8, TO_SUB>), grid, dim3(256), 0, s, sub, nat, M, rows); break;
        case 16: hipLaunchKernelGGL((k_split_shuffle<T, 16, TO_SUB>), grid, dim3(256), 0, s, sub, nat, M, rows); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// k_small: one workgroup per row, the whole schedule in one launch.  Points per thread: 8 up to 2048 samples and for
// complex128 (more wavefronts per row; complex128 with 16 would not fit the registers), 16 for complex64 rows of 4096 / 8192.
constexpr int kSmallE16Min = 4096;
template <typename T> constexpr int small_points(int n) { return sizeof(T) == 4 && n >= kSmallE16Min ? 16 : 8; }
template <typename T> constexpr bool small_supported(int n) {
    return n >= 256 && n <= (sizeof(T) == 4 ? 8192 : 4096) && (n & (n - 1)) == 0;
}
static_assert(kSmallTabs == kMaxTables, "k_small takes one table per cached step size");
template <typename T, int N>
hipError_t launch_small_n(int rows, hipStream_t s, const SmallArgs<T>& a) {
    constexpr int E = small_points<T>(N);
    constexpr size_t lds = (fft_nstages(N, E) > 1 ? (size_t)row_lds_elems(N, E) * sizeof(cx<T>) : 0) + (size_t)fft_tw_lds_entries(N, E) * sizeof(cx<T>);
    static hipError_t attr = allow_lds(k_small<T, N, E>, lds);
    if (attr != hipSuccess) return attr;
    hipLaunchKernelGGL((k_small<T, N, E>), dim3(rows), dim3(N / E), lds, s, a);
    return hipGetLastError();
}
template <typename T>
hipError_t launch_small(int n, int rows, hipStream_t s, const SmallArgs<T>& a) {
    switch (n) {
        case 256:  return launch_small_n<T, 256>(rows, s, a);
        case 512:  return launch_small_n<T, 512>(rows, s, a);
        case 1024: return launch_small_n<T, 1024>(rows, s, a);
        case 2048: return launch_small_n<T, 2048>(rows, s, a);
        case 4096: return launch_small_n<T, 4096>(rows, s, a);
        case 8192: if constexpr (sizeof(T) == 4) return launch_small_n<T, 8192>(rows, s, a); else break;
    }
    return hipErrorInvalidValue;
}

// ---- kernels around k_medium_chirp (chirp_medium below): a complex64 line, the chirp's phase reduced exactly in integers as in chirpz.hip
template <typename T> __device__ __forceinline__ cx<T> cm_chirp_value(long long m, long long n, int conj) {
    const long long r = (m * m) % (2 * n);
    double sn, cs;
    sincospi(-(double)r / (double)n, &sn, &cs);
    return mk<T>((T)cs, (T)(conj ? -sn : sn));
}
// row 0 of the field <- the convolution kernel of Bluestein's identity wrapped around the line: conj(c_m) (which = 0) or c_m (1) at m and M - m, m < n
template <typename T> __global__ __launch_bounds__(256) void k_cm_kernel_line(cx<T>* __restrict__ F, long long n, long long M, int which) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += (long long)gridDim.x * blockDim.x) {
        const long long m = i < n ? i : (M - i < n ? M - i : -1);
        F[i] = m < 0 ? mk<T>((T)0, (T)0) : cm_chirp_value<T>(m, n, which == 0);
    }
}
// out[m] = exp(D~_m h) / n below n, zero above: the products in T as the reference forms them (complex64 * float32, devices.py:1179), the
// transcendental functions in double and rounded once (as k_make_freq_table)
template <typename T> __global__ __launch_bounds__(256) void k_cm_mul_table(const cx<T>* __restrict__ Dt, cx<T>* __restrict__ out, long long n, long long M, T h, T inv) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += (long long)gridDim.x * blockDim.x) {
        cx<T> d = mk<T>((T)0, (T)0);
        if (i < n) {
            const cx<T> x = Dt[i];
            const T xr = x.x * h, xi = x.y * h;
            const T e = (T)exp((double)xr);
            double sn, cs;
            sincos((double)xi, &sn, &cs);
            d = mk<T>((e * (T)cs) * inv, (e * (T)sn) * inv);
        }
        out[i] = d;
    }
}
__global__ __launch_bounds__(256) void k_cm_narrow(const cx<double>* __restrict__ src, cx<float>* __restrict__ dst, long long M, int conj) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += (long long)gridDim.x * blockDim.x)
        dst[i] = mk<float>((float)src[i].x, (float)(conj ? -src[i].y : src[i].y));
}
template <typename T> __global__ __launch_bounds__(256) void k_cm_pre(const cx<T>* __restrict__ A, const cx<T>* __restrict__ chirp, cx<T>* __restrict__ F, long long n, long long M, int batch) {
    const long long total = M * batch;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / M, m = i - r * M;
        F[i] = m < n ? cmul(A[r * n + m], chirp[m]) : mk<T>((T)0, (T)0);
    }
}
// A <- line conj(c); nothing when the run's workgroups did not meet (*error != 0): the caller's field is then as it was
template <typename T> __global__ __launch_bounds__(256) void k_cm_post(cx<T>* __restrict__ A, const cx<T>* __restrict__ chirp, const cx<T>* __restrict__ F, long long n, long long M, int batch,
                                                                      const unsigned* __restrict__ error) {
    if (*error != 0u) return;
    const long long total = n * batch;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / n, m = i - r * n;
        A[i] = cmulc(F[r * M + m], chirp[m]);
    }
}

// k_small_chirp: lengths that are not powers of two on one line of the plan's length (complex128), the whole schedule in one launch
// Points per thread: as k_small.  With 8 points on 512 threads the kernels of the 4096-point line spill (320 bytes of scratch per lane: a workgroup
// of 512 threads has 256 registers per thread); with 16 points on 256 threads (measured in round 4) they do not (256 + 120 accumulation registers),
// but four wavefronts per workgroup hide less: fixed step 12.6 us per step at n = 2032 either way, adaptive 23.3 against 16.7 (eight exp(D~ h) per
// thread and step instead of four).  So 8.
template <typename T> constexpr int chirp_points(int n) { return small_points<T>(n); }
template <typename T, int N>
hipError_t launch_small_chirp_n(int rows, hipStream_t s, const SmallChirpArgs<T>& a) {
    constexpr int E = chirp_points<T>(N);
    constexpr size_t lds = (fft_nstages(N, E) > 1 ? (size_t)row_lds_elems(N, E) * sizeof(cx<T>) : 0) + (size_t)fft_tw_lds_entries(N, E) * sizeof(cx<T>) + (size_t)N * sizeof(cx<T>);
    static_assert(lds <= 160 * 1024, "k_small_chirp: the line, its twiddles and H must fit the LDS");
    static hipError_t attr = allow_lds(k_small_chirp<T, N, E>, lds);
    if (attr != hipSuccess) return attr;
    hipLaunchKernelGGL((k_small_chirp<T, N, E>), dim3(rows), dim3(N / E), lds, s, a);
    return hipGetLastError();
}
template <typename T>
hipError_t launch_small_chirp(int n, int rows, hipStream_t s, const SmallChirpArgs<T>& a) {
    switch (n) {
        case 256:  return launch_small_chirp_n<T, 256>(rows, s, a);
        case 512:  return launch_small_chirp_n<T, 512>(rows, s, a);
        case 1024: return launch_small_chirp_n<T, 1024>(rows, s, a);
        case 2048: return launch_small_chirp_n<T, 2048>(rows, s, a);
        case 4096: return launch_small_chirp_n<T, 4096>(rows, s, a);
    }
    return hipErrorInvalidValue;
}

template <typename T, int N>
hipError_t launch_small_chirp_adapt_n(int rows, hipStream_t s, const SmallChirpAdaptArgs<T>& a) {
    constexpr int E = chirp_points<T>(N);
    constexpr size_t lds = (fft_nstages(N, E) > 1 ? (size_t)row_lds_elems(N, E) * sizeof(cx<T>) : 0) + (size_t)fft_tw_lds_entries(N, E) * sizeof(cx<T>) + (size_t)N * sizeof(cx<T>);
    static_assert(lds + 256 <= 160 * 1024, "k_small_chirp_adapt: the line, its twiddles and H must fit the LDS");
    static hipError_t attr = allow_lds(k_small_chirp_adapt<T, N, E>, lds);
    if (attr != hipSuccess) return attr;
    hipLaunchKernelGGL((k_small_chirp_adapt<T, N, E>), dim3(rows), dim3(N / E), lds, s, a);
    return hipGetLastError();
}
template <typename T>
hipError_t launch_small_chirp_adapt(int n, int rows, hipStream_t s, const SmallChirpAdaptArgs<T>& a) {
    switch (n) {
        case 256:  return launch_small_chirp_adapt_n<T, 256>(rows, s, a);
        case 512:  return launch_small_chirp_adapt_n<T, 512>(rows, s, a);
        case 1024: return launch_small_chirp_adapt_n<T, 1024>(rows, s, a);
        case 2048: return launch_small_chirp_adapt_n<T, 2048>(rows, s, a);
        case 4096: return launch_small_chirp_adapt_n<T, 4096>(rows, s, a);
    }
    return hipErrorInvalidValue;
}

// k_small_adapt: all rows of the plan in ONE workgroup (they share the step size); at most 512 threads, so that a thread
// may use the whole register file
template <typename T> constexpr bool small_adapt_supported(int n, int rows) {
    return small_supported<T>(n) && n <= 4096 && rows >= 1 && rows <= 2 && rows * n / small_points<T>(n) <= 512;     // (8192 x 1: 14.8 us per step, chunked 12.7)
}
template <typename T, int N, int ROWS>
hipError_t launch_small_adapt_nr(hipStream_t s, const SmallAdaptArgs<T>& a) {
    constexpr int E = small_points<T>(N);
    if constexpr (ROWS * N / E <= 512) {
        constexpr size_t lds = (fft_nstages(N, E) > 1 ? (size_t)ROWS * row_lds_elems(N, E) * sizeof(cx<T>) : 0) + (size_t)fft_tw_lds_entries(N, E) * sizeof(cx<T>);
        static hipError_t attr = allow_lds(k_small_adapt<T, N, E, ROWS>, lds);
        if (attr != hipSuccess) return attr;
        hipLaunchKernelGGL((k_small_adapt<T, N, E, ROWS>), dim3(1), dim3(ROWS * N / E), lds, s, a);
        return hipGetLastError();
    } else {
        return hipErrorInvalidValue;
    }
}
template <typename T, int N>
hipError_t launch_small_adapt_n(int rows, hipStream_t s, const SmallAdaptArgs<T>& a) {
    return rows == 1 ? launch_small_adapt_nr<T, N, 1>(s, a) : launch_small_adapt_nr<T, N, 2>(s, a);
}
template <typename T>
hipError_t launch_small_adapt(int n, int rows, hipStream_t s, const SmallAdaptArgs<T>& a) {
    switch (n) {
        case 256:  return launch_small_adapt_n<T, 256>(rows, s, a);
        case 512:  return launch_small_adapt_n<T, 512>(rows, s, a);
        case 1024: return launch_small_adapt_n<T, 1024>(rows, s, a);
        case 2048: return launch_small_adapt_n<T, 2048>(rows, s, a);
        case 4096: return launch_small_adapt_n<T, 4096>(rows, s, a);
    }
    return hipErrorInvalidValue;
}

// A host thread that enqueues ONE lane's launches.  The host issues a launch per ~3.1 us, a 2^20 x 2 step is four launches per
// 17 us: one thread is 72 % busy with it and falls behind as soon as something slows it down -- in a process that has initialised
// RCCL a launch takes ~4.6 us and the run becomes host-bound (18.3 instead of 17.1 us per step, profiles/r03_host_enqueue.txt).
// With the lanes on threads of their own each thread issues two launches per step.
class LaneWorker {
  public:
    ~LaneWorker() { stop(); }
    void start(int device) {
        if (th_.joinable()) return;
        th_ = std::thread([this, device] {
            (void)hipSetDevice(device);
            std::unique_lock<std::mutex> lk(m_);
            for (;;) {
                cv_.wait(lk, [this] { return has_job_ || quit_; });
                if (!has_job_) return;                       // (quit_: a job handed over but not yet picked up still runs -- a capture run's transfers, a lane's launches)
                std::function<int()> job = std::move(job_);
                has_job_ = false;
                lk.unlock();
                const int rc = job();
                std::string err = rc != SSFM_OK ? std::string(ssfm::g_err) : std::string();
                lk.lock();
                rc_ = rc; err_ = std::move(err); done_ = true;
                cv_.notify_all();
            }
        });
    }
    void submit(std::function<int()> job) {
        std::lock_guard<std::mutex> lk(m_);
        job_ = std::move(job); has_job_ = true; done_ = false;
        cv_.notify_all();
    }
    int wait() {                      // the job's status; its error text becomes this thread's
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [this] { return done_; });
        if (rc_ != SSFM_OK) std::snprintf(ssfm::g_err, sizeof(ssfm::g_err), "%s", err_.c_str());
        return rc_;
    }
    void stop() {
        { std::lock_guard<std::mutex> lk(m_); quit_ = true; cv_.notify_all(); }
        if (th_.joinable()) th_.join();
    }
  private:
    std::thread th_;
    std::mutex m_;
    std::condition_variable cv_;
    std::function<int()> job_;
    bool has_job_ = false, done_ = false, quit_ = false;
    int rc_ = SSFM_OK;
    std::string err_;
};

// ----------------------------------------------------------------------------- do two streams share a hardware queue?
// The HIP runtime maps the streams of a priority class onto at most GPU_MAX_HW_QUEUES (4) hardware queues, and a stream created when the class's
// queues are all taken shares one -- possibly the one of the SAME plan's other lane (seen whenever 3 mod 4 other streams of the class were alive:
// profiles/r04_order_dependence.txt).  Kernels of one queue run in order: the plan's lanes would no longer overlap (35 instead of 21 us per step
// in round 1's measurement), and the round-3 engines whose lanes waited for each other inside a kernel stalled until their patience ran out.
// Probe: a kernel on stream a waits (at most `patience` ticks of the 100 MHz clock) for a word that a kernel on stream b sets.
__global__ void k_queue_probe_wait(unsigned* flag, unsigned* seen, long long patience) {
    const long long t0 = wall_clock64();
    unsigned v = 0u;
    while ((v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0u && wall_clock64() - t0 < patience) __builtin_amdgcn_s_sleep(8);
    *seen = v;
}
__global__ void k_queue_probe_set(unsigned* flag) { __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__global__ void k_queue_probe_spin(long long ticks) {            // a kernel that takes `ticks` of the 100 MHz clock, like a real one takes its microseconds
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(4);
}
// 1: b's kernel ran beside a's (queues of their own); 0: it did not within 0.2 ms (one queue -- or a GPU too busy to tell); < 0: HIP error
int streams_run_side_by_side(hipStream_t a, hipStream_t b, unsigned* words /* two device words */, hipEvent_t ev) {
    unsigned seen = 0u;
    if (hipMemsetAsync(words, 0, 2 * sizeof(unsigned), a) != hipSuccess) return -1;
    if (hipEventRecord(ev, a) != hipSuccess || hipStreamWaitEvent(b, ev, 0) != hipSuccess) return -1;
    hipLaunchKernelGGL(k_queue_probe_wait, dim3(1), dim3(1), 0, a, words, words + 1, 20000ll);
    // (a second kernel queued behind the waiting one, as every launch of a run has one, does not change the verdict: tried for the slow state that
    // lane_chain_us below is there for)
    hipLaunchKernelGGL(k_queue_probe_set, dim3(1), dim3(1), 0, b, words);
    if (hipGetLastError() != hipSuccess) return -1;
    if (hipMemcpyAsync(&seen, words + 1, sizeof(unsigned), hipMemcpyDeviceToHost, a) != hipSuccess) return -1;
    if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) return -1;
    return seen != 0u ? 1 : 0;
}

// A lane is good when its launches run beside the other lane's at (nearly) the rate they run alone: the launch period of the plan's OWN kernels with both lanes
// running within kLaneGoodRatio x the period of lane 0 alone (healthy plans measure 1.15-1.3 x, the slow states of profiles/r04_order_dependence.txt 2.3-6 x,
// lanes that share a hardware queue 2 x).  At run time a fixed-step run whose own launch period exceeds kLaneSlowRatio x the best period the plan has seen
// has its lanes rated again (PlanT::lane_health).
constexpr float kLaneGoodRatio = 1.6f;
constexpr float kLaneGoodRatioBig = 2.1f;
constexpr float kLaneSlowRatio = 1.8f;
constexpr float kLaneSuspectRatio = 1.3f;        // two checked runs in a row beyond this (round 6: profiles/r05_final_lane_stability.txt had one plan at 1.37 x that 1.8 never sees)
constexpr int kLaneDroppedCooldown = 32;         // runs after which a plan that dropped to one lane tries its second lane again
constexpr int kLaneProbeSteps = 12;
constexpr int kLaneHealsMax = 8;

// Rated lane pairs outlive their plan (round 6): rating a fresh pair costs ~7 ms of probe launches (k_queue_probe_spin was 25 % of an
// adaptive profile), and what the rating finds out -- which hardware queues the runtime gave the two streams -- is a property of the STREAMS, not of the plan.
// A two-lane plan that ends with good lanes hands its pair (the plan's stream and lane 1's) to a per-device pool instead of destroying it; the next two-lane
// plan of the process on that device takes a pooled pair with its rating and skips the probes.  The run-time check (lane_health) stays as it is.
struct LanePair { int device; hipStream_t main, lane; float score, alone_us, pair_us; };
struct LanePool {
    std::mutex m;
    std::vector<LanePair> free;
    static constexpr size_t kMax = 8;
    bool take(int device, LanePair* out) {
        std::lock_guard<std::mutex> lk(m);
        for (size_t i = free.size(); i-- > 0;)
            if (free[i].device == device) { *out = free[i]; free.erase(free.begin() + (long)i); return true; }
        return false;
    }
    bool give(const LanePair& p) {
        std::lock_guard<std::mutex> lk(m);
        if (free.size() >= kMax) return false;
        free.push_back(p);
        return true;
    }
};
LanePool& lane_pool() { static LanePool* p = new LanePool(); return *p; }        // (never destroyed: plans may be closed from atexit handlers)
std::atomic<long long> g_lane_ratings{0}, g_lane_pairs_reused{0};                  // process-wide: ratings made at plan creation / pairs taken from the pool

// ----------------------------------------------------------------------------- plan
struct PlanBase {
    int precision = 0;
    virtual ~PlanBase() {}
};

constexpr int kMaxLanesConst = 8;
template <typename T> struct PlanT : PlanBase {
    int device = 0;
    int64_t n = 0;
    int batch = 0;
    int N1 = 0, N2 = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    cx<T>* F = nullptr;        // batch * n
    cx<T>* Y = nullptr;        // the field between the kernels: == F (plain layout, in place) or a buffer of its own (U16 layout)
    T* P = nullptr;            // batch * n
    cx<T>* twN = nullptr;      // n
    cx<T>* twA = nullptr;      // (N1/16)*N2: W_N^m
    cx<T>* twB = nullptr;      // 16*N2: W_N^(m*N1/16)
    cx<T>* tw1 = nullptr;      // N1
    cx<T>* tw2 = nullptr;      // N2
    cx<T>* dnat = nullptr;     // n  (D~ or H, natural order, staging)
    cx<T>* dperm = nullptr;    // n  (D~ transposed order)
    cx<T>* dperm_fly = nullptr;  // the same for the rows of k_freq<FM_FLY> where they use another number of points per thread (Ef_fly)
    T* dimag_fly = nullptr;      // a fibre's operator (flat real part) for k_freq<FM_FLY_IM>: the imaginary parts alone (complex64 plans in the unit layout, lazily)
    bool dimag_valid = false;
    cx<T>* tw2_fly = nullptr;
    cx<T>* scratch = nullptr;  // batch * n, lazily
    cx<T>* xfer_tab[2] = {nullptr, nullptr};   // resident transfer functions of ssfm_transfer_table (n each, lazily)
    struct Tab { T h; cx<T>* ptr; bool valid; int kind; };      // kind 0: exp(D~ h)/N as complex numbers; 1: its phases (FM_PHASE)
    Tab tabs[kMaxTables] = {};
    // Re D~ is the same number at every frequency (a fibre: -alpha/2, reference devices.py:1145): |exp(D~ h)| is a scalar and the
    // table may hold phases only.  Decided when the operator is set; env SSFM_PHASE_TABLE=0 keeps the complex tables.
    bool op_flat_re = false;
    T op_re0 = 0;
    bool phase_tables = true;
    bool force_fly = false;    // env SSFM_FORCE_FLY=1 (diagnostic): fixed-step runs form exp(D~ h) in the kernel as adaptive runs do
    int tab_rr = 0;
    // single-launch engine of small plans (ssfm_kernels.hpp k_small): its own row twiddles and operator tables (another order)
    bool small = false;        // env SSFM_SMALL=0 turns it off
    cx<T>* tw_small = nullptr;
    cx<T>* tw_chirp = nullptr; // stage twiddles of the line for the chirp-z kernels where they take other points per thread than k_small (made on first use)
    cx<T>* dsmall = nullptr;   // D~ in the one-line order (k_small_adapt)
    // single-launch engine of medium plans (ssfm_kernels.hpp k_medium): its barrier counters and error word in device memory, a pinned
    // host copy of the error word, and what a repeat of the run on the two-kernel engine needs
    bool medium_ok = true;     // env SSFM_MEDIUM=0, or a run whose barrier once ran out of patience, clears it
    unsigned long long* medium_st = nullptr;       // kBarShards counters, kBarWords flag words, the error word, the ticket counter
    unsigned* medium_err_host = nullptr;        // two words: a dual-polarisation plan of long rows runs its rows as two launches (run_medium)
    hipStream_t medium_stream2 = nullptr;       // ... the second one here, on another XCD
    hipEvent_t medium_ev2 = nullptr;
    bool medium_pending = false;
    bool medium_split_ok = true;   // env SSFM_MEDIUM_SPLIT=0: a dual-polarisation plan's rows stay in one launch
    bool medium_adapt_ok = true;   // env SSFM_MEDIUM_ADAPT=0, or a run whose workgroups once did not all get to run, clears it
    static constexpr long long medium_max_samples = 1ll << 17;      // samples in all (rows x n) up to which the one-XCD engine is used (measured: a gain up to there)
    int medium_xcc = -1, medium_xccs = 8;          // the XCD this plan's single-launch runs use, of so many
    std::vector<T> medium_sched;
    double medium_gamma = 0;
    long long medium_patience = 2000000ll;
    bool fused_ok = true;      // TM_MID_A may be used (env SSFM_ADAPT_FUSED=0, or a grid that once did not run as a whole, clears it)
    cx<T>* fused_backup = nullptr;   // the input of a fused adaptive run, for the fall-back
    Tab stabs[kMaxTables] = {};
    int stab_rr = 0;
    T* d_hs = nullptr;         // the schedule on the device: step sizes, then one table index per step
    size_t d_hs_cap = 0;
    std::vector<unsigned char> h_sched;
    AdaptState<T>* st = nullptr;
    T* zlog = nullptr;
    int64_t zlog_cap = 0;
    bool have_op = false;
    // ---- split plans (ssfm_split.hpp; round 6): rows of more than 2^22 samples.  `n` / `batch` / N1 / N2 below describe the SUB-SEQUENCES (n = M samples,
    // batch = rows x R of them) that every kernel of this file works on; `n_full` / `batch_full` are what the caller sees.  The caller's field lives in
    // `Fnat` in natural time order; F holds its sub-sequences x_a[m] = x[a + R m] while runs are in flight (`sub_valid` / `nat_valid`: which of the two
    // is current -- k_split_shuffle converts lazily, so back-to-back runs do not pay it).
    int split_R = 1;
    int64_t n_full = 0;
    int batch_full = 0;
    cx<T>* Fnat = nullptr;
    double2* split_twA = nullptr;      // W_N^(a k1), W_N^(a N1 k2) in double (one rounding of their product in k_split_mid)
    double2* split_twB = nullptr;
    cx<T>* dsplit = nullptr;           // D~ in the split order [p][k1][k2] (SM_FLY: adaptive runs, more step sizes than tables), lazily
    bool dsplit_valid = false;
    bool sub_valid = true, nat_valid = false;
    bool is_split() const { return split_R > 1; }
    int split_qf() const { return u16 ? N2 / Ef : 0; }          // the order of a sub-row between the row-pass halves: the plan's unit layout, or plain
    T inv_n_full() const { return (T)1 / (T)n_full; }
    // the sub-sequences are current in F (before a run)
    int ensure_sub() {
        if (!is_split() || sub_valid) return SSFM_OK;
        HIP_TRY((launch_split_shuffle<T, true>(split_R, stream, F, Fnat, (long long)n, batch_full)));
        sub_valid = true;
        return SSFM_OK;
    }
    // the natural-order field is current in Fnat (before the caller looks at it)
    int ensure_nat() {
        if (!is_split() || nat_valid) return SSFM_OK;
        HIP_TRY((launch_split_shuffle<T, false>(split_R, stream, F, Fnat, (long long)n, batch_full)));
        nat_valid = true;
        return SSFM_OK;
    }
    void ran() { if (is_split()) { sub_valid = true; nat_valid = false; } }      // a run has advanced F
    // ... and a caller that holds the field's address (ssfm_field_device_ptr / ssfm_stream: `external_order`) finds the natural-order field behind the run on
    // the plan's stream -- and may WRITE it there before the next run, which therefore starts from Fnat again
    int publish_if_external() {
        if (!is_split() || !external_order) return SSFM_OK;
        if (int rc = ensure_nat()) return rc;
        sub_valid = false;
        return SSFM_OK;
    }
    cx<T>* user_field() { return is_split() ? Fnat : F; }
    // the time-order field after a step -> `dst` (DEVICE): a z-resolved capture's snapshot
    hipError_t put_time_order(void* dst) {
        if (is_split()) return launch_split_shuffle<T, false>(split_R, stream, F, static_cast<cx<T>*>(dst), (long long)n, batch_full);
        return hipMemcpyAsync(dst, F, sizeof(cx<T>) * (size_t)n * batch, hipMemcpyDeviceToDevice, stream);
    }
    // the row pass of a split plan: forward row transforms, the pointwise middle (twiddles, radix-R butterfly across the sub-sequences, operator, back),
    // inverse row transforms.  `table`: exp(D~ h) / N in the split order (SM_TABLE), or nullptr: formed in the launch from D~ (SM_FLY; `s` != nullptr:
    // the step size is the device's)
    hipError_t split_freq(const cx<T>* table, T h, const AdaptState<T>* s, int step, int row0, int rows, hipStream_t st_, bool phase = false, T amp = (T)0) {
        FreqArgs<T> fa = fargs(nullptr, 0, s, row0, rows > 0 ? row0 / rows : 0);         // (`s`: launches behind the end of an adaptive run find `done` and leave)
        fa.step = step;
        if (u16) fa.u16 = 2;                 // (the unit layout between the two halves as well: 16-byte accesses on both sides of k_split_mid)
        hipError_t e = launch_freq<T, FM_FWD_ONLY>(N2, N1 * rows, st_, fa, Ef);
        if (e != hipSuccess) return e;
        SplitArgs<T> sa;
        sa.Y = Y + (size_t)row0 * n; sa.G = table ? table : dsplit; sa.twA = split_twA; sa.twB = split_twB; sa.st = s; sa.h = h; sa.inv_n = inv_n_full();
        sa.step = step; sa.N1 = N1; sa.N2 = N2; sa.rows_outer = rows / split_R; sa.Qf = split_qf();
        sa.amp = amp;
        if constexpr (sizeof(T) == 4) { if (table && phase) e = launch_split_mid<T, SM_PHASE>(split_R, st_, sa); else e = table ? launch_split_mid<T, SM_TABLE>(split_R, st_, sa) : launch_split_mid<T, SM_FLY>(split_R, st_, sa); }
        else e = table ? launch_split_mid<T, SM_TABLE>(split_R, st_, sa) : launch_split_mid<T, SM_FLY>(split_R, st_, sa);
        if (e != hipSuccess) return e;
        return launch_freq<T, FM_INV_ONLY>(N2, N1 * rows, st_, fa, Ef);
    }
    // D~ in the split order, for SM_FLY
    int split_fly_ready() {
        if (dsplit_valid) return SSFM_OK;
        if (!dsplit) HIP_TRY(hipMalloc(&dsplit, sizeof(cx<T>) * (size_t)n_full));
        hipLaunchKernelGGL((k_make_split_table<T, 0>), dim3((unsigned)((n_full + 255) / 256)), dim3(256), 0, stream, (const cx<T>*)dnat, dsplit, N1, N2, split_R, (T)0, inv_n_full(), split_qf());
        HIP_TRY(hipGetLastError());
        dsplit_valid = true;
        return SSFM_OK;
    }
    int no_split(const char* what) { return is_split() ? fail(SSFM_ERR_UNSUPPORTED, "%s: not for plans of more than 2^%d samples per row", what, kLog2MaxDirect) : (int)SSFM_OK; }
    // What the staging buffers hold is the PLAN's knowledge, not the caller's: tags[0] labels the linear operator in
    // dnat / dperm, tags[1 + slot] the resident transfer function of a slot.  A caller that has staged something
    // labels it (ssfm_plan_set_tag) and asks later whether it is still there (ssfm_plan_get_tag); every entry point
    // that overwrites or reuses a buffer clears its label here, so a stale label cannot survive.  0 = nothing known.
    uint64_t tags[3] = {0, 0, 0};
    void drop_operator() { have_op = false; tags[0] = 0; dimag_valid = false; dsplit_valid = false; for (auto& t : tabs) t.valid = false; for (auto& t : stabs) t.valid = false; }
    bool timed = false;
    std::atomic<int64_t> last_launches{0};
    // what the last run really did (ssfm_last_run_info): a single-launch engine that falls back is otherwise invisible to the caller
    int last_engine = SSFM_ENGINE_NONE;
    int last_fell_back = 0;
    int64_t fallbacks = 0;
    bool lanes_share_queue = false;    // the plan's lanes could not be given hardware queues of their own (see streams_run_side_by_side)
    // ---- lane health (round 5; rate_lane / lane_health / heal_lanes below)
    int lanes_active = 1;              // lanes a fixed-step run really drives: nlanes, or 1 once a plan that could not be healed has dropped to one lane
    int lanes_remade = 0;              // lane streams replaced at RUN time (a run came out slow, the lanes were rated again and one was not good)
    int lanes_dropped = 0;             // 1: the plan gave up its second lane (two healing attempts in a row found no good stream)
    int lane_heals = 0, lane_strikes = 0, lane_suspect = 0, lane_dropped_runs = 0;
    bool lanes_from_pool = false;      // this plan's lane pair came rated from the process's pool (no probe launches at creation)
    int lane_fault = 0;                // test hook (ssfm_debug_lane_fault): 2 = every rating comes out "in the way"
    float lane_alone_us = 0.f;         // launch period of lane 0's chain alone on the chip (the plan's own kernels; last rating)
    float lane_pair_us = 0.f;          // best launch period seen with all lanes running: the last rating's, or a real run's
    float lane_last_us = 0.f;          // launch period of the last checked fixed-step run (device time / launches per lane)
    float lane_score = 0.f;            // last rating: period with the other lane / period alone
    bool lane_check_pending = false;   // the last run was a two-lane run of >= 64 steps whose period has not been looked at yet
    int64_t lane_check_launches = 0;
    hipEvent_t lane_e0 = nullptr, lane_e1 = nullptr;      // brackets of a rating measurement
    hipEvent_t run_e0 = nullptr, run_e1 = nullptr;        // brackets of the last run that lane_health looks at (its own pair: ev0 / ev1 are re-recorded by every entry point)
    // ---- z-resolved capture that does not stall the loop (propagate_fixed_capture, round 5): a stream of its own for the transfers, a ring of plan-owned
    // device blocks of snapshots (filled by the capture steps' END launches, sent to the host by a helper thread -- cap_worker -- as they fill), the scalar log.
    // Nothing on the device waits across streams: a stream blocked on another stream's event slows every other queue's launches (see propagate_fixed_capture).
    static constexpr int kCapBlocksMax = 8;
    hipStream_t cap_stream = nullptr;
    std::vector<hipEvent_t> cap_ev_ends[kMaxLanesConst];  // per lane and flush: the lane's END that filled the block
    hipEvent_t cap_ev_in = nullptr;                       // on the plan's stream, ahead of the run: the input is in F
    char* cap_blocks = nullptr;                           // kCapBlocksMax blocks at most, cap_block_bytes each
    char* cap_in = nullptr;                               // plans whose engine works in place (Y == F): a copy of the input, taken ahead of the run
    size_t cap_in_bytes = 0;
    size_t cap_block_bytes = 0;
    int cap_nblocks = 0;
    double* cap_scal = nullptr;                           // the wavefronts' pairs, then the reduced log
    size_t cap_scal_bytes = 0;
    LaneWorker cap_worker;                                // the helper thread of a capture run
    bool cap_pending = false;                             // a capture run's helper may still be at work: the next entry point (use_device) joins it
    // A caller that has asked for ssfm_stream() or ssfm_field_device_ptr() may order its own work behind a run without ssfm_synchronize(): for it a
    // run of the one-launch engine of medium plans is resolved (waited for, checked, repeated on the two-kernel engine if need be) before
    // ssfm_propagate_fixed returns
    bool external_order = false;
    // plan-owned device workspaces that grow on demand (ssfm_plan_workspace): what a driver loop outside this file (chirpz.hip) needs per call
    // without a hipMalloc / hipFree pair per call -- hipFree waits for the whole device and stalls the streams of every other plan
    void* work[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t work_cap[4] = {0, 0, 0, 0};
    int workspace(int slot, size_t bytes, void** out) {
        if (slot < 0 || slot > 3 || !out) return fail(SSFM_ERR_INVALID, "ssfm_plan_workspace: slot %d", slot);
        if (int rc = use_device()) return rc;
        if (work_cap[slot] < bytes) {
            HIP_TRY(hipStreamSynchronize(stream));
            (void)hipFree(work[slot]); work[slot] = nullptr; work_cap[slot] = 0;
            const size_t want = bytes + bytes / 2 + 256;
            HIP_TRY(hipMalloc(&work[slot], want));
            work_cap[slot] = want;
        }
        *out = work[slot];
        return SSFM_OK;
    }
    LaneWorker lane_worker[8];        // (lane g > 0: a host thread of its own for its launches; env SSFM_LANE_THREADS=0: all from the caller's thread)
    bool lane_threads = true;
    bool profiling = false;
    // "lanes": rows are independent (they only share the adaptive step size), so a fixed-step run
    // drives disjoint row groups on separate streams; kernels of different groups overlap, which
    // hides launch gaps and lets one group compute while another waits for memory.
    static constexpr int kMaxLanes = 8;
    int nlanes = 1;
    bool u16 = false;          // field layout between the kernels: 16-byte units (ssfm_kernels.hpp "U16")
    int E = 16;                // points per thread of k_time (env SSFM_E = 8 | 16)
    int Ef = 16;               // ... and of k_freq (env SSFM_EF); the two kernels only share the field layout
    int Ef_fly = 16;           // ... and of k_freq when it forms exp(D~ h) itself (adaptive runs, more step sizes than tables)
    hipStream_t lane_stream[kMaxLanes] = {};
    hipEvent_t lane_ev[kMaxLanes] = {};
    hipEvent_t fork_ev = nullptr;
    // per-lane pools of events (profiling).  mode 1: an event after every launch (per-class times, but
    // the marker packets slow a launch-dense run by ~30 %); mode 2: an event after every 64th launch
    // (negligible overhead; the interval is split between the classes by launch count); mode 3: after every
    // kSampleStride launches ONE launch of each class is bracketed by two events (3 extra markers per 16 launches: the
    // host keeps ahead of the GPU) -- kernel_times() then reports those single-launch intervals only.
    struct LaneProf {
        std::vector<hipEvent_t> ev;
        std::vector<int> c0, c1;          // launches of class 0 / 1 between event i-1 and event i
        size_t n = 0;
        int pend0 = 0, pend1 = 0;
        int sampling = 0;                 // mode 3: launches of the current sample group still to be bracketed
    };
    LaneProf prof[8];
    int prof_mode = 0;
    int prof_mode_run = 0;            // the mode the recorded events were taken in
    static constexpr int kSparseStride = 64;
    static constexpr int kSampleStride = 14;

    int prof_record(LaneProf& p, int lane) {
        if (p.n == p.ev.size()) {
            hipEvent_t e;
            HIP_TRY(hipEventCreate(&e));
            p.ev.push_back(e);
            p.c0.push_back(0);
            p.c1.push_back(0);
        }
        HIP_TRY(hipEventRecord(p.ev[p.n], lane_stream[lane]));
        p.c0[p.n] = p.pend0; p.c1[p.n] = p.pend1;
        p.pend0 = p.pend1 = 0;
        ++p.n;
        return SSFM_OK;
    }
    // cls < 0: opening event; cls 0/1: a launch of that class was just enqueued; cls 2: closing
    int prof_mark(int cls, int lane = 0) {
        if (!profiling) return SSFM_OK;
        LaneProf& p = prof[lane];
        if (cls < 0) { p.pend0 = p.pend1 = 0; p.sampling = 0; prof_mode_run = prof_mode; return prof_record(p, lane); }
        if (cls == 0) ++p.pend0;
        if (cls == 1) ++p.pend1;
        if (cls == 2) return (p.pend0 + p.pend1) ? prof_record(p, lane) : (int)SSFM_OK;
        if (prof_mode == 3) {
            if (p.sampling > 0) { --p.sampling; return prof_record(p, lane); }              // this interval holds exactly this launch
            if (p.pend0 + p.pend1 >= kSampleStride) { p.sampling = 2; return prof_record(p, lane); }
            return SSFM_OK;
        }
        if (prof_mode == 1 || p.pend0 + p.pend1 >= kSparseStride) return prof_record(p, lane);
        return SSFM_OK;
    }
    int kernel_times(int64_t counts[2], double total_ms[2]) {
        counts[0] = counts[1] = 0;
        total_ms[0] = total_ms[1] = 0.0;
        for (auto& p : prof) {
            if (p.n < 2) continue;
            HIP_TRY(hipEventSynchronize(p.ev[p.n - 1]));
            for (size_t i = 1; i < p.n; ++i) {
                const int tot = p.c0[i] + p.c1[i];
                if (tot == 0) continue;
                if (prof_mode_run == 3 && tot != 1) continue;          // sampled mode: the bracketed single launches only
                float ms = 0.f;
                HIP_TRY(hipEventElapsedTime(&ms, p.ev[i - 1], p.ev[i]));
                counts[0] += p.c0[i];
                counts[1] += p.c1[i];
                total_ms[0] += (double)ms * p.c0[i] / tot;
                total_ms[1] += (double)ms * p.c1[i] / tot;
            }
        }
        return SSFM_OK;
    }

    T inv_n() const { return (T)1 / (T)n; }

    int free_all() {
        cap_worker.stop();                                  // (finishes a capture run's transfers first)
        for (auto& w : lane_worker) w.stop();
        if (stream) (void)hipStreamSynchronize(stream);
        void* bufs[] = {F, Y != F ? Y : nullptr, P, twN, twA, twB, tw1, tw2, tw2_fly, dnat, dperm, dperm_fly, dimag_fly, scratch, st, zlog, xfer_tab[0], xfer_tab[1]};
        for (void* b : bufs) (void)hipFree(b);
        for (auto& t : tabs) (void)hipFree(t.ptr);
        for (auto& t : stabs) (void)hipFree(t.ptr);
        (void)hipFree(Fnat); (void)hipFree(split_twA); (void)hipFree(split_twB); (void)hipFree(dsplit);
        (void)hipFree(tw_small);
        (void)hipFree(tw_chirp);
        (void)hipFree(dsmall);
        (void)hipFree(fused_backup);
        (void)hipFree(medium_st);
        if (medium_err_host) (void)hipHostFree(medium_err_host);
        if (adapt_look) (void)hipHostFree(adapt_look);
        if (medium_ev2) (void)hipEventDestroy(medium_ev2);
        if (medium_stream2) (void)hipStreamDestroy(medium_stream2);
        (void)hipFree(d_hs);
        for (void* w : work) (void)hipFree(w);
        // a good pair of lane streams goes to the pool with its rating (lane_pool above) instead of being destroyed
        bool pooled = false;
        if (nlanes == 2 && lanes_ok && !lanes_dropped && !lanes_share_queue && lane_fault == 0 && stream && lane_stream[1] && lane_stream[1] != stream
            && !std::getenv("SSFM_LANE_POOL_OFF")) {
            (void)hipStreamSynchronize(lane_stream[1]);
            pooled = lane_pool().give(LanePair{device, stream, lane_stream[1], lane_score, lane_alone_us, lane_pair_us});
            if (pooled) lane_stream[1] = nullptr;
        }
        for (int g = 1; g < kMaxLanes; ++g) {
            if (lane_ev[g]) (void)hipEventDestroy(lane_ev[g]);
            if (lane_stream[g]) (void)hipStreamDestroy(lane_stream[g]);
        }
        if (fork_ev) (void)hipEventDestroy(fork_ev);
        if (lane_e0) (void)hipEventDestroy(lane_e0);
        if (lane_e1) (void)hipEventDestroy(lane_e1);
        if (run_e0) (void)hipEventDestroy(run_e0);
        if (run_e1) (void)hipEventDestroy(run_e1);
        if (cap_stream) (void)hipStreamSynchronize(cap_stream);
        for (auto& v : cap_ev_ends) for (hipEvent_t e : v) (void)hipEventDestroy(e);
        if (cap_ev_in) (void)hipEventDestroy(cap_ev_in);
        for (auto& e : acap_ev) if (e) (void)hipEventDestroy(e);
        (void)hipFree(cap_blocks); (void)hipFree(cap_in); (void)hipFree(cap_scal);
        if (cap_stream) (void)hipStreamDestroy(cap_stream);
        for (auto& p : prof) {
            for (hipEvent_t e : p.ev) (void)hipEventDestroy(e);
            p.ev.clear();
        }
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        if (stream && !pooled) (void)hipStreamDestroy(stream);
        return SSFM_OK;
    }
    ~PlanT() override { free_all(); }

    int init(int dev, int64_t n_, int batch_) {
        n_full = n_; batch_full = batch_;
        {
            int kf = 0;
            while ((1ll << kf) < n_) ++kf;
            int above = kLog2MaxDirect;
            if (const char* e = std::getenv("SSFM_SPLIT_ABOVE")) { const int v = std::atoi(e); if (v >= kSplitLog2M && v < kLog2MaxDirect) above = v; }      // (tests: split plans of 2^21 / 2^22 samples)
            if (kf > above) {
                // a split plan: R sub-sequences of M = 2^20 samples per row (the size the two-kernel engine runs best at; SSFM_SPLIT_LOG2M = 21 | 22: diagnostics)
                int lm = kSplitLog2M;
                if (const char* e = std::getenv("SSFM_SPLIT_LOG2M")) { const int v = std::atoi(e); if (v >= 20 && v <= kLog2MaxDirect) lm = v; }
                if (lm >= kf) lm = kf - 1;
                while ((1 << (kf - lm)) > kSplitMaxR) ++lm;
                split_R = 1 << (kf - lm);
                if ((long long)batch_ * split_R > 65535) return fail(SSFM_ERR_INVALID, "batch=%d rows of 2^%d samples: too many sub-sequences", batch_, kf);
                n_ = 1ll << lm;
                batch_ *= split_R;
                sub_valid = false; nat_valid = true;
            }
        }
        device = dev; n = n_; batch = batch_;
        int k = 0;
        while ((1ll << k) < n) ++k;
        // N = N1 * N2: N1 = 2^floor(k/2) up to 256, N2 up to 4096; beyond 2^20: N1 = 512, N2 up to 8192
        const int k1 = k <= 20 ? (k / 2 < 8 ? k / 2 : 8) : 9;
        N1 = 1 << k1;
        N2 = 1 << (k - k1);
        HIP_TRY(hipSetDevice(device));
        // All streams of a plan live in the HIGHEST priority class.  The lanes only overlap if their streams
        // sit on different hardware queues; the runtime multiplexes streams onto GPU_MAX_HW_QUEUES (default
        // 4) queues per priority class, and in a process that also holds RCCL's (normal-priority) streams
        // -- torch.distributed with backend nccl -- two normal-priority lanes landed on ONE queue: 35
        // instead of 21 us per step.  A priority class of their own gives the lanes queues of their own;
        // giving them the SAME priority keeps the arbitration between them fair (one high, one normal lane
        // starved the normal one: 26 instead of 16 us per field-step with 4 fields).
        int prio_lo = 0, prio_hi = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
        {
            // a plan that will drive two lanes takes a rated pair of streams from the process's pool where there is one (lane_pool above)
            int want2 = (long long)batch * n >= (1ll << 20) ? 2 : 1;
            if (const char* e = std::getenv("SSFM_LANES")) want2 = std::atoi(e);
            LanePair lp;
            if (want2 == 2 && batch_full >= 2 && batch_full % 2 == 0 && !std::getenv("SSFM_LANE_POOL_OFF") && lane_pool().take(device, &lp)) {
                stream = lp.main; lane_stream[1] = lp.lane;
                lane_score = lp.score; lane_alone_us = 0.f; lane_pair_us = 0.f;           // (the periods are this plan's own kernels': the first long run sets them)
                lanes_from_pool = true;
                ++g_lane_pairs_reused;
            }
        }
        if (!stream) HIP_TRY(hipStreamCreateWithPriority(&stream, hipStreamNonBlocking, prio_hi));
        HIP_TRY(hipEventCreate(&ev0));
        HIP_TRY(hipEventCreate(&ev1));
        // measured: complex64 with 8 points per thread (twice the waves, one more exchange) is 10 % slower;
        // complex128 is 5 % FASTER with 8 (16 need the whole register file: 1 wave per SIMD, AGPR spills)
        E = sizeof(T) == 8 ? 8 : 16;
        if (const char* e = std::getenv("SSFM_E")) E = std::atoi(e) == 16 ? 16 : 8;
        // k_freq: complex128 rows run best with 16 points per thread and load-at-use twiddles (256-thread workgroups,
        // two per CU; C1 46.8 -> 44.7 us per step), complex64 rows with 16 and register twiddles (8: 22.2 vs 21.4 us)
        Ef = 16;
        // up to 2^17 points a step is bound by launch and instruction latency, not by the exchanges: 8 points
        // per thread in both kernels (twice the wavefronts per workgroup) is 5-19 % faster there (tools/small_n.py:
        // 7.0 -> 5.7 us per step at 2^10..2^12, 8.6 -> 7.9 at 2^15, 11.3 -> 10.3 at 2^17; equal at 2^18, slower above)
        // (complex128 likewise in k_freq: 9.0 -> 7.7 us per step at 2^10, 13.2 -> 12.3 at 2^16, 17.1 -> 15.2 at 2^17 x 2)
        if (k <= 17 && !std::getenv("SSFM_E")) E = Ef = 8;        // (the environment knobs below still override)
        if (std::getenv("SSFM_E")) Ef = E;
        if (const char* e = std::getenv("SSFM_EF")) Ef = std::atoi(e) == 16 ? 16 : 8;
        if (k > 20) E = Ef = 16;        // the large tiles (N1 = 512, N2 = 8192) exist for 16 points per thread only
        // complex128 rows that form exp(D~ h) in the kernel: 16 points per thread need 256 VGPRs + 164 AGPRs (one wave per SIMD),
        // 8 need 248 (two): adaptive 2^20 x 2 81.3 -> 74.6 us per step (profiles/r02_adaptive_c128_ef.txt).  The operator table's
        // order follows the points per thread, so these rows get their own copy of D~ and of the row twiddles.
        Ef_fly = Ef;
        if (sizeof(T) == 8 && Ef == 16 && k <= 20 && !std::getenv("SSFM_E") && !std::getenv("SSFM_EF")) Ef_fly = 8;
        if (const char* e = std::getenv("SSFM_EF_FLY")) Ef_fly = k > 20 ? 16 : (std::atoi(e) == 16 ? 16 : 8);
        if (const char* e = std::getenv("SSFM_LANE_THREADS")) lane_threads = std::atoi(e) != 0;
        // two lanes pay off once a launch is long enough to hide the other lane's gap; below ~2^20 points in
        // all a step is launch-bound and the second stream only doubles the launches (2^14 x 2: 8.1 us per
        // step with one lane, 12.0 with two; 2^18 x 2: 14.1 / 14.2; 2^20 x 2: 25.6 / 22.9 in a 200-step run)
        int want = (long long)batch * n >= (1ll << 20) ? 2 : 1;
        if (const char* e = std::getenv("SSFM_LANES")) want = std::atoi(e);
        nlanes = want < 1 ? 1 : (want > kMaxLanes ? kMaxLanes : want);
        if (nlanes > batch_full) nlanes = batch_full;
        while (batch_full % nlanes) --nlanes;            // (a lane holds whole rows: all sub-sequences of a row of a split plan)
        lane_stream[0] = stream;
        HIP_TRY(hipEventCreateWithFlags(&fork_ev, hipEventDisableTiming));
        if (lanes_from_pool && nlanes != 2) {           // (cannot happen with the rule above; a pooled lane stream must not leak)
            (void)hipStreamDestroy(lane_stream[1]); lane_stream[1] = nullptr; lanes_from_pool = false;
        }
        for (int g = 1; g < nlanes; ++g) {
            if (!(lanes_from_pool && g == 1)) HIP_TRY(hipStreamCreateWithPriority(&lane_stream[g], hipStreamNonBlocking, prio_hi));
            HIP_TRY(hipEventCreateWithFlags(&lane_ev[g], hipEventDisableTiming));
        }
        lanes_active = nlanes;
        const size_t cb = sizeof(cx<T>);
        HIP_TRY(hipMalloc(&F, cb * n * batch));
        HIP_TRY(hipMalloc(&P, sizeof(T) * n * batch));
        u16 = u16_layout<T>(N1, cols_per_tile<T>(), E);
        Y = F;
        if (u16) HIP_TRY(hipMalloc(&Y, cb * n * batch));
        // inter-pass twiddles W_N^(k1 n2): either the n-entry table in k_time's thread order, or (U16 plans) the two
        // small factor tables the kernel multiplies (ssfm_kernels.hpp twn_compute)
        if (u16 || sizeof(T) == 8) {
            const long long nA = (long long)(N1 / E) * N2, nB = (long long)E * N2;      // [tile][j][c], j < N1/E;  [tile][t][c], t < E
            HIP_TRY(hipMalloc(&twA, cb * nA));
            HIP_TRY(hipMalloc(&twB, cb * nB));
            const int C_ = cols_per_tile<T>();
            hipLaunchKernelGGL(k_make_tw_tiles<T>, dim3((unsigned)((nA + 255) / 256)), dim3(256), 0, stream, twA, N1 / E, 1ll, N2, C_, (long long)n, (int)u16, N2 / Ef);
            hipLaunchKernelGGL(k_make_tw_tiles<T>, dim3((unsigned)((nB + 255) / 256)), dim3(256), 0, stream, twB, E, (long long)(N1 / E), N2, C_, (long long)n, (int)u16, N2 / Ef);
        } else {
            HIP_TRY(hipMalloc(&twN, cb * n));
            hipLaunchKernelGGL(k_make_twN<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, twN, N1, N2, cols_per_tile<T>(), E, (int)u16, N2 / Ef);
        }
        if (int rc = make_line_table(&tw1, N1, E)) return rc;
        if (int rc = make_line_table(&tw2, N2, Ef)) return rc;
        HIP_TRY(hipMalloc(&dnat, cb * n_full));
        HIP_TRY(hipMalloc(&dperm, cb * n));
        if (is_split()) {
            HIP_TRY(hipMalloc(&Fnat, cb * n_full * batch_full));
            HIP_TRY(hipMemsetAsync(Fnat, 0, cb * n_full * batch_full, stream));
            HIP_TRY(hipMalloc(&split_twA, sizeof(double2) * (size_t)split_R * N1));
            HIP_TRY(hipMalloc(&split_twB, sizeof(double2) * (size_t)split_R * N2));
            const long long nt = (long long)split_R * (N1 + N2);
            hipLaunchKernelGGL(k_make_split_twiddles, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, stream, split_twA, split_twB, N1, N2, split_R);
            HIP_TRY(hipGetLastError());
        }
        if (Ef_fly != Ef) {
            if (int rc = make_line_table(&tw2_fly, N2, Ef_fly)) return rc;
            HIP_TRY(hipMalloc(&dperm_fly, cb * n));
        }
        if (const char* e = std::getenv("SSFM_ADAPT_FUSED")) fused_ok = std::atoi(e) != 0;
        if (const char* e = std::getenv("SSFM_MEDIUM")) medium_ok = std::atoi(e) != 0;
        if (const char* e = std::getenv("SSFM_MEDIUM_ADAPT")) medium_adapt_ok = std::atoi(e) != 0;
        if (const char* e = std::getenv("SSFM_MEDIUM_SPLIT")) medium_split_ok = std::atoi(e) != 0;
        if (const char* e = std::getenv("SSFM_FUSED_PATIENCE_TICKS")) medium_patience = std::atoll(e);
        if (const char* e = std::getenv("SSFM_PHASE_TABLE")) phase_tables = std::atoi(e) != 0;
        if (const char* e = std::getenv("SSFM_FORCE_FLY")) force_fly = std::atoi(e) != 0;
        // the XCD this plan's one-launch runs meet on: decided now (the probe launch behind xcc_mask runs once per device, on a stream of its own)
        if (sizeof(T) == 4 && medium_shape(N1, N2) && (medium_ok || medium_adapt_ok)) (void)pick_xcc();
        small = small_supported<T>((int)n);
        if (const char* e = std::getenv("SSFM_SMALL")) small = small && std::atoi(e) != 0;
        if (small) {
            if (int rc = make_line_table(&tw_small, (int)n, small_points<T>((int)n))) return rc;
            HIP_TRY(hipMalloc(&dsmall, cb * n));
        }
        HIP_TRY(hipMalloc(&st, sizeof(AdaptState<T>)));
        HIP_TRY(hipMemsetAsync(P, 0, sizeof(T) * n * batch, stream));
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(stream));
        if (nlanes > 1 && nlanes <= 4) {
            // every lane on a hardware queue of its own, and one on which its kernels really run beside lane 0's (rate_lane): rated with the plan's own
            // kernels on its own buffers, which hold nothing yet -- zeros (a transform of zeros is zeros; D~ = 0 is the identity)
            HIP_TRY(hipEventCreate(&lane_e0));
            HIP_TRY(hipEventCreate(&lane_e1));
            HIP_TRY(hipEventCreate(&run_e0));
            HIP_TRY(hipEventCreate(&run_e1));
            HIP_TRY(hipMemsetAsync(F, 0, cb * n * batch, stream));
            if (Y != F) HIP_TRY(hipMemsetAsync(Y, 0, cb * n * batch, stream));
            HIP_TRY(hipMemsetAsync(dperm, 0, cb * n, stream));
            if (dperm_fly) HIP_TRY(hipMemsetAsync(dperm_fly, 0, cb * n, stream));
            bool replaced = false;
            if (lanes_from_pool && nlanes == 2) lanes_ok = true;               // (rated by the plan that made the pair; lane_health watches the runs as for any plan)
            else {
                ++g_lane_ratings;
                if (int rc = rate_and_replace_lanes(F, Y, P, &replaced)) return rc;
            }
            if (!lanes_ok) lane_strikes = 1;          // (no good stream to be had now: the first run that comes out slow and cannot be healed drops the second lane)
            HIP_TRY(hipStreamSynchronize(stream));
        }
        return SSFM_OK;
    }

    // ---- lane health.  Which hardware queue the runtime gives a lane's stream decides whether its kernels run beside the other lane's at full rate, one
    // after the other (a shared queue) or 3-6 x slower than that (profiles/r04_order_dependence.txt); nothing in the API says which.  Round 4 rated a
    // stream at plan creation by a proxy (dependent launches of a 3 us spin kernel) and still met about one slow plan in 25-40 suite runs, silently.
    // Round 5: (1) the rating is the real thing -- the launch period of the plan's OWN two kernels with both lanes running against lane 0 alone;
    // (2) every fixed-step run of >= 64 steps is looked at afterwards (lane_health: its device time / launches per lane against the best period the
    // plan has seen); a run beyond kLaneSlowRatio has the lanes rated again on scratch buffers, a lane that is not good gets another stream
    // (`lanes_remade`), and a plan that two healing attempts in a row could not repair drops to one lane (`lanes_dropped`: both rows in one launch on
    // the plan's stream -- slower than two good lanes, 2-4 x faster than two bad ones).  ssfm_last_run_info reports all of it.
    //
    // Launch period [us] of `steps` steps of the two-kernel engine (BEGIN, k_freq<FLY>, MID ... END) on the lanes of `mask`, on the buffers given (zeros).
    // The lanes' queues are held by a spin kernel while the host enqueues, so that the host's enqueue rate is not part of the figure.
    int measure_lanes(unsigned mask, int steps, cx<T>* Fx, cx<T>* Yx, T* Px, float* us) {
        cx<T>* F0 = F; cx<T>* Y0 = Y; T* P0 = P;
        F = Fx; Y = Yx; P = Px;
        const int rows = batch / nlanes;
        const T hh = (T)0.0625;
        auto body = [&]() -> int {
            hipLaunchKernelGGL(k_queue_probe_spin, dim3(1), dim3(1), 0, stream, (long long)(400 * (2 * steps + 1) * nlanes));       // (100 MHz ticks: 4 us per launch to come -- the host needs ~3)
            HIP_TRY(hipEventRecord(fork_ev, stream));
            HIP_TRY(hipEventRecord(lane_e0, stream));
            for (int g = 1; g < nlanes; ++g)
                if (mask & (1u << g)) HIP_TRY(hipStreamWaitEvent(lane_stream[g], fork_ev, 0));
            for (int g = 0; g < nlanes; ++g)
                if (mask & (1u << g)) HIP_TRY((launch_time<T, TM_BEGIN>(N1, rows, lane_stream[g], targs((T)1, 0, hh, nullptr, g * rows, g), E)));
            for (int s_ = 0; s_ < steps; ++s_) {
                for (int g = 0; g < nlanes; ++g)
                    if (mask & (1u << g)) HIP_TRY((launch_freq<T, FM_FLY>(N2, N1 * rows, lane_stream[g], fargs_fly(2 * hh, nullptr, g * rows, g), Ef_fly)));
                for (int g = 0; g < nlanes; ++g) {
                    if (!(mask & (1u << g))) continue;
                    if (s_ + 1 < steps) HIP_TRY((launch_time<T, TM_MID>(N1, rows, lane_stream[g], targs((T)1, hh, hh, nullptr, g * rows, g), E)));
                    else HIP_TRY((launch_time<T, TM_END>(N1, rows, lane_stream[g], targs((T)1, hh, 0, nullptr, g * rows, g), E)));
                }
            }
            for (int g = 1; g < nlanes; ++g) {
                if (!(mask & (1u << g))) continue;
                HIP_TRY(hipEventRecord(lane_ev[g], lane_stream[g]));
                HIP_TRY(hipStreamWaitEvent(stream, lane_ev[g], 0));
            }
            HIP_TRY(hipEventRecord(lane_e1, stream));
            HIP_TRY(hipStreamSynchronize(stream));
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, lane_e0, lane_e1));
            *us = ms * 1e3f / (float)(2 * steps + 1);
            return SSFM_OK;
        };
        const int rc = body();
        F = F0; Y = Y0; P = P0;
        return rc;
    }
    // 0 = good, 1 = on a queue of its own but in lane 0's way (score = the ratio), 2 = shares a hardware queue with an earlier lane; < 0: error (g_err set)
    int rate_lane(int g, cx<T>* Fx, cx<T>* Yx, T* Px, float* score, float* alone_out, float* pair_out) {
        *score = 1e30f;
        unsigned* words = nullptr;
        if (hipMalloc(&words, 2 * sizeof(unsigned)) != hipSuccess) { (void)hipGetLastError(); return -1; }
        int shared = 0;
        for (int k = 0; k < g && !shared; ++k) {
            const int side = streams_run_side_by_side(lane_stream[k], lane_stream[g], words, fork_ev);
            if (side < 0) { (void)hipFree(words); return -1; }
            shared = side == 0;
        }
        (void)hipFree(words);
        if (shared) return 2;
        float alone = 1e30f, pair = 0.f;
        // (rows of 32 MiB and more -- complex128 plans of 2^21 points up, the lines of long chirp-z runs: a launch is 30-100 us long, a third of the probe steps measure it as well)
        const int probe_steps = big_rows() ? kLaneProbeSteps / 3 : kLaneProbeSteps;
        for (int r = 0; r < 2; ++r) {           // (the slow state is erratic -- one of its readings may look good: the WORSE of two, against the better alone)
            float a1 = 0.f, p1 = 0.f;
            if (measure_lanes(1u, probe_steps, Fx, Yx, Px, &a1) != SSFM_OK) return -1;
            if (measure_lanes(1u | (1u << g), probe_steps, Fx, Yx, Px, &p1) != SSFM_OK) return -1;
            alone = a1 < alone ? a1 : alone;
            pair = p1 > pair ? p1 : pair;
        }
        *score = pair / alone;
        *alone_out = alone; *pair_out = pair;
#ifdef SSFM_LANE_PROBE_DEBUG
        std::fprintf(stderr, "lane rating: lane %d: %.2f us per launch alone, %.2f with lane 0 beside it\n", g, alone, pair);
#endif
        if (lane_fault == 2) { *score = 9.f; return 1; }
        // Round 6: a kernel that moves 32 MiB and more per row keeps the memory system busy by itself, and its twin beside it legitimately takes up to twice as long per launch
        // (2^22-point complex128 rows: 1.80-1.82, and two lanes are still 18 % faster per step than one launch for both rows) -- rated against 1.6 such a pair was "in the way",
        // had four replacement streams tried in vain, was never pooled, and every such plan took 117 ms to make (tools/plan_create_time.py).  The slow states measure 2.3-6 x.
        return *score > (big_rows() ? kLaneGoodRatioBig : kLaneGoodRatio) ? 1 : 0;
    }
    bool big_rows() const { return (size_t)n * sizeof(cx<T>) >= ((size_t)32 << 20); }
    // Rate every lane beyond the first; a lane that is not good tries replacement streams four at a time -- made together, so that the runtime's
    // least-used rule spreads them over the class's hardware queues -- the first good one (else the best one) replaces the lane's stream, the rest go.
    int rate_and_replace_lanes(cx<T>* Fx, cx<T>* Yx, T* Px, bool* replaced) {
        int prio_lo = 0, prio_hi = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
        *replaced = false;
        bool all_good = true;
        for (int g = 1; g < nlanes; ++g) {
            float score = 0.f, alone = 0.f, pair = 0.f;
            int state = rate_lane(g, Fx, Yx, Px, &score, &alone, &pair);
            if (state < 0) return fail(SSFM_ERR_HIP, "lane rating failed: %s", hipGetErrorString(hipGetLastError()));
            for (int round = 0; round < 2 && state > 0; ++round) {
                hipStream_t cand[4] = {nullptr, nullptr, nullptr, nullptr};
                for (auto& c : cand)
                    if (hipStreamCreateWithPriority(&c, hipStreamNonBlocking, prio_hi) != hipSuccess) { (void)hipGetLastError(); c = nullptr; }
                for (auto& c : cand) {
                    if (!c || state == 0) continue;
                    float sc = 0.f, al = 0.f, pr = 0.f;
                    std::swap(c, lane_stream[g]);                       // (rated in the lane's place; swapped back unless it is better)
                    const int st_c = rate_lane(g, Fx, Yx, Px, &sc, &al, &pr);
                    if (st_c >= 0 && (st_c < state || (st_c == state && sc < score))) { state = st_c; score = sc; alone = al; pair = pr; *replaced = true; }
                    else std::swap(c, lane_stream[g]);
                }
                for (auto& c : cand)
                    if (c) (void)hipStreamDestroy(c);
            }
            if (state != 0) all_good = false;
            if (state == 2) lanes_share_queue = true;
            if (g == 1 || score > lane_score) lane_score = score;
            if (state != 2) { lane_alone_us = alone; if (state == 0) lane_pair_us = pair; }
        }
        if (all_good) lanes_share_queue = false;
        lanes_ok = all_good;
        return SSFM_OK;
    }
    bool lanes_ok = true;
    // After a two-lane run of >= 64 steps: was it as fast as this plan's lanes can be?  Called before the next run is enqueued and from ssfm_synchronize /
    // ssfm_last_propagate_ms / ssfm_last_run_info (where the run has ended anyway); waits for the run's closing event.
    int lane_health() {
        if (lanes_dropped && nlanes == 2 && ++lane_dropped_runs >= kLaneDroppedCooldown && lane_heals < kLaneHealsMax) {
            // a plan that gave up its second lane under contention that may have passed: one more rating on scratch fields; good lanes come back
            lane_dropped_runs = 0;
            const int before = lane_strikes;
            if (int rc = heal_lanes()) return rc;
            if (lanes_ok) { lanes_active = nlanes; lanes_dropped = 0; lane_strikes = 0; }
            else lane_strikes = before;
        }
        if (!lane_check_pending) return SSFM_OK;
        lane_check_pending = false;
        if (lanes_active < 2 || lane_check_launches <= 0 || !run_e1) return SSFM_OK;
        HIP_TRY(hipEventSynchronize(run_e1));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, run_e0, run_e1));
        lane_last_us = ms * 1e3f / (float)lane_check_launches;
        if (lane_pair_us <= 0.f || lane_last_us < lane_pair_us) { lane_pair_us = lane_last_us; lane_suspect = 0; return SSFM_OK; }
        if (lane_heals >= kLaneHealsMax) return SSFM_OK;
        if (lane_last_us > kLaneSlowRatio * lane_pair_us) { lane_suspect = 0; return heal_lanes(); }
        // a second look for the moderately slow: one such run is other work on the GPU, two in a row are rated
        if (lane_last_us > kLaneSuspectRatio * lane_pair_us) { if (++lane_suspect >= 2) { lane_suspect = 0; return heal_lanes(); } }
        else lane_suspect = 0;
        return SSFM_OK;
    }
    int heal_lanes() {
        ++lane_heals;
        HIP_TRY(hipStreamSynchronize(stream));
        const size_t cb = sizeof(cx<T>) * (size_t)n * batch;
        cx<T>* Fx = nullptr; cx<T>* Yx = nullptr; T* Px = nullptr;
        auto drop = [&]() { (void)hipFree(Fx); if (Yx != Fx) (void)hipFree(Yx); (void)hipFree(Px); };
        if (hipMalloc(&Fx, cb) != hipSuccess) { (void)hipGetLastError(); return SSFM_OK; }       // (no memory for the scratch fields: the run's result is right, only slow)
        Yx = Fx;
        if (Y != F && hipMalloc(&Yx, cb) != hipSuccess) { (void)hipGetLastError(); Yx = Fx; drop(); return SSFM_OK; }
        if (hipMalloc(&Px, sizeof(T) * (size_t)n * batch) != hipSuccess) { (void)hipGetLastError(); drop(); return SSFM_OK; }
        (void)hipMemsetAsync(Fx, 0, cb, stream);
        if (Yx != Fx) (void)hipMemsetAsync(Yx, 0, cb, stream);
        (void)hipMemsetAsync(Px, 0, sizeof(T) * (size_t)n * batch, stream);
        const float seen = lane_pair_us;
        bool replaced = false;
        const int rc = rate_and_replace_lanes(Fx, Yx, Px, &replaced);
        (void)hipStreamSynchronize(stream);
        drop();
        if (rc != SSFM_OK) return rc;
        if (replaced) ++lanes_remade;
        if (lanes_ok) {
            lane_strikes = 0;
            // nothing wrong with the lanes (any more).  Not replaced: the run was slow for another reason (other work on the GPU, a profiler) -- its period is
            // the new normal until a faster run is seen, so that every further run at this level is not rated again
            if (!replaced) lane_pair_us = lane_last_us;
            else if (seen > 0.f && seen < lane_pair_us) lane_pair_us = seen;
        } else if (++lane_strikes >= 2) {
            lanes_active = 1;
            lanes_dropped = 1;
        }
        return SSFM_OK;
    }

    // stage twiddles of a line of length L in thread-load order (wgfft.hpp "Table layout"), computed
    // in double on the host and rounded once
    int make_line_table(cx<T>** out, int L, int E) {
        const int M = fft_nstages(L, E);
        const int total = fft_tw_entries(L, E, (int)sizeof(T));
        std::vector<cx<T>> tab((size_t)(total > 0 ? total : 1));
        for (int S = 1; S < M; ++S) {
            const int R = fft_radix(L, S, E), NB = E / R, KU = fft_tw_ku(L, S, E), off = fft_tw_offset(L, S, E, (int)sizeof(T));
            for (int i = 0; i < NB; ++i)
                for (int u = 1; u < R; ++u)
                    for (int ku = 0; ku < KU; ++ku) {
                        const int q = fft_tw_exponent(L, E, S, i, u, ku) % L;
                        const double ang = -2.0 * 3.14159265358979323846 * (double)q / (double)L;
                        cx<T> w; w.x = (T)std::cos(ang); w.y = (T)std::sin(ang);
                        if (q == 0) { w.x = (T)1; w.y = (T)0; }
                        if (4 * q == L) { w.x = (T)0; w.y = (T)-1; }
                        if (2 * q == L) { w.x = (T)-1; w.y = (T)0; }
                        if (4 * q == 3 * L) { w.x = (T)0; w.y = (T)1; }
                        tab[(size_t)off + (size_t)fft_tw_index(L, S, E, (int)sizeof(T), i * (R - 1) + (u - 1), ku)] = w;
                    }
        }
        HIP_TRY(hipMalloc(out, sizeof(cx<T>) * tab.size()));
        HIP_TRY(hipMemcpy(*out, tab.data(), sizeof(cx<T>) * tab.size(), hipMemcpyHostToDevice));
        return SSFM_OK;
    }

#if SSFM_TRACE
    unsigned long long* trace_buf = nullptr;
    int trace_cap = 0; std::atomic<int> trace_next{0};
    std::vector<int> trace_kind;        // per slot: kind*16 + lane
    int trace_begin(int launches) {
        if (!std::getenv("SSFM_TRACE_FILE")) return SSFM_OK;
        if (trace_cap < launches) {
            (void)hipFree(trace_buf);
            HIP_TRY(hipMalloc(&trace_buf, sizeof(unsigned long long) * 1024 * launches));
            trace_cap = launches;
        }
        HIP_TRY(hipMemset(trace_buf, 0, sizeof(unsigned long long) * 1024 * launches));
        trace_next = 0;
        trace_kind.assign(launches, 0);
        return SSFM_OK;
    }
    template <typename A> void trace_tag(A& a, int kind, int lane) {
        a.trace = nullptr; a.trace_slot = 0;
        if (trace_buf) { const int slot = trace_next.fetch_add(1); if (slot < trace_cap) { a.trace = trace_buf; a.trace_slot = slot; trace_kind[slot] = kind * 16 + lane; } }     // (the lanes' host threads tag concurrently)
    }
    int trace_dump() {
        const char* path = std::getenv("SSFM_TRACE_FILE");
        if (!path || !trace_buf) return SSFM_OK;
        HIP_TRY(hipStreamSynchronize(stream));
        const int nt = std::min((int)trace_next, trace_cap);
        std::vector<unsigned long long> raw(1024 * (size_t)nt), h(4 * (size_t)nt);
        HIP_TRY(hipMemcpy(raw.data(), trace_buf, sizeof(unsigned long long) * raw.size(), hipMemcpyDeviceToHost));
        for (int i = 0; i < nt; ++i) {
            unsigned long long fs = ~0ull, ls = 0, fe = ~0ull, le = 0;
            for (int b = 0; b < 256; ++b) {
                const unsigned long long s0 = raw[((size_t)i * 256 + b) * 4], e0 = raw[((size_t)i * 256 + b) * 4 + 1];
                if (s0 == 0) continue;
                fs = s0 < fs ? s0 : fs; ls = s0 > ls ? s0 : ls; fe = e0 < fe ? e0 : fe; le = e0 > le ? e0 : le;
            }
            h[4 * i] = fs; h[4 * i + 1] = ls; h[4 * i + 2] = fe; h[4 * i + 3] = le;
        }
        FILE* f = std::fopen(path, "w");
        if (!f) return SSFM_OK;
        std::fprintf(f, "slot,kind,lane,first_start,last_start,first_end,last_end\n");
        for (int i = 0; i < nt; ++i)
            std::fprintf(f, "%d,%d,%d,%llu,%llu,%llu,%llu\n", i, trace_kind[i] / 16, trace_kind[i] % 16, h[4 * i], h[4 * i + 1], h[4 * i + 2], h[4 * i + 3]);
        std::fclose(f);
        // SSFM_TRACE_RAW=path: every workgroup of launches [first, first + count) of the run (SSFM_TRACE_RAW_FIRST, default the middle; 16 launches)
        if (const char* rp = std::getenv("SSFM_TRACE_RAW")) {
            if (FILE* g = std::fopen(rp, "w")) {
                int first = nt / 2;
                if (const char* e = std::getenv("SSFM_TRACE_RAW_FIRST")) first = std::atoi(e);
                std::fprintf(g, "slot,kind,lane,block,start,end,xcc,hw_id\n");
                for (int i = first; i < first + 16 && i < nt; ++i)
                    for (int b = 0; b < 256; ++b) {
                        const unsigned long long* w = &raw[((size_t)i * 256 + b) * 4];
                        std::fprintf(g, "%d,%d,%d,%d,%llu,%llu,%llu,%llu\n", i, trace_kind[i] / 16, trace_kind[i] % 16, b, w[0], w[1], w[2] >> 32, w[2] & 0xffffffffull);
                    }
                std::fclose(g);
            }
        }
        return SSFM_OK;
    }
#endif

    int use_device() {
        HIP_TRY(hipSetDevice(device));
        if (cap_pending) {                 // a capture run's helper thread: its transfers are complete when it returns
            cap_pending = false;
            if (int rc = cap_worker.wait()) return rc;
        }
        return medium_pending ? finish_medium() : (int)SSFM_OK;
    }
    // A single-launch run of a medium plan is asynchronous like every run; whether one of its barriers ran out of patience is known
    // once it has finished.  The next call that uses the plan looks: on an error the input is restored and the run repeated on the
    // two-kernel engine (and the plan keeps to it).
    int finish_medium() {
        medium_pending = false;
        HIP_TRY(hipStreamSynchronize(stream));
        if (medium_err_host[0] == 0u && medium_err_host[1] == 0u) return SSFM_OK;
        medium_err_host[0] = medium_err_host[1] = 0u;
        medium_ok = false;
        ++fallbacks;
        HIP_TRY(hipMemcpyAsync(F, fused_backup, sizeof(cx<T>) * n * batch, hipMemcpyDeviceToDevice, stream));
        const std::vector<T> sched = medium_sched;
        if (int rc = propagate_fixed(medium_gamma, sched.data(), (int64_t)sched.size(), nullptr)) return rc;
        HIP_TRY(hipStreamSynchronize(stream));
        last_fell_back = 1;
        return SSFM_OK;
    }
    // the XCDs a launch's workgroups are dealt to (probe launch, once per device), and this plan's among them: plans take turns
    int pick_xcc() {
        if (medium_xcc >= 0) return SSFM_OK;
        static std::atomic<int> next_plan{0};
        const unsigned mask = xcc_mask(device);
        if (mask == 0u) { medium_ok = medium_adapt_ok = false; return SSFM_OK; }        // (the probe launch failed: the launch-per-pass engines)
        medium_xccs = __builtin_popcount(mask);
        int k = next_plan.fetch_add(1) % medium_xccs;
        for (int b = 0; b < 32; ++b)
            if ((mask >> b) & 1u) { if (k == 0) { medium_xcc = b; break; } --k; }
        return SSFM_OK;
    }
    // fixed-step runs of a dual-polarisation plan of long rows: one launch per row (see run_medium)
    bool medium_rows_split() const { return medium_split_ok && batch == 2 && n >= (1ll << 16); }
    // the whole schedule in one launch (k_medium); `distinct` holds at most kMaxTables step sizes
    int run_medium(T gamma, double gamma_d, const T* h, int64_t nsteps, const std::vector<T>& distinct, bool phase) {
        if constexpr (sizeof(T) != 4) { (void)gamma; (void)gamma_d; (void)h; (void)nsteps; (void)distinct; (void)phase; return fail(SSFM_ERR_STATE, "the medium engine is complex64 only"); }
        else {
        MediumArgs<T> a;
        std::memset(&a, 0, sizeof(a));
        if (int rc = tables_for(distinct, a.tab, false, phase ? 1 : 0)) return rc;
        for (size_t i = 0; i < distinct.size(); ++i) { const T xr = op_re0 * distinct[i]; a.amp[i] = (T)std::exp((double)xr) * inv_n(); }
        const size_t hb = sizeof(T) * (size_t)nsteps, need = hb + (size_t)nsteps;
        if (d_hs_cap < need) {
            (void)hipFree(d_hs); d_hs = nullptr; d_hs_cap = 0;
            HIP_TRY(hipMalloc(&d_hs, need + need / 2));
            d_hs_cap = need + need / 2;
        }
        HIP_TRY(hipStreamSynchronize(stream));             // the staging vector may still feed the previous run's copy
        h_sched.resize(need);
        std::memcpy(h_sched.data(), h, hb);
        for (int64_t s = 0; s < nsteps; ++s) {
            unsigned char w = 0;
            for (size_t i = 0; i < distinct.size(); ++i)
                if (std::memcmp(&distinct[i], &h[s], sizeof(T)) == 0) w = (unsigned char)i;
            h_sched[hb + (size_t)s] = w;
        }
        HIP_TRY(hipMemcpyAsync(d_hs, h_sched.data(), need, hipMemcpyHostToDevice, stream));
        const size_t fb = sizeof(cx<T>) * n * batch;
        if (!fused_backup) HIP_TRY(hipMalloc(&fused_backup, fb));
        HIP_TRY(hipMemcpyAsync(fused_backup, F, fb, hipMemcpyDeviceToDevice, stream));          // for a repeat on the two-kernel engine
        if (!medium_st) {
            HIP_TRY(hipMalloc(&medium_st, sizeof(unsigned long long) * 2 * (kBarShards + kBarWords + 2)));
            HIP_TRY(hipHostMalloc(&medium_err_host, 2 * sizeof(unsigned)));
            medium_err_host[0] = medium_err_host[1] = 0u;
        }
        constexpr size_t kSet = kBarShards + kBarWords + 2;
        HIP_TRY(hipMemsetAsync(medium_st, 0, sizeof(unsigned long long) * 2 * kSet, stream));
        a.F = F; a.Y = Y; a.P = P; a.twA = twA; a.twB = twB; a.tw1 = tw1; a.tw2 = tw2;
        a.hs = d_hs; a.which = reinterpret_cast<const unsigned char*>(d_hs) + hb;
        a.bar = medium_st; a.error = reinterpret_cast<unsigned*>(medium_st + kBarShards + kBarWords); a.patience = medium_patience;
        if (int rc = pick_xcc()) return rc;
        a.xcc = (unsigned)medium_xcc;
        a.gamma = gamma; a.inv_n = inv_n(); a.nsteps = (int)nsteps; a.Qf = N2 / Ef;
        // The rows of a fixed-step run share nothing: a dual-polarisation plan of long rows (2^16 samples and more) runs them as TWO launches, each
        // on an XCD and a stream of its own (2^16 x 2: 8.3 against 9.25 us per step in one launch; 2^17 x 2 only fits this way)
        const bool two = medium_rows_split();
        a.rows = two ? 1 : batch;
        a.nblk = (unsigned)((N2 / cols_per_tile<T>()) * a.rows);
        if (two) {
            if (!medium_stream2) {
                HIP_TRY(hipStreamCreateWithFlags(&medium_stream2, hipStreamNonBlocking));
                HIP_TRY(hipEventCreateWithFlags(&medium_ev2, hipEventDisableTiming));
            }
            HIP_TRY(hipEventRecord(fork_ev, stream));                  // (behind the uploads and the backup copy, ahead of the first launch)
            HIP_TRY(hipStreamWaitEvent(medium_stream2, fork_ev, 0));
        }
        ++last_launches;
        HIP_TRY(launch_medium(N1, N2, phase, (int)a.nblk, medium_xccs, stream, a));
        HIP_TRY(hipMemcpyAsync(medium_err_host, a.error, sizeof(unsigned), hipMemcpyDeviceToHost, stream));
        medium_err_host[1] = 0u;
        if (two) {
            MediumArgs<T> b = a;
            b.F = F + n; b.Y = Y + n; b.P = P + n;
            b.bar = medium_st + kSet; b.error = reinterpret_cast<unsigned*>(medium_st + kSet + kBarShards + kBarWords);
            // the next XCD of the device's set
            const unsigned mask = xcc_mask(device);
            unsigned nx = (unsigned)medium_xcc;
            do { nx = (nx + 1u) & 31u; } while (!((mask >> nx) & 1u));
            b.xcc = medium_xccs > 1 ? nx : (unsigned)medium_xcc;
            ++last_launches;
            HIP_TRY(launch_medium(N1, N2, phase, (int)b.nblk, medium_xccs, medium_stream2, b));
            HIP_TRY(hipMemcpyAsync(medium_err_host + 1, b.error, sizeof(unsigned), hipMemcpyDeviceToHost, medium_stream2));
            HIP_TRY(hipEventRecord(medium_ev2, medium_stream2));
            HIP_TRY(hipStreamWaitEvent(stream, medium_ev2, 0));
        }
        medium_sched.assign(h, h + nsteps);
        medium_gamma = gamma_d;
        medium_pending = true;
        return SSFM_OK;
        }
    }

    int set_operator(const void* host) {
        if (int rc = use_device()) return rc;
        {
            const T* d = static_cast<const T*>(host);             // (re, im) pairs
            bool flat = true;
            for (int64_t i = 1; i < n_full && flat; ++i) flat = std::memcmp(&d[2 * i], &d[0], sizeof(T)) == 0;
            op_flat_re = flat;
            op_re0 = d[0];
            dimag_valid = false;
            dsplit_valid = false;
        }
        HIP_TRY(hipMemcpyAsync(dnat, host, sizeof(cx<T>) * n_full, hipMemcpyHostToDevice, stream));
        if (is_split()) {           // (the operator's split-order forms are made where they are used: tables_for, split_fly_ready)
            HIP_TRY(hipStreamSynchronize(stream));
            for (auto& t : tabs) t.valid = false;
            have_op = true;
            tags[0] = 0;
            return SSFM_OK;
        }
        hipLaunchKernelGGL((k_make_freq_table<T, 0>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                           (const cx<T>*)dnat, dperm, N1, N2, N2 / Ef, (T)0, inv_n());
        if (dperm_fly)
            hipLaunchKernelGGL((k_make_freq_table<T, 0>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                               (const cx<T>*)dnat, dperm_fly, N1, N2, N2 / Ef_fly, (T)0, inv_n());
        if (dsmall)
            hipLaunchKernelGGL((k_make_freq_table<T, 0>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                               (const cx<T>*)dnat, dsmall, 1, (int)n, (int)n / small_points<T>((int)n), (T)0, inv_n());
        HIP_TRY(hipGetLastError());
        // host buffer may be reused by the caller right after return
        HIP_TRY(hipStreamSynchronize(stream));
        for (auto& t : tabs) t.valid = false;
        for (auto& t : stabs) t.valid = false;
        have_op = true;
        tags[0] = 0;
        return SSFM_OK;
    }

    // exp(D~ h)/N for every step size of a schedule (at most kMaxTables), cached per distinct h.  `one_line`: the table of the
    // single-launch engine (N1 = 1: the table index is the frequency index), else the transposed order of k_freq.
    // A slot that holds a table this schedule needs is never the victim of another of its tables (a round-robin victim could
    // be: schedules {a,b}, {c,d}, {a,e} on one plan would have run a's steps with e's table).
    int tables_for(const std::vector<T>& distinct, const cx<T>** out, bool one_line, int kind = 0) {
        Tab* cache = one_line ? stabs : tabs;
        int& rr = one_line ? stab_rr : tab_rr;
        bool pinned[kMaxTables] = {};
        for (size_t i = 0; i < distinct.size(); ++i) {
            out[i] = nullptr;
            for (int k = 0; k < kMaxTables; ++k)
                if (cache[k].valid && cache[k].kind == kind && std::memcmp(&cache[k].h, &distinct[i], sizeof(T)) == 0) { out[i] = cache[k].ptr; pinned[k] = true; }
        }
        for (size_t i = 0; i < distinct.size(); ++i) {
            if (out[i]) continue;
            int v = rr;
            while (pinned[v]) v = (v + 1) % kMaxTables;      // distinct.size() <= kMaxTables: there is one
            Tab& t = cache[v];
            rr = (v + 1) % kMaxTables;
            pinned[v] = true;
            t.valid = false;
            if (!t.ptr) HIP_TRY(hipMalloc(&t.ptr, sizeof(cx<T>) * n_full));
            t.kind = kind;
            if (is_split() && kind == 1)
                hipLaunchKernelGGL(k_make_split_phase_table<T>, dim3((unsigned)((n_full + 255) / 256)), dim3(256), 0, stream,
                                   (const cx<T>*)dnat, reinterpret_cast<unsigned*>(t.ptr), N1, N2, split_R, distinct[i], split_qf());
            else if (is_split())
                hipLaunchKernelGGL((k_make_split_table<T, 2>), dim3((unsigned)((n_full + 255) / 256)), dim3(256), 0, stream,
                                   (const cx<T>*)dnat, t.ptr, N1, N2, split_R, distinct[i], inv_n_full(), split_qf());
            else if (one_line)
                hipLaunchKernelGGL((k_make_freq_table<T, 2>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                                   (const cx<T>*)dnat, t.ptr, 1, (int)n, (int)n / small_points<T>((int)n), distinct[i], inv_n());
            else if (kind == 1)
                hipLaunchKernelGGL(k_make_phase_table<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                                   (const cx<T>*)dnat, reinterpret_cast<unsigned*>(t.ptr), N1, N2, N2 / Ef, distinct[i]);
            else
                hipLaunchKernelGGL((k_make_freq_table<T, 2>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                                   (const cx<T>*)dnat, t.ptr, N1, N2, N2 / Ef, distinct[i], inv_n());
            HIP_TRY(hipGetLastError());
            t.h = distinct[i]; t.valid = true;
            out[i] = t.ptr;
        }
        return SSFM_OK;
    }

    TimeArgs<T> targs(T gamma, T hh_prev, T hh_next, AdaptState<T>* s, int row0 = 0, int lane = 0) {
        TimeArgs<T> a;
#if SSFM_TRACE
        trace_tag(a, 0, lane);
#endif
        (void)lane;
        a.F = F + (size_t)row0 * n; a.Y = Y + (size_t)row0 * n; a.P = P + (size_t)row0 * n; a.twN = twN; a.twA = twA; a.twB = twB; a.tw1 = tw1; a.st = s; a.zlog = zlog; a.gamma = gamma;
        a.hh_prev = hh_prev; a.hh_next = hh_next; a.N2 = N2; a.rows = 0; a.Qf = N2 / Ef; a.step = 0; a.derive = 0;
        a.s_in = nullptr; a.s_out = nullptr; a.pkeep = nullptr; a.mul = nullptr;
        return a;
    }
    FreqArgs<T> fargs(const cx<T>* tab, T h, const AdaptState<T>* s, int row0 = 0, int lane = 0) {
        FreqArgs<T> a;
#if SSFM_TRACE
        trace_tag(a, 1, lane);
#endif
        (void)lane;
        a.F = Y + (size_t)row0 * n; a.tab = tab; a.tw2 = tw2; a.st = s; a.h = h; a.inv_n = inv_n(); a.N1 = N1; a.rows = 0; a.u16 = u16 ? 1 : 0; a.step = 0; a.amp = (T)0;
        return a;
    }

    // k_freq<FM_FLY_IM> can serve this plan's adaptive steps: a fibre's operator on a complex64 plan in the unit layout; makes the table on first use
    bool fly_imag_ready() {
        if constexpr (sizeof(T) != 4) return false;
        else {
        if (!u16 || !op_flat_re || Ef_fly % 4 != 0) return false;
        if (!dimag_valid) {
            if (!dimag_fly && hipMalloc(&dimag_fly, sizeof(T) * n) != hipSuccess) { (void)hipGetLastError(); return false; }
            hipLaunchKernelGGL(k_make_imag_table<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const cx<T>*)dnat, dimag_fly, N1, N2, N2 / Ef_fly);
            if (hipGetLastError() != hipSuccess) return false;
            dimag_valid = true;
        }
        return true;
        }
    }
    // k_freq<FM_FLY>: D~ and the row twiddles in the order of ITS points per thread
    FreqArgs<T> fargs_fly(T h, const AdaptState<T>* s, int row0 = 0, int lane = 0) {
        FreqArgs<T> a = fargs(dperm_fly ? dperm_fly : dperm, h, s, row0, lane);
        if (tw2_fly) a.tw2 = tw2_fly;
        return a;
    }

    int copy_field_out(void* dst, bool is_device, bool wait) {
        if (int rc = ensure_nat()) return rc;
        HIP_TRY(hipMemcpyAsync(dst, user_field(), sizeof(cx<T>) * n * batch, is_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, stream));
        if (wait) HIP_TRY(hipStreamSynchronize(stream));
        return SSFM_OK;
    }

    // the whole schedule in one launch (k_small); `distinct` holds at most kMaxTables step sizes
    int run_small(T gamma, const T* h, int64_t nsteps, const std::vector<T>& distinct, cx<T>* dsnap = nullptr) {
        SmallArgs<T> a;
        std::memset(&a, 0, sizeof(a));
        if (int rc = tables_for(distinct, a.tab, true)) return rc;
        const size_t hb = sizeof(T) * (size_t)nsteps, need = hb + (size_t)nsteps;
        if (d_hs_cap < need) {
            (void)hipFree(d_hs); d_hs = nullptr; d_hs_cap = 0;
            HIP_TRY(hipMalloc(&d_hs, need + need / 2));
            d_hs_cap = need + need / 2;
        }
        HIP_TRY(hipStreamSynchronize(stream));             // the staging vector may still feed the previous run's copy
        h_sched.resize(need);
        std::memcpy(h_sched.data(), h, hb);
        for (int64_t s = 0; s < nsteps; ++s) {
            unsigned char w = 0;
            for (size_t i = 0; i < distinct.size(); ++i)
                if (std::memcmp(&distinct[i], &h[s], sizeof(T)) == 0) w = (unsigned char)i;
            h_sched[hb + (size_t)s] = w;
        }
        HIP_TRY(hipMemcpyAsync(d_hs, h_sched.data(), need, hipMemcpyHostToDevice, stream));
        a.F = F; a.hs = d_hs; a.which = reinterpret_cast<const unsigned char*>(d_hs) + hb; a.tw = tw_small; a.snap = dsnap; a.gamma = gamma; a.nsteps = (int)nsteps;
        ++last_launches;
        HIP_TRY(launch_small<T>((int)n, batch, stream, a));
        return SSFM_OK;
    }

    // What a capture run (propagate_fixed_capture) adds to propagate_fixed's lanes: where the scalar log of a step goes, and what happens at a capture step.
    // Snapshot 0 (the input) and the last one (the end field) go to the host straight from F; the snapshots in between are written by the capture steps'
    // END launches into slot j % per_block of device block (j / per_block) % nblocks, j = snapshot number - 1, and a block is sent ("flushed") when it is
    // full -- by the helper thread, which waits ON THE HOST for the lanes' END events and tells the lanes' threads, through `flushes_done`, when a block may
    // be written again.
    struct CapRun {
        int64_t every = 0;                  // > 0: snapshots
        char* host = nullptr;               // the caller's buffer
        double* scalars_host = nullptr;
        double* scal = nullptr;             // the wavefronts' pairs: [step][row][per_row][2]
        size_t step_doubles = 0;
        int per_row = 0, nblocks = 0, nlanes = 0;
        size_t fb = 0;
        int64_t per_block = 0, nsnap = 0, nflush = 0, nsteps = 0;
        std::atomic<int64_t> ends_done[kMaxLanesConst];     // flushes whose last END event a lane has recorded
        std::atomic<int64_t> flushes_done{0};               // flushes whose transfer is complete (their block may be written again)
        std::atomic<int64_t> input_done{0};                 // the input's transfer is complete (F may be written)
        std::atomic<int64_t> run_queued{0};                 // 1: the run is queued (ev1 recorded), 2: it failed
        std::atomic<int> failed{0};
        double* scal_at(int64_t step, int row0) const { return scal ? scal + step_doubles * (size_t)step + (size_t)row0 * per_row * 2 : nullptr; }
        bool wait_for(std::atomic<int64_t>& c, int64_t want) {
            for (unsigned spins = 0; c.load(std::memory_order_acquire) < want; ++spins) {
                if (failed.load()) return false;
                if (spins < 4096u) std::this_thread::yield();
                else std::this_thread::sleep_for(std::chrono::microseconds(20));
            }
            return true;
        }
    };
    CapRun cap_cr;                          // (a member: the helper thread works on it after propagate_fixed_capture has returned)
    CapRun* cap_run = nullptr;              // set around the propagate_fixed call of a capture run

    int propagate_fixed(double gamma_d, const T* h, int64_t nsteps, void* snapshots) {
        if (!have_op) return fail(SSFM_ERR_STATE, "ssfm_propagate_fixed: call ssfm_set_linear_operator first");
        if (int rc = use_device()) return rc;
        if (int rc = lane_health()) return rc;             // (the previous run, if it was a long two-lane run: slow lanes are repaired before this one is enqueued)
        const T gamma = (T)gamma_d;
        const int nrows = N1 * batch;
        const int nlanes = lanes_active;                   // (shadows the configured number: a plan that dropped to one lane drives its rows on one stream)
        last_launches = 0;
        timed = false;
        if (nsteps <= 0) return SSFM_OK;
        for (int64_t s = 0; s < nsteps; ++s)
            if (!(h[s] > (T)0) || !std::isfinite((double)h[s]))
                return fail(SSFM_ERR_INVALID, "ssfm_propagate_fixed: step %lld is %g km (must be finite and > 0)", (long long)s, (double)h[s]);
        // distinct step sizes -> operator tables (normally 1, +1 for a short last step)
        std::vector<T> distinct;
        for (int64_t s = 0; s < nsteps && distinct.size() <= (size_t)kMaxTables; ++s) {
            bool seen = false;
            for (T d : distinct) seen = seen || std::memcmp(&d, &h[s], sizeof(T)) == 0;
            if (!seen) distinct.push_back(h[s]);
        }
        const bool use_tables = distinct.size() <= (size_t)kMaxTables && !force_fly;
        std::vector<const cx<T>*> tabptr(distinct.size(), nullptr);
        const bool small_sched = small && use_tables && !profiling && nsteps <= 0x7fffffff;
        // a fibre's operator has one modulus for all frequencies: 4-byte phase tables (ssfm_kernels.hpp FM_PHASE)
        // (complex128: float64 turn fractions, 8 instead of 16 bytes per frequency -- measured -2.5 % at 2^20 x 2 (39.5 against 40.5 us per step) and
        // +5 % at 2^16 x 2, where nothing hides the float64 sincos: from 2^20 samples in all)
        const bool use_phase = use_tables && phase_tables && op_flat_re && (sizeof(T) == 8 ? n * batch >= (1ll << 20) && !is_split() : u16);      // (split plans: 4-byte phases in complex64)
        if (int rc = ensure_sub()) return rc;
        if (is_split() && !use_tables) if (int rc = split_fly_ready()) return rc;
        // plans of 2^12 ... 2^17 samples in the unit layout: the whole schedule in one launch on one XCD (ssfm_kernels.hpp k_medium).  Measured against the
        // one-workgroup-per-row kernel of the small plans (k_small): 5.4 against 6.4 us per step at 8192 samples, 5.6 against 3.5 at 4096 -- so from 8192 on
        const long long med_blocks = (long long)(N2 / cols_per_tile<T>()) * (medium_rows_split() ? 1 : batch);
        const long long med_samples = medium_rows_split() ? n : n * batch;
        const bool med_elig = medium_ok && sizeof(T) == 4 && u16 && E == 8 && Ef == 8 && medium_shape(N1, N2) && use_tables && !profiling
                               && snapshots == nullptr && twA != nullptr && nsteps >= 2 && nsteps <= 0x7fffffff
                               && med_blocks % kBarShards == 0 && med_blocks <= 64 && med_samples <= medium_max_samples && (use_phase || !phase_tables || !op_flat_re);
        const bool go_small = small_sched && snapshots == nullptr && !(med_elig && n >= 8192) && cap_run == nullptr;      // (a capture run: the two-kernel engine)
        const bool go_medium = med_elig && !go_small && cap_run == nullptr;
        last_fell_back = 0;
        last_engine = (go_small || (small_sched && snapshots != nullptr)) ? SSFM_ENGINE_SMALL : go_medium ? SSFM_ENGINE_MEDIUM : is_split() ? SSFM_ENGINE_SPLIT : SSFM_ENGINE_TWO_KERNEL;
        if (use_tables && !go_small && !go_medium)
            if (int rc = tables_for(distinct, tabptr.data(), false, use_phase ? 1 : 0)) return rc;
        auto freq_rows = [&](T hs, int row0, int rows, hipStream_t st_) -> hipError_t {
            const int lane_ = rows > 0 ? row0 / rows : 0;
            ++last_launches;
            if (is_split()) {
                const cx<T>* tp = nullptr;
                if (use_tables)
                    for (size_t i = 0; i < distinct.size(); ++i)
                        if (std::memcmp(&distinct[i], &hs, sizeof(T)) == 0) tp = tabptr[i];
                last_launches += 2;
                const T xr = op_re0 * hs;
                return split_freq(tp, hs, nullptr, 0, row0, rows, st_, use_phase, (T)std::exp((double)xr) * inv_n_full());
            }
            if (use_tables) {
                const cx<T>* tp = nullptr;
                for (size_t i = 0; i < distinct.size(); ++i)
                    if (std::memcmp(&distinct[i], &hs, sizeof(T)) == 0) tp = tabptr[i];
                if (use_phase) {
                    FreqArgs<T> fa = fargs(tp, hs, nullptr, row0, lane_);
                    // the modulus as the complex table forms it: e = exp(Re * h) with the product in T, then e * (1/N) (exact: N = 2^k)
                    const T xr = op_re0 * hs;
                    fa.amp = (T)std::exp((double)xr) * inv_n();
                    return launch_freq<T, FM_PHASE>(N2, N1 * rows, st_, fa, Ef);
                }
                return launch_freq<T, FM_TABLE>(N2, N1 * rows, st_, fargs(tp, hs, nullptr, row0, lane_), Ef);
            }
            return launch_freq<T, FM_FLY>(N2, N1 * rows, st_, fargs_fly(hs, nullptr, row0, lane_), Ef_fly);
        };
        auto freq = [&](T hs) -> hipError_t { return freq_rows(hs, 0, batch, stream); };
        (void)nrows;
        const T half = (T)0.5;
        for (auto& p : prof) p.n = 0;
        HIP_TRY(hipEventRecord(ev0, stream));
        const bool health = snapshots == nullptr && !go_small && !go_medium && nlanes > 1 && !profiling && !SSFM_TRACE && nsteps >= 64 && run_e0 != nullptr && cap_run == nullptr;
        if (health) HIP_TRY(hipEventRecord(run_e0, stream));
        // one lane's launches, start to end.  A capture run (cap_run) adds: the scalar log's address to the column launches, and at a capture step an END
        // launch BESIDE the run -- it only reads the half-transformed field and writes this step's time-order field straight into a slot of a device block
        // (no copy of F); the MID behind it continues exactly as a run without capture does, so the snapshots and the end field are a plain run's, bit for
        // bit.  The lane records an event behind the END that fills a block; the helper thread sends the block to the host once every lane's event is
        // through.  Before a block is written again the lane's THREAD waits (on the host) until the helper has seen its transfer complete: the lanes'
        // threads run ahead of the GPU by at most the ring of blocks, the GPU never waits as long as PCIe keeps up.
        auto lane_run = [&](int g) -> int {
            const int rows = batch / nlanes;
            CapRun* const cr = cap_run;
            const bool snaps = cr && cr->every > 0;
            auto body = [&]() -> int {
                // (round 6 A/B: starting lane 1 of a complex128 run 5 ... 40 us late changes nothing -- 38.6-39.4 against 38.6-38.9 us per step, profiles/r06_c1_ab.txt)
                TimeArgs<T> tb = targs(gamma, 0, h[0] * half, nullptr, g * rows, g);
                if (cr) tb.scal = cr->scal_at(0, g * rows);
                ++last_launches;
                HIP_TRY((launch_time<T, TM_BEGIN>(N1, rows, lane_stream[g], tb, E)));
                int64_t k = 1;                                       // the next snapshot
                for (int64_t s = 0; s < nsteps; ++s) {
                    HIP_TRY(freq_rows(h[s], g * rows, rows, lane_stream[g]));
                    const bool last = s + 1 == nsteps;
                    if (snaps && !last && (s + 1) % cr->every == 0) {
                        const int64_t j = k - 1, fl = j / cr->per_block, slot = j % cr->per_block;
                        const int blk = (int)(fl % cr->nblocks);
                        if (slot == 0 && fl >= cr->nblocks && !cr->wait_for(cr->flushes_done, fl - cr->nblocks + 1))      // the block is written again
                            return fail(SSFM_ERR_HIP, "ssfm_propagate_fixed_capture: the transfers failed");
                        TimeArgs<T> te = targs(gamma, h[s] * half, 0, nullptr, g * rows, g);
                        te.F = reinterpret_cast<cx<T>*>(cap_blocks + cap_block_bytes * (size_t)blk + cr->fb * (size_t)slot) + (size_t)(g * rows) * n;
                        ++last_launches;
                        HIP_TRY((launch_time<T, TM_END>(N1, rows, lane_stream[g], te, E)));
                        if (slot == cr->per_block - 1 || k == cr->nsnap - 2) {          // the block is full (or the last one of the run): it leaves
                            HIP_TRY(hipEventRecord(cap_ev_ends[g][fl], lane_stream[g]));
                            cr->ends_done[g].store(fl + 1, std::memory_order_release);
                        }
                        ++k;
                    }
                    if (last) {
                        if (snaps && !in_place() && !cr->wait_for(cr->input_done, 1))    // (the input's transfer reads F, which this launch overwrites)
                            return fail(SSFM_ERR_HIP, "ssfm_propagate_fixed_capture: the transfers failed");
                        TimeArgs<T> te = targs(gamma, h[s] * half, 0, nullptr, g * rows, g);
                        if (cr) te.scal = cr->scal_at(s + 1, g * rows);
                        ++last_launches;
                        HIP_TRY((launch_time<T, TM_END>(N1, rows, lane_stream[g], te, E)));
                    } else {
                        TimeArgs<T> tm = targs(gamma, h[s] * half, h[s + 1] * half, nullptr, g * rows, g);
                        if (cr) tm.scal = cr->scal_at(s + 1, g * rows);
                        ++last_launches;
                        HIP_TRY((launch_time<T, TM_MID>(N1, rows, lane_stream[g], tm, E)));
                    }
                }
                return SSFM_OK;
            };
            const int rc = body();
            if (rc != SSFM_OK && cr) cr->failed.store(1);
            return rc;
        };
        auto enqueue_steps = [&]() -> int {
            if (cap_run != nullptr && nlanes == 1) return lane_run(0);
            if (nlanes > 1 && ((lane_threads && !profiling && nsteps >= 16) || cap_run != nullptr)) {
                // every lane from a host thread of its own (LaneWorker): lane 0 from this one
                HIP_TRY(hipEventRecord(fork_ev, stream));
                for (int g = 1; g < nlanes; ++g) HIP_TRY(hipStreamWaitEvent(lane_stream[g], fork_ev, 0));
                for (int g = 1; g < nlanes; ++g) { lane_worker[g].start(device); lane_worker[g].submit([&lane_run, g] { return lane_run(g); }); }
                int rc = lane_run(0);
                for (int g = 1; g < nlanes; ++g) { const int r = lane_worker[g].wait(); if (rc == SSFM_OK) rc = r; }
                if (rc != SSFM_OK) return rc;
                for (int g = 1; g < nlanes; ++g) {
                    HIP_TRY(hipEventRecord(lane_ev[g], lane_stream[g]));
                    HIP_TRY(hipStreamWaitEvent(stream, lane_ev[g], 0));
                }
                return SSFM_OK;
            }
            if (nlanes > 1) {
                const int rows = batch / nlanes;
                HIP_TRY(hipEventRecord(fork_ev, stream));
                for (int g = 1; g < nlanes; ++g) HIP_TRY(hipStreamWaitEvent(lane_stream[g], fork_ev, 0));
                for (int g = 0; g < nlanes; ++g) {
                    if (int rc = prof_mark(-1, g)) return rc;
                    ++last_launches;
                    HIP_TRY((launch_time<T, TM_BEGIN>(N1, rows, lane_stream[g], targs(gamma, 0, h[0] * half, nullptr, g * rows, g), E)));
                    if (int rc = prof_mark(0, g)) return rc;
                }
                for (int64_t s = 0; s < nsteps; ++s) {
                    for (int g = 0; g < nlanes; ++g) {
                        HIP_TRY(freq_rows(h[s], g * rows, rows, lane_stream[g]));
                        if (int rc = prof_mark(1, g)) return rc;
                    }
                    for (int g = 0; g < nlanes; ++g) {
                        ++last_launches;
                        if (s + 1 < nsteps)
                            HIP_TRY((launch_time<T, TM_MID>(N1, rows, lane_stream[g], targs(gamma, h[s] * half, h[s + 1] * half, nullptr, g * rows, g), E)));
                        else
                            HIP_TRY((launch_time<T, TM_END>(N1, rows, lane_stream[g], targs(gamma, h[s] * half, 0, nullptr, g * rows, g), E)));
                        if (int rc = prof_mark(0, g)) return rc;
                    }
                }
                for (int g = 0; g < nlanes; ++g) if (int rc = prof_mark(2, g)) return rc;
                for (int g = 1; g < nlanes; ++g) {
                    HIP_TRY(hipEventRecord(lane_ev[g], lane_stream[g]));
                    HIP_TRY(hipStreamWaitEvent(stream, lane_ev[g], 0));
                }
            } else {
                if (int rc = prof_mark(-1)) return rc;
                ++last_launches;
                HIP_TRY((launch_time<T, TM_BEGIN>(N1, batch, stream, targs(gamma, 0, h[0] * half, nullptr), E)));
                if (int rc = prof_mark(0)) return rc;
                for (int64_t s = 0; s < nsteps; ++s) {
                    HIP_TRY(freq(h[s]));
                    if (int rc = prof_mark(1)) return rc;
                    ++last_launches;
                    if (s + 1 < nsteps)
                        HIP_TRY((launch_time<T, TM_MID>(N1, batch, stream, targs(gamma, h[s] * half, h[s + 1] * half, nullptr), E)));
                    else
                        HIP_TRY((launch_time<T, TM_END>(N1, batch, stream, targs(gamma, h[s] * half, 0, nullptr), E)));
                    if (int rc = prof_mark(0)) return rc;
                }
                if (int rc = prof_mark(2)) return rc;
            }
            return SSFM_OK;
        };
        if (snapshots == nullptr) {
#if SSFM_TRACE
            if (int rc = trace_begin((int)(2 * nsteps + 1) * nlanes)) return rc;
#endif
            if (go_small) {
                if (int rc = run_small(gamma, h, nsteps, distinct)) return rc;
            } else if (go_medium) {
                if (int rc = run_medium(gamma, gamma_d, h, nsteps, distinct, use_phase)) return rc;
                if (external_order) {          // (see external_order: the caller may consume the field stream-ordered, without ssfm_synchronize)
                    HIP_TRY(hipEventRecord(ev1, stream));
                    timed = true;
                    return finish_medium();
                }
            } else if (int rc = enqueue_steps()) return rc;
#if SSFM_TRACE
            if (int rc = trace_dump()) return rc;
#endif
        } else {
            // z-resolved capture (reference devices.py:1150-1152, 1184-1186).  The field after every step is
            // copied device-to-device into a block of up to kSnapBlock snapshots -- asynchronous, the step loop
            // never waits for the host -- and a block goes to the host in one transfer.
            const size_t fb = sizeof(cx<T>) * n * batch;
            char* snap = static_cast<char*>(snapshots);
            size_t free_b = 0, total_b = 0;
            HIP_TRY(hipMemGetInfo(&free_b, &total_b));
            int64_t block = (int64_t)std::min<size_t>((size_t)(nsteps + 1), std::max<size_t>(1, std::min<size_t>(free_b / 2, size_t(8) << 30) / fb));
            char* dsnap = nullptr;
            HIP_TRY(hipMalloc(&dsnap, fb * (size_t)block));
            if (!(small_sched && block == nsteps + 1)) last_engine = is_split() ? SSFM_ENGINE_SPLIT : SSFM_ENGINE_TWO_KERNEL;
            if (small_sched && block == nsteps + 1) {
                // a small plan whose whole capture fits the device: the single launch writes every snapshot itself
                hipError_t e = hipMemcpyAsync(dsnap, F, fb, hipMemcpyDeviceToDevice, stream);             // the input
                int rc = e == hipSuccess ? run_small(gamma, h, nsteps, distinct, reinterpret_cast<cx<T>*>(dsnap)) : fail(SSFM_ERR_HIP, "snapshot copy failed: %s", hipGetErrorString(e));
                if (rc == SSFM_OK) {
                    e = hipMemcpyAsync(snap, dsnap, fb * (size_t)(nsteps + 1), hipMemcpyDeviceToHost, stream);
                    if (e == hipSuccess) e = hipStreamSynchronize(stream);
                    if (e != hipSuccess) rc = fail(SSFM_ERR_HIP, "snapshot download failed: %s", hipGetErrorString(e));
                }
                (void)hipFree(dsnap);
                if (rc != SSFM_OK) return rc;
                HIP_TRY(hipEventRecord(ev1, stream));
                timed = true;
                return SSFM_OK;
            }
            auto flush = [&](int64_t first, int64_t count) -> int {
                hipError_t e = hipMemcpyAsync(snap + fb * first, dsnap, fb * (size_t)count, hipMemcpyDeviceToHost, stream);
                if (e == hipSuccess) e = hipStreamSynchronize(stream);
                if (e != hipSuccess) { (void)hipFree(dsnap); return fail(SSFM_ERR_HIP, "snapshot download failed: %s", hipGetErrorString(e)); }
                return SSFM_OK;
            };
            int64_t first = 0, held = 0;                    // snapshots [first, first + held) are in dsnap
            auto capture = [&]() -> hipError_t { return put_time_order(dsnap + fb * held++); };
            hipError_t ce = capture();
            for (int64_t s = 0; s < nsteps && ce == hipSuccess; ++s) {
                if (held == block) { if (int rc = flush(first, held)) return rc; first += held; held = 0; }
                last_launches += 2;
                ce = launch_time<T, TM_BEGIN>(N1, batch, stream, targs(gamma, 0, h[s] * half, nullptr), E);
                if (ce == hipSuccess) ce = freq(h[s]);
                if (ce == hipSuccess) ce = launch_time<T, TM_END>(N1, batch, stream, targs(gamma, h[s] * half, 0, nullptr), E);
                if (ce == hipSuccess) ce = capture();
            }
            if (ce != hipSuccess) { (void)hipFree(dsnap); return fail(SSFM_ERR_HIP, "snapshot run failed: %s", hipGetErrorString(ce)); }
            if (int rc = flush(first, held)) return rc;
            HIP_TRY(hipFree(dsnap));
        }
        ran();
        if (int rc = publish_if_external()) return rc;          // (a caller that orders its own work on the stream finds the natural-order field)
        HIP_TRY(hipEventRecord(ev1, stream));
        timed = true;
        if (health) {
            HIP_TRY(hipEventRecord(run_e1, stream));
            lane_check_pending = true;
            lane_check_launches = (is_split() ? 4 : 2) * nsteps + 1;
        }
        if (snapshots != nullptr) HIP_TRY(hipStreamSynchronize(stream));
        return SSFM_OK;
    }

    // ---- z-resolved capture with a stride and a scalar log (the reference collects the field after EVERY step, devices.py:1150-1152,
    // 1184-1186 -- 16 GiB for the headline run -- and its consumers, devices.py:2326-2563, plot a few hundred of them).  The run keeps the fused two-kernel
    // engine: a capture step ADDS an END launch beside the run, which writes that step's time-order field into a slot of a ring of plan-owned device
    // blocks; a helper thread sends a full block to the caller's (page-locked) buffer on a stream of its own while the run goes on.  The input and the end
    // field go straight from F.  The scalar log is a 16-byte store per wavefront of the column kernels (time_body<..., LOG>, TimeArgs::scal), reduced behind
    // the run.  Asynchronous: the buffers are valid after ssfm_synchronize (or any later call on the plan).
    //
    // NOTHING on the device waits across streams, and that is the point of the helper thread.  The first form of this function ordered the transfers with
    // events (hipStreamWaitEvent on the copy stream behind every lane's END, on the lanes behind the transfers): a stream that sits blocked on an event
    // which is milliseconds away slows the launches of EVERY other queue of the process -- the command processor keeps polling the blocked barrier packet
    // between their dispatches.  tools/attic/barrier_cost.hip, profiles/r05_barrier_cost.txt: two chains of 8.9 us kernels take 17.8 ms with an idle
    // third stream, 19.9 / 21.9 / 40 ms with a third stream blocked on an event 5 / 10 / 40 ms away (whatever its priority), 17.7 ms with a spinning
    // wavefront in its place.  In the capture run that was 1.2-1.4 ms per call (the copy stream's wait for the end of the run, queued 8 ms early -- the
    // host enqueues ahead) and 0.1 ms per snapshot (the marker behind each 0.7 ms transfer): +18-21 % at every = 100 (profiles/r05_capture_ab.txt).
    // Host waits (hipEventSynchronize, hipStreamSynchronize) put nothing into a queue.
    static int64_t capture_count(int64_t nsteps, int64_t every) { return every > 0 ? 1 + (nsteps + every - 1) / every : 0; }
    bool in_place() const { return static_cast<const void*>(Y) == static_cast<const void*>(F); }
    int capture_helper() {
        CapRun* const cr = &cap_cr;
        auto body = [&]() -> int {
            const bool snaps = cr->every > 0;
            if (snaps) {                                                // snapshot 0: the input, from F
                HIP_TRY(hipEventSynchronize(cap_ev_in));
                HIP_TRY(hipMemcpyAsync(cr->host, in_place() ? cap_in : reinterpret_cast<const char*>(F), cr->fb, hipMemcpyDeviceToHost, cap_stream));
                HIP_TRY(hipStreamSynchronize(cap_stream));
                cr->input_done.store(1, std::memory_order_release);
            }
            for (int64_t fl = 0; fl < cr->nflush; ++fl) {
                for (int g = 0; g < cr->nlanes; ++g) {
                    if (!cr->wait_for(cr->ends_done[g], fl + 1)) return fail(SSFM_ERR_HIP, "ssfm_propagate_fixed_capture: the run failed");
                    HIP_TRY(hipEventSynchronize(cap_ev_ends[g][fl]));
                }
                const int64_t first = fl * cr->per_block, count = std::min<int64_t>(cr->per_block, cr->nsnap - 2 - first);
                HIP_TRY(hipMemcpyAsync(cr->host + cr->fb * (size_t)(1 + first), cap_blocks + cap_block_bytes * (size_t)(fl % cr->nblocks), cr->fb * (size_t)count,
                                       hipMemcpyDeviceToHost, cap_stream));
                HIP_TRY(hipStreamSynchronize(cap_stream));
                cr->flushes_done.store(fl + 1, std::memory_order_release);
            }
            if (!cr->wait_for(cr->run_queued, 1) || cr->run_queued.load() != 1) return fail(SSFM_ERR_HIP, "ssfm_propagate_fixed_capture: the run failed");
            HIP_TRY(hipEventSynchronize(ev1));                         // the end of the run on the plan's stream: every lane has joined
            if (snaps) HIP_TRY(hipMemcpyAsync(cr->host + cr->fb * (size_t)(cr->nsnap - 1), F, cr->fb, hipMemcpyDeviceToHost, cap_stream));      // the end field is the last snapshot
            if (cr->scalars_host) {
                // the log goes to the host behind the run: a pair (mean |A|^2, max |A|^2) per row and step
                double* red = cr->scal + cr->step_doubles * (size_t)(cr->nsteps + 1);
                hipLaunchKernelGGL(k_scal_reduce<0>, dim3((unsigned)((cr->nsteps + 1) * batch)), dim3(64), 0, cap_stream, (const double*)cr->scal, red, cr->per_row, 1.0 / (double)n);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipMemcpyAsync(cr->scalars_host, red, sizeof(double) * 2 * (size_t)batch * (size_t)(cr->nsteps + 1), hipMemcpyDeviceToHost, cap_stream));
            }
            HIP_TRY(hipStreamSynchronize(cap_stream));
            return SSFM_OK;
        };
        const int rc = body();
        if (rc != SSFM_OK) cr->failed.store(1);
        return rc;
    }
    int propagate_fixed_capture(double gamma_d, const T* h, int64_t nsteps, int64_t every, void* fields_host, double* scalars_host) {
        if (!have_op) return fail(SSFM_ERR_STATE, "ssfm_propagate_fixed_capture: call ssfm_set_linear_operator first");
        if (nsteps < 1 || nsteps > 0x7fffffff) return fail(SSFM_ERR_INVALID, "ssfm_propagate_fixed_capture: nsteps=%lld", (long long)nsteps);
        if ((fields_host != nullptr) != (every > 0)) return fail(SSFM_ERR_INVALID, "ssfm_propagate_fixed_capture: `fields` and `every` > 0 go together");
        if (!fields_host && !scalars_host) return fail(SSFM_ERR_INVALID, "ssfm_propagate_fixed_capture: nothing to capture");
        if (int rc = no_split("ssfm_propagate_fixed_capture")) return rc;
        if (int rc = use_device()) return rc;                          // (joins the previous capture run's helper)
        if (int rc = lane_health()) return rc;
        for (int64_t s = 0; s < nsteps; ++s)
            if (!(h[s] > (T)0) || !std::isfinite((double)h[s]))
                return fail(SSFM_ERR_INVALID, "ssfm_propagate_fixed_capture: step %lld is %g km (must be finite and > 0)", (long long)s, (double)h[s]);
        const int nl = lanes_active;
        // the capture's own resources, made on first use (the copy stream: normal priority -- the plan's high-priority class keeps its queues for the lanes)
        if (!cap_stream) {
            HIP_TRY(hipStreamCreateWithFlags(&cap_stream, hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&cap_ev_in, hipEventDisableTiming));
        }
        const size_t fb = sizeof(cx<T>) * (size_t)n * batch;
        const int64_t nsnap = capture_count(nsteps, every);
        int64_t per_block = 1, nflush = 0;
        int nblocks = 0;
        if (fields_host && nsnap > 2) {
            // a block leaves for the host as soon as it is full: small blocks (8 MiB; one snapshot of a large field) keep the transfers beside the run and
            // leave little of them behind its end; a ring of up to eight (1 GiB at most) lets the lanes' threads enqueue that far ahead of the transfers
            per_block = (int64_t)std::max<size_t>(1, std::min<size_t>((size_t(8) << 20) / fb, 64));
            per_block = std::min<int64_t>(per_block, nsnap - 2);
            nflush = (nsnap - 2 + per_block - 1) / per_block;
            const size_t bb = fb * (size_t)per_block;
            nblocks = (int)std::min<int64_t>(nflush, std::max<int64_t>(2, std::min<int64_t>(kCapBlocksMax, (int64_t)((size_t(1) << 30) / bb))));
            if (cap_block_bytes != bb || cap_nblocks < nblocks) {
                (void)hipFree(cap_blocks); cap_blocks = nullptr; cap_block_bytes = 0; cap_nblocks = 0;
                HIP_TRY(hipMalloc(&cap_blocks, bb * (size_t)nblocks));
                cap_block_bytes = bb; cap_nblocks = nblocks;
            }
        }
        // the scalar log: a pair per wavefront of the column kernels, row and step; reduced to a pair per row and step behind the run (k_scal_reduce)
        const int per_row = (N2 / cols_per_tile<T>()) * ((N1 * cols_per_tile<T>() / E + 63) / 64);
        const size_t step_doubles = (size_t)batch * per_row * 2, raw_b = sizeof(double) * step_doubles * (size_t)(nsteps + 1);
        const size_t red_b = sizeof(double) * 2 * (size_t)batch * (size_t)(nsteps + 1);
        if (scalars_host && cap_scal_bytes < raw_b + red_b) {
            (void)hipFree(cap_scal); cap_scal = nullptr; cap_scal_bytes = 0;
            HIP_TRY(hipMalloc(&cap_scal, raw_b + red_b));
            cap_scal_bytes = raw_b + red_b;
        }
        // an event per lane and flush (an event that is recorded again before its waiter has looked at it would tie a transfer to a LATER END)
        for (int g = 0; g < nl; ++g)
            while ((int64_t)cap_ev_ends[g].size() < nflush) {
                hipEvent_t e;
                HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
                cap_ev_ends[g].push_back(e);
            }
        CapRun& cr = cap_cr;
        cr.every = fields_host ? every : 0; cr.host = static_cast<char*>(fields_host); cr.scalars_host = scalars_host; cr.scal = scalars_host ? cap_scal : nullptr;
        cr.step_doubles = step_doubles; cr.per_row = per_row; cr.fb = fb; cr.per_block = per_block; cr.nsnap = nsnap; cr.nflush = nflush; cr.nsteps = nsteps;
        cr.nblocks = nblocks; cr.nlanes = nl;
        for (auto& e : cr.ends_done) e.store(0);
        cr.flushes_done.store(0); cr.input_done.store(0); cr.run_queued.store(0); cr.failed.store(0);
        if (fields_host) {
            // the input is in F once everything queued on the plan's stream so far is through.  The plans whose engine transforms F in place (Y == F:
            // complex128 and the complex64 plans outside the unit layout) overwrite it with their first launch: they keep a copy, made on the plan's stream
            if (in_place()) {
                if (cap_in_bytes < fb) {
                    (void)hipFree(cap_in); cap_in = nullptr; cap_in_bytes = 0;
                    HIP_TRY(hipMalloc(&cap_in, fb));
                    cap_in_bytes = fb;
                }
                HIP_TRY(hipMemcpyAsync(cap_in, F, fb, hipMemcpyDeviceToDevice, stream));
            }
            HIP_TRY(hipEventRecord(cap_ev_in, stream));
        }
        cap_worker.start(device);
        cap_worker.submit([this] { return capture_helper(); });
        cap_run = &cr;
        const int run_rc = propagate_fixed(gamma_d, h, nsteps, nullptr);             // (the run itself: propagate_fixed's lanes with the hooks above)
        cap_run = nullptr;
        cr.run_queued.store(run_rc == SSFM_OK ? 1 : 2, std::memory_order_release);
        if (run_rc != SSFM_OK) {
            const std::string err = ssfm::g_err;
            cr.failed.store(1);
            (void)cap_worker.wait();
            (void)hipDeviceSynchronize();
            std::snprintf(ssfm::g_err, sizeof(ssfm::g_err), "%s", err.c_str());
            return run_rc;
        }
        cap_pending = true;
        return SSFM_OK;
    }

    // ---- adaptive run (reference devices.py:1155-1161, 1172-1196 with h = None), in three parts so that a z-resolved
    // capture can go to the host in bounded blocks: adaptive_begin (first step size), adaptive_run (up to `budget`
    // steps; all of them without a capture), adaptive_finish (time-order field, z log).
    struct AdaptRun {
        bool active = false;
        bool tile_private = false;     // U16 plans without capture: the field between END and BEGIN stays in the Y buffer
        T gamma = 0;
        T length = 0;
        int max_steps = 0;
        int step = 0;                  // index of the next step to launch
        StepState<T> now = {};         // state after the last launched step (host copy)
        bool fused = false;            // column kernel of at most 128 workgroups, no capture: END + BEGIN in one launch (TM_MID_A)
        bool medium = false;           // (deferred) a medium plan: the single-launch kernel is k_medium_adapt (one XCD)
        bool deferred = false;         // small plan without capture: nothing launched yet -- the first adaptive_run decides between
        int single_step = 0;           // the single-launch kernel (budget covers the run) and the chunked engine
        T phi_max = 0;
    } ar;
    // ---- z-resolved capture of an adaptive run that keeps the run's engine (ssfm_adaptive_set_capture, round 6).  The reference's own consumer calls FIBER with
    // h = None and return_steps (devices.py:2342); ssfm_adaptive_run(snapshots) reproduces that with three launches per step and a host wait per step.  Here the run
    // keeps its engine (two launches per step, fused, where the plan has it) and a capture step ADDS a launch: k_time<TM_END> behind the step's row pass reads
    // the half-transformed field and writes that step's time-order field into a slot of a ring of plan-owned device blocks (as ssfm_propagate_fixed_capture does);
    // the step number is the host's when it enqueues, the step SIZE the device's (cur[step & 1]: what the fused column launch behind it uses), a launch queued
    // behind the end of the run finds `done` and leaves.  An adaptive run is queued in chunks with a look at the state in between (that look exists anyway):
    // a chunk's snapshots go to one half of the ring and travel to the host on a stream of their own while the next chunk runs into the other half.
    struct AdaptCap {
        bool on = false;
        int64_t every = 0;
        std::vector<int64_t> list;     // or: ascending step numbers (1-based: the field AFTER step s)
        size_t next_in_list = 0;
        char* host = nullptr;
        int64_t capacity = 0;
        int64_t* taken = nullptr;      // step number of every snapshot written
        int64_t* n_taken = nullptr;
        int64_t count = 0;             // snapshots written to the host buffer so far (or queued for it)
        int half_slots = 0;
        int chunk_no = 0;
    } acap;
    hipEvent_t acap_ev[2] = {nullptr, nullptr};      // behind the transfers of the chunk that used half 0 / 1 of the ring (the host waits for it before that half is written again)
    bool acap_wants(int64_t after) {
        if (!acap.on) return false;
        if (acap.every > 0) return after % acap.every == 0;
        while (acap.next_in_list < acap.list.size() && acap.list[acap.next_in_list] < after) ++acap.next_in_list;
        return acap.next_in_list < acap.list.size() && acap.list[acap.next_in_list] == after;
    }
    int adaptive_set_capture(int64_t every, const int64_t* steps, int64_t n_steps, void* fields_host, int64_t capacity, int64_t* taken, int64_t* n_taken) {
        if (!ar.active) return fail(SSFM_ERR_STATE, "ssfm_adaptive_set_capture: call ssfm_adaptive_begin first");
        if (ar.step != 0 || (!ar.deferred && ar.now.steps != 0)) return fail(SSFM_ERR_STATE, "ssfm_adaptive_set_capture: the run has started");
        if ((every > 0) == (steps != nullptr && n_steps > 0)) return fail(SSFM_ERR_INVALID, "ssfm_adaptive_set_capture: either a stride or a list of step numbers");
        if (!fields_host || capacity < 1 || !taken || !n_taken) return fail(SSFM_ERR_INVALID, "ssfm_adaptive_set_capture: NULL argument");
        if (int rc = no_split("ssfm_adaptive_set_capture")) return rc;
        if (int rc = use_device()) return rc;
        acap = AdaptCap();
        acap.on = true; acap.every = every > 0 ? every : 0;
        if (every <= 0) {
            acap.list.assign(steps, steps + n_steps);
            for (size_t i = 0; i < acap.list.size(); ++i)
                if (acap.list[i] < 1 || (i && acap.list[i] <= acap.list[i - 1])) { acap.on = false; return fail(SSFM_ERR_INVALID, "ssfm_adaptive_set_capture: step numbers must ascend from 1"); }
        }
        acap.host = static_cast<char*>(fields_host); acap.capacity = capacity; acap.taken = taken; acap.n_taken = n_taken;
        *n_taken = 0;
        const size_t fb = sizeof(cx<T>) * (size_t)n * batch;
        // the ring: two halves of up to 32 snapshots, 1 GiB in all at most
        int half = (int)std::max<size_t>(1, std::min<size_t>(32, (size_t(512) << 20) / fb));
        acap.half_slots = half;
        const size_t need = fb * (size_t)half * 2;
        if (cap_block_bytes * (size_t)cap_nblocks < need) {
            (void)hipFree(cap_blocks); cap_blocks = nullptr; cap_block_bytes = 0; cap_nblocks = 0;
            HIP_TRY(hipMalloc(&cap_blocks, need));
            cap_block_bytes = need; cap_nblocks = 1;
        }
        if (!cap_stream) {
            HIP_TRY(hipStreamCreateWithFlags(&cap_stream, hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&cap_ev_in, hipEventDisableTiming));
        }
        for (auto& e : acap_ev) if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        if (ar.deferred) {            // (a small plan: the capture needs the launch-per-pass engine)
            const AdaptRun keep = ar;
            if (int rc = adaptive_begin_chunked(keep.gamma, keep.length, keep.phi_max, keep.single_step, keep.max_steps, 0)) return rc;
        }
        return SSFM_OK;
    }

    int adaptive_begin(double gamma_d, double length, double phi_max, int single_step, int64_t max_steps, int capture) {
        if (!have_op) return fail(SSFM_ERR_STATE, "ssfm_propagate_adaptive: call ssfm_set_linear_operator first");
        if (max_steps < 1 || max_steps > (1ll << 30)) return fail(SSFM_ERR_INVALID, "ssfm_propagate_adaptive: max_steps=%lld", (long long)max_steps);
        if (int rc = use_device()) return rc;
        const T gamma = (T)gamma_d;
        if (zlog_cap < max_steps + 1) {
            (void)hipFree(zlog); zlog = nullptr; zlog_cap = 0;
            HIP_TRY(hipMalloc(&zlog, sizeof(T) * (max_steps + 1)));
            zlog_cap = max_steps + 1;
        }
        last_launches = 0;
        last_fell_back = 0;
        last_engine = SSFM_ENGINE_NONE;
        acap.on = false;
        HIP_TRY(hipEventRecord(ev0, stream));
        // (4096 x 2: the one-XCD engine, 8.1 us per step, before the one-workgroup kernel that holds both rows, 10.4)
        bool med_adapt = false;
        if constexpr (sizeof(T) == 4) {
            const long long blocks = (long long)(N2 / cols_per_tile<T>()) * batch;
            med_adapt = medium_ok && medium_adapt_ok && fused_ok && !capture && !single_step && u16 && E == 8 && Ef == 8 && Ef_fly == 8 && medium_shape(N1, N2)
                        && twA != nullptr && blocks % kBarShards == 0 && blocks <= kBarWords && n * batch <= medium_max_samples;
        }
        if (small && !capture && small_adapt_supported<T>((int)n, batch) && !(med_adapt && n == 4096 && batch == 2)) {
            ar = AdaptRun();
            ar.active = true; ar.deferred = true;
            ar.gamma = gamma; ar.length = (T)length; ar.phi_max = (T)phi_max; ar.max_steps = (int)max_steps; ar.single_step = single_step;
            return SSFM_OK;
        }
        if (med_adapt) {
            ar = AdaptRun();
            ar.active = true; ar.deferred = true; ar.medium = true;
            ar.gamma = gamma; ar.length = (T)length; ar.phi_max = (T)phi_max; ar.max_steps = (int)max_steps; ar.single_step = single_step;
            return SSFM_OK;
        }
        return adaptive_begin_chunked(gamma, (T)length, (T)phi_max, single_step, (int)max_steps, capture);
    }
    static constexpr int kAdaptChunkMax = 512;          // steps queued between two looks at the state, at most
    AdaptState<T> adapt_host;          // staging copy of the run's parameters (a member: the copy below is asynchronous)
    // what the host reads back during an adaptive run (the step state, the give-up word) lands in page-locked memory: a transfer of a few bytes into
    // pageable memory goes through the runtime's staging path (measured per look at the state: profiles/r05_adaptive_looks.txt)
    struct AdaptLook { StepState<T> now; unsigned gave_up; unsigned pad[3]; };
    AdaptLook* adapt_look = nullptr;
    int ensure_adapt_look() {
        if (!adapt_look) HIP_TRY(hipHostMalloc(&adapt_look, sizeof(AdaptLook), hipHostMallocDefault));
        return SSFM_OK;
    }
    int upload_adapt_state(T gamma, T length, T phi_max, int max_steps) {
        std::memset(&adapt_host, 0, sizeof(adapt_host));
        adapt_host.length = length;
        adapt_host.phi_max = phi_max;
        adapt_host.abs_gamma = gamma < 0 ? -gamma : gamma;
        adapt_host.adaptive = 1;
        adapt_host.max_steps = max_steps;
        adapt_host.patience = 2000000ll;
        if (const char* e = std::getenv("SSFM_FUSED_PATIENCE_TICKS")) adapt_host.patience = std::atoll(e);
        HIP_TRY(hipMemcpyAsync(st, &adapt_host, sizeof(adapt_host), hipMemcpyHostToDevice, stream));
        return SSFM_OK;
    }
    int adaptive_begin_chunked(T gamma, T length, T phi_max, int single_step, int max_steps, int capture) {
        if (int rc = ensure_sub()) return rc;
        if (is_split()) if (int rc = split_fly_ready()) return rc;
        if (int rc = upload_adapt_state(gamma, length, phi_max, max_steps)) return rc;
        if (!single_step) {
            hipLaunchKernelGGL(k_absmax<T>, dim3(1024), dim3(256), 0, stream, (const cx<T>*)F, (long long)n * batch, st);
            ++last_launches;
        }
        hipLaunchKernelGGL(k_step_control<T>, dim3(1), dim3(64), 0, stream, st, zlog, 0, single_step, 0);
        ++last_launches;
        HIP_TRY(hipGetLastError());
        // the first step size comes back at once: it tells how many steps to queue before the first look at the state
        // (a run of a few dozen steps is then two or three chunks, not five: every look is a 25 us stall)
        if (int rc = ensure_adapt_look()) return rc;
        HIP_TRY(hipMemcpyAsync(&adapt_look->now, reinterpret_cast<const char*>(st) + offsetof(AdaptState<T>, cur), sizeof(StepState<T>), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        const StepState<T> first = adapt_look->now;
        ar = AdaptRun();
        ar.now = first;
        ar.active = true;
        ar.gamma = gamma;
        ar.length = (T)length;
        ar.max_steps = (int)max_steps;
        // U16 plans keep the time-domain field between END and the next BEGIN in the Y buffer, in tile-private 16-byte
        // units (ssfm_kernels.hpp TM_END_Y / TM_BEGIN_Y); a z-resolved capture needs the time-order field after every
        // step and uses the plain modes
        ar.phi_max = phi_max;
        ar.single_step = single_step;
        // a column kernel of at most 64 workgroups (measured: a gain up to there, profiles/r02_medium_adaptive.txt) waits for the global maximum inside the launch (TM_MID_A); the input is kept
        // for the case that the GPU does not run the grid as a whole (then: the three-launch engine, for good)
        // Larger complex64 grids (up to 512 workgroups = two per CU: 2^20 x 2) take the same kernel with a hand-over that has no counter to
        // serialise on (AdaptState::wgmax): the column pass is then ONE launch per step instead of two -- 64 MiB of field traffic per step
        // instead of 96.
        const long long col_blocks = (long long)(N2 / cols_per_tile<T>()) * batch;
        long long fused_max = sizeof(T) == 4 ? kAdaptWords : kAdaptSlots;
        int cus = 0;
        if (col_blocks > kAdaptSlots && (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || col_blocks > 2ll * cus || col_blocks % 64 != 0))
            fused_max = kAdaptSlots;                              // (the whole grid must be resident at once: two workgroups per CU)
        ar.fused = fused_ok && !capture && (N1 == 128 || N1 == 256) && col_blocks <= fused_max && !is_split();
        if (ar.fused) {
            const size_t fb = sizeof(cx<T>) * n * batch;
            if (!fused_backup) HIP_TRY(hipMalloc(&fused_backup, fb));
            HIP_TRY(hipMemcpyAsync(fused_backup, F, fb, hipMemcpyDeviceToDevice, stream));
        }
        // (Two engines that drove the polarisations of an adaptive run on two streams, a lane's kernel waiting INSIDE the launch for the other lane's
        // maxima, were built in round 3 and removed in round 4: both lost their A/B -- 36-38 and 28.9-29.7 against 32-33 and 29.6-31.1 us per step at
        // 2^20 x 2, profiles/r03_adaptive_two_lanes.txt, r03_adaptive_fused_large.txt -- and both stall until their patience runs out whenever the
        // runtime maps the plan's two streams to ONE hardware queue, which it does when the number of live streams of the priority class is 3 mod 4:
        // profiles/r04_order_dependence.txt.)
        ar.tile_private = u16 && !capture && !ar.fused;
        return SSFM_OK;
    }

    // Up to `budget` more steps.  snapshots != NULL: HOST buffer of `budget` fields, the field after each step taken by
    // this call (the caller captured the input itself); the run then synchronises after every step.
    int adaptive_run(int64_t budget, void* snapshots, int64_t* steps_total, int* done) {
        if (!ar.active) return fail(SSFM_ERR_STATE, "ssfm_adaptive_run: no adaptive run in progress");
        if (budget < 1) return fail(SSFM_ERR_INVALID, "ssfm_adaptive_run: budget=%lld", (long long)budget);
        if (int rc = use_device()) return rc;
        if (ar.deferred) {
            if (snapshots != nullptr) return fail(SSFM_ERR_STATE, "ssfm_adaptive_run: the run was not begun with capture");
            const AdaptRun keep = ar;
            if (keep.medium && budget >= (int64_t)keep.max_steps) {
                if constexpr (sizeof(T) == 4) {
                    // the whole run in one launch on one XCD (k_medium_adapt); the input is kept for the case that its workgroups do not all get to run
                    const size_t fb = sizeof(cx<T>) * n * batch;
                    if (!fused_backup) HIP_TRY(hipMalloc(&fused_backup, fb));
                    HIP_TRY(hipMemcpyAsync(fused_backup, F, fb, hipMemcpyDeviceToDevice, stream));
                    if (int rc = upload_adapt_state(keep.gamma, keep.length, keep.phi_max, keep.max_steps)) return rc;
                    hipLaunchKernelGGL(k_absmax<T>, dim3(1024), dim3(256), 0, stream, (const cx<T>*)F, (long long)n * batch, st);
                    hipLaunchKernelGGL(k_step_control<T>, dim3(1), dim3(64), 0, stream, st, zlog, 0, 0, 0);
                    last_launches += 2;
                    if (!medium_st) {
                        HIP_TRY(hipMalloc(&medium_st, sizeof(unsigned long long) * 2 * (kBarShards + kBarWords + 2)));
                        HIP_TRY(hipHostMalloc(&medium_err_host, 2 * sizeof(unsigned)));
                        medium_err_host[0] = medium_err_host[1] = 0u;
                    }
                    HIP_TRY(hipMemsetAsync(medium_st, 0, sizeof(unsigned long long) * (kBarShards + kBarWords + 2), stream));
                    if (int rc = pick_xcc()) return rc;
                    MediumAdaptArgs<T> ma;
                    ma.F = F; ma.Y = Y; ma.P = P; ma.twA = twA; ma.twB = twB; ma.tw1 = tw1; ma.tw2 = tw2; ma.D = dperm; ma.st = st; ma.zlog = zlog;
                    ma.bar = medium_st; ma.error = reinterpret_cast<unsigned*>(medium_st + kBarShards + kBarWords); ma.patience = medium_patience;
                    ma.xcc = (unsigned)medium_xcc; ma.nblk = (unsigned)((N2 / cols_per_tile<T>()) * batch);
                    ma.gamma = keep.gamma; ma.inv_n = inv_n(); ma.rows = batch; ma.Qf = N2 / Ef;
                    ++last_launches;
                    HIP_TRY(launch_medium_adapt(N1, N2, (int)ma.nblk, medium_xccs, stream, ma));
                    unsigned gave_up = 0, gave_up2 = 0;
                    HIP_TRY(hipMemcpyAsync(&gave_up, ma.error, sizeof(unsigned), hipMemcpyDeviceToHost, stream));
                    HIP_TRY(hipMemcpyAsync(&gave_up2, reinterpret_cast<const char*>(st) + offsetof(AdaptState<T>, error), sizeof(unsigned), hipMemcpyDeviceToHost, stream));
                    HIP_TRY(hipMemcpyAsync(&ar.now, reinterpret_cast<const char*>(st) + offsetof(AdaptState<T>, cur), sizeof(ar.now), hipMemcpyDeviceToHost, stream));
                    HIP_TRY(hipStreamSynchronize(stream));
                    if (!gave_up && !gave_up2) {
                        ar.deferred = false;
                        last_engine = SSFM_ENGINE_MEDIUM_ADAPT;
                        if (steps_total) *steps_total = ar.now.steps;
                        if (done) *done = ar.now.done;
                        return SSFM_OK;
                    }
                    // part of the grid never ran beside the rest: the same run on the chunked engine, which this plan then keeps to
                    medium_adapt_ok = false;
                    last_fell_back = 1;
                    ++fallbacks;
                    HIP_TRY(hipMemcpyAsync(F, fused_backup, fb, hipMemcpyDeviceToDevice, stream));
                }
            } else if (keep.medium) {
                // (a caller that takes the run in pieces: the chunked engine)
            } else
            if (budget >= (int64_t)keep.max_steps) {
                // the whole run in one launch (k_small_adapt); the state and the z log are read once, here
                if (int rc = upload_adapt_state(keep.gamma, keep.length, keep.phi_max, keep.max_steps)) return rc;
                SmallAdaptArgs<T> sa;
                sa.F = F; sa.D = dsmall; sa.tw = tw_small; sa.st = st; sa.zlog = zlog; sa.gamma = keep.gamma; sa.inv_n = inv_n(); sa.single_step = keep.single_step;
                ++last_launches;
                HIP_TRY(launch_small_adapt<T>((int)n, batch, stream, sa));
                HIP_TRY(hipMemcpyAsync(&ar.now, reinterpret_cast<const char*>(st) + offsetof(AdaptState<T>, cur), sizeof(ar.now), hipMemcpyDeviceToHost, stream));
                HIP_TRY(hipStreamSynchronize(stream));
                ar.deferred = false;
                last_engine = SSFM_ENGINE_SMALL_ADAPT;
                if (steps_total) *steps_total = ar.now.steps;
                if (done) *done = ar.now.done;
                return SSFM_OK;
            }
            // a budgeted caller: the chunked engine from the start
            if (int rc = adaptive_begin_chunked(keep.gamma, keep.length, keep.phi_max, keep.single_step, keep.max_steps, 0)) return rc;
        }
        if (ar.fused && ar.step == 0 && budget < (int64_t)ar.max_steps) {
            // a caller that takes the run in pieces gets the chunked engine (the fused kernel's launches are queued a whole run ahead);
            // the field itself stays in the tile-private order until ssfm_adaptive_finish (include/ssfm_amd.h)
            ar.fused = false;
            ar.tile_private = u16;
        }
        const int nrows = N1 * batch;
        const size_t fb = sizeof(cx<T>) * n * batch;
        char* snap = static_cast<char*>(snapshots);
        if (snap && ar.tile_private) return fail(SSFM_ERR_STATE, "ssfm_adaptive_run: the run was not begun with capture");
        const int first_step = ar.now.steps;
        auto estimate = [&]() {              // about (L - z) / h steps remain (the step only shrinks towards the clamp at L)
            // Three quarters of them are queued before the next look at the state (a look drains the queue: 25-40 us) -- but the last two dozen all at
            // once and two more: launches behind the end of the run find `done` and leave (a few us each), where the looks of a tail of 7, 2, 1 steps cost more
            const double remain = ar.now.h > (T)0 ? ((double)ar.length - (double)ar.now.z) / (double)ar.now.h : 1.0;
            const int est = remain > 1e6 ? kAdaptChunkMax : (remain <= 24.0 ? (int)remain + 3 : (int)(0.75 * remain) + 1);
            return est < 2 ? 2 : (est > kAdaptChunkMax ? kAdaptChunkMax : est);
        };
        int chunk = snap ? 1 : estimate();
        const bool fly_imag = !is_split() && fly_imag_ready();
        if (acap.on && snap) return fail(SSFM_ERR_STATE, "ssfm_adaptive_run: either ssfm_adaptive_set_capture or `snapshots`");
        while (!ar.now.done && ar.now.steps - first_step < budget) {
            last_engine = is_split() ? SSFM_ENGINE_SPLIT_ADAPT : ar.fused ? SSFM_ENGINE_ADAPT_FUSED : SSFM_ENGINE_ADAPT_3;
            if ((int64_t)chunk > budget - (ar.now.steps - first_step)) chunk = (int)(budget - (ar.now.steps - first_step));
            // a capture run: this chunk's snapshots go to half (chunk_no & 1) of the ring, whose transfers of two chunks ago are waited for ON THE HOST
            std::vector<int64_t> chunk_caps;
            char* ring_half = nullptr;
            if (acap.on) {
                // (short chunks: a chunk's snapshots only leave at the look behind it, and the transfers should run beside the NEXT chunk's kernels -- four
                // snapshots per chunk, 64 steps at least; a look drains the queue for ~40 us)
                if (acap.every > 0 && (int64_t)chunk > std::max<int64_t>(64, acap.every * 4)) chunk = (int)std::max<int64_t>(64, acap.every * 4);
                // ... and a chunk ends right behind a capture step where it can, so that the snapshot leaves at once (the last ones of a run would otherwise
                // travel behind its end)
                if (acap.every > 0 && (int64_t)chunk > acap.every) {
                    const int64_t end = (((int64_t)ar.step + chunk) / acap.every) * acap.every;
                    if (end > (int64_t)ar.step) chunk = (int)(end - (int64_t)ar.step);
                }
                if (acap.chunk_no >= 2) HIP_TRY(hipEventSynchronize(acap_ev[acap.chunk_no & 1]));      // (the transfers of the chunk before the last: NOT the last chunk's, which run beside this one)
                ring_half = cap_blocks + fb * (size_t)acap.half_slots * (size_t)(acap.chunk_no & 1);
            }
            for (int i = 0; i < chunk; ++i, ++ar.step) {
                TimeArgs<T> tb = targs(ar.gamma, 0, 0, st), te = tb;
                tb.step = te.step = ar.step;
                tb.derive = i != 0;         // the first BEGIN of a chunk finds its state in cur[] (k_step_control wrote it)
                if (ar.fused) {
                    // BEGIN once; then FREQ + MID_A per step (MID_A(s) leaves the state of step s + 1 in cur[(s + 1) & 1])
                    if (ar.step == 0) { tb.derive = 0; HIP_TRY((launch_time<T, TM_BEGIN>(N1, batch, stream, tb, E))); ++last_launches; }
                } else if (ar.tile_private && ar.step > 0) HIP_TRY((launch_time<T, TM_BEGIN_Y>(N1, batch, stream, tb, E)));
                else HIP_TRY((launch_time<T, TM_BEGIN>(N1, batch, stream, tb, E)));
                FreqArgs<T> fa = fargs_fly(0, st);
                fa.step = ar.step;
                if (is_split()) {
                    HIP_TRY(split_freq(nullptr, 0, st, ar.step, 0, batch, stream));
                    last_launches += 2;
                } else
                if (fly_imag) {           // (half the operator's bytes: 23.7 -> 22.6 us per step at 2^20 x 2, the same bits)
                    fa.tab = reinterpret_cast<const cx<T>*>(dimag_fly); fa.amp = op_re0;
                    HIP_TRY((launch_freq<T, FM_FLY_IM>(N2, nrows, stream, fa, Ef_fly)));
                } else
                HIP_TRY((launch_freq<T, FM_FLY>(N2, nrows, stream, fa, Ef_fly)));
                if (acap.on && acap_wants((int64_t)ar.step + 1) && acap.count + (int64_t)chunk_caps.size() < acap.capacity) {
                    // the capture step's extra launch: this step's time-order field into a ring slot (the step size is the device's: cur[step & 1])
                    TimeArgs<T> tc = te;
                    tc.F = reinterpret_cast<cx<T>*>(ring_half + fb * chunk_caps.size());
                    HIP_TRY((launch_time<T, TM_END>(N1, batch, stream, tc, E)));
                    ++last_launches;
                    chunk_caps.push_back((int64_t)ar.step + 1);
                    if ((int)chunk_caps.size() >= acap.half_slots || (acap.every == 0 && chunk_caps.size() >= 4 && i + 1 >= 64)) chunk = i + 1;          // (the half is full / enough to send: look at the state)
                }
                if (ar.fused) HIP_TRY((launch_time<T, TM_MID_A>(N1, batch, stream, te, E)));
                else if (ar.tile_private) HIP_TRY((launch_time<T, TM_END_Y>(N1, batch, stream, te, E)));
                else HIP_TRY((launch_time<T, TM_END>(N1, batch, stream, te, E)));
                last_launches += ar.fused ? 2 : 3;
            }
            // the state after the last launched step, for the host (and for the first BEGIN of the next chunk)
            if (!ar.fused) {
                hipLaunchKernelGGL(k_step_control<T>, dim3(1), dim3(64), 0, stream, st, zlog, 1, 0, ar.step);
                ++last_launches;
            }
            const int before = ar.now.steps;
            adapt_look->gave_up = 0u;
            HIP_TRY(hipMemcpyAsync(&adapt_look->now, reinterpret_cast<const char*>(st) + offsetof(AdaptState<T>, cur) + sizeof(StepState<T>) * (ar.step & 1),
                                   sizeof(StepState<T>), hipMemcpyDeviceToHost, stream));
            if (ar.fused) HIP_TRY(hipMemcpyAsync(&adapt_look->gave_up, reinterpret_cast<const char*>(st) + offsetof(AdaptState<T>, error), sizeof(unsigned), hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipStreamSynchronize(stream));
            ar.now = adapt_look->now;
            const unsigned gave_up = adapt_look->gave_up;
            if (acap.on && !(ar.fused && gave_up)) {
                // the chunk has run (the look above waited for it): its snapshots of steps really taken leave for the host beside the next chunk
                for (size_t k = 0; k < chunk_caps.size(); ++k) {
                    if (chunk_caps[k] > (int64_t)ar.now.steps) break;                       // (queued behind the end of the run: never written)
                    HIP_TRY(hipMemcpyAsync(acap.host + fb * (size_t)acap.count, ring_half + fb * k, fb, hipMemcpyDeviceToHost, cap_stream));
                    acap.taken[acap.count++] = chunk_caps[k];
                }
                *acap.n_taken = acap.count;
                HIP_TRY(hipEventRecord(acap_ev[acap.chunk_no & 1], cap_stream));
                ++acap.chunk_no;
            }
            if (ar.fused && gave_up) {
                // the grid did not run as a whole (another job holds CUs): the same run again on the three-launch engine
                const AdaptRun keep = ar;
                if (acap.on) { HIP_TRY(hipStreamSynchronize(cap_stream)); acap.count = 0; acap.next_in_list = 0; acap.chunk_no = 0; *acap.n_taken = 0; }
                fused_ok = false;
                last_fell_back = 1;
                ++fallbacks;
                HIP_TRY(hipMemcpyAsync(F, fused_backup, fb, hipMemcpyDeviceToDevice, stream));
                if (int rc = adaptive_begin_chunked(keep.gamma, keep.length, keep.phi_max, keep.single_step, keep.max_steps, 0)) return rc;
                chunk = estimate();
                continue;
            }
            if (snap && ar.now.steps == before + 1) {        // (chunk = 1: the step just launched was really taken)
                ran();
                if (int rc = copy_field_out(snap + fb * (size_t)(before - first_step), false, true)) return rc;
            }
            if (!snap && !ar.now.done) chunk = estimate();       // do not queue far beyond the end
        }
        if (steps_total) *steps_total = ar.now.steps;
        if (done) *done = ar.now.done;
        return SSFM_OK;
    }

    int adaptive_finish(int64_t* steps_out, double* z_out) {
        if (!ar.active) return fail(SSFM_ERR_STATE, "ssfm_adaptive_finish: no adaptive run in progress");
        if (int rc = use_device()) return rc;
        if (ar.deferred) {           // finished before any step was asked for: the chunked engine's begin leaves the same state behind
            const AdaptRun keep = ar;
            if (int rc = adaptive_begin_chunked(keep.gamma, keep.length, keep.phi_max, keep.single_step, keep.max_steps, 0)) return rc;
        }
        ar.active = false;
        if (acap.on) { HIP_TRY(hipStreamSynchronize(cap_stream)); acap.on = false; }          // (the caller's snapshots are complete)
        if (ar.tile_private && ar.now.steps > 0) {
            HIP_TRY((launch_time<T, TM_UNPACK>(N1, batch, stream, targs(ar.gamma, 0, 0, nullptr), E)));      // Y buffer -> time-order field
            ++last_launches;
        }
        if (ar.now.steps > 0) ran();
        if (int rc = publish_if_external()) return rc;
        HIP_TRY(hipEventRecord(ev1, stream));
        timed = true;
        if (steps_out) *steps_out = ar.now.steps;
        if (z_out) {
            std::vector<T> zl(ar.now.steps + 1);
            HIP_TRY(hipMemcpyAsync(zl.data(), zlog, sizeof(T) * (ar.now.steps + 1), hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipStreamSynchronize(stream));
            for (int i = 0; i <= ar.now.steps; ++i) z_out[i] = (double)zl[i];
        }
        HIP_TRY(hipStreamSynchronize(stream));
        if (!(ar.now.z >= ar.length) && ar.now.steps >= ar.max_steps)
            return fail(SSFM_ERR_INVALID, "ssfm_propagate_adaptive: max_steps=%lld reached at z=%g of %g km",
                        (long long)ar.max_steps, (double)ar.now.z, (double)ar.length);
        return SSFM_OK;
    }

    int propagate_adaptive(double gamma_d, double length, double phi_max, int single_step, int64_t max_steps,
                           int64_t* steps_out, double* z_out, void* snapshots) {
        if (int rc = adaptive_begin(gamma_d, length, phi_max, single_step, max_steps, snapshots != nullptr)) return rc;
        char* snap = static_cast<char*>(snapshots);
        if (snap) if (int rc = copy_field_out(snap, false, true)) return rc;
        if (int rc = adaptive_run(max_steps, snap ? snap + sizeof(cx<T>) * n * batch : nullptr, nullptr, nullptr)) return rc;
        return adaptive_finish(steps_out, z_out);
    }

    // out = ifft(fft(x) * H): BEGIN/END with gamma = 0 are pure column transforms
    int apply_transfer(const void* H_host) {
        if (int rc = use_device()) return rc;
        const int nrows = N1 * batch;
        if (!scratch) HIP_TRY(hipMalloc(&scratch, sizeof(cx<T>) * n * batch));
        cx<T>* hperm = scratch;   // n entries are enough (a split plan: n_full <= n * batch)
        // dnat is a staging buffer: the propagator's own D~ must be re-set after a DM call
        HIP_TRY(hipMemcpyAsync(dnat, H_host, sizeof(cx<T>) * n_full, hipMemcpyHostToDevice, stream));
        drop_operator();
        if (int rc = ensure_sub()) return rc;
        if (is_split())
            hipLaunchKernelGGL((k_make_split_table<T, 1>), dim3((unsigned)((n_full + 255) / 256)), dim3(256), 0, stream, (const cx<T>*)dnat, hperm, N1, N2, split_R, (T)0, inv_n_full(), split_qf());
        else
        hipLaunchKernelGGL((k_make_freq_table<T, 1>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                           (const cx<T>*)dnat, hperm, N1, N2, N2 / Ef, (T)0, inv_n());
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(ev0, stream));
        HIP_TRY((launch_time<T, TM_BEGIN>(N1, batch, stream, targs(0, 0, 0, nullptr), E)));
        if (is_split()) HIP_TRY(split_freq(hperm, 0, nullptr, 0, 0, batch, stream));
        else
        HIP_TRY((launch_freq<T, FM_TABLE>(N2, nrows, stream, fargs(hperm, 0, nullptr), Ef)));
        HIP_TRY((launch_time<T, TM_END>(N1, batch, stream, targs(0, 0, 0, nullptr), E)));
        ran();
        if (int rc = publish_if_external()) return rc;
        HIP_TRY(hipEventRecord(ev1, stream));
        last_launches = is_split() ? 5 : 3;
        timed = true;
        HIP_TRY(hipStreamSynchronize(stream));   // H_host may be released by the caller
        return SSFM_OK;
    }

    int apply_dispersion(double dt_s, double D_s2, void* H_out) {
        if (int rc = use_device()) return rc;
        const int nrows = N1 * batch;
        if (!scratch) HIP_TRY(hipMalloc(&scratch, sizeof(cx<T>) * n * batch));
        cx<T>* hperm = scratch;
        cx<T>* hnat = nullptr;
        if (H_out) {                       // dnat doubles as staging for the natural-order H
            hnat = dnat;
            drop_operator();
        }
        const double val = 1.0 / ((double)n_full * dt_s);
        if (int rc = ensure_sub()) return rc;
        if (is_split())
            hipLaunchKernelGGL(k_make_split_dm_table<T>, dim3((unsigned)((n_full + 255) / 256)), dim3(256), 0, stream, hperm, hnat, N1, N2, split_R, val, D_s2, inv_n_full(), split_qf());
        else
        hipLaunchKernelGGL(k_make_dm_table<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                           hperm, hnat, N1, N2, N2 / Ef, val, D_s2, inv_n());
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(ev0, stream));
        HIP_TRY((launch_time<T, TM_BEGIN>(N1, batch, stream, targs(0, 0, 0, nullptr), E)));
        if (is_split()) HIP_TRY(split_freq(hperm, 0, nullptr, 0, 0, batch, stream));
        else
        HIP_TRY((launch_freq<T, FM_TABLE>(N2, nrows, stream, fargs(hperm, 0, nullptr), Ef)));
        HIP_TRY((launch_time<T, TM_END>(N1, batch, stream, targs(0, 0, 0, nullptr), E)));
        ran();
        if (int rc = publish_if_external()) return rc;
        HIP_TRY(hipEventRecord(ev1, stream));
        last_launches = is_split() ? 5 : 3;
        timed = true;
        if (H_out) {
            HIP_TRY(hipMemcpyAsync(H_out, hnat, sizeof(cx<T>) * n_full, hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipStreamSynchronize(stream));
        }
        return SSFM_OK;
    }

    // A transfer function kept on the device (slot 0 / 1), and its application x <- ifft(fft(x) * H) WITHOUT a
    // host synchronisation: the building block of the chirp-z path for sizes that are not powers of two.
    int transfer_table(const void* H_host, int slot) {
        if (slot < 0 || slot > 1) return fail(SSFM_ERR_INVALID, "ssfm_transfer_table: slot %d", slot);
        if (int rc = no_split("ssfm_transfer_table")) return rc;
        if (int rc = use_device()) return rc;
        if (!xfer_tab[slot]) HIP_TRY(hipMalloc(&xfer_tab[slot], sizeof(cx<T>) * n));
        HIP_TRY(hipMemcpyAsync(dnat, H_host, sizeof(cx<T>) * n, hipMemcpyHostToDevice, stream));      // dnat = staging
        drop_operator();
        tags[1 + slot] = 0;
        hipLaunchKernelGGL((k_make_freq_table<T, 1>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                           (const cx<T>*)dnat, xfer_tab[slot], N1, N2, N2 / Ef, (T)0, inv_n());
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(stream));                  // H_host may be released by the caller
        return SSFM_OK;
    }
    // x <- ifft(fft(ifft(fft(x) * H0) * mul) * H1) with the tables of slots 0 and 1 and a time-domain table `mul` (n entries per row, DEVICE, the plan's
    // precision): the two applications of apply_table around a pointwise product, the middle of it -- inverse pass, product, forward pass -- in ONE column
    // launch (k_time<TM_MID> with TimeArgs::mul).  Five launches instead of seven.  Plain layout only (complex128 plans; complex64 plans of the plain layout).
    // `io` (nullable; complex128 plans): a chirp-z step's two ends folded into the first and the last launch (ChirpIO, ssfm_kernels.hpp) -- the caller's
    // field in, the caller's field out, the plan's own field buffer untouched.
    int apply_tables_mul(const void* mul_dev, const ssfm::ChirpStepIO* io = nullptr) {
        if (int rc = no_split("ssfm_apply_tables_mul")) return rc;
        if (!xfer_tab[0] || !xfer_tab[1]) return fail(SSFM_ERR_STATE, "ssfm_apply_tables_mul: slots 0 and 1 must hold tables");
        if (u16) return fail(SSFM_ERR_UNSUPPORTED, "ssfm_apply_tables_mul: not for plans in the 16-byte-unit layout");
        if (!mul_dev) return fail(SSFM_ERR_INVALID, "ssfm_apply_tables_mul: NULL table");
        TimeArgs<T> tb = targs(0, 0, 0, nullptr);
        if (io) {
            if (sizeof(T) != 8) return fail(SSFM_ERR_UNSUPPORTED, "ssfm_chirp_step: complex128 plans only");
            if (!io->A || !io->P || !io->chirp || io->n < 2 || 2 * io->n - 1 > n) return fail(SSFM_ERR_INVALID, "ssfm_chirp_step: bad field description");
            tb.cz.chirp = static_cast<const cx<T>*>(io->chirp); tb.cz.A = static_cast<cx<T>*>(io->A); tb.cz.P = static_cast<T*>(io->P); tb.cz.n = io->n;
            tb.cz.gamma = (T)io->gamma; tb.cz.hh = (T)io->hh; tb.cz.scale = (T)(1.0 / (double)io->n);
            tb.cz.h = static_cast<const double*>(io->h_dev); tb.cz.done = static_cast<const int*>(io->done_dev);
            tb.cz.maxbits = static_cast<unsigned long long*>(io->maxbits_dev);
        }
        if (int rc = use_device()) return rc;
        if (io) last_engine = SSFM_ENGINE_CHIRP_STEPS;
        if (io && (io->lean || io->c64_line)) {
            // Round 6 (the adaptive runs of long lines): the rows on the plan's lanes, the middle pass reads `n` table entries instead of the line's length, and -- a
            // complex64 caller -- the line holds complex64 values between the passes (chirp_line_run has the reasons and the numbers)
            const bool c64_line = io->c64_line != 0;
            if (c64_line && !line_half_ok()) return fail(SSFM_ERR_UNSUPPORTED, "ssfm_chirp_step: no complex64-storage passes for this plan");
            const int L = (nlanes > 1 && batch % nlanes == 0) ? nlanes : 1;
            const int rows = batch / L;
            auto row_ptr = [&](cx<T>* base, int row0) -> cx<T>* {
                return c64_line ? reinterpret_cast<cx<T>*>(reinterpret_cast<cx<float>*>(base) + (size_t)row0 * n) : base + (size_t)row0 * n;
            };
            auto time_launch = [&](auto mode_tag, const TimeArgs<T>& a, hipStream_t st) -> hipError_t {
                constexpr int MODE = decltype(mode_tag)::value;
                if constexpr (sizeof(T) == 8) { if (c64_line) return launch_time_h<MODE>(N1, rows, st, a, E); }
                return launch_time<T, MODE>(N1, rows, st, a, E);
            };
            auto freq_launch = [&](const cx<T>* tab, int g, hipStream_t st) -> hipError_t {
                FreqArgs<T> a = fargs(tab, 0, nullptr, g * rows, g);
                a.F = row_ptr(Y, g * rows);
                if constexpr (sizeof(T) == 8) { if (c64_line) return launch_freq_h(N2, N1 * rows, st, a, tb.cz.done); }
                return launch_freq<T, FM_TABLE>(N2, N1 * rows, st, a, Ef);
            };
            auto lane_st = [&](int g) { return L > 1 ? lane_stream[g] : stream; };
            auto ends = [&](int g) {
                TimeArgs<T> a = tb;
                a.F = row_ptr(F, g * rows); a.Y = row_ptr(Y, g * rows); a.P = P + (size_t)(g * rows) * n;
                a.cz.A = tb.cz.A + (size_t)(g * rows) * io->n; a.cz.P = tb.cz.P + (size_t)(g * rows) * io->n;
                return a;
            };
            if (L > 1) {
                HIP_TRY(hipEventRecord(fork_ev, stream));
                for (int g = 1; g < L; ++g) HIP_TRY(hipStreamWaitEvent(lane_stream[g], fork_ev, 0));
            }
            for (int g = 0; g < L; ++g) HIP_TRY(time_launch(std::integral_constant<int, TM_BEGIN>{}, ends(g), lane_st(g)));
            for (int g = 0; g < L; ++g) HIP_TRY(freq_launch(xfer_tab[0], g, lane_st(g)));
            for (int g = 0; g < L; ++g) {
                TimeArgs<T> tm = targs(0, 0, 0, nullptr, g * rows, g);
                tm.F = row_ptr(F, g * rows); tm.Y = row_ptr(Y, g * rows);
                tm.mul = static_cast<const cx<T>*>(mul_dev);
                tm.cz.done = tb.cz.done;
                if (io->lean) tm.keep = (int)io->n;
                HIP_TRY(time_launch(std::integral_constant<int, TM_MID>{}, tm, lane_st(g)));
            }
            for (int g = 0; g < L; ++g) HIP_TRY(freq_launch(xfer_tab[1], g, lane_st(g)));
            for (int g = 0; g < L; ++g) HIP_TRY(time_launch(std::integral_constant<int, TM_END>{}, ends(g), lane_st(g)));
            for (int g = 1; g < L; ++g) {
                HIP_TRY(hipEventRecord(lane_ev[g], lane_stream[g]));
                HIP_TRY(hipStreamWaitEvent(stream, lane_ev[g], 0));
            }
            last_launches += 5 * L;
            return SSFM_OK;
        }
        HIP_TRY((launch_time<T, TM_BEGIN>(N1, batch, stream, tb, E)));
        HIP_TRY((launch_freq<T, FM_TABLE>(N2, N1 * batch, stream, fargs(xfer_tab[0], 0, nullptr), Ef)));
        TimeArgs<T> tm = targs(0, 0, 0, nullptr);
        tm.mul = static_cast<const cx<T>*>(mul_dev);
        HIP_TRY((launch_time<T, TM_MID>(N1, batch, stream, tm, E)));
        HIP_TRY((launch_freq<T, FM_TABLE>(N2, N1 * batch, stream, fargs(xfer_tab[1], 0, nullptr), Ef)));
        HIP_TRY((launch_time<T, TM_END>(N1, batch, stream, tb, E)));
        last_launches += 5;
        return SSFM_OK;
    }
    // A fixed-step chirp-z run on the plan's line (plain layout), the field already there as A c, zero from `keep` up: the chirps on either side of a step
    // cancel against the neighbouring steps' (c conj(c) = 1; a rotation commutes with them), so a step is FOUR launches -- row pass (slot 0), column
    // pass with the table product (TM_MID_L, mul[which[s]]: exp(D~ h) / n below keep, zero above), row pass (slot 1), column pass with the second half
    // rotation of this step and the first of the next in one (the padding set to zero first) -- instead of the five of ssfm_chirp_step; the caller
    // multiplies by conj(c) when the run is over.
    // half: the line holds complex64 values between the passes (a complex64 caller's run; line_half_ok() says whether this plan has those kernels)
    bool line_half_ok() const { return sizeof(T) == 8 && !u16 && !is_split() && line_half_shape(N1, N2, E, Ef); }
    int chirp_line_run(const void* const* mul, const unsigned char* which, const double* hs, int64_t nsteps, double gamma_d, int64_t keep, int c64_line = 0) {
        if (int rc = no_split("chirp-z")) return rc;
        if (c64_line && !line_half_ok()) return fail(SSFM_ERR_UNSUPPORTED, "ssfm_chirp_line_run: no complex64-storage passes for this plan");
        if (!xfer_tab[0] || !xfer_tab[1]) return fail(SSFM_ERR_STATE, "ssfm_chirp_line_run: slots 0 and 1 must hold tables");
        if (u16) return fail(SSFM_ERR_UNSUPPORTED, "ssfm_chirp_line_run: not for plans in the 16-byte-unit layout");
        if (!mul || !which || !hs || nsteps < 1 || keep < 2 || keep > n) return fail(SSFM_ERR_INVALID, "ssfm_chirp_line_run: bad arguments");
        for (int64_t s = 0; s < nsteps; ++s)
            if (!mul[which[s]]) return fail(SSFM_ERR_INVALID, "ssfm_chirp_line_run: step %lld has no table", (long long)s);
        if (int rc = use_device()) return rc;
        const T gamma = (T)gamma_d, half = (T)0.5;
        // Round 6: the rows go out on the plan's LANES (one stream per half of the rows, as the fixed-step run of a power of two does) -- a column pass is bound by
        // latencies (one wavefront per SIMD in complex128), a row pass by bytes, and side by side they fill each other's gaps: 2^22-point rows x 2, complex128,
        // 176 us per step in one launch per pass against 149 on two lanes (tools/big_n_c128.py).  SSFM_LANES=1: one launch per pass as before.
        const int L = (nlanes > 1 && batch % nlanes == 0) ? nlanes : 1;
        const int rows = batch / L;
        // (c64_line: the field buffers hold complex64 values -- a row starts `n` float pairs after the one before)
        auto row_ptr = [&](cx<T>* base, int row0) -> cx<T>* {
            return c64_line ? reinterpret_cast<cx<T>*>(reinterpret_cast<cx<float>*>(base) + (size_t)row0 * n) : base + (size_t)row0 * n;
        };
        auto TA = [&](T g_, T hp, T hn, int g) {
            TimeArgs<T> a = targs(g_, hp, hn, nullptr, g * rows, g);
            a.F = row_ptr(F, g * rows); a.Y = row_ptr(Y, g * rows);
            return a;
        };
        auto FA = [&](const cx<T>* tab, int g) {
            FreqArgs<T> a = fargs(tab, 0, nullptr, g * rows, g);
            a.F = row_ptr(Y, g * rows);
            return a;
        };
        auto time_launch = [&](auto mode_tag, const TimeArgs<T>& a, hipStream_t st) -> hipError_t {
            constexpr int MODE = decltype(mode_tag)::value;
            if constexpr (sizeof(T) == 8) { if (c64_line) return launch_time_h<MODE>(N1, rows, st, a, E); }
            return launch_time<T, MODE>(N1, rows, st, a, E);
        };
        auto freq_launch = [&](const FreqArgs<T>& a, hipStream_t st) -> hipError_t {
            if constexpr (sizeof(T) == 8) { if (c64_line) return launch_freq_h(N2, N1 * rows, st, a); }
            return launch_freq<T, FM_TABLE>(N2, N1 * rows, st, a, Ef);
        };
        auto lane_st = [&](int g) { return L > 1 ? lane_stream[g] : stream; };
        if (L > 1) {
            HIP_TRY(hipEventRecord(fork_ev, stream));
            for (int g = 1; g < L; ++g) HIP_TRY(hipStreamWaitEvent(lane_stream[g], fork_ev, 0));
        }
        for (int g = 0; g < L; ++g) HIP_TRY(time_launch(std::integral_constant<int, TM_BEGIN>{}, TA(gamma, (T)0, (T)hs[0] * half, g), lane_st(g)));
        for (int64_t s = 0; s < nsteps; ++s) {
            for (int g = 0; g < L; ++g) HIP_TRY(freq_launch(FA(xfer_tab[0], g), lane_st(g)));
            for (int g = 0; g < L; ++g) {
                TimeArgs<T> tl = TA((T)0, (T)0, (T)0, g);
                tl.mul = static_cast<const cx<T>*>(mul[which[s]]);
                tl.keep = (int)keep;
                HIP_TRY(time_launch(std::integral_constant<int, TM_MID_L>{}, tl, lane_st(g)));
            }
            for (int g = 0; g < L; ++g) HIP_TRY(freq_launch(FA(xfer_tab[1], g), lane_st(g)));
            for (int g = 0; g < L; ++g) {
                if (s + 1 < nsteps) {
                    TimeArgs<T> tm = TA(gamma, (T)hs[s] * half, (T)hs[s + 1] * half, g);
                    tm.keep = (int)keep;
                    HIP_TRY(time_launch(std::integral_constant<int, TM_MID>{}, tm, lane_st(g)));
                } else {
                    HIP_TRY(time_launch(std::integral_constant<int, TM_END>{}, TA(gamma, (T)hs[s] * half, (T)0, g), lane_st(g)));
                }
            }
        }
        for (int g = 1; g < L; ++g) {
            HIP_TRY(hipEventRecord(lane_ev[g], lane_stream[g]));
            HIP_TRY(hipStreamWaitEvent(stream, lane_ev[g], 0));
        }
        last_launches += (1 + 4 * nsteps) * L;
        last_engine = SSFM_ENGINE_CHIRP_STEPS;
        last_fell_back = 0;
        return SSFM_OK;
    }
    // A fixed-step chirp-z run of a field of nn <= n / 2 samples per row in ONE launch (k_small_chirp); SSFM_ERR_UNSUPPORTED (nothing launched) when the
    // plan is not a complex128 plan of the one-workgroup-per-row engine.
    int chirp_small(void* A, const void* chirp, const void* Dt, int64_t nn, double gamma, const double* hs, int64_t nsteps) {
        if (int rc = no_split("chirp-z")) return rc;
        if (!small || !tw_small || n > 4096) return fail(SSFM_ERR_UNSUPPORTED, "ssfm_chirp_small: a plan of 256 ... 4096 samples is needed");
        if (!A || !chirp || !Dt || !hs || nn < 2 || 2 * nn - 1 > n || nsteps < 1 || nsteps > 0x7fffffff) return fail(SSFM_ERR_INVALID, "ssfm_chirp_small: bad arguments");
        for (int64_t s = 0; s < nsteps; ++s)
            if (!(hs[s] > 0) || !std::isfinite(hs[s])) return fail(SSFM_ERR_INVALID, "ssfm_chirp_small: step %lld is %g km (must be finite and > 0)", (long long)s, hs[s]);
        if (int rc = use_device()) return rc;
        const size_t need = sizeof(double) * (size_t)nsteps;
        if (d_hs_cap < need) {
            (void)hipFree(d_hs); d_hs = nullptr; d_hs_cap = 0;
            HIP_TRY(hipMalloc(&d_hs, need + need / 2));
            d_hs_cap = need + need / 2;
        }
        HIP_TRY(hipStreamSynchronize(stream));
        HIP_TRY(hipMemcpyAsync(d_hs, hs, need, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream));             // (hs is the caller's)
        SmallChirpArgs<T> a;
        a.A = static_cast<cx<T>*>(A); a.chirp = static_cast<const cx<T>*>(chirp); a.Dt = static_cast<const cx<T>*>(Dt); a.hs = reinterpret_cast<const double*>(d_hs);
        if (int rc = chirp_line_table(&a.tw)) return rc;
        a.gamma = (T)gamma; a.n = (int)nn; a.nsteps = (int)nsteps;
        last_launches = 1;
        last_engine = SSFM_ENGINE_CHIRP_SMALL; last_fell_back = 0;
        HIP_TRY(launch_small_chirp<T>((int)n, batch, stream, a));
        return SSFM_OK;
    }
    // ... and the adaptive run (k_small_chirp_adapt): the rows' workgroups exchange their maxima through memory every step, so all of them must be
    // resident at once (at most kChirpAdaptRows rows).  Synchronous: the z log and the step count come back.
    int chirp_line_table(const cx<T>** tw) {
        *tw = tw_small;
        if (chirp_points<T>((int)n) == small_points<T>((int)n)) return SSFM_OK;
        if (!tw_chirp)
            if (int rc = make_line_table(&tw_chirp, (int)n, chirp_points<T>((int)n))) return rc;
        *tw = tw_chirp;
        return SSFM_OK;
    }
    static constexpr int kChirpAdaptRows = 16;
    int chirp_small_adapt(void* A, const void* chirp, const void* Dt, int64_t nn, double gamma, double length, double phi_max, int f32, int64_t max_steps,
                          double* z_out, int64_t* steps_out) {
        if (!small || !tw_small || n > 4096 || batch > kChirpAdaptRows)
            return fail(SSFM_ERR_UNSUPPORTED, "ssfm_chirp_small_adapt: a plan of 256 ... 4096 samples and at most %d rows is needed", kChirpAdaptRows);
        if (!A || !chirp || !Dt || nn < 2 || 2 * nn - 1 > n || max_steps < 1 || max_steps > 0x7ffffff0 || !(length > 0) || !(phi_max > 0) || gamma == 0.0)
            return fail(SSFM_ERR_INVALID, "ssfm_chirp_small_adapt: bad arguments");
        if (int rc = use_device()) return rc;
        const size_t words = 2 * (size_t)batch * 2, need = sizeof(double) * (size_t)(max_steps + 1) + 8 * words + 16;
        if (d_hs_cap < need) {
            HIP_TRY(hipStreamSynchronize(stream));
            (void)hipFree(d_hs); d_hs = nullptr; d_hs_cap = 0;
            HIP_TRY(hipMalloc(&d_hs, need + need / 2));
            d_hs_cap = need + need / 2;
        }
        unsigned char* base = reinterpret_cast<unsigned char*>(d_hs);
        SmallChirpAdaptArgs<T> a;
        a.A = static_cast<cx<T>*>(A); a.chirp = static_cast<const cx<T>*>(chirp); a.Dt = static_cast<const cx<T>*>(Dt);
        if (int rc = chirp_line_table(&a.tw)) return rc;
        a.xw = reinterpret_cast<unsigned long long*>(base); a.out = reinterpret_cast<int*>(base + 8 * words); a.zlog = reinterpret_cast<double*>(base + 8 * words + 16);
        a.phi_max = phi_max; a.abs_gamma = gamma < 0 ? -gamma : gamma; a.length = length; a.patience = medium_patience;
        a.gamma = (T)gamma; a.n = (int)nn; a.max_steps = (int)max_steps; a.f32 = f32;
        HIP_TRY(hipMemsetAsync(base, 0, 8 * words + 16, stream));
        // the caller's field is kept (a few dozen KiB): rows that had finished when another row's workgroup gave up have stored their result
        void* keep = nullptr;
        const size_t fbytes = sizeof(cx<T>) * (size_t)nn * (size_t)batch;
        if (int rc = workspace(3, fbytes, &keep)) return rc;
        HIP_TRY(hipMemcpyAsync(keep, A, fbytes, hipMemcpyDeviceToDevice, stream));
        last_launches = 1;
        last_engine = SSFM_ENGINE_CHIRP_SMALL_ADAPT; last_fell_back = 0;
        HIP_TRY(launch_small_chirp_adapt<T>((int)n, batch, stream, a));
        int out[3] = {0, 0, 0};
        HIP_TRY(hipMemcpyAsync(out, a.out, sizeof(out), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        if (out[1]) {
            // a busy device: not all rows' workgroups were resident together.  The field is put back as it came (rows that had finished have
            // stored theirs) and the caller's other path can run -- as every other single-launch engine of this library does
            if (out[2] != 0) {
                HIP_TRY(hipMemcpyAsync(A, keep, fbytes, hipMemcpyDeviceToDevice, stream));
                HIP_TRY(hipStreamSynchronize(stream));
            }
            ++fallbacks;
            last_fell_back = 1;
            return fail(SSFM_ERR_UNSUPPORTED, "ssfm_chirp_small_adapt: the rows' workgroups did not meet within the patience; the field is unchanged");
        }
        if (z_out) HIP_TRY(hipMemcpy(z_out, a.zlog, sizeof(double) * (size_t)(out[0] + 1), hipMemcpyDeviceToHost));
        if (steps_out) *steps_out = out[0];
        return SSFM_OK;
    }
    // Can this plan run the one-launch chirp-z engine of 2048 < nn <= n / 2 samples (chirp_medium / chirp_medium_adapt below)?  If so, slots 0 and 1
    // hold the two convolutions' transfer functions for that length when this returns.
    int chirp_medium_tables(int64_t nn) {
        if (int rc = no_split("chirp-z")) return rc;
        if constexpr (sizeof(T) != 4) { (void)nn; return fail(SSFM_ERR_UNSUPPORTED, "ssfm_chirp_medium: complex64 plans only"); }
        else {
        const long long blocks = (long long)(N2 / cols_per_tile<T>()) * batch;
        if (!medium_ok || !u16 || E != 8 || Ef != 8 || !medium_shape(N1, N2) || n < 8192 || blocks % kBarShards != 0 || blocks > 64
            || n * batch > medium_max_samples || medium_xcc < 0)
            return fail(SSFM_ERR_UNSUPPORTED, "ssfm_chirp_medium: a complex64 plan of 2^13 ... 2^17 points (2^17 in all rows) with the one-XCD engine is needed");
        if (nn < 2 || 2 * nn - 1 > n) return fail(SSFM_ERR_INVALID, "ssfm_chirp_medium: %lld samples do not fit a line of %lld", (long long)nn, (long long)n);
        if (int rc = use_device()) return rc;
        const unsigned gM = (unsigned)std::min<long long>((n + 255) / 256, 4096);
        // the two convolutions' transfer functions, generated and transformed on the device once per length (the slots' tags say what they hold)
        const uint64_t want = 0xC412000000000000ull ^ (uint64_t)nn;
        if (!xfer_tab[0] || !xfer_tab[1] || tags[1] != want || tags[2] != want) {
            // ... in DOUBLE on a complex128 plan of the line's length that lives for this block, rounded once: an error of the tables is the same at every
            // step and adds up over a run (tables from the complex64 transform itself: 1.6e-5 ... 2.6e-5 from the reference result after 100 steps, the
            // tolerance is 2e-5; rounded from double: what the data's own rounding leaves).  The kernel of the inverse transform is the conjugate of
            // the forward one's, and both are symmetric on the line: H1 = conj(H0).
            tags[1] = tags[2] = 0;
            for (int slot = 0; slot < 2; ++slot)
                if (!xfer_tab[slot]) HIP_TRY(hipMalloc(&xfer_tab[slot], sizeof(cx<T>) * n));
            if (!scratch) HIP_TRY(hipMalloc(&scratch, sizeof(cx<T>) * n * batch));
            std::unique_ptr<PlanT<double>> wide(new (std::nothrow) PlanT<double>());
            if (!wide) return fail(SSFM_ERR_INVALID, "out of host memory");
            wide->precision = SSFM_C128;
            if (int rc = wide->init(device, n, 1)) return rc;
            hipLaunchKernelGGL(k_cm_kernel_line<double>, dim3(gM), dim3(256), 0, wide->stream, wide->F, (long long)nn, (long long)n, 0);
            if (int rc = wide->field_spectrum()) return rc;
            HIP_TRY(hipStreamSynchronize(wide->stream));
            for (int slot = 0; slot < 2; ++slot) {
                hipLaunchKernelGGL(k_cm_narrow, dim3(gM), dim3(256), 0, stream, (const cx<double>*)wide->scratch, scratch, (long long)n, slot);
                hipLaunchKernelGGL((k_make_freq_table<T, 1>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                                   (const cx<T>*)scratch, xfer_tab[slot], N1, N2, N2 / Ef, (T)0, inv_n());
            }
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipStreamSynchronize(stream));
            HIP_TRY(hipSetDevice(device));
            last_launches += 8;
            tags[1] = tags[2] = want;
        }
        return SSFM_OK;
        }
    }
    // A fixed-step chirp-z run of nn samples per row (2 nn - 1 <= n) on the line of a MEDIUM complex64 plan (2^13 ... 2^17 points) in one launch on one XCD
    // (k_medium_chirp) between two pointwise launches.  Synchronous.  SSFM_ERR_UNSUPPORTED with A as it was: not such a plan, more than kMaxTables step
    // sizes, or the launch's workgroups did not meet within the patience (the engine is then off for this plan).
    int chirp_medium(void* A_, const void* chirp_, const void* Dt_, int64_t nn, double gamma, const double* hs, int64_t nsteps) {
        if constexpr (sizeof(T) != 4) { (void)A_; (void)chirp_; (void)Dt_; (void)nn; (void)gamma; (void)hs; (void)nsteps; return fail(SSFM_ERR_UNSUPPORTED, "ssfm_chirp_medium: complex64 plans only"); }
        else {
        if (!A_ || !chirp_ || !Dt_ || !hs || nsteps < 1 || nsteps > 0x7fffffff) return fail(SSFM_ERR_INVALID, "ssfm_chirp_medium: bad arguments");
        std::vector<T> hf((size_t)nsteps), distinct;
        for (int64_t s = 0; s < nsteps; ++s) {
            if (!(hs[s] > 0) || !std::isfinite(hs[s])) return fail(SSFM_ERR_INVALID, "ssfm_chirp_medium: step %lld is %g km (must be finite and > 0)", (long long)s, hs[s]);
            hf[(size_t)s] = (T)hs[s];
            bool seen = false;
            for (const T d : distinct) seen = seen || std::memcmp(&d, &hf[(size_t)s], sizeof(T)) == 0;
            if (!seen) distinct.push_back(hf[(size_t)s]);
            if (distinct.size() > (size_t)kMaxTables) return fail(SSFM_ERR_UNSUPPORTED, "ssfm_chirp_medium: more than %d distinct step sizes", kMaxTables);
        }
        last_launches = 0;
        if (int rc = chirp_medium_tables(nn)) return rc;
        const long long blocks = (long long)(N2 / cols_per_tile<T>()) * batch;
        cx<T>* A = static_cast<cx<T>*>(A_);
        const cx<T>* chirp = static_cast<const cx<T>*>(chirp_);
        const cx<T>* Dt = static_cast<const cx<T>*>(Dt_);
        const unsigned gM = (unsigned)std::min<long long>((n + 255) / 256, 4096);
        cx<T>* mulbase = nullptr;
        if (int rc = workspace(0, sizeof(cx<T>) * (size_t)n * distinct.size(), reinterpret_cast<void**>(&mulbase))) return rc;
        MediumChirpArgs<T> a;
        std::memset(&a, 0, sizeof(a));
        for (size_t i = 0; i < distinct.size(); ++i) {
            hipLaunchKernelGGL(k_cm_mul_table<T>, dim3(gM), dim3(256), 0, stream, Dt, mulbase + i * (size_t)n, (long long)nn, (long long)n, distinct[i], (T)(1.0 / (double)nn));
            a.mul[i] = mulbase + i * (size_t)n;
            ++last_launches;
        }
        const size_t hb = sizeof(T) * (size_t)nsteps, need = hb + (size_t)nsteps;
        if (d_hs_cap < need) {
            HIP_TRY(hipStreamSynchronize(stream));
            (void)hipFree(d_hs); d_hs = nullptr; d_hs_cap = 0;
            HIP_TRY(hipMalloc(&d_hs, need + need / 2));
            d_hs_cap = need + need / 2;
        }
        HIP_TRY(hipStreamSynchronize(stream));             // the staging vector may still feed the previous run's copy
        h_sched.resize(need);
        std::memcpy(h_sched.data(), hf.data(), hb);
        for (int64_t s = 0; s < nsteps; ++s) {
            unsigned char w = 0;
            for (size_t i = 0; i < distinct.size(); ++i)
                if (std::memcmp(&distinct[i], &hf[(size_t)s], sizeof(T)) == 0) w = (unsigned char)i;
            h_sched[hb + (size_t)s] = w;
        }
        HIP_TRY(hipMemcpyAsync(d_hs, h_sched.data(), need, hipMemcpyHostToDevice, stream));
        if (!medium_st) {
            HIP_TRY(hipMalloc(&medium_st, sizeof(unsigned long long) * 2 * (kBarShards + kBarWords + 2)));
            HIP_TRY(hipHostMalloc(&medium_err_host, 2 * sizeof(unsigned)));
            medium_err_host[0] = medium_err_host[1] = 0u;
        }
        constexpr size_t kSet = kBarShards + kBarWords + 2;
        HIP_TRY(hipMemsetAsync(medium_st, 0, sizeof(unsigned long long) * 2 * kSet, stream));
        a.F = F; a.Y = Y; a.P = P; a.twA = twA; a.twB = twB; a.tw1 = tw1; a.tw2 = tw2;
        a.H[0] = xfer_tab[0]; a.H[1] = xfer_tab[1];
        a.hs = d_hs; a.which = reinterpret_cast<const unsigned char*>(d_hs) + hb;
        a.bar = medium_st; a.error = reinterpret_cast<unsigned*>(medium_st + kBarShards + kBarWords); a.patience = medium_patience;
        a.xcc = (unsigned)medium_xcc;
        a.gamma = (T)gamma; a.n = (int)nn; a.nsteps = (int)nsteps; a.rows = batch; a.Qf = N2 / Ef;
        a.nblk = (unsigned)blocks;
        const unsigned gA = (unsigned)std::min<long long>((n * batch + 255) / 256, 4096);
        hipLaunchKernelGGL(k_cm_pre<T>, dim3(gA), dim3(256), 0, stream, (const cx<T>*)A, chirp, F, (long long)nn, (long long)n, batch);
        HIP_TRY(launch_medium_chirp(N1, N2, (int)a.nblk, medium_xccs, stream, a));
        hipLaunchKernelGGL(k_cm_post<T>, dim3(gA), dim3(256), 0, stream, A, chirp, (const cx<T>*)F, (long long)nn, (long long)n, batch, (const unsigned*)a.error);
        HIP_TRY(hipGetLastError());
        last_launches += 3;
        HIP_TRY(hipMemcpyAsync(medium_err_host, a.error, sizeof(unsigned), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        last_engine = SSFM_ENGINE_CHIRP_MEDIUM;
        last_fell_back = 0;
        if (medium_err_host[0] != 0u) {
            medium_err_host[0] = 0u;
            medium_ok = false;
            ++fallbacks;
            last_fell_back = 1;
            return fail(SSFM_ERR_UNSUPPORTED, "ssfm_chirp_medium: the launch's workgroups did not meet within the patience; the field is unchanged");
        }
        return SSFM_OK;
        }
    }
    // ... and the adaptive run (k_medium_chirp_adapt): the step rule of ssfm_chirp_small_adapt in float32.  Synchronous.
    int chirp_medium_adapt(void* A_, const void* chirp_, const void* Dt_, int64_t nn, double gamma_d, double length, double phi_max, int64_t max_steps,
                           double* z_out, int64_t* steps_out) {
        if constexpr (sizeof(T) != 4) { (void)A_; (void)chirp_; (void)Dt_; (void)nn; (void)gamma_d; (void)length; (void)phi_max; (void)max_steps; (void)z_out; (void)steps_out;
                                        return fail(SSFM_ERR_UNSUPPORTED, "ssfm_chirp_medium_adapt: complex64 plans only"); }
        else {
        const long long blocks = (long long)(N2 / cols_per_tile<T>()) * batch;
        if (!medium_adapt_ok || !fused_ok || Ef_fly != 8 || blocks > kBarWords) return fail(SSFM_ERR_UNSUPPORTED, "ssfm_chirp_medium_adapt: the plan's one-launch adaptive engine is off");
        if (!A_ || !chirp_ || !Dt_ || !(gamma_d != 0.0) || !(length > 0) || !(phi_max > 0) || max_steps < 1 || max_steps > (1ll << 30))
            return fail(SSFM_ERR_INVALID, "ssfm_chirp_medium_adapt: bad arguments");
        last_launches = 0;
        if (int rc = chirp_medium_tables(nn)) return rc;
        cx<T>* A = static_cast<cx<T>*>(A_);
        const cx<T>* chirp = static_cast<const cx<T>*>(chirp_);
        if (zlog_cap < max_steps + 1) {
            HIP_TRY(hipStreamSynchronize(stream));
            (void)hipFree(zlog); zlog = nullptr; zlog_cap = 0;
            HIP_TRY(hipMalloc(&zlog, sizeof(T) * (max_steps + 1)));
            zlog_cap = max_steps + 1;
        }
        if (int rc = upload_adapt_state((T)gamma_d, (T)length, (T)phi_max, (int)max_steps)) return rc;
        hipLaunchKernelGGL(k_absmax<T>, dim3(256), dim3(256), 0, stream, (const cx<T>*)A, (long long)nn * batch, st);
        hipLaunchKernelGGL(k_step_control<T>, dim3(1), dim3(64), 0, stream, st, zlog, 0, 0, 0);
        if (!medium_st) {
            HIP_TRY(hipMalloc(&medium_st, sizeof(unsigned long long) * 2 * (kBarShards + kBarWords + 2)));
            HIP_TRY(hipHostMalloc(&medium_err_host, 2 * sizeof(unsigned)));
            medium_err_host[0] = medium_err_host[1] = 0u;
        }
        HIP_TRY(hipMemsetAsync(medium_st, 0, sizeof(unsigned long long) * (kBarShards + kBarWords + 2), stream));
        MediumChirpAdaptArgs<T> a;
        std::memset(&a, 0, sizeof(a));
        a.F = F; a.Y = Y; a.P = P; a.twA = twA; a.twB = twB; a.tw1 = tw1; a.tw2 = tw2;
        a.H[0] = xfer_tab[0]; a.H[1] = xfer_tab[1]; a.Dt = static_cast<const cx<T>*>(Dt_); a.st = st; a.zlog = zlog;
        a.bar = medium_st; a.error = reinterpret_cast<unsigned*>(medium_st + kBarShards + kBarWords); a.patience = medium_patience;
        a.xcc = (unsigned)medium_xcc; a.nblk = (unsigned)blocks;
        a.gamma = (T)gamma_d; a.inv_len = (T)(1.0 / (double)nn); a.n = (int)nn; a.rows = batch; a.Qf = N2 / Ef;
        const unsigned gA = (unsigned)std::min<long long>((n * batch + 255) / 256, 4096);
        hipLaunchKernelGGL(k_cm_pre<T>, dim3(gA), dim3(256), 0, stream, (const cx<T>*)A, chirp, F, (long long)nn, (long long)n, batch);
        HIP_TRY(launch_medium_chirp_adapt(N1, N2, (int)a.nblk, medium_xccs, stream, a));
        // (the barriers report through the engine's error word, the hand-over of the maxima through the step state's: the field goes back to the caller
        // -- the post kernel -- only when both are clear and the run is done)
        unsigned gave_up[2] = {0u, 0u};
        StepState<T> fin = {};
        HIP_TRY(hipMemcpyAsync(&gave_up[0], a.error, sizeof(unsigned), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipMemcpyAsync(&gave_up[1], reinterpret_cast<const char*>(st) + offsetof(AdaptState<T>, error), sizeof(unsigned), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipMemcpyAsync(&fin, reinterpret_cast<const char*>(st) + offsetof(AdaptState<T>, cur), sizeof(fin), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        last_launches += 5;
        last_engine = SSFM_ENGINE_CHIRP_MEDIUM_ADAPT;
        last_fell_back = 0;
        if (gave_up[0] != 0u || gave_up[1] != 0u || !fin.done) {
            medium_adapt_ok = false;
            ++fallbacks;
            last_fell_back = 1;
            return fail(SSFM_ERR_UNSUPPORTED, "ssfm_chirp_medium_adapt: the launch's workgroups did not meet within the patience; the field is unchanged");
        }
        hipLaunchKernelGGL(k_cm_post<T>, dim3(gA), dim3(256), 0, stream, A, chirp, (const cx<T>*)F, (long long)nn, (long long)n, batch, (const unsigned*)a.error);
        HIP_TRY(hipGetLastError());
        if (z_out) {
            std::vector<T> zl((size_t)fin.steps + 1);
            HIP_TRY(hipMemcpyAsync(zl.data(), zlog, sizeof(T) * zl.size(), hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipStreamSynchronize(stream));
            for (size_t i = 0; i < zl.size(); ++i) z_out[i] = (double)zl[i];
        }
        HIP_TRY(hipStreamSynchronize(stream));
        if (steps_out) *steps_out = fin.steps;
        return SSFM_OK;
        }
    }
    // slot <- fft(field): the field itself becomes a resident transfer function (row 0; the field is consumed)
    int table_from_field(int slot) {
        if (int rc = no_split("ssfm_table_from_field")) return rc;
        if (slot < 0 || slot > 1) return fail(SSFM_ERR_INVALID, "ssfm_table_from_field: slot %d", slot);
        if (int rc = use_device()) return rc;
        if (!xfer_tab[slot]) HIP_TRY(hipMalloc(&xfer_tab[slot], sizeof(cx<T>) * n));
        tags[1 + slot] = 0;
        if (int rc = field_spectrum()) return rc;
        hipLaunchKernelGGL((k_make_freq_table<T, 1>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                           (const cx<T>*)scratch, xfer_tab[slot], N1, N2, N2 / Ef, (T)0, inv_n());
        HIP_TRY(hipGetLastError());
        ++last_launches;
        return SSFM_OK;
    }
    // scratch <- fft(field) of every row, natural frequency order (the field is consumed)
    int field_spectrum() {
        if (!scratch) HIP_TRY(hipMalloc(&scratch, sizeof(cx<T>) * n * batch));
        HIP_TRY((launch_time<T, TM_BEGIN>(N1, batch, stream, targs(0, 0, 0, nullptr), E)));
        HIP_TRY((launch_freq<T, FM_FWD_ONLY>(N2, N1 * batch, stream, fargs(dperm, 0, nullptr), Ef)));
        const long long total = (long long)n * batch;
        hipLaunchKernelGGL(k_unpermute<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (const cx<T>*)Y, scratch, N1, N2, batch);
        HIP_TRY(hipGetLastError());
        last_launches += 3;
        return SSFM_OK;
    }
    int apply_table(int slot) {
        if (int rc = no_split("ssfm_apply_table")) return rc;
        if (slot < 0 || slot > 1 || !xfer_tab[slot]) return fail(SSFM_ERR_STATE, "ssfm_apply_table: slot %d holds no table", slot);
        if (int rc = use_device()) return rc;
        HIP_TRY((launch_time<T, TM_BEGIN>(N1, batch, stream, targs(0, 0, 0, nullptr), E)));
        HIP_TRY((launch_freq<T, FM_TABLE>(N2, N1 * batch, stream, fargs(xfer_tab[slot], 0, nullptr), Ef)));
        HIP_TRY((launch_time<T, TM_END>(N1, batch, stream, targs(0, 0, 0, nullptr), E)));
        last_launches += 3;
        return SSFM_OK;
    }

    int debug_fft(void* dst) {
        if (int rc = no_split("ssfm_debug")) return rc;
        if (int rc = use_device()) return rc;
        const int nrows = N1 * batch;
        if (!scratch) HIP_TRY(hipMalloc(&scratch, sizeof(cx<T>) * n * batch));
        HIP_TRY((launch_time<T, TM_BEGIN>(N1, batch, stream, targs(0, 0, 0, nullptr), E)));
        HIP_TRY((launch_freq<T, FM_FWD_ONLY>(N2, nrows, stream, fargs(dperm, 0, nullptr), Ef)));
        const long long total = (long long)n * batch;
        hipLaunchKernelGGL(k_unpermute<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                           (const cx<T>*)Y, scratch, N1, N2, batch);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(dst, scratch, sizeof(cx<T>) * total, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        return SSFM_OK;
    }
};

}  // namespace

struct ssfm_plan {
    PlanBase* impl;
};

#define WITH_PLAN(p, expr)                                                          \
    do {                                                                            \
        if ((p) == nullptr || (p)->impl == nullptr) return fail(SSFM_ERR_INVALID, "null plan"); \
        if ((p)->impl->precision == SSFM_C64) { auto* P_ = static_cast<PlanT<float>*>((p)->impl); return (expr); } \
        auto* P_ = static_cast<PlanT<double>*>((p)->impl);                          \
        return (expr);                                                              \
    } while (0)

namespace {
template <typename PT> static int set_field_impl(PT* P_, const void* src, int is_device) {
    if (int rc = P_->use_device()) return rc;
    HIP_TRY(hipMemcpyAsync(P_->user_field(), src, sizeof(*P_->F) * P_->n * P_->batch, is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, P_->stream));
    if (P_->is_split()) { P_->nat_valid = true; P_->sub_valid = false; }
    if (!is_device) HIP_TRY(hipStreamSynchronize(P_->stream));
    return SSFM_OK;
}

template <typename PT> static int sync_impl(PT* P_) {
    if (int rc = P_->use_device()) return rc;
    HIP_TRY(hipStreamSynchronize(P_->stream));
    return P_->lane_health();
}

template <typename PT> static int last_ms_impl(PT* P_, float* ms, int64_t* launches) {
    if (launches) *launches = P_->last_launches;
    if (ms) {
        *ms = 0.f;
        if (P_->timed) {
            HIP_TRY(hipEventSynchronize(P_->ev1));
            HIP_TRY(hipEventElapsedTime(ms, P_->ev0, P_->ev1));
        }
    }
    return P_->lane_health();
}
}  // namespace

namespace {
template <typename PT> static int run_info_impl(PT* P_, ssfm_run_info* info, size_t bytes) {
    if (int rc = P_->use_device()) return rc;
    if (int rc = P_->lane_health()) return rc;
    ssfm_run_info r;
    std::memset(&r, 0, sizeof(r));
    r.engine = P_->last_engine; r.fell_back = P_->last_fell_back; r.fallbacks_total = P_->fallbacks;
    r.lanes = P_->lanes_active; r.lanes_configured = P_->nlanes; r.lanes_share_queue = P_->lanes_share_queue ? 1 : 0; r.lanes_remade = P_->lanes_remade; r.lanes_dropped = P_->lanes_dropped;
    r.lane_heals = P_->lane_heals;
    r.lane_alone_us = P_->lane_alone_us; r.lane_pair_us = P_->lane_pair_us; r.lane_last_us = P_->lane_last_us; r.lane_score = P_->lane_score;
    r.lanes_from_pool = P_->lanes_from_pool ? 1 : 0; r.lane_ratings_total = g_lane_ratings.load(); r.lane_pairs_reused = g_lane_pairs_reused.load();
    std::memcpy(info, &r, bytes < sizeof(r) ? bytes : sizeof(r));
    return SSFM_OK;
}
}  // namespace

namespace {
// A caller that takes the stream or the field pointer may order its own work behind a run without ssfm_synchronize -- also behind a run that is ALREADY
// queued: a one-launch run of a medium plan that is still pending is resolved here (waited for, checked, repeated on the two-kernel engine if its
// workgroups did not meet), so that what the caller orders on the stream finds the run's real result.  nullptr (ssfm_last_error set) if that fails.
template <typename PT> static bool mark_external(PT* P_) {
    P_->external_order = true;
    // (a capture run's helper thread copies the end field from F on a stream of its own: it is joined here, before the caller can order anything that
    // writes F behind the run -- ADVICE r5: the snapshots' validity "after any later call on the plan" includes these two calls)
    if (P_->is_split() && (P_->use_device() != SSFM_OK || P_->publish_if_external() != SSFM_OK)) return false;      // (the natural-order field, queued on the plan's stream; the caller may write it)
    if (!P_->medium_pending && !P_->cap_pending) return true;
    return P_->use_device() == SSFM_OK;               // (joins the capture helper, resolves a pending one-launch run)
}
// The library's own cross-unit accessors (chirpz.hip writes F on the plan's stream through them): the same join.
template <typename PT> static bool join_helpers(PT* P_) {
    if (!P_->medium_pending && !P_->cap_pending) return true;
    return P_->use_device() == SSFM_OK;
}
}  // namespace

extern "C" {

int ssfm_abi_version(void) { return SSFM_ABI_VERSION; }

const char* ssfm_last_error(void) { return g_err; }

int ssfm_device_count(int* count) {
    if (!count) return fail(SSFM_ERR_INVALID, "count is NULL");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { *count = 0; return fail(SSFM_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *count = c;
    return SSFM_OK;
}

int ssfm_supported_log2n(int precision, int* lo, int* hi) {
    if (precision != SSFM_C64 && precision != SSFM_C128) return fail(SSFM_ERR_UNSUPPORTED, "unknown precision %d", precision);
    if (lo) *lo = kLog2Min;
    if (hi) *hi = kLog2Max;
    return SSFM_OK;
}

int ssfm_plan_create(ssfm_plan** out, int device, int64_t n, int batch, int precision) {
    if (!out) return fail(SSFM_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (precision != SSFM_C64 && precision != SSFM_C128) return fail(SSFM_ERR_UNSUPPORTED, "unknown precision %d", precision);
    if (batch < 1 || batch > 65535) return fail(SSFM_ERR_INVALID, "batch=%d out of range [1, 65535]", batch);
    if (n < (1ll << kLog2Min) || n > (1ll << kLog2Max) || (n & (n - 1)) != 0)
        return fail(SSFM_ERR_UNSUPPORTED, "n=%lld: the HIP path needs a power of two in [2^%d, 2^%d]", (long long)n, kLog2Min, kLog2Max);
    int count = 0;
    if (int rc = ssfm_device_count(&count)) return rc;
    if (device < 0 || device >= count) return fail(SSFM_ERR_NO_DEVICE, "device %d not available (%d visible)", device, count);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(SSFM_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
    ssfm_plan* p = new (std::nothrow) ssfm_plan{nullptr};
    if (!p) return fail(SSFM_ERR_INVALID, "out of host memory");
    int rc;
    if (precision == SSFM_C64) {
        auto* impl = new (std::nothrow) PlanT<float>();
        if (!impl) { delete p; return fail(SSFM_ERR_INVALID, "out of host memory"); }
        impl->precision = precision;
        p->impl = impl;
        rc = impl->init(device, n, batch);
    } else {
        auto* impl = new (std::nothrow) PlanT<double>();
        if (!impl) { delete p; return fail(SSFM_ERR_INVALID, "out of host memory"); }
        impl->precision = precision;
        p->impl = impl;
        rc = impl->init(device, n, batch);
    }
    if (rc != SSFM_OK) { delete p->impl; delete p; return rc; }
    *out = p;
    return SSFM_OK;
}

int ssfm_plan_destroy(ssfm_plan* plan) {
    if (!plan) return SSFM_OK;
    delete plan->impl;
    delete plan;
    return SSFM_OK;
}

int ssfm_set_linear_operator(ssfm_plan* plan, const void* dtilde_host) {
    if (!dtilde_host) return fail(SSFM_ERR_INVALID, "dtilde_host is NULL");
    WITH_PLAN(plan, P_->set_operator(dtilde_host));
}

int ssfm_set_field(ssfm_plan* plan, const void* src, int is_device) {
    if (!src) return fail(SSFM_ERR_INVALID, "src is NULL");
    WITH_PLAN(plan, set_field_impl(P_, src, is_device));
}
int ssfm_get_field(ssfm_plan* plan, void* dst, int is_device) {
    if (!dst) return fail(SSFM_ERR_INVALID, "dst is NULL");
    WITH_PLAN(plan, (P_->use_device() ? SSFM_ERR_HIP : P_->copy_field_out(dst, is_device != 0, true)));
}
void* ssfm_field_device_ptr(ssfm_plan* plan) {
    if (!plan || !plan->impl) return nullptr;
    if (plan->impl->precision == SSFM_C64) { auto* P_ = static_cast<PlanT<float>*>(plan->impl); return mark_external(P_) ? P_->user_field() : nullptr; }
    auto* P_ = static_cast<PlanT<double>*>(plan->impl);
    return mark_external(P_) ? P_->user_field() : nullptr;
}

int ssfm_propagate_fixed(ssfm_plan* plan, double gamma, const void* h_schedule, int64_t nsteps, void* snapshots) {
    if (nsteps < 0) return fail(SSFM_ERR_INVALID, "nsteps=%lld", (long long)nsteps);
    if (nsteps > 0 && !h_schedule) return fail(SSFM_ERR_INVALID, "h_schedule is NULL");
    if (!plan || !plan->impl) return fail(SSFM_ERR_INVALID, "null plan");
    if (plan->impl->precision == SSFM_C64)
        return static_cast<PlanT<float>*>(plan->impl)->propagate_fixed(gamma, static_cast<const float*>(h_schedule), nsteps, snapshots);
    return static_cast<PlanT<double>*>(plan->impl)->propagate_fixed(gamma, static_cast<const double*>(h_schedule), nsteps, snapshots);
}

int ssfm_propagate_fixed_capture(ssfm_plan* plan, double gamma, const void* h_schedule, int64_t nsteps, const ssfm_capture* cap) {
    if (!h_schedule || !cap) return fail(SSFM_ERR_INVALID, "ssfm_propagate_fixed_capture: NULL argument");
    if (!plan || !plan->impl) return fail(SSFM_ERR_INVALID, "null plan");
    if (plan->impl->precision == SSFM_C64)
        return static_cast<PlanT<float>*>(plan->impl)->propagate_fixed_capture(gamma, static_cast<const float*>(h_schedule), nsteps, cap->every, cap->fields, cap->scalars);
    return static_cast<PlanT<double>*>(plan->impl)->propagate_fixed_capture(gamma, static_cast<const double*>(h_schedule), nsteps, cap->every, cap->fields, cap->scalars);
}
int ssfm_propagate_adaptive(ssfm_plan* plan, double gamma, double length, double phi_max, int single_step,
                            int64_t max_steps, int64_t* steps_out, double* z_out, void* snapshots) {
    WITH_PLAN(plan, P_->propagate_adaptive(gamma, length, phi_max, single_step, max_steps, steps_out, z_out, snapshots));
}

int ssfm_adaptive_begin(ssfm_plan* plan, double gamma, double length, double phi_max, int single_step, int64_t max_steps, int capture) {
    WITH_PLAN(plan, P_->adaptive_begin(gamma, length, phi_max, single_step, max_steps, capture));
}
int ssfm_adaptive_run(ssfm_plan* plan, int64_t budget, void* snapshots, int64_t* steps_total, int* done) {
    WITH_PLAN(plan, P_->adaptive_run(budget, snapshots, steps_total, done));
}
int ssfm_adaptive_finish(ssfm_plan* plan, int64_t* steps_out, double* z_out) { WITH_PLAN(plan, P_->adaptive_finish(steps_out, z_out)); }
int ssfm_adaptive_set_capture(ssfm_plan* plan, const ssfm_adaptive_capture* cap) {
    if (!cap) return fail(SSFM_ERR_INVALID, "ssfm_adaptive_set_capture: NULL description");
    WITH_PLAN(plan, P_->adaptive_set_capture(cap->every, cap->steps, cap->n_steps, cap->fields, cap->capacity, cap->taken, cap->n_taken));
}

int ssfm_apply_transfer(ssfm_plan* plan, const void* H_host) {
    if (!H_host) return fail(SSFM_ERR_INVALID, "H_host is NULL");
    WITH_PLAN(plan, P_->apply_transfer(H_host));
}

int ssfm_apply_dispersion(ssfm_plan* plan, double dt_s, double D_s2, void* H_out) {
    if (!(dt_s > 0)) return fail(SSFM_ERR_INVALID, "dt_s must be positive");
    WITH_PLAN(plan, P_->apply_dispersion(dt_s, D_s2, H_out));
}

int ssfm_transfer_table(ssfm_plan* plan, const void* H_host, int slot) {
    if (!H_host) return fail(SSFM_ERR_INVALID, "H_host is NULL");
    WITH_PLAN(plan, P_->transfer_table(H_host, slot));
}
int ssfm_apply_table(ssfm_plan* plan, int slot) { WITH_PLAN(plan, P_->apply_table(slot)); }
int ssfm_table_from_field(ssfm_plan* plan, int slot) { WITH_PLAN(plan, P_->table_from_field(slot)); }
int ssfm_synchronize(ssfm_plan* plan) { WITH_PLAN(plan, sync_impl(P_)); }

void* ssfm_stream(ssfm_plan* plan) {
    if (!plan || !plan->impl) return nullptr;
    if (plan->impl->precision == SSFM_C64) { auto* P_ = static_cast<PlanT<float>*>(plan->impl); return mark_external(P_) ? P_->stream : nullptr; }
    auto* P_ = static_cast<PlanT<double>*>(plan->impl);
    return mark_external(P_) ? P_->stream : nullptr;
}


int ssfm_debug(ssfm_plan* plan, int what, int64_t arg, void* dst) {
    if (what == SSFM_DEBUG_FFT) {
        if (!dst) return fail(SSFM_ERR_INVALID, "ssfm_debug: dst is NULL");
        WITH_PLAN(plan, P_->debug_fft(dst));
    }
    if (what == SSFM_DEBUG_LANE_FAULT) {
        const int mode = (int)arg;
        WITH_PLAN(plan, (P_->lane_fault = mode, P_->lane_pair_us = mode ? P_->lane_pair_us * 0.25f : P_->lane_pair_us, (int)SSFM_OK));
    }
    return fail(SSFM_ERR_INVALID, "ssfm_debug: what = %d", what);
}
int ssfm_last_run_info(ssfm_plan* plan, ssfm_run_info* info, size_t info_bytes) {
    if (!info || info_bytes < 2 * sizeof(int)) return fail(SSFM_ERR_INVALID, "ssfm_last_run_info: info is NULL or too small");
    WITH_PLAN(plan, run_info_impl(P_, info, info_bytes));
}

int ssfm_last_propagate_ms(ssfm_plan* plan, float* ms, int64_t* launches) { WITH_PLAN(plan, last_ms_impl(P_, ms, launches)); }

int ssfm_set_profiling(ssfm_plan* plan, int mode) { WITH_PLAN(plan, (P_->profiling = mode != 0, P_->prof_mode = mode, (int)SSFM_OK)); }
int ssfm_plan_set_tag(ssfm_plan* plan, int which, uint64_t tag) {
    if (which < 0 || which > 2) return fail(SSFM_ERR_INVALID, "ssfm_plan_set_tag: which = %d", which);
    WITH_PLAN(plan, (P_->tags[which] = tag, (int)SSFM_OK));
}
int ssfm_plan_get_tag(ssfm_plan* plan, int which, uint64_t* tag) {
    if (which < 0 || which > 2 || !tag) return fail(SSFM_ERR_INVALID, "ssfm_plan_get_tag: bad argument");
    WITH_PLAN(plan, (*tag = (which == 0 && !P_->have_op) ? 0 : P_->tags[which], (int)SSFM_OK));
}
int ssfm_kernel_times(ssfm_plan* plan, int64_t counts[2], double total_ms[2]) {
    if (!counts || !total_ms) return fail(SSFM_ERR_INVALID, "NULL output");
    WITH_PLAN(plan, P_->kernel_times(counts, total_ms));
}

}  // extern "C"

// ---- the library's own cross-unit entry points (ssfm_common.hpp): not part of the C ABI
namespace ssfm {
void* plan_stream(ssfm_plan* plan) {
    if (!plan || !plan->impl) return nullptr;
    if (plan->impl->precision == SSFM_C64) { auto* P_ = static_cast<PlanT<float>*>(plan->impl); return join_helpers(P_) ? P_->stream : nullptr; }
    auto* P_ = static_cast<PlanT<double>*>(plan->impl);
    return join_helpers(P_) ? P_->stream : nullptr;
}
void* plan_field(ssfm_plan* plan) {
    if (!plan || !plan->impl) return nullptr;
    // (the library's other units work on the field buffer as a line of `plan_length` points: not for split plans, whose F holds sub-sequences)
    if (plan->impl->precision == SSFM_C64) { auto* P_ = static_cast<PlanT<float>*>(plan->impl); return join_helpers(P_) && !P_->is_split() ? (void*)P_->F : nullptr; }
    auto* P_ = static_cast<PlanT<double>*>(plan->impl);
    return join_helpers(P_) && !P_->is_split() ? (void*)P_->F : nullptr;
}
int64_t plan_length(ssfm_plan* plan, int* batch, int* precision) {
    if (!plan || !plan->impl) return 0;
    if (precision) *precision = plan->impl->precision;
    if (plan->impl->precision == SSFM_C64) { auto* P_ = static_cast<PlanT<float>*>(plan->impl); if (batch) *batch = P_->batch_full; return P_->n_full; }
    auto* P_ = static_cast<PlanT<double>*>(plan->impl);
    if (batch) *batch = P_->batch_full;
    return P_->n_full;
}
int plan_workspace(ssfm_plan* plan, int slot, size_t bytes, void** out) { WITH_PLAN(plan, P_->workspace(slot, bytes, out)); }
int plan_chirp_step(ssfm_plan* plan, const void* mul_dev, const ChirpStepIO* io) {
    if (!io) return fail(SSFM_ERR_INVALID, "chirp step: NULL field description");
    WITH_PLAN(plan, P_->apply_tables_mul(mul_dev, io));
}
int plan_chirp_line_run(ssfm_plan* plan, const void* const* mul, const unsigned char* which, const double* hs, int64_t nsteps, double gamma, int64_t keep, int half) {
    WITH_PLAN(plan, P_->chirp_line_run(mul, which, hs, nsteps, gamma, keep, half));
}
int plan_line_half_ok(ssfm_plan* plan) {
    if (!plan || !plan->impl || plan->impl->precision == SSFM_C64) return 0;
    return static_cast<PlanT<double>*>(plan->impl)->line_half_ok() ? 1 : 0;
}
int plan_chirp_small(ssfm_plan* plan, void* A, const void* chirp, const void* Dt, int64_t n, double gamma, const double* hs, int64_t nsteps) {
    WITH_PLAN(plan, P_->chirp_small(A, chirp, Dt, n, gamma, hs, nsteps));
}
int plan_chirp_small_adapt(ssfm_plan* plan, void* A, const void* chirp, const void* Dt, int64_t n, double gamma, double length, double phi_max, int f32,
                           int64_t max_steps, double* z_out, int64_t* steps_out) {
    WITH_PLAN(plan, P_->chirp_small_adapt(A, chirp, Dt, n, gamma, length, phi_max, f32, max_steps, z_out, steps_out));
}
int plan_chirp_medium(ssfm_plan* plan, void* A, const void* chirp, const void* Dt, int64_t n, double gamma, const double* hs, int64_t nsteps) {
    WITH_PLAN(plan, P_->chirp_medium(A, chirp, Dt, n, gamma, hs, nsteps));
}
int plan_chirp_medium_adapt(ssfm_plan* plan, void* A, const void* chirp, const void* Dt, int64_t n, double gamma, double length, double phi_max, int64_t max_steps,
                            double* z_out, int64_t* steps_out) {
    WITH_PLAN(plan, P_->chirp_medium_adapt(A, chirp, Dt, n, gamma, length, phi_max, max_steps, z_out, steps_out));
}
}  // namespace ssfm
