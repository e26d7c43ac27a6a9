// frontend.hip -- receiver front-end arithmetic on gfx950: the square-law detection at the heart of the
// reference's PD (devices.py:1512-1515)
//     i_ph = r * (x * x.conj()).real,  summed over the polarisations  (then `post`, the load resistance
//     of devices.py:1547, applied to the sum),
// with the reference's signal / noise bookkeeping (typing.py:1337-1344): the signal current is r |s|^2,
// the noise current r Re(s n* + n s* + n n*) -- the beat terms kept separate from the signal so that the
// detector's low-pass filter (ssfm_sosfiltfilt) runs on both.  One pass, 16-byte loads, float64.
#include <hip/hip_runtime.h>

#include <mutex>

#include "ssfm_amd.h"
#include "ssfm_common.hpp"

using ssfm::fail;

namespace {

// the sums are formed in the reference's order without fused multiply-adds
template <bool NOISE>
__global__ __launch_bounds__(256) void k_square_law(const double2* __restrict__ s, const double2* __restrict__ nz, int n_pol, long long n, double r, double post,
                                                    double* __restrict__ i_sig, double* __restrict__ i_noise) {
#pragma clang fp contract(off)
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        double acc_s = 0.0, acc_n = 0.0;
        for (int p = 0; p < n_pol; ++p) {
            const double2 a = s[(long long)p * n + i];
            const double ps = (a.x * a.x + a.y * a.y) * r;
            acc_s = p ? acc_s + ps : ps;
            if (NOISE) {
                const double2 b = nz[(long long)p * n + i];
                const double m = a.x * b.x + a.y * b.y;                         // Re(s n*) = Re(n s*)
                const double pn = ((m + m) + (b.x * b.x + b.y * b.y)) * r;
                acc_n = p ? acc_n + pn : pn;
            }
        }
        i_sig[i] = acc_s * post;
        if (NOISE) i_noise[i] = acc_n * post;
    }
}

struct Scratch {
    std::mutex mu;
    void* buf[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t cap[4] = {0, 0, 0, 0};
    hipError_t need(int i, size_t bytes) {
        if (bytes <= cap[i]) return hipSuccess;
        if (buf[i]) { (void)hipFree(buf[i]); buf[i] = nullptr; cap[i] = 0; }
        hipError_t e = hipMalloc(&buf[i], bytes);
        if (e == hipSuccess) cap[i] = bytes;
        return e;
    }
};
constexpr int kMaxDevices = 64;
Scratch g_scratch[kMaxDevices];

}  // namespace

namespace {
int square_law_device(int device, const void* sig, const void* noise, int n_pol, int64_t n, double r, double post, double* i_sig, double* i_noise) {
    if (!sig || !i_sig) return fail(SSFM_ERR_INVALID, "ssfm_square_law: NULL argument");
    if ((noise == nullptr) != (i_noise == nullptr)) return fail(SSFM_ERR_INVALID, "ssfm_square_law: noise and i_noise must be given together");
    if (n_pol < 1 || n_pol > 2 || n < 1) return fail(SSFM_ERR_INVALID, "ssfm_square_law: n_pol=%d n=%lld", n_pol, (long long)n);
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count)
        return fail(SSFM_ERR_NO_DEVICE, "ssfm_square_law: device %d not available", device);
    HIP_TRY(hipSetDevice(device));
    const unsigned blocks = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    if (noise)
        hipLaunchKernelGGL(k_square_law<true>, dim3(blocks), dim3(256), 0, 0, (const double2*)sig, (const double2*)noise, n_pol, (long long)n, r, post, i_sig, i_noise);
    else
        hipLaunchKernelGGL(k_square_law<false>, dim3(blocks), dim3(256), 0, 0, (const double2*)sig, (const double2*)nullptr, n_pol, (long long)n, r, post, i_sig,
                           (double*)nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    return SSFM_OK;
}

}  // namespace
extern "C" int ssfm_square_law(int device, const void* sig, const void* noise, int n_pol, int64_t n, double r, double post, double* i_sig, double* i_noise, int on_device) {
    if (on_device) return square_law_device(device, sig, noise, n_pol, n, r, post, i_sig, i_noise);
    if (!sig || !i_sig) return fail(SSFM_ERR_INVALID, "ssfm_square_law: NULL argument");
    if ((noise == nullptr) != (i_noise == nullptr)) return fail(SSFM_ERR_INVALID, "ssfm_square_law: noise and i_noise must be given together");
    if (n_pol < 1 || n_pol > 2 || n < 1) return fail(SSFM_ERR_INVALID, "ssfm_square_law: n_pol=%d n=%lld", n_pol, (long long)n);
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count || device >= kMaxDevices)
        return fail(SSFM_ERR_NO_DEVICE, "ssfm_square_law: device %d not available", device);
    HIP_TRY(hipSetDevice(device));
    Scratch& w = g_scratch[device];
    std::lock_guard<std::mutex> lock(w.mu);
    const size_t in_bytes = sizeof(double2) * (size_t)n * n_pol, out_bytes = sizeof(double) * (size_t)n;
    HIP_TRY(w.need(0, in_bytes));
    HIP_TRY(w.need(2, out_bytes));
    HIP_TRY(hipMemcpy(w.buf[0], sig, in_bytes, hipMemcpyHostToDevice));
    if (noise) {
        HIP_TRY(w.need(1, in_bytes));
        HIP_TRY(w.need(3, out_bytes));
        HIP_TRY(hipMemcpy(w.buf[1], noise, in_bytes, hipMemcpyHostToDevice));
    }
    const unsigned blocks = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    if (noise)
        hipLaunchKernelGGL(k_square_law<true>, dim3(blocks), dim3(256), 0, 0, (const double2*)w.buf[0], (const double2*)w.buf[1], n_pol, (long long)n, r, post,
                           (double*)w.buf[2], (double*)w.buf[3]);
    else
        hipLaunchKernelGGL(k_square_law<false>, dim3(blocks), dim3(256), 0, 0, (const double2*)w.buf[0], (const double2*)nullptr, n_pol, (long long)n, r, post,
                           (double*)w.buf[2], (double*)nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(i_sig, w.buf[2], out_bytes, hipMemcpyDeviceToHost));
    if (noise) HIP_TRY(hipMemcpy(i_noise, w.buf[3], out_bytes, hipMemcpyDeviceToHost));
    return SSFM_OK;
}
