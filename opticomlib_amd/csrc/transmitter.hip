// transmitter.hip -- the elementwise device work of the transmitter chain, so that PRBS -> DAC -> MZM(LASER) -> FIBER
// stays in GPU memory: the Mach-Zehnder transfer (reference devices.py:762-767)
//     g = pi/(2 Vpi) (v + bias) [+ pi/(2 Vpi) v_noise],   h = sqrt(loss) (cos g + j eta/2 sin g),   out = in * h
// for the optical signal and the optical noise alike, with the unused polarisation emptied (:771-778), and the two
// array operations DAC needs after its convolution (devices.py:319-336): a x + b, and the real part of a complex array.
#include <hip/hip_runtime.h>

#include "ssfm_amd.h"
#include "ssfm_common.hpp"

using ssfm::fail;

namespace {

int use_dev(int device) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return fail(SSFM_ERR_NO_DEVICE, "device %d not available", device);
    HIP_TRY(hipSetDevice(device));
    return SSFM_OK;
}
unsigned blocks_of(long long n) { return (unsigned)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192); }

// drive: float64 (drive_complex = 0) or complex128 (1: a chirped / Gaussian DAC output); noise of the drive likewise
__global__ __launch_bounds__(256) void k_mzm(double2* __restrict__ out_s, double2* __restrict__ out_n, const double2* __restrict__ in_s, const double2* __restrict__ in_n,
                                             int n_pol, long long n, const double* __restrict__ v, const double* __restrict__ vn, int drive_complex, double k,
                                             double bias, double sqrt_loss, double half_eta, int dead_pol) {
#pragma clang fp contract(off)
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        double gr, gi = 0.0;
        if (drive_complex) {
            gr = k * (v[2 * i] + bias);
            gi = k * v[2 * i + 1];
            if (vn) { gr += k * vn[2 * i]; gi += k * vn[2 * i + 1]; }
        } else {
            gr = k * (v[i] + bias);
            if (vn) gr += k * vn[i];
        }
        double s, c;
        sincos(gr, &s, &c);
        double cr = c, ci = 0.0, sr = s, si = 0.0;              // cos g, sin g (complex for a complex drive)
        if (drive_complex && gi != 0.0) {
            const double ch = cosh(gi), sh = sinh(gi);
            cr = c * ch; ci = -s * sh;
            sr = s * ch; si = c * sh;
        }
        // h = sqrt_loss * (cos g + j half_eta sin g)
        const double hr = sqrt_loss * (cr - half_eta * si), hi = sqrt_loss * (ci + half_eta * sr);
        for (int p = 0; p < n_pol; ++p) {
            const long long o = (long long)p * n + i;
            const bool dead = n_pol == 2 && p == dead_pol;
            const double2 a = in_s[o];
            out_s[o] = dead ? make_double2(0.0, 0.0) : make_double2(a.x * hr - a.y * hi, a.x * hi + a.y * hr);
            if (in_n) {
                const double2 b = in_n[o];
                out_n[o] = dead ? make_double2(0.0, 0.0) : make_double2(b.x * hr - b.y * hi, b.x * hi + b.y * hr);
            }
        }
    }
}

// LASER (reference devices.py:353-510) over t = linspace(0, stop, n): out = amp [exp(j phase)] [sqrt(1 + rin)] [exp(j w t)],
// each factor applied in the reference's order and only when present; real output (n doubles) while no phase factor
// has been applied, complex otherwise.  `phase` is the running sum of the Wiener increments and `rin` the intensity
// noise, both drawn by the caller (NumPy's generator, so that a seeded script gets the reference's realisation).
__global__ __launch_bounds__(256) void k_laser(double* __restrict__ out, long long n, double amp, const double* __restrict__ phase, const double* __restrict__ rin,
                                               int has_df, double w, double step, double stop) {
#pragma clang fp contract(off)
    const bool cplx = phase || has_df;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        double re = amp, im = 0.0;
        if (phase) {
            double s, c;
            sincos(phase[i], &s, &c);
            re = amp * c; im = amp * s;
        }
        if (rin) {
            const double q = sqrt(1.0 + rin[i]);
            re = re * q; im = im * q;
        }
        if (has_df) {
            const double t = (i == n - 1 && n > 1) ? stop : (double)i * step;
            double s, c;
            sincos(w * t, &s, &c);
            const double r2 = re * c - im * s, i2 = re * s + im * c;
            re = r2; im = i2;
        }
        if (cplx) { out[2 * i] = re; out[2 * i + 1] = im; }
        else out[i] = re;
    }
}

// dst = src * alpha + beta; complex: both parts scaled, beta added to the real part
__global__ __launch_bounds__(256) void k_axpb(double* __restrict__ dst, const double* __restrict__ src, double alpha, double beta, long long n, int is_complex) {
#pragma clang fp contract(off)
    const long long total = is_complex ? 2 * n : n;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
        dst[i] = (is_complex && (i & 1)) ? src[i] * alpha : src[i] * alpha + beta;
}

__global__ __launch_bounds__(256) void k_real(double* __restrict__ dst, const double2* __restrict__ src, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) dst[i] = src[i].x;
}

}  // namespace

extern "C" int ssfm_mzm(int device, void* out_sig, void* out_noise, const void* in_sig, const void* in_noise, int n_pol, int64_t n, const void* drive,
                        const void* drive_noise, int drive_complex, double k, double bias, double sqrt_loss, double half_eta, int dead_pol) {
    if (!out_sig || !in_sig || !drive || n < 1 || n_pol < 1 || n_pol > 2) return fail(SSFM_ERR_INVALID, "ssfm_mzm: bad argument");
    if ((in_noise == nullptr) != (out_noise == nullptr)) return fail(SSFM_ERR_INVALID, "ssfm_mzm: in_noise and out_noise must be given together");
    if (int rc = use_dev(device)) return rc;
    hipLaunchKernelGGL(k_mzm, dim3(blocks_of(n)), dim3(256), 0, 0, (double2*)out_sig, (double2*)out_noise, (const double2*)in_sig, (const double2*)in_noise, n_pol,
                       (long long)n, (const double*)drive, (const double*)drive_noise, drive_complex, k, bias, sqrt_loss, half_eta, dead_pol);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    return SSFM_OK;
}

extern "C" int ssfm_device_axpb(int device, void* dst, const void* src, double alpha, double beta, int64_t n, int is_complex) {
    if (!dst || !src || n < 1) return fail(SSFM_ERR_INVALID, "ssfm_device_axpb: bad argument");
    if (int rc = use_dev(device)) return rc;
    hipLaunchKernelGGL(k_axpb, dim3(blocks_of(n)), dim3(256), 0, 0, (double*)dst, (const double*)src, alpha, beta, (long long)n, is_complex);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    return SSFM_OK;
}

extern "C" int ssfm_laser(int device, void* out, int64_t n, double amp, const double* phase, const double* rin, int has_df, double w, double step, double stop) {
    if (!out || n < 1) return fail(SSFM_ERR_INVALID, "ssfm_laser: bad argument");
    if (int rc = use_dev(device)) return rc;
    hipLaunchKernelGGL(k_laser, dim3(blocks_of(n)), dim3(256), 0, 0, (double*)out, (long long)n, amp, phase, rin, has_df, w, step, stop);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    return SSFM_OK;
}

// ---------------------------------------------------------------------------------- small array helpers
// ssfm_device_mean2: mean of a float64 array (out[0]) or of the real and imaginary parts of a complex128 one
// (out[0], out[1]); ssfm_device_shift: dst = src + (re + j im) (AC coupling of a complex pulse, devices.py:339-340);
// ssfm_device_zero: `bytes` zero bytes (an empty polarisation, devices.py:924); ssfm_device_power: mean |x|^2 of
// each of `rows` rows of n complex128 (float64: x^2) values.
namespace {

__global__ __launch_bounds__(256) void k_sum_strided(const double* __restrict__ a, long long n, int stride, int offset, int square, double* __restrict__ partial) {
    double acc = 0.0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        if (square) {
            double v = 0.0;
            for (int k = 0; k < stride; ++k) { const double x = a[i * stride + k]; v += x * x; }
            acc += v;
        } else {
            acc += a[i * stride + offset];
        }
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    __shared__ double w[4];
    if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = w[0] + w[1] + w[2] + w[3];
}

__global__ __launch_bounds__(256) void k_shift(double* __restrict__ dst, const double* __restrict__ src, long long n, int is_complex, double re, double im) {
    const long long total = is_complex ? 2 * n : n;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
        dst[i] = src[i] + ((is_complex && (i & 1)) ? im : re);
}

int reduce_sum(int device, const double* a, long long n, int stride, int offset, int square, double* sum_out) {
    constexpr int kBlocks = 512;
    double* partial = nullptr;
    if (int rc = ssfm_device_alloc(device, sizeof(double) * kBlocks, (void**)&partial)) return rc;
    hipLaunchKernelGGL(k_sum_strided, dim3(kBlocks), dim3(256), 0, 0, a, n, stride, offset, square, partial);
    double host[kBlocks];
    hipError_t e = hipMemcpy(host, partial, sizeof(host), hipMemcpyDeviceToHost);
    (void)ssfm_device_free(device, partial, sizeof(double) * kBlocks);
    if (e != hipSuccess) return fail(SSFM_ERR_HIP, "device reduction: %s", hipGetErrorString(e));
    double acc = 0.0;
    for (int i = 0; i < kBlocks; ++i) acc += host[i];
    *sum_out = acc;
    return SSFM_OK;
}

}  // namespace

namespace {
int device_mean2(int device, const void* src, int64_t n, int is_complex, double* out) {
    if (!src || !out || n < 1) return fail(SSFM_ERR_INVALID, "ssfm_device_reduce (mean2): bad argument");
    if (int rc = use_dev(device)) return rc;
    const int stride = is_complex ? 2 : 1;
    for (int k = 0; k < stride; ++k) {
        double s = 0.0;
        if (int rc = reduce_sum(device, (const double*)src, (long long)n, stride, k, 0, &s)) return rc;
        out[k] = s / (double)n;
    }
    return SSFM_OK;
}

int device_power(int device, const void* src, int rows, int64_t n, int is_complex, double* out) {
    if (!src || !out || n < 1 || rows < 1) return fail(SSFM_ERR_INVALID, "ssfm_device_reduce (power): bad argument");
    if (int rc = use_dev(device)) return rc;
    const int stride = is_complex ? 2 : 1;
    for (int r = 0; r < rows; ++r) {
        double s = 0.0;
        if (int rc = reduce_sum(device, (const double*)src + (size_t)r * n * stride, (long long)n, stride, 0, 1, &s)) return rc;
        out[r] = s / (double)n;
    }
    return SSFM_OK;
}

}  // namespace
namespace ssfm { SSFM_INTERNAL int device_mean(int device, const double* a, const double* b, int64_t n, double* mean_out); SSFM_INTERNAL int device_min(int device, const double* a, int64_t n, double* min_out); }
// One entry point for the reductions a device-resident signal needs (all synchronous, results on the HOST):
//   SSFM_REDUCE_MEAN   out[0] = mean of the n float64 of a (+ b, nullable: the mean of the elementwise sum)
//   SSFM_REDUCE_MEAN2  numpy.mean of n float64 (out[0]) or complex128 values (out[0] + j out[1]) of a
//   SSFM_REDUCE_POWER  out[r] = mean |x|^2 of each of `rows` rows of n float64 / complex128 values of a
//   SSFM_REDUCE_MIN    out[0] = minimum of the n float64 of a
extern "C" int ssfm_device_reduce(int device, int kind, const void* a, const void* b, int rows, int64_t n, int is_complex, double* out) {
    switch (kind) {
        case SSFM_REDUCE_MEAN:  return ssfm::device_mean(device, (const double*)a, (const double*)b, n, out);
        case SSFM_REDUCE_MEAN2: return device_mean2(device, a, n, is_complex, out);
        case SSFM_REDUCE_POWER: return device_power(device, a, rows, n, is_complex, out);
        case SSFM_REDUCE_MIN:   return ssfm::device_min(device, (const double*)a, n, out);
    }
    return fail(SSFM_ERR_INVALID, "ssfm_device_reduce: kind %d", kind);
}
extern "C" int ssfm_device_shift(int device, void* dst, const void* src, int64_t n, int is_complex, double re, double im) {
    if (!dst || !src || n < 1) return fail(SSFM_ERR_INVALID, "ssfm_device_shift: bad argument");
    if (int rc = use_dev(device)) return rc;
    hipLaunchKernelGGL(k_shift, dim3(blocks_of(n)), dim3(256), 0, 0, (double*)dst, (const double*)src, (long long)n, is_complex, re, im);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    return SSFM_OK;
}

