// device_mem.hip -- device-resident signals: raw HBM buffers a host-side signal object can own between
// calls, so that a chain FIBER -> DBP -> BPF -> PD moves no data over PCIe until a result is looked at.
// The reference has no counterpart (its arrays are NumPy arrays); these are the plumbing behind the host
// mirror's lazily materialised `.signal` / `.noise`.
#include <hip/hip_runtime.h>

#include <mutex>
#include <unordered_map>
#include <vector>

#include "ssfm_amd.h"
#include "ssfm_common.hpp"

using ssfm::fail;

namespace {

// Freed buffers are kept per (device, size) and handed out again: link simulations allocate the same few
// sizes over and over, and hipMalloc/hipFree cost more than the kernels of a short filter call.
struct Pool {
    std::mutex mu;
    std::unordered_map<size_t, std::vector<void*>> free_by_size;
    size_t cached_bytes = 0;
};
constexpr int kMaxDevices = 64;
constexpr size_t kMaxCachedBytes = size_t(2) << 30;        // per device
Pool g_pool[kMaxDevices];
Pool g_host_pool;                                          // page-locked host buffers (ssfm_host_alloc)
constexpr size_t kMaxCachedHostBytes = size_t(1) << 30;

int use(int device) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count || device >= kMaxDevices)
        return fail(SSFM_ERR_NO_DEVICE, "device %d not available", device);
    HIP_TRY(hipSetDevice(device));
    return SSFM_OK;
}

template <typename S, typename D>
__global__ __launch_bounds__(256) void k_convert(const S* __restrict__ src, D* __restrict__ dst, long long count2) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count2; i += (long long)gridDim.x * blockDim.x) dst[i] = (D)src[i];
}
template <typename D>
__global__ __launch_bounds__(256) void k_widen(const double* __restrict__ src, D* __restrict__ dst, long long count) {      // real -> complex
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long long)gridDim.x * blockDim.x) {
        dst[2 * i] = (D)src[i];
        dst[2 * i + 1] = (D)0;
    }
}
template <typename T>
__global__ __launch_bounds__(256) void k_add(T* __restrict__ dst, const T* __restrict__ a, const T* __restrict__ b, long long count2) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count2; i += (long long)gridDim.x * blockDim.x) dst[i] = a[i] + b[i];
}
unsigned blocks_for(long long n) { return (unsigned)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192); }


// Running sum (numpy.cumsum) of n doubles in three launches: every workgroup scans a tile of 256 x 16 samples and
// leaves its total; one workgroup turns the totals into offsets; the offsets are added.  A thread sums its 16
// samples in order, the tile's 256 partial sums are scanned through LDS (Hillis-Steele), so the additions are grouped
// differently from NumPy's left-to-right loop: equal to rounding (~1e-16 of the largest partial sum per level).
constexpr int kScanItems = 16;
constexpr int kScanTile = 256 * kScanItems;

__device__ __forceinline__ double block_exclusive_scan(double v, double* lds, double* total) {
    const int t = threadIdx.x;
    lds[t] = v;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        const double add = t >= o ? lds[t - o] : 0.0;
        __syncthreads();
        lds[t] += add;
        __syncthreads();
    }
    if (total) *total = lds[255];
    return lds[t] - v;
}

__global__ __launch_bounds__(256) void k_cumsum_tiles(const double* __restrict__ src, double* __restrict__ dst, long long n, double* __restrict__ totals) {
    __shared__ double lds[256];
    const long long base = (long long)blockIdx.x * kScanTile + (long long)threadIdx.x * kScanItems;
    double x[kScanItems], run = 0.0;
    for (int k = 0; k < kScanItems; ++k) {
        x[k] = base + k < n ? src[base + k] : 0.0;
        run += x[k];
        x[k] = run;
    }
    double total;
    const double before = block_exclusive_scan(run, lds, &total);
    for (int k = 0; k < kScanItems; ++k)
        if (base + k < n) dst[base + k] = before + x[k];
    if (threadIdx.x == 0) totals[blockIdx.x] = total;
}

__global__ __launch_bounds__(256) void k_cumsum_totals(double* __restrict__ totals, int ntiles) {      // one workgroup: exclusive scan in place
    __shared__ double lds[256];
    __shared__ double carry_s;
    if (threadIdx.x == 0) carry_s = 0.0;
    __syncthreads();
    for (int b0 = 0; b0 < ntiles; b0 += 256) {
        const int i = b0 + threadIdx.x;
        const double v = i < ntiles ? totals[i] : 0.0;
        double total;
        const double before = block_exclusive_scan(v, lds, &total);
        const double carry = carry_s;
        __syncthreads();
        if (i < ntiles) totals[i] = carry + before;
        if (threadIdx.x == 0) carry_s = carry + total;
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_cumsum_add(double* __restrict__ dst, long long n, const double* __restrict__ offsets) {
    const double off = offsets[blockIdx.x];
    const long long base = (long long)blockIdx.x * kScanTile;
    for (int k = threadIdx.x; k < kScanTile; k += 256)
        if (base + k < n) dst[base + k] += off;
}

__global__ __launch_bounds__(256) void k_min(const double* __restrict__ a, long long n, double* __restrict__ partial) {
    double m = __builtin_inf();
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) m = fmin(m, a[i]);
    for (int o = 32; o > 0; o >>= 1) m = fmin(m, __shfl_xor(m, o));
    __shared__ double w[4];
    if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = fmin(fmin(w[0], w[1]), fmin(w[2], w[3]));
}

}  // namespace

namespace {
__global__ __launch_bounds__(256) void k_real_part(double* __restrict__ dst, const double2* __restrict__ src, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) dst[i] = src[i].x;
}
int host_alloc(size_t bytes, void** out);
int host_free(void* ptr, size_t bytes);
}
extern "C" int ssfm_device_alloc(int device, size_t bytes, void** out) {
    if (!out || bytes == 0) return fail(SSFM_ERR_INVALID, "ssfm_device_alloc: bytes=%zu", bytes);
    if (device == SSFM_HOST_PINNED) return host_alloc(bytes, out);
    if (int rc = use(device)) return rc;
    Pool& p = g_pool[device];
    {
        std::lock_guard<std::mutex> lock(p.mu);
        auto it = p.free_by_size.find(bytes);
        if (it != p.free_by_size.end() && !it->second.empty()) {
            *out = it->second.back();
            it->second.pop_back();
            p.cached_bytes -= bytes;
            return SSFM_OK;
        }
    }
    hipError_t e = hipMalloc(out, bytes);
    if (e != hipSuccess) {
        // give the cache back to the allocator and retry once
        std::vector<void*> drop;
        {
            std::lock_guard<std::mutex> lock(p.mu);
            for (auto& kv : p.free_by_size) { drop.insert(drop.end(), kv.second.begin(), kv.second.end()); kv.second.clear(); }
            p.cached_bytes = 0;
        }
        for (void* q : drop) (void)hipFree(q);
        e = hipMalloc(out, bytes);
    }
    if (e != hipSuccess) return fail(SSFM_ERR_HIP, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    return SSFM_OK;
}

extern "C" int ssfm_device_free(int device, void* ptr, size_t bytes) {
    if (!ptr) return SSFM_OK;
    if (device == SSFM_HOST_PINNED) return host_free(ptr, bytes);
    if (int rc = use(device)) return rc;
    Pool& p = g_pool[device];
    {
        std::lock_guard<std::mutex> lock(p.mu);
        if (bytes > 0 && p.cached_bytes + bytes <= kMaxCachedBytes) {
            p.free_by_size[bytes].push_back(ptr);
            p.cached_bytes += bytes;
            return SSFM_OK;
        }
    }
    HIP_TRY(hipFree(ptr));
    return SSFM_OK;
}

// Page-locked host memory for results that are read back: a device-to-host copy into fresh pageable memory makes the
// runtime lock and later unlock the destination pages (measured on the MI355X box: two 2 MiB read-backs leave the NEXT
// transfer of the process stalled for 20 ms) and takes page faults on every first touch; into a pooled page-locked
// buffer it is one DMA at link speed.  Pooled by size like the device buffers.
namespace {
int host_alloc(size_t bytes, void** out) {
    {
        std::lock_guard<std::mutex> lock(g_host_pool.mu);
        auto it = g_host_pool.free_by_size.find(bytes);
        if (it != g_host_pool.free_by_size.end() && !it->second.empty()) {
            *out = it->second.back();
            it->second.pop_back();
            g_host_pool.cached_bytes -= bytes;
            return SSFM_OK;
        }
    }
    HIP_TRY(hipHostMalloc(out, bytes, hipHostMallocDefault));
    return SSFM_OK;
}

int host_free(void* ptr, size_t bytes) {
    {
        std::lock_guard<std::mutex> lock(g_host_pool.mu);
        if (bytes > 0 && g_host_pool.cached_bytes + bytes <= kMaxCachedHostBytes) {
            g_host_pool.free_by_size[bytes].push_back(ptr);
            g_host_pool.cached_bytes += bytes;
            return SSFM_OK;
        }
    }
    HIP_TRY(hipHostFree(ptr));
    return SSFM_OK;
}
}  // namespace

extern "C" int ssfm_device_copy(int device, void* dst, const void* src, size_t bytes, int kind) {
    if (!dst || (!src && kind != 3)) return fail(SSFM_ERR_INVALID, "ssfm_device_copy: NULL argument");
    if (kind < 0 || kind > 3) return fail(SSFM_ERR_INVALID, "ssfm_device_copy: kind=%d (0 host->device, 1 device->host, 2 device->device, 3 zero bytes)", kind);
    if (int rc = use(device)) return rc;
    if (kind == 3) { HIP_TRY(hipMemset(dst, 0, bytes)); return SSFM_OK; }
    const hipMemcpyKind k = kind == 0 ? hipMemcpyHostToDevice : kind == 1 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    HIP_TRY(hipMemcpy(dst, src, bytes, k));
    if (kind == 2) HIP_TRY(hipDeviceSynchronize());          // device-to-device hipMemcpy may return early
    return SSFM_OK;
}

extern "C" int ssfm_device_convert(int device, const void* src, int src_precision, void* dst, int dst_precision, int64_t count) {
    if (!dst || !src || count < 1) return fail(SSFM_ERR_INVALID, "ssfm_device_convert: bad argument");
    if (int rc = use(device)) return rc;
    const long long c2 = 2 * (long long)count;
    if (src_precision == SSFM_C64 && dst_precision == SSFM_C128)
        hipLaunchKernelGGL((k_convert<float, double>), dim3(blocks_for(c2)), dim3(256), 0, 0, (const float*)src, (double*)dst, c2);
    else if (src_precision == SSFM_C128 && dst_precision == SSFM_C64)
        hipLaunchKernelGGL((k_convert<double, float>), dim3(blocks_for(c2)), dim3(256), 0, 0, (const double*)src, (float*)dst, c2);
    else if (src_precision == SSFM_F64_REAL && dst_precision == SSFM_C128)
        hipLaunchKernelGGL((k_widen<double>), dim3(blocks_for(count)), dim3(256), 0, 0, (const double*)src, (double*)dst, (long long)count);
    else if (src_precision == SSFM_F64_REAL && dst_precision == SSFM_C64)
        hipLaunchKernelGGL((k_widen<float>), dim3(blocks_for(count)), dim3(256), 0, 0, (const double*)src, (float*)dst, (long long)count);
    else if (src_precision == SSFM_C128 && dst_precision == SSFM_F64_REAL)
        hipLaunchKernelGGL(k_real_part, dim3(blocks_for(count)), dim3(256), 0, 0, (double*)dst, (const double2*)src, (long long)count);
    else
        return fail(SSFM_ERR_INVALID, "ssfm_device_convert: precisions %d -> %d", src_precision, dst_precision);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    return SSFM_OK;
}

extern "C" int ssfm_device_add(int device, void* dst, const void* a, const void* b, int precision, int64_t count) {
    if (!dst || !a || !b || count < 1) return fail(SSFM_ERR_INVALID, "ssfm_device_add: bad argument");
    if (int rc = use(device)) return rc;
    const long long c2 = 2 * (long long)count;
    if (precision == SSFM_C64) hipLaunchKernelGGL(k_add<float>, dim3(blocks_for(c2)), dim3(256), 0, 0, (float*)dst, (const float*)a, (const float*)b, c2);
    else if (precision == SSFM_C128) hipLaunchKernelGGL(k_add<double>, dim3(blocks_for(c2)), dim3(256), 0, 0, (double*)dst, (const double*)a, (const double*)b, c2);
    else return fail(SSFM_ERR_INVALID, "ssfm_device_add: precision %d", precision);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    return SSFM_OK;
}

extern "C" int ssfm_device_mem_info(int device, size_t* free_bytes, size_t* total_bytes, size_t* pooled_bytes) {
    if (!free_bytes || !total_bytes) return fail(SSFM_ERR_INVALID, "ssfm_device_mem_info: NULL argument");
    if (int rc = use(device)) return rc;
    HIP_TRY(hipMemGetInfo(free_bytes, total_bytes));
    if (pooled_bytes) {
        std::lock_guard<std::mutex> lock(g_pool[device].mu);
        *pooled_bytes = g_pool[device].cached_bytes;
    }
    return SSFM_OK;
}

// ------------------------------------------------------------------------------------- device random numbers
// The reference draws its thermal / shot / ASE noise from NumPy's global generator (devices.py:1521-1527, :930);
// the host mirror reproduces those draws on the host when asked for seed-for-seed parity.  For Monte-Carlo runs
// that only need the statistics, this is the documented device generator: Philox4x32-10 (Salmon et al., SC'11)
// with key = the 64-bit seed and counter = (pair index, stream); one call yields 128 random bits = two 53-bit
// uniforms u1, u2 in (0, 1) = one Box-Muller pair  sqrt(-2 ln u1) * (cos, sin)(2 pi u2).  Element 2p and 2p+1 of
// the output come from pair p, so a buffer's content depends only on (seed, stream), not on the launch shape.
namespace {

__device__ __forceinline__ void philox4x32_10(unsigned (&c)[4], unsigned k0, unsigned k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c[0];
        const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c[2];
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n1 = (unsigned)p1;
        const unsigned n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1, n3 = (unsigned)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

__global__ __launch_bounds__(256) void k_randn(double* __restrict__ out, long long n, unsigned long long seed, unsigned long long stream, double mean, double std) {
    const long long pairs = (n + 1) / 2;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < pairs; p += (long long)gridDim.x * blockDim.x) {
        unsigned c[4] = {(unsigned)p, (unsigned)((unsigned long long)p >> 32), (unsigned)stream, (unsigned)(stream >> 32)};
        philox4x32_10(c, (unsigned)seed, (unsigned)(seed >> 32));
        const double u1 = ((double)(((unsigned long long)(c[0] >> 5) << 26) | (c[1] >> 6)) + 0.5) * 0x1p-53;
        const double u2 = ((double)(((unsigned long long)(c[2] >> 5) << 26) | (c[3] >> 6)) + 0.5) * 0x1p-53;
        const double r = sqrt(-2.0 * log(u1));
        double s, co;
        sincospi(2.0 * u2, &s, &co);
        out[2 * p] = mean + std * r * co;
        if (2 * p + 1 < n) out[2 * p + 1] = mean + std * r * s;
    }
}

// out = (a + b + c + offset) * scale over `n` doubles; a, b, c nullable (PD's noise current, devices.py:1529-1547)
__global__ __launch_bounds__(256) void k_sum3(double* __restrict__ out, const double* __restrict__ a, const double* __restrict__ b, const double* __restrict__ c,
                                              double offset, double scale, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        double v = 0.0;
        bool any = false;
        if (a) { v = a[i]; any = true; }
        if (b) { v = any ? v + b[i] : b[i]; any = true; }
        if (c) { v = any ? v + c[i] : c[i]; any = true; }
        out[i] = (any ? v + offset : offset) * scale;
    }
}

// dst = a * factor (+ b) over `n` doubles (complex arrays are passed as 2n doubles); b nullable
__global__ __launch_bounds__(256) void k_scale_add(double* __restrict__ dst, const double* __restrict__ a, double factor, const double* __restrict__ b, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        dst[i] = b ? a[i] * factor + b[i] : a[i] * factor;
}

__global__ __launch_bounds__(256) void k_sum(const double* __restrict__ a, const double* __restrict__ b, long long n, double* __restrict__ partial) {
    double acc = 0.0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) acc += b ? a[i] + b[i] : a[i];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    __shared__ double w[4];
    if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = w[0] + w[1] + w[2] + w[3];
}

}  // namespace

extern "C" int ssfm_device_randn(int device, double* out_dev, int64_t n, uint64_t seed, uint64_t stream, double mean, double std) {
    if (!out_dev || n < 1) return fail(SSFM_ERR_INVALID, "ssfm_device_randn: bad argument");
    if (int rc = use(device)) return rc;
    hipLaunchKernelGGL(k_randn, dim3(blocks_for((n + 1) / 2)), dim3(256), 0, 0, out_dev, (long long)n, (unsigned long long)seed, (unsigned long long)stream, mean, std);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    return SSFM_OK;
}

extern "C" int ssfm_device_sum3(int device, double* out_dev, const double* a, const double* b, const double* c, double offset, double scale, int64_t n) {
    if (!out_dev || n < 1) return fail(SSFM_ERR_INVALID, "ssfm_device_sum3: bad argument");
    if (int rc = use(device)) return rc;
    hipLaunchKernelGGL(k_sum3, dim3(blocks_for(n)), dim3(256), 0, 0, out_dev, a, b, c, offset, scale, (long long)n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    return SSFM_OK;
}

extern "C" int ssfm_device_scale_add(int device, double* dst, const double* a, double factor, const double* b, int64_t n) {
    if (!dst || !a || n < 1) return fail(SSFM_ERR_INVALID, "ssfm_device_scale_add: bad argument");
    if (int rc = use(device)) return rc;
    hipLaunchKernelGGL(k_scale_add, dim3(blocks_for(n)), dim3(256), 0, 0, dst, a, factor, b, (long long)n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    return SSFM_OK;
}

namespace ssfm { SSFM_INTERNAL int device_mean(int device, const double* a, const double* b, int64_t n, double* mean_out); SSFM_INTERNAL int device_min(int device, const double* a, int64_t n, double* min_out); }
int ssfm::device_mean(int device, const double* a, const double* b, int64_t n, double* mean_out) {
    if (!a || !mean_out || n < 1) return fail(SSFM_ERR_INVALID, "ssfm_device_reduce (mean): bad argument");
    if (int rc = use(device)) return rc;
    constexpr int kBlocks = 1024;
    double* partial = nullptr;
    if (int rc = ssfm_device_alloc(device, sizeof(double) * kBlocks, (void**)&partial)) return rc;
    hipLaunchKernelGGL(k_sum, dim3(kBlocks), dim3(256), 0, 0, a, b, (long long)n, partial);
    double host[kBlocks];
    hipError_t e = hipMemcpy(host, partial, sizeof(host), hipMemcpyDeviceToHost);
    (void)ssfm_device_free(device, partial, sizeof(double) * kBlocks);
    if (e != hipSuccess) return fail(SSFM_ERR_HIP, "ssfm_device_reduce (mean): %s", hipGetErrorString(e));
    double acc = 0.0;
    for (int i = 0; i < kBlocks; ++i) acc += host[i];
    *mean_out = acc / (double)n;
    return SSFM_OK;
}

extern "C" int ssfm_device_cumsum(int device, double* dst, const double* src, int64_t n) {
    if (!dst || !src || n < 1) return fail(SSFM_ERR_INVALID, "ssfm_device_cumsum: bad argument");
    if (int rc = use(device)) return rc;
    const int ntiles = (int)((n + kScanTile - 1) / kScanTile);
    double* totals = nullptr;
    if (int rc = ssfm_device_alloc(device, sizeof(double) * ntiles, (void**)&totals)) return rc;
    hipLaunchKernelGGL(k_cumsum_tiles, dim3(ntiles), dim3(256), 0, 0, src, dst, (long long)n, totals);
    if (ntiles > 1) {
        hipLaunchKernelGGL(k_cumsum_totals, dim3(1), dim3(256), 0, 0, totals, ntiles);
        hipLaunchKernelGGL(k_cumsum_add, dim3(ntiles), dim3(256), 0, 0, dst, (long long)n, totals);
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();
    (void)ssfm_device_free(device, totals, sizeof(double) * ntiles);
    if (e != hipSuccess) return fail(SSFM_ERR_HIP, "ssfm_device_cumsum: %s", hipGetErrorString(e));
    return SSFM_OK;
}

int ssfm::device_min(int device, const double* a, int64_t n, double* min_out) {
    if (!a || !min_out || n < 1) return fail(SSFM_ERR_INVALID, "ssfm_device_reduce (min): bad argument");
    if (int rc = use(device)) return rc;
    constexpr int kBlocks = 1024;
    double* partial = nullptr;
    if (int rc = ssfm_device_alloc(device, sizeof(double) * kBlocks, (void**)&partial)) return rc;
    hipLaunchKernelGGL(k_min, dim3(kBlocks), dim3(256), 0, 0, a, (long long)n, partial);
    double host[kBlocks];
    hipError_t e = hipMemcpy(host, partial, sizeof(host), hipMemcpyDeviceToHost);
    (void)ssfm_device_free(device, partial, sizeof(double) * kBlocks);
    if (e != hipSuccess) return fail(SSFM_ERR_HIP, "ssfm_device_reduce (min): %s", hipGetErrorString(e));
    double m = host[0];
    for (int i = 1; i < kBlocks; ++i) m = host[i] < m ? host[i] : m;
    *min_out = m;
    return SSFM_OK;
}
