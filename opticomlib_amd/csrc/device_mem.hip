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
template <typename T>
__global__ __launch_bounds__(256) void k_add(T* __restrict__ dst, const T* __restrict__ a, const T* __restrict__ b, long long count2) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count2; i += (long long)gridDim.x * blockDim.x) dst[i] = a[i] + b[i];
}
unsigned blocks_for(long long n) { return (unsigned)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192); }

}  // namespace

extern "C" int ssfm_device_alloc(int device, size_t bytes, void** out) {
    if (!out || bytes == 0) return fail(SSFM_ERR_INVALID, "ssfm_device_alloc: bytes=%zu", bytes);
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

extern "C" int ssfm_device_copy(int device, void* dst, const void* src, size_t bytes, int kind) {
    if (!dst || !src) return fail(SSFM_ERR_INVALID, "ssfm_device_copy: NULL argument");
    if (kind < 0 || kind > 2) return fail(SSFM_ERR_INVALID, "ssfm_device_copy: kind=%d (0 host->device, 1 device->host, 2 device->device)", kind);
    if (int rc = use(device)) return rc;
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
