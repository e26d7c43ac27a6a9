// prbs.hip -- the reference's pseudo-random bit generator (devices.py:63-182) on the device, and the symbol
// loaders that consume its bits where they lie.
//
// The reference walks a Fibonacci LFSR of `order` bits one bit per Python iteration: emit bit 0 of the state, form
// new = bit[t1] ^ bit[t2] (taps of devices.py:134-142, zero-based), shift it in from the right (devices.py:166-175).
// A shift is a linear map M over GF(2) on the state, so the state at position j is M^j seed.  Every thread jumps
// to the start of its chunk with the binary decomposition of j (M^(2^k) for k < 40, built on the host once per
// order: 40 x order words) and then walks kChunk bits exactly as the reference does.  Integer arithmetic only:
// the sequence and the final register state (return_seed) are bit for bit the reference's.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <mutex>

#include "ssfm_amd.h"
#include "ssfm_common.hpp"

using ssfm::fail;
namespace ssfm { SSFM_INTERNAL int plan_load_bits(ssfm_plan* plan, int64_t plan_n, const void* bits_dev, int64_t nbits, int up); }

namespace {

constexpr int kChunk = 256;          // bits per thread
constexpr int kPowers = 40;          // sequences of up to 2^40 bits

struct Jump {
    uint32_t m[kPowers][32];         // row r of M^(2^k): bit r of the new state = parity(row & state)
};

int taps_of(int order, int* t1, int* t2) {
    static const int tab[][3] = {{7, 7, 6}, {9, 9, 5}, {11, 11, 9}, {15, 15, 14}, {20, 20, 3}, {23, 23, 18}, {31, 31, 28}};
    for (auto& t : tab)
        if (t[0] == order) { *t1 = t[1] - 1; *t2 = t[2] - 1; return 1; }
    return 0;
}

// one shift of the reference (devices.py:172-174): state' = ((state << 1) | (bit t1 ^ bit t2)) & mask
void shift_matrix(int order, int t1, int t2, uint32_t m[32]) {
    std::memset(m, 0, sizeof(uint32_t) * 32);
    m[0] = (1u << t1) ^ (1u << t2);
    for (int r = 1; r < order; ++r) m[r] = 1u << (r - 1);
}
// c = a * b over GF(2): row r of c = XOR of the rows of b selected by the bits of row r of a
void mat_mul(const uint32_t a[32], const uint32_t b[32], uint32_t c[32], int order) {
    for (int r = 0; r < order; ++r) {
        uint32_t acc = 0;
        for (int k = 0; k < order; ++k)
            if ((a[r] >> k) & 1u) acc ^= b[k];
        c[r] = acc;
    }
    for (int r = order; r < 32; ++r) c[r] = 0;
}

__device__ __forceinline__ uint32_t apply(const uint32_t* __restrict__ m, uint32_t s, int order) {
    uint32_t out = 0;
    for (int r = 0; r < order; ++r) out |= (uint32_t)(__popc(m[r] & s) & 1) << r;
    return out;
}

__global__ __launch_bounds__(256) void k_prbs(uint8_t* __restrict__ out, long long len, int order, int t1, int t2, uint32_t seed,
                                              const Jump* __restrict__ jump, uint32_t* __restrict__ final_state) {
    __shared__ uint32_t mats[kPowers][32];
    for (int e = threadIdx.x; e < kPowers * 32; e += blockDim.x) mats[e / 32][e % 32] = jump->m[e / 32][e % 32];
    __syncthreads();
    const long long chunk = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long j0 = chunk * kChunk;
    if (j0 > len) return;
    uint32_t s = seed;
    for (int k = 0; k < kPowers; ++k)
        if ((j0 >> k) & 1) s = apply(mats[k], s, order);
    const uint32_t mask = order == 32 ? 0xffffffffu : ((1u << order) - 1u);
    const long long end = j0 + kChunk < len ? j0 + kChunk : len;
    for (long long j = j0; j < end; ++j) {
        out[j] = (uint8_t)(s & 1u);                                   // devices.py:171
        const uint32_t nw = ((s >> t1) ^ (s >> t2)) & 1u;             // devices.py:172
        s = ((s << 1) | nw) & mask;                                   // devices.py:173
    }
    // the thread whose walk ends at `len` holds the state after `len` shifts (devices.py:181, return_seed)
    if (final_state && chunk == len / kChunk) *final_state = s;
}

// field (complex128, zero-padded to M) <- nsym amplitudes from bits (0.0 / 1.0), zero-stuffed to `up` samples per bit
__global__ __launch_bounds__(256) void k_load_bits(const uint8_t* __restrict__ bits, long long nbits, int up, double2* __restrict__ F, long long M) {
    const int at = up / 2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / up;
        F[i] = make_double2((b < nbits && i - b * up == at) ? (double)bits[b] : 0.0, 0.0);
    }
}

// QPSK-like test field of the benchmark configurations: row r, symbol k takes bits
// (b0, b1) = bits[2 (r nsym + k)], bits[2 (r nsym + k) + 1] -> ((2 b0 - 1) + j (2 b1 - 1)) / sqrt 2 at sample k sps + sps/2
__global__ __launch_bounds__(256) void k_load_qpsk(const uint8_t* __restrict__ bits, long long nsym, int rows, int sps, double2* __restrict__ F, long long n) {
    const int at = sps / 2;
    const long long total = n * rows;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / n, m = i - r * n, k = m / sps;
        double2 v = make_double2(0.0, 0.0);
        if (k < nsym && m - k * sps == at) {
            const long long b = 2 * (r * nsym + k);
            v = make_double2((2.0 * bits[b] - 1.0) / 1.4142135623730951, (2.0 * bits[b + 1] - 1.0) / 1.4142135623730951);
        }
        F[i] = v;
    }
}

std::mutex g_jump_mu;
struct JumpCache { int order; Jump* dev; };
JumpCache g_jump[8][8] = {};          // [device][slot]

int jump_for(int device, int order, int t1, int t2, const Jump** out) {
    std::lock_guard<std::mutex> lk(g_jump_mu);
    if (device < 0 || device >= 8) return fail(SSFM_ERR_INVALID, "ssfm_prbs: device %d", device);
    for (auto& c : g_jump[device])
        if (c.dev && c.order == order) { *out = c.dev; return SSFM_OK; }
    Jump* h = new (std::nothrow) Jump();
    if (!h) return fail(SSFM_ERR_INVALID, "out of host memory");
    shift_matrix(order, t1, t2, h->m[0]);
    for (int k = 1; k < kPowers; ++k) mat_mul(h->m[k - 1], h->m[k - 1], h->m[k], order);
    Jump* d = nullptr;
    hipError_t e = hipMalloc(&d, sizeof(Jump));
    if (e == hipSuccess) e = hipMemcpy(d, h, sizeof(Jump), hipMemcpyHostToDevice);
    delete h;
    if (e != hipSuccess) { (void)hipFree(d); return fail(SSFM_ERR_HIP, "ssfm_prbs: %s", hipGetErrorString(e)); }
    for (auto& c : g_jump[device])
        if (!c.dev) { c.order = order; c.dev = d; *out = d; return SSFM_OK; }
    (void)hipFree(d);
    return fail(SSFM_ERR_INVALID, "ssfm_prbs: jump-table cache full");
}

unsigned blocks_for(long long n) { return (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096); }

}  // namespace

extern "C" int ssfm_prbs(int device, void* bits_dev, int64_t len, int order, uint32_t seed, uint32_t* final_state) {
    int t1 = 0, t2 = 0;
    if (!taps_of(order, &t1, &t2)) return fail(SSFM_ERR_INVALID, "ssfm_prbs: order %d is not one of 7, 9, 11, 15, 20, 23, 31", order);
    if (!bits_dev || len < 1 || len >= (1ll << kPowers)) return fail(SSFM_ERR_INVALID, "ssfm_prbs: %lld bits", (long long)len);
    if (seed == 0 || (order < 32 && (seed >> order) != 0)) return fail(SSFM_ERR_INVALID, "ssfm_prbs: seed %u is not a state of %d bits", seed, order);
    HIP_TRY(hipSetDevice(device));
    const Jump* jump = nullptr;
    if (int rc = jump_for(device, order, t1, t2, &jump)) return rc;
    uint32_t* fs_dev = nullptr;
    if (final_state) HIP_TRY(hipMalloc(&fs_dev, sizeof(uint32_t)));
    const long long chunks = len / kChunk + 1;                        // one more: the state after a whole number of chunks
    hipLaunchKernelGGL(k_prbs, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, 0, (uint8_t*)bits_dev, (long long)len, order, t1, t2, seed, jump, fs_dev);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && final_state) e = hipMemcpy(final_state, fs_dev, sizeof(uint32_t), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    (void)hipFree(fs_dev);
    if (e != hipSuccess) return fail(SSFM_ERR_HIP, "ssfm_prbs: %s", hipGetErrorString(e));
    return SSFM_OK;
}

// (behind ssfm_load_symbols(..., src_kind = 1), csrc/chirpz.hip)
int ssfm::plan_load_bits(ssfm_plan* plan, int64_t plan_n, const void* bits_dev, int64_t nbits, int up) {
    if (!plan) return fail(SSFM_ERR_INVALID, "null plan");
    double2* F = static_cast<double2*>(ssfm::plan_field(plan));
    if (!F || !bits_dev || nbits < 1 || up < 1 || nbits * up > plan_n)
        return fail(SSFM_ERR_INVALID, "ssfm_load_symbols: %lld bits x %d samples for a plan of %lld", (long long)nbits, up, (long long)plan_n);
    hipLaunchKernelGGL(k_load_bits, dim3(blocks_for(plan_n)), dim3(256), 0, static_cast<hipStream_t>(ssfm::plan_stream(plan)), (const uint8_t*)bits_dev,
                       (long long)nbits, up, F, (long long)plan_n);
    HIP_TRY(hipGetLastError());
    return SSFM_OK;
}

extern "C" int ssfm_load_qpsk(ssfm_plan* plan, int64_t plan_n, int rows, const void* bits_dev, int64_t nsym, int sps) {
    if (!plan) return fail(SSFM_ERR_INVALID, "null plan");
    double2* F = static_cast<double2*>(ssfm::plan_field(plan));
    if (!F || !bits_dev || nsym < 1 || sps < 1 || rows < 1 || nsym * sps > plan_n)
        return fail(SSFM_ERR_INVALID, "ssfm_load_qpsk: %lld symbols x %d samples for a plan of %lld", (long long)nsym, sps, (long long)plan_n);
    hipLaunchKernelGGL(k_load_qpsk, dim3(blocks_for(plan_n * rows)), dim3(256), 0, static_cast<hipStream_t>(ssfm::plan_stream(plan)), (const uint8_t*)bits_dev,
                       (long long)nsym, rows, sps, F, (long long)plan_n);
    HIP_TRY(hipGetLastError());
    return SSFM_OK;
}
