// sos_filter.hip -- zero-phase IIR filtering (cascade of second-order sections, forward-backward) on
// gfx950: the arithmetic behind the reference's LPF / BPF (devices.py:1363-1368, :814-823), i.e.
// scipy.signal.sosfiltfilt(sos, x, axis=-1) with its default odd padding of 3*ntaps samples and
// steady-state initial conditions.
//
// A cascade of NS biquads in direct form II transposed is a linear recurrence with a K = 2*NS state
// per channel.  It is parallelised over time by chunks of L = 12 samples, TWO launches per direction:
//   k_chunk_scan  a thread runs the recurrence over ITS chunk from a ZERO state (the chunk's particular
//                 solution p_c); the 64 chunks of a wavefront are combined by a shuffle scan of the affine
//                 maps s -> M s + p_c (M = A^L, the homogeneous map over one chunk), the 4 wavefronts of
//                 the workgroup through LDS: every chunk gets e_c = its start state if the workgroup's
//                 GROUP of 256 chunks started from zero, the group its total T_g;
//   k_apply       a workgroup first forms the true start state of its group,
//                     S_g = (M^256)^g s_0 + sum_{j<g} (M^256)^(g-1-j) T_j,      s_0 = zi * x_0,
//                 by a reduction over all earlier group totals (each lane multiplies by its own power
//                 (M^256)^lane, a wavefront sums with shuffles and applies (M^16384)^a); then a thread
//                 forms its chunk's start state M^(c mod 256) S_g + e_c, re-runs the chunk and writes
//                 the outputs.  The reduction is O(groups^2) over the grid -- 342 groups per 2^20-sample
//                 row, nothing against the sample work -- and replaces a separate, serial scan kernel
//                 (measured 25 us of a 127 us call).
// The powers M^j, (M^256)^j, (M^16384)^j, j = 0..64, are built once per filter on the host from M, which
// itself comes from running the same recurrence on unit states.  A wavefront's 64 chunks are 768
// consecutive samples: they are read (and the outputs written) with coalesced accesses and transposed
// through LDS, so that a thread still walks ITS chunk in order.  Inside a chunk the operation order is
// exactly SciPy's sample loop (fp contraction off); only the chunk start states see a different
// summation order (relative 1e-16 effects).  A complex row is two real channels with the same real
// coefficients, carried by ONE thread (16-byte accesses).  float64 throughout, like SciPy.
//
// Traffic per real sample: forward 8 (chunk) + 8 + 8 (apply: read x, write y1), backward the same
// = 48 B against the algorithmic 16 B (read x once, write y once).
//
// That is the THREE-LAUNCH form (orders 5-8 of long calls, SSFM_SOS_ONE_LAUNCH=0, and the fallback).  Calls whose workgroups are all resident at
// once take ONE launch (k_filtfilt, sos_filter_impl.inc): a workgroup keeps its chunks in registers / LDS from the first load to the last store and
// meets the others twice through the group totals -- 16 B per real sample.  Round 5: the totals of orders up to 4 cross as 16-byte units {value, tag}
// that carry their own validity (no flag, no acknowledgement awaited), the wavefront scan runs inside DPP rows; 32-33 us for 2^20 x 2 complex128
// (profiles/r05_sos_handover_ab.txt).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "ssfm_common.hpp"

using ssfm::fail;

namespace {

#ifndef SOS_CHUNK
#define SOS_CHUNK 12
#endif
// Samples per thread.  Two lengths (profiles/r03_sosfilt.txt, section 11): SOS_CHUNK for the three-launch form and for short calls
// (first version: 16 -> 121 us, 32 -> 130 us, 64 -> 158 us per call at 2^20 x 2 complex; later 8 -> 101, 12 -> 90, 16 -> 97; 12 is 4-14 %
// faster than 16 for every shape from 2^14 to 2^20 x 2), kChunkLong for the one-launch form of long calls: fewer threads pay the
// per-chunk scan and look-back, two workgroups per CU instead of three hold the same samples (2^20 x 2 complex128: 47.5 -> 39.8 us,
// 2^20 real 33.0 -> 25.3; short calls lose 10-20 % with it).
constexpr int kChunkLong = 18;
constexpr int kWave = 64;          // lanes per wavefront
#ifndef SOS_WAVES
#define SOS_WAVES 4
#endif
// Wavefronts per workgroup W (a template parameter of the kernels; chunks per group = threads per workgroup = 64 W).
// Measured, kernels per call (profiles/r02_sosfilt_shape.txt): 2^20 x 2 complex 87 us with 4, 95 with 2, 88.5 with 3
// (first version: 1 -> 131 us); 2^20 real 45 / 48-50 / 43; but the short calls of a 2^14 ... 2^16-sample link, which
// are a latency chain of four small kernels, run 11-12 % faster with 2 (2^16 real 32.3 -> 28.5 us, 2^14 complex 38.1 ->
// 34.0): `sos_waves()` picks 2 up to kSmallChunks chunks in all, SOS_WAVES above.
// Round 6: the small shape is ONE wavefront (SOS_WAVES_SMALL) and is taken wherever its grid is resident at once -- with the look-back near
// (sos_filter_impl.inc group_start_near) a smaller group costs nothing and has no carry between wavefronts: 4-15 % under two wavefronts from 2^14 to
// 2^20 samples (profiles/r06_sos_w1.txt), 15-25 % under the shapes round 5 chose for 2^17 ... 2^20 (profiles/r06_sos_shape_sweep.txt).  Asking the
// compiler for three wavefronts per SIMD so that 2^20 x 2 complex128 fits too (SOS_SMALL_EU=3): no gain there, 2-4 us lost at 2^18 x 2 -- stays 1.
constexpr int kWavesLarge = SOS_WAVES;
#ifndef SOS_WAVES_SMALL
#define SOS_WAVES_SMALL 1
#endif
#ifndef SOS_SMALL_EU
#define SOS_SMALL_EU 1
#endif
#ifndef SOS_SMALL_AHEAD
#define SOS_SMALL_AHEAD 1
#endif
constexpr int kWavesSmall = SOS_WAVES_SMALL;
constexpr long long kSmallChunks = 16384;
constexpr int kMaxSections = 4;    // Bessel orders up to 8

struct SosCoefs {
    double b0[kMaxSections], b1[kMaxSections], b2[kMaxSections], a1[kMaxSections], a2[kMaxSections];
};

// The 16-byte sample stores go out write-through (sc1), as the propagator's do -- the next kernel's readers sit on
// other CUs, so a dirty line in this XCD's L2 only adds a flush at the kernel boundary.  2^20 x 2 complex: 90.5 -> 87.5 us,
// other shapes unchanged; non-temporal loads measured 4 % slower (profiles/r02_sosfilt_policy_ab.txt).
// One direction of the forward-backward pass.  Rows hold CH interleaved channels.
struct SosPass {
    const double* src;      // forward: caller's x; backward: y1 (padded forward output, [row][m][CH])
    long long n;            // original samples per row
    long long m;            // padded length n + 2*edge
    long long mv;           // length of the sequence the pass runs over: forward m; backward the whole chunk grid, ngroups * group * kChunk >= m:
                            // the backward sequence is b[i'] = y1[min(mv - 1 - i', m - 1)] -- it STARTS with mv - m copies of y1[m - 1], under which
                            // the initial state zi * y1[m - 1] (scipy: zi * y[-1]) stays where it is (steady state) -- so that backward chunk c'
                            // is forward chunk (chunks - 1 - c') sample for sample, and the forward output pass can run the backward chunk pass
                            // on the outputs it holds in registers
    int edge;
    int backward;
};

// ---------------------------------------------------------------------------------------------- host
// Scratch memory of the filter, one set per device, grown on demand and kept (a hipMalloc/hipFree pair
// per call cost more than the kernels).  Calls on one device are serialised by `mu`.
struct Workspace {
    std::mutex mu;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double* buf[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};      // x staging, y1, E, T (both passes), E of the backward pass, tables
    size_t cap[6] = {0, 0, 0, 0, 0, 0};
    std::vector<double> table_key;       // sos coefficients the device tables were built for
    float last_ms = 0.f;
    // one-launch form (k_filtfilt): group totals + their flags, the call counter the flags are compared with, the host-visible give-up word
    double* link_T = nullptr;
    unsigned* link_flag = nullptr;
    size_t link_T_cap = 0, link_flag_cap = 0;
    unsigned long long* link_U = nullptr;    // totals that carry their own validity: 16-byte units {value, tag} (sos_filter_impl.inc group_start)
    size_t link_U_cap = 0;
    unsigned long long link_tag = 0;         // the call counter the tags are; never repeated, never 0
    double* meet_tab = nullptr;              // the single-meeting kernels' maps (sos_filter_impl.inc build_meet_tables), for the filter and row end of `meet_key`
    size_t meet_cap = 0;
    std::vector<double> meet_key;
    int look = 1 << 30;                      // powers of the group map that matter for the filter of `table_key` (sos_filter_impl.inc group_start_near)
    unsigned epoch = 0;
    int* status = nullptr;
    int give_ups = 0;                    // calls IN A ROW that fell back to three launches after waiting in vain; from kMaxGiveUps on the form rests
    int rested = 0;                      // ... for kRestCalls calls, then gets another try
    int last_launches = 0;
    hipError_t need(int i, size_t bytes) {
        if (bytes <= cap[i]) return hipSuccess;
        if (buf[i]) { (void)hipFree(buf[i]); buf[i] = nullptr; cap[i] = 0; }
        hipError_t e = hipMalloc(&buf[i], bytes);
        if (e == hipSuccess) cap[i] = bytes;
        return e;
    }
};
constexpr int kMaxDevices = 64;
Workspace g_ws[kMaxDevices];

template <int K> void matmul(const double* A, const double* B, double* C) {
    for (int r = 0; r < K; ++r)
        for (int q = 0; q < K; ++q) {
            double acc = 0.0;
            for (int m2 = 0; m2 < K; ++m2) acc += A[r * K + m2] * B[m2 * K + q];
            C[r * K + q] = acc;
        }
}

// tables[level][j] = (M_level)^j, j = 0..64; level 0: M = A^kChunk, level 1: M^(64 W) (one group of W x 64
// chunks), level 2: M^16384 (64 groups)
#define WS_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(SSFM_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); } while (0)

// ---- the one-launch form's conditions
constexpr int kMaxGiveUps = 3, kRestCalls = 1000;
// SSFM_SOS_ONE_LAUNCH=0: always three launches (read per call)
inline bool one_launch_enabled() {
    const char* e = std::getenv("SSFM_SOS_ONE_LAUNCH");
    return !(e && std::atoi(e) == 0);
}
// SSFM_SOS_NEAR=0: a group's start state from ALL the earlier totals of its row, whatever the group map's powers are (read per call)
inline bool sos_near_enabled() {
    const char* e = std::getenv("SSFM_SOS_NEAR");
    return !(e && e[0] == '0');
}
// SSFM_SOS_MEET=1: ONE meeting of the workgroups per call instead of two (sos_filter_impl.inc meet_states; read per call).  Built for the verdict of round 5,
// correct (the parity tests run it), and NOT the default: the wait it saves (2.8 us at 2^20 x 2 once the look-back is near) is paid back by the second pass over
// the forward outputs and by the contention of workgroups that no wait takes out of step any more -- 27.7-28.8 us against 26.8-27.2 (profiles/r06_sos_meet_ab.txt).
inline bool sos_meet_enabled() {
    const char* e = std::getenv("SSFM_SOS_MEET");
    return e && e[0] == '1';
}
// A sufficient condition, from the cascade's poles alone, for the powers of the map over `len` samples to be negligible within kNearCertain groups (the
// kernels' own test is on the matrices the host builds, Workspace::look; this one lets the dispatcher choose a shape before any table exists): the largest
// pole modulus r has r^(len * kNearCertain) < 1e-60 -- twenty orders below the kernels' threshold for whatever the transient in front of the decay is.
constexpr int kNearCertain = 4;
template <int NS> bool near_is_certain(const SosCoefs& c, long long len) {
    if (!sos_near_enabled()) return false;
    double r2 = 0.0;
    for (int s = 0; s < NS; ++s) {
        const double a1 = c.a1[s], a2 = c.a2[s], disc = a1 * a1 - 4.0 * a2;
        double m2;
        if (disc < 0.0) m2 = a2;                                             // a complex pair: |z|^2 = a2
        else { const double z = 0.5 * (std::fabs(a1) + std::sqrt(disc)); m2 = z * z; }
        r2 = m2 > r2 ? m2 : r2;
    }
    if (!(r2 < 1.0)) return false;
    return 0.5 * std::log(r2) * (double)len * kNearCertain < -138.2;          // ln 1e-60
}
// how long a workgroup waits for a total before it gives the call up, in 10 ns ticks (SSFM_SOS_PATIENCE_US, default 2 ms)
inline long long one_launch_patience() {
    if (const char* e = std::getenv("SSFM_SOS_PATIENCE_US")) { const long long v = std::atoll(e); if (v > 0) return v * 100; }
    return 200000;
}
namespace chunk_short {
constexpr int kChunk = SOS_CHUNK;
#include "sos_filter_impl.inc"
}  // namespace chunk_short
namespace chunk_long {
constexpr int kChunk = kChunkLong;
#include "sos_filter_impl.inc"
}  // namespace chunk_long

// the workgroup shape follows the size of the call (see kWavesSmall)
inline int sos_waves(long long n, int rows, int edge) {
    const long long chunks = (n + 2ll * edge + chunk_short::kChunk - 1) / chunk_short::kChunk * rows;
    if (const char* e = std::getenv("SOS_WAVES_FORCE")) { const int v = std::atoi(e); if (v == kWavesSmall || v == kWavesLarge) return v; }
    return chunks <= kSmallChunks ? kWavesSmall : kWavesLarge;
}
// ... and so does the chunk length: long calls that fit the one-launch form with the long chunk take it (SSFM_SOS_LONG_CHUNK=0: never)
template <int NS, int CH>
int run_filter(Workspace& w, const SosCoefs& c, const double* sos_key, const double* zi_h, const double* x, double* y,
               long long n, int rows, int edge, bool on_device) {
    // Which workgroup shape a call takes (round 6; profiles/r06_sos_shape_sweep.txt, r06_sos_w1.txt -- orders 2 to 8, 8192 ... 2^20 samples, three layouts):
    //  1. ONE wavefront per workgroup, short chunk, wherever that grid is resident at once: with the look-back near (group_start_near) a small group costs nothing and has no
    //     carry between wavefronts -- the best shape or within 4 % of it everywhere it fits (2^20 float64 15.3 us against 17.7 in four wavefronts of the long chunk);
    //  2. one wavefront with the LONG chunk where only that fits (1.5 ... 2.1 M complex samples: 22.5 against 25.4 us at 786 432 x 2);
    //  3. four wavefronts of the short chunk while that is at most one workgroup per CU (orders 4 to 8: 15.5 / 34 / 42 us against the long chunk's 17 / 37 / 45);
    //  4. four wavefronts of the long chunk (two workgroups per CU hold what three of the short one would); 5. the short chunk in three launches.
    // SOS_WAVES_FORCE (test aid) pins the wavefronts, SSFM_SOS_LONG_CHUNK=0 the short chunk.
    const bool forced = std::getenv("SOS_WAVES_FORCE") != nullptr;
    const char* lc = std::getenv("SSFM_SOS_LONG_CHUNK");
    const bool long_ok = !(lc && std::atoi(lc) == 0);
    if (kWavesSmall != kWavesLarge && sos_waves(n, rows, edge) == kWavesSmall)          // (short calls -- in whatever form -- and the test aid)
        return chunk_short::run_filter_w<NS, CH, kWavesSmall>(w, c, sos_key, zi_h, x, y, n, rows, edge, on_device);
    if (kWavesSmall != kWavesLarge && !forced) {
        if (chunk_short::one_launch_would_run<NS, CH, kWavesSmall>(w, c, n, rows, edge))
            return chunk_short::run_filter_w<NS, CH, kWavesSmall>(w, c, sos_key, zi_h, x, y, n, rows, edge, on_device);
        if (long_ok && chunk_long::one_launch_would_run<NS, CH, kWavesSmall>(w, c, n, rows, edge))
            return chunk_long::run_filter_w<NS, CH, kWavesSmall>(w, c, sos_key, zi_h, x, y, n, rows, edge, on_device);
        if (long_ok) {
            const long long groups = ((n + 2ll * edge + chunk_short::kChunk - 1) / chunk_short::kChunk + kWave * kWavesLarge - 1) / (kWave * kWavesLarge);
            static int cus = 0;
            if (!cus) { int dev = 0, v = 0; if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess) cus = v; }
            if (cus > 0 && groups * rows <= cus && chunk_short::one_launch_would_run<NS, CH, kWavesLarge>(w, c, n, rows, edge))
                return chunk_short::run_filter_w<NS, CH, kWavesLarge>(w, c, sos_key, zi_h, x, y, n, rows, edge, on_device);
        }
    }
    if (long_ok && chunk_long::one_launch_would_run<NS, CH, kWavesLarge>(w, c, n, rows, edge))
        return chunk_long::run_filter_w<NS, CH, kWavesLarge>(w, c, sos_key, zi_h, x, y, n, rows, edge, on_device);
    return chunk_short::run_filter_w<NS, CH, kWavesLarge>(w, c, sos_key, zi_h, x, y, n, rows, edge, on_device);
}

int sosfiltfilt_impl(int device, const double* sos, const double* zi, int n_sections, const void* x, void* y,
                     int64_t n, int batch, int is_complex, bool on_device) {
    if (!sos || !zi || !x || !y) return fail(SSFM_ERR_INVALID, "ssfm_sosfiltfilt: NULL argument");
    if (n_sections < 1 || n_sections > kMaxSections)
        return fail(SSFM_ERR_UNSUPPORTED, "ssfm_sosfiltfilt: %d sections (supported: 1..%d)", n_sections, kMaxSections);
    if (batch < 1 || n < 1) return fail(SSFM_ERR_INVALID, "ssfm_sosfiltfilt: batch=%d n=%lld", batch, (long long)n);
    SosCoefs c;
    std::memset(&c, 0, sizeof(c));
    int zb = 0, za = 0;
    for (int s = 0; s < n_sections; ++s) {
        const double* r = sos + 6 * s;
        if (r[3] != 1.0) return fail(SSFM_ERR_INVALID, "ssfm_sosfiltfilt: section %d has a0 = %g (must be 1)", s, r[3]);
        c.b0[s] = r[0]; c.b1[s] = r[1]; c.b2[s] = r[2]; c.a1[s] = r[4]; c.a2[s] = r[5];
        zb += r[2] == 0.0; za += r[5] == 0.0;
    }
    const int ntaps = 2 * n_sections + 1 - (zb < za ? zb : za);       // scipy sosfiltfilt
    const int edge = 3 * ntaps;
    if (n <= edge) return fail(SSFM_ERR_INVALID, "The length of the input vector x must be greater than padlen, which is %d.", edge);
    if (is_complex && on_device && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15))
        return fail(SSFM_ERR_INVALID, "ssfm_sosfiltfilt_device: complex buffers must be 16-byte aligned");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count || device >= kMaxDevices)
        return fail(SSFM_ERR_NO_DEVICE, "ssfm_sosfiltfilt: device %d not available", device);
    HIP_TRY(hipSetDevice(device));
    Workspace& w = g_ws[device];
    std::lock_guard<std::mutex> lock(w.mu);
    const double* xd = static_cast<const double*>(x);
    double* yd = static_cast<double*>(y);
#define SOS_CASE(NS)                                                                                                  \
    case NS: return is_complex ? run_filter<NS, 2>(w, c, sos, zi, xd, yd, n, batch, edge, on_device)                   \
                               : run_filter<NS, 1>(w, c, sos, zi, xd, yd, n, batch, edge, on_device);
    switch (n_sections) {
        SOS_CASE(1) SOS_CASE(2) SOS_CASE(3)
        default: return is_complex ? run_filter<4, 2>(w, c, sos, zi, xd, yd, n, batch, edge, on_device)
                                   : run_filter<4, 1>(w, c, sos, zi, xd, yd, n, batch, edge, on_device);
    }
#undef SOS_CASE
}

}  // namespace

extern "C" int ssfm_sosfiltfilt(int device, const double* sos, const double* zi, int n_sections, const void* x, void* y,
                                int64_t n, int batch, int is_complex, int on_device) {
    return sosfiltfilt_impl(device, sos, zi, n_sections, x, y, n, batch, is_complex, on_device != 0);
}

extern "C" int ssfm_sosfiltfilt_last(int device, float* ms, int* launches) {
    if (device < 0 || device >= kMaxDevices) return fail(SSFM_ERR_NO_DEVICE, "ssfm_sosfiltfilt_last: device %d", device);
    std::lock_guard<std::mutex> lock(g_ws[device].mu);
    if (ms) *ms = g_ws[device].last_ms;
    if (launches) *launches = g_ws[device].last_launches;
    return SSFM_OK;
}
