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
#include <hip/hip_runtime.h>

#include <algorithm>
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
constexpr int kChunk = SOS_CHUNK;  // samples per thread (measured at 2^20 x 2 complex, first version: 16 -> 121 us, 32 -> 130 us, 64 -> 158 us per call;
                                   // this version: 8 -> 101 us, 12 -> 90 us, 16 -> 97 us; 12 is 4-14 % faster than 16 for every shape from 2^14 to 2^20 x 2)
constexpr int kWave = 64;          // lanes per wavefront
#ifndef SOS_WAVES
#define SOS_WAVES 4
#endif
// Wavefronts per workgroup W (a template parameter of the kernels; chunks per group = threads per workgroup = 64 W).
// Measured, kernels per call (profiles/r02_sosfilt_shape.txt): 2^20 x 2 complex 87 us with 4, 95 with 2, 88.5 with 3
// (first version: 1 -> 131 us); 2^20 real 45 / 48-50 / 43; but the short calls of a 2^14 ... 2^16-sample link, which
// are a latency chain of four small kernels, run 11-12 % faster with 2 (2^16 real 32.3 -> 28.5 us, 2^14 complex 38.1 ->
// 34.0): `sos_waves()` picks 2 up to kSmallChunks chunks in all, SOS_WAVES above.
constexpr int kWavesLarge = SOS_WAVES;
constexpr int kWavesSmall = 2;
constexpr long long kSmallChunks = 16384;
constexpr int kMaxSections = 4;    // Bessel orders up to 8

struct SosCoefs {
    double b0[kMaxSections], b1[kMaxSections], b2[kMaxSections], a1[kMaxSections], a2[kMaxSections];
};

// SOS_WT: the 16-byte sample stores go out write-through (sc1), as the propagator's do -- the next kernel's readers sit on
// other CUs, so a dirty line in this XCD's L2 only adds a flush at the kernel boundary.  2^20 x 2 complex: 90.5 -> 87.5 us,
// other shapes unchanged; non-temporal loads measured 4 % slower (profiles/r02_sosfilt_policy_ab.txt).
#ifndef SOS_WT
#define SOS_WT 1
#endif
#ifndef SOS_FUSE
#define SOS_FUSE 1
#endif
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

template <int CH> struct Smp { double v[CH]; };

template <int CH> __device__ __forceinline__ Smp<CH> ld(const double* p, long long e) {
    Smp<CH> r;
    if constexpr (CH == 2) {
        const double2 q = *reinterpret_cast<const double2*>(p + 2 * e);
        r.v[0] = q.x; r.v[1] = q.y;
    } else {
        r.v[0] = p[e];
    }
    return r;
}
template <int CH> __device__ __forceinline__ void st(double* p, long long e, const Smp<CH>& s) {
#if SOS_WT
    if constexpr (CH == 2) {
        typedef double d2v __attribute__((ext_vector_type(2)));
        d2v q = {s.v[0], s.v[1]};
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(reinterpret_cast<d2v*>(p + 2 * e)), "v"(q) : "memory");
    } else p[e] = s.v[0];
#else
    if constexpr (CH == 2) *reinterpret_cast<double2*>(p + 2 * e) = make_double2(s.v[0], s.v[1]);
    else p[e] = s.v[0];
#endif
}

// sample i of the pass input of the row starting at `base`
// Branch-free on purpose: the lanes of a wavefront at a row's edge take different paths here, and loads behind divergent
// branches cannot be issued together -- the 12 samples of a chunk became 12 to 36 dependent trips to memory, and the two or four
// edge wavefronts of a launch ran twice as long as all the others: they WERE the kernel's duration (19 us with a bulk that is done
// after 10, profiles/r03_sosfilt.txt).  Both candidate samples are loaded unconditionally (indices clamped into the row), the
// odd extension 2 x[end] - x[mirror] (scipy's odd_ext) is formed and one select picks.
template <int CH> __device__ __forceinline__ Smp<CH> sos_input(const SosPass& p, const double* base, long long i) {
    if (p.backward) {                                                            // reversed forward output (wave-uniform branch)
        long long k = p.mv - 1 - i;
        k = k < 0 ? 0 : (k > p.m - 1 ? p.m - 1 : k);                             // (beyond the end: the virtual copies of y1[m - 1])
        return ld<CH>(base, k);
    }
    const long long j = i - p.edge;                                              // position in the row; outside [0, n): odd extension
    const bool left = j < 0, right = j >= p.n;
    long long ia = left ? 0 : (right ? p.n - 1 : j);                             // the row's end sample, or the sample itself
    long long ib = left ? -j : (right ? 2 * (p.n - 1) - j : j);                  // its mirror image
    ib = ib < 0 ? 0 : (ib > p.n - 1 ? p.n - 1 : ib);                             // (only lanes beyond the padded length get here clamped)
    const Smp<CH> a = ld<CH>(base, ia), b = ld<CH>(base, ib);
    Smp<CH> r;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const double ext = 2.0 * a.v[c] - b.v[c];
        r.v[c] = (left || right) ? ext : a.v[c];
    }
    return r;
}
template <int CH> __device__ __forceinline__ const double* pass_row(const SosPass& p, int row) {
    return p.src + (long long)row * (p.backward ? p.m : p.n) * CH;
}

template <int NS>
__device__ __forceinline__ double sos_step(const SosCoefs& c, double (&z)[NS][2], double x) {
#pragma clang fp contract(off)       // SciPy's C loop rounds every product: keep the same roundings
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const double xn = x;
        x = c.b0[s] * xn + z[s][0];
        z[s][0] = c.b1[s] * xn - c.a1[s] * x + z[s][1];
        z[s][1] = c.b2[s] * xn - c.a2[s] * x;
    }
    return x;
}

// The same step with fused multiply-adds (5 operations per section instead of 9; each product is no longer rounded on its own, so the
// result differs from SciPy's loop in the last bits -- as the chunk start states do anyway).  The one-launch kernel, whose workgroups
// all compute at the same time, is bound by its float64 instruction count.
template <int NS>
__device__ __forceinline__ double sos_step_fma(const SosCoefs& c, double (&z)[NS][2], double x) {
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const double xn = x;
        x = fma(c.b0[s], xn, z[s][0]);
        z[s][0] = fma(c.b1[s], xn, fma(-c.a1[s], x, z[s][1]));
        z[s][1] = fma(c.b2[s], xn, -c.a2[s] * x);
    }
    return x;
}

// LDS staging of a wavefront's 64 x kChunk samples: element r = chunk * kChunk + t lives at r + r / kChunk
// (one pad element per chunk: a lane's ds_read/ds_write of "its" sample t is bank-conflict free).
constexpr int kWaveSamples = kWave * kChunk;
constexpr int kLdsElems = kWaveSamples + kWave;
__device__ __forceinline__ int lds_pos(int r) { return r + r / kChunk; }

// Is the wavefront's sample range [w0, w0 + 64 * kChunk) of the padded sequence plain memory (no odd extension,
// no ragged end)?  Wave-uniform.
__device__ __forceinline__ bool wave_is_plain(const SosPass& p, long long w0) {
    if (p.backward) return w0 >= p.mv - p.m;                  // (past the virtual copies; the grid ends with the sequence)
    return w0 >= p.edge && w0 + kWaveSamples <= p.edge + p.n;
}

// A wavefront exchanges data through ITS slice of LDS only: DS instructions of one wavefront execute in
// order, so no workgroup barrier is needed -- just keep the compiler from reordering across this point.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// the kChunk input samples of the lane's chunk (all loads issued before the recurrence starts)
template <int CH> __device__ __forceinline__ void load_chunk(const SosPass& p, const double* base, long long w0, int lane, int len, bool plain,
                                                              Smp<CH>* lds, Smp<CH> (&xs)[kChunk]) {
    if (plain) {
        // coalesced: instruction t moves samples w0 + 64 t + lane
        Smp<CH> tmp[kChunk];
#pragma unroll
        for (int t = 0; t < kChunk; ++t) {
            const long long i = w0 + t * kWave + lane;
            tmp[t] = ld<CH>(base, p.backward ? p.mv - 1 - i : i - p.edge);
        }
#pragma unroll
        for (int t = 0; t < kChunk; ++t) lds[lds_pos(t * kWave + lane)] = tmp[t];
        wave_lds_sync();
#pragma unroll
        for (int t = 0; t < kChunk; ++t) xs[t] = lds[lane * (kChunk + 1) + t];
        wave_lds_sync();
    } else {
        const long long i0 = w0 + (long long)lane * kChunk;
#pragma unroll
        for (int t = 0; t < kChunk; ++t) xs[t] = sos_input<CH>(p, base, i0 + t);     // (samples beyond `len` are loaded from clamped positions and never used)
        (void)len;
    }
}

// The two halves of load_chunk's plain path, for a caller with work in between that should not have to keep 12 samples per channel in
// registers meanwhile (k_apply's group reduction): coalesced loads -> the wavefront's LDS slice, and later slice -> this lane's chunk.
template <int CH> __device__ __forceinline__ void chunk_to_lds(const SosPass& p, const double* base, long long w0, int lane, Smp<CH>* lds) {
    Smp<CH> tmp[kChunk];
#pragma unroll
    for (int t = 0; t < kChunk; ++t) {
        const long long i = w0 + t * kWave + lane;
        tmp[t] = ld<CH>(base, p.backward ? p.mv - 1 - i : i - p.edge);
    }
#pragma unroll
    for (int t = 0; t < kChunk; ++t) lds[lds_pos(t * kWave + lane)] = tmp[t];
    wave_lds_sync();
}
// The same for a wavefront at a row's edge or end: sample by sample through sos_input (two coalesced loads and a select each) instead of
// lane by lane -- in the one-launch kernel every workgroup waits for the totals of the row's first group.
template <int CH> __device__ __forceinline__ void chunk_to_lds_edge(const SosPass& p, const double* base, long long w0, int lane, Smp<CH>* lds) {
    Smp<CH> tmp[kChunk];
#pragma unroll
    for (int t = 0; t < kChunk; ++t) {
        const long long b = w0 + t * kWave;                          // (wave-uniform: most of the 12 instructions of an edge wavefront are plain)
        if (b >= p.edge && b + kWave <= p.edge + p.n) tmp[t] = ld<CH>(base, b + lane - p.edge);
        else {
            long long i = b + lane;
            i = i < p.mv ? i : p.mv - 1;                          // (beyond the sequence: never used)
            tmp[t] = sos_input<CH>(p, base, i);
        }
    }
#pragma unroll
    for (int t = 0; t < kChunk; ++t) lds[lds_pos(t * kWave + lane)] = tmp[t];
    wave_lds_sync();
}
template <int CH> __device__ __forceinline__ void chunk_from_lds(Smp<CH>* lds, int lane, Smp<CH> (&xs)[kChunk]) {
#pragma unroll
    for (int t = 0; t < kChunk; ++t) xs[t] = lds[lane * (kChunk + 1) + t];
    wave_lds_sync();
}

// y += M x  (M row-major K x K, wave-uniform address).  Every map of this file is a power of the cascade's one-chunk map, and in a
// cascade a section never sees the ones after it: the maps are block lower triangular (2 x 2 blocks, exact zeros above -- the host
// builds them by running the recurrence on unit states), so those products are skipped: 12 of 16 for two sections, 40 of 64 for four.
template <int K, int CH> __device__ __forceinline__ void mat_acc(const double* M, const double (&x)[CH][K], double (&y)[CH][K]) {
#pragma unroll
    for (int r = 0; r < K; ++r)
#pragma unroll
        for (int q = 0; q < K; ++q) {
            if (q / 2 > r / 2) continue;
            const double mrq = M[r * K + q];
#pragma unroll
            for (int c = 0; c < CH; ++c) y[c][r] += mrq * x[c][q];
        }
}

// The scan's doubling steps use M^1, M^2, M^4 ... M^32, the combination of the wavefronts M^64.  Fetching them step by step from
// global memory puts dependent cache misses on the critical path of kernels that only live for microseconds, so a workgroup keeps
// a copy of the seven matrices in LDS (k_chunk_scan: fetched up front by its last wavefront, written after that one's transposes).
constexpr int kScanSteps = 6;
// Inclusive scan over the 64 lanes of a wavefront of the affine maps s -> M s + x_lane (same M in every
// lane): afterwards x_lane = sum_{i <= lane} M^(lane - i) x_i.  lds_pw = the workgroup's LDS copy of the powers.
template <int K, int CH> __device__ __forceinline__ void wave_scan(double (&x)[CH][K], const double* lds_pw) {
    const int lane = threadIdx.x & (kWave - 1);
#pragma unroll
    for (int k = 0; k < kScanSteps; ++k) {
        const int d = 1 << k;
        double up[CH][K];
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
            for (int q = 0; q < K; ++q) {
                const double o = __shfl_up(x[c][q], d, kWave);
                up[c][q] = lane >= d ? o : 0.0;
            }
        mat_acc<K, CH>(lds_pw + k * K * K, up, x);
    }
}
// The same scan in the OPPOSITE lane order (lane 63 first): afterwards x_lane = sum_{i >= lane} M^(i - lane) x_i -- the backward pass's
// chunk order inside the forward pass's wavefront (see SosPass::mv).
template <int K, int CH> __device__ __forceinline__ void wave_scan_rev(double (&x)[CH][K], const double* lds_pw) {
    const int lane = threadIdx.x & (kWave - 1);
#pragma unroll
    for (int k = 0; k < kScanSteps; ++k) {
        const int d = 1 << k;
        double up[CH][K];
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
            for (int q = 0; q < K; ++q) {
                const double o = __shfl_down(x[c][q], d, kWave);
                up[c][q] = lane + d < kWave ? o : 0.0;
            }
        mat_acc<K, CH>(lds_pw + k * K * K, up, x);
    }
}
// x <- M x
template <int K, int CH> __device__ __forceinline__ void mat_apply(const double* M, double (&x)[CH][K]) {
    double y[CH][K];
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int k = 0; k < K; ++k) y[c][k] = 0.0;
    mat_acc<K, CH>(M, x, y);
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int k = 0; k < K; ++k) x[c][k] = y[c][k];
}

// state vector s[CH][K] <-> z[NS][2] per channel
template <int NS, int CH, int W>
__global__ __launch_bounds__(kWave * W) void k_chunk_scan(SosPass p, int nchunks, int ngroups, const double* __restrict__ pw,
                                                       const double* __restrict__ G, double* __restrict__ E, double* __restrict__ T) {
    constexpr int K = 2 * NS;
    constexpr int kWaves = W, kGroup = kWave * W;
    __shared__ Smp<CH> lds_all[kWaves][kLdsElems];
    // The staging slices are ALL the LDS of the kernel: the wavefront totals go into each wavefront's own slice and the matrix powers
    // of the scan into the last wavefront's (offset 1 KiB), both written once the owner has finished its transposes.  52 KiB per
    // workgroup: three fit a CU whatever the allocation granule is (54 272 B did not: a 2^20 x 2 call's 684 workgroups ran as two
    // generations, and a launch took twice a workgroup's lifetime -- profiles/r03_sosfilt.txt).
    static_assert((kScanSteps + 1) * K * K * sizeof(double) + 1024 <= sizeof(Smp<CH>) * kLdsElems, "the powers fit the slice");
    double* const lds_pw = reinterpret_cast<double*>(lds_all[kWaves - 1]) + 128;
    const int row = blockIdx.x / ngroups, g = blockIdx.x % ngroups;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = tid / kWave;
    const int ch = g * kGroup + tid;
    const long long w0 = ((long long)g * kGroup + wv * kWave) * kChunk;
    const bool plain = wave_is_plain(p, w0);
    // (the last wavefront fetches the scan's matrix powers now and writes them to LDS after its transposes)
    constexpr int kPwEntries = (kScanSteps + 1) * K * K, kPwPerLane = (kPwEntries + kWave - 1) / kWave;
    double pw_stage[kPwPerLane];
    if (wv == kWaves - 1) {
#pragma unroll
        for (int i = 0; i < kPwPerLane; ++i) {
            const int e = lane + i * kWave;
            pw_stage[i] = e < kPwEntries ? pw[((long long)1 << (e / (K * K))) * K * K + e % (K * K)] : 0.0;
        }
    }
    // this lane's M^lane, for the offset of its wavefront inside the group (fetched ahead of its use)
    double Ml[K][K];
#pragma unroll
    for (int r = 0; r < K; ++r)
#pragma unroll
        for (int q = 0; q < K; ++q) Ml[r][q] = pw[(long long)lane * K * K + r * K + q];
    double s[CH][K];
#pragma unroll
    for (int a = 0; a < CH; ++a)
#pragma unroll
        for (int k = 0; k < K; ++k) s[a][k] = 0.0;
    if (plain || ch < nchunks) {
        const double* base = pass_row<CH>(p, row);
        const long long i0 = (long long)ch * kChunk;
        const int len = (int)(i0 + kChunk <= p.mv ? kChunk : p.mv - i0);
        Smp<CH> xs[kChunk];
        load_chunk<CH>(p, base, w0, lane, len, plain, lds_all[wv], xs);
        // The chunk's particular solution -- the state after its samples from a ZERO state -- is linear in the samples:
        // p_c = sum_t G[:, t] x_t, G[:, t] = the state kChunk - 1 - t steps after a unit sample (built on the host by
        // running the recurrence on unit samples).  K fma per sample and channel instead of the 9 NS float64 operations
        // of the recurrence itself, which only the output pass (k_apply) has to run.  A short last chunk ends `len`
        // samples in: its sample t has the weight of sample kChunk - len + t of a full one.
        if (len == kChunk) {
#pragma unroll
            for (int t = 0; t < kChunk; ++t)
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const double g = G[k * kChunk + t];
#pragma unroll
                    for (int a = 0; a < CH; ++a) s[a][k] = fma(g, xs[t].v[a], s[a][k]);
                }
        } else {
            const int off = kChunk - len;
#pragma unroll
            for (int t = 0; t < kChunk; ++t)
                if (t < len) {
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        const double g = G[k * kChunk + off + t];
#pragma unroll
                        for (int a = 0; a < CH; ++a) s[a][k] = fma(g, xs[t].v[a], s[a][k]);
                    }
                }
        }
    }
    if (wv == kWaves - 1) {
#pragma unroll
        for (int i = 0; i < kPwPerLane; ++i) {
            const int e = lane + i * kWave;
            if (e < kPwEntries) lds_pw[e] = pw_stage[i];
        }
    }
    __syncthreads();                      // lds_pw staged
    wave_scan<K, CH>(s, lds_pw);
    // exclusive prefix = the chunk's start state if its WAVEFRONT started from zero
    double ex[CH][K];
#pragma unroll
    for (int a = 0; a < CH; ++a)
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const double prev = __shfl_up(s[a][k], 1, kWave);
            ex[a][k] = lane ? prev : 0.0;
            if (lane == kWave - 1) reinterpret_cast<double*>(lds_all[wv])[a * K + k] = s[a][k];
        }
    __syncthreads();
    // start state of this wavefront if the GROUP started from zero: ws_{w+1} = M^64 ws_w + tot_w
    const double* M64 = lds_pw + kScanSteps * K * K;
    double ws[CH][K];
#pragma unroll
    for (int a = 0; a < CH; ++a)
#pragma unroll
        for (int k = 0; k < K; ++k) ws[a][k] = 0.0;
    for (int w = 0; w < wv; ++w) {
        mat_apply<K, CH>(M64, ws);
#pragma unroll
        for (int a = 0; a < CH; ++a)
#pragma unroll
            for (int k = 0; k < K; ++k) ws[a][k] += reinterpret_cast<const double*>(lds_all[w])[a * K + k];
    }
    // e_c = M^lane ws + (wavefront-local exclusive prefix)
#pragma unroll
    for (int r = 0; r < K; ++r)
#pragma unroll
        for (int q = 0; q < K; ++q) {
#pragma unroll
            for (int a = 0; a < CH; ++a) ex[a][r] += Ml[r][q] * ws[a][q];
        }
    double* Eo = E + ((long long)row * ngroups * kGroup + ch) * (CH * K);
#pragma unroll
    for (int a = 0; a < CH; ++a)
#pragma unroll
        for (int k = 0; k < K; ++k) Eo[a * K + k] = ex[a][k];
    if (wv == kWaves - 1) {
        // group total = state after the last wavefront
        mat_apply<K, CH>(M64, ws);
        if (lane == 0) {
            double* To = T + ((long long)row * ngroups + g) * (CH * K);
#pragma unroll
            for (int a = 0; a < CH; ++a)
#pragma unroll
                for (int k = 0; k < K; ++k) To[a * K + k] = ws[a][k] + reinterpret_cast<const double*>(lds_all[kWaves - 1])[a * K + k];
        }
    }
}

// SOS_APPLY_WAVES = 3 asks the register allocator for three workgroups per CU (230 -> 168 registers, 120 bytes of scratch per lane), so
// that a 2^20 x 2 call's 683 workgroups would be resident at once instead of 512 + 171: measured SLOWER, 95.9 against 86.5 us of kernels per
// call (profiles/r03_sosfilt.txt) -- the spills cost more than the second, third-full generation.
#ifndef SOS_APPLY_WAVES
#define SOS_APPLY_WAVES 3
#endif
template <int NS, int CH, int W>
__global__ __launch_bounds__(kWave * W) __attribute__((amdgpu_waves_per_eu(NS <= 2 ? SOS_APPLY_WAVES : 1, 8))) void k_apply(SosCoefs c, SosPass p, int nchunks, int ngroups, const double* __restrict__ zi,
                                                  const double* __restrict__ pw, const double* __restrict__ pwG, const double* __restrict__ pwH,
                                                  const double* __restrict__ E, const double* __restrict__ T,
                                                  double* __restrict__ y1, double* __restrict__ out,
                                                  const double* __restrict__ G, double* __restrict__ E2, double* __restrict__ T2) {
    // E2 != nullptr (forward pass): the kernel also runs the BACKWARD pass's chunk pass on the outputs it holds -- backward chunk c' is
    // this thread's chunk read the other way round (SosPass::mv) -- and leaves its E / T in E2 / T2: no second k_chunk_scan launch, no
    // second read of y1
    constexpr int K = 2 * NS;
    constexpr int kWaves = W, kGroup = kWave * W;
    __shared__ Smp<CH> lds_all[kWaves][kLdsElems];
    __shared__ double lds_m64[K * K];
    __shared__ double red[kWaves][CH * K];
    __shared__ __attribute__((aligned(16))) double c_lds[2];
    const int row = blockIdx.x / ngroups, g = blockIdx.x % ngroups;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = tid / kWave;
    const int ch = g * kGroup + tid;
    const long long w0 = ((long long)g * kGroup + wv * kWave) * kChunk;
    const bool plain = wave_is_plain(p, w0);
    const bool active = plain || ch < nchunks;
    const double* base = pass_row<CH>(p, row);
    for (int e = tid; e < K * K; e += kGroup) lds_m64[e] = pw[(long long)kWave * K * K + e];
    // operands fetched ahead of the samples: e_c, this lane's M^lane and (M^256)^lane
    double s[CH][K], Ml[K][K], MG[K][K];
    const double* Eo = E + ((long long)row * ngroups * kGroup + ch) * (CH * K);
#pragma unroll
    for (int a = 0; a < CH; ++a)
#pragma unroll
        for (int k = 0; k < K; ++k) s[a][k] = Eo[a * K + k];
#pragma unroll
    for (int r = 0; r < K; ++r)
#pragma unroll
        for (int q = 0; q < K; ++q) { Ml[r][q] = pw[(long long)lane * K * K + r * K + q]; MG[r][q] = pwG[(long long)lane * K * K + r * K + q]; }
    const long long i0 = (long long)ch * kChunk;
    const int len = (int)(i0 + kChunk <= p.mv ? kChunk : p.mv - i0);
    // the samples: a plain wavefront parks them in its LDS slice until the group's start state is known (they would be 48 registers
    // through the whole reduction: 230 in all, two workgroups per CU, two generations for a 2^20 x 2 call); an edge wavefront keeps them
    // (an edge wavefront loads lane by lane and parks its samples the same way)
    if (active) {
        if (plain) chunk_to_lds<CH>(p, base, w0, lane, lds_all[wv]);
        else {
            Smp<CH> xe[kChunk];
            load_chunk<CH>(p, base, w0, lane, len, false, lds_all[wv], xe);
#pragma unroll
            for (int t = 0; t < kChunk; ++t) lds_all[wv][lane * (kChunk + 1) + t] = xe[t];
            wave_lds_sync();
        }
    }

    // ---- start state of the group: S_g = sum over distances d = 0..g of (M^256)^d v_d,
    //      v_d = T_{g-1-d} for d < g, v_g = s_0 = zi * x_0.  d = 64 a + lane: lane power, then wavefront power.
    double acc[CH][K];
#pragma unroll
    for (int a = 0; a < CH; ++a)
#pragma unroll
        for (int k = 0; k < K; ++k) acc[a][k] = 0.0;
    for (int d0 = 0; d0 <= g; d0 += kGroup) {
        const int d = d0 + tid;
        double v[CH][K], u[CH][K];
        if (d < g) {
            const double* Tj = T + ((long long)row * ngroups + (g - 1 - d)) * (CH * K);
#pragma unroll
            for (int a = 0; a < CH; ++a)
#pragma unroll
                for (int k = 0; k < K; ++k) v[a][k] = Tj[a * K + k];
        } else if (d == g) {
            const Smp<CH> u0 = sos_input<CH>(p, base, 0);
#pragma unroll
            for (int a = 0; a < CH; ++a)
#pragma unroll
                for (int k = 0; k < K; ++k) v[a][k] = zi[k] * u0.v[a];          // zi * x_0 (sosfiltfilt)
        } else {
#pragma unroll
            for (int a = 0; a < CH; ++a)
#pragma unroll
                for (int k = 0; k < K; ++k) v[a][k] = 0.0;
        }
#pragma unroll
        for (int a = 0; a < CH; ++a)
#pragma unroll
            for (int r = 0; r < K; ++r) {
                double t = 0.0;
#pragma unroll
                for (int q = 0; q < K; ++q) t += MG[r][q] * v[a][q];
                u[a][r] = t;
            }
#pragma unroll
        for (int o = kWave / 2; o > 0; o >>= 1)
#pragma unroll
            for (int a = 0; a < CH; ++a)
#pragma unroll
                for (int k = 0; k < K; ++k) u[a][k] += __shfl_xor(u[a][k], o, kWave);
        if (d0 + wv * kWave <= g) mat_acc<K, CH>(pwH + (long long)(d0 / kWave + wv) * K * K, u, acc);      // wave-uniform
    }
    if (lane == 0) {
#pragma unroll
        for (int a = 0; a < CH; ++a)
#pragma unroll
            for (int k = 0; k < K; ++k) red[wv][a * K + k] = acc[a][k];
    }
    __syncthreads();
    double sg[CH][K];
#pragma unroll
    for (int a = 0; a < CH; ++a)
#pragma unroll
        for (int k = 0; k < K; ++k) {
            double t = red[0][a * K + k];
#pragma unroll
            for (int w = 1; w < kWaves; ++w) t += red[w][a * K + k];
            sg[a][k] = t;
        }
    const bool fuse = E2 != nullptr;
    if (!active && !fuse) return;
    Smp<CH> ys[kChunk];                    // this chunk's outputs (kept for the fused chunk pass)
    Smp<CH>* lds = lds_all[wv];
    if (active) {
    Smp<CH> xs[kChunk];
    chunk_from_lds<CH>(lds_all[wv], lane, xs);
    // true start state of the chunk = M^lane (M^64)^wv S_g + e_c
    for (int w = 0; w < wv; ++w) mat_apply<K, CH>(lds_m64, sg);
#pragma unroll
    for (int r = 0; r < K; ++r)
#pragma unroll
        for (int q = 0; q < K; ++q) {
#pragma unroll
            for (int a = 0; a < CH; ++a) s[a][r] += Ml[r][q] * sg[a][q];
        }
    double z[CH][NS][2];
#pragma unroll
    for (int a = 0; a < CH; ++a)
#pragma unroll
        for (int q = 0; q < NS; ++q) { z[a][q][0] = s[a][2 * q]; z[a][q][1] = s[a][2 * q + 1]; }
    double* const dst_f = y1 + (long long)row * p.m * CH;
    double* const dst_b = out + (long long)row * p.n * CH;
    if (plain) {
        // outputs back through LDS, then coalesced stores
#pragma unroll
        for (int t = 0; t < kChunk; ++t) {
#pragma unroll
            for (int a = 0; a < CH; ++a) ys[t].v[a] = sos_step<NS>(c, z[a], xs[t].v[a]);
            lds[lane * (kChunk + 1) + t] = ys[t];
        }
        wave_lds_sync();
#pragma unroll
        for (int t = 0; t < kChunk; ++t) {
            const Smp<CH> y = lds[lds_pos(t * kWave + lane)];
            const long long i = w0 + t * kWave + lane;
            if (!p.backward) st<CH>(dst_f, i, y);
            else {
                // y = reverse(y2)[edge : m - edge]  ->  out[nn] = y2 at forward position nn + edge
                const long long nn = p.mv - 1 - i - p.edge;
                if (nn >= 0 && nn < p.n) st<CH>(dst_b, nn, y);
            }
        }
        wave_lds_sync();
    } else {
#pragma unroll
        for (int t = 0; t < kChunk; ++t)
            if (t < len) {
#pragma unroll
                for (int a = 0; a < CH; ++a) ys[t].v[a] = sos_step<NS>(c, z[a], xs[t].v[a]);
                if (!p.backward) st<CH>(dst_f, i0 + t, ys[t]);
                else {
                    const long long nn = p.mv - 1 - (i0 + t) - p.edge;
                    if (nn >= 0 && nn < p.n) st<CH>(dst_b, nn, ys[t]);
                }
            }
    }
    }
    if (!fuse) return;

    // ---- the backward pass's chunk pass (k_chunk_scan's work) on the outputs of this chunk, read from its end to its start.
    // Positions at and beyond m hold y1[m - 1] (SosPass::mv): the thread that produced it shares it; all such chunks are in the last group.
    const long long last = p.m - 1;
    if (g == ngroups - 1) {
        if ((long long)ch * kChunk <= last && last < (long long)ch * kChunk + kChunk) {
            Smp<CH> cv = ys[0];
#pragma unroll
            for (int t = 1; t < kChunk; ++t) if (i0 + t == last) cv = ys[t];
#pragma unroll
            for (int a = 0; a < CH; ++a) c_lds[a] = cv.v[a];
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < kChunk; ++t)
            if (i0 + t > last) {
#pragma unroll
                for (int a = 0; a < CH; ++a) ys[t].v[a] = c_lds[a];
            }
    }
    // the scan's matrix powers: the last wavefront copies them into its slice (free now), offset 1 KiB
    constexpr int kPwEntries = (kScanSteps + 1) * K * K;
    double* const lds_pw = reinterpret_cast<double*>(lds_all[kWaves - 1]) + 128;
    if (wv == kWaves - 1)
        for (int e = lane; e < kPwEntries; e += kWave) lds_pw[e] = pw[((long long)1 << (e / (K * K))) * K * K + e % (K * K)];
    // this lane's power for the offset of its wavefront inside the group: its backward position in the wavefront is 63 - lane
    double Mb[K][K];
#pragma unroll
    for (int r = 0; r < K; ++r)
#pragma unroll
        for (int q = 0; q < K; ++q) Mb[r][q] = pw[(long long)(kWave - 1 - lane) * K * K + r * K + q];
    double sb[CH][K];
#pragma unroll
    for (int a = 0; a < CH; ++a)
#pragma unroll
        for (int k = 0; k < K; ++k) sb[a][k] = 0.0;
#pragma unroll
    for (int t = 0; t < kChunk; ++t)
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const double gk = G[k * kChunk + t];
#pragma unroll
            for (int a = 0; a < CH; ++a) sb[a][k] = fma(gk, ys[kChunk - 1 - t].v[a], sb[a][k]);
        }
    __syncthreads();                      // lds_pw staged
    wave_scan_rev<K, CH>(sb, lds_pw);
    double exb[CH][K];
    double* const totb = reinterpret_cast<double*>(lds_all[wv]);          // (the wavefront's total, in its own slice)
#pragma unroll
    for (int a = 0; a < CH; ++a)
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const double prev = __shfl_down(sb[a][k], 1, kWave);
            exb[a][k] = lane != kWave - 1 ? prev : 0.0;
            if (lane == 0) totb[a * K + k] = sb[a][k];
        }
    __syncthreads();
    // backward order of the wavefronts: the last one comes first
    double wsb[CH][K];
#pragma unroll
    for (int a = 0; a < CH; ++a)
#pragma unroll
        for (int k = 0; k < K; ++k) wsb[a][k] = 0.0;
    for (int b = 0; b < kWaves - 1 - wv; ++b) {
        mat_apply<K, CH>(lds_m64, wsb);
        const double* tb = reinterpret_cast<const double*>(lds_all[kWaves - 1 - b]);
#pragma unroll
        for (int a = 0; a < CH; ++a)
#pragma unroll
            for (int k = 0; k < K; ++k) wsb[a][k] += tb[a * K + k];
    }
#pragma unroll
    for (int r = 0; r < K; ++r)
#pragma unroll
        for (int q = 0; q < K; ++q) {
#pragma unroll
            for (int a = 0; a < CH; ++a) exb[a][r] += Mb[r][q] * wsb[a][q];
        }
    // backward chunk index of this thread: (chunks in the grid) - 1 - ch; backward group: ngroups - 1 - g
    const long long chb = (long long)ngroups * kGroup - 1 - ch;
    double* Eb = E2 + ((long long)row * ngroups * kGroup + chb) * (CH * K);
#pragma unroll
    for (int a = 0; a < CH; ++a)
#pragma unroll
        for (int k = 0; k < K; ++k) Eb[a * K + k] = exb[a][k];
    if (wv == 0) {
        // group total = state after the last wavefront in backward order (this one)
        mat_apply<K, CH>(lds_m64, wsb);
        if (lane == 0) {
            double* Tb = T2 + ((long long)row * ngroups + (ngroups - 1 - g)) * (CH * K);
            const double* tb = reinterpret_cast<const double*>(lds_all[0]);
#pragma unroll
            for (int a = 0; a < CH; ++a)
#pragma unroll
                for (int k = 0; k < K; ++k) Tb[a * K + k] = wsb[a][k] + tb[a * K + k];
        }
    }
}

// ------------------------------------------------------------------------------------------ one launch
// The whole forward-backward filter in ONE launch, for calls whose workgroups are all resident at once (host: `one_launch_capacity`).
// A workgroup keeps its 256 chunks in registers / LDS from the first load to the last store and meets the others twice, through
// the group totals in HBM: forward it needs the totals of every EARLIER group of its row, backward those of every LATER one.  Each
// total goes out with a flag (= the call's epoch; the flags are never reset); a reader polls the flags of the totals it needs with
// agent-scope loads.  x is read once, the result written once -- 16 B per real sample, the algorithmic minimum, against 51 of the
// three-launch form (no y1, no chunk states in HBM at all).
// With every workgroup in the same phase at the same time nothing hides a trip to memory, and while the samples stream in or out
// such a trip takes microseconds: between the sample loads and the sample stores the kernel touches global memory only for the
// totals -- the scan's matrix powers live in LDS for the whole launch (in the pad elements of the staging slices, which the sample
// traffic never touches), the per-lane powers are asked for ahead of the waits.
// A workgroup that waits longer than `patience` (another process's grid sharing the GPU could keep part of this one out) raises
// `*status` and leaves; the host then repeats the call with the three-launch form (x is never written).
struct SosLink {
    double* T;            // [2][gridDim.x][CH * K]: forward totals (+ the start state's image in group 0's), then backward
    unsigned* flag;       // [2][gridDim.x]
    int* status;          // host-visible
    unsigned epoch;
    long long patience;   // ticks of the 100 MHz clock
    long long* tl;        // SOS_TIMELINE builds: [gridDim.x][16] clock readings of a workgroup's phases (dev aid), else unused
};
#ifndef SOS_TIMELINE
#define SOS_TIMELINE 0
#endif
#if SOS_TIMELINE
#define SOS_MARK(i) do { if (threadIdx.x == 0 && L.tl) L.tl[(long long)blockIdx.x * 16 + (i)] = wall_clock64(); } while (0)
#else
#define SOS_MARK(i) do { } while (0)
#endif
#ifndef SOS_AHEAD_LARGE
#define SOS_AHEAD_LARGE 0
#endif
constexpr int kLookIter = 2;      // a workgroup looks back over at most kLookIter x (its threads) groups: the one-launch form's limit on groups per row

// No fences in the hand-over: an agent-scope release / acquire is a write-back / invalidate of the XCD's whole L2 on gfx950
// (buffer_wbl2 / buffer_inv sc1), and with 684 workgroups doing both twice a 2^20 x 2 call took 141 us, its workgroups 12 to 55 us
// from "samples loaded" to "total published" (profiles/r03_sosfilt.txt).  Instead the totals are written through (sc1 stores), the
// writer waits for their acknowledgement before it raises the flag, and readers fetch them with sc1 loads, which go past the caches.
typedef double sos_d2v __attribute__((ext_vector_type(2)));

// Doubles `e` of the workgroup's LDS-resident table (scan powers M^1 .. M^32, M^64, then the wavefronts' carries and totals): pad
// element number e / CH of the staging slices (slice-major), component e % CH.  Compile-time `e` -> an immediate offset.
template <int CH> struct PadTable {
    Smp<CH>* slices;       // lds_all[0]
    __device__ __forceinline__ double& operator[](int e) const {
        const int pad = e / CH;
        return slices[(pad / kWave) * kLdsElems + (pad % kWave) * (kChunk + 1) + kChunk].v[e % CH];
    }
};
template <int K, int CH, class Tab> __device__ __forceinline__ void mat_acc_t(const Tab& M, int base, const double (&x)[CH][K], double (&y)[CH][K]) {
#pragma unroll
    for (int r = 0; r < K; ++r)
#pragma unroll
        for (int q = 0; q < K; ++q) {
            if (q / 2 > r / 2) continue;
            const double mrq = M[base + r * K + q];
#pragma unroll
            for (int c = 0; c < CH; ++c) y[c][r] += mrq * x[c][q];
        }
}
template <int K, int CH, class Tab> __device__ __forceinline__ void mat_apply_t(const Tab& M, int base, double (&x)[CH][K]) {
    double y[CH][K];
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int k = 0; k < K; ++k) y[c][k] = 0.0;
    mat_acc_t<K, CH>(M, base, x, y);
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int k = 0; k < K; ++k) x[c][k] = y[c][k];
}
// wave_scan / wave_scan_rev with the powers in a PadTable
template <int K, int CH, bool REV, class Tab> __device__ __forceinline__ void wave_scan_t(double (&x)[CH][K], const Tab& pwt) {
    const int lane = threadIdx.x & (kWave - 1);
#pragma unroll
    for (int k = 0; k < kScanSteps; ++k) {
        const int d = 1 << k;
        double up[CH][K];
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
            for (int q = 0; q < K; ++q) up[c][q] = REV ? __shfl_down(x[c][q], d, kWave) : __shfl_up(x[c][q], d, kWave);
        if (REV ? lane + d < kWave : lane >= d) mat_acc_t<K, CH>(pwt, k * K * K, up, x);       // (the others have no partner: one branch instead of 2 N selects)
    }
}

// acc = start state of group number `g` of its pass = sum_{d < g} (M^group)^d T'_{g-1-d}; T'_j = the published totals of the row
// (`Tb`, `Fb`: the row's totals and flags in the pass's own group order).  Waits for each of them: all flags first, then all totals
// in one round of loads.
template <int K, int CH, int W>
__device__ __forceinline__ void group_start(const SosLink& L, const double* Tb, const unsigned* Fb, int g, const double* __restrict__ pwG,
                                            const double* __restrict__ pwH, double (&acc)[CH][K], int* s_fail) {
    constexpr int kGroup = kWave * W, N = CH * K;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = __builtin_amdgcn_readfirstlane(tid / kWave);
    double MG[K][K];
#pragma unroll
    for (int r = 0; r < K; ++r)
#pragma unroll
        for (int q = 0; q < K; ++q) MG[r][q] = pwG[(long long)lane * K * K + r * K + q];
#pragma unroll
    for (int a = 0; a < CH; ++a)
#pragma unroll
        for (int k = 0; k < K; ++k) acc[a][k] = 0.0;
    const long long t0 = wall_clock64();
#pragma unroll
    for (int it = 0; it < kLookIter; ++it) {
        const int d = it * kGroup + tid;
        if (d < g) {
            while (__hip_atomic_load(Fb + (g - 1 - d), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != L.epoch) {
                if (wall_clock64() - t0 > L.patience) { *s_fail = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
    }
    // the totals: 16-byte sc1 buffer loads (compiler-tracked; the flat-address form of an sc1 load only exists as inline assembly, whose
    // result registers the compiler would be free to copy or spill before the data is there)
    typedef unsigned int sos_u4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(Tb), 0, g * N * (int)sizeof(double), 0x00020000);
    sos_d2v q[kLookIter][N / 2];
#pragma unroll
    for (int it = 0; it < kLookIter; ++it) {
        const int d = it * kGroup + tid;
        const int off = d < g ? (g - 1 - d) * N * (int)sizeof(double) : 0x7ffffff0;        // (beyond the descriptor's range: the load returns zeros)
        if (it * kGroup < g) {                       // (uniform)
#pragma unroll
            for (int i = 0; i < N / 2; ++i) {
                const sos_u4 raw = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 16 * i, 0, 16 /* sc1 */);
                q[it][i] = __builtin_bit_cast(sos_d2v, raw);
            }
        }
    }
    // per lane: (M^group)^(64 (it W + wv)) (M^group)^lane T' for each round, summed; ONE reduction over the wavefront at the end
#pragma unroll
    for (int it = 0; it < kLookIter; ++it) {
        if (it * kGroup >= g) break;                 // (uniform)
        double u[CH][K], vv[CH][K];
#pragma unroll
        for (int a = 0; a < CH; ++a)
#pragma unroll
            for (int qq = 0; qq < K; ++qq) {
                const int e = a * K + qq;
                vv[a][qq] = e % 2 ? q[it][e / 2].y : q[it][e / 2].x;
                u[a][qq] = 0.0;
            }
        mat_acc<K, CH>(&MG[0][0], vv, u);
        if (it * kGroup + wv * kWave < g) mat_acc<K, CH>(pwH + (long long)(it * W + wv) * K * K, u, acc);      // wave-uniform, scalar loads
    }
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1)
#pragma unroll
        for (int a = 0; a < CH; ++a)
#pragma unroll
            for (int k = 0; k < K; ++k) acc[a][k] += __shfl_xor(acc[a][k], o, kWave);
}

template <int NS, int CH, int W>
__global__ __launch_bounds__(kWave * W) __attribute__((amdgpu_waves_per_eu(NS <= 2 && W != kWavesSmall ? SOS_APPLY_WAVES : 1, 8))) void k_filtfilt(
    SosCoefs c, SosPass p, int nchunks, int ngroups, const double* __restrict__ zi, const double* __restrict__ pw,
    const double* __restrict__ pwG, const double* __restrict__ pwH, const double* __restrict__ G, double* __restrict__ out, SosLink L) {
    constexpr int K = 2 * NS, N = CH * K;
    constexpr int kWaves = W, kGroup = kWave * W;
    constexpr int kPwEntries = (kScanSteps + 1) * K * K;       // M^1, M^2 .. M^32, M^64
    constexpr int kM64 = kScanSteps * K * K, kCarry = kPwEntries, kTot = kCarry + kWaves * N, kG = kTot + kWaves * N, kTabSize = kG + K * kChunk;
    constexpr bool kPads = kTabSize <= kWave * kWaves * CH;       // the table fits the slices' pad elements
    constexpr bool kAhead = W == kWavesSmall || SOS_AHEAD_LARGE;
    __shared__ Smp<CH> lds_all[kWaves][kLdsElems];
    __shared__ double red[kWaves][N];
    __shared__ __attribute__((aligned(16))) double c_lds[2];
    __shared__ __attribute__((aligned(16))) int s_fail[4];
    __shared__ __attribute__((aligned(16))) double v0_lds[N];            // the pass's start state zi * (first sample), in the group that holds it
    __shared__ __attribute__((aligned(16))) double tab_lds[kPads ? 2 : kTabSize];
    struct FlatTable { double* t; __device__ __forceinline__ double& operator[](int e) const { return t[e]; } };
    const PadTable<CH> padt{&lds_all[0][0]};
    const FlatTable flatt{tab_lds};
    const auto tabv = [&](int e) -> double& { if constexpr (kPads) return padt[e]; else return flatt[e]; };
    struct TabRef { decltype(tabv) f; __device__ __forceinline__ double& operator[](int e) const { return f(e); } };
    const TabRef tab{tabv};
    const int row = blockIdx.x / ngroups, g = blockIdx.x % ngroups, gb = ngroups - 1 - g;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int ch = g * kGroup + tid;
    const long long w0 = ((long long)g * kGroup + wv * kWave) * kChunk;
    const bool plain = wave_is_plain(p, w0);
    const bool active = plain || ch < nchunks;
    const double* base = pass_row<CH>(p, row);
    const long long i0 = (long long)ch * kChunk;
    const int len = (int)(i0 + kChunk <= p.mv ? kChunk : p.mv - i0);
    Smp<CH>* const lds = lds_all[wv];
    double* const Tf = L.T + (long long)row * ngroups * N;
    double* const Tbk = L.T + ((long long)gridDim.x + (long long)row * ngroups) * N;
    unsigned* const Ff = L.flag + (long long)row * ngroups;
    unsigned* const Fbk = L.flag + (long long)gridDim.x + (long long)row * ngroups;
    if (tid == 0) s_fail[0] = 0;
    SOS_MARK(0);
    // the scan's powers -> the LDS table (the pads are nobody else's: no barrier needed before the samples use the slices)
    for (int e0 = tid; e0 < kPwEntries + K * kChunk; e0 += kGroup) {
        const bool is_g = e0 >= kPwEntries;
        const int e = is_g ? kG + (e0 - kPwEntries) : e0;
        const double val = is_g ? G[e0 - kPwEntries] : pw[((long long)1 << (e0 / (K * K))) * K * K + e0 % (K * K)];
        if constexpr (kPads) {
            const int pad = e / CH;
            lds_all[0][(pad / kWave) * kLdsElems + (pad % kWave) * (kChunk + 1) + kChunk].v[e % CH] = val;
        } else tab_lds[e] = val;
    }
    // the image of the pass's start state zi * x_0 under one group map goes out with group 0's total; group 0 itself starts from it
    if (g == 0 && tid == 0) {
        const Smp<CH> u0 = sos_input<CH>(p, base, 0);
#pragma unroll
        for (int a = 0; a < CH; ++a)
#pragma unroll
            for (int k = 0; k < K; ++k) v0_lds[a * K + k] = zi[k] * u0.v[a];
    }

    // ---- forward chunk pass (k_chunk_scan's work; the chunk states stay in registers)
    Smp<CH> xs[kChunk];
    double ex[CH][K];                     // the chunk's start state if its WAVEFRONT started from zero
    {
        double s[CH][K];
#pragma unroll
        for (int a = 0; a < CH; ++a)
#pragma unroll
            for (int k = 0; k < K; ++k) s[a][k] = 0.0;
        if (plain || g * kGroup + wv * kWave < nchunks) {          // (the whole wavefront moves its 768 samples, whichever lanes hold chunks)
            if (plain) chunk_to_lds<CH>(p, base, w0, lane, lds);
            else chunk_to_lds_edge<CH>(p, base, w0, lane, lds);
            chunk_from_lds<CH>(lds, lane, xs);
        }
        SOS_MARK(1);
        __syncthreads();                      // table, s_fail, v0 staged
        if (active) {
            if (len == kChunk) {
#pragma unroll
                for (int t = 0; t < kChunk; ++t)
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        const double gk = tab[kG + k * kChunk + t];
#pragma unroll
                        for (int a = 0; a < CH; ++a) s[a][k] = fma(gk, xs[t].v[a], s[a][k]);
                    }
            } else {
                const int off = kChunk - len;
#pragma unroll
                for (int t = 0; t < kChunk; ++t)
                    if (t < len) {
#pragma unroll
                        for (int k = 0; k < K; ++k) {
                            const double gk = tab[kG + k * kChunk + off + t];
#pragma unroll
                            for (int a = 0; a < CH; ++a) s[a][k] = fma(gk, xs[t].v[a], s[a][k]);
                        }
                    }
            }
        }
        SOS_MARK(2);
        wave_scan_t<K, CH, false>(s, tab);
#pragma unroll
        for (int a = 0; a < CH; ++a)
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const double prev = __shfl_up(s[a][k], 1, kWave);
                ex[a][k] = lane ? prev : 0.0;
                if (lane == kWave - 1) tab[kTot + wv * N + a * K + k] = s[a][k];
            }
        __syncthreads();
        SOS_MARK(3);
        // start state of this wavefront if the GROUP started from zero: ws_{w+1} = M^64 ws_w + tot_w.  Kept in the table (wave-uniform).
        double ws[CH][K];
#pragma unroll
        for (int a = 0; a < CH; ++a)
#pragma unroll
            for (int k = 0; k < K; ++k) ws[a][k] = 0.0;
        for (int w = 0; w < wv; ++w) {
            mat_apply_t<K, CH>(tab, kM64, ws);
#pragma unroll
            for (int a = 0; a < CH; ++a)
#pragma unroll
                for (int k = 0; k < K; ++k) ws[a][k] += tab[kTot + w * N + a * K + k];
        }
        if (lane == 0) {
#pragma unroll
            for (int a = 0; a < CH; ++a)
#pragma unroll
                for (int k = 0; k < K; ++k) tab[kCarry + wv * N + a * K + k] = ws[a][k];
        }
        if (wv == kWaves - 1 && g + 1 < ngroups) {
            mat_apply_t<K, CH>(tab, kM64, ws);
#pragma unroll
            for (int a = 0; a < CH; ++a)
#pragma unroll
                for (int k = 0; k < K; ++k) ws[a][k] += tab[kTot + (kWaves - 1) * N + a * K + k];
            if (g == 0) {
                double v0[CH][K];
#pragma unroll
                for (int a = 0; a < CH; ++a)
#pragma unroll
                    for (int k = 0; k < K; ++k) v0[a][k] = v0_lds[a * K + k];
                mat_acc<K, CH>(pwG + K * K, v0, ws);
            }
            if (lane == 0) {
#pragma unroll
                for (int a = 0; a < CH; ++a)
#pragma unroll
                    for (int k = 0; k < K; ++k) __hip_atomic_store(Tf + (long long)g * N + a * K + k, ws[a][k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    SOS_MARK(4);
    // the samples wait in the wavefront's slice while the group's start state is formed
    if (active) {
#pragma unroll
        for (int t = 0; t < kChunk; ++t) lds[lane * (kChunk + 1) + t] = xs[t];
        wave_lds_sync();
    }
    // (the flag follows the total once that is acknowledged: by now it mostly is)
    if (wv == kWaves - 1 && g + 1 < ngroups) {
        asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
        if (lane == 0) __hip_atomic_store(Ff + g, L.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    SOS_MARK(5);
    // this lane's M^lane, for the offset of its chunk inside the wavefront: asked for ahead of the wait where the registers allow it
    // (the two-wavefront shape of short calls, which are pure latency chains), behind it otherwise (three workgroups per CU: 168 registers)
    double Ml[K][K];
    if constexpr (kAhead) {
#pragma unroll
        for (int r = 0; r < K; ++r)
#pragma unroll
            for (int q = 0; q < K; ++q) Ml[r][q] = pw[(long long)lane * K * K + r * K + q];
    }
    double sg[CH][K];
    {
        double acc[CH][K];
        group_start<K, CH, W>(L, Tf, Ff, g, pwG, pwH, acc, s_fail);
        if (lane == 0) {
#pragma unroll
            for (int a = 0; a < CH; ++a)
#pragma unroll
                for (int k = 0; k < K; ++k) red[wv][a * K + k] = acc[a][k];
        }
    }
    if constexpr (!kAhead) {
#pragma unroll
        for (int r = 0; r < K; ++r)
#pragma unroll
            for (int q = 0; q < K; ++q) Ml[r][q] = pw[(long long)lane * K * K + r * K + q];
    }
    __syncthreads();
    if (s_fail[0]) {
        if (tid == 0) __hip_atomic_store(L.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
#pragma unroll
    for (int a = 0; a < CH; ++a)
#pragma unroll
        for (int k = 0; k < K; ++k) {
            double t = red[0][a * K + k];
#pragma unroll
            for (int w = 1; w < kWaves; ++w) t += red[w][a * K + k];
            sg[a][k] = g == 0 ? v0_lds[a * K + k] : t;
        }
    SOS_MARK(6);
    // ---- forward output pass: the chunk from its true start state M^lane ((M^64)^wv S_g + ws) + ex; the outputs replace the samples
    for (int w = 0; w < wv; ++w) mat_apply_t<K, CH>(tab, kM64, sg);
#pragma unroll
    for (int a = 0; a < CH; ++a)
#pragma unroll
        for (int k = 0; k < K; ++k) sg[a][k] += tab[kCarry + wv * N + a * K + k];
    if (active) {
        chunk_from_lds<CH>(lds, lane, xs);
#pragma unroll
        for (int r = 0; r < K; ++r)
#pragma unroll
            for (int q = 0; q < K; ++q) {
#pragma unroll
                for (int a = 0; a < CH; ++a) ex[a][r] += Ml[r][q] * sg[a][q];
            }
        SOS_MARK(13);
        double z[CH][NS][2];
#pragma unroll
        for (int a = 0; a < CH; ++a)
#pragma unroll
            for (int q = 0; q < NS; ++q) { z[a][q][0] = ex[a][2 * q]; z[a][q][1] = ex[a][2 * q + 1]; }
        if (len == kChunk) {
#pragma unroll
            for (int t = 0; t < kChunk; ++t)
#pragma unroll
                for (int a = 0; a < CH; ++a) xs[t].v[a] = sos_step_fma<NS>(c, z[a], xs[t].v[a]);
        } else {
#pragma unroll
            for (int t = 0; t < kChunk; ++t)
                if (t < len) {
#pragma unroll
                    for (int a = 0; a < CH; ++a) xs[t].v[a] = sos_step_fma<NS>(c, z[a], xs[t].v[a]);
                }
        }
    }
    SOS_MARK(7);
    // ---- backward chunk pass on the outputs, read from the chunk's end to its start (SosPass::mv).  Positions at and beyond m hold
    // y1[m - 1]: the thread that produced it shares it; all such chunks are in the last group.
    const long long last = p.m - 1;
    if (gb == 0) {
        if (i0 <= last && last < i0 + kChunk) {
            Smp<CH> cv = xs[0];
#pragma unroll
            for (int t = 1; t < kChunk; ++t) if (i0 + t == last) cv = xs[t];
#pragma unroll
            for (int a = 0; a < CH; ++a) c_lds[a] = cv.v[a];
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < kChunk; ++t)
            if (i0 + t > last) {
#pragma unroll
                for (int a = 0; a < CH; ++a) xs[t].v[a] = c_lds[a];
            }
        if (tid == 0) {
#pragma unroll
            for (int a = 0; a < CH; ++a)
#pragma unroll
                for (int k = 0; k < K; ++k) v0_lds[a * K + k] = zi[k] * c_lds[a];
        }
    }
    double exb[CH][K];
    {
        double sb[CH][K];
#pragma unroll
        for (int a = 0; a < CH; ++a)
#pragma unroll
            for (int k = 0; k < K; ++k) sb[a][k] = 0.0;
#pragma unroll
        for (int t = 0; t < kChunk; ++t)
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const double gk = tab[kG + k * kChunk + t];
#pragma unroll
                for (int a = 0; a < CH; ++a) sb[a][k] = fma(gk, xs[kChunk - 1 - t].v[a], sb[a][k]);
            }
        wave_scan_t<K, CH, true>(sb, tab);
#pragma unroll
        for (int a = 0; a < CH; ++a)
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const double prev = __shfl_down(sb[a][k], 1, kWave);
                exb[a][k] = lane != kWave - 1 ? prev : 0.0;
                if (lane == 0) tab[kTot + wv * N + a * K + k] = sb[a][k];
            }
        __syncthreads();                      // (also: v0_lds of the backward pass staged, every wavefront past its use of the forward carries)
        SOS_MARK(8);
        // backward order of the wavefronts: the last one comes first
        double wsb[CH][K];
#pragma unroll
        for (int a = 0; a < CH; ++a)
#pragma unroll
            for (int k = 0; k < K; ++k) wsb[a][k] = 0.0;
        for (int b = 0; b < kWaves - 1 - wv; ++b) {
            mat_apply_t<K, CH>(tab, kM64, wsb);
#pragma unroll
            for (int a = 0; a < CH; ++a)
#pragma unroll
                for (int k = 0; k < K; ++k) wsb[a][k] += tab[kTot + (kWaves - 1 - b) * N + a * K + k];
        }
        if (lane == 0) {
#pragma unroll
            for (int a = 0; a < CH; ++a)
#pragma unroll
                for (int k = 0; k < K; ++k) tab[kCarry + wv * N + a * K + k] = wsb[a][k];
        }
        if (wv == 0 && gb + 1 < ngroups) {
            mat_apply_t<K, CH>(tab, kM64, wsb);
#pragma unroll
            for (int a = 0; a < CH; ++a)
#pragma unroll
                for (int k = 0; k < K; ++k) wsb[a][k] += tab[kTot + a * K + k];
            if (gb == 0) {
                double v0[CH][K];
#pragma unroll
                for (int a = 0; a < CH; ++a)
#pragma unroll
                    for (int k = 0; k < K; ++k) v0[a][k] = v0_lds[a * K + k];
                mat_acc<K, CH>(pwG + K * K, v0, wsb);
            }
            if (lane == 0) {
#pragma unroll
                for (int a = 0; a < CH; ++a)
#pragma unroll
                    for (int k = 0; k < K; ++k) __hip_atomic_store(Tbk + (long long)gb * N + a * K + k, wsb[a][k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < kChunk; ++t) lds[lane * (kChunk + 1) + t] = xs[t];
    wave_lds_sync();
    if (wv == 0 && gb + 1 < ngroups) {
        asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
        if (lane == 0) __hip_atomic_store(Fbk + gb, L.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    SOS_MARK(9);
    // this lane's power for the offset of its chunk inside the wavefront: its backward position there is 63 - lane
    double Mb[K][K];
    if constexpr (kAhead) {
#pragma unroll
        for (int r = 0; r < K; ++r)
#pragma unroll
            for (int q = 0; q < K; ++q) Mb[r][q] = pw[(long long)(kWave - 1 - lane) * K * K + r * K + q];
    }
    double sgb[CH][K];
    {
        double acc[CH][K];
        group_start<K, CH, W>(L, Tbk, Fbk, gb, pwG, pwH, acc, s_fail);
        if (lane == 0) {
#pragma unroll
            for (int a = 0; a < CH; ++a)
#pragma unroll
                for (int k = 0; k < K; ++k) red[wv][a * K + k] = acc[a][k];
        }
    }
    if constexpr (!kAhead) {
#pragma unroll
        for (int r = 0; r < K; ++r)
#pragma unroll
            for (int q = 0; q < K; ++q) Mb[r][q] = pw[(long long)(kWave - 1 - lane) * K * K + r * K + q];
    }
    __syncthreads();
    if (s_fail[0]) {
        if (tid == 0) __hip_atomic_store(L.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
#pragma unroll
    for (int a = 0; a < CH; ++a)
#pragma unroll
        for (int k = 0; k < K; ++k) {
            double t = red[0][a * K + k];
#pragma unroll
            for (int w = 1; w < kWaves; ++w) t += red[w][a * K + k];
            sgb[a][k] = gb == 0 ? v0_lds[a * K + k] : t;
        }
    SOS_MARK(10);
    // ---- backward output pass: the chunk from its end, start state M^(63 - lane) ((M^64)^(W - 1 - wv) S_gb + wsb) + exb
    for (int w = 0; w < kWaves - 1 - wv; ++w) mat_apply_t<K, CH>(tab, kM64, sgb);
#pragma unroll
    for (int a = 0; a < CH; ++a)
#pragma unroll
        for (int k = 0; k < K; ++k) sgb[a][k] += tab[kCarry + wv * N + a * K + k];
    chunk_from_lds<CH>(lds, lane, xs);
#pragma unroll
    for (int r = 0; r < K; ++r)
#pragma unroll
        for (int q = 0; q < K; ++q) {
#pragma unroll
            for (int a = 0; a < CH; ++a) exb[a][r] += Mb[r][q] * sgb[a][q];
        }
    {
        double z[CH][NS][2];
#pragma unroll
        for (int a = 0; a < CH; ++a)
#pragma unroll
            for (int q = 0; q < NS; ++q) { z[a][q][0] = exb[a][2 * q]; z[a][q][1] = exb[a][2 * q + 1]; }
#pragma unroll
        for (int t = kChunk - 1; t >= 0; --t) {
#pragma unroll
            for (int a = 0; a < CH; ++a) xs[t].v[a] = sos_step_fma<NS>(c, z[a], xs[t].v[a]);
            lds[lane * (kChunk + 1) + t] = xs[t];
        }
    }
    wave_lds_sync();
    SOS_MARK(11);
    // the result sits where the sample was: padded position k -> out[k - edge]
    double* const dst = out + (long long)row * p.n * CH;
#pragma unroll
    for (int t = 0; t < kChunk; ++t) {
        const Smp<CH> y = lds[lds_pos(t * kWave + lane)];
        const long long nn = w0 + t * kWave + lane - p.edge;
        if (nn >= 0 && nn < p.n) st<CH>(dst, nn, y);
    }
    SOS_MARK(12);
}

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
template <int NS, int W> void build_tables(const SosCoefs& c, std::vector<double>& tab) {
    constexpr int K = 2 * NS;
    constexpr int kWaves = W;
    double M[K * K];
    // column q of A^kChunk = state after kChunk zero-input steps from the unit state e_q
    for (int q = 0; q < K; ++q) {
        double z[NS][2];
        for (int s = 0; s < NS; ++s) z[s][0] = z[s][1] = 0.0;
        z[q / 2][q % 2] = 1.0;
        for (int i = 0; i < kChunk; ++i) {
            double x = 0.0;
            for (int s = 0; s < NS; ++s) {
                const double xn = x;
                x = c.b0[s] * xn + z[s][0];
                z[s][0] = c.b1[s] * xn - c.a1[s] * x + z[s][1];
                z[s][1] = c.b2[s] * xn - c.a2[s] * x;
            }
        }
        for (int r = 0; r < K; ++r) M[r * K + q] = z[r / 2][r % 2];
    }
    // G[k][t]: component k of the state at the end of a chunk whose only non-zero sample is a 1 at position t
    std::vector<double> G((size_t)K * kChunk, 0.0);
    for (int t = 0; t < kChunk; ++t) {
        double z[NS][2];
        for (int s = 0; s < NS; ++s) z[s][0] = z[s][1] = 0.0;
        for (int i = t; i < kChunk; ++i) {
            double x = i == t ? 1.0 : 0.0;
            for (int s = 0; s < NS; ++s) {
                const double xn = x;
                x = c.b0[s] * xn + z[s][0];
                z[s][0] = c.b1[s] * xn - c.a1[s] * x + z[s][1];
                z[s][1] = c.b2[s] * xn - c.a2[s] * x;
            }
        }
        for (int k = 0; k < K; ++k) G[(size_t)k * kChunk + t] = z[k / 2][k % 2];
    }
    tab.assign((size_t)3 * (kWave + 1) * K * K, 0.0);
    for (int level = 0; level < 3; ++level) {
        double* P = tab.data() + (size_t)level * (kWave + 1) * K * K;
        for (int r = 0; r < K; ++r) P[r * K + r] = 1.0;
        for (int j = 0; j < kWave; ++j) matmul<K>(M, P + (size_t)j * K * K, P + (size_t)(j + 1) * K * K);
        std::memcpy(M, P + (size_t)kWave * K * K, sizeof(M));                 // M^64 of this level
        if (level == 0) {                                                     // group map = (M^64)^kWaves
            double A[K * K], B[K * K];
            std::memcpy(A, M, sizeof(M));
            for (int w = 1; w < kWaves; ++w) { matmul<K>(M, A, B); std::memcpy(A, B, sizeof(A)); }
            std::memcpy(M, A, sizeof(M));
        }
    }
    tab.insert(tab.end(), G.begin(), G.end());                 // (behind the three levels of matrix powers)
}

#define WS_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(SSFM_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); } while (0)

// ---- the one-launch form's conditions
constexpr int kMaxGiveUps = 3, kRestCalls = 1000;
// SSFM_SOS_ONE_LAUNCH=0: always three launches (read per call)
inline bool one_launch_enabled() {
    const char* e = std::getenv("SSFM_SOS_ONE_LAUNCH");
    return !(e && std::atoi(e) == 0);
}
// how long a workgroup waits for a total before it gives the call up, in 10 ns ticks (SSFM_SOS_PATIENCE_US, default 2 ms)
inline long long one_launch_patience() {
    if (const char* e = std::getenv("SSFM_SOS_PATIENCE_US")) { const long long v = std::atoll(e); if (v > 0) return v * 100; }
    return 200000;
}
// Workgroups of k_filtfilt the device holds at once.  The runtime's occupancy answer is cut to what the LDS allows with the 512-byte
// allocation granule and the 161 280 usable bytes measured on gfx950 (three workgroups of 53 760 B fit a CU, three of 54 272 B do not:
// profiles/r03_sosfilt.txt) -- a grid that waits for workgroups that cannot start would only end by its patience.
template <int NS, int CH, int W> long long one_launch_capacity() {
    static long long cap = -1;                                  // (calls are serialised by Workspace::mu)
    if (cap >= 0) return cap;
    int per_cu = 0, cus = 0, dev = 0;
    hipFuncAttributes fa;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_filtfilt<NS, CH, W>, kWave * W, 0) != hipSuccess ||
        hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(k_filtfilt<NS, CH, W>)) != hipSuccess) {
        (void)hipGetLastError();
        return cap = 0;
    }
    const long long lds = ((long long)fa.sharedSizeBytes + 511) / 512 * 512;
    const long long by_lds = lds > 0 ? 161280 / lds : per_cu;
    if (const char* e = std::getenv("SSFM_SOS_ONE_LAUNCH_CAP")) return cap = std::atoll(e);      // (tests: force the three-launch form by size)
    return cap = (long long)cus * (per_cu < by_lds ? per_cu : by_lds);
}

template <int NS, int CH, int W>
int run_filter_w(Workspace& w, const SosCoefs& c, const double* sos_key, const double* zi_h, const double* x, double* y,
               long long n, int rows, int edge, bool on_device) {
    constexpr int K = 2 * NS;
    constexpr int kGroup = kWave * W;
    const long long m = n + 2ll * edge;
    const long long nchunks_ll = (m + kChunk - 1) / kChunk;
    if (nchunks_ll > (1ll << 30)) return fail(SSFM_ERR_UNSUPPORTED, "ssfm_sosfiltfilt: n=%lld too long", n);
    const int nchunks = (int)nchunks_ll;
    const int ngroups = (nchunks + kGroup - 1) / kGroup;
    if (ngroups > kWave * kWave)
        return fail(SSFM_ERR_UNSUPPORTED, "ssfm_sosfiltfilt: n=%lld exceeds %d samples per row", n, kWave * kWave * kGroup * kChunk - 2 * edge);
    if ((long long)ngroups * rows > 0x7fffffffll) return fail(SSFM_ERR_UNSUPPORTED, "ssfm_sosfiltfilt: %d rows of %lld samples exceed the grid", rows, n);
    const size_t xbytes = sizeof(double) * (size_t)n * rows * CH;
    if (!w.stream) {
        WS_TRY(hipStreamCreateWithFlags(&w.stream, hipStreamNonBlocking));
        WS_TRY(hipEventCreate(&w.ev0));
        WS_TRY(hipEventCreate(&w.ev1));
    }
    if (!on_device) WS_TRY(w.need(0, xbytes));
    WS_TRY(w.need(1, sizeof(double) * (size_t)m * rows * CH));
    WS_TRY(w.need(2, sizeof(double) * (size_t)ngroups * kGroup * rows * CH * K));
    WS_TRY(w.need(3, sizeof(double) * (size_t)ngroups * rows * CH * K * 2));           // T of the forward and of the backward pass
    WS_TRY(w.need(4, sizeof(double) * (size_t)ngroups * kGroup * rows * CH * K));       // E of the backward pass (written by the forward output kernel)
    const size_t pow_doubles = (size_t)3 * (kWave + 1) * K * K;
    const size_t tab_doubles = pow_doubles + (size_t)K * kChunk;          // matrix powers, then G
    WS_TRY(w.need(5, sizeof(double) * (tab_doubles + K)));
    // tables + zi: rebuilt only when the filter changes
    std::vector<double> key(sos_key, sos_key + 6 * NS);
    key.insert(key.end(), zi_h, zi_h + K);
    key.push_back((double)W);                              // the group map is (M^64)^W
    if (key != w.table_key) {
        std::vector<double> tab;
        build_tables<NS, W>(c, tab);
        tab.insert(tab.end(), zi_h, zi_h + K);
        WS_TRY(hipMemcpyAsync(w.buf[5], tab.data(), sizeof(double) * tab.size(), hipMemcpyHostToDevice, w.stream));
        WS_TRY(hipStreamSynchronize(w.stream));            // `tab` is a local
        w.table_key = key;
    }
    const double* d_pw = w.buf[5];
    const double* d_pwG = w.buf[5] + (size_t)(kWave + 1) * K * K;
    const double* d_pwH = w.buf[5] + (size_t)2 * (kWave + 1) * K * K;
    const double* d_G = w.buf[5] + pow_doubles;
    const double* d_zi = w.buf[5] + tab_doubles;
    const double* d_x = x;
    double* d_out = y;
    if (!on_device) {
        WS_TRY(hipMemcpyAsync(w.buf[0], x, xbytes, hipMemcpyHostToDevice, w.stream));
        d_x = w.buf[0];
        d_out = w.buf[0];          // the backward pass writes the trimmed result over the staged input
    }
    SosPass p;
    p.n = n; p.m = m; p.edge = edge;
    const dim3 grid((unsigned)(ngroups * rows)), block(kGroup);
    // ---- one launch, when the whole grid is resident at once.  The result never lands on the input (a call that gives up is repeated
    // from it): an in-place call's goes to the scratch buffer first and is copied over the input afterwards.
    const bool overlap = on_device && !(reinterpret_cast<const char*>(y) + xbytes <= reinterpret_cast<const char*>(x) ||
                                        reinterpret_cast<const char*>(x) + xbytes <= reinterpret_cast<const char*>(y));
    if (w.give_ups >= kMaxGiveUps && ++w.rested >= kRestCalls) { w.give_ups = kMaxGiveUps - 1; w.rested = 0; }
    if (one_launch_enabled() && w.give_ups < kMaxGiveUps && ngroups <= kLookIter * kGroup &&
        (long long)grid.x <= one_launch_capacity<NS, CH, W>()) {
        const size_t tdoubles = (size_t)2 * grid.x * CH * K;
        if (tdoubles > w.link_T_cap) {
            if (w.link_T) (void)hipFree(w.link_T);
            w.link_T = nullptr; w.link_T_cap = 0;
            WS_TRY(hipMalloc(&w.link_T, sizeof(double) * tdoubles));
            w.link_T_cap = tdoubles;
        }
        if ((size_t)2 * grid.x > w.link_flag_cap) {
            if (w.link_flag) (void)hipFree(w.link_flag);
            w.link_flag = nullptr; w.link_flag_cap = 0;
            WS_TRY(hipMalloc(&w.link_flag, sizeof(unsigned) * 2 * grid.x));
            WS_TRY(hipMemsetAsync(w.link_flag, 0, sizeof(unsigned) * 2 * grid.x, w.stream));
            w.link_flag_cap = (size_t)2 * grid.x;
            w.epoch = 0;
        }
        if (!w.status) WS_TRY(hipHostMalloc(&w.status, 64, hipHostMallocMapped));
        if (++w.epoch == 0) {                                  // (the counter wrapped: old flags could match again)
            WS_TRY(hipMemsetAsync(w.link_flag, 0, sizeof(unsigned) * w.link_flag_cap, w.stream));
            w.epoch = 1;
        }
        *w.status = 0;
        SosLink L;
        L.T = w.link_T; L.flag = w.link_flag; L.status = w.status; L.epoch = w.epoch; L.patience = one_launch_patience();
        L.tl = nullptr;
#if SOS_TIMELINE
        static long long* d_tl = nullptr; static size_t tl_cap = 0;
        if (std::getenv("SOS_TIMELINE_DUMP")) {
            if (tl_cap < grid.x) { if (d_tl) (void)hipFree(d_tl); WS_TRY(hipMalloc(&d_tl, sizeof(long long) * 16 * grid.x)); tl_cap = grid.x; }
            L.tl = d_tl;
        }
#endif
        double* const d_res = (!on_device || overlap) ? w.buf[1] : y;
        p.backward = 0; p.src = d_x; p.mv = m;
        WS_TRY(hipEventRecord(w.ev0, w.stream));
        hipLaunchKernelGGL((k_filtfilt<NS, CH, W>), grid, block, 0, w.stream, c, p, nchunks, ngroups, d_zi, d_pw, d_pwG, d_pwH, d_G, d_res, L);
        WS_TRY(hipGetLastError());
        WS_TRY(hipEventRecord(w.ev1, w.stream));
        if (!on_device) WS_TRY(hipMemcpyAsync(y, d_res, xbytes, hipMemcpyDeviceToHost, w.stream));
        WS_TRY(hipStreamSynchronize(w.stream));
        if (*w.status == 0 && on_device && overlap) {
            WS_TRY(hipMemcpyAsync(y, d_res, xbytes, hipMemcpyDeviceToDevice, w.stream));
            WS_TRY(hipStreamSynchronize(w.stream));
        }
#if SOS_TIMELINE
        if (L.tl) {
            // per phase: when the first, the median and the last workgroup got there, in us after the first workgroup's start; then a few workgroups' own lines
            std::vector<long long> h((size_t)16 * grid.x);
            WS_TRY(hipMemcpy(h.data(), L.tl, sizeof(long long) * h.size(), hipMemcpyDeviceToHost));
            long long t0 = h[0];
            for (unsigned b = 0; b < grid.x; ++b) t0 = h[(size_t)b * 16] < t0 ? h[(size_t)b * 16] : t0;
            static const char* names[14] = {"start", "loaded", "chunk sums", "scanned", "total stored", "flag raised", "forward start state", "forward outputs",
                                            "backward scanned", "backward flag raised", "backward start state", "backward outputs", "end", "(forward chunk state)"};
            constexpr int kMarks = 14;
            for (int i = 0; i < kMarks; ++i) {
                std::vector<double> v;
                for (unsigned b = 0; b < grid.x; ++b) v.push_back((h[(size_t)b * 16 + i] - t0) * 0.01);
                std::sort(v.begin(), v.end());
                std::fprintf(stderr, "%2d %-22s first %6.2f  median %6.2f  last %6.2f us\n", i, names[i], v.front(), v[v.size() / 2], v.back());
            }
            for (unsigned b = 0; b < grid.x; b += grid.x / 6 ? grid.x / 6 : 1) {
                std::fprintf(stderr, "  workgroup %4u:", b);
                for (int i = 0; i < kMarks; ++i) std::fprintf(stderr, " %6.2f", (h[(size_t)b * 16 + i] - t0) * 0.01);
                std::fprintf(stderr, "\n");
            }
        }
#endif
        if (*w.status == 0) {
            WS_TRY(hipEventElapsedTime(&w.last_ms, w.ev0, w.ev1));
            w.last_launches = 1;
            w.give_ups = 0;
            return SSFM_OK;
        }
        ++w.give_ups;                                          // part of the grid never ran beside the rest: three launches from the untouched input
    }
    double *d_y1 = w.buf[1], *d_E = w.buf[2], *d_T = w.buf[3];
    w.last_launches = 3;
    WS_TRY(hipEventRecord(w.ev0, w.stream));
    // One stream: splitting the rows over two streams (as the propagator does) was measured SLOWER here
    // (145 vs 121 us for 2 x 2^20 complex) -- every kernel is a short latency chain, not a bandwidth phase.
    // Three launches per call: chunk pass (forward), output pass (forward) + chunk pass (backward) in one kernel, output pass (backward).
    // SOS_FUSE=0 (build knob) keeps the backward chunk pass a launch of its own.
    double *d_E2 = w.buf[4], *d_T2 = w.buf[3] + (size_t)ngroups * rows * CH * K;
    p.backward = 0; p.src = d_x; p.mv = m;
    hipLaunchKernelGGL((k_chunk_scan<NS, CH, W>), grid, block, 0, w.stream, p, nchunks, ngroups, d_pw, d_G, d_E, d_T);
    hipLaunchKernelGGL((k_apply<NS, CH, W>), grid, block, 0, w.stream, c, p, nchunks, ngroups, d_zi, d_pw, d_pwG, d_pwH, (const double*)d_E,
                       (const double*)d_T, d_y1, d_out, d_G, SOS_FUSE ? d_E2 : nullptr, SOS_FUSE ? d_T2 : nullptr);
    p.backward = 1; p.src = d_y1; p.mv = (long long)ngroups * kGroup * kChunk;
    if (!SOS_FUSE) hipLaunchKernelGGL((k_chunk_scan<NS, CH, W>), grid, block, 0, w.stream, p, ngroups * kGroup, ngroups, d_pw, d_G, d_E2, d_T2);
    hipLaunchKernelGGL((k_apply<NS, CH, W>), grid, block, 0, w.stream, c, p, ngroups * kGroup, ngroups, d_zi, d_pw, d_pwG, d_pwH, (const double*)d_E2,
                       (const double*)d_T2, d_y1, d_out, d_G, (double*)nullptr, (double*)nullptr);
    WS_TRY(hipGetLastError());
    WS_TRY(hipEventRecord(w.ev1, w.stream));
    if (!on_device) WS_TRY(hipMemcpyAsync(y, w.buf[0], xbytes, hipMemcpyDeviceToHost, w.stream));
    WS_TRY(hipStreamSynchronize(w.stream));
    WS_TRY(hipEventElapsedTime(&w.last_ms, w.ev0, w.ev1));
    return SSFM_OK;
}

// the workgroup shape follows the size of the call (see kWavesSmall)
inline int sos_waves(long long n, int rows, int edge) {
    const long long chunks = (n + 2ll * edge + kChunk - 1) / kChunk * rows;
    if (const char* e = std::getenv("SOS_WAVES_FORCE")) { const int v = std::atoi(e); if (v == kWavesSmall || v == kWavesLarge) return v; }
    return chunks <= kSmallChunks ? kWavesSmall : kWavesLarge;
}
template <int NS, int CH>
int run_filter(Workspace& w, const SosCoefs& c, const double* sos_key, const double* zi_h, const double* x, double* y,
               long long n, int rows, int edge, bool on_device) {
    if (kWavesSmall != kWavesLarge && sos_waves(n, rows, edge) == kWavesSmall)
        return run_filter_w<NS, CH, kWavesSmall>(w, c, sos_key, zi_h, x, y, n, rows, edge, on_device);
    return run_filter_w<NS, CH, kWavesLarge>(w, c, sos_key, zi_h, x, y, n, rows, edge, on_device);
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
                                int64_t n, int batch, int is_complex) {
    return sosfiltfilt_impl(device, sos, zi, n_sections, x, y, n, batch, is_complex, false);
}

extern "C" int ssfm_sosfiltfilt_device(int device, const double* sos, const double* zi, int n_sections, const void* x_dev, void* y_dev,
                                       int64_t n, int batch, int is_complex) {
    return sosfiltfilt_impl(device, sos, zi, n_sections, x_dev, y_dev, n, batch, is_complex, true);
}

extern "C" int ssfm_sosfiltfilt_last_launches(int device, int* launches) {
    if (!launches) return fail(SSFM_ERR_INVALID, "launches is NULL");
    if (device < 0 || device >= kMaxDevices) return fail(SSFM_ERR_NO_DEVICE, "ssfm_sosfiltfilt_last_launches: device %d", device);
    std::lock_guard<std::mutex> lock(g_ws[device].mu);
    *launches = g_ws[device].last_launches;
    return SSFM_OK;
}

extern "C" int ssfm_sosfiltfilt_last_ms(int device, float* ms) {
    if (!ms) return fail(SSFM_ERR_INVALID, "ms is NULL");
    if (device < 0 || device >= kMaxDevices) return fail(SSFM_ERR_NO_DEVICE, "ssfm_sosfiltfilt_last_ms: device %d", device);
    std::lock_guard<std::mutex> lock(g_ws[device].mu);
    *ms = g_ws[device].last_ms;
    return SSFM_OK;
}
