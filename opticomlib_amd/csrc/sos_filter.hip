// sos_filter.hip -- zero-phase IIR filtering (cascade of second-order sections, forward-backward) on
// gfx950: the arithmetic behind the reference's LPF / BPF (devices.py:1363-1368, :814-823), i.e.
// scipy.signal.sosfiltfilt(sos, x, axis=-1) with its default odd padding of 3*ntaps samples and
// steady-state initial conditions.
//
// A cascade of NS biquads in direct form II transposed is a linear recurrence with a K = 2*NS state.
// It is parallelised over time by chunking:
//   k_chunk   every thread runs the recurrence over ITS chunk of L consecutive samples from a ZERO
//             state and keeps the final state p_c (the chunk's particular solution);
//   k_scan_*  the chunk start states s_{c+1} = A^L s_c + p_c by a two-level scan (groups of 64 chunks);
//             A^L, the K x K homogeneous map over L samples, is built once on the host by running the
//             same recurrence on unit states, A^(64 L) by six squarings;
//   k_apply   every thread re-runs its chunk from the true start state and writes the outputs.
// Inside a chunk the operation order is exactly SciPy's sample loop; only the chunk start states see a
// different summation order (relative 1e-16 effects).  Complex rows are two independent real channels
// (real coefficients), addressed with stride 2.  float64 throughout, like SciPy.
#include <hip/hip_runtime.h>

#include <cstring>
#include <vector>

#include "ssfm_common.hpp"

using ssfm::fail;

namespace {

constexpr int kChunk = 256;        // samples per thread
constexpr int kMaxSections = 4;    // Bessel orders up to 8

struct SosCoefs {
    double b0[kMaxSections], b1[kMaxSections], b2[kMaxSections], a1[kMaxSections], a2[kMaxSections];
};

struct SosPass {
    const double* src;      // forward: caller's x (first element of row 0); backward: y1 buffer
    long long row_pitch;    // elements between consecutive virtual rows of src
    long long stride;       // element stride inside a row (1 real, 2 complex)
    long long n;            // original samples per row
    long long m;            // padded length n + 2*edge
    int edge;
    int backward;
};

// sample i of the pass input for a virtual row starting at `base`
__device__ __forceinline__ double sos_input(const SosPass& p, const double* base, long long i) {
    if (p.backward) return base[p.m - 1 - i];                                   // reversed forward output
    if (i < p.edge) return 2.0 * base[0] - base[(p.edge - i) * p.stride];       // odd extension, left
    if (i < p.edge + p.n) return base[(i - p.edge) * p.stride];
    const long long r = i - p.edge - p.n;                                       // odd extension, right
    return 2.0 * base[(p.n - 1) * p.stride] - base[(p.n - 2 - r) * p.stride];
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

// virtual row v -> base pointer of the pass input
__device__ __forceinline__ const double* row_base(const SosPass& p, int v, int chan_per_row) {
    if (p.backward) return p.src + (long long)v * p.row_pitch;
    return p.src + (long long)(v / chan_per_row) * p.row_pitch + (v % chan_per_row);
}

template <int NS>
__global__ void k_chunk(SosCoefs c, SosPass p, int nchunks, int vrows, int chan, double* __restrict__ pfinal) {
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (long long)nchunks * vrows) return;
    const int v = (int)(g / nchunks), ch = (int)(g % nchunks);
    const double* base = row_base(p, v, chan);
    double z[NS][2];
#pragma unroll
    for (int s = 0; s < NS; ++s) z[s][0] = z[s][1] = 0.0;
    const long long i0 = (long long)ch * kChunk;
    const long long i1 = i0 + kChunk < p.m ? i0 + kChunk : p.m;
    for (long long i = i0; i < i1; ++i) sos_step<NS>(c, z, sos_input(p, base, i));
    double* out = pfinal + g * (2 * NS);
#pragma unroll
    for (int s = 0; s < NS; ++s) { out[2 * s] = z[s][0]; out[2 * s + 1] = z[s][1]; }
}

// Chunk start states by a two-level scan of the affine maps s -> A s + p_c (A = A^kChunk):
//   k_scan_group   thread per (row, group of kGroup chunks): the group's zero-state response
//   k_scan_top     thread per row: walks the groups with AG = A^kGroup (n_chunks / kGroup steps)
//   k_scan_starts  thread per (row, group): re-walks its chunks from the group's true start state
// so the longest serial chain is 2*kGroup + n_chunks/kGroup matrix-vector steps instead of n_chunks.
constexpr int kGroup = 64;

template <int K> __device__ __forceinline__ void affine_step(const double (&A)[K][K], double (&s)[K], const double* __restrict__ p) {
    double t[K];
#pragma unroll
    for (int r = 0; r < K; ++r) {
        double acc = p[r];
#pragma unroll
        for (int q = 0; q < K; ++q) acc += A[r][q] * s[q];
        t[r] = acc;
    }
#pragma unroll
    for (int k = 0; k < K; ++k) s[k] = t[k];
}
template <int K> __device__ __forceinline__ void load_mat(double (&A)[K][K], const double* __restrict__ M) {
#pragma unroll
    for (int r = 0; r < K; ++r)
#pragma unroll
        for (int q = 0; q < K; ++q) A[r][q] = M[r * K + q];
}

template <int NS>
__global__ void k_scan_group(int nchunks, int ngroups, int vrows, const double* __restrict__ Amat,
                             const double* __restrict__ pfinal, double* __restrict__ gfinal) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ngroups * vrows) return;
    constexpr int K = 2 * NS;
    const int v = g / ngroups, gi = g % ngroups;
    double A[K][K], s[K];
    load_mat<K>(A, Amat);
#pragma unroll
    for (int k = 0; k < K; ++k) s[k] = 0.0;
    const int c1 = (gi + 1) * kGroup < nchunks ? (gi + 1) * kGroup : nchunks;
    for (int ch = gi * kGroup; ch < c1; ++ch) affine_step<K>(A, s, pfinal + ((long long)v * nchunks + ch) * K);
#pragma unroll
    for (int k = 0; k < K; ++k) gfinal[(long long)g * K + k] = s[k];
}

template <int NS>
__global__ void k_scan_top(SosPass p, int ngroups, int vrows, int chan, const double* __restrict__ zi,
                           const double* __restrict__ AGmat, const double* __restrict__ gfinal, double* __restrict__ gstart) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= vrows) return;
    constexpr int K = 2 * NS;
    const double u0 = sos_input(p, row_base(p, v, chan), 0);
    double AG[K][K], s[K];
    load_mat<K>(AG, AGmat);
#pragma unroll
    for (int k = 0; k < K; ++k) s[k] = zi[k] * u0;                  // zi * x_0 (sosfiltfilt)
    for (int gi = 0; gi < ngroups; ++gi) {
        double* st = gstart + ((long long)v * ngroups + gi) * K;
#pragma unroll
        for (int k = 0; k < K; ++k) st[k] = s[k];
        affine_step<K>(AG, s, gfinal + ((long long)v * ngroups + gi) * K);
    }
}

template <int NS>
__global__ void k_scan_starts(int nchunks, int ngroups, int vrows, const double* __restrict__ Amat,
                              const double* __restrict__ pfinal, const double* __restrict__ gstart, double* __restrict__ start) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ngroups * vrows) return;
    constexpr int K = 2 * NS;
    const int v = g / ngroups, gi = g % ngroups;
    double A[K][K], s[K];
    load_mat<K>(A, Amat);
#pragma unroll
    for (int k = 0; k < K; ++k) s[k] = gstart[(long long)g * K + k];
    const int c1 = (gi + 1) * kGroup < nchunks ? (gi + 1) * kGroup : nchunks;
    for (int ch = gi * kGroup; ch < c1; ++ch) {
        double* st = start + ((long long)v * nchunks + ch) * K;
#pragma unroll
        for (int k = 0; k < K; ++k) st[k] = s[k];
        affine_step<K>(A, s, pfinal + ((long long)v * nchunks + ch) * K);
    }
}

template <int NS>
__global__ void k_apply(SosCoefs c, SosPass p, int nchunks, int vrows, int chan, const double* __restrict__ start,
                        double* __restrict__ y1, double* __restrict__ out, long long out_pitch, long long out_stride) {
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (long long)nchunks * vrows) return;
    const int v = (int)(g / nchunks), ch = (int)(g % nchunks);
    const double* base = row_base(p, v, chan);
    double z[NS][2];
    const double* st = start + g * (2 * NS);
#pragma unroll
    for (int s = 0; s < NS; ++s) { z[s][0] = st[2 * s]; z[s][1] = st[2 * s + 1]; }
    const long long i0 = (long long)ch * kChunk;
    const long long i1 = i0 + kChunk < p.m ? i0 + kChunk : p.m;
    if (!p.backward) {
        double* dst = y1 + (long long)v * p.m;
        for (long long i = i0; i < i1; ++i) dst[i] = sos_step<NS>(c, z, sos_input(p, base, i));
    } else {
        // y = reverse(y2)[edge : m - edge]  ->  out[n] = y2[m - 1 - (n + edge)]
        double* dst = out + (long long)(v / chan) * out_pitch + (v % chan);
        for (long long i = i0; i < i1; ++i) {
            const double y = sos_step<NS>(c, z, sos_input(p, base, i));
            const long long nn = p.m - 1 - i - p.edge;
            if (nn >= 0 && nn < p.n) dst[nn * out_stride] = y;
        }
    }
}

template <int NS>
int run_filter(const SosCoefs& c, const double* zi_h, const double* x_h, double* y_h, long long n, int batch, int chan, int edge) {
    constexpr int K = 2 * NS;
    const long long m = n + 2ll * edge;
    const int vrows = batch * chan;
    const int nchunks = (int)((m + kChunk - 1) / kChunk);
    // A^kChunk: column q = state after kChunk zero-input steps from unit state e_q (host, same recurrence)
    double Amat[K * K];
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
        for (int r = 0; r < K; ++r) Amat[r * K + q] = z[r / 2][r % 2];
    }
    // AG = A^kGroup: the homogeneous map over a whole group of chunks (kGroup = 2^6: six squarings)
    double AG[K * K];
    std::memcpy(AG, Amat, sizeof(AG));
    for (int it = 0; (1 << it) < kGroup; ++it) {
        double T2[K * K];
        for (int r = 0; r < K; ++r)
            for (int q = 0; q < K; ++q) {
                double acc = 0.0;
                for (int m2 = 0; m2 < K; ++m2) acc += AG[r * K + m2] * AG[m2 * K + q];
                T2[r * K + q] = acc;
            }
        std::memcpy(AG, T2, sizeof(AG));
    }
    const int ngroups = (nchunks + kGroup - 1) / kGroup;
    const size_t xbytes = sizeof(double) * (size_t)n * vrows;
    double *d_x = nullptr, *d_y1 = nullptr, *d_pf = nullptr, *d_st = nullptr, *d_zi = nullptr, *d_A = nullptr, *d_AG = nullptr, *d_gf = nullptr, *d_gs = nullptr;
    auto cleanup = [&]() { void* b[] = {d_x, d_y1, d_pf, d_st, d_zi, d_A, d_AG, d_gf, d_gs}; for (void* q : b) (void)hipFree(q); };
#define TRY_OR_CLEAN(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { cleanup(); return fail(SSFM_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); } } while (0)
    TRY_OR_CLEAN(hipMalloc(&d_x, xbytes));
    TRY_OR_CLEAN(hipMalloc(&d_y1, sizeof(double) * (size_t)m * vrows));
    TRY_OR_CLEAN(hipMalloc(&d_pf, sizeof(double) * (size_t)nchunks * vrows * K));
    TRY_OR_CLEAN(hipMalloc(&d_st, sizeof(double) * (size_t)nchunks * vrows * K));
    TRY_OR_CLEAN(hipMalloc(&d_zi, sizeof(double) * K));
    TRY_OR_CLEAN(hipMalloc(&d_A, sizeof(double) * K * K));
    TRY_OR_CLEAN(hipMalloc(&d_AG, sizeof(double) * K * K));
    TRY_OR_CLEAN(hipMalloc(&d_gf, sizeof(double) * (size_t)ngroups * vrows * K));
    TRY_OR_CLEAN(hipMalloc(&d_gs, sizeof(double) * (size_t)ngroups * vrows * K));
    TRY_OR_CLEAN(hipMemcpy(d_AG, AG, sizeof(double) * K * K, hipMemcpyHostToDevice));
    TRY_OR_CLEAN(hipMemcpy(d_x, x_h, xbytes, hipMemcpyHostToDevice));
    TRY_OR_CLEAN(hipMemcpy(d_zi, zi_h, sizeof(double) * K, hipMemcpyHostToDevice));
    TRY_OR_CLEAN(hipMemcpy(d_A, Amat, sizeof(double) * K * K, hipMemcpyHostToDevice));
    const long long work = (long long)nchunks * vrows;
    const dim3 gw((unsigned)((work + 127) / 128)), bw(128);
    SosPass p;
    p.n = n; p.m = m; p.edge = edge;
    for (int dir = 0; dir < 2; ++dir) {
        p.backward = dir;
        if (dir == 0) { p.src = d_x; p.row_pitch = n * chan; p.stride = chan; }
        else          { p.src = d_y1; p.row_pitch = m; p.stride = 1; }
        hipLaunchKernelGGL(k_chunk<NS>, gw, bw, 0, 0, c, p, nchunks, vrows, chan, d_pf);
        const dim3 gg((unsigned)((ngroups * vrows + 63) / 64)), bg(64);
        hipLaunchKernelGGL(k_scan_group<NS>, gg, bg, 0, 0, nchunks, ngroups, vrows, (const double*)d_A, (const double*)d_pf, d_gf);
        hipLaunchKernelGGL(k_scan_top<NS>, dim3((vrows + 63) / 64), dim3(64), 0, 0, p, ngroups, vrows, chan, (const double*)d_zi, (const double*)d_AG, (const double*)d_gf, d_gs);
        hipLaunchKernelGGL(k_scan_starts<NS>, gg, bg, 0, 0, nchunks, ngroups, vrows, (const double*)d_A, (const double*)d_pf, (const double*)d_gs, d_st);
        // the backward pass writes the trimmed, re-reversed result over the input buffer
        hipLaunchKernelGGL(k_apply<NS>, gw, bw, 0, 0, c, p, nchunks, vrows, chan, (const double*)d_st, d_y1, d_x, n * chan, (long long)chan);
        TRY_OR_CLEAN(hipGetLastError());
    }
    TRY_OR_CLEAN(hipMemcpy(y_h, d_x, xbytes, hipMemcpyDeviceToHost));
#undef TRY_OR_CLEAN
    cleanup();
    return SSFM_OK;
}

}  // namespace

extern "C" int ssfm_sosfiltfilt(int device, const double* sos, const double* zi, int n_sections, const void* x, void* y,
                                int64_t n, int batch, int is_complex) {
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
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count)
        return fail(SSFM_ERR_NO_DEVICE, "ssfm_sosfiltfilt: device %d not available", device);
    HIP_TRY(hipSetDevice(device));
    const int chan = is_complex ? 2 : 1;
    const double* xh = static_cast<const double*>(x);
    double* yh = static_cast<double*>(y);
    switch (n_sections) {
        case 1: return run_filter<1>(c, zi, xh, yh, n, batch, chan, edge);
        case 2: return run_filter<2>(c, zi, xh, yh, n, batch, chan, edge);
        case 3: return run_filter<3>(c, zi, xh, yh, n, batch, chan, edge);
        default: return run_filter<4>(c, zi, xh, yh, n, batch, chan, edge);
    }
}
