// ssfm_split.hpp -- fields of more than 2^22 samples per row ("split plans", round 6).
//
// The reference takes any length its host's memory allows (devices.py:1147,1178-1180: numpy.fft has no size cap); the two-kernel engine stops where a row of
// the four-step transform no longer fits a workgroup (N = N1 N2, N1 <= 512 column points, N2 <= 8192 row points: 2^22).  Beyond that the transform gets a THIRD
// factor, N = R M with M = 2^20 (the size the two-kernel engine runs best at) and R = 2 ... 16, by decimation in TIME:
//
//     x_a[m] = x[a + R m]   (a < R sub-sequences of M samples: real time-domain samples, so the nonlinear operator -- pointwise in time -- acts on them as they are)
//     Y_a    = FFT_M(x_a)                                                         the plan's own column + row passes, on R x rows sub-rows
//     X[q + M p] = sum_a W_R^(a p) (W_N^(a q) Y_a[q])        q < M, p < R         k_split_mid: pointwise in q, a radix-R butterfly across the sub-rows
//     X'[k] = X[k] exp(D~_k h) / N                                                (the same launch)
//     Z_a[q] = conj(W_N^(a q)) sum_p W_R^(-a p) X'[q + M p]                       (the same launch)
//     x'_a   = IFFT_M(Z_a) (unnormalised: 1 / N is in the operator)               the plan's row + column passes
//
// One split step = k_time<MID> (unchanged: inverse column pass, both half rotations, forward column pass -- on sub-rows), k_freq<FM_FWD_ONLY> (row pass forward),
// k_split_mid, k_freq<FM_INV_ONLY> (row pass inverse): FOUR passes over the field instead of two -- the minimum for a transform of three factors when the
// pass on either side of the operator is fused with it and the passes on either side of the nonlinear step with each other.  Algorithmic bytes stay 2 P E per
// sample and step; the design floor doubles.  Between the two halves of the row pass a sub-row keeps the plan's own order -- the 16-byte
// unit layout of the complex64 plans (FreqArgs::u16 == 2), the plain transposed order [k1][k2] (q = k1 + N1 k2) otherwise; k_split_mid is pointwise, so only its
// tables and its twiddle look-up know the order (split_k2_of_pos).  A fibre's operator (one modulus for all frequencies) travels as 4-byte phases in complex64
// (SM_PHASE, as k_freq<FM_PHASE>): A/B over three rounds (profiles/r06_split_ab.txt): 2^23 x 2 203-207 -> 194 us per step, 2^24 x 2 450-479 -> 440-449; the unit
// layout alone: nothing.
#pragma once
#include "ssfm_kernels.hpp"

namespace ssfm {

constexpr int kSplitLog2M = 20;            // sub-sequence length of a split plan (env SSFM_SPLIT_LOG2M = 20 | 21 | 22 overrides: diagnostics)
constexpr int kSplitMaxR = 16;

enum SplitMode { SM_TABLE = 0, SM_FLY = 1, SM_PHASE = 2 };       // SM_PHASE (complex64, a fibre's operator: one modulus for all frequencies): 4-byte phases, as k_freq<FM_PHASE>

template <typename T> struct SplitArgs {
    cx<T>* Y;                 // the sub-rows between the two row passes: rows_outer x R sub-rows of M, plain transposed order
    const cx<T>* G;           // SM_TABLE: exp(D~ h) / N at [p][k1][k2];  SM_FLY: D~ there
    const double2* twA;       // W_N^(a k1)      [a][k1], a < R
    const double2* twB;       // W_N^(a N1 k2)   [a][k2]
    const AdaptState<T>* st;  // SM_FLY: step size source when non-null
    T h;                      // SM_FLY with st == nullptr
    T inv_n;                  // 1 / N (SM_FLY)
    T amp;                    // SM_PHASE: exp(Re D~ h) / N, the modulus of every table entry
    int step;
    int N1, N2;
    int rows_outer;           // rows of N samples covered by this launch
    int Qf;                   // > 0: the sub-rows lie in the plan's 16-byte-unit order between the row-pass halves (position -> k2 by u16_col_of_pos); 0: plain order
};
// the row frequency k2 at position `pos` of a sub-row between the two halves of the row pass
__host__ __device__ __forceinline__ long long split_k2_of_pos(long long pos, int Qf) { return Qf > 0 ? u16_col_of_pos(pos, Qf) : pos; }

// One thread: V consecutive positions (16 bytes) of all R sub-rows of every outer row -- twiddles and operator formed once, applied to every row.  R = 16 in complex64
// needs 256 VGPRs + 42 AGPRs (one wave per SIMD); the same kernel with ONE position per thread (166 VGPRs, three waves per SIMD, 8-byte accesses) takes the same time
// (2^24 x 2: 441-450 against 446-449 us per step, profiles/r06_split_ab.txt): sixteen 8 MiB-apart streams per wavefront are bound by the memory system, not by occupancy.
template <typename T, int R> __host__ __device__ constexpr int split_positions() { return sizeof(T) == 4 ? 2 : 1; }
template <typename T, int R, int MODE>
__global__ __launch_bounds__(256) void k_split_mid(const SplitArgs<T> a) {
    constexpr int V = split_positions<T, R>();
    typedef T v4_t __attribute__((ext_vector_type(V * 2)));
    const long long M = (long long)a.N1 * a.N2;
    const long long unit = (long long)blockIdx.x * 256 + threadIdx.x;
    if (unit * V >= M) return;
    const long long p0 = unit * V;
    const int k1 = (int)(p0 / a.N2);                                 // (N2 is even: the V positions share k1)
    int k2v[V];
#pragma unroll
    for (int v = 0; v < V; ++v) k2v[v] = (int)split_k2_of_pos(p0 % a.N2 + v, a.Qf);
    T h = a.h;
    if (MODE == SM_FLY && a.st != nullptr) {
        const StepState<T> S = a.st->cur[a.step & 1];
        if (S.done || a.st->error != 0u) return;
        h = S.h;
    }
    // the operator of the R frequencies q + M p behind every position
    cx<T> g[R][V];
    if constexpr (MODE == SM_PHASE) {
        typedef unsigned uv_t __attribute__((ext_vector_type(V)));
        const unsigned* Gu = reinterpret_cast<const unsigned*>(a.G);
#pragma unroll
        for (int p = 0; p < R; ++p) {
            const uv_t q = *reinterpret_cast<const uv_t*>(Gu + (long long)p * M + p0);
#pragma unroll
            for (int v = 0; v < V; ++v) g[p][v] = phase32_factor(q[v], a.amp);
        }
    } else {
#pragma unroll
    for (int p = 0; p < R; ++p) {
        const v4_t q = *reinterpret_cast<const v4_t*>(a.G + (long long)p * M + p0);
#pragma unroll
        for (int v = 0; v < V; ++v) g[p][v] = mk<T>(q[2 * v], q[2 * v + 1]);
    }
    }
    if constexpr (MODE == SM_FLY) {
        // exp(D~ h) / N: the reference's products in T (devices.py:1179), as k_freq<FM_FLY> forms them
#pragma unroll
        for (int v = 0; v < V; ++v) {
            cx<T> m[R];
            T ph[R];
            bool flat = true;
#pragma unroll
            for (int p = 0; p < R; ++p) { m[p] = g[p][v]; ph[p] = m[p].y * h; flat = flat && (m[p].x == m[0].x); }
            const T e0 = exp_acc<T>(m[0].x * h);
            fly_factors<R>(m, ph, flat, e0, h, a.inv_n);
#pragma unroll
            for (int p = 0; p < R; ++p) g[p][v] = m[p];
        }
    }
    // W_N^(a q), q = k1 + N1 k2: two factors from tables in double, ONE rounding
    cx<T> w[R][V];
#pragma unroll
    for (int s = 1; s < R; ++s) {
        const double2 A = a.twA[s * a.N1 + k1];
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const double2 B = a.twB[(long long)s * a.N2 + k2v[v]];
            w[s][v] = mk<T>((T)(A.x * B.x - A.y * B.y), (T)(A.x * B.y + A.y * B.x));
        }
    }
    for (int row = 0; row < a.rows_outer; ++row) {
        cx<T>* base = a.Y + (long long)row * R * M + p0;
        cx<T> y[R][V];
#pragma unroll
        for (int s = 0; s < R; ++s) {
            const v4_t q = *reinterpret_cast<const v4_t*>(base + (long long)s * M);
#pragma unroll
            for (int v = 0; v < V; ++v) y[s][v] = mk<T>(q[2 * v], q[2 * v + 1]);
        }
#pragma unroll
        for (int v = 0; v < V; ++v) {
            cx<T> t[R];
            t[0] = y[0][v];
#pragma unroll
            for (int s = 1; s < R; ++s) t[s] = cmul(y[s][v], w[s][v]);
            Dft<R>::template run<-1>(t);
#pragma unroll
            for (int p = 0; p < R; ++p) t[p] = cmul(t[p], g[p][v]);
            Dft<R>::template run<+1>(t);
            y[0][v] = t[0];
#pragma unroll
            for (int s = 1; s < R; ++s) y[s][v] = cmul(t[s], mk<T>(w[s][v].x, -w[s][v].y));
        }
#pragma unroll
        for (int s = 0; s < R; ++s) {
            v4_t q;
#pragma unroll
            for (int v = 0; v < V; ++v) { q[2 * v] = y[s][v].x; q[2 * v + 1] = y[s][v].y; }
            *reinterpret_cast<v4_t*>(base + (long long)s * M) = q;
        }
    }
}

// out[p][k1][k2] = f(src[(k1 + N1 k2) + M p]); MODE 0: copy (D~ for SM_FLY), 1: * inv_n (a transfer function), 2: exp(src h) * inv_n (k_make_freq_table's arithmetic)
template <typename T, int MODE>
__global__ void k_make_split_table(const cx<T>* __restrict__ src, cx<T>* __restrict__ out, int N1, int N2, int R, T h, T inv_n, int Qf) {
    const long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long M = (long long)N1 * N2;
    if (o >= M * R) return;
    const long long p = o / M, r = o % M;
    const long long k1 = r / N2, k2 = split_k2_of_pos(r % N2, Qf);
    cx<T> d = src[k1 + (long long)N1 * k2 + M * p];
    if (MODE == 1) { d.x *= inv_n; d.y *= inv_n; }
    if (MODE == 2) {
        const T xr = d.x * h, xi = d.y * h;
        const T e = (T)exp((double)xr);
        double s, c;
        sincos((double)xi, &s, &c);
        d.x = (e * (T)c) * inv_n;
        d.y = (e * (T)s) * inv_n;
    }
    out[o] = d;
}
// phase table of exp(D~ h) in the split order (k_make_phase_table's arithmetic: the reference's float32 product Im(D~) h taken to double, the turn fraction in 32 bits)
template <typename T>
__global__ void k_make_split_phase_table(const cx<T>* __restrict__ src, unsigned* __restrict__ out, int N1, int N2, int R, T h, int Qf) {
    const long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long M = (long long)N1 * N2;
    if (o >= M * R) return;
    const long long p = o / M, r = o % M;
    const long long k1 = r / N2, k2 = split_k2_of_pos(r % N2, Qf);
    const T xi = src[k1 + (long long)N1 * k2 + M * p].y * h;
    double turns = (double)xi * 0.15915494309189533577;
    turns -= floor(turns);
    out[o] = (unsigned)(unsigned long long)llrint(turns * 4294967296.0);
}
// DM transfer function in the split order (k_make_dm_table's arithmetic on the grid of N = R M frequencies); `nat` != nullptr: H in natural order as well
template <typename T>
__global__ void k_make_split_dm_table(cx<T>* __restrict__ out, cx<T>* __restrict__ nat, int N1, int N2, int R, double val, double D, T inv_n, int Qf) {
    const long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long M = (long long)N1 * N2, N = M * R;
    if (o >= N) return;
    const long long p = o / M, r = o % M;
    const long long k = (r / N2) + (long long)N1 * split_k2_of_pos(r % N2, Qf) + M * p;
    const long long ks = k < (N + 1) / 2 ? k : k - N;
    const double w = ((double)ks * val) * 2.0 * 3.141592653589793;
    const double ph = ((w * w) * D) / 2.0;
    double s, c;
    sincos(ph, &s, &c);
    out[o] = mk<T>((T)c * inv_n, (T)s * inv_n);
    if (nat != nullptr) nat[k] = mk<T>((T)c, (T)s);
}
// twA[a][k1] = W_N^(a k1), twB[a][k2] = W_N^(a N1 k2): exact integer angle reduction, sincospi in double
__global__ inline void k_make_split_twiddles(double2* twA, double2* twB, int N1, int N2, int R) {
    const long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long N = (long long)N1 * N2 * R;
    const long long nA = (long long)R * N1, nB = (long long)R * N2;
    if (o >= nA + nB) return;
    long long m;
    if (o < nA) { const long long s = o / N1, k1 = o % N1; m = (s * k1) % N; }
    else { const long long s = (o - nA) / N2, k2 = (o - nA) % N2; m = ((s * N1) % N) * k2 % N; }
    double sn, cs;
    sincospi(-2.0 * (double)m / (double)N, &sn, &cs);
    (o < nA ? twA[o] : twB[o - nA]) = make_double2(cs, sn);
}
// natural time order <-> sub-sequences: sub[(row R + a) M + m] = nat[row N + a + R m].  One thread per (row, m): R consecutive samples on the natural side.
template <typename T, int R, bool TO_SUB>
__global__ __launch_bounds__(256) void k_split_shuffle(cx<T>* __restrict__ sub, cx<T>* __restrict__ nat, long long M, int rows) {
    const long long o = (long long)blockIdx.x * 256 + threadIdx.x;
    if (o >= M * rows) return;
    const long long row = o / M, m = o % M;
    cx<T>* n0 = nat + (row * M + m) * R;
    cx<T>* s0 = sub + row * R * M + m;
#pragma unroll
    for (int a = 0; a < R; ++a) {
        if (TO_SUB) s0[(long long)a * M] = n0[a];
        else n0[a] = s0[(long long)a * M];
    }
}

}  // namespace ssfm
