// wgfft.hpp -- workgroup-level power-of-two FFT building block for gfx950 (wave64).
//
// A "line" of L = 2^m complex points (16 <= L <= 4096) is transformed by Q = L/16
// threads; every thread owns 16 points in registers.  Register slot t of thread j holds
// line element j + t*Q on entry AND on exit ("pattern P"), so a forward and an inverse
// transform can be chained without any data movement in between, and the global-memory
// access pattern on both sides of a kernel is the same.
//
// The transform is a Stockham autosort: 1..3 radix-{16,8,4} stages.  The first stage works
// straight out of registers, every later stage after one exchange through LDS, and the
// last stage leaves its results in registers.  Stage twiddles W_L^q come from one table of
// L entries per line length (generated in double, rounded once).
#pragma once
#include <hip/hip_runtime.h>

namespace ssfm {

template <typename T> struct cx_of;
template <> struct cx_of<float>  { using type = float2; };
template <> struct cx_of<double> { using type = double2; };
template <typename T> using cx = typename cx_of<T>::type;

template <typename T> __device__ __forceinline__ cx<T> mk(T x, T y) { cx<T> r; r.x = x; r.y = y; return r; }
template <typename C> __device__ __forceinline__ C cadd(C a, C b) { a.x += b.x; a.y += b.y; return a; }
template <typename C> __device__ __forceinline__ C csub(C a, C b) { a.x -= b.x; a.y -= b.y; return a; }
// a * b
template <typename C> __device__ __forceinline__ C cmul(C a, C b) {
    C r; r.x = a.x * b.x - a.y * b.y; r.y = a.x * b.y + a.y * b.x; return r;
}
// a * conj(b)
template <typename C> __device__ __forceinline__ C cmulc(C a, C b) {
    C r; r.x = a.x * b.x + a.y * b.y; r.y = a.y * b.x - a.x * b.y; return r;
}
// Tables hold forward twiddles exp(-i*theta); DIR < 0 = forward, DIR > 0 = inverse (conjugate).
template <int DIR, typename C> __device__ __forceinline__ C cmuld(C a, C w) {
    return DIR < 0 ? cmul(a, w) : cmulc(a, w);
}
// multiply by -i (forward) or +i (inverse)
template <int DIR, typename C> __device__ __forceinline__ C rot90(C a) {
    C r;
    if (DIR < 0) { r.x = a.y;  r.y = -a.x; }
    else         { r.x = -a.y; r.y = a.x;  }
    return r;
}

// ---------------------------------------------------------------- small DFTs in registers
template <int DIR, typename C> __device__ __forceinline__ void dft2(C& a, C& b) {
    C t = a; a = cadd(t, b); b = csub(t, b);
}
template <int DIR, typename C> __device__ __forceinline__ void dft4(C& x0, C& x1, C& x2, C& x3) {
    C a0 = cadd(x0, x2), a1 = csub(x0, x2), a2 = cadd(x1, x3), a3 = rot90<DIR>(csub(x1, x3));
    x0 = cadd(a0, a2); x1 = cadd(a1, a3); x2 = csub(a0, a2); x3 = csub(a1, a3);
}

template <int R> struct Dft;
template <> struct Dft<2> {
    template <int DIR, typename C> static __device__ __forceinline__ void run(C* x) { dft2<DIR>(x[0], x[1]); }
};
template <> struct Dft<4> {
    template <int DIR, typename C> static __device__ __forceinline__ void run(C* x) { dft4<DIR>(x[0], x[1], x[2], x[3]); }
};
template <> struct Dft<8> {
    template <int DIR, typename C> static __device__ __forceinline__ void run(C* x) {
        using T = decltype(x[0].x);
        const T s = (T)0.70710678118654752440;
        // even / odd DFT-4
        dft4<DIR>(x[0], x[2], x[4], x[6]);
        dft4<DIR>(x[1], x[3], x[5], x[7]);
        // odd *= W8^k
        C o1, o2, o3;
        if (DIR < 0) {   // W8 = (1 - i)/sqrt2 ; W8^3 = (-1 - i)/sqrt2
            o1.x = (x[3].x + x[3].y) * s; o1.y = (x[3].y - x[3].x) * s;
            o3.x = (x[7].y - x[7].x) * s; o3.y = -(x[7].x + x[7].y) * s;
        } else {         // conj
            o1.x = (x[3].x - x[3].y) * s; o1.y = (x[3].y + x[3].x) * s;
            o3.x = -(x[7].x + x[7].y) * s; o3.y = (x[7].x - x[7].y) * s;
        }
        o2 = rot90<DIR>(x[5]);
        C e0 = x[0], e1 = x[2], e2 = x[4], e3 = x[6], o0 = x[1];
        x[0] = cadd(e0, o0); x[4] = csub(e0, o0);
        x[1] = cadd(e1, o1); x[5] = csub(e1, o1);
        x[2] = cadd(e2, o2); x[6] = csub(e2, o2);
        x[3] = cadd(e3, o3); x[7] = csub(e3, o3);
    }
};
template <> struct Dft<16> {
    template <int DIR, typename C> static __device__ __forceinline__ void run(C* x) {
        using T = decltype(x[0].x);
        const T c1 = (T)0.92387953251128675613, s1 = (T)0.38268343236508977173;   // cos, sin(pi/8)
        const T c2 = (T)0.70710678118654752440;
        // four DFT-4 over stride-4 subsequences: x[b + 4q] <- G_b[q]
#pragma unroll
        for (int b = 0; b < 4; ++b) dft4<DIR>(x[b], x[b + 4], x[b + 8], x[b + 12]);
        // twiddle G_b[q] *= W16^(b*q); forward W16^k = (cos(k pi/8), -sin(k pi/8))
        const T wr[10] = { (T)1, c1, c2, s1, (T)0, -s1, -c2, -c1, (T)-1, -c1 };   // cos(k pi/8), k=0..9
        const T wi[10] = { (T)0, s1, c2, c1, (T)1, c1, c2, s1, (T)0, -s1 };       // sin(k pi/8)
#pragma unroll
        for (int b = 1; b < 4; ++b) {
#pragma unroll
            for (int q = 1; q < 4; ++q) {
                const int k = b * q;                       // 1..9
                C w; w.x = wr[k]; w.y = -wi[k];
                x[b + 4 * q] = cmuld<DIR>(x[b + 4 * q], w);
            }
        }
        // DFT-4 across b for every q; result p of group q is output q + 4p
        C y[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            C a = x[4 * q], b = x[4 * q + 1], c = x[4 * q + 2], d = x[4 * q + 3];
            dft4<DIR>(a, b, c, d);
            y[q] = a; y[q + 4] = b; y[q + 8] = c; y[q + 12] = d;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = y[i];
    }
};

// ---------------------------------------------------------------- stage plan
__host__ __device__ constexpr int fft_nstages(int L) { return L <= 16 ? 1 : (L <= 256 ? 2 : 3); }
// radix of stage s (0-based) for line length L
__host__ __device__ constexpr int fft_radix(int L, int s) {
    return L == 16   ? 16
         : L == 32   ? (s == 0 ? 8 : 4)
         : L == 64   ? 8
         : L == 128  ? (s == 0 ? 16 : 8)
         : L == 256  ? 16
         : L == 512  ? 8
         : L == 1024 ? (s == 0 ? 16 : 8)
         : L == 2048 ? (s <= 1 ? 16 : 8)
         :             16;   // 4096
}
__host__ __device__ constexpr int fft_ls(int L, int s) {   // product of radices before stage s
    return s == 0 ? 1 : fft_ls(L, s - 1) * fft_radix(L, s - 1);
}

// One Stockham stage.  IDX maps a line element index to an LDS element index.
template <typename T, int L, int DIR, int S, typename IDX>
__device__ __forceinline__ void fft_stage(cx<T> (&v)[16], cx<T>* lds, const int j, const IDX& idx,
                                          const cx<T>* __restrict__ tw) {
    constexpr int M  = fft_nstages(L);
    constexpr int R  = fft_radix(L, S);
    constexpr int NB = 16 / R;          // butterflies per thread
    constexpr int Q  = L / 16;          // threads per line
    constexpr int LS = fft_ls(L, S);
    constexpr int STEP = L / (LS * R);  // table stride of this stage's twiddles

    if (S > 0) {
        // all reads of the previous exchange happen here; writers finished before the barrier
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int jb = j + i * Q;
#pragma unroll
            for (int u = 0; u < R; ++u) v[i + u * NB] = lds[idx(jb + u * (L / R))];
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int k = (j + i * Q) & (LS - 1);
#pragma unroll
            for (int u = 1; u < R; ++u) v[i + u * NB] = cmuld<DIR>(v[i + u * NB], tw[k * u * STEP]);
        }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        cx<T> tmp[R];
#pragma unroll
        for (int u = 0; u < R; ++u) tmp[u] = v[i + u * NB];
        Dft<R>::template run<DIR>(tmp);
#pragma unroll
        for (int u = 0; u < R; ++u) v[i + u * NB] = tmp[u];
    }
    if (S < M - 1) {
        if (S > 0) __syncthreads();     // every thread has read its inputs of this stage
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int jb = j + i * Q;
            const int k = jb & (LS - 1);
            const int base = (jb - k) * R + k;
#pragma unroll
            for (int u = 0; u < R; ++u) lds[idx(base + u * LS)] = v[i + u * NB];
        }
        __syncthreads();
    }
}

// Full line transform.  The caller guarantees that nobody is still reading `lds` from an
// earlier exchange when this is entered (i.e. there was a barrier since).
template <typename T, int L, int DIR, typename IDX>
__device__ __forceinline__ void fft_line(cx<T> (&v)[16], cx<T>* lds, const int j, const IDX& idx,
                                         const cx<T>* __restrict__ tw) {
    constexpr int M = fft_nstages(L);
    fft_stage<T, L, DIR, 0, IDX>(v, lds, j, idx, tw);
    if (M > 1) fft_stage<T, L, DIR, (M > 1 ? 1 : 0), IDX>(v, lds, j, idx, tw);
    if (M > 2) fft_stage<T, L, DIR, (M > 2 ? 2 : 0), IDX>(v, lds, j, idx, tw);
}

}  // namespace ssfm
