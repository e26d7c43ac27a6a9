// wgfft.hpp -- workgroup-level power-of-two FFT building block for gfx950 (wave64).
//
// A "line" of L = 2^m complex points (16 <= L <= 8192) is transformed by Q = L/E
// threads; every thread owns E (16 or 8) points in registers.  Register slot t of thread j holds
// line element j + t*Q on entry AND on exit ("pattern P"), so a forward and an inverse
// transform can be chained without any data movement in between, and the global-memory
// access pattern on both sides of a kernel is the same.
//
// The transform is a Stockham autosort: 1..4 radix-{16,8,4,2} stages.  The first stage works
// straight out of registers, every later stage after one exchange through LDS, and the
// last stage leaves its results in registers.  Stage twiddles W_L^q come from one table of
// L entries per line length (generated in double, rounded once) and are kept in registers.
//
// complex64 arithmetic is written directly in CDNA packed-FP32 instructions: one complex
// number is one 64-bit VGPR pair (re, im); v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 with
// op_sel (half swizzle) and neg_lo/neg_hi modifiers give a complex add or a +-i rotation folded
// into an add in ONE instruction and a complex multiply in TWO, with no register shuffling.
// (hipcc's own SLP packing of scalar code needs ~25 % extra v_mov and folds no half negation.)
#pragma once
#include <hip/hip_runtime.h>

// In-kernel cycle stamps for tools/stamp_harness.hip (diagnostic builds only; the product never defines it)
#ifndef SSFM_STAMPS
#define SSFM_STAMPS 0
#endif
#if SSFM_STAMPS
extern __device__ unsigned long long* g_stamp_buf;
#define SSFM_STAMP(i)                                                                         \
    do {                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        if (threadIdx.x == 0) {                                                               \
            unsigned long long t_;                                                            \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");         \
            g_stamp_buf[blockIdx.x * 16 + (i)] = t_;                                          \
        }                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                    \
    } while (0)
#else
#define SSFM_STAMP(i) do { } while (0)
#endif


namespace ssfm {

typedef float  cf32 __attribute__((ext_vector_type(2)));
typedef double cf64 __attribute__((ext_vector_type(2)));
template <typename T> struct cx_of;
template <> struct cx_of<float>  { using type = cf32; };
template <> struct cx_of<double> { using type = cf64; };
template <typename T> using cx = typename cx_of<T>::type;
template <typename C> struct real_of;
template <> struct real_of<cf32> { using type = float; };
template <> struct real_of<cf64> { using type = double; };

template <typename T> __device__ __forceinline__ cx<T> mk(T x, T y) { cx<T> r; r.x = x; r.y = y; return r; }

// ---- generic (double) forms
__device__ __forceinline__ cf64 cadd(cf64 a, cf64 b) { return a + b; }
__device__ __forceinline__ cf64 csub(cf64 a, cf64 b) { return a - b; }
__device__ __forceinline__ cf64 cmul(cf64 a, cf64 b) { cf64 r; r.x = a.x * b.x - a.y * b.y; r.y = a.x * b.y + a.y * b.x; return r; }
__device__ __forceinline__ cf64 cmulc(cf64 a, cf64 b) { cf64 r; r.x = a.x * b.x + a.y * b.y; r.y = a.y * b.x - a.x * b.y; return r; }
// a + (-i) d   and   a + (+i) d
__device__ __forceinline__ cf64 add_mi(cf64 a, cf64 d) { cf64 r; r.x = a.x + d.y; r.y = a.y - d.x; return r; }
__device__ __forceinline__ cf64 add_pi(cf64 a, cf64 d) { cf64 r; r.x = a.x - d.y; r.y = a.y + d.x; return r; }
__device__ __forceinline__ cf64 mul_mi(cf64 a) { cf64 r; r.x = a.y; r.y = -a.x; return r; }
__device__ __forceinline__ cf64 mul_pi(cf64 a) { cf64 r; r.x = -a.y; r.y = a.x; return r; }
__device__ __forceinline__ cf64 scale2(cf64 a, cf64 s) { return a * s; }

// (Plain component-wise float arithmetic instead of the packed forms, built with -fno-slp-vectorize, is SLOWER in every configuration -- C2 18.4 vs 17.2 us per step,
// profiles/r03_ubench_pk_issue.txt: a v_pk_fma_f32 costs a lone wave 9.0 cycles against 12.2 for the two v_fma_f32 it replaces, and the kernels' compute
// phases run with one wave per SIMD most of the time.  That arm was removed in round 5.)
// ---- float forms: packed FP32 with modifiers (VOP3P).  op_sel[i] / op_sel_hi[i] pick the half of
// source i that feeds the low / high result lane; neg_lo / neg_hi negate a source per lane.
__device__ __forceinline__ cf32 cadd(cf32 a, cf32 b) { return a + b; }      // v_pk_add_f32
__device__ __forceinline__ cf32 csub(cf32 a, cf32 b) { return a - b; }      // v_pk_add_f32 neg
__device__ __forceinline__ cf32 scale2(cf32 a, cf32 s) { return a * s; }    // v_pk_mul_f32
__device__ __forceinline__ cf32 cmul(cf32 a, cf32 w) {
    cf32 t, r;      // t = (ay wy, ay wx);  r = (ax wx - t.lo, ax wy + t.hi)
    asm("v_pk_mul_f32 %1, %2, %3 op_sel:[1,1] op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %0, %2, %3, %1 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[0,0,1]"
        : "=v"(r), "=&v"(t) : "v"(a), "v"(w));
    return r;
}
__device__ __forceinline__ cf32 cmulc(cf32 a, cf32 w) {
    cf32 t, r;      // t = (ay wy, ay wx);  r = (ax wx + t.lo, -ax wy + t.hi)
    asm("v_pk_mul_f32 %1, %2, %3 op_sel:[1,1] op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %0, %2, %3, %1 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]"
        : "=v"(r), "=&v"(t) : "v"(a), "v"(w));
    return r;
}
__device__ __forceinline__ cf32 add_mi(cf32 a, cf32 d) {        // (a.x + d.y, a.y - d.x)
    cf32 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(d));
    return r;
}
__device__ __forceinline__ cf32 add_pi(cf32 a, cf32 d) {        // (a.x - d.y, a.y + d.x)
    cf32 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(d));
    return r;
}
__device__ __forceinline__ cf32 mul_mi(cf32 a) {                // (a.y, -a.x)
    cf32 r;
    asm("v_pk_mul_f32 %0, 1.0, %1 op_sel:[0,1] op_sel_hi:[0,0] neg_hi:[0,1]" : "=v"(r) : "v"(a));
    return r;
}
__device__ __forceinline__ cf32 mul_pi(cf32 a) {                // (-a.y, a.x)
    cf32 r;
    asm("v_pk_mul_f32 %0, 1.0, %1 op_sel:[0,1] op_sel_hi:[0,0] neg_lo:[0,1]" : "=v"(r) : "v"(a));
    return r;
}


// Tables hold forward twiddles exp(-i*theta); DIR < 0 = forward, DIR > 0 = inverse (conjugate).
template <int DIR, typename C> __device__ __forceinline__ C cmuld(C a, C w) { return DIR < 0 ? cmul(a, w) : cmulc(a, w); }
// a + (DIR: -i forward, +i inverse) * d,  and the opposite sign
template <int DIR, typename C> __device__ __forceinline__ C add_rot(C a, C d) { return DIR < 0 ? add_mi(a, d) : add_pi(a, d); }
template <int DIR, typename C> __device__ __forceinline__ C sub_rot(C a, C d) { return DIR < 0 ? add_pi(a, d) : add_mi(a, d); }
template <int DIR, typename C> __device__ __forceinline__ C rot90(C a) { return DIR < 0 ? mul_mi(a) : mul_pi(a); }

// ---------------------------------------------------------------- small DFTs in registers
template <int DIR, typename C> __device__ __forceinline__ void dft2(C& a, C& b) {
    C t = a; a = cadd(t, b); b = csub(t, b);
}
template <int DIR, typename C> __device__ __forceinline__ void dft4(C& x0, C& x1, C& x2, C& x3) {
    const C a0 = cadd(x0, x2), a1 = csub(x0, x2), a2 = cadd(x1, x3), d = csub(x1, x3);
    x0 = cadd(a0, a2); x2 = csub(a0, a2);
    x1 = add_rot<DIR>(a1, d); x3 = sub_rot<DIR>(a1, d);       // a1 -+ i d
}

template <int R> struct Dft;
template <> struct Dft<2> {
    template <int DIR, typename C> static __device__ __forceinline__ void run(C* x) { dft2<DIR>(x[0], x[1]); }
};
template <> struct Dft<4> {
    template <int DIR, typename C> static __device__ __forceinline__ void run(C* x) { dft4<DIR>(x[0], x[1], x[2], x[3]); }
};
// Radix constants.  In float32 the correctly rounded cos/sin(pi/8) and sqrt(1/2) have
// |w|^2 - 1 = -5.7e-8 and -3.4e-8: a SYSTEMATIC amplitude loss that every radix-8/16 butterfly
// applies and that adds up over 10 butterfly layers x 1000 steps to 2e-4 of the signal energy
// (measured).  No single float is better, so the bias is cancelled ACROSS the elements of one
// butterfly at no cost in instructions:
//   sqrt(1/2): R2D = round-down float (-1.71e-8 relative), R2U = next float up (+6.72e-8); half of
//              the W8 rotations use (R2D, R2U) for their two components, the others (R2D, R2D);
//   pi/8:      two (cos, sin) float pairs whose norms err in opposite directions,
//              A: |w|^2 - 1 = +7.4e-9, B: -1.14e-8 (angles off by < 7.3e-8 rad, the size of an
//              ordinary float32 twiddle rounding error).
// Net bias of a radix-16 butterfly: +3e-10 (was -2.3e-8).
template <typename T> struct RadixConst;
template <> struct RadixConst<float> {
    static constexpr float R2D = 0.7071067690849304f, R2U = 0.7071068286895752f;
    static constexpr float cA = 0.9238795638084412f, sA = 0.3826833665370941f;
    static constexpr float cB = 0.9238795042037964f, sB = 0.38268348574638367f;
};
template <> struct RadixConst<double> {
    static constexpr double R2D = 0.70710678118654752440, R2U = R2D;
    static constexpr double cA = 0.92387953251128675613, sA = 0.38268343236508977173;
    static constexpr double cB = cA, sB = sA;
};
// x * W8^1 (forward (1-i)/sqrt2, inverse (1+i)/sqrt2) and x * W8^3 (forward (-1-i)/sqrt2, inverse
// (-1+i)/sqrt2):  x (1 -+ i) = x -+ i x is one rotated add, then one scale by a constant pair.
// VAR = 1: the imaginary component is scaled with the rounded-up constant.
template <int DIR, int VAR, typename C> __device__ __forceinline__ C mul_w8_1(C a) {
    using T = typename real_of<C>::type;
    return scale2(add_rot<DIR>(a, a), mk<T>(RadixConst<T>::R2D, VAR ? RadixConst<T>::R2U : RadixConst<T>::R2D));
}
template <int DIR, int VAR, typename C> __device__ __forceinline__ C mul_w8_3(C a) {
    using T = typename real_of<C>::type;
    // forward: x(-1-i) = -(x + i x);  inverse: x(-1+i) = -(x - i x)
    return scale2(sub_rot<DIR>(a, a), mk<T>(-RadixConst<T>::R2D, VAR ? -RadixConst<T>::R2U : -RadixConst<T>::R2D));
}

template <> struct Dft<8> {
    template <int DIR, typename C> static __device__ __forceinline__ void run(C* x) {
        // even / odd DFT-4, odd *= W8^k, combine
        dft4<DIR>(x[0], x[2], x[4], x[6]);
        dft4<DIR>(x[1], x[3], x[5], x[7]);
        const C o0 = x[1], o1 = mul_w8_1<DIR, 1>(x[3]), o3 = mul_w8_3<DIR, 0>(x[7]);
        const C e0 = x[0], e1 = x[2], e2 = x[4], e3 = x[6], o2 = x[5];
        x[0] = cadd(e0, o0); x[4] = csub(e0, o0);
        x[1] = cadd(e1, o1); x[5] = csub(e1, o1);
        x[2] = add_rot<DIR>(e2, o2); x[6] = sub_rot<DIR>(e2, o2);      // o2 * W8^2 = -+ i o2
        x[3] = cadd(e3, o3); x[7] = csub(e3, o3);
    }
};
template <> struct Dft<16> {
    template <int DIR, typename C> static __device__ __forceinline__ void run(C* x) {
        using T = typename real_of<C>::type;
        using K = RadixConst<T>;
        // four DFT-4 over stride-4 subsequences: x[b + 4q] <- G_b[q]
#pragma unroll
        for (int b = 0; b < 4; ++b) dft4<DIR>(x[b], x[b + 4], x[b + 8], x[b + 12]);
        // G_b[q] *= W16^(b*q); forward W16^k = (cos(k pi/8), -sin(k pi/8))
        x[1 + 4]  = cmuld<DIR>(x[1 + 4], mk<T>(K::cA, -K::sA));      // b=1 q=1: W^1
        x[1 + 8]  = mul_w8_1<DIR, 1>(x[1 + 8]);                       // b=1 q=2: W^2
        x[1 + 12] = cmuld<DIR>(x[1 + 12], mk<T>(K::sB, -K::cB));     // b=1 q=3: W^3
        x[2 + 4]  = mul_w8_1<DIR, 0>(x[2 + 4]);                       // b=2 q=1: W^2
        x[2 + 8]  = rot90<DIR>(x[2 + 8]);                             // b=2 q=2: W^4 = -i
        x[2 + 12] = mul_w8_3<DIR, 1>(x[2 + 12]);                      // b=2 q=3: W^6
        x[3 + 4]  = cmuld<DIR>(x[3 + 4], mk<T>(K::sB, -K::cB));      // b=3 q=1: W^3
        x[3 + 8]  = mul_w8_3<DIR, 0>(x[3 + 8]);                       // b=3 q=2: W^6
        x[3 + 12] = cmuld<DIR>(x[3 + 12], mk<T>(-K::cB, K::sB));     // b=3 q=3: W^9 = -W^1
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
// E = points per thread (16 or 8).  E = 16: radix-16 stages, fewest LDS exchanges, ~200 VGPRs;
// E = 8: radix-8 stages, one more exchange for most lengths but half the registers and twice the
// waves for the same line -- what a small problem needs to keep two computing waves per SIMD.
__host__ __device__ constexpr int ilog2(int v) { return v <= 1 ? 0 : 1 + ilog2(v >> 1); }
__host__ __device__ constexpr int fft_nstages(int L, int E) {
    return E == 16 ? (L <= 16 ? 1 : (L <= 256 ? 2 : (L <= 4096 ? 3 : 4))) : (ilog2(L) + 2) / 3;
}
// radix of stage s (0-based) for line length L
__host__ __device__ constexpr int fft_radix(int L, int s, int E) {
    return E == 16 ? (L == 16   ? 16
                    : L == 32   ? (s == 0 ? 8 : 4)
                    : L == 64   ? 8
                    : L == 128  ? (s == 0 ? 16 : 8)
                    : L == 256  ? 16
                    : L == 512  ? 8
                    : L == 1024 ? (s == 0 ? 16 : 8)
                    : L == 2048 ? (s <= 1 ? 16 : 8)
                    : L == 4096 ? 16
                    :             (s <= 2 ? 16 : 2))   // 8192 = 16 * 16 * 16 * 2
                   // E == 8: radix 8 while three bits remain, then the remainder
                   : (s < ilog2(L) / 3 ? 8 : (1 << (ilog2(L) % 3)));
}
__host__ __device__ constexpr int fft_ls(int L, int s, int E) {   // product of radices before stage s
    return s == 0 ? 1 : fft_ls(L, s - 1, E) * fft_radix(L, s - 1, E);
}

// Stage twiddles of one thread, loaded once per kernel and kept in registers: they depend only on
// the thread's position in the line, and the forward and the inverse transform share them.
//
// complex128 cannot afford that: 3 x 15 twiddles of 4 registers each are 180 of the 256 a thread can have, and
// the 16-point kernels spilled to AGPRs and scratch.  There a thread keeps only WHERE its
// twiddles are -- the workgroup's LDS copy of a small stage, its own column of the global table for a large
// one -- and loads each factor when a stage needs it (twice per kernel: forward and inverse transform).
// (Only with 16 points per thread, i.e. the 2^21 / 2^22 plans: with 8 points per thread the registers suffice and
// keeping the twiddles is 2 % faster -- C1 46.6 vs 47.6 us per step; for complex64 and for 8-point rows of >= 2048 points it was measured and lost.)
template <typename T, int E, int L = 0> __host__ __device__ constexpr bool tw_lazy() {
    return sizeof(T) == 8 && E == 16;
}

template <typename T, int L, int E, bool LAZY = tw_lazy<T, E, L>()> struct LineTw {
    static constexpr int M = fft_nstages(L, E);
    cx<T> w[M > 1 ? M - 1 : 1][E - 1];
};
template <typename T, int L, int E> struct LineTw<T, L, E, true> {
    static constexpr int M = fft_nstages(L, E);
    const cx<T>* g[M > 1 ? M - 1 : 1];      // stage read from the global table: this thread's first entry
    const cx<T>* l[M > 1 ? M - 1 : 1];      // stage staged through LDS: this thread's first entry there
};
// Table layout: the twiddles are stored in the order the threads load them: for stage S >= 1,
// entry (slot, ku) = W_L^(k u STEP), slot enumerating (butterfly i, factor u >= 1) and ku the thread
// class: a stage's twiddles depend on the thread only through k = j mod LS, so there are KU = min(LS, Q)
// classes.  The last stage has KU = Q (every thread its own set): lanes with consecutive j read
// consecutive entries, one coalesced access per wave instruction (a gather from a natural W_L^q table
// cost ~100 cycles of issue per instruction).  Earlier stages have few classes (16 for 4096 = 16^3):
// their (R-1)*KU values go through LDS once per workgroup instead of 15 loads per thread -- the
// per-thread copies were a quarter of a workgroup's whole load traffic.
__host__ __device__ constexpr int fft_tw_slots_of(int L, int S, int E) { return (E / fft_radix(L, S, E)) * (fft_radix(L, S, E) - 1); }
__host__ __device__ constexpr int fft_tw_ku(int L, int S, int E) { return fft_ls(L, S, E) < L / E ? fft_ls(L, S, E) : L / E; }
// a stage whose table is small is staged through LDS (shared by the workgroup's lines too)
__host__ __device__ constexpr bool fft_tw_via_lds(int L, int S, int E) { return fft_tw_slots_of(L, S, E) * fft_tw_ku(L, S, E) <= 512; }
// A complex64 stage that every thread reads straight from the global table (the last stage of a 4096-point row: 15 factors
// per thread) stores the factors of slots 2p and 2p+1 side by side: 8 loads of 16 bytes per lane instead of 15 of 8 -- a vector-memory
// instruction costs a wave the same issue time whether a lane moves 8 or 16 bytes (tools/ubench_load_issue.hip), and these loads stand
// at the head of k_freq behind the field's.  ts = sizeof of the real type.
__host__ __device__ constexpr bool fft_tw_paired(int L, int S, int E, int ts) {
    return ts == 4 && !fft_tw_via_lds(L, S, E);
}
// table entries of stage S (always an even number, so that every stage starts on a 16-byte boundary)
__host__ __device__ constexpr int fft_tw_stage_entries(int L, int S, int E, int ts) {
    return fft_tw_paired(L, S, E, ts) ? ((fft_tw_slots_of(L, S, E) + 1) / 2) * 2 * fft_tw_ku(L, S, E)
                                      : ((fft_tw_slots_of(L, S, E) * fft_tw_ku(L, S, E) + 1) / 2) * 2;
}
__host__ __device__ constexpr int fft_tw_offset(int L, int S, int E, int ts) {       // first entry of stage S (S >= 1)
    return S <= 1 ? 0 : fft_tw_offset(L, S - 1, E, ts) + fft_tw_stage_entries(L, S - 1, E, ts);
}
__host__ __device__ constexpr int fft_tw_entries(int L, int E, int ts) { return fft_tw_offset(L, fft_nstages(L, E), E, ts); }
// position of (slot, thread class ku) inside stage S
__host__ __device__ constexpr int fft_tw_index(int L, int S, int E, int ts, int slot, int ku) {
    return fft_tw_paired(L, S, E, ts) ? ((slot >> 1) * fft_tw_ku(L, S, E) + ku) * 2 + (slot & 1) : slot * fft_tw_ku(L, S, E) + ku;
}
__host__ __device__ constexpr int fft_tw_lds_offset(int L, int S, int E) {   // position of stage S in the LDS copy
    return S <= 1 ? 0 : fft_tw_lds_offset(L, S - 1, E) + (fft_tw_via_lds(L, S - 1, E) ? fft_tw_slots_of(L, S - 1, E) * fft_tw_ku(L, S - 1, E) : 0);
}
__host__ __device__ constexpr int fft_tw_lds_entries(int L, int E) { return fft_tw_lds_offset(L, fft_nstages(L, E), E); }
// exponent q of W_L for (stage S, butterfly i, factor u, thread class ku)
__host__ __device__ constexpr int fft_tw_exponent(int L, int E, int S, int i, int u, int ku) {
    return ((ku + i * (L / E)) & (fft_ls(L, S, E) - 1)) * u * (L / (fft_ls(L, S, E) * fft_radix(L, S, E)));
}

// Phase 1 (before the workgroup's barrier): global -> LDS copy of the small stages, global -> register
// loads of the large ones.  `tid`/`nthreads` enumerate the whole workgroup.
template <typename T, int L, int E, int S>
__device__ __forceinline__ void tw_stage_issue(LineTw<T, L, E>& tw, const int j, const cx<T>* __restrict__ tab, cx<T>* ldsT,
                                               const int tid, const int nthreads) {
    constexpr int SLOTS = fft_tw_slots_of(L, S, E);
    constexpr int KU = fft_tw_ku(L, S, E);
    constexpr int OFF = fft_tw_offset(L, S, E, (int)sizeof(T));
    if constexpr (fft_tw_paired(L, S, E, (int)sizeof(T))) {
        typedef T w4_t __attribute__((ext_vector_type(4)));
        const w4_t* __restrict__ p4 = reinterpret_cast<const w4_t*>(tab + OFF) + (j & (KU - 1));
#pragma unroll
        for (int pr = 0; pr < (SLOTS + 1) / 2; ++pr) {
            const w4_t q = p4[pr * KU];
            tw.w[S - 1][2 * pr] = mk<T>(q.x, q.y);
            if (2 * pr + 1 < SLOTS) tw.w[S - 1][2 * pr + 1] = mk<T>(q.z, q.w);
        }
    } else if constexpr (fft_tw_via_lds(L, S, E)) {
        constexpr int LOFF = fft_tw_lds_offset(L, S, E);
        for (int e = tid; e < SLOTS * KU; e += nthreads) ldsT[LOFF + e] = tab[OFF + e];
    } else if constexpr (tw_lazy<T, E, L>()) {
        tw.g[S - 1] = tab + OFF + (j & (KU - 1));
    } else {
#pragma unroll
        for (int sl = 0; sl < SLOTS; ++sl) tw.w[S - 1][sl] = tab[OFF + sl * KU + (j & (KU - 1))];
    }
}
template <typename T, int L, int E, int S>
__device__ __forceinline__ void tw_stage_fetch(LineTw<T, L, E>& tw, const int j, const cx<T>* ldsT) {
    if constexpr (fft_tw_via_lds(L, S, E)) {
        constexpr int SLOTS = fft_tw_slots_of(L, S, E);
        constexpr int KU = fft_tw_ku(L, S, E);
        constexpr int LOFF = fft_tw_lds_offset(L, S, E);
        if constexpr (tw_lazy<T, E, L>()) {
            tw.l[S - 1] = ldsT + LOFF + (j & (KU - 1));
        } else {
#pragma unroll
            for (int sl = 0; sl < SLOTS; ++sl) tw.w[S - 1][sl] = ldsT[LOFF + sl * KU + (j & (KU - 1))];
        }
    }
}
// twiddle `slot` of stage S (S >= 1) of this thread
template <typename T, int L, int E, int S>
__device__ __forceinline__ cx<T> tw_get(const LineTw<T, L, E>& tw, const int slot) {
    if constexpr (tw_lazy<T, E, L>()) {
        constexpr int KU = fft_tw_ku(L, S, E);
        if constexpr (fft_tw_via_lds(L, S, E)) return tw.l[S - 1][slot * KU];
        else return tw.g[S - 1][slot * KU];
    } else {
        return tw.w[S - 1][slot];
    }
}
template <typename T, int L, int E>
__device__ __forceinline__ void line_twiddles_issue(LineTw<T, L, E>& tw, const int j, const cx<T>* __restrict__ tab, cx<T>* ldsT,
                                                    const int tid, const int nthreads) {
    constexpr int M = fft_nstages(L, E);
    if constexpr (M > 1) tw_stage_issue<T, L, E, 1>(tw, j, tab, ldsT, tid, nthreads);
    if constexpr (M > 2) tw_stage_issue<T, L, E, 2>(tw, j, tab, ldsT, tid, nthreads);
    if constexpr (M > 3) tw_stage_issue<T, L, E, 3>(tw, j, tab, ldsT, tid, nthreads);
}
// The same in three parts, for kernels whose head is a latency chain (k_time, k_freq): the SMALL stages' table entries are asked for
// FIRST -- before the field loads -- and go to LDS later, while the field is still on its way.  (Vector-memory results return in issue
// order: staged behind the field loads, as line_twiddles_issue does it, the LDS copy waits for the whole field to land, and the barrier and
// every load issued after it start a second round trip behind the first: 0.3-0.5 us per kernel, profiles/r04_c2_micro.txt.)
__host__ __device__ constexpr int fft_tw_prefetch_count(int L, int E, int NT, int S = 1) {
    return S >= fft_nstages(L, E) ? 0
         : (fft_tw_via_lds(L, S, E) ? (fft_tw_slots_of(L, S, E) * fft_tw_ku(L, S, E) + NT - 1) / NT : 0) + fft_tw_prefetch_count(L, E, NT, S + 1);
}
template <typename T, int L, int E, int NT> struct TwStaged {
    static constexpr int PF = fft_tw_prefetch_count(L, E, NT);
    cx<T> r[PF > 0 ? PF : 1];
};
template <typename T, int L, int E, int NT, int S>
__device__ __forceinline__ void tw_stage_prefetch(TwStaged<T, L, E, NT>& p, int& k, const cx<T>* __restrict__ tab, const int tid) {
    if constexpr (fft_tw_via_lds(L, S, E)) {
        constexpr int NS = fft_tw_slots_of(L, S, E) * fft_tw_ku(L, S, E), OFF = fft_tw_offset(L, S, E, (int)sizeof(T));
#pragma unroll
        for (int i = 0; i * NT < NS; ++i) {
            const int e = tid + i * NT;
            p.r[k++] = e < NS ? tab[OFF + e] : mk<T>((T)0, (T)0);
        }
    }
}
template <typename T, int L, int E, int NT, int S>
__device__ __forceinline__ void tw_stage_commit(const TwStaged<T, L, E, NT>& p, int& k, cx<T>* ldsT, const int tid) {
    if constexpr (fft_tw_via_lds(L, S, E)) {
        constexpr int NS = fft_tw_slots_of(L, S, E) * fft_tw_ku(L, S, E), LOFF = fft_tw_lds_offset(L, S, E);
#pragma unroll
        for (int i = 0; i * NT < NS; ++i) {
            const int e = tid + i * NT;
            if (e < NS) ldsT[LOFF + e] = p.r[k];
            ++k;
        }
    }
}
template <typename T, int L, int E, int NT>
__device__ __forceinline__ void line_twiddles_prefetch(TwStaged<T, L, E, NT>& p, const cx<T>* __restrict__ tab, const int tid) {
    constexpr int M = fft_nstages(L, E);
    int k = 0;
    if constexpr (M > 1) tw_stage_prefetch<T, L, E, NT, 1>(p, k, tab, tid);
    if constexpr (M > 2) tw_stage_prefetch<T, L, E, NT, 2>(p, k, tab, tid);
    if constexpr (M > 3) tw_stage_prefetch<T, L, E, NT, 3>(p, k, tab, tid);
}
template <typename T, int L, int E, int NT>
__device__ __forceinline__ void line_twiddles_commit(const TwStaged<T, L, E, NT>& p, cx<T>* ldsT, const int tid) {
    constexpr int M = fft_nstages(L, E);
    int k = 0;
    if constexpr (M > 1) tw_stage_commit<T, L, E, NT, 1>(p, k, ldsT, tid);
    if constexpr (M > 2) tw_stage_commit<T, L, E, NT, 2>(p, k, ldsT, tid);
    if constexpr (M > 3) tw_stage_commit<T, L, E, NT, 3>(p, k, ldsT, tid);
}
// the stages that every thread reads for itself (registers, or the lazy pointers)
template <typename T, int L, int E>
__device__ __forceinline__ void line_twiddles_issue_regs(LineTw<T, L, E>& tw, const int j, const cx<T>* __restrict__ tab) {
    constexpr int M = fft_nstages(L, E);
    if constexpr (M > 1 && !fft_tw_via_lds(L, 1, E)) tw_stage_issue<T, L, E, 1>(tw, j, tab, nullptr, 0, 1);
    if constexpr (M > 2 && !fft_tw_via_lds(L, 2, E)) tw_stage_issue<T, L, E, 2>(tw, j, tab, nullptr, 0, 1);
    if constexpr (M > 3 && !fft_tw_via_lds(L, 3, E)) tw_stage_issue<T, L, E, 3>(tw, j, tab, nullptr, 0, 1);
}
// Phase 2 (after the barrier): LDS -> registers
template <typename T, int L, int E>
__device__ __forceinline__ void line_twiddles_fetch(LineTw<T, L, E>& tw, const int j, const cx<T>* ldsT) {
    constexpr int M = fft_nstages(L, E);
    if constexpr (M > 1) tw_stage_fetch<T, L, E, 1>(tw, j, ldsT);
    if constexpr (M > 2) tw_stage_fetch<T, L, E, 2>(tw, j, ldsT);
    if constexpr (M > 3) tw_stage_fetch<T, L, E, 3>(tw, j, ldsT);
}

// One Stockham stage.  IDX maps a line element index to an LDS element index.  Exchanges alternate
// between two LDS buffers (`lds`, `lds + BUF`): exchange x uses buffer x & 1, so a buffer is rewritten
// only after a full barrier has separated it from its last readers and ONE barrier per exchange
// (between its writes and its reads) is enough.  XP = parity of this transform's first exchange.
// BUF = 0 selects a single buffer (half the LDS) with a second barrier before every rewrite.
template <typename T, int L, int E, int DIR, int S, int XP, typename IDX>
__device__ __forceinline__ void fft_stage(cx<T> (&v)[E], cx<T>* lds, const int BUF, const int j, const IDX& idx,
                                          const LineTw<T, L, E>& tw) {
    constexpr int M  = fft_nstages(L, E);
    constexpr int R  = fft_radix(L, S, E);
    constexpr int NB = E / R;           // butterflies per thread
    constexpr int Q  = L / E;           // threads per line
    constexpr int LS = fft_ls(L, S, E);

#ifdef SSFM_STAMP_FFT
    if (DIR < 0) SSFM_STAMP_FFT(7 + 3 * S);          // stage entry
#endif
    if constexpr (S > 0) {
        const cx<T>* src = lds + ((XP + S - 1) & 1) * BUF;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int jb = j + i * Q;
#pragma unroll
            for (int u = 0; u < R; ++u) v[i + u * NB] = src[idx(jb + u * (L / R))];
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
#pragma unroll
            for (int u = 1; u < R; ++u)
                v[i + u * NB] = cmuld<DIR>(v[i + u * NB], tw_get<T, L, E, S>(tw, i * (R - 1) + (u - 1)));
        }
    }
#ifdef SSFM_STAMP_FFT
    if (DIR < 0) SSFM_STAMP_FFT(8 + 3 * S);          // LDS read + twiddle done
#endif
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        cx<T> tmp[R];
#pragma unroll
        for (int u = 0; u < R; ++u) tmp[u] = v[i + u * NB];
        Dft<R>::template run<DIR>(tmp);
#pragma unroll
        for (int u = 0; u < R; ++u) v[i + u * NB] = tmp[u];
    }
#ifdef SSFM_STAMP_FFT
    if (DIR < 0) SSFM_STAMP_FFT(9 + 3 * S);          // butterflies done
#endif
    if constexpr (S < M - 1) {
        cx<T>* dst = lds + ((XP + S) & 1) * BUF;
        if (BUF == 0 && (S > 0 || XP != 0)) __syncthreads();   // single buffer: its readers must be done
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int jb = j + i * Q;
            const int k = jb & (LS - 1);
            const int base = (jb - k) * R + k;
#pragma unroll
            for (int u = 0; u < R; ++u) dst[idx(base + u * LS)] = v[i + u * NB];
        }
        __syncthreads();
    }
}

// Full line transform.  XP: parity of its first exchange (0 for the first transform of a kernel,
// (number of exchanges so far) & 1 for a later one).
template <typename T, int L, int E, int DIR, int XP, typename IDX>
__device__ __forceinline__ void fft_line(cx<T> (&v)[E], cx<T>* lds, const int BUF, const int j, const IDX& idx,
                                         const LineTw<T, L, E>& tw) {
    constexpr int M = fft_nstages(L, E);
    fft_stage<T, L, E, DIR, 0, XP, IDX>(v, lds, BUF, j, idx, tw);
    if constexpr (M > 1) fft_stage<T, L, E, DIR, 1, XP, IDX>(v, lds, BUF, j, idx, tw);
    if constexpr (M > 2) fft_stage<T, L, E, DIR, 2, XP, IDX>(v, lds, BUF, j, idx, tw);
    if constexpr (M > 3) fft_stage<T, L, E, DIR, 3, XP, IDX>(v, lds, BUF, j, idx, tw);
}

// The same with a hook between the first stage and the rest: `mid()` runs after the first stage's butterflies and LDS writes
// (a place to issue loads whose data is only needed after the transform).
template <typename T, int L, int E, int DIR, int XP, typename IDX, typename F>
__device__ __forceinline__ void fft_line_hook(cx<T> (&v)[E], cx<T>* lds, const int BUF, const int j, const IDX& idx,
                                              const LineTw<T, L, E>& tw, F&& mid) {
    constexpr int M = fft_nstages(L, E);
    fft_stage<T, L, E, DIR, 0, XP, IDX>(v, lds, BUF, j, idx, tw);
    mid();
    if constexpr (M > 1) fft_stage<T, L, E, DIR, 1, XP, IDX>(v, lds, BUF, j, idx, tw);
    if constexpr (M > 2) fft_stage<T, L, E, DIR, 2, XP, IDX>(v, lds, BUF, j, idx, tw);
    if constexpr (M > 3) fft_stage<T, L, E, DIR, 3, XP, IDX>(v, lds, BUF, j, idx, tw);
}

}  // namespace ssfm
