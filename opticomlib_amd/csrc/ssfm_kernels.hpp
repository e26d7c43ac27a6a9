// ssfm_kernels.hpp -- the split-step Fourier kernels for gfx950.
//
// N = N1 * N2 samples per row (four-step FFT).  Between kernels the field lives in the
// "half transformed" layout Y[k1][n2] (k1 < N1 column-frequency, n2 < N2 row-time index):
//
//   k_time  works on a tile of C columns x all N1 rows:  twiddle^-1 -> IFFT_N1 -> second
//           nonlinear half step of step k (stale |A|^2) -> |A|^2 -> first nonlinear half
//           step of step k+1 -> FFT_N1 -> twiddle.  Time-domain samples exist only in
//           registers.  (reference devices.py:1175,1177,1181)
//   k_freq  works on whole rows (fixed k1, all n2): FFT_N2 -> multiply by exp(D~ h)/N (table
//           stored in the transposed order this pass produces) -> IFFT_N2.
//           (reference devices.py:1178-1180)
//
// so one SSFM step is two launches and each reads and writes the field exactly once.
#pragma once
#include "wgfft.hpp"

// Timing-only ablation builds for tools/ablate.sh (results are WRONG when any bit is set): -DSSFM_ABLATE=<bits>
//   1 no inter-pass twiddle loads   2 no |A|^2 traffic   4 no operator table loads   8 no transforms   16 no nonlinear rotation
#ifndef SSFM_ABLATE
#define SSFM_ABLATE 0
#endif
#define SSFM_ABL_NO_TWN ((SSFM_ABLATE & 1) != 0)
#define SSFM_ABL_NO_P ((SSFM_ABLATE & 2) != 0)
#define SSFM_ABL_NO_TAB ((SSFM_ABLATE & 4) != 0)
#define SSFM_ABL_NO_FFT ((SSFM_ABLATE & 8) != 0)
#define SSFM_ABL_NO_NL ((SSFM_ABLATE & 16) != 0)

// Decided by measurement in rounds 2-4 and no longer switchable (the A/Bs are under profiles/: r02_ab_u16_sc1_twnc.txt, r02_ab_policy.txt,
// r02_c128_policy_ab.txt, r03_ablation_and_knobs.txt, r03_relaxed_barrier.txt, r04_c2_micro.txt, r04_c1_variants.txt):
//  * the small tables a workgroup stages through LDS are asked for BEFORE the field loads, k_freq's phase / operator loads go out before the field loads
//    and the factors are formed while the field is on its way; kernel arguments arrive preloaded in SGPRs;
//  * |A|^2 of the step's start is loaded after the inverse transform's first stage (every mode that reads it, both precisions);
//  * inter-pass twiddles W_N^(k1 n2) are formed in the kernel from two small tables: U16 plans (20.8 vs 21.6 us per step alone, 18.1 vs 20.7 with the memory
//    policy) and complex128 plans (43.7 -> 42.8); the write-through / non-temporal policy is for the U16 plans only (it loses 2 % in complex128);
//  * in-kernel barriers that only atomics or acknowledged write-through data cross carry no fence (an agent-scope fence writes back / invalidates the XCD's L2).
template <typename T, bool U16> __host__ __device__ constexpr bool stream_policy() { return U16; }
template <typename T, bool U16> __host__ __device__ constexpr bool twn_compute() { return U16 || sizeof(T) == 8; }

// Launch-level trace for tools/trace_timeline.py (diagnostic builds only: -DSSFM_TRACE=1).  Every
// workgroup folds its start / end time (s_memrealtime, 100 MHz) into 4 words of its launch's slot.
#ifndef SSFM_TRACE
#define SSFM_TRACE 0
#endif
#if SSFM_TRACE
#define SSFM_TRACE_ARGS unsigned long long* trace; int trace_slot;
// four words per workgroup: start, end, where it ran (XCC id << 32 | HW_ID: CU, SE, SIMD of the stamping wave), spare
#define SSFM_TRACE_BEGIN(a)                                                                     \
    if ((a).trace && threadIdx.x == 0) {                                                        \
        unsigned hw_, xc_;                                                                      \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_));                       \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xc_));                      \
        (a).trace[((long long)(a).trace_slot * 256 + (blockIdx.x & 255)) * 4 + 0] = __builtin_amdgcn_s_memrealtime(); \
        (a).trace[((long long)(a).trace_slot * 256 + (blockIdx.x & 255)) * 4 + 2] = ((unsigned long long)xc_ << 32) | hw_; \
    }
#define SSFM_TRACE_END(a)                                                                       \
    if ((a).trace && threadIdx.x == 0)                                                          \
        (a).trace[((long long)(a).trace_slot * 256 + (blockIdx.x & 255)) * 4 + 1] = __builtin_amdgcn_s_memrealtime();
#else
#define SSFM_TRACE_ARGS
#define SSFM_TRACE_BEGIN(a)
#define SSFM_TRACE_END(a)
#endif

namespace ssfm {

// BEGIN / END: first / last half of MID with the time-order field on one side.  BEGIN_Y / END_Y (U16 plans, adaptive
// runs): the same, but the time-domain field between END and the next BEGIN stays in the Y buffer in the tile-private
// 16-byte-unit order (only the same tile reads it back; write-through stores, non-temporal loads) instead of going
// through time order with 8-byte accesses; UNPACK turns that buffer into the time-order field at the end of the run.
// TM_MID_A: the MID of an ADAPTIVE run of a plan whose column kernel is at most 128 workgroups (2^14 ... 2^18 samples): END and
// the next BEGIN in one launch, the workgroups waiting INSIDE the kernel for the global max |A|^2 (a grid barrier over that few
// workgroups costs 1-2 us, profiles/r02_barrier_probe.txt; the third launch it replaces 3.5 us).
// (TM_MID_L, k_medium_chirp only: the column pass BETWEEN the two convolutions of a chirp-z step -- inverse transform, product with a table in the
// time order, forward transform; no nonlinear operator, the |A|^2 the thread keeps is left alone)
enum TimeMode { TM_BEGIN = 0, TM_MID = 1, TM_END = 2, TM_BEGIN_Y = 3, TM_END_Y = 4, TM_UNPACK = 5, TM_MID_A = 6, TM_MID_L = 7 };
__host__ __device__ constexpr bool tm_inverse(int m) { return m == TM_MID || m == TM_END || m == TM_END_Y || m == TM_MID_A || m == TM_MID_L; }     // starts in the half-transformed domain
__host__ __device__ constexpr bool tm_forward(int m) { return m == TM_BEGIN || m == TM_MID || m == TM_BEGIN_Y || m == TM_MID_A || m == TM_MID_L; } // ends in it
__host__ __device__ constexpr bool tm_ends(int m) { return m == TM_END || m == TM_END_Y; }

// Device-resident step control of the adaptive mode (reference devices.py:1155-1161,1193-1196).
//
// State S_s = (z reached, step h about to be taken, done, steps taken) BEFORE step s lives in cur[s & 1].  END(s)
// only accumulates max |A|^2: one atomicMax per workgroup on slot (block id mod 64) of slots[s & 1] -- thousands of
// waves on ONE address serialise (about 13 ns each).  The next BEGIN(s + 1) derives S_(s+1) = f(S_s, max) itself:
// every wavefront reads the 64 slots and replays devices.py:1173,1193-1196 (the same float operations in every
// wavefront: the same bits), workgroup 0 also records S_(s+1) in cur[(s + 1) & 1], z in zlog, and clears the slots
// END(s + 1) will fill.  No ticket chain, no last-arrival logic, no fence: everything a kernel reads was written by an
// EARLIER kernel of the stream.  (Round 1 let the last workgroup of END run the step control behind a chain of
// dependent device-scope atomics: 4 us of END's 14.)  k_adapt_finish derives the state after the last launched step
// for the host; the first BEGIN of the next chunk is told that cur[] already holds its state.
template <typename T> struct StepState {
    T z;            // position reached
    T h;            // step about to be taken
    int done;
    int steps;      // steps taken so far
};
constexpr int kAdaptSlots = 64;
constexpr int kAdaptWords = 512;       // workgroups of the largest fused adaptive column kernel (2^20 x 2: 256 tiles x 2 rows)
template <typename T> struct AdaptState {
    T length;
    T phi_max;
    T abs_gamma;
    int adaptive;   // 0: keep h (only clamp), 1: h = phi_max / max(|gamma| |A|^2)
    int max_steps;
    int pad_;
    StepState<T> cur[2];
    unsigned long long slots[2][kAdaptSlots];      // bit patterns of max |A|^2 (non-negative => monotone as integers); [step parity][slot]
    // TM_MID_A with more than kAdaptSlots workgroups (complex64): one word per workgroup and step parity, (step + 1) << 32 | bits of its
    // maximum -- flag and value in ONE 8-byte store, no atomics, no counter that 512 workgroups would serialise on (a counter barrier
    // over 512 workgroups costs 7-9 us, profiles/r02_barrier_probe.txt; this hand-over about 2).  Zeroed by the host before a run.
    unsigned long long wgmax[2][kAdaptWords];
    unsigned arrive[2];     // TM_MID_A: workgroups that have delivered their maximum for the step of this parity
    unsigned error;         // TM_MID_A: a workgroup gave up waiting (the GPU did not run the whole grid at once): the host falls back
    long long patience;     // TM_MID_A: ticks of the 100 MHz clock a workgroup waits for the others (20 ms; tests set 0)
};

// Block id -> (unit, row) so that the `rows` blocks working on the same unit (column tile / spectrum
// row) share its read-only table through ONE XCD's L2: blocks are dealt round-robin over the 8 XCDs
// (id % 8 labels the group, MI355X_MICROARCH.md), so the rows of a unit take consecutive slots of one
// group.  Pure speed: any mapping that is a bijection is correct.
__device__ __forceinline__ void xcd_unit_row(unsigned id, int units, int rows, int& unit, int& row) {
    if (rows == 1) {                       // (a lane's launch: no division at the head of the kernel)
        unit = (int)id;
        row = 0;
    } else if ((units & 7) == 0) {
        const unsigned xcd = id & 7, slot = id >> 3;
        unit = (int)((slot / rows) * 8 + xcd);
        row = (int)(slot % rows);
    } else {
        unit = (int)(id % units);
        row = (int)(id / units);
    }
}

// Read-only tables are stored in the order the consuming thread holds them, two register slots
// interleaved, so that every access is 16 bytes per lane and coalesced (a vector-memory instruction
// costs one wave ~70 cycles of issue whether it moves 8 or 16 bytes per lane, and with one wave per
// SIMD that issue time is on the workgroup's critical path).
// k_freq operator table, row k1: element k2 = j + t*Q  ->  ((t>>1)*Q + j)*2 + (t&1)
__host__ __device__ __forceinline__ long long freq_tab_pos(long long k2, int Q) {
    const long long j = k2 % Q, t = k2 / Q;
    return ((t >> 1) * Q + j) * 2 + (t & 1);
}
// k_time inter-pass twiddle W_N^(k1 n2): tile = n2 / C, thread = (k1 % Q1) * C + n2 % C, slot t = k1 / Q1
//   -> ((tile*(E/2) + (t>>1)) * NT + thread) * 2 + (t&1),   NT = Q1*C threads per tile
__host__ __device__ __forceinline__ long long time_tw_pos(long long k1, long long n2, int Q1, int C, int E) {
    const long long tile = n2 / C, thread = (k1 % Q1) * C + n2 % C, t = k1 / Q1;
    return ((tile * (E / 2) + (t >> 1)) * ((long long)Q1 * C) + thread) * 2 + (t & 1);
}

// ---- "U16" field layout between the two kernels (complex64): 16-byte units.
// A vector-memory instruction costs a wave about the same issue time whether a lane moves 8 or 16 bytes, a
// write-through (sc1) store only runs at the plain rate with 16 bytes per lane, and a kernel's loads and stores
// are on its critical path (one wave per SIMD, every workgroup of a launch in step).  k_freq's thread j holds row
// elements n2 = j + Qf t: the elements of register slots t = 2g and 2g+1 are stored side by side as ONE 16-byte
// unit at position freq_tab_pos(n2, Qf) of the row -- the operator table's own order.  A k_time tile is then 8
// consecutive units = one 128-byte line per row k1 = the columns {jf .. jf+7} + Qf (2 gf + h), h = 0, 1; its lane
// (h, j mod 4, c8) loads the unit of row j + Q1 (2g + h), and ONE v_permlane32_swap per dword hands the h = 1
// column's half to lane + 32 and takes that lane's h = 0 half (wavefront shuffle instead of a second pass
// through memory): afterwards every thread holds rows 2g and 2g+1 of ITS column, as in the plain layout.
template <typename T> __host__ __device__ constexpr bool u16_layout(int N1, int C, int E) {
    return sizeof(T) == 4 && C == 16 && ((N1 / E) % 4) == 0 && N1 / E >= 4;
}
// natural column of position `pos` of a row in the U16 order (inverse of freq_tab_pos)
__host__ __device__ __forceinline__ long long u16_col_of_pos(long long pos, int Qf) {
    const long long u = pos >> 1, g = u / Qf, j = u % Qf;
    return j + (long long)Qf * (2 * g + (pos & 1));
}
// tile order: tiles 2m and 2m+1 share every 128-byte line of the NATURAL-order side (8 + 8 columns), so they
// take consecutive slots of one XCD (block ids b and b + 8): the second finds the line in that XCD's L2
__host__ __device__ __forceinline__ int u16_tile_of_unit(int t, int ntiles) {
    return (ntiles & 15) == 0 ? ((t & ~15) | ((t & 7) << 1) | ((t >> 3) & 1)) : t;
}
// lanes 32-63 of `lo` <-> lanes 0-31 of `hi` (v_permlane32_swap_b32), both components of a complex pair
__device__ __forceinline__ void lane32_swap(cf32& lo, cf32& hi) {
    const auto rx = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo.x), __float_as_uint(hi.x), false, false);
    const auto ry = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo.y), __float_as_uint(hi.y), false, false);
    lo = mk<float>(__uint_as_float(rx[0]), __uint_as_float(ry[0]));
    hi = mk<float>(__uint_as_float(rx[1]), __uint_as_float(ry[1]));
}
__device__ __forceinline__ void lane32_swap(cf64&, cf64&) {}      // (complex128 elements are 16 bytes already)
// k_time inter-pass twiddle in the U16 order: the tile and the column inside it follow the row position
__host__ __device__ __forceinline__ long long time_tw_pos_u16(long long k1, long long n2, int Q1, int E, int Qf, int ntiles) {
    const long long pos = freq_tab_pos(n2, Qf);
    const long long tile = pos / 16, c = ((pos & 1) << 3) | ((pos >> 1) & 7), thread = (k1 % Q1) * 16 + c, t = k1 / Q1;
    (void)ntiles;
    return ((tile * (E / 2) + (t >> 1)) * ((long long)Q1 * 16) + thread) * 2 + (t & 1);
}

// A chirp-z step's two ends folded into the column passes of the plain layout (csrc/chirpz.hip has the algebra; chirp == nullptr: off).
//   TM_BEGIN reads the caller's field A (n samples per row, natural order) instead of the plan's: |A|^2 -> P, half a nonlinear step, times the chirp c,
//            zeros from n up to the plan's length -- what k_chirp_pre wrote into the plan's field for BEGIN to read back;
//   TM_END   writes A instead of the plan's field: times conj(c) * scale, the other half nonlinear step with the P of BEGIN, and the maximum of
//            |A|^2 into *maxbits (nullable) -- what k_chirp_post did from the field END had stored.
// Steps driven from the device (ssfm_chirp_propagate, adaptive): the step size comes from *h, and a set *done turns the pass into a no-op.
template <typename T> struct ChirpIO {
    const cx<T>* chirp;
    cx<T>* A;
    T* P;
    long long n;
    T gamma, hh, scale;
    const double* h;
    const int* done;
    unsigned long long* maxbits;
};

template <typename T> struct TimeArgs {
    cx<T>* F;                 // field in time order, batch rows of N (read by BEGIN, written by END)
    cx<T>* Y;                 // field in the half-transformed layout between the kernels.  Plain layout: Y == F, a tile
                              // reads and writes the same columns, every kernel works in place.  U16 layout: a tile's
                              // columns and its positions in the row order differ, so BEGIN / END go from one buffer to the other
    T* P;                     // stale |A|^2, tile-major (private to k_time)
    const cx<T>* twN;         // W_N^(k1*n2) at [k1*N2 + n2]                      (plans without the in-kernel twiddles)
    const cx<T>* twA;         // [tile][j][c] = W_N^(j n2),        j < N1/E   W_N^(k1 n2) = twA * twB,  k1 = j + t N1/E,
    const cx<T>* twB;         // [tile][t][c] = W_N^(t n2 N1/E),   t < E      n2 = column c of the tile  (twn_compute plans)
    const cx<T>* tw1;         // W_N1^q
    AdaptState<T>* st;        // nullptr in fixed-step mode
    T* zlog;                  // adaptive mode: z after every step
    T gamma;
    T hh_prev;                // h/2 of the step being finished
    T hh_next;                // h/2 of the step being started
    int N2;
    int rows;                 // rows covered by this launch (grid = N2/C * rows blocks)
    int Qf;                   // threads per row of k_freq (U16 layout: which columns form a tile)
    int step;                 // adaptive mode: index of the step this launch belongs to (its state is cur[step & 1])
    int derive;               // adaptive BEGIN: 1 = derive the step's state from the previous step's (see AdaptState)
    const StepState<T>* s_in; // k_medium_adapt (PK): TM_MID_A takes the state of the step it finishes from here (LDS) instead of st->cur[],
    StepState<T>* s_out;      // ... and leaves the next step's here (untouched if the hand-over ran out of patience)
    const cx<T>* mul;         // TM_MID of the plain layout: the time-domain samples are multiplied by mul[position in the row] between the inverse and the
                              // forward pass (chirp-z: exp(D~ h) between Bluestein's two convolutions -- one column launch instead of END, a
                              // pointwise kernel and BEGIN); nullptr otherwise
    T* pkeep;                 // PK: the thread's E values of |A|^2 stay in registers from one column pass to the next (the same workgroup has the tile
                              // every time) instead of going through the P buffer
    int keep = 0;             // PK, TM_MID: > 0 = the samples from this position of the row on are set to zero behind the rotation (chirp-z: the line's padding)
    double* scal = nullptr;   // per-step scalar log of a z-resolved capture (ssfm_propagate_fixed_capture), this launch's step and first row: [row][wavefront of the
                              // launch's row][2] doubles -- every wavefront stores the sum and the maximum of ITS |A|^2 (one 16-byte store, no atomics: k_scal_reduce
                              // adds them up in a fixed order behind the run); nullptr: none
    ChirpIO<T> cz = {};       // chirp-z steps (plain layout, TM_BEGIN / TM_END): see ChirpIO
    SSFM_TRACE_ARGS
};

// sin/cos of the nonlinear phase.  float: Cody-Waite reduction by pi/2 (three fma terms whose sum
// is pi/2 to double precision) + the fdlibm float kernels; absolute error < 9e-8 for |x| <= 65000
// (0.5 ulp for the small phases an SSFM step produces).  Larger arguments are reduced in double
// (good to |x| ~ 1e14; beyond that one float ulp spans many turns and the phase carries no
// information, but the result is still a unit rotation).  The callers test once per thread whether
// any of its 16 phases needs the slow reduction, so the common path has no branches.
__device__ __forceinline__ void sincos_reduced(float r, int q, float& s, float& c) {
    const float z = r * r;
    float ps = fmaf(z, 2.7183114939898219064e-6f, -1.98393348360966317347e-4f);
    ps = fmaf(z, ps, 8.3333293858894631756e-3f);
    ps = fmaf(z, ps, -1.66666666416265235595e-1f);
    const float sr = fmaf(r * z, ps, r);
    float pc = fmaf(z, 2.43904487962774090654e-5f, -1.38867637746099294692e-3f);
    pc = fmaf(z, pc, 4.16666233237390631894e-2f);
    pc = fmaf(z, pc, -4.99999997251031003120e-1f);
    const float cr = fmaf(z, pc, 1.0f);
    const float a = (q & 1) ? cr : sr;
    const float b = (q & 1) ? sr : cr;
    s = (q & 2) ? -a : a;
    c = ((q + 1) & 2) ? -b : b;
}
constexpr float kSincosSmallMax = 65000.0f;
constexpr float kSincosTinyMax = 0.78f;       // |x| <= pi/4: the kernels apply without any reduction
// |x| <= pi/4 (every nonlinear phase of a sane split-step run): no reduction, no quadrant select
__device__ __forceinline__ void sincos_tiny(float r, float& s, float& c) {
    const float z = r * r;
    float ps = fmaf(z, 2.7183114939898219064e-6f, -1.98393348360966317347e-4f);
    ps = fmaf(z, ps, 8.3333293858894631756e-3f);
    ps = fmaf(z, ps, -1.66666666416265235595e-1f);
    s = fmaf(r * z, ps, r);
    float pc = fmaf(z, 2.43904487962774090654e-5f, -1.38867637746099294692e-3f);
    pc = fmaf(z, pc, 4.16666233237390631894e-2f);
    pc = fmaf(z, pc, -4.99999997251031003120e-1f);
    c = fmaf(z, pc, 1.0f);
}
template <bool BIG> __device__ __forceinline__ void sincos_f32(float x, float& s, float& c) {
    float r;
    int q;
    if (!BIG) {
        const float k = rintf(x * 0.6366197723675814f);
        r = fmaf(k, -1.570770263671875f, x);
        r = fmaf(k, -2.6063062250614166e-05f, r);
        r = fmaf(k, -6.077094383272197e-11f, r);
        q = (int)k;
    } else {
        const double xd = (double)x;
        const double kd = rint(xd * 0.63661977236758134308);
        double rd = fma(kd, -1.57079632679489655800e+00, xd);
        rd = fma(kd, -6.12323399573676603587e-17, rd);
        r = (float)rd;
        q = (int)((long long)kd & 3);
    }
    sincos_reduced(r, q, s, c);
    if (BIG && !(fabsf(x) < __builtin_inff())) s = c = __builtin_nanf("");
}
// |x| <= 2^-5 (the phase of a split step that resolves the nonlinearity: phi_max is 0.01 ... 0.05 rad by default, a fixed-step run of
// the headline configuration turns 4e-4 rad per half step): cos x = 1 - z/2 + z^2/24 (truncation z^3/720 < 2e-12), sin x = x (1 - z/6)
// (relative truncation z^2/120 < 8e-9, an eighth of the float32 half-ulp) with z = x^2 -- both halves of ONE packed fma, then one fma
// and one product: 4 instructions per point instead of 11 (k_time spends 2/3 of a SIMD's issue slots when both lanes run, so instructions
// are time: profiles/r04_c2_micro.txt)
constexpr float kSincosMicroMax = 0.03125f;
__device__ __forceinline__ cf32 expi_micro(float x) {
    const float z = x * x;
    const cf32 acc = mk<float>(4.16666679084300994873e-2f, -1.66666671633720397949e-1f) * mk<float>(z, z) + mk<float>(-0.5f, 1.0f);
    return mk<float>(fmaf(acc.x, z, 1.0f), acc.y * x);
}
// rotate E values by their phases
template <int E> __device__ __forceinline__ void rotate_all(cf32 (&v)[E], const float (&phi)[E]) {
    float amax = 0.0f;
#pragma unroll
    for (int t = 0; t < E; ++t) amax = fmaxf(amax, fabsf(phi[t]));
    const bool big = !(amax <= kSincosSmallMax);           // also true for NaN
    if (__builtin_expect(amax <= kSincosMicroMax, 1)) {
#pragma unroll
        for (int t = 0; t < E; ++t) v[t] = cmul(v[t], expi_micro(phi[t]));
    } else if (__builtin_expect(amax <= kSincosTinyMax, 1)) {
#pragma unroll
        for (int t = 0; t < E; ++t) {
            float s, c;
            sincos_tiny(phi[t], s, c);
            v[t] = cmul(v[t], mk<float>(c, s));
        }
    } else if (__builtin_expect(big, 0)) {
#pragma unroll
        for (int t = 0; t < E; ++t) {
            float s, c;
            sincos_f32<true>(phi[t], s, c);
            v[t] = cmul(v[t], mk<float>(c, s));
        }
    } else {
#pragma unroll
        for (int t = 0; t < E; ++t) {
            float s, c;
            sincos_f32<false>(phi[t], s, c);
            v[t] = cmul(v[t], mk<float>(c, s));
        }
    }
}
// double, |x| <= pi/4: the fdlibm kernels (__kernel_sin / __kernel_cos without the tail argument), < 1 ulp.  The library
// sincos() carries its argument reduction (Cody-Waite and Payne-Hanek paths, ~60 float64 instructions and a dozen
// branches per call): 40 % of the float64 instructions of k_time<double>, which is bound by them.
__device__ __forceinline__ void sincos_tiny(double x, double& s, double& c) {
    const double z = x * x;
    double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = fma(z, ps, 2.75573137070700676789e-06);
    ps = fma(z, ps, -1.98412698298579493134e-04);
    ps = fma(z, ps, 8.33333333332248946124e-03);
    ps = fma(z, ps, -1.66666666666666324348e-01);
    s = fma(x * z, ps, x);
    double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = fma(z, pc, -2.75573143513906633035e-07);
    pc = fma(z, pc, 2.48015872894767294178e-05);
    pc = fma(z, pc, -1.38888888888741095749e-03);
    pc = fma(z, pc, 4.16666666666666019037e-02);
    const double hz = 0.5 * z;
    c = 1.0 - (hz - z * (z * pc));
}
template <int E> __device__ __forceinline__ void rotate_all(cf64 (&v)[E], const double (&phi)[E]) {
    double amax = 0.0;
#pragma unroll
    for (int t = 0; t < E; ++t) amax = fmax(amax, fabs(phi[t]));
    if (__builtin_expect(amax <= (double)kSincosTinyMax, 1)) {          // (false for NaN: the library path propagates it)
#pragma unroll
        for (int t = 0; t < E; ++t) {
            double s, c;
            sincos_tiny(phi[t], s, c);
            v[t] = cmul(v[t], mk<double>(c, s));
        }
    } else {
#pragma unroll
        for (int t = 0; t < E; ++t) {
            double s, c;
            sincos(phi[t], &s, &c);
            v[t] = cmul(v[t], mk<double>(c, s));
        }
    }
}
// amp * (cos x, sin x) for |x| <= pi/2 as ONE packed polynomial in z = x^2: cos and sin are the two halves of the accumulator
// (5 v_pk_fma_f32); leading coefficients 1, -1/2 and 1 exact, the rest least-squares fits on [-pi/2, pi/2] rounded to
// float32.  Against the correctly rounded values: rms 4e-8, |c|^2 + |s|^2 - 1 = -1e-8 on average (correctly rounded
// float32 pairs themselves: 0 +- 8e-8).
__device__ __forceinline__ cf32 expi_half_turn(float x, float amp) {
    const float z = x * x;
    const cf32 z2 = mk<float>(z, z);
    cf32 acc = mk<float>(-2.6185691126556776e-07f, -2.39068338458992e-08f);
    acc = acc * z2 + mk<float>(2.4768854927970096e-05f, 2.7526464236871107e-06f);
    acc = acc * z2 + mk<float>(-0.0013888560933992267f, -0.00019840890308842063f);
    acc = acc * z2 + mk<float>(0.041666656732559204f, 0.008333330973982811f);
    acc = acc * z2 + mk<float>(-0.5f, -0.1666666716337204f);
    acc = acc * z2 + mk<float>(1.0f, 1.0f);
    return acc * mk<float>(amp, amp * x);
}
// amp * exp(i x), |x| <= kSincosSmallMax: Cody-Waite reduction by pi (k = rint(x / pi), three fma terms whose sum is pi to
// double precision) leaves |r| <= pi/2, an odd k flips the sign of both components (folded into amp) -- no quadrant
// selects: about 16 instructions instead of 33 per point of the on-the-fly operator.
__device__ __forceinline__ cf32 expi_f32(float x, float amp) {
    const float k = rintf(x * 0.3183098861837907f);
    float r = fmaf(k, -3.14154052734375f, x);
    r = fmaf(k, -5.2126124501228333e-05f, r);
    r = fmaf(k, -1.2154188766544394e-10f, r);
    const unsigned flip = (unsigned)(int)k << 31;
    return expi_half_turn(r, __uint_as_float(__float_as_uint(amp) ^ flip));
}
// scalar form
template <typename T> __device__ __forceinline__ T exp_acc(T x);
template <typename T> __device__ __forceinline__ void sincos_acc(T x, T& s, T& c);
template <> __device__ __forceinline__ void sincos_acc<float>(float x, float& s, float& c) {
    if (__builtin_expect(fabsf(x) <= kSincosSmallMax, 1)) sincos_f32<false>(x, s, c);
    else sincos_f32<true>(x, s, c);
}
template <> __device__ __forceinline__ void sincos_acc<double>(double x, double& s, double& c) { sincos(x, &s, &c); }

// Memory policy of the streamed data (field, |A|^2) of the U16 plans.  The consumer of every store is the NEXT
// kernel on other CUs -- the all-to-all between the two passes crosses XCDs -- so nothing is gained by keeping the
// lines in this XCD's L2, and a kernel that leaves megabytes dirty there pays for their write-back at its end:
// dependent-launch gap = 1.45 us + dirty bytes / 6 TB/s (MI355X_MICROARCH.md, row "boundary").  Measured
// (tools/trace_timeline.py, profiles/r02_launch_timeline.txt): gap 2.4 us behind plain stores, 1.5 us behind
// write-through ones; timing-only ablations that REMOVE the table loads make the kernels slower (23.5 vs 21.2 us per
// step, profiles/r02_ablation_lanes2.txt): with less read traffic fewer dirty lines are evicted while the kernel runs.
//   field stores write-through (sc1).  Needs the 16-byte units of the U16 layout: an 8-byte sc1 store runs at 0.4x the rate (round 1 measured 22.4 vs
//     21.2 us per step with it);
//   the |A|^2 buffer likewise (its 16-byte tile-major stores);
//   field and |A|^2 are read once per kernel: non-temporal loads.  Only together with the write-through stores (alone it LOSES: 23.7 vs 21.6 us per
//     step -- again the dirty lines).
// A/B on MI355X, C2 single field / 4 fields resident, us per (field-)step (profiles/r02_ab_u16_sc1_twnc.txt, r02_ab_policy.txt):
//   plain 21.6 / 15.0   sc1 22.3 / 15.7   sc1 + P 20.7 / 15.4   + in-kernel twiddles 19.7 / 15.0   + nt loads 18.1 / 15.1
// Everything outside the U16 path (complex128, the small 8-byte layouts, the time-order side) keeps plain accesses.
// P16: the stale |A|^2 of the large complex64 plans (16 points per thread, unit layout) crosses the launch boundary as 16-bit fixed
// point relative to the THREAD's own maximum: 2 x 16 bytes + one 4-byte scale per thread instead of 4 x 16 bytes -- 9 instead of 16 of the 91
// bytes a dual-polarisation sample*step moves.  |A|^2 only ever enters the phase h/2 gamma |A|^2 (devices.py:1175,1181), so what counts is the
// ABSOLUTE error of that phase: at most 2^-17 of the thread's largest phase (7.6e-6 of <= 0.05 rad; the float32 product itself carries 6e-8
// relative).  The buffer is private to a tile (the same thread reads back what it wrote), so the scale needs no agreement between threads.
// Stated bound: absolute phase error <= 2^-17 x the thread's largest half-step phase, per half step -- inside the suite's 2e-5 @ 100 steps for peaks up to
// ~0.1 rad per half step (tests/test_gpu_parity.py::test_the_16_bit_stale_power_holds_its_stated_phase_bound; the reference's adaptive rule keeps 0.005 ... 0.05,
// the benchmark configurations 4e-4).  A fixed h that turns more per step is outside it (as it is outside any sensible splitting error).
template <typename T, bool U16, int E> __host__ __device__ constexpr bool p16_layout() { return U16 && sizeof(T) == 4 && E == 16; }
template <bool NT, typename V> __device__ __forceinline__ V stream_load(const V* p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));
// 16-byte unit of the U16 layout
__device__ __forceinline__ void stream_store(f32x4* p, f32x4 v) {
    // The s_nop 1 are the two wait states a VALU write to the data registers of a > 8-byte store needs after it on
    // gfx940+: the compiler inserts them for its own stores but cannot see into an asm statement (without them the
    // next address computation overwrote the data of the store before it had been read)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void stream_store(f64x4* p, f64x4 v) { *p = v; }
__device__ __forceinline__ void stream_store(cf32* p, cf32 v) { *p = v; }
__device__ __forceinline__ void stream_store(cf64* p, cf64 v) { *p = v; }

// k_time's exchange buffer: element e of column c at [e][c].  SW = 0: as it is -- the plain lane order (16 lanes = the 16
// columns of one row) reads and writes whole 128-byte rows, conflict-free.  SW = log2(first radix) (U16 lane order: a
// 16-lane group = 8 columns of TWO rows j, a 32-lane group = 8 columns of FOUR rows): the two halves of a row swap places
// according to bits of e chosen so that both the first stage's writes (rows 2^SW apart per lane pair) and the next stage's
// reads (consecutive rows) spread over all banks; measured 40 % of k_time's LDS cycles were bank conflicts without it.
template <int C, int SW = 0> struct ColIdx {
    int c;
    __device__ __forceinline__ int operator()(int e) const {
        if constexpr (SW == 0) return e * C + c;
        else return e * C + (c ^ ((((e >> 1) ^ (e >> SW)) & 1) << 3));
    }
};
// row layout in LDS: one pad element per 2^SH elements (SH = log2 of the first-stage radix), which
// makes the stride-R first-stage writes and the contiguous later reads conflict-free
template <int SH> struct RowIdx {
    int off;
    __device__ __forceinline__ int operator()(int e) const { return off + e + (e >> SH); }
};
__host__ __device__ constexpr int row_pad_shift(int E) { return E == 16 ? 4 : 3; }
__host__ __device__ constexpr int row_lds_elems(int n2, int E) { return n2 + (n2 >> row_pad_shift(E)); }

template <typename T> __device__ __forceinline__ unsigned long long float_bits(T v);
template <> __device__ __forceinline__ unsigned long long float_bits<float>(float v) { return (unsigned long long)__float_as_uint(v); }
template <> __device__ __forceinline__ unsigned long long float_bits<double>(double v) { return (unsigned long long)__double_as_longlong(v); }

template <typename T> __device__ __forceinline__ T bits_float(unsigned long long b);
template <> __device__ __forceinline__ float bits_float<float>(unsigned long long b) { return __uint_as_float((unsigned)b); }
template <> __device__ __forceinline__ double bits_float<double>(unsigned long long b) { return __longlong_as_double((long long)b); }

// S_(s+1) from S_s and the maximum of |A|^2 after step s (reference devices.py:1173, 1193-1196)
template <typename T> __device__ __forceinline__ StepState<T> step_advance(const AdaptState<T>* st, const StepState<T> prev, unsigned long long maxbits) {
    if (prev.done) return prev;
    StepState<T> n;
    n.z = prev.z + prev.h;
    T h = prev.h;
    if (st->adaptive) h = st->phi_max / (st->abs_gamma * bits_float<T>(maxbits));
    const T rem = st->length - n.z;
    n.h = h < rem ? h : rem;
    n.steps = prev.steps + 1;
    n.done = !(n.z < st->length) || n.steps >= st->max_steps;
    return n;
}
// maximum over the 64 slots, by one wavefront (every lane gets it)
template <typename T> __device__ __forceinline__ unsigned long long slots_max(const AdaptState<T>* st, int parity) {
    const int i = threadIdx.x & (kAdaptSlots - 1);
    unsigned long long mb = __hip_atomic_load(&st->slots[parity][i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (filled by memory-side atomics of an earlier kernel)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(mb, o);
        mb = other > mb ? other : mb;
    }
    return mb;
}
// The state of step `step` as every wavefront of a BEGIN / k_freq / END kernel sees it.  derive: BEGIN of a step whose
// state is not in cur[] yet.
template <typename T> __device__ __forceinline__ StepState<T> step_state(const AdaptState<T>* st, int step, bool derive) {
    if (!derive) return st->cur[step & 1];
    return step_advance<T>(st, st->cur[(step - 1) & 1], slots_max<T>(st, (step - 1) & 1));
}


// ------------------------------------------------------------------------------ k_time
// (Alternating the LDS exchanges between two buffers saves one barrier per exchange but doubles the LDS footprint; measured on MI355X it is SLOWER,
// 23.2 -> 25.6 us per step in round 1 and 18.1 vs 16.5 in round 4: fewer workgroups of the concurrently running kernels fit a CU.  Removed in round 5.)
// waves per SIMD to ask for: enough for two workgroups of THREADS threads per CU in complex64
// (a 512-thread workgroup with 16 points per thread needs the full 256 registers: one workgroup per CU)
__host__ __device__ constexpr int min_waves(int threads, int tsize, int e = 16) {
    return tsize == 4 && threads >= 256 ? (threads == 256 ? 2 : (e == 16 ? 2 : threads / 128)) : 1;
}
// complex64 workgroups of 256 threads are meant to sit two per CU -- a launch of one lane is 256 of them, both lanes together fill the chip evenly.  A
// kernel that needs fewer than 171 registers would be admitted three per CU and the dispatcher packs them unevenly; the attribute pins the register
// budget to two waves per SIMD.
#define SSFM_KERNEL_BOUNDS(threads, tsize, e) \
    __launch_bounds__(threads) __attribute__((amdgpu_waves_per_eu(min_waves(threads, tsize, e), (tsize) == 4 && (threads) == 256 ? 2 : 8)))
// 16 bytes at byte offset `off` of the buffer behind `rsrc`, sc1: agent-scope coherent (served past this CU's L1) and tracked by the
// compiler's wait counters -- how a workgroup reads what ANOTHER workgroup of the same launch has stored (k_medium)
// k_medium's workgroups all sit on ONE XCD and meet in that XCD's L2 -- plain stores (the CU's L1 writes through), loads with sc0 nt (they miss
// the L1 and are answered by the L2), the barrier a flag word per workgroup: a store + barrier + load round costs 1.1-1.3 us for 16-32 workgroups
// against 2.1-2.8 us across the XCDs, which meet in memory (tools/xcd_barrier_probe.hip, profiles/r03_xcd_barrier.txt; sc0 alone or an L1
// invalidate + plain load return STALE data).  (The form across the XCDs -- sc1 stores / sc1 loads / agent-scope atomics -- was round 3's first
// k_medium, 8.0-12.0 vs 8.0-10.5 us per step, removed in round 5; |A|^2 stays in registers between a workgroup's column passes: -4...-5 %.)
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4_t load16_sc1(__amdgpu_buffer_rsrc_t rsrc, int off) {
    return __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 3 /* sc0 nt */);
}
// a field store of a pass: write-through to memory, except inside k_medium on one XCD (see above)
template <bool PK, typename P, typename V> __device__ __forceinline__ void pass_store(P* p, V v) {
    if constexpr (PK) *p = v;
    else stream_store(p, v);
}
// an atomic add executed in this XCD's L2, returning the old value
__device__ __forceinline__ unsigned long long l2_add_u64(unsigned long long* p, unsigned long long v) {
    unsigned long long old;
    asm volatile("global_atomic_add_x2 %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(old) : "v"(p), "v"(v) : "memory");
    return old;
}
__device__ __forceinline__ unsigned long long ld_l2_u64(const unsigned long long* p) {         // misses the CU's L1, answered by the XCD's L2
    unsigned long long v;
    asm volatile("global_load_dwordx2 %0, %1, off sc0 nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void st_l2_u64(unsigned long long* p, unsigned long long v) {       // a plain store: written through the L1 into the XCD's L2
    asm volatile("global_store_dwordx2 %0, %1, off" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 15u;
}

// Sum (MAX = false) or maximum of a value over the live lanes of a wavefront, for lane 0: the 16 lanes of a row by four DPP steps (quad swaps, then the
// two mirrors -- lanes that already agree pairwise need no true butterfly), the rows by v_readlane.  About a dozen VALU instructions; a butterfly of
// ds_bpermute (__shfl_xor) is six dependent trips through the LDS pipeline, ~0.5 us on a wave's critical path (measured: the scalar log of a capture
// cost 14 % of a step with it).  NT: threads of the workgroup (the smallest plans run less than a wavefront).
template <int CTRL> __device__ __forceinline__ float dpp_move(float x) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true)); }
template <int CTRL> __device__ __forceinline__ double dpp_move(double x) {
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, 0xF, 0xF, true), hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xF, 0xF, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ float lane_value(float x, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), lane)); }
__device__ __forceinline__ double lane_value(double x, int lane) {
    const long long b = __double_as_longlong(x);
    return __longlong_as_double(((long long)__builtin_amdgcn_readlane((int)(b >> 32), lane) << 32) | (unsigned)__builtin_amdgcn_readlane((int)(b & 0xffffffffll), lane));
}
template <int NT, bool MAX, typename T> __device__ __forceinline__ double wave_total(T x) {
    auto op = [](T p, T q) -> T { if constexpr (MAX) return q > p ? q : p; else return p + q; };
    x = op(x, dpp_move<0xB1>(x));            // quad_perm [1,0,3,2]
    x = op(x, dpp_move<0x4E>(x));            // quad_perm [2,3,0,1]
    x = op(x, dpp_move<0x141>(x));           // row_half_mirror
    x = op(x, dpp_move<0x140>(x));           // row_mirror: all 16 lanes of a row hold the row's total
    constexpr int ROWS16 = NT < 16 ? 1 : (NT < 64 ? NT : 64) / 16;      // (lanes beyond the workgroup's threads are not live: a DPP read of them gives 0, which neither a sum nor a maximum of |A|^2 minds)
    T r = lane_value(x, 0);
#pragma unroll
    for (int i = 1; i < ROWS16; ++i) r = op(r, lane_value(x, 16 * i));
    return (double)r;
}

// maximum of a 32-bit unsigned value over the 64 lanes of a wavefront, for every lane (see wave_total)
__device__ __forceinline__ unsigned wave_max_u32(unsigned x) {
    unsigned o;
    o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xF, 0xF, true); x = o > x ? o : x;
    o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xF, 0xF, true); x = o > x ? o : x;
    o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x141, 0xF, 0xF, true); x = o > x ? o : x;
    o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x140, 0xF, 0xF, true); x = o > x ? o : x;
    unsigned r = (unsigned)__builtin_amdgcn_readlane((int)x, 0);
#pragma unroll
    for (int i = 1; i < 4; ++i) { const unsigned q = (unsigned)__builtin_amdgcn_readlane((int)x, 16 * i); r = q > r ? q : r; }
    return r;
}

// The body of k_time for workgroup `bid` of `nblk`.  PK: called from the persistent kernel of the medium plans (k_medium), where
// the field between the passes was stored by other workgroups of the SAME launch: it is read with sc1 loads (the stores are
// write-through already), MI355X_MICROARCH.md "Valid forms".
// LOG: the instantiation that keeps the scalar log of a capture run (TimeArgs::scal).  A template parameter rather than a test of the pointer: the kernels of
// every other run are then the very code they were without it -- the same registers, the same contractions, the same bits (a run-time branch in the column
// kernels changed which products the compiler fuses, and the adaptive step's z log with them, in the last bit).
// H (round 6): the FIELD between the passes is stored as complex64 while everything this kernel computes stays in T = double -- the line of a chirp-z run of a complex64
// caller (ssfm_chirp_line_run): a pass then moves 8 instead of 16 bytes per point and direction, and the result is rounded to complex64 once per PASS (four times a step,
// 6e-8 each, unbiased) instead of at every butterfly of a complex64 transform of 2^18 ... 2^22 points.  The buffers are the complex128 plan's own, half used.
template <typename T, int N1, int C, int E, int MODE, bool U16, bool PK = false, bool LOG = false, bool H = false>
__device__ __forceinline__ void time_body(const TimeArgs<T>& a, const unsigned bid, const unsigned nblk) {
    static_assert(!H || (sizeof(T) == 8 && !U16 && !PK && !LOG && (MODE == TM_BEGIN || MODE == TM_MID || MODE == TM_MID_L || MODE == TM_END)), "H: the plain layout of a complex128 plan, the line's four passes");
    if constexpr (H) {
        // (a pass queued behind the end of an adaptive run does nothing at all: round 5's left the caller's field alone but still transformed the line -- 100-120 us each
        // at 2^22 points -- and an adaptive run queues well past its end, its step only grows along the fibre)
        if (a.cz.done != nullptr && *a.cz.done) return;
    }
    constexpr int Q = N1 / E;                      // threads per column
    static_assert(!U16 || (sizeof(T) == 4 && C == 16 && Q % 4 == 0 && E % 2 == 0), "U16 layout: complex64, 16 columns, whole waves of 4 j");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cx<T>* lds = reinterpret_cast<cx<T>*>(smem_raw);

    constexpr bool INV = tm_inverse(MODE), FWD = tm_forward(MODE);
    static_assert(U16 || MODE <= TM_END || MODE == TM_MID_A || MODE == TM_MID_L, "the tile-private time-domain modes exist for the U16 layout only");
    static_assert(MODE != TM_MID_L || PK || !U16, "TM_MID_L: a pass of the one-launch engines, or a launch of its own on a plan in the plain layout");
    T hh_prev = a.hh_prev, hh_next = a.hh_next;
    const int tid = threadIdx.x;
    SSFM_TRACE_BEGIN(a);
    // plain: thread = j * C + c.  U16: lane = h * 32 + (j mod 4) * 8 + c8, column c = h * 8 + c8 (see "U16" above)
    const int c = U16 ? (((tid >> 5) & 1) << 3) | (tid & 7) : tid % C;
    const int j = U16 ? ((tid >> 6) << 2) | ((tid >> 3) & 3) : tid / C;
    const int ltid = U16 ? j * C + c : tid;       // thread index the tables are laid out by
    const long long N = (long long)N1 * a.N2;
    int tile, brow;
    xcd_unit_row(bid, a.N2 / C, a.rows, tile, brow);
    if (U16) tile = u16_tile_of_unit(tile, a.N2 / C);
    // natural (time-order) column of this thread: the tile's columns are contiguous in the row ORDER of the layout
    const int ncol = U16 ? (int)u16_col_of_pos((long long)tile * C + 2 * (c & 7) + (c >> 3), a.Qf) : tile * C + c;
    // wave-uniform bases (SGPRs) + one 32-bit lane offset
    cx<T>* __restrict__ Fb = a.F + (long long)brow * N;
    cx<T>* __restrict__ Yb = a.Y + (long long)brow * N;
    // |A|^2 is private to this kernel (written and read back by the same thread of the same tile),
    // so it is stored tile-major as 4 x (4 values per thread): 16-byte accesses, 1 KiB per wave.
    typedef T p4_t __attribute__((ext_vector_type(4)));
    p4_t* __restrict__ Pb = reinterpret_cast<p4_t*>(a.P + (long long)brow * N + (long long)tile * (N1 * C)) + tid;
    constexpr int PSTR = N1 * C / E;      // threads per tile
    // P16: the tile's 16 KiB hold 2 x 16 bytes of 16-bit values per thread (8 KiB), then one float scale per thread
    constexpr bool P16 = p16_layout<T, U16, E>() && !PK;
    // P32 (round 6): the large complex128 plans' stale |A|^2 crosses the launch boundary as 32-bit fixed point relative to the thread's own maximum -- E / 4 16-byte
    // units of 32-bit values per thread, then one float64 scale per thread: 40 instead of 64 bytes per thread and direction at E = 8.  |A|^2 only ever enters the
    // phase h/2 gamma |A|^2 (devices.py:1175,1181): its absolute error is <= 2^-33 of the thread's largest phase (6e-12 rad at 0.05 rad per half step; the
    // configuration C1 fixture sits 2.8e-13 from the float64 restatement with it, 3.0e-14 without, bound 1e-10).  A/B over three interleaved rounds
    // (profiles/r06_c1_ab.txt): C1 38.6-38.9 -> 37.5-37.6 us per step.  Plans of N1 >= 256 column points (2^16 samples and more), where the bytes are.
    constexpr bool P32 = sizeof(T) == 8 && N1 >= 256 && E % 4 == 0 && !PK && !U16;
    u32x4_t* __restrict__ Pq32 = reinterpret_cast<u32x4_t*>(a.P + (long long)brow * N + (long long)tile * (N1 * C)) + tid;
    double* __restrict__ Ps32 = reinterpret_cast<double*>(a.P + (long long)brow * N + (long long)tile * (N1 * C)) + (N1 * C / 2) + tid;
    u32x4_t* __restrict__ Pq = reinterpret_cast<u32x4_t*>(a.P + (long long)brow * N + (long long)tile * (N1 * C)) + tid;
    float* __restrict__ Ps = reinterpret_cast<float*>(a.P + (long long)brow * N + (long long)tile * (N1 * C)) + (N1 * C / 2) + tid;
    const int off = j * a.N2 + ncol;             // time-order side: element (n1 = j + Q t, n2 = ncol)
    const int stride = Q * a.N2;
    // half-transformed side, U16: the 16-byte unit of row j + Q (2g + h) at columns (c8, h = 0 | 1) of the tile
    typedef T u4_t __attribute__((ext_vector_type(4)));
    const int offy = U16 ? (j + Q * (c >> 3)) * a.N2 + tile * C + 2 * (c & 7) : off;
    using CI = ColIdx<C, U16 ? ilog2(fft_radix(N1, 0, E)) : 0>;
    const CI idx{c};

    // issue every global load of the tile up front
    cx<T> v[E];
    cx<T> w[E];
    T pold[E];
    LineTw<T, N1, E> tw;
    constexpr bool TWC = twn_compute<T, U16>();
    constexpr int NT = N1 * C / E;
    constexpr bool HEAD = MODE != TM_UNPACK;
    constexpr int NBS = (E * C + NT - 1) / NT;     // loads per thread of the tile's E x C inter-pass factors
    TwStaged<T, N1, E, NT> tws;
    cx<T> bs_pre[NBS];
    cx<T> wA = mk<T>((T)1, (T)0);
    if constexpr (HEAD) {
        // the small tables first (they come back first): the tile's inter-pass factors and the LDS-staged stage twiddles
        line_twiddles_prefetch<T, N1, E, NT>(tws, a.tw1, tid);
        if constexpr (TWC) {
#pragma unroll
            for (int i = 0; i < NBS; ++i) {
                const int e = tid + i * NT;
                bs_pre[i] = (SSFM_ABL_NO_TWN || e >= E * C) ? mk<T>((T)1, (T)0) : a.twB[(long long)tile * (E * C) + e];
            }
            if (!SSFM_ABL_NO_TWN) wA = a.twA[(long long)tile * (Q * C) + ltid];
        }
    }
    if (U16 && MODE != TM_BEGIN) {               // half-transformed field, or the tile-private time-domain field: 16-byte units
        if constexpr (PK && sizeof(T) == 4) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)Yb, 0, (int)(N * (long long)sizeof(cx<T>)), 0x00020000);
#pragma unroll
            for (int g = 0; g < E / 2; ++g) {
                const u32x4_t q = load16_sc1(rs, (offy + 2 * g * stride) * (int)sizeof(cx<T>));
                v[2 * g] = mk<T>(__uint_as_float(q.x), __uint_as_float(q.y));
                v[2 * g + 1] = mk<T>(__uint_as_float(q.z), __uint_as_float(q.w));
            }
        } else {
#pragma unroll
        for (int g = 0; g < E / 2; ++g) {
            const u4_t q = stream_load<true>(reinterpret_cast<const u4_t*>(&Yb[offy + 2 * g * stride]));
            v[2 * g] = mk<T>(q.x, q.y);
            v[2 * g + 1] = mk<T>(q.z, q.w);
        }
        }
    } else {
        const cx<T>* __restrict__ src = MODE == TM_BEGIN ? Fb : Yb;
        bool loaded = false;
        if constexpr (MODE == TM_BEGIN && !U16 && sizeof(T) == 8) {
            if (a.cz.chirp != nullptr) {
                if (a.cz.done != nullptr && *a.cz.done) return;
                const T hh = a.cz.h != nullptr ? (T)(0.5 * *a.cz.h) : a.cz.hh;
                const cx<T>* __restrict__ Ab = a.cz.A + (long long)brow * a.cz.n;
                T* __restrict__ Pn = a.cz.P + (long long)brow * a.cz.n;
#pragma unroll
                for (int t = 0; t < E; ++t) {
                    const int m = off + t * stride;
                    v[t] = mk<T>((T)0, (T)0);
                    if (m < a.cz.n) {
                        cx<T> x = Ab[m];
                        const T p = x.x * x.x + x.y * x.y;
                        Pn[m] = p;
                        if (a.cz.gamma != (T)0) {
                            T sn, cs;
                            sincos(a.cz.gamma * p * hh, &sn, &cs);
                            x = mk<T>(x.x * cs - x.y * sn, x.x * sn + x.y * cs);
                        }
                        v[t] = cmul(x, a.cz.chirp[m]);
                    }
                }
                loaded = true;
            }
        }
        if (!loaded) {
        if constexpr (H) {
            const cx<float>* __restrict__ s32 = reinterpret_cast<const cx<float>*>(MODE == TM_BEGIN ? a.F : a.Y) + (long long)brow * N;
#pragma unroll
            for (int t = 0; t < E; ++t) { const cx<float> q = s32[off + t * stride]; v[t] = mk<T>((T)q.x, (T)q.y); }
        } else {
#pragma unroll
        for (int t = 0; t < E; ++t) v[t] = src[off + t * stride];
        }
        }
    }
    cx<T>* Bs = lds + (fft_nstages(N1, E) > 1 ? N1 * C : 0);
    if constexpr (MODE == TM_UNPACK) {
        // nothing but the field moves
    } else if constexpr (TWC) {
        // W_N^((j + t Q) n2) = W_N^(j n2) * W_N^(Q1 t n2): the second factor depends on (t, column) only, the tile's E x C values go through LDS once
        // (asked for at the head, above); 1 MiB of tables instead of an 8 MiB stream per row, both stored tile by tile in the order they are read
#pragma unroll
        for (int i = 0; i < NBS; ++i) {
            const int e = tid + i * NT;
            if (e < E * C) Bs[e] = bs_pre[i];
        }
    } else {
        typedef T w4_t __attribute__((ext_vector_type(4)));
        const w4_t* __restrict__ W4 = reinterpret_cast<const w4_t*>(a.twN) + (long long)tile * (E / 2) * (N1 * C / E) + ltid;
#pragma unroll
        for (int g = 0; g < E / 2; ++g) {
            w4_t q;
            if (SSFM_ABL_NO_TWN) { q.x = (T)1; q.y = (T)0; q.z = (T)1; q.w = (T)0; } else q = W4[g * (N1 * C / E)];
            w[2 * g] = mk<T>(q.x, q.y);
            w[2 * g + 1] = mk<T>(q.z, q.w);
        }
    }
    // |A|^2 of the step's start is only needed after the inverse transform: its loads are issued after the
    // transform's first stage instead of with the field loads -- a smaller burst at the head of the kernel, which the tail of the
    // other lane's kernel queues behind (launch timeline: the last workgroup of a launch ends 2-3 us after the first).  Six
    // interleaved rounds, 2^20 x 2: 16.97 against 17.52 us per step (profiles/r03_ablation_and_knobs.txt)
    auto load_pold = [&]() {
        if constexpr (MODE == TM_MID_L) return;
        if constexpr (PK) {
#pragma unroll
            for (int t = 0; t < E; ++t) pold[t] = a.pkeep[t];
            return;
        }
        if constexpr (P32) {
            const double sc = (double)*Ps32 * (1.0 / 4294967295.0);
#pragma unroll
            for (int g = 0; g < E / 4; ++g) {
                const u32x4_t q = Pq32[g * PSTR];
                pold[4 * g] = (T)((double)q.x * sc); pold[4 * g + 1] = (T)((double)q.y * sc); pold[4 * g + 2] = (T)((double)q.z * sc); pold[4 * g + 3] = (T)((double)q.w * sc);
            }
            return;
        }
        if constexpr (P16) {
            // 2^23 + q is the float whose low mantissa bits are q: (2^23 + q) sc - 2^23 sc = q sc, one byte permute and one fma per value
            const float sc = stream_load<true>(Ps) * (1.0f / 65535.0f), off23 = -8388608.0f * sc;
#pragma unroll
            for (int g = 0; g < E / 8; ++g) {
                const u32x4_t q = stream_load<true>(&Pq[g * PSTR]);
                const unsigned w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    pold[8 * g + 2 * i] = fmaf(__uint_as_float(__builtin_amdgcn_perm(0x4B000000u, w[i], 0x07060100u)), sc, off23);
                    pold[8 * g + 2 * i + 1] = fmaf(__uint_as_float(__builtin_amdgcn_perm(0x4B000000u, w[i], 0x07060302u)), sc, off23);
                }
            }
            return;
        }
#pragma unroll
        for (int g = 0; g < E / 4; ++g) {
            p4_t q;
            if (SSFM_ABL_NO_P) q = (T)1e-3; else q = stream_load<stream_policy<T, U16>()>(&Pb[g * PSTR]);
            pold[4 * g] = q.x; pold[4 * g + 1] = q.y; pold[4 * g + 2] = q.z; pold[4 * g + 3] = q.w;
        }
    };
    constexpr bool LATE_P = (U16 || sizeof(T) == 8) && INV;       // (every mode that reads |A|^2)
    if (INV && !LATE_P) load_pold();
    // The step control state (adaptive runs) is read HERE, after every global load of the tile has been issued: it was written
    // by the previous launch, so its load is a miss of ~2 us that would otherwise stand in front of the whole kernel.  (A launch
    // queued beyond the end of the run returns now; its loads went to registers only.)
    StepState<T> S_this = {};            // TM_MID_A: the state of the step this launch finishes
    if constexpr (MODE == TM_MID_A && PK) {
        S_this = *a.s_in;
        hh_prev = S_this.h * (T)0.5;
    } else if constexpr (MODE == TM_MID_A) {
        S_this = a.st->cur[a.step & 1];
        // (a workgroup of an EARLIER launch gave up waiting: the run is void and the host will repeat it -- the launches still
        // queued return at once instead of each waiting out its own patience on a field that no longer means anything)
        if (a.st->error != 0u) return;
        if (S_this.done) {
            // a launch queued beyond the end of the run: hand the final state on (the host reads cur[] of the LAST launched step)
            if (bid == 0 && tid == 0) a.st->cur[(a.step + 1) & 1] = S_this;
            return;
        }
        hh_prev = S_this.h * (T)0.5;
    } else if (a.st != nullptr && MODE != TM_UNPACK) {
        if (a.st->error != 0u) return;                        // (see TM_MID_A above)
        const StepState<T> S = step_state<T>(a.st, a.step, FWD && a.derive != 0);
        if (FWD && bid == 0 && tid == 0) {
            // workgroup 0 records the state of this step (BEGIN is the first kernel of a step) and empties the slots its END fills
            if (a.derive) {
                a.st->cur[a.step & 1] = S;
                if (!a.st->cur[(a.step - 1) & 1].done) a.zlog[S.steps] = S.z;
            }
        }
        if (FWD && bid == 0 && tid < kAdaptSlots) a.st->slots[a.step & 1][tid] = 0ull;
        if (S.done) return;
        hh_prev = hh_next = S.h * (T)0.5;
    }
    cx<T>* ldsT = Bs + (TWC ? E * C : 0);
    if constexpr (MODE != TM_UNPACK) {
        line_twiddles_issue_regs<T, N1, E>(tw, j, a.tw1);
        line_twiddles_commit<T, N1, E, NT>(tws, ldsT, tid);
        if (fft_tw_lds_entries(N1, E) > 0 || TWC) __syncthreads();
        if constexpr (TWC) {
            w[0] = wA;
#pragma unroll
            for (int t = 1; t < E; ++t) w[t] = cmul(wA, Bs[t * C + c]);
            // (round 6 diagnostic, profiles/r06_fourstep_error.txt: these factors rounded ONCE from double instead of formed as a float32 product change the engines'
            // distance from the float64 solution by nothing -- 1.25 / 1.50 / 1.71e-5 -> 1.21 / 1.53 / 1.74e-5 at 2^12 / 2^14 / 2^16 after 83 steps)
        }
        line_twiddles_fetch<T, N1, E>(tw, j, ldsT);
    }
    if (MODE == TM_BEGIN_Y || MODE == TM_UNPACK) {
        // the exchange that undoes END_Y's: every thread gets its own column back
#pragma unroll
        for (int g = 0; g < E / 2; ++g) lane32_swap(v[2 * g], v[2 * g + 1]);
    }
    if (MODE == TM_UNPACK) {
#pragma unroll
        for (int t = 0; t < E; ++t) Fb[off + t * stride] = v[t];
        return;
    }
    if (INV) {
        if (U16) {
            // lane h = 0 keeps its low half and takes lane + 32's low half; lane h = 1 takes lane - 32's high half
#pragma unroll
            for (int g = 0; g < E / 2; ++g) {
                lane32_swap(v[2 * g], v[2 * g + 1]);
            }
        }
#pragma unroll
        for (int t = 0; t < E; ++t) v[t] = cmulc(v[t], w[t]);
        if constexpr (LATE_P) fft_line_hook<T, N1, E, +1, 0, CI>(v, lds, 0, j, idx, tw, load_pold);
        else if (!SSFM_ABL_NO_FFT) fft_line<T, N1, E, +1, 0, CI>(v, lds, 0, j, idx, tw);
    }
    if constexpr (((MODE == TM_MID || MODE == TM_MID_A) && PK) || (MODE == TM_MID && !U16)) {
        if (a.keep > 0) {                // (chirp-z: what the convolution left in the line's padding is not part of the field -- nor of its maximum)
#pragma unroll
            for (int t = 0; t < E; ++t)
                if (off + t * stride >= a.keep) v[t] = mk<T>((T)0, (T)0);
        }
    }
    // v = time-domain samples A(n1, n2).  Nonlinear operator (reference devices.py:1175-1181):
    // the second half step of the step being finished uses the |A|^2 of its START (pold), the
    // first half step of the next one the |A|^2 of the field after that rotation -- a rotation
    // does not change |A|, so both phases are known here and are applied as ONE rotation.
    T pmax = (T)0;
    T phi[E];
    T pnew[E];
    // z-resolved scalars (ssfm_propagate_fixed_capture): a wavefront's sum and maximum of |A|^2, taken where |A|^2 is in registers (below) and stored behind
    // the kernel's last transform, just ahead of the field stores and write-through like them.  Not earlier: the store is inline assembly, which the compiler's wait counters
    // do not see -- a later `s_waitcnt vmcnt` meant for this kernel's loads would wait for the store's trip to memory as well (measured: + 2 us per launch);
    // and not a plain store: a kernel that leaves any line dirty in its XCD's L2 pays ~0.9 us more at its end.
    double log_sum = 0.0, log_max = 0.0;
    auto log_store = [&]() {
        if constexpr (LOG) {
            if ((tid & 63) == 0) {
                constexpr int NW = (NT + 63) / 64;
                typedef double d2_t __attribute__((ext_vector_type(2)));
                d2_t q; q.x = log_sum; q.y = log_max;
                d2_t* dst = &reinterpret_cast<d2_t*>(a.scal)[((long long)brow * (a.N2 / C) + tile) * NW + (tid >> 6)];
                asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(dst), "v"(q) : "memory");
            }
        }
    };
    bool fwd_active = true;               // TM_MID_A: false when this step ends the run
    auto store_pnew = [&]() {
        if constexpr (PK) {
            if (FWD) {
#pragma unroll
                for (int t = 0; t < E; ++t) a.pkeep[t] = pnew[t];
            }
        } else
        if constexpr (P32) {
            if (FWD) {
                double m = (double)pnew[0];
#pragma unroll
                for (int t = 1; t < E; ++t) m = fmax(m, (double)pnew[t]);
                const double inv = m > 0.0 ? 4294967295.0 / m : 0.0;
                *Ps32 = m;
#pragma unroll
                for (int g = 0; g < E / 4; ++g) {
                    u32x4_t q;
                    q.x = __double2uint_rn(fmin((double)pnew[4 * g] * inv, 4294967295.0)); q.y = __double2uint_rn(fmin((double)pnew[4 * g + 1] * inv, 4294967295.0));
                    q.z = __double2uint_rn(fmin((double)pnew[4 * g + 2] * inv, 4294967295.0)); q.w = __double2uint_rn(fmin((double)pnew[4 * g + 3] * inv, 4294967295.0));
                    Pq32[g * PSTR] = q;
                }
            }
        } else
        if constexpr (P16) {
            if (FWD) {
                float m = pnew[0];
#pragma unroll
                for (int t = 1; t < E; ++t) m = fmaxf(m, pnew[t]);
                const float inv = m > 0.0f ? __builtin_amdgcn_rcpf(m) : 0.0f;          // (v_cvt_pknorm clamps to [0, 1]: the 1-ulp reciprocal cannot overflow)
                *Ps = m;                                                               // 256 bytes per wave; a plain store (a 4-byte write-through store is 6x the time per byte)
#pragma unroll
                for (int g = 0; g < E / 8; ++g) {
                    u32x4_t q;
                    unsigned w[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const auto h2 = __builtin_amdgcn_cvt_pknorm_u16(pnew[8 * g + 2 * i] * inv, pnew[8 * g + 2 * i + 1] * inv);
                        w[i] = __builtin_bit_cast(unsigned, h2);
                    }
                    q.x = w[0]; q.y = w[1]; q.z = w[2]; q.w = w[3];
                    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(&Pq[g * PSTR]), "v"(q) : "memory");
                }
            }
        } else
        if (FWD && !SSFM_ABL_NO_P) {
#pragma unroll
            for (int g = 0; g < E / 4; ++g) {
                p4_t q;
                q.x = pnew[4 * g]; q.y = pnew[4 * g + 1]; q.z = pnew[4 * g + 2]; q.w = pnew[4 * g + 3];
                if constexpr (sizeof(T) == 4) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(&Pb[g * PSTR]), "v"(q) : "memory");
                else Pb[g * PSTR] = q;
            }
        }
    };
    if constexpr (MODE == TM_MID_A) {
#pragma unroll
        for (int t = 0; t < E; ++t) {
            pnew[t] = v[t].x * v[t].x + v[t].y * v[t].y;
            pmax = pnew[t] > pmax ? pnew[t] : pmax;
        }
        // (Round 4 tried to do everything that does not need the next step size BEFORE the workgroups meet -- |A|^2 to the P buffer, the second half
        // rotation of the finished step, the other rotation after the meeting: 28.1-28.4 against 27.4-27.5 us per step, the second rotation costs
        // more than the wait hides; and to start the odd rows' workgroups of k_freq<FLY> 1.7 / 3.4 / 5.1 us late, so that the two rows of the one
        // launch are not in the same phase at the same time: 28.4 / 30.4 / 32.2 against 27.3; and to store |A|^2 alone before the meeting (4.7 of the
        // launch's 21 MB of writes under the wait): 28.5 against 27.3 -- the workgroup's word is queued behind those stores.  profiles/r04_adaptive.txt)
        // ---- the step control, inside the launch: every workgroup delivers its maximum, waits until all have, and replays
        // step_advance() on the same 64 slots (the same float operations: the same bits in every workgroup)
        // (the wavefront's maximum by DPP rows and v_readlane, not a ds_bpermute butterfly: six dependent trips through the LDS pipe on the path to the
        // meeting -- with the one behind the poll below 22.4 -> 21.x us per step at 2^20 x 2, profiles/r05_adaptive.txt)
        pmax = (T)wave_total<64, true>(pmax);
        // (every static a multiple of 16 bytes: they precede the dynamic LDS region, whose base must stay 16-byte aligned)
        __shared__ __attribute__((aligned(16))) T wave_max_a[16];
        __shared__ __attribute__((aligned(16))) StepState<T> s_next_[2];
        __shared__ __attribute__((aligned(16))) int s_ok_[4];
        StepState<T>& s_next = s_next_[0];
        int& s_ok = s_ok_[0];
        constexpr int NWAVES_A = (N1 * C / E + 63) / 64;
        if ((tid & 63) == 0) wave_max_a[tid >> 6] = pmax;
        __syncthreads();
        if (sizeof(T) == 4) {
            // ---- large grids: every workgroup publishes ONE word, a wavefront of every workgroup reads them all
            if (tid < 64) {
                const int set = a.step & 1;
                const unsigned long long epoch = (unsigned long long)(unsigned)(a.step + 1) << 32;
                const unsigned total = nblk, mine = bid;
                if (tid == 0) {
                    T m = wave_max_a[0];
#pragma unroll
                    for (int w = 1; w < NWAVES_A; ++w) m = wave_max_a[w] > m ? wave_max_a[w] : m;
                    if constexpr (PK) st_l2_u64(&a.st->wgmax[set][mine], epoch | (float_bits<T>(m) & 0xffffffffull));
                    else __hip_atomic_store(&a.st->wgmax[set][mine], epoch | (float_bits<T>(m) & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                const unsigned long long* words = a.st->wgmax[set];
                const long long t0 = wall_clock64();
                int good = 0;
                unsigned long long mb = 0ull;
                // a lane asks again only for the words it has not seen complete (512 workgroups x 512 words per round would be 2 MiB of
                // 8-byte uncached reads per round: the first form of this loop made the hand-over 6 us long)
                unsigned pending = 0u;
                unsigned rounds_ = a.st->patience < 0 ? 15u : 0u;          // (no patience at all -- the give-up test's setting: the first incomplete round ends the wait)
#pragma unroll
                for (int i = 0; i < kAdaptWords / 64; ++i)
                    if ((unsigned)tid + 64u * (unsigned)i < total) pending |= 1u << i;
                for (;;) {
#pragma unroll
                    for (int i = 0; i < kAdaptWords / 64; ++i) {
                        if (pending & (1u << i)) {
                            unsigned long long wd;
                            if constexpr (PK) wd = ld_l2_u64(&words[(unsigned)tid + 64u * (unsigned)i]);
                            else wd = __hip_atomic_load(&words[(unsigned)tid + 64u * (unsigned)i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if ((wd >> 32) == (epoch >> 32)) {
                                pending &= ~(1u << i);
                                const unsigned long long vb = wd & 0xffffffffull;
                                mb = vb > mb ? vb : mb;
                            }
                        }
                    }
                    if (__all(pending == 0u)) { good = 1; break; }
                    // (the clock is looked at every 16th round only -- s_memrealtime is a trip through the scalar memory path -- and the rounds are 64 instead of
                    // 256 cycles apart: 22.95 -> 22.43 us per step of the 652-step run at 2^20 x 2, profiles/r05_adaptive.txt)
                    if ((++rounds_ & 15u) == 0u && wall_clock64() - t0 > a.st->patience) break;              // the grid is not running as a whole -- give up, never hang
                    __builtin_amdgcn_s_sleep(1);
                }
                if (good) mb = (unsigned long long)wave_max_u32((unsigned)mb);          // (the words' low halves: the bit patterns of non-negative floats order like the floats)
                if (tid == 0) {
                    s_ok = good;
                    if (good) s_next = step_advance<T>(a.st, S_this, mb);
                    else atomicExch(&a.st->error, 1u);
                }
            }
        } else
        if (tid < 64) {
            const int set = a.step & 1;
            int good = 0;
            if (tid == 0) {
                T m = wave_max_a[0];
#pragma unroll
                for (int w = 1; w < NWAVES_A; ++w) m = wave_max_a[w] > m ? wave_max_a[w] : m;
                atomicMax(&a.st->slots[set][bid % kAdaptSlots], float_bits<T>(m));
                // Only atomics cross this barrier (the slots, read back with agent-scope loads below), so no fence is needed -- an agent-scope
                // release / acquire writes back / invalidates the XCD's whole L2 on gfx950 -- just the order: the maximum is acknowledged
                // before the arrival is counted.
                asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
                __hip_atomic_fetch_add(&a.st->arrive[set], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const long long t0 = wall_clock64();                  // 100 MHz
                for (;;) {
                    if (__hip_atomic_load(&a.st->arrive[set], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= nblk) { good = 1; break; }
                    if (wall_clock64() - t0 > a.st->patience) break;  // (20 ms) the grid is not running as a whole -- give up, never hang
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            good = __shfl(good, 0);
            unsigned long long mb = 0ull;
            if (good) {
                mb = __hip_atomic_load(&a.st->slots[set][tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    const unsigned long long other = __shfl_xor(mb, o);
                    mb = other > mb ? other : mb;
                }
            }
            if (tid == 0) {
                s_ok = good;
                if (good) s_next = step_advance<T>(a.st, S_this, mb);
                else atomicExch(&a.st->error, 1u);
            }
        }
        __syncthreads();
        if (!s_ok) return;
        const StepState<T> Sn = s_next;
        if constexpr (PK) { if (tid == 0) *a.s_out = Sn; }
        if (bid == 0) {
            // workgroup 0 records the state of the next step and empties what the next step's launch will fill
            if (tid == 0) { a.st->cur[(a.step + 1) & 1] = Sn; a.zlog[Sn.steps] = Sn.z; a.st->arrive[(a.step + 1) & 1] = 0u; }
            if (tid < kAdaptSlots) a.st->slots[(a.step + 1) & 1][tid] = 0ull;
        }
        fwd_active = !Sn.done;
        hh_next = Sn.h * (T)0.5;
#pragma unroll
        for (int t = 0; t < E; ++t) {
            T ph = hh_prev * (a.gamma * pold[t]);
            if (fwd_active) ph += hh_next * (a.gamma * pnew[t]);
            phi[t] = ph;
        }
    } else if constexpr (MODE != TM_MID_L) {
#pragma unroll
    for (int t = 0; t < E; ++t) {
        const T p = v[t].x * v[t].x + v[t].y * v[t].y;
        T ph = (T)0;
        if (INV) ph = hh_prev * (a.gamma * pold[t]);
        if (FWD) ph += hh_next * (a.gamma * p);
        pnew[t] = p;
        phi[t] = ph;
        pmax = p > pmax ? p : pmax;
    }
    if constexpr (LOG) {
        // z-resolved scalars (ssfm_propagate_fixed_capture): the field's power and peak after the step this launch ends (BEGIN: of the input) -- |A|^2 is in
        // registers here anyway; a wavefront stores the sum and the maximum of its values as one 16-byte pair of its own (no atomics, nothing waits).
        {
            T sum = pnew[0];
#pragma unroll
            for (int t = 1; t < E; ++t) sum += pnew[t];
            log_sum = wave_total<NT, false>(sum);
            log_max = wave_total<NT, true>(pmax);
        }
    }
    }
    if constexpr (MODE != TM_MID_L) {
    store_pnew();
    if (!SSFM_ABL_NO_NL) rotate_all<E>(v, phi);
    else {
#pragma unroll
        for (int t = 0; t < E; ++t) v[t].x += phi[t];
    }
    }
    if constexpr (MODE == TM_MID_L) {
        if (a.s_in != nullptr) {
            // adaptive runs: `mul` holds D~ (natural order, `keep` entries), the factor exp(D~ h) * hh_next is formed here with the step size of the
            // state in LDS -- the products in T as the reference forms them, the functions as k_freq<FM_FLY> takes them
            const T h = a.s_in->h;
#pragma unroll
            for (int t = 0; t < E; ++t) {
                const int m = off + t * stride;
                if (m < a.keep) {
                    const cx<T> d = a.mul[m];
                    const T e = exp_acc<T>(d.x * h) * a.hh_next;
                    T sn, cs;
                    sincos_acc<T>(d.y * h, sn, cs);
                    v[t] = cmul(v[t], mk<T>(e * cs, e * sn));
                } else v[t] = mk<T>((T)0, (T)0);
            }
        } else
        if (a.mul != nullptr) {
            if (a.keep > 0) {
                // (a launch of its own on a long line, ssfm_chirp_line_run: the table is zero from `keep` up -- three quarters of a line just above a power of
                // two -- and reading the zeros doubled what this pass reads: 150 us per 2^22-point row against the 78 of the pass that carries the rotation)
#pragma unroll
                for (int t = 0; t < E; ++t) {
                    const int m = off + t * stride;
                    v[t] = m < a.keep ? cmul(v[t], a.mul[m]) : mk<T>((T)0, (T)0);
                }
            } else {
#pragma unroll
                for (int t = 0; t < E; ++t) v[t] = cmul(v[t], a.mul[off + t * stride]);
            }
        }
    }
    if constexpr (MODE == TM_MID && !U16) {
        if (a.mul != nullptr) {
            if (a.keep > 0) {            // (the table has `keep` entries; the samples from there on were set to zero above)
#pragma unroll
                for (int t = 0; t < E; ++t) {
                    const int m = off + t * stride;
                    if (m < a.keep) v[t] = cmul(v[t], a.mul[m]);
                }
            } else {
#pragma unroll
            for (int t = 0; t < E; ++t) v[t] = cmul(v[t], a.mul[off + t * stride]);
            }
        }
    }
    if constexpr (MODE == TM_MID_A) {
        if (!fwd_active) {
            // the run ends with this step: time-order field, as k_time<TM_END> leaves it
#pragma unroll
            for (int t = 0; t < E; ++t) Fb[off + t * stride] = v[t];
            return;
        }
    }
    if (tm_ends(MODE)) {
        if (MODE == TM_END) {
            if constexpr (MODE == TM_END && !U16 && sizeof(T) == 8) {
                if (a.cz.chirp != nullptr) {
                    if (a.cz.done != nullptr && *a.cz.done) return;
                    const T hh = a.cz.h != nullptr ? (T)(0.5 * *a.cz.h) : a.cz.hh;
                    cx<T>* __restrict__ Ab = a.cz.A + (long long)brow * a.cz.n;
                    const T* __restrict__ Pn = a.cz.P + (long long)brow * a.cz.n;
                    T pmx = (T)0;
#pragma unroll
                    for (int t = 0; t < E; ++t) {
                        const int m = off + t * stride;
                        if (m < a.cz.n) {
                            const cx<T> w = a.cz.chirp[m];
                            cx<T> x = mk<T>((v[t].x * w.x + v[t].y * w.y) * a.cz.scale, (v[t].y * w.x - v[t].x * w.y) * a.cz.scale);     // z conj(c) / n
                            if (a.cz.gamma != (T)0) {
                                T sn, cs;
                                sincos(a.cz.gamma * Pn[m] * hh, &sn, &cs);
                                x = mk<T>(x.x * cs - x.y * sn, x.x * sn + x.y * cs);
                            }
                            Ab[m] = x;
                            const T p = x.x * x.x + x.y * x.y;
                            pmx = p > pmx ? p : pmx;
                        }
                    }
                    if (a.cz.maxbits != nullptr) {
                        for (int o = 32; o > 0; o >>= 1) {
                            const T other = __shfl_xor(pmx, o);
                            pmx = other > pmx ? other : pmx;
                        }
                        if ((threadIdx.x & 63) == 0 && pmx > (T)0) atomicMax(a.cz.maxbits, (unsigned long long)__double_as_longlong((double)pmx));
                    }
                    return;
                }
            }
            // (time-order output: plain stores -- it is read by whatever comes after the run, not by the next pass)
            if constexpr (H) {
                cx<float>* __restrict__ F32 = reinterpret_cast<cx<float>*>(a.F) + (long long)brow * N;
#pragma unroll
                for (int t = 0; t < E; ++t) F32[off + t * stride] = mk<float>((float)v[t].x, (float)v[t].y);
            } else {
#pragma unroll
            for (int t = 0; t < E; ++t) Fb[off + t * stride] = v[t];
            }
        } else {
            // tile-private 16-byte units in the Y buffer: only this tile's BEGIN_Y (or UNPACK) reads them back
#pragma unroll
            for (int g = 0; g < E / 2; ++g) {
                lane32_swap(v[2 * g], v[2 * g + 1]);
                u4_t q;
                q.x = v[2 * g].x; q.y = v[2 * g].y; q.z = v[2 * g + 1].x; q.w = v[2 * g + 1].y;
                pass_store<PK>(reinterpret_cast<u4_t*>(&Yb[offy + 2 * g * stride]), q);
            }
        }
        if (a.st != nullptr) {
            // max |A|^2 of the workgroup -> its slot; the next BEGIN reads the slots (see AdaptState)
            pmax = (T)wave_total<N1 * C / E, true>(pmax);
            __shared__ T wave_max[16];
            constexpr int NWAVES = (N1 * C / E + 63) / 64;
            if ((tid & 63) == 0) wave_max[tid >> 6] = pmax;
            __syncthreads();
            if (tid == 0) {
                T m = wave_max[0];
#pragma unroll
                for (int w = 1; w < NWAVES; ++w) m = wave_max[w] > m ? wave_max[w] : m;
                atomicMax(&a.st->slots[a.step & 1][bid % kAdaptSlots], float_bits<T>(m));
            }
        }
        log_store();
        SSFM_TRACE_END(a);
        return;
    }
    // (one LDS buffer: a transform that follows another starts with a barrier before its first exchange, XP = 1)
    constexpr int NX = fft_nstages(N1, E) - 1;      // exchanges of the inverse transform
    constexpr int XP_FWD = ((MODE != TM_MID && MODE != TM_MID_A && MODE != TM_MID_L) || NX == 0) ? 0 : 1;
    if (!SSFM_ABL_NO_FFT) fft_line<T, N1, E, -1, XP_FWD, CI>(v, lds, 0, j, idx, tw);
    log_store();                 // (behind the kernel's last load, ahead of its field stores: see log_store)
    if (U16) {
#pragma unroll
        for (int t = 0; t < E; ++t) v[t] = cmul(v[t], w[t]);
#pragma unroll
        for (int g = 0; g < E / 2; ++g) {
            lane32_swap(v[2 * g], v[2 * g + 1]);           // (the exchange is its own inverse)
            u4_t q;
            q.x = v[2 * g].x; q.y = v[2 * g].y; q.z = v[2 * g + 1].x; q.w = v[2 * g + 1].y;
            pass_store<PK>(reinterpret_cast<u4_t*>(&Yb[offy + 2 * g * stride]), q);
        }
    } else if constexpr (H) {
        cx<float>* __restrict__ Y32 = reinterpret_cast<cx<float>*>(a.Y) + (long long)brow * N;
#pragma unroll
        for (int t = 0; t < E; ++t) {
            const cx<T> y = cmul(v[t], w[t]);
            pass_store<PK>(&Y32[off + t * stride], mk<float>((float)y.x, (float)y.y));
        }
    } else {
#pragma unroll
        for (int t = 0; t < E; ++t) pass_store<PK>(&Yb[off + t * stride], cmul(v[t], w[t]));
    }
#if SSFM_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    SSFM_TRACE_END(a);
}

// Kernel arguments.  A struct passed by value lives in the kernarg segment and is fetched by scalar loads at the head of the kernel: a miss of
// its own (the segment was written by the host for this very launch) in front of everything else.  gfx950 can PRELOAD the first 16 dwords of
// the segment into SGPRs while the wave is launched (-mllvm -amdgpu-kernarg-preload-count=16; .amdhsa_user_sgpr_kernarg_preload_length) -- for
// scalar and pointer arguments, not for aggregates, 14 dwords at most (16 user SGPRs less the segment pointer).  So the two kernels of a step take
// what their heads need as leading scalars and the rest -- the fields of the other modes -- as a trailing struct that is fetched where it is used.  (k_medium and the
// single-launch engines call time_body / freq_body themselves and keep their structs.)
// The scalar log of a capture: `per` (sum, max) pairs per (step, row), as the column kernels' wavefronts left them -> out[(step, row)] = (mean |A|^2, max |A|^2).
// One wavefront per (step, row), a fixed order of additions: the log is reproducible bit for bit.
template <int UNUSED = 0> __global__ __launch_bounds__(64) void k_scal_reduce(const double* __restrict__ raw, double* __restrict__ out, int per, double inv_n) {      // (a template: one definition across the translation units)
    const long long item = blockIdx.x;
    const double* r = raw + item * (long long)per * 2;
    double sum = 0.0, mx = 0.0;
    for (int i = threadIdx.x; i < per; i += 64) {
        sum += r[2 * i];
        mx = r[2 * i + 1] > mx ? r[2 * i + 1] : mx;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        sum += __shfl_xor(sum, o);
        const double other = __shfl_xor(mx, o);
        mx = other > mx ? other : mx;
    }
    if (threadIdx.x == 0) { out[2 * item] = sum * inv_n; out[2 * item + 1] = mx; }
}

template <typename T> struct TimeArgsCold {
    cx<T>* F;
    const cx<T>* twN;
    AdaptState<T>* st;
    T* zlog;
    int step, derive;
    const StepState<T>* s_in;
    StepState<T>* s_out;
    const cx<T>* mul;
    T* pkeep;
    int keep;
    double* scal;
    ChirpIO<T> cz;
    SSFM_TRACE_ARGS
};
template <typename T> __host__ __device__ inline TimeArgsCold<T> time_args_cold(const TimeArgs<T>& a) {
    TimeArgsCold<T> c;
    c.keep = a.keep; c.scal = a.scal;
    c.F = a.F; c.twN = a.twN; c.st = a.st; c.zlog = a.zlog; c.step = a.step; c.derive = a.derive; c.s_in = a.s_in; c.s_out = a.s_out; c.mul = a.mul; c.pkeep = a.pkeep; c.cz = a.cz;
#if SSFM_TRACE
    c.trace = a.trace; c.trace_slot = a.trace_slot;
#endif
    return c;
}
template <typename T, int N1, int C, int E, int MODE, bool U16 = false, bool LOG = false>
__global__ SSFM_KERNEL_BOUNDS(N1 * C / E, sizeof(T), E) void k_time(cx<T>* Y, T* P, const cx<T>* twA, const cx<T>* twB, const cx<T>* tw1, int N2, int rows, int Qf,
                                                                     T gamma, T hh_prev, T hh_next, const TimeArgsCold<T> c) {
    // (14 dwords are preloaded: the five pointers, N2, rows, Qf and gamma -- what stands in front of the first load; the two half steps follow by scalar load)
    TimeArgs<T> a;
    a.Y = Y; a.P = P; a.twA = twA; a.twB = twB; a.tw1 = tw1; a.gamma = gamma; a.hh_prev = hh_prev; a.hh_next = hh_next; a.N2 = N2; a.rows = rows; a.Qf = Qf;
    a.F = c.F; a.twN = c.twN; a.st = c.st; a.zlog = c.zlog; a.step = c.step; a.derive = c.derive; a.s_in = c.s_in; a.s_out = c.s_out; a.mul = c.mul; a.pkeep = c.pkeep; a.cz = c.cz;
    a.keep = c.keep; a.scal = c.scal;
#if SSFM_TRACE
    a.trace = c.trace; a.trace_slot = c.trace_slot;
#endif
    time_body<T, N1, C, E, MODE, U16, false, LOG>(a, blockIdx.x, gridDim.x);
}
// ... and with the field between the passes in complex64 (time_body, H): complex128 plans, the passes of a chirp-z line
template <typename T, int N1, int C, int E, int MODE>
__global__ SSFM_KERNEL_BOUNDS(N1 * C / E, sizeof(T), E) void k_time_h(cx<T>* Y, T* P, const cx<T>* twA, const cx<T>* twB, const cx<T>* tw1, int N2, int rows, int Qf,
                                                                       T gamma, T hh_prev, T hh_next, const TimeArgsCold<T> c) {
    TimeArgs<T> a;
    a.Y = Y; a.P = P; a.twA = twA; a.twB = twB; a.tw1 = tw1; a.gamma = gamma; a.hh_prev = hh_prev; a.hh_next = hh_next; a.N2 = N2; a.rows = rows; a.Qf = Qf;
    a.F = c.F; a.twN = c.twN; a.st = c.st; a.zlog = c.zlog; a.step = c.step; a.derive = c.derive; a.s_in = c.s_in; a.s_out = c.s_out; a.mul = c.mul; a.pkeep = c.pkeep; a.cz = c.cz;
    a.keep = c.keep; a.scal = c.scal;
#if SSFM_TRACE
    a.trace = c.trace; a.trace_slot = c.trace_slot;
#endif
    time_body<T, N1, C, E, MODE, false, false, false, true>(a, blockIdx.x, gridDim.x);
}
// host side: the launch of k_time from a TimeArgs
#define SSFM_TIME_KERNEL_ARGS(a) (a).Y, (a).P, (a).twA, (a).twB, (a).tw1, (a).N2, (a).rows, (a).Qf, (a).gamma, (a).hh_prev, (a).hh_next, ssfm::time_args_cold(a)

// ------------------------------------------------------------------------------ k_freq
// FM_PHASE: the operator table of a FIBRE (|exp(D~ h)| is the same number at every frequency: Re D~ = -alpha/2,
// devices.py:1145) holds only the PHASE of every entry, as a 32-bit fraction of a turn (4 bytes per frequency instead of
// 8: the table is 16 of the 96 bytes a dual-pol sample*step moves); the kernel forms amp * (cos, sin) itself.
// (FM_FLY_IM: FM_FLY for an operator whose real part is the same number at every frequency -- a fibre's -alpha/2 -- with `tab` holding the IMAGINARY
// parts only, four per 16 bytes in FM_PHASE's order, and `amp` the real part: half the operator's bytes, the same arithmetic)
enum FreqMode { FM_TABLE = 0, FM_FLY = 1, FM_FWD_ONLY = 2, FM_PHASE = 3, FM_FLY_IM = 4, FM_INV_ONLY = 5 };
// (FM_FWD_ONLY: the plan's layout in, the forward row transform, the PLAIN transposed spectrum out; FM_INV_ONLY, round 6: the plain transposed spectrum in,
// the unnormalised inverse row transform, the plan's layout out -- the two halves of a row pass around k_split_mid, ssfm_split.hpp)

template <typename T> struct FreqArgs {
    cx<T>* F;
    const cx<T>* tab;        // FM_TABLE: exp(D~ h)/N (or H/N) at [k1*N2 + k2];  FM_FLY: D~ at the same place
    const cx<T>* tw2;        // W_N2^q
    const AdaptState<T>* st; // FM_FLY: step size source when non-null (the state of step `step`)
    T h;                     // FM_FLY with st == nullptr
    T amp;                   // FM_PHASE: exp(Re D~ h) / N, the modulus of every table entry
    int step;
    T inv_n;
    int N1;
    int rows;                // batch rows covered by this launch
    int u16;                 // field rows in the 16-byte-unit order (host side picks the kernel variant)
    SSFM_TRACE_ARGS
};

template <typename T> __device__ __forceinline__ T exp_acc(T x);
template <> __device__ __forceinline__ float exp_acc<float>(float x) { return expf(x); }
template <> __device__ __forceinline__ double exp_acc<double>(double x) { return exp(x); }

// ---- phase tables (FM_PHASE).  Entry = round(frac(phi / 2 pi) * 2^32), phi = the reference's float32 product Im(D~) * h
// (devices.py:1179) taken to double: resolution 1.5e-9 rad, finer than the float32 ulp of any phase above 0.013 rad.
// Row k1, element k2 = j + t Q  ->  ((t >> 2) Q + j) 4 + (t & 3): a thread's slots 4g .. 4g+3 are one 16-byte load.
__host__ __device__ __forceinline__ long long freq_phase_pos(long long k2, int Q) {
    const long long j = k2 % Q, t = k2 / Q;
    return ((t >> 2) * Q + j) * 4 + (t & 3);
}
// amp * exp(i 2 pi u / 2^32).  The turn fraction needs no range reduction: u << 1 read as a signed number IS the angle
// modulo pi in [-pi/2, pi/2) (in units of pi / 2^32), and bit 31 of u + 2^30 says whether the half turn taken off was odd
// (then both components change sign: folded into amp).  cos and sin run as the two halves of ONE packed polynomial in
// z = x^2 (expi_half_turn above).  About 15 instructions per frequency.
__device__ __forceinline__ cf32 phase32_factor(unsigned u, float amp) {
    const int r = (int)(u << 1);
    const unsigned flip = (u + 0x40000000u) & 0x80000000u;
    const float x = (float)r * 7.3145906e-10f;                   // pi / 2^32
    return expi_half_turn(x, __uint_as_float(__float_as_uint(amp) ^ flip));
}
__device__ __forceinline__ cf64 phase32_factor(unsigned, double) { return mk<double>(0.0, 0.0); }      // (complex64 plans only)
// complex128 plans (round 4): the phase as a float64 fraction of a turn in [-0.5, 0.5] -- 8 bytes per frequency instead of the 16 of the complex
// table, which is a third of what k_freq<double> moves (51 MB per launch, DESIGN.md section 5: C1 is bound by traffic).  amp * exp(2 pi i x):
// k = rint(4 x) quarter turns come off exactly (x - k / 4 has no rounding), the rest |theta| <= pi / 4 goes through the fdlibm kernels (< 1 ulp),
// the quarter turns are a swap / sign of the two components.  About 30 float64 instructions per frequency, formed under the field loads.
__device__ __forceinline__ cf64 phase64_factor(double x, double amp) {
    const double k = rint(4.0 * x);
    const double theta = fma(k, -0.25, x) * 6.283185307179586476925;
    double s, c;
    sincos_tiny(theta, s, c);
    const int q = (int)k & 3;
    const double cr = (q & 1) ? -s : c, sr = (q & 1) ? c : s;          // one quarter turn: (c, s) -> (-s, c)
    const double sg = (q & 2) ? -amp : amp;                            // two: both signs
    return mk<double>(sg * cr, sg * sr);
}
__device__ __forceinline__ cf32 phase64_factor(double, float) { return mk<float>(0.0f, 0.0f); }        // (complex128 plans only)

// m[t] <- exp(D~_t h) / N with D~_t = m[t] on entry and ph[t] = Im(D~_t) h; `flat`: Re D~ is the same at every t and
// e0 = exp(Re D~ h).  (e * c) * inv_n == (e * inv_n) * c exactly: N is a power of two.
template <int E> __device__ __forceinline__ void fly_factors(cf32 (&m)[E], const float (&ph)[E], bool flat, float e0, float h, float inv_n) {
    float amax = 0.0f;
#pragma unroll
    for (int t = 0; t < E; ++t) amax = fmaxf(amax, fabsf(ph[t]));
    if (__builtin_expect(amax <= kSincosSmallMax, 1)) {
#pragma unroll
        for (int t = 0; t < E; ++t) {
            const float e = flat ? e0 : exp_acc<float>(m[t].x * h);
            m[t] = expi_f32(ph[t], e * inv_n);
        }
    } else {
#pragma unroll
        for (int t = 0; t < E; ++t) {
            float sn, cs;
            sincos_f32<true>(ph[t], sn, cs);
            const float e = flat ? e0 : exp_acc<float>(m[t].x * h);
            m[t] = mk<float>((e * cs) * inv_n, (e * sn) * inv_n);
        }
    }
}
template <int E> __device__ __forceinline__ void fly_factors(cf64 (&m)[E], const double (&ph)[E], bool flat, double e0, double h, double inv_n) {
#pragma unroll
    for (int t = 0; t < E; ++t) {
        double sn, cs;
        sincos(ph[t], &sn, &cs);
        const double e = flat ? e0 : exp_acc<double>(m[t].x * h);
        m[t] = mk<double>((e * cs) * inv_n, (e * sn) * inv_n);
    }
}

// The body of k_freq for workgroup `bid` (PK: see time_body)
template <typename T, int N2, int ROWS, int E, int MODE, bool U16, bool PK = false, bool H = false>
__device__ __forceinline__ void freq_body(const FreqArgs<T>& a, const unsigned bid) {
    static_assert(!H || (sizeof(T) == 8 && !U16 && !PK && MODE == FM_TABLE), "H (see time_body): the plain layout of a complex128 plan, a table of complex numbers");
    constexpr int Q = N2 / E;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cx<T>* lds = reinterpret_cast<cx<T>*>(smem_raw);

    T h = a.h;
    constexpr bool FLY = MODE == FM_FLY || MODE == FM_FLY_IM;
    SSFM_TRACE_BEGIN(a);
    const int tid = threadIdx.x;
    const int j = tid % Q;
    const int rr = tid / Q;
    // block -> (group of ROWS consecutive k1, batch row); batch rows of one k1 group share an XCD
    int kgrp, brow;
    xcd_unit_row(bid, a.N1 / ROWS, a.rows, kgrp, brow);
    const int k1 = kgrp * ROWS + rr;
    const long long row = (long long)brow * a.N1 + k1;
    cx<T>* __restrict__ Frow = a.F + row * N2;
    const cx<T>* __restrict__ trow = a.tab + (long long)k1 * N2;
    using RI = RowIdx<row_pad_shift(E)>;
    const RI idx{rr * row_lds_elems(N2, E)};

    cx<T> v[E];
    cx<T> m[E];
    LineTw<T, N2, E> tw;
    SSFM_STAMP(0);
    typedef T u4_t __attribute__((ext_vector_type(4)));
    // the phase loads go out BEFORE the field loads (results return in issue order) and amp * exp(i phase) is formed while the
    // field is on its way -- 200 instructions per thread off the workgroup's critical path (the head of a kernel is a wait of 1-2 us)
    constexpr bool EARLY_PHASE = MODE == FM_PHASE;
    TwStaged<T, N2, E, ROWS * N2 / E> tws;
    line_twiddles_prefetch<T, N2, E, ROWS * N2 / E>(tws, a.tw2, tid);
    unsigned pu[E];
    double pt[sizeof(T) == 8 ? E : 1];          // complex128: the phases as float64 turn fractions, slots 2g and 2g+1 side by side (freq_tab_pos)
    auto load_phases = [&]() {
        if constexpr (sizeof(T) == 8) {
            typedef double d2_t __attribute__((ext_vector_type(2)));
            const d2_t* __restrict__ P2 = reinterpret_cast<const d2_t*>(reinterpret_cast<const double*>(a.tab) + (long long)k1 * N2) + j;
#pragma unroll
            for (int g = 0; g < E / 2; ++g) {
                const d2_t q = P2[g * Q];
                pt[2 * g] = q.x; pt[2 * g + 1] = q.y;
            }
        } else {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const u32x4* __restrict__ P4 = reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned*>(a.tab) + (long long)k1 * N2) + j;
#pragma unroll
        for (int g = 0; g < E / 4; ++g) {
            u32x4 q;
            if (SSFM_ABL_NO_TAB) q = 0x12345678u; else q = P4[g * Q];
            pu[4 * g] = q.x; pu[4 * g + 1] = q.y; pu[4 * g + 2] = q.z; pu[4 * g + 3] = q.w;
        }
        }
    };
    auto phase_factors = [&]() {
#pragma unroll
        for (int t = 0; t < E; ++t) {
            if constexpr (sizeof(T) == 8) m[t] = phase64_factor(pt[t], a.amp);
            else m[t] = phase32_factor(pu[t], a.amp);
        }
    };
    // (the same for the operator itself where the kernel forms exp(D~ h): adaptive runs.  Only with 16-byte elements... of either precision)
    constexpr bool EARLY_FLY = FLY;
    auto load_table = [&]() {
        if constexpr (MODE == FM_FLY_IM) {
            static_assert(MODE != FM_FLY_IM || E % 4 == 0, "four imaginary parts per load");
            typedef T i4_t __attribute__((ext_vector_type(4)));
            const i4_t* __restrict__ I4 = reinterpret_cast<const i4_t*>(reinterpret_cast<const T*>(a.tab) + (long long)k1 * N2) + j;
#pragma unroll
            for (int g = 0; g < E / 4; ++g) {
                const i4_t q = I4[g * Q];
                m[4 * g] = mk<T>(a.amp, q.x); m[4 * g + 1] = mk<T>(a.amp, q.y); m[4 * g + 2] = mk<T>(a.amp, q.z); m[4 * g + 3] = mk<T>(a.amp, q.w);
            }
            return;
        }
        typedef T m4_t __attribute__((ext_vector_type(4)));
        const m4_t* __restrict__ T4 = reinterpret_cast<const m4_t*>(trow) + j;
#pragma unroll
        for (int g = 0; g < E / 2; ++g) {
            m4_t q;
            if (SSFM_ABL_NO_TAB) { q.x = a.inv_n; q.y = (T)0; q.z = a.inv_n; q.w = (T)0; } else q = T4[g * Q];
            m[2 * g] = mk<T>(q.x, q.y);
            m[2 * g + 1] = mk<T>(q.z, q.w);
        }
    };
    if constexpr (EARLY_PHASE) load_phases();
    if constexpr (EARLY_FLY) load_table();
    constexpr bool INV_ONLY = MODE == FM_INV_ONLY;
    // (a.u16 == 2: the two halves of a split plan's row pass keep the UNIT layout between them -- 16-byte accesses on both sides of k_split_mid, which is pointwise and
    // only needs to know which frequency a position holds; a.u16 == 1: the plain transposed spectrum, what ssfm_debug's forward transform hands out)
    if (U16 && (!INV_ONLY || a.u16 == 2)) {
        // U16 layout: register slots 2g and 2g+1 (elements j + Q 2g, j + Q (2g+1)) lie side by side in the row
        if constexpr (PK && sizeof(T) == 4) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)Frow, 0, N2 * (int)sizeof(cx<T>), 0x00020000);
#pragma unroll
            for (int g = 0; g < E / 2; ++g) {
                const u32x4_t q = load16_sc1(rs, (g * Q + j) * 16);
                v[2 * g] = mk<T>(__uint_as_float(q.x), __uint_as_float(q.y));
                v[2 * g + 1] = mk<T>(__uint_as_float(q.z), __uint_as_float(q.w));
            }
        } else {
#pragma unroll
        for (int g = 0; g < E / 2; ++g) {
            const u4_t q = stream_load<true>(reinterpret_cast<const u4_t*>(Frow) + g * Q + j);
            v[2 * g] = mk<T>(q.x, q.y);
            v[2 * g + 1] = mk<T>(q.z, q.w);
        }
        }
    } else if constexpr (H) {
        const cx<float>* __restrict__ F32 = reinterpret_cast<const cx<float>*>(a.F) + row * N2;
#pragma unroll
        for (int t = 0; t < E; ++t) { const cx<float> q = F32[j + t * Q]; v[t] = mk<T>((T)q.x, (T)q.y); }
    } else {
#pragma unroll
        for (int t = 0; t < E; ++t) v[t] = Frow[j + t * Q];
    }
    cx<T>* ldsT = lds + (fft_nstages(N2, E) > 1 ? ROWS * row_lds_elems(N2, E) : 0);
    line_twiddles_issue_regs<T, N2, E>(tw, j, a.tw2);
    if constexpr (MODE == FM_TABLE) load_table();
    if ((FLY || MODE == FM_FWD_ONLY || MODE == FM_INV_ONLY) && a.st != nullptr) {
        // (read after the row and the operator have been asked for: the state was written by the previous launch, a ~2 us miss)
        // (the two halves of a split plan's row pass: launches queued behind the end of an adaptive run leave the field alone, like every other kernel)
        const StepState<T> S = a.st->cur[a.step & 1];
        if (S.done || a.st->error != 0u) return;             // (error: see k_time<TM_MID_A>)
        h = S.h;
    }
    SSFM_STAMP(1);
#if SSFM_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SSFM_STAMP(2);
#endif
    line_twiddles_commit<T, N2, E, ROWS * N2 / E>(tws, ldsT, tid);
    auto fly = [&]() {
        // exp(D~ h): real factor exp(Re*h), phase Im*h (reference devices.py:1179), then 1/N
        T ph[E];
#pragma unroll
        for (int t = 0; t < E; ++t) ph[t] = m[t].y * h;
        // Re D~ = -alpha/2 is the same number at every frequency for a fibre (devices.py:1145): then exp(Re h) is ONE
        // exponential per thread instead of 16 (bit-identical: the same float product, the same function); any other
        // operator takes the general path
        bool flat = true;
#pragma unroll
        for (int t = 1; t < E; ++t) flat = flat && (m[t].x == m[0].x);
        const T e0 = exp_acc<T>(m[0].x * h);
        fly_factors<E>(m, ph, flat, e0, h, a.inv_n);
    };
    if constexpr (EARLY_PHASE) {
        phase_factors();
        __builtin_amdgcn_sched_barrier(0);          // (keeps the factors ahead of the wait for the field)
    }
    if constexpr (EARLY_FLY) {
        fly();
        __builtin_amdgcn_sched_barrier(0);
    }
    if (fft_tw_lds_entries(N2, E) > 0) __syncthreads();
    line_twiddles_fetch<T, N2, E>(tw, j, ldsT);
    if (!SSFM_ABL_NO_FFT && !INV_ONLY) fft_line<T, N2, E, -1, 0, RI>(v, lds, 0, j, idx, tw);
    if (MODE == FM_FWD_ONLY) {
        if (U16 && a.u16 == 2) {
#pragma unroll
            for (int g = 0; g < E / 2; ++g) {
                u4_t q;
                q.x = v[2 * g].x; q.y = v[2 * g].y; q.z = v[2 * g + 1].x; q.w = v[2 * g + 1].y;
                pass_store<PK>(reinterpret_cast<u4_t*>(Frow) + g * Q + j, q);
            }
            return;
        }
#pragma unroll
        for (int t = 0; t < E; ++t) pass_store<PK>(&Frow[j + t * Q], v[t]);
        return;
    }
    SSFM_STAMP(3);
    if constexpr (!INV_ONLY) {
#pragma unroll
    for (int t = 0; t < E; ++t) v[t] = cmul(v[t], m[t]);
    }
    SSFM_STAMP(4);
    if (!SSFM_ABL_NO_FFT) fft_line<T, N2, E, +1, (fft_nstages(N2, E) == 1 ? 0 : 1), RI>(v, lds, 0, j, idx, tw);
    SSFM_STAMP(5);
    if (U16) {
#pragma unroll
        for (int g = 0; g < E / 2; ++g) {
            u4_t q;
            q.x = v[2 * g].x; q.y = v[2 * g].y; q.z = v[2 * g + 1].x; q.w = v[2 * g + 1].y;
            pass_store<PK>(reinterpret_cast<u4_t*>(Frow) + g * Q + j, q);
        }
    } else if constexpr (H) {
        cx<float>* __restrict__ F32 = reinterpret_cast<cx<float>*>(a.F) + row * N2;
#pragma unroll
        for (int t = 0; t < E; ++t) pass_store<PK>(&F32[j + t * Q], mk<float>((float)v[t].x, (float)v[t].y));
    } else {
#pragma unroll
        for (int t = 0; t < E; ++t) pass_store<PK>(&Frow[j + t * Q], v[t]);
    }
#if SSFM_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    SSFM_STAMP(6);
#if SSFM_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    SSFM_TRACE_END(a);
}

template <typename T, int N2, int ROWS, int E, int MODE, bool U16 = false>
__global__ SSFM_KERNEL_BOUNDS(ROWS * N2 / E, sizeof(T), E) void k_freq(cx<T>* F, const cx<T>* tab, const cx<T>* tw2, const AdaptState<T>* st, T h, T amp, T inv_n, int step, int N1,
                                                                     int rows, int u16
#if SSFM_TRACE
                                                                     , unsigned long long* trace, int trace_slot
#endif
                                                                     ) {
    FreqArgs<T> a;           // (see k_time: every argument of this kernel is a leading scalar, 14 dwords in complex64)
    a.F = F; a.tab = tab; a.tw2 = tw2; a.st = st; a.h = h; a.amp = amp; a.inv_n = inv_n; a.step = step; a.N1 = N1; a.rows = rows; a.u16 = u16;
#if SSFM_TRACE
    a.trace = trace; a.trace_slot = trace_slot;
#endif
    freq_body<T, N2, ROWS, E, MODE, U16, false>(a, blockIdx.x);
}
// ... with the field in complex64 on either side (freq_body, H)
template <typename T, int N2, int ROWS, int E, int MODE>
__global__ SSFM_KERNEL_BOUNDS(ROWS * N2 / E, sizeof(T), E) void k_freq_h(cx<T>* F, const cx<T>* tab, const cx<T>* tw2, const AdaptState<T>* st, T h, T amp, T inv_n, int step, int N1,
                                                                       int rows, int u16, const int* done
#if SSFM_TRACE
                                                                       , unsigned long long* trace, int trace_slot
#endif
                                                                       ) {
    if (done != nullptr && *done) return;            // (queued behind the end of an adaptive run: see time_body, H)
    FreqArgs<T> a;
    a.F = F; a.tab = tab; a.tw2 = tw2; a.st = st; a.h = h; a.amp = amp; a.inv_n = inv_n; a.step = step; a.N1 = N1; a.rows = rows; a.u16 = u16;
#if SSFM_TRACE
    a.trace = trace; a.trace_slot = trace_slot;
#endif
    freq_body<T, N2, ROWS, E, MODE, false, false, true>(a, blockIdx.x);
}
// The two halves of a split plan's row pass (FM_FWD_ONLY, FM_INV_ONLY: one transform, no operator, 113-122 registers) without the two-per-CU pin of k_freq: they
// stream rows of a field that no cache holds (2^23 samples and more per row), and three or four workgroups per CU hide more of that latency than two.
template <typename T, int N2, int ROWS, int E, int MODE, bool U16 = false>
__global__ __launch_bounds__(ROWS * N2 / E) void k_freq_half(cx<T>* F, const cx<T>* tab, const cx<T>* tw2, const AdaptState<T>* st, T h, T amp, T inv_n, int step, int N1,
                                                            int rows, int u16
#if SSFM_TRACE
                                                            , unsigned long long* trace, int trace_slot
#endif
                                                            ) {
    static_assert(MODE == FM_FWD_ONLY || MODE == FM_INV_ONLY, "k_freq_half: the transform-only modes");
    FreqArgs<T> a;
    a.F = F; a.tab = tab; a.tw2 = tw2; a.st = st; a.h = h; a.amp = amp; a.inv_n = inv_n; a.step = step; a.N1 = N1; a.rows = rows; a.u16 = u16;
#if SSFM_TRACE
    a.trace = trace; a.trace_slot = trace_slot;
#endif
    freq_body<T, N2, ROWS, E, MODE, U16, false>(a, blockIdx.x);
}
#if SSFM_TRACE
#define SSFM_FREQ_KERNEL_ARGS(a) (a).F, (a).tab, (a).tw2, (a).st, (a).h, (a).amp, (a).inv_n, (a).step, (a).N1, (a).rows, (a).u16, (a).trace, (a).trace_slot
#else
#define SSFM_FREQ_KERNEL_ARGS(a) (a).F, (a).tab, (a).tw2, (a).st, (a).h, (a).amp, (a).inv_n, (a).step, (a).N1, (a).rows, (a).u16
#endif

constexpr int kSmallTabs = 4;          // operator tables per schedule of the single-launch engines (k_medium, k_small)
// ------------------------------------------------------------------------------ k_medium
// Plans of 2^14 ... 2^17 samples (the reference's example and test sizes): a field is a few dozen column tiles / row groups, a
// step of the two-kernel engine is two launches of 2.6 us each whatever the size (dependent-launch boundary), i.e. 7-10 us.
// Here ONE launch runs the whole fixed-step schedule: workgroup b owns column tile b in the time passes and row group b in the
// frequency passes (both counts are equal by construction: ROWS = N1 * 16 / N2 rows per group), and between two passes all
// workgroups meet at a barrier in device memory -- at most 64 of them, 0.5-1.4 us (profiles/r02_barrier_probe.txt), less than the
// launch it replaces.  The passes are the bodies of k_time / k_freq; what changes is how the field travels between them:
// written with write-through (sc1) stores as ever, read with sc1 loads (PK = true), because reader and writer now belong to
// the same launch (MI355X_MICROARCH.md "Valid forms": sc1 stores, every storing wave drained, one arrival per workgroup, a
// relaxed poll, then sc1 loads only).  |A|^2 stays private to a workgroup (its tile) and keeps its loads: the same CU wrote it.
// A workgroup never waits longer than `patience` (the grid is far below the chip's capacity, but the chip may be shared):
// then the error word is set and the host repeats the run with the two-kernel engine.
template <typename T> struct MediumArgs {
    cx<T>* F;                          // time-order field, in and out
    cx<T>* Y;                          // the field between the passes
    T* P;
    const cx<T>* twA;
    const cx<T>* twB;
    const cx<T>* tw1;
    const cx<T>* tw2;
    const cx<T>* tab[kSmallTabs];      // per distinct step size: exp(D~ h)/N (FM_TABLE) or its phases (FM_PHASE), k_freq's order
    T amp[kSmallTabs];                 // FM_PHASE: the modulus
    const T* hs;                       // the schedule: nsteps step sizes [km]
    const unsigned char* which;        // per step: its table
    unsigned long long* bar;           // 8 arrival counters, the error word, the ticket counter: zeroed before the launch
    unsigned xcc;                      // the XCD the launch's workgroups meet on (plans take turns, so that concurrent plans do not share one)
    unsigned nblk;                     // workgroups that do the work (the launch has that many per XCD)
    unsigned* error;
    long long patience;                // ticks of the 100 MHz clock
    T gamma;
    T inv_n;
    int nsteps;
    int rows;                          // batch rows
    int Qf;
};
constexpr int kBarShards = 8;
constexpr int kBarWords = 64;          // one-XCD form: a flag word per workgroup (no atomics), behind the shards; then the error word and the ticket counter

// every workgroup of the launch has passed here `epoch` times once each counter shows epoch * nblk / 8 arrivals
__device__ __forceinline__ bool medium_barrier(unsigned long long* bar, unsigned* error, long long patience, unsigned long long& epoch,
                                               const unsigned bid, const unsigned nblk, const int tid) {
    __shared__ __attribute__((aligned(16))) int s_bar_ok[4];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // every wave: its write-through stores have landed
    __syncthreads();
    ++epoch;
    if (tid < 64) {
        // every workgroup raises its own word to the epoch; a wavefront of every workgroup reads them all (one per lane).  No atomics at all: sharded
        // L2 atomics were 0.4-0.6 us per step slower
        if (tid == 0) st_l2_u64(&bar[kBarShards + bid], epoch);
        const unsigned long long want = epoch * (nblk / kBarShards);
        const long long t0 = wall_clock64();
        int good = 0;
        // (Two polls in flight half a round trip apart, so that the last arrival is seen sooner, changed nothing: 4.71 / 5.77 / 8.12 / 12.38 against
        // 4.59 / 5.74 / 8.07 / 12.17 us per step at 2^13 x 2 / 2^14 / 2^16 x 2 / the 8176-sample chirp-z line -- the wait is for the slowest workgroup's
        // stores, not for the news of them.  Round 4, tools/medium_time.py.)
        for (;;) {
            unsigned long long got = want;
            if ((unsigned)tid < nblk) got = ld_l2_u64(&bar[kBarShards + tid]) >= epoch ? want : 0ull;
            if (__all(got >= want)) { good = 1; break; }
            if (wall_clock64() - t0 > patience) break;                  // (looking at the clock only every 16th round changes nothing here: 4.59 / 5.70 / 5.82 against
            __builtin_amdgcn_s_sleep(1);                                //  4.61 / 5.70 / 5.81 us per step at 2^13 x 2 / 2^14 / 2^14 x 2, profiles/r05_adaptive.txt)
        }
        if (tid == 0) { s_bar_ok[0] = good; if (!good) atomicExch(error, 1u); }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");       // (no instruction: the loads of the next pass stay below)
    return s_bar_ok[0] != 0;
}
// The one-XCD engines are launched with 8 x nblk workgroups, dealt round robin to the XCDs: the ones on XCD `xcc` take the tiles in the order they arrive
// (the others leave); should fewer than nblk arrive there, the first barrier runs out of patience and the host repeats the run on the launch-per-pass engine.
__device__ __forceinline__ bool medium_ticket(const unsigned xcc, unsigned long long* bar, const unsigned nblk, const int tid, unsigned& bid) {
    if (xcc_id() != xcc) return false;
    __shared__ __attribute__((aligned(16))) unsigned s_bid[4];
    if (tid == 0) s_bid[0] = (unsigned)l2_add_u64(bar + kBarShards + kBarWords + 1, 1ull);
    __syncthreads();
    bid = s_bid[0];
    return bid < nblk;
}
template <typename T, int N1, int N2, int E, int FMODE>
__global__ __launch_bounds__(N1 * 16 / E) void k_medium(const MediumArgs<T> a) {
    constexpr int C = 16, ROWS = N1 * C / N2;
    static_assert(ROWS >= 1 && ROWS * N2 == N1 * C, "the two passes share the workgroup shape");
    const int tid = threadIdx.x;
    const unsigned nblk = a.nblk;
    unsigned bid;
    if (!medium_ticket(a.xcc, a.bar, nblk, tid, bid)) return;
    unsigned long long epoch = 0;
    T pk[E];
    TimeArgs<T> ta;
    ta.F = a.F; ta.Y = a.Y; ta.P = a.P; ta.twN = nullptr; ta.twA = a.twA; ta.twB = a.twB; ta.tw1 = a.tw1; ta.st = nullptr; ta.zlog = nullptr;
    ta.gamma = a.gamma; ta.N2 = N2; ta.rows = a.rows; ta.Qf = a.Qf; ta.step = 0; ta.derive = 0;
    ta.s_in = nullptr; ta.s_out = nullptr; ta.pkeep = pk; ta.mul = nullptr;
    FreqArgs<T> fa;
    fa.F = a.Y; fa.tw2 = a.tw2; fa.st = nullptr; fa.inv_n = a.inv_n; fa.N1 = N1; fa.rows = a.rows; fa.u16 = 1; fa.step = 0;
    const T half = (T)0.5;
    ta.hh_prev = (T)0; ta.hh_next = a.hs[0] * half;
    time_body<T, N1, C, E, TM_BEGIN, true, true>(ta, bid, nblk);
    if (!medium_barrier(a.bar, a.error, a.patience, epoch, bid, nblk, tid)) return;
    for (int s = 0; s < a.nsteps; ++s) {
        const int w = a.which[s];
        fa.tab = a.tab[w]; fa.amp = a.amp[w]; fa.h = a.hs[s];
        freq_body<T, N2, ROWS, E, FMODE, true, true>(fa, bid);
        if (!medium_barrier(a.bar, a.error, a.patience, epoch, bid, nblk, tid)) return;
        ta.hh_prev = a.hs[s] * half;
        if (s + 1 < a.nsteps) {
            ta.hh_next = a.hs[s + 1] * half;
            time_body<T, N1, C, E, TM_MID, true, true>(ta, bid, nblk);
            if (!medium_barrier(a.bar, a.error, a.patience, epoch, bid, nblk, tid)) return;
        } else {
            ta.hh_next = (T)0;
            time_body<T, N1, C, E, TM_END, true, true>(ta, bid, nblk);
        }
    }
}

// ------------------------------------------------------------------------------ k_medium_chirp
// A fixed-step run of a field of ANY length n, 2048 < n <= N / 2, through the chirp-z identity on a line of N = 2^13 ... 2^17 points in one launch on
// one XCD (csrc/chirpz.hip has the algebra; the launch-per-pass form takes five launches, 21-33 us, per step).  The chirps on either side of a step
// cancel against the next step's (c conj(c) = 1 and a rotation commutes with them), so only the run's two ends carry them -- two pointwise launches
// around this one (ssfm_host.hip chirp_medium); in here the line holds v = A c, zero from n up, and a step is four passes:
//     row pass     fft . H0 . ifft        H0 = fft_N(conj(c) wrapped) / N: the n-point forward transform up to a factor c_k
//     column pass  TM_MID_L               ifft, times exp(D~ h) / n (zero from n up: `mul`), fft
//     row pass     fft . H1 . ifft        H1 = fft_N(c wrapped) / N: the inverse one
//     column pass  TM_MID / TM_END        ifft, the second half rotation of this step and the first of the next in one, zero from n up, fft
template <typename T> struct MediumChirpArgs {
    cx<T>* F;                          // BEGIN reads A c (padded) here, END leaves the line here
    cx<T>* Y;
    T* P;
    const cx<T>* twA;
    const cx<T>* twB;
    const cx<T>* tw1;
    const cx<T>* tw2;
    const cx<T>* H[2];                 // the two convolutions' transfer functions, k_freq<FM_TABLE>'s order
    const cx<T>* mul[kSmallTabs];      // per distinct step size: exp(D~ h) / n at the positions below n, zero above (time order of the line)
    const T* hs;
    const unsigned char* which;
    unsigned long long* bar;           // as MediumArgs::bar
    unsigned xcc, nblk;
    unsigned* error;
    long long patience;
    T gamma;
    int n;                             // samples of the field
    int nsteps;
    int rows;
    int Qf;
};
template <typename T, int N1, int N2, int E>
__global__ __launch_bounds__(N1 * 16 / E) void k_medium_chirp(const MediumChirpArgs<T> a) {
    constexpr int C = 16, ROWS = N1 * C / N2;
    static_assert(ROWS >= 1 && ROWS * N2 == N1 * C, "the two passes share the workgroup shape");
    const int tid = threadIdx.x;
    const unsigned nblk = a.nblk;
    unsigned bid;
    if (!medium_ticket(a.xcc, a.bar, nblk, tid, bid)) return;
    unsigned long long epoch = 0;
    T pk[E];
    TimeArgs<T> ta;
    ta.F = a.F; ta.Y = a.Y; ta.P = a.P; ta.twN = nullptr; ta.twA = a.twA; ta.twB = a.twB; ta.tw1 = a.tw1; ta.st = nullptr; ta.zlog = nullptr;
    ta.gamma = a.gamma; ta.N2 = N2; ta.rows = a.rows; ta.Qf = a.Qf; ta.step = 0; ta.derive = 0;
    ta.s_in = nullptr; ta.s_out = nullptr; ta.pkeep = pk; ta.mul = nullptr; ta.keep = a.n;
    TimeArgs<T> tl = ta;
    tl.gamma = (T)0; tl.hh_prev = tl.hh_next = (T)0; tl.keep = 0;
    FreqArgs<T> fa;
    fa.F = a.Y; fa.tw2 = a.tw2; fa.st = nullptr; fa.inv_n = (T)0; fa.N1 = N1; fa.rows = a.rows; fa.u16 = 1; fa.step = 0; fa.h = (T)0; fa.amp = (T)0;
    const T half = (T)0.5;
    ta.hh_prev = (T)0; ta.hh_next = a.hs[0] * half;
    time_body<T, N1, C, E, TM_BEGIN, true, true>(ta, bid, nblk);
    if (!medium_barrier(a.bar, a.error, a.patience, epoch, bid, nblk, tid)) return;
    for (int s = 0; s < a.nsteps; ++s) {
        tl.mul = a.mul[a.which[s]];
#pragma unroll 1
        for (int half_step = 0; half_step < 2; ++half_step) {
            fa.tab = a.H[half_step];
            freq_body<T, N2, ROWS, E, FM_TABLE, true, true>(fa, bid);
            if (!medium_barrier(a.bar, a.error, a.patience, epoch, bid, nblk, tid)) return;
            if (half_step == 0) {
                time_body<T, N1, C, E, TM_MID_L, true, true>(tl, bid, nblk);
                if (!medium_barrier(a.bar, a.error, a.patience, epoch, bid, nblk, tid)) return;
            }
        }
        ta.hh_prev = a.hs[s] * half;
        if (s + 1 < a.nsteps) {
            ta.hh_next = a.hs[s + 1] * half;
            time_body<T, N1, C, E, TM_MID, true, true>(ta, bid, nblk);
            if (!medium_barrier(a.bar, a.error, a.patience, epoch, bid, nblk, tid)) return;
        } else {
            ta.hh_next = (T)0;
            time_body<T, N1, C, E, TM_END, true, true>(ta, bid, nblk);
        }
    }
}

// ------------------------------------------------------------------------------ k_medium_chirp_adapt
// ... and the ADAPTIVE run of such a field: k_medium_chirp's four passes per step with k_medium_adapt's step control (the maxima of |A|^2 -- of the
// line below n: |A c| = |A| -- meet in TM_MID_A); exp(D~ h) / n is formed in TM_MID_L from D~ and the step size of the state in LDS.
template <typename T> struct MediumChirpAdaptArgs {
    cx<T>* F;
    cx<T>* Y;
    T* P;
    const cx<T>* twA;
    const cx<T>* twB;
    const cx<T>* tw1;
    const cx<T>* tw2;
    const cx<T>* H[2];
    const cx<T>* Dt;                   // D~, n entries, natural frequency order
    AdaptState<T>* st;
    T* zlog;
    unsigned long long* bar;
    unsigned xcc, nblk;
    unsigned* error;
    long long patience;
    T gamma;
    T inv_len;                         // 1 / n
    int n;
    int rows;
    int Qf;
};
template <typename T, int N1, int N2, int E>
__global__ __launch_bounds__(N1 * 16 / E) void k_medium_chirp_adapt(const MediumChirpAdaptArgs<T> a) {
    constexpr int C = 16, ROWS = N1 * C / N2;
    static_assert(ROWS >= 1 && ROWS * N2 == N1 * C, "the two passes share the workgroup shape");
    const int tid = threadIdx.x;
    const unsigned nblk = a.nblk;
    unsigned bid;
    if (!medium_ticket(a.xcc, a.bar, nblk, tid, bid)) return;
    __shared__ __attribute__((aligned(16))) StepState<T> s_state[2];       // [0] the step being taken, [1] where TM_MID_A leaves the next one
    unsigned long long epoch = 0;
    StepState<T> S = a.st->cur[0];
    if (S.done) return;
    TimeArgs<T> ta;
    ta.F = a.F; ta.Y = a.Y; ta.P = a.P; ta.twN = nullptr; ta.twA = a.twA; ta.twB = a.twB; ta.tw1 = a.tw1; ta.st = nullptr; ta.zlog = a.zlog;
    ta.gamma = a.gamma; ta.N2 = N2; ta.rows = a.rows; ta.Qf = a.Qf; ta.step = 0; ta.derive = 0;
    T pk[E];
    ta.s_in = &s_state[0]; ta.s_out = &s_state[1]; ta.pkeep = pk; ta.mul = nullptr; ta.keep = a.n;
    TimeArgs<T> tl = ta;
    tl.gamma = (T)0; tl.hh_prev = (T)0; tl.hh_next = a.inv_len; tl.mul = a.Dt; tl.zlog = nullptr;
    FreqArgs<T> fa;
    fa.F = a.Y; fa.tw2 = a.tw2; fa.st = nullptr; fa.inv_n = (T)0; fa.N1 = N1; fa.rows = a.rows; fa.u16 = 1; fa.step = 0; fa.h = (T)0; fa.amp = (T)0;
    const T half = (T)0.5;
    ta.hh_prev = (T)0; ta.hh_next = S.h * half;
    time_body<T, N1, C, E, TM_BEGIN, true, true>(ta, bid, nblk);
    if (tid == 0) { s_state[0] = S; s_state[1].steps = -0x7fffffff; }
    if (!medium_barrier(a.bar, a.error, a.patience, epoch, bid, nblk, tid)) return;           // (its barriers also publish s_state to the workgroup)
    ta.st = a.st;
    for (int step = 0;; ++step) {
        fa.tab = a.H[0];
        freq_body<T, N2, ROWS, E, FM_TABLE, true, true>(fa, bid);
        if (!medium_barrier(a.bar, a.error, a.patience, epoch, bid, nblk, tid)) return;
        time_body<T, N1, C, E, TM_MID_L, true, true>(tl, bid, nblk);
        if (!medium_barrier(a.bar, a.error, a.patience, epoch, bid, nblk, tid)) return;
        fa.tab = a.H[1];
        freq_body<T, N2, ROWS, E, FM_TABLE, true, true>(fa, bid);
        if (!medium_barrier(a.bar, a.error, a.patience, epoch, bid, nblk, tid)) return;
        ta.step = step;
        time_body<T, N1, C, E, TM_MID_A, true, true>(ta, bid, nblk);
        __syncthreads();
        const StepState<T> Sn = s_state[1];
        if (Sn.steps == -0x7fffffff) return;                   // the hand-over of the maxima ran out of patience (error word set)
        S = Sn;
        if (S.done) break;
        __syncthreads();
        if (tid == 0) { s_state[0] = S; s_state[1].steps = -0x7fffffff; }
        if (!medium_barrier(a.bar, a.error, a.patience, epoch, bid, nblk, tid)) return;
    }
    if (bid == 0 && tid == 0) a.st->cur[0] = S;                // (the host reads cur[0])
}

// ------------------------------------------------------------------------------ k_medium_adapt
// The ADAPTIVE run of a medium plan in one launch on one XCD: k_medium's passes with the step size found on the way.  Every workgroup
// keeps the step control state itself (the same arithmetic on the same maxima: the same bits everywhere, as in TM_MID_A); the maxima
// travel as one word per workgroup through the XCD's L2 (AdaptState::wgmax); workgroup 0 writes the z log and, at the end, the state
// the host reads.  The first step's size is in st->cur[0] (k_absmax + k_step_control ran before the launch).
template <typename T> struct MediumAdaptArgs {
    cx<T>* F;
    cx<T>* Y;
    T* P;
    const cx<T>* twA;
    const cx<T>* twB;
    const cx<T>* tw1;
    const cx<T>* tw2;
    const cx<T>* D;                    // D~ in k_freq<FM_FLY>'s order
    AdaptState<T>* st;
    T* zlog;
    unsigned long long* bar;           // as MediumArgs::bar
    unsigned xcc, nblk;
    unsigned* error;
    long long patience;
    T gamma;
    T inv_n;
    int rows;
    int Qf;
};
template <typename T, int N1, int N2, int E>
__global__ __launch_bounds__(N1 * 16 / E) void k_medium_adapt(const MediumAdaptArgs<T> a) {
    constexpr int C = 16, ROWS = N1 * C / N2;
    static_assert(ROWS >= 1 && ROWS * N2 == N1 * C, "the two passes share the workgroup shape");
    const int tid = threadIdx.x;
    const unsigned nblk = a.nblk;
    unsigned bid;
    if (!medium_ticket(a.xcc, a.bar, nblk, tid, bid)) return;
    __shared__ __attribute__((aligned(16))) StepState<T> s_state[2];       // [0] the step being taken, [1] where TM_MID_A leaves the next one
    unsigned long long epoch = 0;
    StepState<T> S = a.st->cur[0];
    if (S.done) return;
    TimeArgs<T> ta;
    ta.F = a.F; ta.Y = a.Y; ta.P = a.P; ta.twN = nullptr; ta.twA = a.twA; ta.twB = a.twB; ta.tw1 = a.tw1; ta.st = nullptr; ta.zlog = a.zlog;
    ta.gamma = a.gamma; ta.N2 = N2; ta.rows = a.rows; ta.Qf = a.Qf; ta.step = 0; ta.derive = 0;
    T pk[E];
    ta.s_in = &s_state[0]; ta.s_out = &s_state[1]; ta.pkeep = pk; ta.mul = nullptr;
    FreqArgs<T> fa;
    fa.F = a.Y; fa.tab = a.D; fa.tw2 = a.tw2; fa.st = nullptr; fa.inv_n = a.inv_n; fa.N1 = N1; fa.rows = a.rows; fa.u16 = 1; fa.step = 0; fa.amp = (T)0;
    const T half = (T)0.5;
    ta.hh_prev = (T)0; ta.hh_next = S.h * half;
    time_body<T, N1, C, E, TM_BEGIN, true, true>(ta, bid, nblk);
    if (!medium_barrier(a.bar, a.error, a.patience, epoch, bid, nblk, tid)) return;
    ta.st = a.st;
    for (int step = 0;; ++step) {
        fa.h = S.h;
        freq_body<T, N2, ROWS, E, FM_FLY, true, true>(fa, bid);
        if (tid == 0) { s_state[0] = S; s_state[1].steps = -0x7fffffff; }
        if (!medium_barrier(a.bar, a.error, a.patience, epoch, bid, nblk, tid)) return;       // (its barriers also publish s_state to the workgroup)
        ta.step = step;
        time_body<T, N1, C, E, TM_MID_A, true, true>(ta, bid, nblk);
        __syncthreads();
        const StepState<T> Sn = s_state[1];
        if (Sn.steps == -0x7fffffff) return;                   // the hand-over of the maxima ran out of patience (error word set)
        S = Sn;
        if (S.done) break;
        if (!medium_barrier(a.bar, a.error, a.patience, epoch, bid, nblk, tid)) return;
    }
    if (bid == 0 && tid == 0) a.st->cur[0] = S;                // (the host reads cur[0])
}

// ------------------------------------------------------------------------------ k_small
// A field of up to 8192 samples is ONE line of the row transform: a workgroup keeps a whole row in its registers and runs
// the complete fixed-step schedule in a single launch -- rotation, N-point FFT, exp(D~ h)/N, inverse FFT, rotation, for every
// step, with LDS for the exchanges only.  The two-kernel engine spends 2.6 us per kernel on such a plan whatever its size
// (dependent-launch boundary; profiles/r02_small_graph.txt: eager and hipGraph alike), i.e. 5.2-6 us per step.
// The arithmetic per step is k_time's and k_freq<FM_TABLE>'s: the second half rotation of a step and the first of the next
// are one rotation by the sum of the two phases, |A|^2 of the step's start stays in registers (`pold`).
template <typename T> struct SmallArgs {
    cx<T>* F;                          // batch rows of N samples, time order, advanced in place
    const cx<T>* tab[kSmallTabs];      // exp(D~ h)/N per distinct step size, at freq_tab_pos(k, N / E)
    const T* hs;                       // the schedule: nsteps step sizes [km]
    const unsigned char* which;        // per step: its table
    const cx<T>* tw;                   // stage twiddles of the N-point line (make_line_table(N, E))
    cx<T>* snap;                       // z-resolved capture (devices.py:1184-1186): the field after step s at snap + (s + 1) * rows * N, or NULL
    T gamma;
    int nsteps;
};
template <typename T, int N, int E>
__global__ __launch_bounds__(N / E) void k_small(const SmallArgs<T> a) {
    constexpr int Q = N / E;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cx<T>* lds = reinterpret_cast<cx<T>*>(smem_raw);
    const int j = threadIdx.x;
    cx<T>* __restrict__ Frow = a.F + (long long)blockIdx.x * N;
    using RI = RowIdx<row_pad_shift(E)>;
    const RI idx{0};
    cx<T> v[E];
    cx<T> m[E];
#pragma unroll
    for (int t = 0; t < E; ++t) v[t] = Frow[j + t * Q];
    LineTw<T, N, E> tw;
    cx<T>* ldsT = lds + (fft_nstages(N, E) > 1 ? row_lds_elems(N, E) : 0);
    line_twiddles_issue<T, N, E>(tw, j, a.tw, ldsT, j, Q);
    if (fft_tw_lds_entries(N, E) > 0) __syncthreads();
    line_twiddles_fetch<T, N, E>(tw, j, ldsT);
    const T half = (T)0.5;
    T pold[E];
    T phi[E];
    {   // first half step of step 0 (k_time<TM_BEGIN>)
        const T hh = half * a.hs[0];
#pragma unroll
        for (int t = 0; t < E; ++t) {
            const T p = v[t].x * v[t].x + v[t].y * v[t].y;
            T ph = (T)0;
            ph += hh * (a.gamma * p);
            pold[t] = p;
            phi[t] = ph;
        }
        rotate_all<E>(v, phi);
    }
    typedef T m4_t __attribute__((ext_vector_type(4)));
    for (int s = 0; s < a.nsteps; ++s) {
        const T h = a.hs[s];
        const m4_t* __restrict__ T4 = reinterpret_cast<const m4_t*>(a.tab[a.which[s]]) + j;
#pragma unroll
        for (int g = 0; g < E / 2; ++g) {
            const m4_t q = T4[g * Q];
            m[2 * g] = mk<T>(q.x, q.y);
            m[2 * g + 1] = mk<T>(q.z, q.w);
        }
        // (exchange parity 1: the LDS buffer was last read by the previous transform, a barrier precedes its rewrite)
        fft_line<T, N, E, -1, 1, RI>(v, lds, 0, j, idx, tw);
#pragma unroll
        for (int t = 0; t < E; ++t) v[t] = cmul(v[t], m[t]);
        fft_line<T, N, E, +1, 1, RI>(v, lds, 0, j, idx, tw);
        // k_time<TM_MID> (TM_END after the last step): both half-step phases in one rotation
        const T hh_prev = half * h;
        const bool more = s + 1 < a.nsteps;
        const T hh_next = more ? half * a.hs[s + 1] : (T)0;
#pragma unroll
        for (int t = 0; t < E; ++t) {
            const T p = v[t].x * v[t].x + v[t].y * v[t].y;
            T ph = hh_prev * (a.gamma * pold[t]);
            if (more && a.snap == nullptr) ph += hh_next * (a.gamma * p);
            pold[t] = p;
            phi[t] = ph;
        }
        rotate_all<E>(v, phi);
        if (a.snap != nullptr) {
            // a capture needs the field BETWEEN the two half rotations (k_time<TM_END>, then <TM_BEGIN>, as the two-kernel engine does)
            cx<T>* __restrict__ srow = a.snap + ((long long)(s + 1) * gridDim.x + blockIdx.x) * N;
#pragma unroll
            for (int t = 0; t < E; ++t) srow[j + t * Q] = v[t];
            if (more) {
#pragma unroll
                for (int t = 0; t < E; ++t) {
                    T ph = (T)0;
                    ph += hh_next * (a.gamma * pold[t]);
                    phi[t] = ph;
                }
                rotate_all<E>(v, phi);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < E; ++t) Frow[j + t * Q] = v[t];
}

// ------------------------------------------------------------------------------ k_small_chirp
// A field of ANY length n <= 2048 (the reference's own generators: a PRBS-7 word at 16 samples per bit is 2032 samples) through the chirp-z identity
// on ONE line of N >= 2 n - 1 points, the whole fixed-step schedule in a single launch (csrc/chirpz.hip has the algebra; the launch-per-pass form
// takes 5 launches, 22 us, per step).  Thread j keeps the samples m = j + t Q of its row; per step
//     v = A exp(i phi) c            (c_m = exp(-i pi m^2 / n); zero from n up to N)
//     v = ifft_N(fft_N(v) H)        H = fft_N(conj(c), wrapped around the line) / N: the forward n-point transform, up to a factor c_k
//     v = v exp(D~ h)               (zero from n up)
//     v = ifft_N(fft_N(v) conj(H))  the inverse one: its kernel is the conjugate of a sequence that is symmetric on the line, so is its transform
//     A = v conj(c) / n, rotated by the step's second half phase and the next step's first in one rotation (as k_small does).
// H is computed here (one transform per launch) in the order the line transform leaves its output, so no table in any order is needed; exp(D~ h) is
// formed in registers whenever the step size changes (normally twice per run).
template <typename T> struct SmallChirpArgs {
    cx<T>* A;                 // rows of n samples, natural order, advanced in place
    const cx<T>* chirp;       // c, n entries
    const cx<T>* Dt;          // D~, n entries, natural frequency order
    const double* hs;         // the schedule: nsteps step sizes [km]
    const cx<T>* tw;          // stage twiddles of the N-point line
    T gamma;
    int n;
    int nsteps;
};
template <typename T, int N, int E>
__global__ __launch_bounds__(N / E) void k_small_chirp(const SmallChirpArgs<T> a) {
    constexpr int Q = N / E;
    constexpr int EH = E / 2;              // n <= N / 2: the samples m = j + t Q with t >= E / 2 are the zero padding, whatever the thread
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cx<T>* lds = reinterpret_cast<cx<T>*>(smem_raw);
    const int j = threadIdx.x;
    const int n = a.n;
    cx<T>* __restrict__ Arow = a.A + (long long)blockIdx.x * n;
    using RI = RowIdx<row_pad_shift(E)>;
    const RI idx{0};
    cx<T> v[E];
    LineTw<T, N, E> tw;
    cx<T>* ldsT = lds + (fft_nstages(N, E) > 1 ? row_lds_elems(N, E) : 0);
    cx<T>* ldsH = ldsT + fft_tw_lds_entries(N, E);          // H, N entries, as this thread's E values of the line transform's output: [t Q + j]
    line_twiddles_issue<T, N, E>(tw, j, a.tw, ldsT, j, Q);
    if (fft_tw_lds_entries(N, E) > 0) __syncthreads();
    line_twiddles_fetch<T, N, E>(tw, j, ldsT);
    const cx<T> zero = mk<T>((T)0, (T)0);
#pragma unroll
    for (int t = 0; t < E; ++t) {
        const int m = j + t * Q, d = m < N - m ? m : N - m;
        v[t] = zero;
        if (d < n) { const cx<T> c = a.chirp[d]; v[t] = mk<T>(c.x, -c.y); }
    }
    fft_line<T, N, E, -1, 1, RI>(v, lds, 0, j, idx, tw);
    const T inv_line = (T)1 / (T)N, inv_n = (T)1 / (T)n;
#pragma unroll
    for (int t = 0; t < E; ++t) ldsH[t * Q + j] = mk<T>(v[t].x * inv_line, v[t].y * inv_line);       // (read back by this thread only)
    cx<T> cw[EH], ex[EH];
#pragma unroll
    for (int t = 0; t < EH; ++t) {
        const int m = j + t * Q;
        v[t] = zero; cw[t] = zero; ex[t] = zero;
        if (m < n) { v[t] = Arow[m]; cw[t] = a.chirp[m]; }
    }
    const T half = (T)0.5;
    T pold[EH];
    T phi[EH];
    cx<T> vh[EH];
    {
        const T hh = half * (T)a.hs[0];
#pragma unroll
        for (int t = 0; t < EH; ++t) {
            const T p = v[t].x * v[t].x + v[t].y * v[t].y;
            T ph = (T)0;
            ph += hh * (a.gamma * p);
            pold[t] = p;
            phi[t] = ph;
            vh[t] = v[t];
        }
        rotate_all<EH>(vh, phi);
#pragma unroll
        for (int t = 0; t < EH; ++t) v[t] = vh[t];
    }
    double hprev = 0.0;
    for (int s = 0; s < a.nsteps; ++s) {
        const double hd = a.hs[s];
        if (s == 0 || hd != hprev) {
#pragma unroll
            for (int t = 0; t < EH; ++t) {
                const int m = j + t * Q;
                if (m < n) {
                    const cx<T> d = a.Dt[m];
                    T sn, cs;
                    sincos_acc<T>(d.y * (T)hd, sn, cs);
                    const T g = exp_acc<T>(d.x * (T)hd);
                    ex[t] = mk<T>(g * cs, g * sn);
                }
            }
            hprev = hd;
        }
        const T h = (T)hd;
#pragma unroll
        for (int t = 0; t < EH; ++t) { v[t] = cmul(v[t], cw[t]); v[t + EH] = zero; }
        fft_line<T, N, E, -1, 1, RI>(v, lds, 0, j, idx, tw);
#pragma unroll
        for (int t = 0; t < E; ++t) v[t] = cmul(v[t], ldsH[t * Q + j]);
        fft_line<T, N, E, +1, 1, RI>(v, lds, 0, j, idx, tw);
#pragma unroll
        for (int t = 0; t < EH; ++t) { v[t] = cmul(v[t], ex[t]); v[t + EH] = zero; }
        fft_line<T, N, E, -1, 1, RI>(v, lds, 0, j, idx, tw);
#pragma unroll
        for (int t = 0; t < E; ++t) v[t] = cmulc(v[t], ldsH[t * Q + j]);
        fft_line<T, N, E, +1, 1, RI>(v, lds, 0, j, idx, tw);
        const T hh_prev = half * h;
        const bool more = s + 1 < a.nsteps;
        const T hh_next = more ? half * (T)a.hs[s + 1] : (T)0;
#pragma unroll
        for (int t = 0; t < EH; ++t) {
            const cx<T> y = cmulc(v[t], cw[t]);
            vh[t] = mk<T>(y.x * inv_n, y.y * inv_n);
            const T p = vh[t].x * vh[t].x + vh[t].y * vh[t].y;
            T ph = hh_prev * (a.gamma * pold[t]);
            if (more) ph += hh_next * (a.gamma * p);
            pold[t] = p;
            phi[t] = ph;
        }
        rotate_all<EH>(vh, phi);
#pragma unroll
        for (int t = 0; t < EH; ++t) v[t] = vh[t];
    }
#pragma unroll
    for (int t = 0; t < EH; ++t) {
        const int m = j + t * Q;
        if (m < n) Arow[m] = v[t];
    }
}

// The ADAPTIVE run (h = phi_max / (|gamma| max |A|^2), the reference's default: devices.py:1155-1161, 1172-1196 with h = None) of such a field in one
// launch.  The rows of a signal share the step size, and a row needs a whole workgroup's registers, so the workgroups exchange their maxima every
// step: each stores its own as two 8-byte words (tag << 32 | half of the double's bit pattern; tag = exchange number, the slots alternate) and
// reads the others' -- relaxed agent-scope accesses and nothing else, the tag inside the word is the ordering.  |A|^2 is the same before and after a
// rotation, so the next step size is known before the phases are applied: one rotation per step here too.  The step rule is ssfm_chirp_propagate's
// (chirpz.hip k_chirp_control), evaluated by every workgroup from the same maximum in the caller's arithmetic (RT = float in complex64 mode).
template <typename T> struct SmallChirpAdaptArgs {
    cx<T>* A;
    const cx<T>* chirp;
    const cx<T>* Dt;
    const cx<T>* tw;
    unsigned long long* xw;   // [2][rows][2] exchange words, zeroed before the launch
    double* zlog;             // z after every step, max_steps + 1 entries
    int* out;                 // [0] steps taken, [1] != 0: an exchange ran out of patience, [2] rows that stored their result (0 with [1] set: A is untouched)
    double phi_max, abs_gamma, length;
    long long patience;       // 100 MHz ticks
    T gamma;
    int n;
    int max_steps;
    int f32;
};
template <typename RT>
__device__ __forceinline__ void chirp_step_rule(double amax, double phi_max, double abs_gamma, double L, double z, double& h, double& znext) {
    const RT zz = (RT)z;
    RT hh = (RT)phi_max / ((RT)abs_gamma * (RT)amax);
    const RT left = (RT)((RT)L - zz);
    hh = hh < left ? hh : left;
    h = (double)hh;
    znext = (double)(RT)(zz + hh);
}
template <typename T, int N, int E>
__global__ __launch_bounds__(N / E) void k_small_chirp_adapt(const SmallChirpAdaptArgs<T> a) {
    constexpr int Q = N / E;
    constexpr int EH = E / 2;
    constexpr int NW = Q / 64 > 0 ? Q / 64 : 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __shared__ double s_red[NW + 1];
    cx<T>* lds = reinterpret_cast<cx<T>*>(smem_raw);
    const int j = threadIdx.x;
    const int n = a.n;
    const int rows = gridDim.x, row = blockIdx.x;
    cx<T>* __restrict__ Arow = a.A + (long long)row * n;
    using RI = RowIdx<row_pad_shift(E)>;
    const RI idx{0};
    cx<T> v[E];
    LineTw<T, N, E> tw;
    cx<T>* ldsT = lds + (fft_nstages(N, E) > 1 ? row_lds_elems(N, E) : 0);
    cx<T>* ldsH = ldsT + fft_tw_lds_entries(N, E);
    line_twiddles_issue<T, N, E>(tw, j, a.tw, ldsT, j, Q);
    if (fft_tw_lds_entries(N, E) > 0) __syncthreads();
    line_twiddles_fetch<T, N, E>(tw, j, ldsT);
    const cx<T> zero = mk<T>((T)0, (T)0);
#pragma unroll
    for (int t = 0; t < E; ++t) {
        const int m = j + t * Q, d = m < N - m ? m : N - m;
        v[t] = zero;
        if (d < n) { const cx<T> c = a.chirp[d]; v[t] = mk<T>(c.x, -c.y); }
    }
    fft_line<T, N, E, -1, 1, RI>(v, lds, 0, j, idx, tw);
    const T inv_line = (T)1 / (T)N, inv_n = (T)1 / (T)n;
#pragma unroll
    for (int t = 0; t < E; ++t) ldsH[t * Q + j] = mk<T>(v[t].x * inv_line, v[t].y * inv_line);
    cx<T> cw[EH], ex[EH], vh[EH];
    T pold[EH], phi[EH];
#pragma unroll
    for (int t = 0; t < EH; ++t) {
        const int m = j + t * Q;
        vh[t] = zero; cw[t] = zero; ex[t] = zero;
        if (m < n) { vh[t] = Arow[m]; cw[t] = a.chirp[m]; }
    }
    // the maximum of |A|^2 over all rows: exchange number `tag` (1, 2, ...).  false: out of patience
    auto all_rows_max = [&](double mine, unsigned tag, double& out) -> bool {
        mine = wave_total<N / E, true>(mine);
        if ((j & 63) == 0) s_red[j >> 6] = mine;
        __syncthreads();
        if (j == 0) {
            double m = s_red[0];
            for (int w = 1; w < NW; ++w) m = s_red[w] > m ? s_red[w] : m;
            const unsigned long long bits = (unsigned long long)__double_as_longlong(m), tg = (unsigned long long)tag << 32;
            unsigned long long* slot = a.xw + ((long long)(tag & 1) * rows + row) * 2;
            __hip_atomic_store(&slot[0], tg | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&slot[1], tg | (bits & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const long long t0 = wall_clock64();
            bool ok = true;
            for (int r = 0; r < rows && ok; ++r) {
                if (r == row) continue;
                const unsigned long long* src = a.xw + ((long long)(tag & 1) * rows + r) * 2;
                unsigned long long w0, w1;
                for (;;) {
                    w0 = __hip_atomic_load(&src[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    w1 = __hip_atomic_load(&src[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((w0 >> 32) == tag && (w1 >> 32) == tag) break;
                    if (wall_clock64() - t0 > a.patience) { ok = false; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (ok) {
                    const double other = __longlong_as_double((long long)((w0 << 32) | (w1 & 0xffffffffull)));
                    m = other > m ? other : m;
                }
            }
            s_red[NW] = ok ? m : -1.0;
        }
        __syncthreads();
        out = s_red[NW];
        __syncthreads();                 // (s_red is rewritten by the next exchange)
        return out >= 0.0;
    };
    auto fail_out = [&]() { if (j == 0) a.out[1] = 1; };
    double pm = 0.0;
#pragma unroll
    for (int t = 0; t < EH; ++t) {
        const T p = vh[t].x * vh[t].x + vh[t].y * vh[t].y;
        pold[t] = p;
        pm = (double)p > pm ? (double)p : pm;
    }
    unsigned tag = 1;
    double amax;
    if (!all_rows_max(pm, tag, amax)) { fail_out(); return; }
    double z = 0.0, h, znext;
    int steps = 0;
    if (a.f32) chirp_step_rule<float>(amax, a.phi_max, a.abs_gamma, a.length, z, h, znext);
    else chirp_step_rule<double>(amax, a.phi_max, a.abs_gamma, a.length, z, h, znext);
    if (row == 0 && j == 0) a.zlog[0] = 0.0;
    const T half = (T)0.5;
    {
        const T hh = half * (T)h;
#pragma unroll
        for (int t = 0; t < EH; ++t) {
            T ph = (T)0;
            ph += hh * (a.gamma * pold[t]);
            phi[t] = ph;
        }
        rotate_all<EH>(vh, phi);
    }
    for (;;) {
#pragma unroll
        for (int t = 0; t < EH; ++t) {
            const int m = j + t * Q;
            if (m < n) {
                const cx<T> d = a.Dt[m];
                T sn, cs;
                sincos_acc<T>(d.y * (T)h, sn, cs);
                const T g = exp_acc<T>(d.x * (T)h);
                ex[t] = mk<T>(g * cs, g * sn);
            }
        }
#pragma unroll
        for (int t = 0; t < EH; ++t) { v[t] = cmul(vh[t], cw[t]); v[t + EH] = zero; }
        fft_line<T, N, E, -1, 1, RI>(v, lds, 0, j, idx, tw);
#pragma unroll
        for (int t = 0; t < E; ++t) v[t] = cmul(v[t], ldsH[t * Q + j]);
        fft_line<T, N, E, +1, 1, RI>(v, lds, 0, j, idx, tw);
#pragma unroll
        for (int t = 0; t < EH; ++t) { v[t] = cmul(v[t], ex[t]); v[t + EH] = zero; }
        fft_line<T, N, E, -1, 1, RI>(v, lds, 0, j, idx, tw);
#pragma unroll
        for (int t = 0; t < E; ++t) v[t] = cmulc(v[t], ldsH[t * Q + j]);
        fft_line<T, N, E, +1, 1, RI>(v, lds, 0, j, idx, tw);
        T pnew[EH];
        pm = 0.0;
#pragma unroll
        for (int t = 0; t < EH; ++t) {
            const cx<T> y = cmulc(v[t], cw[t]);
            vh[t] = mk<T>(y.x * inv_n, y.y * inv_n);
            pnew[t] = vh[t].x * vh[t].x + vh[t].y * vh[t].y;
            pm = (double)pnew[t] > pm ? (double)pnew[t] : pm;
        }
        ++tag;
        if (!all_rows_max(pm, tag, amax)) { fail_out(); return; }
        // (k_chirp_control: the step just taken is logged, then the run ends or the next size follows from the maximum)
        z = znext;
        ++steps;
        if (row == 0 && j == 0) a.zlog[steps] = z;
        const bool done = a.f32 ? (!((float)z < (float)a.length) || steps >= a.max_steps) : (!(z < a.length) || steps >= a.max_steps);
        double hn = 0.0;
        if (!done) {
            if (a.f32) chirp_step_rule<float>(amax, a.phi_max, a.abs_gamma, a.length, z, hn, znext);
            else chirp_step_rule<double>(amax, a.phi_max, a.abs_gamma, a.length, z, hn, znext);
        }
        const T hh_prev = half * (T)h, hh_next = half * (T)hn;
#pragma unroll
        for (int t = 0; t < EH; ++t) {
            T ph = hh_prev * (a.gamma * pold[t]);
            if (!done) ph += hh_next * (a.gamma * pnew[t]);
            pold[t] = pnew[t];
            phi[t] = ph;
        }
        rotate_all<EH>(vh, phi);
        if (done) break;
        h = hn;
    }
#pragma unroll
    for (int t = 0; t < EH; ++t) {
        const int m = j + t * Q;
        if (m < n) Arow[m] = vh[t];
    }
    if (j == 0) atomicAdd(&a.out[2], 1);
    if (row == 0 && j == 0) a.out[0] = steps;
}

// The adaptive run of a small plan (reference devices.py:1155-1161, 1172-1196 with h = None) in ONE launch: ROWS rows (the
// polarisations of a signal share the step size: the maximum is taken over all of them) in one workgroup, D~ in registers
// (exp(D~ h) formed per step as k_freq<FM_FLY> does), the step control of step_advance() between the inverse transform and
// the rotation: |A|^2 is the same before and after a rotation, so the next step size is known before the phases are applied.
template <typename T> struct SmallAdaptArgs {
    cx<T>* F;                 // ROWS rows of N samples, time order, advanced in place
    const cx<T>* D;           // D~ at freq_tab_pos(k, N / E)
    const cx<T>* tw;
    AdaptState<T>* st;        // length, phi_max, abs_gamma, adaptive, max_steps in; cur[0] = final state out
    T* zlog;
    T gamma;
    T inv_n;
    int single_step;
};
template <typename T, int N, int E, int ROWS>
__global__ __launch_bounds__(ROWS * N / E) void k_small_adapt(const SmallAdaptArgs<T> a) {
    constexpr int Q = N / E;
    constexpr int NW = (ROWS * Q + 63) / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cx<T>* lds = reinterpret_cast<cx<T>*>(smem_raw);
    __shared__ unsigned long long wave_max[2][NW];
    const int tid = threadIdx.x;
    const int j = tid % Q;
    const int rr = tid / Q;
    cx<T>* __restrict__ Frow = a.F + (long long)rr * N;
    using RI = RowIdx<row_pad_shift(E)>;
    const RI idx{rr * row_lds_elems(N, E)};
    cx<T> v[E];
    cx<T> d[E];
    cx<T> m[E];
#pragma unroll
    for (int t = 0; t < E; ++t) v[t] = Frow[j + t * Q];
    {
        typedef T m4_t __attribute__((ext_vector_type(4)));
        const m4_t* __restrict__ D4 = reinterpret_cast<const m4_t*>(a.D) + j;
#pragma unroll
        for (int g = 0; g < E / 2; ++g) {
            const m4_t q = D4[g * Q];
            d[2 * g] = mk<T>(q.x, q.y);
            d[2 * g + 1] = mk<T>(q.z, q.w);
        }
    }
    LineTw<T, N, E> tw;
    cx<T>* ldsT = lds + (fft_nstages(N, E) > 1 ? ROWS * row_lds_elems(N, E) : 0);
    line_twiddles_issue<T, N, E>(tw, j, a.tw, ldsT, tid, ROWS * Q);
    if (fft_tw_lds_entries(N, E) > 0) __syncthreads();
    line_twiddles_fetch<T, N, E>(tw, j, ldsT);
    bool flat = true;
#pragma unroll
    for (int t = 1; t < E; ++t) flat = flat && (d[t].x == d[0].x);
    // max over the workgroup of a per-thread value (bit patterns of non-negative numbers are monotone)
    int par = 0;
    auto wg_max = [&](T x) -> unsigned long long {
        x = (T)wave_total<ROWS * Q, true>(x);          // (DPP rows + v_readlane: see wave_total -- a butterfly of shuffles is six trips through the LDS pipe, every step)
        if ((tid & 63) == 0) wave_max[par][tid >> 6] = float_bits<T>(x);
        __syncthreads();
        unsigned long long mb = wave_max[par][0];
#pragma unroll
        for (int w = 1; w < NW; ++w) mb = wave_max[par][w] > mb ? wave_max[par][w] : mb;
        par ^= 1;                    // the next reduction writes the other set: no second barrier
        return mb;
    };
    const T half = (T)0.5;
    T pold[E];
    T phi[E];
    StepState<T> S;
    {   // first step size (k_absmax + k_step_control phase 0) and the first half rotation (k_time<TM_BEGIN>)
        T pmax = (T)0;
#pragma unroll
        for (int t = 0; t < E; ++t) {
            const T p = v[t].x * v[t].x + v[t].y * v[t].y;
            pold[t] = p;
            pmax = p > pmax ? p : pmax;
        }
        const unsigned long long mb = wg_max(pmax);
        T h;
        if (a.single_step) h = a.st->length;
        else h = a.st->phi_max / (a.st->abs_gamma * bits_float<T>(mb));
        h = h < a.st->length ? h : a.st->length;
        S.h = h; S.z = (T)0; S.steps = 0; S.done = !((T)0 < a.st->length);
        if (tid == 0) a.zlog[0] = (T)0;
        if (!S.done) {
            const T hh = S.h * half;
#pragma unroll
            for (int t = 0; t < E; ++t) {
                T ph = (T)0;
                ph += hh * (a.gamma * pold[t]);
                phi[t] = ph;
            }
            rotate_all<E>(v, phi);
        }
    }
    while (!S.done) {
        const T h = S.h;
        fft_line<T, N, E, -1, 1, RI>(v, lds, 0, j, idx, tw);
        {   // exp(D~ h)/N as k_freq<FM_FLY>
            T ph[E];
#pragma unroll
            for (int t = 0; t < E; ++t) { ph[t] = d[t].y * h; m[t] = d[t]; }
            const T e0 = exp_acc<T>(d[0].x * h);
            fly_factors<E>(m, ph, flat, e0, h, a.inv_n);
        }
#pragma unroll
        for (int t = 0; t < E; ++t) v[t] = cmul(v[t], m[t]);
        fft_line<T, N, E, +1, 1, RI>(v, lds, 0, j, idx, tw);
        T pnew[E];
        T pmax = (T)0;
#pragma unroll
        for (int t = 0; t < E; ++t) {
            const T p = v[t].x * v[t].x + v[t].y * v[t].y;
            pnew[t] = p;
            pmax = p > pmax ? p : pmax;
        }
        const StepState<T> Sn = step_advance<T>(a.st, S, wg_max(pmax));
        if (tid == 0) a.zlog[Sn.steps] = Sn.z;
        const T hh_prev = h * half, hh_next = Sn.h * half;
#pragma unroll
        for (int t = 0; t < E; ++t) {
            T ph = hh_prev * (a.gamma * pold[t]);
            if (!Sn.done) ph += hh_next * (a.gamma * pnew[t]);
            pold[t] = pnew[t];
            phi[t] = ph;
        }
        rotate_all<E>(v, phi);
        S = Sn;
    }
#pragma unroll
    for (int t = 0; t < E; ++t) Frow[j + t * Q] = v[t];
    if (tid == 0) a.st->cur[0] = S;
}

// ------------------------------------------------------------------------------ tables
// Factor tables of the inter-pass twiddles, tile by tile: out[(tile * R + r) * C + c] = W_N^(r mult n2(tile, c)), r < R
// (twA: R = N1/E, mult = 1;  twB: R = E, mult = N1/E)
template <typename T> __global__ void k_make_tw_tiles(cx<T>* out, int R, long long mult, int N2, int C, long long N, int u16, int Qf) {
    const long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= (long long)R * N2) return;
    const long long tile = o / ((long long)R * C), r = (o / C) % R, c = o % C;
    const long long n2 = u16 ? u16_col_of_pos(tile * C + 2 * (c & 7) + (c >> 3), Qf) : tile * C + c;
    const long long m = (r * mult % N) * n2 % N;
    double sn, cs;
    sincospi(-2.0 * (double)m / (double)N, &sn, &cs);
    out[o] = mk<T>((T)cs, (T)sn);
}
// W_N^(k1*n2) at [k1*N2 + n2]
template <typename T> __global__ void k_make_twN(cx<T>* tab, int N1, int N2, int C, int E, int u16, int Qf) {
    const long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long N = (long long)N1 * N2;
    if (o >= N) return;
    const long long k1 = o / N2, n2 = o % N2;
    const long long m = (k1 * n2) % N;
    double s, c;
    sincospi(-2.0 * (double)m / (double)N, &s, &c);
    tab[u16 ? time_tw_pos_u16(k1, n2, N1 / E, E, Qf, N2 / C) : time_tw_pos(k1, n2, N1 / E, C, E)] = mk<T>((T)c, (T)s);
}
// out[k1*N2 + freq_tab_pos(k2)] = f(src[k1 + N1*k2]); MODE 0: copy, 1: * inv_n, 2: exp(src*h) * inv_n
template <typename T, int MODE>
__global__ void k_make_freq_table(const cx<T>* __restrict__ src, cx<T>* __restrict__ out, int N1, int N2, int Q, T h, T inv_n) {
    const long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long N = (long long)N1 * N2;
    if (o >= N) return;
    const long long k1 = o / N2, k2 = o % N2;
    cx<T> d = src[k1 + (long long)N1 * k2];
    if (MODE == 1) { d.x *= inv_n; d.y *= inv_n; }
    if (MODE == 2) {
        // the products are taken in T exactly as the reference does (complex64 * float32), the
        // transcendental functions in double and rounded once: within 1 ulp of any libm's result
        const T xr = d.x * h, xi = d.y * h;
        const T e = (T)exp((double)xr);
        double s, c;
        sincos((double)xi, &s, &c);
        d.x = (e * (T)c) * inv_n;
        d.y = (e * (T)s) * inv_n;
    }
    out[k1 * N2 + freq_tab_pos(k2, Q)] = d;
}
// the operator's imaginary parts alone (FM_FLY_IM): out[k1*N2 + freq_phase_pos(k2)] = Im(src[k1 + N1*k2])
template <typename T>
__global__ void k_make_imag_table(const cx<T>* __restrict__ src, T* __restrict__ out, int N1, int N2, int Q) {
    const long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long N = (long long)N1 * N2;
    if (o >= N) return;
    const long long k1 = o / N2, k2 = o % N2;
    out[k1 * N2 + freq_phase_pos(k2, Q)] = src[k1 + (long long)N1 * k2].y;
}
// phase table of exp(D~ h) (FM_PHASE): out[k1*N2 + freq_phase_pos(k2)] = round(frac(Im(src[k1 + N1*k2]) * h / 2 pi) * 2^32),
// the product in T exactly as the reference forms it (complex64 * float32), the rest in double
template <typename T>
__global__ void k_make_phase_table(const cx<T>* __restrict__ src, unsigned* __restrict__ out, int N1, int N2, int Q, T h) {
    const long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long N = (long long)N1 * N2;
    if (o >= N) return;
    const long long k1 = o / N2, k2 = o % N2;
    const T xi = src[k1 + (long long)N1 * k2].y * h;
    if constexpr (sizeof(T) == 8) {
        // complex128 plans: the float64 fraction of a turn in [-0.5, 0.5], slots 2g / 2g+1 of a thread side by side (freq_tab_pos).  1 / (2 pi) as
        // a double-double, so that the fraction carries no more than its own final rounding
        const double c1 = 0.15915494309189535, c2 = -9.839338337591243e-18;
        const double hi = xi * c1;
        const double lo = fma(xi, c1, -hi) + xi * c2;
        const double f = hi - rint(hi);
        reinterpret_cast<double*>(out)[k1 * N2 + freq_tab_pos(k2, Q)] = f + lo;
        return;
    }
    double turns = (double)xi * 0.15915494309189533577;          // 1 / (2 pi)
    turns -= floor(turns);
    out[k1 * N2 + freq_phase_pos(k2, Q)] = (unsigned)(unsigned long long)llrint(turns * 4294967296.0);      // (2^32 wraps to 0)
}
// DM transfer function (reference devices.py:1025-1027): H_k = exp(1j * w_k^2 * D / 2), w_k = fftfreq(n, dt)[k] * 2 * pi,
// every product in float64 in the reference's order.  Writes H/N in the transposed order and, if
// `nat` != nullptr, H in natural order.
template <typename T>
__global__ void k_make_dm_table(cx<T>* __restrict__ perm, cx<T>* __restrict__ nat, int N1, int N2, int Q, double val, double D, T inv_n) {
    const long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long N = (long long)N1 * N2;
    if (o >= N) return;
    const long long k1 = o / N2, k2 = o % N2;
    const long long k = k1 + (long long)N1 * k2;
    const long long ks = k < (N + 1) / 2 ? k : k - N;          // numpy.fft.fftfreq integer grid
    const double w = ((double)ks * val) * 2.0 * 3.141592653589793;
    const double ph = ((w * w) * D) / 2.0;
    double s, c;
    sincos(ph, &s, &c);
    perm[k1 * N2 + freq_tab_pos(k2, Q)] = mk<T>((T)c * inv_n, (T)s * inv_n);
    if (nat != nullptr) nat[k] = mk<T>((T)c, (T)s);
}
// natural[k1 + N1*k2] = perm[k1*N2 + k2]   (debug: spectrum back to natural order)
template <typename T>
__global__ void k_unpermute(const cx<T>* __restrict__ perm, cx<T>* __restrict__ nat, int N1, int N2, int batch) {
    const long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long N = (long long)N1 * N2;
    if (o >= N * batch) return;
    const long long b = o / N, r = o % N;
    const long long k1 = r / N2, k2 = r % N2;
    nat[b * N + k1 + (long long)N1 * k2] = perm[o];
}

// max |A|^2 over the whole (natural-order) field -> st->maxbits
template <typename T> __global__ void k_absmax(const cx<T>* __restrict__ F, long long total, AdaptState<T>* st) {
    T pmax = (T)0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const cx<T> v = F[i];
        const T p = v.x * v.x + v.y * v.y;
        pmax = p > pmax ? p : pmax;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const T o = __shfl_xor(pmax, off);
        pmax = o > pmax ? o : pmax;
    }
    // one atomic per workgroup, spread over the slots (thousands of waves on one address serialise)
    __shared__ T wave_max[16];
    if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = pmax;
    __syncthreads();
    if (threadIdx.x == 0) {
        T m = wave_max[0];
        for (unsigned w = 1; w < (blockDim.x + 63) / 64; ++w) m = wave_max[w] > m ? wave_max[w] : m;
        atomicMax(&st->slots[1][blockIdx.x % kAdaptSlots], float_bits<T>(m));       // (the slots "before step 0")
    }
}

// Launched with one wavefront.  phase 0: the first step from k_absmax's slots (reference devices.py:1155-1161) -> cur[0];
// phase 1 (k_adapt_finish): the state after the last launched step `step - 1` -> cur[step & 1], for the host and for
// the first BEGIN of the next chunk.
template <typename T> __global__ void k_step_control(AdaptState<T>* st, T* zlog, int phase, int single_step, int step) {
    if (phase == 0) {
        const unsigned long long mb = slots_max<T>(st, 1);
        if (threadIdx.x < kAdaptSlots) {            // (both stores inside the guard: the kernel is launched with 64 threads, but nothing here may depend on that)
            st->slots[0][threadIdx.x] = 0ull;
            st->slots[1][threadIdx.x] = 0ull;
        }
        if (threadIdx.x != 0) return;
        T h;
        if (single_step) h = st->length;
        else h = st->phi_max / (st->abs_gamma * bits_float<T>(mb));
        h = h < st->length ? h : st->length;
        StepState<T> S;
        S.h = h;
        S.z = (T)0;
        S.steps = 0;
        S.done = !((T)0 < st->length);
        st->cur[0] = S;
        st->cur[1] = S;
        zlog[0] = (T)0;
        return;
    }
    const StepState<T> S = step_state<T>(st, step, true);
    if (threadIdx.x != 0) return;
    if (!st->cur[(step - 1) & 1].done) zlog[S.steps] = S.z;
    st->cur[step & 1] = S;
}

}  // namespace ssfm
