// chirpz.hip -- fields whose length N is NOT a power of two (the reference takes any N: numpy.fft).
//
// Bluestein's identity turns a length-N DFT into a circular convolution of length M = 2^m >= 2N - 1, which the
// power-of-two engine already does in three launches (ssfm_apply_table: x <- ifft(fft(x) H)):
//     c_n = exp(-i pi n^2 / N)
//     fft_N(x)_k  = c_k       sum_n (x_n c_n)        conj(c)_{k-n}
//     ifft_N(X)_n = conj(c_n) sum_k (X_k conj(c_k))  c_{n-k}  / N
// One split step (reference devices.py:1172-1181) becomes
//     chirp_pre   y_n = A_n exp(i gamma |A_n|^2 h/2) c_n, zero-padded to M     (also stores |A_n|^2: the stale N^)
//     apply_table (slot 0: fft_M of conj(c))
//     chirp_mid   y_k = z_k exp(D~_k h)            -- c_k conj(c_k) = 1: the two chirps between the transforms cancel
//     apply_table (slot 1: fft_M of c)
//     chirp_post  A_n = z_n conj(c_n) / N * exp(i gamma |A_n|^2_stale h/2),  max |A|^2 for the adaptive step
// All of it in complex128 on a complex128 plan of length M, whatever the caller's precision: the result then differs
// from the reference's complex64 arithmetic only by the reference's own rounding (stated tolerance), and the
// chirp's n^2 phase is exact.  These kernels run on the plan's stream, between caller-owned device arrays
// (batch x N, natural order) and the plan's field buffer (batch x M); nothing synchronises with the host.
#include <hip/hip_runtime.h>

#include <vector>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "ssfm_amd.h"
#include "ssfm_common.hpp"

using ssfm::fail;

namespace {

// Step control of an adaptive run driven from C (ssfm_chirp_propagate): the kernels of a step take its size from here and return at once when the
// run is over (steps are queued ahead of the host's knowledge of the end).  NULL in the call-per-kernel entry points.
struct ChirpCtl {
    double h;                  // size of the step being taken [km] (a float32 value in complex64 mode)
    double z;                  // position reached at its end
    int done;
    int steps;                 // steps taken so far
    unsigned long long maxbits;
};

// (half: the line is stored as complex64 -- ssfm_chirp_line_run for a complex64 caller, ssfm_kernels.hpp time_body H)
__global__ __launch_bounds__(256) void k_chirp_pre(const double2* __restrict__ A, double* __restrict__ P, const double2* __restrict__ chirp,
                                                   double2* __restrict__ F, long long n, long long M, int batch, double gamma, double hh, const ChirpCtl* __restrict__ ctl, int half = 0) {
    if (ctl) { if (ctl->done) return; hh = 0.5 * ctl->h; }
    const long long total = M * batch;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / M, m = i - r * M;
        double2 y = make_double2(0.0, 0.0);
        if (m < n) {
            double2 a = A[r * n + m];
            const double p = a.x * a.x + a.y * a.y;
            if (P) P[r * n + m] = p;
            if (gamma != 0.0) {
                double s, c;
                sincos(gamma * p * hh, &s, &c);
                a = make_double2(a.x * c - a.y * s, a.x * s + a.y * c);
            }
            const double2 w = chirp[m];
            y = make_double2(a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x);
        }
        if (half) reinterpret_cast<float2*>(F)[i] = make_float2((float)y.x, (float)y.y);
        else F[i] = y;
    }
}

// mode 0: multiply by exp(D~ h) (D~ in `tab`); mode 1: multiply by tab itself (a transfer function, e.g. DM's H)
__global__ __launch_bounds__(256) void k_chirp_mid(const double2* __restrict__ tab, double2* __restrict__ F, long long n, long long M, int batch, double h, int mode,
                                                   const ChirpCtl* __restrict__ ctl) {
    if (ctl) { if (ctl->done) return; h = ctl->h; }
    const long long total = M * batch;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long m = i % M;
        double2 y = make_double2(0.0, 0.0);
        if (m < n) {
            const double2 z = F[i];
            double2 e = tab[m];
            if (mode == 0) {
                double s, c;
                sincos(e.y * h, &s, &c);
                const double g = exp(e.x * h);
                e = make_double2(g * c, g * s);
            }
            y = make_double2(z.x * e.x - z.y * e.y, z.x * e.y + z.y * e.x);
        }
        F[i] = y;
    }
}

__global__ __launch_bounds__(256) void k_chirp_post(double2* __restrict__ A, const double* __restrict__ P, const double2* __restrict__ chirp,
                                                    const double2* __restrict__ F, long long n, long long M, int batch, double gamma, double hh,
                                                    double scale, unsigned long long* __restrict__ maxbits, const ChirpCtl* __restrict__ ctl, int half = 0) {
    if (ctl) { if (ctl->done) return; hh = 0.5 * ctl->h; }
    const long long total = n * batch;
    double pmax = 0.0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / n, m = i - r * n;
        double2 z;
        if (half) { const float2 q = reinterpret_cast<const float2*>(F)[r * M + m]; z = make_double2((double)q.x, (double)q.y); }
        else z = F[r * M + m];
        const double2 w = chirp[m];
        double2 a = make_double2((z.x * w.x + z.y * w.y) * scale, (z.y * w.x - z.x * w.y) * scale);     // z * conj(c) / N
        if (gamma != 0.0) {
            double s, c;
            sincos(gamma * P[i] * hh, &s, &c);
            a = make_double2(a.x * c - a.y * s, a.x * s + a.y * c);
        }
        A[i] = a;
        const double p = a.x * a.x + a.y * a.y;
        pmax = p > pmax ? p : pmax;
    }
    if (maxbits) {
        for (int o = 32; o > 0; o >>= 1) {
            const double other = __shfl_xor(pmax, o);
            pmax = other > pmax ? other : pmax;
        }
        if ((threadIdx.x & 63) == 0) atomicMax(maxbits, (unsigned long long)__double_as_longlong(pmax));   // non-negative doubles order like integers
    }
}

unsigned blocks_for(long long n) { return (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096); }

struct Target {
    double2* F;
    hipStream_t stream;
    long long M;
};
// Every chirp-z entry point works on the field buffer of a COMPLEX128 plan of exactly `M` points per row and at least `rows` rows: checked here, once for
// all of them (only ssfm_chirp_propagate checked; a complex64 plan or a mismatched plan_n / batch wrote past the end of the field).
int target_of(ssfm_plan* plan, long long n, long long M, int rows, Target* t) {
    if (!plan) return fail(SSFM_ERR_INVALID, "null plan");
    {
        int prec = 0, pb = 0;
        const int64_t pn = ssfm::plan_length(plan, &pb, &prec);
        if (prec != SSFM_C128 || pn != M || rows < 1 || pb < rows)
            return fail(SSFM_ERR_INVALID, "chirp-z: a complex128 plan of %lld points x >= %d rows is needed (this one: %s, %lld x %d)", M, rows,
                        prec == SSFM_C128 ? "complex128" : "complex64", (long long)pn, pb);
    }
    t->F = static_cast<double2*>(ssfm::plan_field(plan));              // (the internal accessors: taking them does not make the plan's one-launch runs synchronous)
    t->stream = static_cast<hipStream_t>(ssfm::plan_stream(plan));
    t->M = M;
    if (!t->F) return fail(SSFM_ERR_INVALID, "plan has no field buffer");
    if (n < 2 || M < 2 * n - 1) return fail(SSFM_ERR_INVALID, "chirp-z: plan length %lld is shorter than 2 * %lld - 1", M, n);
    return SSFM_OK;
}

}  // namespace

namespace {
int chirp_pre(ssfm_plan* plan, int64_t plan_n, int batch, const void* A, void* P, const void* chirp, int64_t n, double gamma, double hh) {
    Target t;
    if (int rc = target_of(plan, n, plan_n, batch, &t)) return rc;
    if (!A || !chirp) return fail(SSFM_ERR_INVALID, "ssfm_chirp_pre: NULL argument");
    hipLaunchKernelGGL(k_chirp_pre, dim3(blocks_for(t.M * batch)), dim3(256), 0, t.stream, (const double2*)A, (double*)P, (const double2*)chirp, t.F,
                       (long long)n, t.M, batch, gamma, hh, (const ChirpCtl*)nullptr);
    HIP_TRY(hipGetLastError());
    return SSFM_OK;
}

int chirp_mid(ssfm_plan* plan, int64_t plan_n, int batch, const void* tab, int64_t n, double h, int mode) {
    Target t;
    if (int rc = target_of(plan, n, plan_n, batch, &t)) return rc;
    if (!tab || mode < 0 || mode > 1) return fail(SSFM_ERR_INVALID, "ssfm_chirp_mid: bad argument");
    hipLaunchKernelGGL(k_chirp_mid, dim3(blocks_for(t.M * batch)), dim3(256), 0, t.stream, (const double2*)tab, t.F, (long long)n, t.M, batch, h, mode, (const ChirpCtl*)nullptr);
    HIP_TRY(hipGetLastError());
    return SSFM_OK;
}

int chirp_post(ssfm_plan* plan, int64_t plan_n, int batch, void* A, const void* P, const void* chirp, int64_t n, double gamma, double hh, void* maxbits_dev, double scale) {
    Target t;
    if (int rc = target_of(plan, n, plan_n, batch, &t)) return rc;
    if (!A || !chirp || (gamma != 0.0 && !P)) return fail(SSFM_ERR_INVALID, "ssfm_chirp_post: NULL argument");
    if (maxbits_dev) HIP_TRY(hipMemsetAsync(maxbits_dev, 0, sizeof(unsigned long long), t.stream));
    hipLaunchKernelGGL(k_chirp_post, dim3(blocks_for((long long)n * batch)), dim3(256), 0, t.stream, (double2*)A, (const double*)P, (const double2*)chirp,
                       (const double2*)t.F, (long long)n, t.M, batch, gamma, hh, scale, (unsigned long long*)maxbits_dev, (const ChirpCtl*)nullptr);
    HIP_TRY(hipGetLastError());
    return SSFM_OK;
}
}  // namespace

// x <- ifft_n(fft_n(x) * tab) (exponent = 0: DM's H, any transfer function) or * exp(tab) (exponent != 0) for every row of the DEVICE array A (batch x n
// complex128, in place) on a complex128 plan of plan_n >= 2n - 1 points that ssfm_chirp_setup has prepared for n; `tab`: n complex128 on the DEVICE.
// Asynchronous on the plan's stream.  (reference: DM for any length, devices.py:1019-1035 over numpy.fft)
extern "C" int ssfm_chirp_transfer(ssfm_plan* plan, int64_t plan_n, int batch, void* A, const void* chirp, const void* tab, int64_t n, int exponent) {
    if (int rc = chirp_pre(plan, plan_n, batch, A, nullptr, chirp, n, 0.0, 0.0)) return rc;
    if (int rc = ssfm_apply_table(plan, 0)) return rc;
    if (int rc = chirp_mid(plan, plan_n, batch, tab, n, exponent ? 1.0 : 0.0, exponent ? 0 : 1)) return rc;
    if (int rc = ssfm_apply_table(plan, 1)) return rc;
    return chirp_post(plan, plan_n, batch, A, nullptr, chirp, n, 0.0, 0.0, nullptr, 1.0 / (double)n);
}
// fft (inverse = 0, unscaled like numpy.fft.fft) or ifft (inverse != 0, with its 1/n) of every row of A (batch x n complex128, DEVICE, in place):
//   fft(x)_k = c_k sum_m (x_m c_m) conj(c)_{k-m};   ifft(X)_m = conj(c_m) / n sum_k (X_k conj(c_k)) c_{m-k}.     chirp = c, chirp_conj = conj(c) (ssfm_device_chirp).
// Asynchronous on the plan's stream.  (reference: signal('w') / signal('t'), typing.py:1421-1462)
extern "C" int ssfm_chirp_fourier(ssfm_plan* plan, int64_t plan_n, int batch, void* A, const void* chirp, const void* chirp_conj, int64_t n, int inverse) {
    if (inverse) {
        if (int rc = chirp_pre(plan, plan_n, batch, A, nullptr, chirp_conj, n, 0.0, 0.0)) return rc;
        if (int rc = ssfm_apply_table(plan, 1)) return rc;
        return chirp_post(plan, plan_n, batch, A, nullptr, chirp, n, 0.0, 0.0, nullptr, 1.0 / (double)n);         // * conj(c) / n
    }
    if (int rc = chirp_pre(plan, plan_n, batch, A, nullptr, chirp, n, 0.0, 0.0)) return rc;
    if (int rc = ssfm_apply_table(plan, 0)) return rc;
    return chirp_post(plan, plan_n, batch, A, nullptr, chirp_conj, n, 0.0, 0.0, nullptr, 1.0);                     // * c: fft carries no 1/n
}

// ------------------------------------------------------------------------------- a whole run from C
namespace {
// tab[m] = exp(D~[m] h) for m < n, 0 up to M: what k_chirp_mid multiplies with, as a table for the fused middle pass (ssfm_apply_tables_mul)
__global__ __launch_bounds__(256) void k_chirp_mktab(const double2* __restrict__ D, double2* __restrict__ tab, long long n, long long M, double h, const ChirpCtl* __restrict__ ctl,
                                                     double scale = 1.0) {
    if (ctl) { if (ctl->done) return; h = ctl->h; }
    for (long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += (long long)gridDim.x * blockDim.x) {
        double2 e = make_double2(0.0, 0.0);
        if (m < n) {
            const double2 d = D[m];
            double sn, cs;
            sincos(d.y * h, &sn, &cs);
            const double g = exp(d.x * h) * scale;
            e = make_double2(g * cs, g * sn);
        }
        tab[m] = e;
    }
}
// max |A|^2 of the input (the first step size)
__global__ __launch_bounds__(256) void k_chirp_absmax(const double2* __restrict__ A, long long total, ChirpCtl* __restrict__ ctl) {
    double pmax = 0.0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const double2 a = A[i];
        const double p = a.x * a.x + a.y * a.y;
        pmax = p > pmax ? p : pmax;
    }
    for (int o = 32; o > 0; o >>= 1) {
        const double other = __shfl_xor(pmax, o);
        pmax = other > pmax ? other : pmax;
    }
    if ((threadIdx.x & 63) == 0) atomicMax(&ctl->maxbits, (unsigned long long)__double_as_longlong(pmax));
}
// The step rule of devices._fiber_chirpz (reference devices.py:1172-1196) in the caller's step arithmetic RT (float in complex64 mode):
//   h = phi_max / (|gamma| max|A|^2), clamped to L - z;  z += h.   first = 1: the size of step 0 from the input's maximum.
template <typename RT>
__global__ void k_chirp_control(ChirpCtl* __restrict__ ctl, double* __restrict__ zlog, double phi_max_d, double abs_gamma_d, double length_d, int max_steps, int first) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (ctl->done) return;
    const RT phi_max = (RT)phi_max_d, abs_gamma = (RT)abs_gamma_d, L = (RT)length_d;
    RT z = (RT)ctl->z;
    if (!first) {
        const int steps = ctl->steps + 1;
        ctl->steps = steps;
        zlog[steps] = (double)z;
        if (!(z < L) || steps >= max_steps) { ctl->done = 1; return; }
    } else zlog[0] = 0.0;
    const RT amax = (RT)__longlong_as_double((long long)ctl->maxbits);
    RT h = phi_max / (abs_gamma * amax);
    const RT left = (RT)(L - z);
    h = h < left ? h : left;
    ctl->h = (double)h;
    ctl->z = (double)(RT)(z + h);
    ctl->maxbits = 0ull;
}
}  // namespace

// FIBER / DBP on a field of ANY length through the chirp-z identity, the whole run driven from here (devices._fiber_chirpz called one entry point
// per kernel and, in adaptive mode, waited for every step's maximum on the host: 33 / 78 us per step at n = 2032, where the step itself is 25).
//   hs != NULL: fixed step, `nsteps` sizes (HOST);  hs == NULL: adaptive, h = phi_max / (|gamma| max|A|^2) in float32 (f32 != 0) or float64 arithmetic.
//   z_out (HOST, nullable): z after every step, steps + 1 entries (capacity max_steps + 1);  steps_out: steps taken.  Synchronous.
extern "C" int ssfm_chirp_propagate(ssfm_plan* plan, int64_t plan_n, int batch, void* A, void* P, const void* chirp, const void* Dt, int64_t n, double gamma,
                                    const double* hs, int64_t nsteps, double length, double phi_max, int f32, int64_t max_steps, double* z_out, int64_t* steps_out) {
    Target t;
    if (int rc = target_of(plan, n, plan_n, batch, &t)) return rc;
    if (!A || !P || !chirp || !Dt) return fail(SSFM_ERR_INVALID, "ssfm_chirp_propagate: NULL argument");
    {
        int prec = 0, pb = 0;
        const int64_t pn = ssfm::plan_length(plan, &pb, &prec);
        if (prec != SSFM_C128 || pn != plan_n || pb != batch) return fail(SSFM_ERR_INVALID, "ssfm_chirp_propagate: a complex128 plan of %lld x %d is needed", (long long)plan_n, batch);
    }
    const unsigned gM = blocks_for(t.M * batch), gN = blocks_for((long long)n * batch);
    const double scale = 1.0 / (double)n;
    // A step is FIVE launches (seven in adaptive mode): the middle of it -- inverse pass of the first convolution, product with exp(D~ h), forward pass
    // of the second -- is one column launch with the factors from a table (kept while the step size repeats), and the step's two ends -- chirp
    // products, half nonlinear steps, zero padding, the maximum of |A|^2 -- run inside the first and the last column launch (ssfm_chirp_step).
    // (The nine- and seven-launch forms this grew out of in round 3 -- 33 / 25.6 against 22 us per step at n = 2032, profiles/r03_anyn_small.txt --
    // were removed in round 4 together with their environment switches.)
    double2* mtab = nullptr;
    if (int rc = ssfm::plan_workspace(plan, 0, sizeof(double2) * (size_t)t.M, reinterpret_cast<void**>(&mtab))) return rc;
    double mtab_h = 0.0;
    bool mtab_set = false;
    (void)gM; (void)gN; (void)scale;
    // (An adaptive step without the table launch -- the middle column launch forming exp(D~ h) itself from the control block's step size, six launches
    // instead of seven -- was measured at the end of round 4 and is not faster: 32.4 / 36.7 / 41.5 against 30.5 / 36.8 / 41.3 us per step at 3000 x 2 /
    // 15060 x 2 / 32752 x 2: sixteen float64 sincos + exp per thread weigh what the launch saved.)
    // Long lines (2^18 points and more): the rows on the plan's two lanes, a table of n entries instead of M (the middle pass sets the padding to zero itself), and for a
    // complex64 caller complex64 values between the float64 passes -- as the four-launch run below (round 6; ssfm_host.hip chirp_line_run).  SSFM_CHIRP_HALF=0: as before.
    const char* he_ = std::getenv("SSFM_CHIRP_HALF");
    const bool lean = t.M >= (1ll << 18) && !(he_ && he_[0] == '0');
    const bool c64_steps = lean && f32 && ssfm::plan_line_half_ok(plan);
    auto step = [&](double h, const ChirpCtl* ctl, unsigned long long* mb) -> int {
        if (ctl || !mtab_set || std::memcmp(&mtab_h, &h, sizeof(h)) != 0) {
            const long long entries = lean ? (long long)n : t.M;
            hipLaunchKernelGGL(k_chirp_mktab, dim3(blocks_for(entries)), dim3(256), 0, t.stream, (const double2*)Dt, mtab, (long long)n, entries, h, ctl);
            mtab_h = h; mtab_set = true;
        }
        ssfm::ChirpStepIO io;
        io.A = A; io.P = P; io.chirp = chirp; io.n = n; io.gamma = gamma; io.hh = 0.5 * h;
        io.h_dev = ctl ? &ctl->h : nullptr; io.done_dev = ctl ? &ctl->done : nullptr; io.maxbits_dev = mb;
        io.lean = lean ? 1 : 0; io.c64_line = c64_steps ? 1 : 0;
        return ssfm::plan_chirp_step(plan, mtab, &io);
    };
    std::vector<double> hs_used;
    if (hs) {
        if (nsteps < 0) return fail(SSFM_ERR_INVALID, "ssfm_chirp_propagate: nsteps=%lld", (long long)nsteps);
        // a step of length zero is the identity (the reference's loop does not run for length == 0, devices.py:1172): dropped here, so that both
        // engines below see the same schedule; anything else that is not a positive finite number is an error
        for (int64_t s = 0; s < nsteps; ++s) {
            if (hs[s] == 0.0) continue;
            if (!(hs[s] > 0) || !std::isfinite(hs[s])) return fail(SSFM_ERR_INVALID, "ssfm_chirp_propagate: step %lld is %g km (must be finite and >= 0)", (long long)s, hs[s]);
            hs_used.push_back(hs[s]);
        }
        const int64_t nsteps_given = nsteps;
        hs = hs_used.data();
        nsteps = (int64_t)hs_used.size();
        if (nsteps == 0) { if (steps_out) *steps_out = nsteps_given; return SSFM_OK; }
        // plans of up to 4096 samples (n <= 2048): the whole schedule in one launch, a workgroup per row (k_small_chirp).  SSFM_CHIRP_SMALL=0: off.
        const char* se = std::getenv("SSFM_CHIRP_SMALL");
        if (nsteps > 0 && t.M <= 4096 && !(se && std::atoi(se) == 0)) {
            const int rc = ssfm::plan_chirp_small(plan, A, chirp, Dt, n, gamma, hs, nsteps);
            if (rc == SSFM_OK) {
                HIP_TRY(hipStreamSynchronize(t.stream));
                if (steps_out) *steps_out = nsteps_given;
                return SSFM_OK;
            }
            if (rc != SSFM_ERR_UNSUPPORTED) return rc;
        }
        // A schedule of at most four step sizes (the reference's fixed-h runs have two): FOUR launches per step.  The chirp products on either side of
        // a step cancel against the neighbouring steps' (c conj(c) = 1, and the nonlinear rotation commutes with them), so they are taken once -- a
        // pointwise launch before and one after the run -- and the line holds A c in between (ssfm_chirp_line_run; round 4: 21.4 -> 17 us per step at
        // n = 3000 x 2, profiles/r04_chirp_medium.txt).  The |A|^2 of a step's start stays in the plan's own buffer: P is not used.
        {
            std::vector<double> distinct;
            std::vector<unsigned char> which((size_t)nsteps);
            for (int64_t s = 0; s < nsteps && distinct.size() <= 4; ++s) {
                size_t i = 0;
                while (i < distinct.size() && std::memcmp(&distinct[i], &hs[s], sizeof(double)) != 0) ++i;
                if (i == distinct.size()) distinct.push_back(hs[s]);
                which[(size_t)s] = (unsigned char)i;
            }
            if (distinct.size() <= 4) {
                double2* tabs = nullptr;
                if (int rc = ssfm::plan_workspace(plan, 3, sizeof(double2) * (size_t)t.M * distinct.size(), reinterpret_cast<void**>(&tabs))) return rc;       // (a slot of its own: `mtab` above stays valid for the step loop below)
                const void* mulp[4] = {nullptr, nullptr, nullptr, nullptr};
                for (size_t i = 0; i < distinct.size(); ++i) {
                    hipLaunchKernelGGL(k_chirp_mktab, dim3(blocks_for(t.M)), dim3(256), 0, t.stream, (const double2*)Dt, tabs + i * (size_t)t.M, (long long)n, t.M, distinct[i],
                                       (const ChirpCtl*)nullptr, 1.0 / (double)n);
                    mulp[i] = tabs + i * (size_t)t.M;
                }
                // A complex64 caller's run (f32) on a line of 2^18 points and more: the line holds complex64 values BETWEEN the passes, every pass computes in
                // float64 (ssfm_kernels.hpp time_body, H) -- half the bytes of the field per pass, one rounding to complex64 per pass.  SSFM_CHIRP_HALF=0: never.
                const char* he = std::getenv("SSFM_CHIRP_HALF");
                const int half = (f32 && t.M >= (1ll << 18) && !(he && he[0] == '0') && ssfm::plan_line_half_ok(plan)) ? 1 : 0;
                hipLaunchKernelGGL(k_chirp_pre, dim3(blocks_for(t.M * batch)), dim3(256), 0, t.stream, (const double2*)A, (double*)nullptr, (const double2*)chirp, t.F,
                                   (long long)n, t.M, batch, 0.0, 0.0, (const ChirpCtl*)nullptr, half);
                HIP_TRY(hipGetLastError());
                const int rc = ssfm::plan_chirp_line_run(plan, mulp, which.data(), hs, nsteps, gamma, n, half);
                if (rc == SSFM_OK) {
                    hipLaunchKernelGGL(k_chirp_post, dim3(blocks_for((long long)n * batch)), dim3(256), 0, t.stream, (double2*)A, (const double*)nullptr, (const double2*)chirp,
                                       (const double2*)t.F, (long long)n, t.M, batch, 0.0, 0.0, 1.0, (unsigned long long*)nullptr, (const ChirpCtl*)nullptr, half);
                    HIP_TRY(hipGetLastError());
                    HIP_TRY(hipStreamSynchronize(t.stream));
                    if (steps_out) *steps_out = nsteps_given;
                    return SSFM_OK;
                }
                if (rc != SSFM_ERR_UNSUPPORTED) return rc;             // (a plan in the unit layout: the five-launch step below)
            }
        }
        for (int64_t s = 0; s < nsteps; ++s)
            if (int rc = step(hs[s], nullptr, nullptr)) return rc;
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(t.stream));
        if (steps_out) *steps_out = nsteps_given;
        return SSFM_OK;
    }
    if (max_steps < 1 || max_steps > (1ll << 30)) return fail(SSFM_ERR_INVALID, "ssfm_chirp_propagate: max_steps=%lld", (long long)max_steps);
    {   // plans of up to 4096 samples: the whole adaptive run in one launch (k_small_chirp_adapt)
        const char* se = std::getenv("SSFM_CHIRP_SMALL");
        if (t.M <= 4096 && gamma != 0.0 && !(se && std::atoi(se) == 0)) {
            const int rc = ssfm::plan_chirp_small_adapt(plan, A, chirp, Dt, n, gamma, length, phi_max, f32, max_steps, z_out, steps_out);
            if (rc != SSFM_ERR_UNSUPPORTED) return rc;
        }
    }
    ChirpCtl* ctl = nullptr;
    double* zlog = nullptr;
    if (int wrc = ssfm::plan_workspace(plan, 1, sizeof(ChirpCtl), reinterpret_cast<void**>(&ctl))) return wrc;
    if (int wrc = ssfm::plan_workspace(plan, 2, sizeof(double) * (size_t)(max_steps + 1), reinterpret_cast<void**>(&zlog))) return wrc;
    int rc = SSFM_OK;
    ChirpCtl now;
    std::memset(&now, 0, sizeof(now));
    auto hip_ok = [&](hipError_t e, const char* what) { if (e != hipSuccess && rc == SSFM_OK) rc = fail(SSFM_ERR_HIP, "ssfm_chirp_propagate: %s failed: %s", what, hipGetErrorString(e)); return e == hipSuccess; };
    const double ag = gamma < 0 ? -gamma : gamma;
    do {
        if (!hip_ok(hipMemsetAsync(ctl, 0, sizeof(ChirpCtl), t.stream), "hipMemsetAsync")) break;
        hipLaunchKernelGGL(k_chirp_absmax, dim3(gN), dim3(256), 0, t.stream, (const double2*)A, (long long)n * batch, ctl);
        if (f32) hipLaunchKernelGGL(k_chirp_control<float>, dim3(1), dim3(64), 0, t.stream, ctl, zlog, phi_max, ag, length, (int)max_steps, 1);
        else hipLaunchKernelGGL(k_chirp_control<double>, dim3(1), dim3(64), 0, t.stream, ctl, zlog, phi_max, ag, length, (int)max_steps, 1);
        if (!hip_ok(hipMemcpyAsync(&now, ctl, sizeof(now), hipMemcpyDeviceToHost, t.stream), "hipMemcpyAsync") || !hip_ok(hipStreamSynchronize(t.stream), "hipStreamSynchronize")) break;
        while (!now.done) {
            // about (L - z) / h steps remain (the step only shrinks towards the clamp at L): queue most of them, then look
            const double remain = now.h > 0 ? (length - (now.z - now.h)) / now.h : 1.0;
            int chunk = remain > 1e6 ? 128 : (int)(0.75 * remain) + 1;
            chunk = chunk < 2 ? 2 : (chunk > 128 ? 128 : chunk);
            for (int i = 0; i < chunk && rc == SSFM_OK; ++i) {
                rc = step(0.0, ctl, &ctl->maxbits);
                if (f32) hipLaunchKernelGGL(k_chirp_control<float>, dim3(1), dim3(64), 0, t.stream, ctl, zlog, phi_max, ag, length, (int)max_steps, 0);
                else hipLaunchKernelGGL(k_chirp_control<double>, dim3(1), dim3(64), 0, t.stream, ctl, zlog, phi_max, ag, length, (int)max_steps, 0);
            }
            if (rc != SSFM_OK) break;
            if (!hip_ok(hipGetLastError(), "a launch") || !hip_ok(hipMemcpyAsync(&now, ctl, sizeof(now), hipMemcpyDeviceToHost, t.stream), "hipMemcpyAsync") ||
                !hip_ok(hipStreamSynchronize(t.stream), "hipStreamSynchronize")) break;
        }
        if (rc != SSFM_OK) break;
        if (z_out) hip_ok(hipMemcpy(z_out, zlog, sizeof(double) * (size_t)(now.steps + 1), hipMemcpyDeviceToHost), "hipMemcpy");
        if (steps_out) *steps_out = now.steps;
    } while (false);
    return rc;
}

// The one-launch engines of complex64 callers behind one entry point: a run of a field of n samples per row (DEVICE, complex64, natural order, advanced in
// place) on a COMPLEX64 plan whose length is the line's, M = 2^k >= 2n - 1 -- a workgroup per row for M <= 4096 (k_small_chirp[_adapt]), the one-XCD engine for
// M = 2^13 ... 2^17 with at most 2^17 points in all rows (k_medium_chirp[_adapt]).  chirp (n complex64, DEVICE; ssfm_device_chirp rounded), Dt = D~ (n
// complex64, DEVICE, natural frequency order).  hs != NULL: fixed step, `nsteps` sizes (HOST; the medium engine: at most four distinct ones).  hs == NULL:
// the adaptive rule of devices.py:1172-1196 in float32 arithmetic over `length`, at most max_steps steps; z_out (HOST, nullable, max_steps + 1 entries)
// receives z after every step, *steps_out the steps taken.  Synchronous for the medium engine and adaptive runs; a fixed-step small run is asynchronous on the
// plan's stream.  SSFM_ERR_UNSUPPORTED with A as it was: the plan has no such engine, or its workgroups did not meet within their patience (the caller takes
// the complex128 line: ssfm_chirp_propagate).
extern "C" int ssfm_chirp_propagate_c64(ssfm_plan* plan, void* A, const void* chirp, const void* Dt, int64_t n, double gamma, const double* hs, int64_t nsteps,
                                        double length, double phi_max, int64_t max_steps, double* z_out, int64_t* steps_out) {
    int prec = 0, batch = 0;
    const int64_t M = ssfm::plan_length(plan, &batch, &prec);
    if (M <= 0) return fail(SSFM_ERR_INVALID, "null plan");
    if (prec != SSFM_C64) return fail(SSFM_ERR_UNSUPPORTED, "ssfm_chirp_propagate_c64: a complex64 plan is needed");
    if (hs) {
        const int rc = M <= 4096 ? ssfm::plan_chirp_small(plan, A, chirp, Dt, n, gamma, hs, nsteps) : ssfm::plan_chirp_medium(plan, A, chirp, Dt, n, gamma, hs, nsteps);
        if (rc == SSFM_OK && steps_out) *steps_out = nsteps;
        return rc;
    }
    if (M <= 4096) return ssfm::plan_chirp_small_adapt(plan, A, chirp, Dt, n, gamma, length, phi_max, 1, max_steps, z_out, steps_out);
    return ssfm::plan_chirp_medium_adapt(plan, A, chirp, Dt, n, gamma, length, phi_max, max_steps, z_out, steps_out);
}

// --------------------------------------------------------------------------------- pulse shaping (DAC)
// The reference's upfir (utils.py:1949-1981): bits zero-stuffed to `up` samples per bit (sample at up/2), linear
// convolution with the pulse h, 'same' part.  On the plan: ssfm_load_padded writes h (zero-padded) into the field,
// ssfm_table_from_field turns the field into its transfer function fft(h) (slot), ssfm_load_symbols writes the
// zero-stuffed symbol amplitudes (the bits as 0.0 / 1.0), ssfm_apply_table convolves; the caller copies the 'same' window out of the field buffer.
namespace {

__global__ __launch_bounds__(256) void k_load_padded(const double* __restrict__ src, int src_complex, long long n_src, double2* __restrict__ F, long long M) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += (long long)gridDim.x * blockDim.x) {
        double2 v = make_double2(0.0, 0.0);
        if (i < n_src) v = src_complex ? make_double2(src[2 * i], src[2 * i + 1]) : make_double2(src[i], 0.0);
        F[i] = v;
    }
}

__global__ __launch_bounds__(256) void k_load_symbols(const double* __restrict__ sym, long long nsym, int up, double2* __restrict__ F, long long M) {
    const int at = up / 2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / up;
        F[i] = make_double2((b < nsym && i - b * up == at) ? sym[b] : 0.0, 0.0);
    }
}


// The DAC's built-in pulses generated where they are used (reference utils.py:1791-1947: nrz_pulse, gauss_pulse,
// rcos_pulse over t = linspace(-span/2, span/2, span*sps + 1)); the pulse is as long as the signal, so building it
// with NumPy and uploading it cost more than the whole convolution.  Every expression keeps the reference's order of
// operations (no fused multiply-add: the grid t_i = i*step + start is then NumPy's linspace bit for bit, and the
// values differ from NumPy's only by the last-bit differences of exp / sin / cos).
struct PulseSpec {
    int kind;            // 0 nrz, 1 gaussian, 2 raised cosine ('normal'), 3 root raised cosine ('sqrt'), 4 sinc (beta = 0)
    int pow2m;           // gaussian: the exponent 2 m
    long long npts;
    double start, step, stop;
    double a, b, c, d, e, f, g;      // per kind, see k_load_pulse
};

// One rounding per operation: HIP's __dmul_rn / __dadd_rn are plain operators the compiler may still fuse into an
// fma, so the products and sums that must round like NumPy's are written under an explicit no-contraction pragma.
__device__ __forceinline__ double mul_r(double a, double b) {
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ double add_r(double a, double b) {
#pragma clang fp contract(off)
    return a + b;
}
__device__ __forceinline__ double sub_r(double a, double b) {
#pragma clang fp contract(off)
    return a - b;
}

__device__ __forceinline__ double2 cmul_plain(double2 x, double2 y) {
    return make_double2(sub_r(mul_r(x.x, y.x), mul_r(x.y, y.y)), add_r(mul_r(x.x, y.y), mul_r(x.y, y.x)));
}

__device__ __forceinline__ double np_sinc(double t) {            // numpy.sinc: sin(y) / y, y = pi * (t == 0 ? 1e-20 : t)
    const double y = mul_r(3.141592653589793, t == 0.0 ? 1.0e-20 : t);
    return sin(y) / y;
}

__global__ __launch_bounds__(256) void k_load_pulse(PulseSpec ps, double2* __restrict__ F, long long M) {
    const double pi = 3.141592653589793;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += (long long)gridDim.x * blockDim.x) {
        double2 v = make_double2(0.0, 0.0);
        if (i < ps.npts) {
            const double t = (i == ps.npts - 1 && ps.npts > 1) ? ps.stop : add_r(mul_r((double)i, ps.step), ps.start);
            if (ps.kind == 0) {                                   // a = -T/2, b = T/2
                v.x = (t >= ps.a && t < ps.b) ? 1.0 : 0.0;
            } else if (ps.kind == 1) {                            // (a, b) = alpha (1 + j c): exp(-((a + j b) t)^(2m))
                const double2 z = make_double2(mul_r(ps.a, t), mul_r(ps.b, t));
                double2 acc = make_double2(1.0, 0.0), p = z;      // NumPy's integer complex power: binary powering
                if (ps.pow2m == 2) acc = cmul_plain(z, z);
                else if (z.x == 0.0 && z.y == 0.0) acc = make_double2(0.0, 0.0);
                else for (int mask = 1;;) {
                    if (ps.pow2m & mask) acc = cmul_plain(acc, p);
                    mask <<= 1;
                    if (ps.pow2m < mask) break;
                    p = cmul_plain(p, p);
                }
                const double ex = exp(-acc.x);
                double sn, cs;
                sincos(-acc.y, &sn, &cs);
                v = make_double2(ex * cs, ex * sn);
            } else if (ps.kind == 2) {                            // a = 2 beta, b = pi beta, c = value where den ~ 0
                const double u = mul_r(ps.a, t);
                const double den = sub_r(1.0, mul_r(u, u));
                if (fabs(den) < 1e-8) v.x = ps.c;
                else v.x = mul_r(np_sinc(t), cos(mul_r(ps.b, t))) / den;
            } else if (ps.kind == 3) {                            // a = beta, b = 4 beta, c = 1 - beta, d = 1 + beta, e = 1/(4 beta), f = value at 0, g = value at 1/(4 beta)
                const double ta = fabs(t);
                if (ta < 1e-8) v.x = ps.f;
                else if (fabs(sub_r(ta, ps.e)) < 1e-8) v.x = ps.g;
                else {
                    const double pt = mul_r(pi, t), bt = mul_r(ps.b, t);
                    const double num = add_r(sin(mul_r(pt, ps.c)), mul_r(bt, cos(mul_r(pt, ps.d))));
                    v.x = num / mul_r(pt, sub_r(1.0, mul_r(bt, bt)));
                }
            } else {
                v.x = np_sinc(t);
            }
        }
        F[i] = v;
    }
}

}  // namespace

extern "C" int ssfm_load_padded(ssfm_plan* plan, int64_t plan_n, const void* src_dev, int src_complex, int64_t n_src) {
    Target t;
    if (int rc = target_of(plan, 2, plan_n, 1, &t)) return rc;      // (no length relation to check here)
    if (!src_dev || n_src < 1 || n_src > plan_n) return fail(SSFM_ERR_INVALID, "ssfm_load_padded: %lld source samples for a plan of %lld", (long long)n_src, (long long)plan_n);
    hipLaunchKernelGGL(k_load_padded, dim3(blocks_for(t.M)), dim3(256), 0, t.stream, (const double*)src_dev, src_complex, (long long)n_src, t.F, t.M);
    HIP_TRY(hipGetLastError());
    return SSFM_OK;
}

namespace ssfm { SSFM_INTERNAL int plan_load_bits(ssfm_plan* plan, int64_t plan_n, const void* bits_dev, int64_t nbits, int up); }      // (prbs.hip)
extern "C" int ssfm_load_symbols(ssfm_plan* plan, int64_t plan_n, const void* src_dev, int src_kind, int64_t nsym, int up) {
    if (src_kind == 1) return ssfm::plan_load_bits(plan, plan_n, src_dev, nsym, up);
    if (src_kind != 0) return fail(SSFM_ERR_INVALID, "ssfm_load_symbols: src_kind %d", src_kind);
    const double* sym_dev = static_cast<const double*>(src_dev);
    Target t;
    if (int rc = target_of(plan, 2, plan_n, 1, &t)) return rc;      // (no length relation to check here)
    if (!sym_dev || nsym < 1 || up < 1 || nsym * up > plan_n) return fail(SSFM_ERR_INVALID, "ssfm_load_symbols: %lld symbols x %d samples for a plan of %lld", (long long)nsym, up, (long long)plan_n);
    hipLaunchKernelGGL(k_load_symbols, dim3(blocks_for(t.M)), dim3(256), 0, t.stream, sym_dev, (long long)nsym, up, t.F, t.M);
    HIP_TRY(hipGetLastError());
    return SSFM_OK;
}

extern "C" int ssfm_load_pulse(ssfm_plan* plan, int64_t plan_n, int kind, int64_t npts, double start, double step, double stop, int pow2m, const double* params) {
    Target t;
    if (int rc = target_of(plan, 2, plan_n, 1, &t)) return rc;      // (no length relation to check here)
    if (kind < 0 || kind > 4 || npts < 1 || npts > plan_n || !params) return fail(SSFM_ERR_INVALID, "ssfm_load_pulse: kind %d, %lld points for a plan of %lld", kind, (long long)npts, (long long)plan_n);
    if (kind == 1 && (pow2m < 2 || pow2m > 98 || (pow2m & 1))) return fail(SSFM_ERR_INVALID, "ssfm_load_pulse: gaussian order 2m = %d outside 2 ... 98", pow2m);
    PulseSpec ps{kind, pow2m, (long long)npts, start, step, stop, params[0], params[1], params[2], params[3], params[4], params[5], params[6]};
    hipLaunchKernelGGL(k_load_pulse, dim3(blocks_for(t.M)), dim3(256), 0, t.stream, ps, t.F, t.M);
    HIP_TRY(hipGetLastError());
    return SSFM_OK;
}

// ---------------------------------------------------------------------------------- the chirp itself
// c_m = exp(-i pi m^2 / n): the phase is reduced exactly in integers (m^2 mod 2n) and taken with sincospi, so
// the chirp carries no argument-rounding error at all.  k_chirp: out[m] = c_m or conj(c_m), m < n.  k_chirp_kernel:
// the convolution kernel of Bluestein's identity in the plan's field (row 0, length M): v[m] = v[M - m] = conj(c_m)
// (which = 0: forward transform) or c_m (which = 1: inverse), zero elsewhere; ssfm_table_from_field turns it into
// the resident transfer function of a slot -- no host transform, no upload.
namespace {

__device__ __forceinline__ double2 chirp_value(long long m, long long n, int conj) {
    const long long r = (m * m) % (2 * n);
    double s, c;
    sincospi(-(double)r / (double)n, &s, &c);
    return make_double2(c, conj ? -s : s);
}

__global__ __launch_bounds__(256) void k_chirp(double2* __restrict__ out, long long n, int conj) {
    for (long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x; m < n; m += (long long)gridDim.x * blockDim.x) out[m] = chirp_value(m, n, conj);
}

__global__ __launch_bounds__(256) void k_chirp_kernel(double2* __restrict__ F, long long n, long long M, int which) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += (long long)gridDim.x * blockDim.x) {
        const long long m = i < n ? i : (M - i < n ? M - i : -1);
        F[i] = m < 0 ? make_double2(0.0, 0.0) : chirp_value(m, n, which == 0);
    }
}

}  // namespace

extern "C" int ssfm_device_chirp(int device, void* out_dev, int64_t n, int conj) {
    if (!out_dev || n < 2 || n > (1ll << 30)) return fail(SSFM_ERR_INVALID, "ssfm_device_chirp: bad argument");
    HIP_TRY(hipSetDevice(device));
    hipLaunchKernelGGL(k_chirp, dim3(blocks_for(n)), dim3(256), 0, 0, (double2*)out_dev, (long long)n, conj);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    return SSFM_OK;
}

// Both convolution kernels of Bluestein's identity for fields of n samples, generated in the plan's field and transformed there into the resident transfer
// functions of slots 0 (forward transform) and 1 (inverse): what every other ssfm_chirp_* call on a complex128 plan of plan_n >= 2n - 1 points relies on.
// Neither a host transform nor an upload.  Asynchronous on the plan's stream; the plan's field is consumed.
extern "C" int ssfm_chirp_setup(ssfm_plan* plan, int64_t plan_n, int64_t n) {
    Target t;
    if (int rc = target_of(plan, n, plan_n, 1, &t)) return rc;
    if (n > (1ll << 30)) return fail(SSFM_ERR_INVALID, "ssfm_chirp_setup: bad argument");
    for (int which = 0; which < 2; ++which) {
        hipLaunchKernelGGL(k_chirp_kernel, dim3(blocks_for(t.M)), dim3(256), 0, t.stream, t.F, (long long)n, t.M, which);
        HIP_TRY(hipGetLastError());
        if (int rc = ssfm_table_from_field(plan, which)) return rc;
    }
    return SSFM_OK;
}
