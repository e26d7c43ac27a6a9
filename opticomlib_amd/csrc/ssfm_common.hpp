// ssfm_common.hpp -- error reporting shared by the translation units of _ssfm_amd.so
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "ssfm_amd.h"

namespace ssfm {

inline thread_local char g_err[512] = "";

inline int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// ---- entry points of ssfm_host.hip that the library's other translation units use and the C ABI does NOT export (hidden visibility).  Round 5: the
// chirp-z step fragments -- round 4's ABI exported twelve of them, which only devices.py could sequence correctly; the exported surface is now
// ssfm_chirp_setup / _propagate / _propagate_c64 / _transfer / _fourier (csrc/chirpz.hip).
#define SSFM_INTERNAL __attribute__((visibility("hidden")))
// A chirp-z step's two ends folded into the first and the last column launch (TM_BEGIN / TM_END with ChirpIO, ssfm_kernels.hpp): the caller's field A, `n`
// complex128 per row, is read and written directly.  h_dev (nullable, DEVICE double): hh = *h_dev / 2 instead of `hh`; done_dev (nullable, DEVICE int): the
// first and the last launch do nothing when it is set; maxbits_dev (nullable, DEVICE 8 bytes): atomic maximum of the bit pattern of |A|^2 after the step.
struct ChirpStepIO {
    void* A;
    void* P;
    const void* chirp;
    int64_t n;
    double gamma, hh;
    const void* h_dev;
    const void* done_dev;
    void* maxbits_dev;
    int c64_line = 0;        // the plan's line holds complex64 values between the passes (plan_line_half_ok; a complex64 caller's run)
    int lean = 0;            // the table holds `n` entries only (nothing is read from `n` up: the padding is set to zero instead) and the rows go out on the plan's lanes
};
SSFM_INTERNAL void* plan_stream(ssfm_plan* plan);            // the plan's stream / field buffer WITHOUT marking the plan as externally ordered (ssfm_stream does)
SSFM_INTERNAL void* plan_field(ssfm_plan* plan);
// one chirp-z step in five launches: x <- ifft(fft(ifft(fft(x) H0) mul) H1) with the step's ends folded in (complex128 plans, plain layout); asynchronous
SSFM_INTERNAL int plan_chirp_step(ssfm_plan* plan, const void* mul_dev, const ChirpStepIO* io);
// a whole fixed-step run on the plan's line in four launches per step: the line holds A c (zero from `keep` up) on entry and on return; slots 0 / 1 hold the
// convolutions' tables; mul[which[s]] = exp(D~ h_s) / keep below `keep`, zero above.  SSFM_ERR_UNSUPPORTED: a plan in the 16-byte-unit layout
SSFM_INTERNAL int plan_chirp_line_run(ssfm_plan* plan, const void* const* mul, const unsigned char* which, const double* hs, int64_t nsteps, double gamma, int64_t keep, int half);
SSFM_INTERNAL int plan_line_half_ok(ssfm_plan* plan);        // 1: the plan has the passes that keep the line in complex64 between them (ssfm_kernels.hpp time_body, H)
// the one-launch engines: a workgroup per row on a line of <= 4096 points (the plan's precision), or the one-XCD engine of a complex64 line of 2^13 ... 2^17
// points; SSFM_ERR_UNSUPPORTED with the field as it came when the plan has no such engine or its workgroups did not meet
SSFM_INTERNAL int plan_chirp_small(ssfm_plan* plan, void* A, const void* chirp, const void* Dt, int64_t n, double gamma, const double* hs, int64_t nsteps);
SSFM_INTERNAL int plan_chirp_small_adapt(ssfm_plan* plan, void* A, const void* chirp, const void* Dt, int64_t n, double gamma, double length, double phi_max, int f32,
                                         int64_t max_steps, double* z_out, int64_t* steps_out);
SSFM_INTERNAL int plan_chirp_medium(ssfm_plan* plan, void* A, const void* chirp, const void* Dt, int64_t n, double gamma, const double* hs, int64_t nsteps);
SSFM_INTERNAL int plan_chirp_medium_adapt(ssfm_plan* plan, void* A, const void* chirp, const void* Dt, int64_t n, double gamma, double length, double phi_max,
                                          int64_t max_steps, double* z_out, int64_t* steps_out);
SSFM_INTERNAL int64_t plan_length(ssfm_plan* plan, int* batch, int* precision);
// A device buffer of at least `bytes` bytes owned by the plan (slot 0 ... 3; grows on demand, freed with the plan; contents undefined between calls; a growing
// call waits for the plan's stream): scratch for the driver loops of chirpz.hip (the exp(D~ h) tables, the step control block, the z log) without a
// hipMalloc / hipFree pair per call -- hipFree waits for the whole device and stalls the streams of every other plan
SSFM_INTERNAL int plan_workspace(ssfm_plan* plan, int slot, size_t bytes, void** out);

}  // namespace ssfm

#define HIP_TRY(expr)                                                                               \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess)                                                                       \
            return ssfm::fail(SSFM_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),  \
                              __FILE__, __LINE__);                                                  \
    } while (0)
