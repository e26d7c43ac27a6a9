/*
 * ssfm_amd.h -- C ABI of the MI355X (gfx950) split-step Fourier fibre propagator.
 *
 * This is the drop-in boundary for ONE path of armando-palacio/opticomlib: the body of
 * `opticomlib.devices.FIBER` (reference opticomlib/devices.py:1113-1206), its sign-flipped
 * twin `DBP` (devices.py:1280-1283) and the single linear step `DM` (devices.py:1019-1035).
 * The reference has no FFI of its own: the seam is the Python function call, and the only
 * accelerator hook is the CuPy array-API subset used at devices.py:1128-1134.  What a
 * binding for this path needs is therefore exactly:
 *
 *   reference line(s)                              entry point here
 *   ---------------------------------------------  -------------------------------------------
 *   devices.py:1137-1145 (coefficients, D~)        ssfm_set_linear_operator (host computes D~ in
 *                                                  the reference's float32 expression order; the
 *                                                  library never re-derives it)
 *   devices.py:1147 (A = complex64(signal+noise))  ssfm_set_field
 *   devices.py:1155-1161,1172-1196 fixed h         ssfm_propagate_fixed (host supplies the float32
 *                                                  step schedule z/h produce, devices.py:1173,1196)
 *   devices.py:1155-1156,1193-1196 h=None          ssfm_propagate_adaptive (step control on device)
 *   devices.py:1150-1152,1184-1186 return_steps    `snapshots` argument of both propagate calls
 *   devices.py:1204 (.get() / wrap)                ssfm_get_field
 *   devices.py:1027-1029 (DM: ifft(fft(x)*H))      ssfm_apply_transfer / ssfm_apply_dispersion (H generated on the device)
 *   devices.py:1363-1368, :814-823 (LPF / BPF:     ssfm_sosfiltfilt (`on_device` flag: host or device arrays; ssfm_sosfiltfilt_last
 *     scipy.signal.sosfiltfilt)                    says which form the last call took)
 *   devices.py:1512-1515 (PD: r * |x|^2, pol sum)  ssfm_square_law (`on_device` flag)
 *   devices.py:1521-1549, :930-936 (PD / EDFA      ssfm_device_randn (rng = "device"), ssfm_device_reduce (mean / min / max), ssfm_device_sum3,
 *                                                  ssfm_device_scale_add
 *     noise currents, gain and ASE loading)
 *   numpy.fft of any N (devices.py:1178-1180)      ssfm_chirp_setup, ssfm_chirp_propagate / _propagate_c64 (FIBER / DBP), ssfm_chirp_transfer
 *                                                  (DM), ssfm_chirp_fourier (signal('w') / ('t')) on a power-of-two plan
 *   utils.py:1791-1981 (DAC: pulses, upfir)        ssfm_load_pulse / _load_padded / _load_symbols, ssfm_table_from_field,
 *                                                  ssfm_apply_table, ssfm_device_axpb (real part: `is_complex` flag)
 *   devices.py:480-510 (LASER), :762-778 (MZM)     ssfm_laser (+ ssfm_device_cumsum, ssfm_device_reduce), ssfm_mzm
 *   (none: NumPy arrays are the reference's only   ssfm_device_alloc / _free / _copy / _convert / _add
 *     data format)                                 -- device-resident signals between calls
 *
 * Conventions: plain C types, caller-owned buffers, every call returns an int status
 * (0 = SSFM_OK), no exceptions cross the ABI, no global mutable state except the
 * thread-local text behind ssfm_last_error().  One plan owns one HIP stream on one
 * device, so N plans drive N GPUs from one process or one plan per rank drives one GPU.
 *
 * Field layout: `batch` rows of `n` complex samples, row-major, interleaved (re, im);
 * float32 pairs for SSFM_C64, float64 pairs for SSFM_C128.  A dual-polarisation
 * optical_signal is 2 rows; F independent fields are 2F rows (rows never interact
 * except through the shared adaptive step size).
 */
#ifndef SSFM_AMD_H
#define SSFM_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SSFM_ABI_VERSION 3

/* The library is built with -fvisibility=hidden: the entry points declared here are its ONLY dynamic symbols (nm -D: 59 + the runtime's). */
#ifndef SSFM_API
#define SSFM_API __attribute__((visibility("default")))
#endif

enum ssfm_status {
    SSFM_OK = 0,
    SSFM_ERR_INVALID = 1,      /* bad argument (message in ssfm_last_error) */
    SSFM_ERR_UNSUPPORTED = 2,  /* n not a supported power of two, unknown precision */
    SSFM_ERR_HIP = 3,          /* a HIP runtime call failed */
    SSFM_ERR_NO_DEVICE = 4,    /* no usable gfx950 device */
    SSFM_ERR_STATE = 5         /* call order violated (e.g. propagate before set_linear_operator) */
};

enum ssfm_precision { SSFM_C64 = 0, SSFM_C128 = 1 };
#define SSFM_F64_REAL 2   /* ssfm_device_convert only: a real float64 source (a CW laser, a drive voltage) */

typedef struct ssfm_plan ssfm_plan;

/* Library / device discovery.  ssfm_device_count never initialises a HIP context beyond
 * hipGetDeviceCount. */
SSFM_API int ssfm_abi_version(void);
SSFM_API int ssfm_device_count(int* count);
SSFM_API const char* ssfm_last_error(void);

/* Smallest / largest supported log2(n) for a precision (currently 8..24).  Rows of more than 2^22 samples run as "split plans" (round 6: R = 2 ... 16
 * sub-sequences of 2^20 samples through the same kernels plus a pointwise launch across them, csrc/ssfm_split.hpp): ssfm_set_field / _get_field /
 * _field_device_ptr, ssfm_set_linear_operator, ssfm_propagate_fixed (with `snapshots`), ssfm_propagate_adaptive / ssfm_adaptive_*, ssfm_apply_transfer and
 * ssfm_apply_dispersion work as for any plan; the entry points that treat the plan's field buffer as ONE line (ssfm_transfer_table / _apply_table /
 * _table_from_field, the ssfm_chirp_* and ssfm_load_* calls, ssfm_propagate_fixed_capture) return SSFM_ERR_UNSUPPORTED for them. */
SSFM_API int ssfm_supported_log2n(int precision, int* lo, int* hi);

/* Allocate every device buffer once: field, stale |A|^2, operator tables, twiddles. */
SSFM_API int ssfm_plan_create(ssfm_plan** out, int device, int64_t n, int batch, int precision);
SSFM_API int ssfm_plan_destroy(ssfm_plan* plan);

/* D~(w) [1/km], complex (precision's element type), length n, natural FFT order, HOST memory.
 * Shared by all rows.  (reference devices.py:1145) */
SSFM_API int ssfm_set_linear_operator(ssfm_plan* plan, const void* dtilde_host);

/* Copy a (batch, n) complex field in / out of the plan.  `is_device` != 0: the pointer is
 * device memory on the plan's device. Both calls are ordered on the plan's stream;
 * ssfm_get_field returns after the copy has completed. */
SSFM_API int ssfm_set_field(ssfm_plan* plan, const void* src, int is_device);
SSFM_API int ssfm_get_field(ssfm_plan* plan, void* dst, int is_device);
/* Device address of the plan's resident field (batch*n complex), natural time order
 * between propagate calls.
 * WHEN THE FIELD IS VALID.  Runs are asynchronous: the result is in the field buffer once ssfm_synchronize / ssfm_get_field has
 * returned SSFM_OK.  A caller that orders its OWN work behind a run on the plan's stream (ssfm_stream) instead may do so: from the
 * first call of ssfm_field_device_ptr or ssfm_stream on a plan, ssfm_propagate_fixed resolves the one kind of run whose success is only
 * known at its end -- the one-launch engine of plans of 2^12 ... 2^17 samples, whose workgroups meet inside the launch and give up
 * when the GPU does not run them side by side; the run is then repeated on the launch-per-pass engine -- BEFORE it returns (for such
 * plans the call is then synchronous).  Every other engine either needs no fallback or resolves it inside the call already.
 * ssfm_last_run_info tells which engine a run took and whether it fell back. */
SSFM_API void* ssfm_field_device_ptr(ssfm_plan* plan);

/* Fixed-step run (reference devices.py:1172-1196 with h given).
 *   gamma        nonlinear coefficient [1/(W km)] (rounded to the plan's real type)
 *   h_schedule   nsteps step sizes [km], HOST, float32 (C64) or float64 (C128)
 *   snapshots    NULL, or HOST buffer for (nsteps+1, batch, n) complex: the field before the
 *                first step and after every step (reference return_steps, devices.py:1184-1186)
 * Asynchronous on the plan's stream unless snapshots != NULL; see ssfm_synchronize. */
SSFM_API int ssfm_propagate_fixed(ssfm_plan* plan, double gamma, const void* h_schedule, int64_t nsteps,
                         void* snapshots);

/* z-resolved capture that does not stall the loop.  The reference's return_steps keeps the field after EVERY step
 * (devices.py:1150-1152,1184-1186: 16 GiB for the 1000-step run of a 2^20-sample dual-polarisation field) and its consumers plot a few hundred of them
 * (devices.py:2326-2563); ssfm_propagate_fixed(..., snapshots) reproduces that.  This entry point takes
 *   every     > 0 with `fields`: the field after every `every`-th step and after the last one, behind the input (snapshot 0):
 *             1 + ceil(nsteps / every) snapshots of batch x n complex values of the plan's type; 0 with fields == NULL: none
 *   fields    HOST -- page-locked (ssfm_device_alloc(SSFM_HOST_PINNED, ...)) for the transfers to run beside the kernels -- or NULL
 *   scalars   HOST or NULL: (nsteps + 1) x batch x 2 doubles -- after every step s (0: the input) and for every row the mean and the maximum of |A|^2 (the
 *             column kernels have |A|^2 in registers: every wavefront stores its sum and maximum, a small kernel adds them up behind the run in a fixed order)
 * and keeps the fused two-kernel engine: a capture step ADDS a launch that writes the time-order field of that step into a ring of plan-owned device
 * blocks; a helper thread of the plan sends full blocks to `fields` on a stream of its own while the run goes on (the GPU never waits for the host, nor one
 * stream for another: DESIGN.md 8b).  Asynchronous: the call returns when the run is enqueued; `fields` / `scalars` are valid after ssfm_synchronize or
 * any later call on the plan, and must stay valid until then.  The run itself is ssfm_propagate_fixed's, kernel for kernel, so the end field and every
 * snapshot are bit for bit what a plain run of that many steps leaves (with `scalars` the column kernels are another instantiation -- the log is
 * compiled in, not tested for -- and agree with the plain run's to the last bits of a fused product). */
typedef struct ssfm_capture {
    int64_t every;
    void* fields;
    double* scalars;
} ssfm_capture;
SSFM_API int ssfm_propagate_fixed_capture(ssfm_plan* plan, double gamma, const void* h_schedule, int64_t nsteps, const ssfm_capture* cap);

/* Adaptive run (reference devices.py:1155-1156, 1193-1196): h = phi_max / max(|gamma| |A|^2),
 * maximum over all rows of the plan, clamped to length - z; z and h live on the device.
 *   single_step  the caller's evaluation of `(beta_2 == 0 and beta_3 == 0) or gamma == 0`
 *                (devices.py:1156): the first step is then the whole length
 *   max_steps    capacity of z_out / snapshots (run stops with SSFM_ERR_INVALID beyond it)
 *   steps_out    number of steps taken
 *   z_out        NULL or HOST float64[max_steps+1]: z after every step (z_out[0] = 0)
 *   snapshots    NULL or HOST (max_steps+1, batch, n) complex
 * Synchronous (the step count is only known at the end). */
SSFM_API int ssfm_propagate_adaptive(ssfm_plan* plan, double gamma, double length, double phi_max,
                            int single_step, int64_t max_steps, int64_t* steps_out, double* z_out,
                            void* snapshots);

/* The same run in three calls, so that a z-resolved capture (return_steps, devices.py:1184-1186) reaches the host in
 * bounded blocks instead of one (max_steps + 1)-field buffer: ssfm_adaptive_begin chooses the first step (capture != 0
 * announces snapshots); ssfm_adaptive_run takes up to `budget` more steps -- snapshots: NULL, or HOST memory for `budget`
 * fields, the field after each step this call takes (ask ssfm_get_field for the input beforehand) -- and reports the
 * steps taken so far and whether z has reached the length; ssfm_adaptive_finish returns the step count and z_out as
 * ssfm_propagate_adaptive does (z_out: steps + 1 values; size it from ssfm_adaptive_run's count).
 * Between ssfm_adaptive_begin and ssfm_adaptive_finish the plan's field buffer (ssfm_get_field, ssfm_field_device_ptr) is only
 * defined when the run was begun with capture != 0: without a capture the field of a complex64 plan stays in an internal
 * tile-private order between steps, and ssfm_adaptive_finish produces the time-order field. */
SSFM_API int ssfm_adaptive_begin(ssfm_plan* plan, double gamma, double length, double phi_max, int single_step, int64_t max_steps, int capture);
SSFM_API int ssfm_adaptive_run(ssfm_plan* plan, int64_t budget, void* snapshots, int64_t* steps_total, int* done);
SSFM_API int ssfm_adaptive_finish(ssfm_plan* plan, int64_t* steps_out, double* z_out);

/* A z-resolved capture of an ADAPTIVE run that keeps the run's own engine (round 6; the reference's consumer of return_steps calls FIBER with h = None,
 * devices.py:2342).  Between ssfm_adaptive_begin(capture = 0) and the first ssfm_adaptive_run(snapshots = NULL): the field after every `every`-th step
 * (every > 0), or after the steps of an ascending list of step numbers (`steps`, 1-based; e.g. the steps that first reach given z positions, taken from the z
 * log of a run of the same field -- an adaptive run repeats bit for bit), into `fields`: HOST memory (page-locked for the transfers to run beside the
 * kernels) for `capacity` snapshots of batch x n complex values.  The input and the end field are not among them (ssfm_get_field before and after).  A capture
 * step adds ONE launch to the run (the step's time-order field into a ring of plan-owned device blocks); the snapshots of a chunk of queued steps travel
 * while the next chunk runs.  `taken[k]` = step number of snapshot k, `*n_taken` = their count: both valid after ssfm_adaptive_finish. */
typedef struct ssfm_adaptive_capture {
    int64_t every;
    const int64_t* steps;
    int64_t n_steps;
    void* fields;
    int64_t capacity;
    int64_t* taken;
    int64_t* n_taken;
} ssfm_adaptive_capture;
SSFM_API int ssfm_adaptive_set_capture(ssfm_plan* plan, const ssfm_adaptive_capture* cap);

/* out = ifft(fft(field) * H) on every row; H complex, length n, natural FFT order, HOST.
 * (reference DM, devices.py:1027-1029) */
SSFM_API int ssfm_apply_transfer(ssfm_plan* plan, const void* H_host);

/* DM with the transfer function generated on the device (reference devices.py:1025-1029):
 * H(w_k) = exp(+1j w_k^2 D/2), w_k = 2 pi fftfreq(n, dt)[k] in float64 with the reference's operation
 * order, D in s^2 (the caller has applied devices.py:1025's `D *= 1e-12**2`).  If H_out != NULL the
 * natural-order H (n complex, precision's type, HOST) is returned as well (for retH). */
SSFM_API int ssfm_apply_dispersion(ssfm_plan* plan, double dt_s, double D_s2, void* H_out);

/* Zero-phase IIR filtering with a cascade of second-order sections = scipy.signal.sosfiltfilt(sos, x,
 * axis=-1) (odd padding of 3*ntaps samples, steady-state initial conditions): the arithmetic of the
 * reference's LPF (devices.py:1363-1368) and BPF (devices.py:814-823).  Plan-less.
 *   sos  n_sections x 6 float64 (b0 b1 b2 a0 a1 a2, a0 == 1), HOST;  1 <= n_sections <= 4
 *   zi   n_sections x 2 float64 = scipy.signal.sosfilt_zi(sos), HOST
 *   x,y  batch x n float64 (is_complex = 0) or complex128 interleaved (is_complex = 1); y may alias x.  on_device = 0: HOST arrays (uploaded, filtered,
 *        read back); on_device = 1: DEVICE memory of `device` (complex buffers 16-byte aligned), so a field can be filtered where it was
 *        propagated.  Synchronous. */
SSFM_API int ssfm_sosfiltfilt(int device, const double* sos, const double* zi, int n_sections, const void* x, void* y,
                     int64_t n, int batch, int is_complex, int on_device);
/* The last ssfm_sosfiltfilt call on `device` (either pointer nullable): device time [ms] of its kernels (HIP events on the filter's stream, transfers
 * excluded), and its kernel launches: 1 when the whole forward-backward pass ran as one
 * launch (every workgroup of the call resident at once: up to about 2.3 M real or complex samples in all on an
 * MI355X, at most 512 groups of 3072 samples per row), 3 otherwise (longer calls; SSFM_SOS_ONE_LAUNCH=0; a call
 * whose workgroups did not all get to run side by side within SSFM_SOS_PATIENCE_US -- default 2000 -- is
 * repeated in this form, and after three such calls in a row the next 1000 calls skip the one-launch form).  The two
 * forms agree to rounding (the one-launch form uses fused multiply-adds), not bit for bit. */
SSFM_API int ssfm_sosfiltfilt_last(int device, float* ms, int* launches);

/* Square-law detection of the reference's PD (devices.py:1512-1515): i_ph = r * (x * x.conj()).real summed
 * over the polarisations, signal and noise kept apart as the reference's signal algebra does
 * (typing.py:1337-1344): i_sig = r * sum_p |s_p|^2,  i_noise = r * sum_p Re(s_p n_p* + n_p s_p* + n_p n_p*).
 *   sig, noise   n_pol x n complex128; noise may be NULL (then i_noise must be NULL)
 *   post   factor applied to the summed currents (1 for currents; R_load of devices.py:1547 for voltages)
 *   i_sig, i_noise   n float64
 *   on_device   0: all four are HOST arrays; 1: DEVICE buffers (16-byte aligned).  Synchronous. */
SSFM_API int ssfm_square_law(int device, const void* sig, const void* noise, int n_pol, int64_t n, double r, double post, double* i_sig, double* i_noise, int on_device);

/* ---- device-resident signals ----------------------------------------------------------------------
 * Raw HBM buffers that a host-side signal object can own between calls, so that a chain such as
 * FIBER -> DBP -> BPF -> PD moves nothing over PCIe until a result is looked at.  The reference keeps
 * every signal as a NumPy array (typing.py:1022-1165, :2124-2196); the host mirror's `.signal` /
 * `.noise` are materialised lazily from these.  All calls are synchronous.
 *   ssfm_device_alloc / _free   pooled per (device, size); pass the allocation size to _free.  device = SSFM_HOST_PINNED: a page-locked HOST buffer
 *                               (pooled by size) for results that are read back -- the destination of a device-to-host copy
 *   ssfm_device_copy            kind 0 host->device, 1 device->host, 2 device->device, 3 `bytes` zero bytes (src ignored: the empty y polarisation an
 *                               EDFA gives a single-polarisation input, devices.py:924)
 *   ssfm_device_convert         complex64 <-> complex128, SSFM_F64_REAL -> either (zero imaginary part), or complex128 -> SSFM_F64_REAL (the real
 *                               part); `count` elements
 *   ssfm_device_add             dst = a + b, `count` complex elements of `precision` */
#define SSFM_HOST_PINNED (-1)
SSFM_API int ssfm_device_alloc(int device, size_t bytes, void** out);
SSFM_API int ssfm_device_free(int device, void* ptr, size_t bytes);
SSFM_API int ssfm_device_copy(int device, void* dst, const void* src, size_t bytes, int kind);
SSFM_API int ssfm_device_convert(int device, const void* src, int src_precision, void* dst, int dst_precision, int64_t count);
SSFM_API int ssfm_device_add(int device, void* dst, const void* a, const void* b, int precision, int64_t count);
/* Device random numbers and the few float64 array operations the receiver front-end needs around them.
 * ssfm_device_randn: out[i] = mean + std * N(0,1), Philox4x32-10 keyed by `seed`, counter = (pair index, `stream`),
 * Box-Muller on two 53-bit uniforms per pair -- the documented generator behind PD / EDFA with rng="device" (the
 * reference draws from NumPy's global generator, devices.py:1521-1527, :930; the default rng="numpy" reproduces
 * those draws on the host).  sum3: out = (a + b + c + offset) * scale (a, b, c nullable); scale_add: dst = a*factor
 * (+ b).  All on `n` float64 elements in DEVICE memory, synchronous. */
SSFM_API int ssfm_device_randn(int device, double* out_dev, int64_t n, uint64_t seed, uint64_t stream, double mean, double std);
SSFM_API int ssfm_device_sum3(int device, double* out_dev, const double* a, const double* b, const double* c, double offset, double scale, int64_t n);
SSFM_API int ssfm_device_scale_add(int device, double* dst, const double* a, double factor, const double* b, int64_t n);
/* dst = running sum of src (numpy.cumsum; the laser's Wiener phase, devices.py:490), n float64 on the DEVICE. */
SSFM_API int ssfm_device_cumsum(int device, double* dst, const double* src, int64_t n);
/* Elementwise transmitter work on DEVICE arrays (csrc/transmitter.hip), synchronous:
 * ssfm_mzm: the Mach-Zehnder transfer of devices.py:762-778 -- g = k (drive + bias) [+ k drive_noise], h = sqrt_loss (cos g +
 *   j half_eta sin g), out = in * h for signal and (nullable) noise, n_pol x n complex128; polarisation `dead_pol` of a
 *   dual-polarisation input is emptied; the drive is n float64 (drive_complex = 0) or complex128 (1);
 * ssfm_device_axpb: dst = src * alpha + beta on n float64 (is_complex = 0) or complex128 (1; beta to the real part). */
SSFM_API int ssfm_mzm(int device, void* out_sig, void* out_noise, const void* in_sig, const void* in_noise, int n_pol, int64_t n, const void* drive,
             const void* drive_noise, int drive_complex, double k, double bias, double sqrt_loss, double half_eta, int dead_pol);
/* LASER (devices.py:353-510) over t = linspace(0, stop, n) (t_i = i*step): out = amp [exp(j phase)] [sqrt(1 + rin)] [exp(j w t)],
 * the factors in the reference's order; `phase` (running sum of the Wiener increments) and `rin` are nullable float64 DEVICE
 * arrays drawn by the caller, `w` = 2 pi df with has_df.  `out`: n float64 when neither phase nor df is given, else n complex128. */
SSFM_API int ssfm_laser(int device, void* out, int64_t n, double amp, const double* phase, const double* rin, int has_df, double w, double step, double stop);
SSFM_API int ssfm_device_axpb(int device, void* dst, const void* src, double alpha, double beta, int64_t n, int is_complex);
/* Small DEVICE-array helpers that keep the transmitter / amplifier off the host, synchronous:
 * ssfm_device_shift: dst = src + (re + j im);
 * ssfm_device_reduce: the reductions a device-resident signal needs, results on the HOST --
 *   SSFM_REDUCE_MEAN   out[0] = mean of the n float64 of a (+ b, nullable: the mean of the elementwise sum; the photocurrent's mean, devices.py:1521)
 *   SSFM_REDUCE_MEAN2  numpy.mean of n float64 (out[0]) or complex128 values (out[0] + j out[1]) of a -- the DC level that AC coupling removes (DAC, devices.py:339-340)
 *   SSFM_REDUCE_POWER  out[r] = mean |x|^2 of each of `rows` rows of n float64 / complex128 values of a
 *   SSFM_REDUCE_MIN    out[0] = the minimum of the n float64 of a
 * (b, rows and is_complex are ignored where a kind has no use for them). */
enum { SSFM_REDUCE_MEAN = 0, SSFM_REDUCE_MEAN2 = 1, SSFM_REDUCE_POWER = 2, SSFM_REDUCE_MIN = 3 };
SSFM_API int ssfm_device_shift(int device, void* dst, const void* src, int64_t n, int is_complex, double re, double im);
SSFM_API int ssfm_device_reduce(int device, int kind, const void* a, const void* b, int rows, int64_t n, int is_complex, double* out);
/* PRBS (reference devices.py:63-182): `len` bits of the Fibonacci LFSR x^order + x^t2 + 1 (orders 7, 9, 11, 15, 20,
 * 23, 31; taps of devices.py:134-142) started from the non-zero state `seed` (the caller has applied devices.py:143-149:
 * modulo 2^order, default all ones, 0 -> 1), one uint8 0/1 per bit in DEVICE memory; the sequence and the register state
 * after `len` shifts (`final_state`, nullable HOST word: return_seed, devices.py:181) are bit for bit the reference's.
 * Every thread jumps to its 256-bit chunk with powers of the shift matrix over GF(2) and walks it as devices.py:170-175.
 * (ssfm_load_symbols with src_kind 1 takes the bits from the device as they lie: PRBS -> DAC without a host copy.)
 * ssfm_load_qpsk: plan field (complex128, `rows` rows of plan_n) <- the QPSK-like test symbols of the benchmark
 * configurations: row r, symbol k = ((2 b0 - 1) + j (2 b1 - 1)) / sqrt(2) from bits 2 (r nsym + k) and
 * + 1, at sample k sps + sps / 2, zeros elsewhere.  Asynchronous on the plan's stream. */
SSFM_API int ssfm_prbs(int device, void* bits_dev, int64_t len, int order, uint32_t seed, uint32_t* final_state);
SSFM_API int ssfm_load_qpsk(ssfm_plan* plan, int64_t plan_n, int rows, const void* bits_dev, int64_t nsym, int sps);
/* Free / total HBM of the device and the bytes held in the library's buffer pool (nullable). */
SSFM_API int ssfm_device_mem_info(int device, size_t* free_bytes, size_t* total_bytes, size_t* pooled_bytes);

/* ---- lengths that are not powers of two (the reference takes any N: numpy.fft, devices.py:1178-1180) --------
 * Bluestein's identity maps a length-n transform onto the circular convolution of a power-of-two plan of length
 * M >= 2n - 1, with the chirp c_m = exp(-i pi m^2 / n):
 *     fft_n(x)_k  = c_k       sum_m (x_m c_m)        conj(c)_{k-m}
 *     ifft_n(X)_m = conj(c_m) sum_k (X_k conj(c_k))  c_{m-k}  / n
 * (csrc/chirpz.hip has the algebra of a split step).  Round 5: the exported surface is FIVE calls -- round 4 exported the step's twelve
 * fragments (chirp_pre / _mid / _post / _step / _line_run / _small / _medium ...), which only one particular caller could sequence correctly;
 * they are internal now (csrc/ssfm_common.hpp) and every engine is chosen inside.
 *
 * ssfm_transfer_table keeps a transfer function H (HOST, plan length complex, the plan's type) on the device in slot 0 or 1;
 * ssfm_apply_table does x <- ifft(fft(x) * H) on the plan's field WITHOUT waiting for the host (also the DAC's pulse shaping, below). */
SSFM_API int ssfm_transfer_table(ssfm_plan* plan, const void* H_host, int slot);
SSFM_API int ssfm_apply_table(ssfm_plan* plan, int slot);
/* The chirp c (conj = 0) or conj(c) (1), n complex128, into DEVICE memory, its phase reduced exactly in integers (synchronous). */
SSFM_API int ssfm_device_chirp(int device, void* out_dev, int64_t n, int conj);
/* One-time set-up of a COMPLEX128 plan of plan_n >= 2n - 1 points for fields of n samples: both convolution kernels of the identity
 * (v[m] = v[plan_n - m] = conj(c_m) and c_m) are generated in the plan's field and transformed there into the resident transfer functions of
 * slots 0 and 1 -- neither a host transform nor an upload.  What ssfm_chirp_propagate / _transfer / _fourier rely on.  Asynchronous; the
 * plan's field is consumed. */
SSFM_API int ssfm_chirp_setup(ssfm_plan* plan, int64_t plan_n, int64_t n);
/* A whole FIBER / DBP run on a field of ANY length, driven from C on the complex128 line: A (batch x n complex128, DEVICE) is
 * advanced in place, P is scratch (batch x n float64, DEVICE), chirp = c (ssfm_device_chirp), Dt = D~ (n complex128, DEVICE, natural
 * frequency order); the plan: complex128, plan_n >= 2n - 1, prepared by ssfm_chirp_setup(plan, plan_n, n).  hs != NULL: fixed step, `nsteps`
 * step sizes (HOST; steps of length 0 are the identity).  hs == NULL: the adaptive rule of devices.py:1172-1196, h = phi_max / (|gamma| max|A|^2)
 * clamped to the rest of `length`, evaluated on the device in float32 (f32 != 0: the reference's arithmetic in complex64 mode) or float64; z_out
 * (HOST, nullable, max_steps + 1 doubles) receives z after every step, *steps_out the steps taken.  Engines, chosen inside: the whole run in one
 * launch for plan_n <= 4096 (a workgroup per row); four launches per fixed step for schedules of up to four step sizes (the chirps of neighbouring
 * steps cancel: one pointwise launch before and after the run); five otherwise, seven per adaptive step.  A caller that wants the field after
 * every step (return_steps) calls it a step at a time (nsteps = 1, or max_steps = 1 over the rest of the length).  Synchronous.
 * f32 != 0 also says that the CALLER's field is complex64 (widened into A for the call): such a four-launch run on a line of 2^18 points and more keeps the
 * line as complex64 values BETWEEN its passes -- every pass computes in float64 -- which halves the field's bytes per pass and sits a few 1e-7 from the
 * complex128 line after 100 steps (SSFM_CHIRP_HALF=0: never); the rows go out on the plan's two lanes. */
SSFM_API int ssfm_chirp_propagate(ssfm_plan* plan, int64_t plan_n, int batch, void* A, void* P, const void* chirp, const void* Dt, int64_t n, double gamma,
                         const double* hs, int64_t nsteps, double length, double phi_max, int f32, int64_t max_steps, double* z_out, int64_t* steps_out);
/* The same run for complex64 callers in ONE launch on a complex64 line: the plan is COMPLEX64 and its length is the line's, M = 2^k >= 2n - 1 -- a
 * workgroup per row for M <= 4096 (at most 16 rows in adaptive mode), the one-XCD engine for M = 2^13 ... 2^17 with at most 2^17 points in all rows
 * (four passes per step; fixed step: at most four distinct sizes).  A, chirp (c rounded to complex64), Dt: complex64, DEVICE; needs no
 * ssfm_chirp_setup (the line's tables are the plan's own).  hs / nsteps / length / phi_max / max_steps / z_out / steps_out as above, the adaptive rule
 * in float32.  Synchronous except for a fixed-step run with M <= 4096 (asynchronous on the plan's stream).  SSFM_ERR_UNSUPPORTED with A as it was: the
 * plan has no such engine or schedule, or the launch's workgroups did not meet within their patience (the engine is then off for this plan) -- the
 * caller takes ssfm_chirp_propagate.  (The reference transforms such a length in single precision itself -- pocketfft's Bluestein -- so this is its
 * arithmetic class; its accuracy margin and when a caller should prefer the complex128 line: opticomlib_amd/devices.py _c64_line_has_margin.) */
SSFM_API int ssfm_chirp_propagate_c64(ssfm_plan* plan, void* A, const void* chirp, const void* Dt, int64_t n, double gamma, const double* hs, int64_t nsteps,
                             double length, double phi_max, int64_t max_steps, double* z_out, int64_t* steps_out);
/* x <- ifft_n(fft_n(x) * tab) (exponent = 0: DM's H for any length, devices.py:1019-1035) or * exp(tab) (exponent != 0) for every row of the DEVICE array
 * A (batch x n complex128, in place); tab: n complex128, DEVICE.  Plan as for ssfm_chirp_propagate.  Asynchronous on the plan's stream. */
SSFM_API int ssfm_chirp_transfer(ssfm_plan* plan, int64_t plan_n, int batch, void* A, const void* chirp, const void* tab, int64_t n, int exponent);
/* numpy.fft.fft (inverse = 0, unscaled) or ifft (inverse != 0, with its 1/n) of every row of A (batch x n complex128, DEVICE, in place);
 * chirp = c, chirp_conj = conj(c).  The reference's signal('w') / signal('t'), typing.py:1421-1462.  Asynchronous on the plan's stream. */
SSFM_API int ssfm_chirp_fourier(ssfm_plan* plan, int64_t plan_n, int batch, void* A, const void* chirp, const void* chirp_conj, int64_t n, int inverse);
/* Pulse shaping of the reference's DAC (upfir, utils.py:1949-1981) on a complex128 plan of batch 1: ssfm_load_padded
 * writes `n_src` float64 (src_complex = 0) or complex128 (1) samples from DEVICE memory into the field, zero-padded;
 * ssfm_table_from_field makes slot <- fft(field) (the field is consumed); ssfm_load_symbols writes `nsym` symbols --
 * float64 amplitudes (src_kind 0) or uint8 bits taken as 0.0 / 1.0 (src_kind 1), DEVICE -- zero-stuffed to `up` samples per symbol with the sample at up / 2; ssfm_apply_table then convolves.  Asynchronous. */
SSFM_API int ssfm_load_padded(ssfm_plan* plan, int64_t plan_n, const void* src_dev, int src_complex, int64_t n_src);
SSFM_API int ssfm_load_symbols(ssfm_plan* plan, int64_t plan_n, const void* src_dev, int src_kind, int64_t nsym, int up);
/* The DAC's built-in pulses (utils.py:1791-1947) generated in the field instead of uploaded: `npts` samples over
 * t_i = i*step + start (t_{npts-1} = stop: numpy.linspace), zero-padded.  kind 0 nrz (params: -T/2, T/2); 1 gaussian
 * exp(-((p0 + j p1) t)^pow2m) with (p0, p1) = alpha (1 + j c); 2 raised cosine (2 beta, pi beta, value where
 * 1 - (2 beta t)^2 ~ 0); 3 root raised cosine (beta, 4 beta, 1 - beta, 1 + beta, 1/(4 beta), value at 0, value at
 * 1/(4 beta)); 4 sinc.  `params`: 7 doubles on the HOST.  Asynchronous. */
SSFM_API int ssfm_load_pulse(ssfm_plan* plan, int64_t plan_n, int kind, int64_t npts, double start, double step, double stop, int pow2m, const double* params);
SSFM_API int ssfm_table_from_field(ssfm_plan* plan, int slot);
/* Validation aids behind one entry point.  SSFM_DEBUG_FFT: the forward FFT of every row of the plan's field into HOST `dst` (natural frequency order,
 * unscaled).  SSFM_DEBUG_LANE_FAULT: test hook of the lane health check (ssfm_last_run_info) -- arg 1 makes the plan believe a four times better launch
 * period than it has seen (the next long two-lane run looks slow and the lanes are rated again), arg 2 additionally makes every rating come out bad (the
 * repair fails; the second failure drops the plan to one lane), arg 0 back to normal. */
enum { SSFM_DEBUG_FFT = 0, SSFM_DEBUG_LANE_FAULT = 1 };
SSFM_API int ssfm_debug(ssfm_plan* plan, int what, int64_t arg, void* dst);

SSFM_API int ssfm_synchronize(ssfm_plan* plan);
/* The plan's hipStream_t (as void*), so callers can record events around propagate calls or order their own work behind a run (see
 * "WHEN THE FIELD IS VALID" at ssfm_field_device_ptr). */
SSFM_API void* ssfm_stream(ssfm_plan* plan);

/* Which engine the plan's last run took, and whether it had to be repeated on a fallback.  The engines differ in speed only (same results
 * within the stated tolerances); the single-launch ones need the GPU to run all their workgroups side by side and fall back, after a bounded
 * wait, when it does not (a shared GPU, a profiler): without this query that is invisible to the caller.
 *   engine            a value of enum ssfm_engine below: the engine that produced the result of the LAST run (after a fallback: the fallback)
 *   fell_back         1: the last run was started on a single-launch engine, gave up and was repeated
 *   fallbacks_total   such repeats over the life of the plan (a plan keeps to the fallback engine after the first)
 *   lanes             row groups ("lanes") the plan's fixed-step runs drive now on separate streams: `lanes_configured` (env SSFM_LANES, default 2 from 2^20
 *                     samples in all, never more than the batch; one kernel launch covers batch / lanes rows), or 1 once the plan has dropped its second lane (below)
 *   lanes_share_queue 1: a lane's stream shares a hardware queue with an earlier lane's and no replacement stream did better (more than four
 *                     high-priority streams alive): the lanes' kernels run one after the other
 *   lanes_remade      lane streams replaced at RUN time.  Which hardware queue the runtime gives a stream decides whether a lane's kernels run beside
 *                     the other lane's at full rate or 3-6 x slower (profiles/r04_order_dependence.txt); ssfm_plan_create rates every lane with the
 *                     plan's own two kernels (launch period with the other lane running / alone, `lane_score`; good <= 1.6) and replaces a stream
 *                     that is not good, and every fixed-step run of >= 64 steps on two lanes is looked at afterwards: a launch period
 *                     (`lane_last_us`) beyond 1.8 x the best the plan has seen (`lane_pair_us`) has the lanes rated again, on scratch fields,
 *                     and repaired before the next run is enqueued
 *   lanes_dropped     1: two repairs in a row found no good stream and the plan runs both rows in one launch on one stream from now on
 *   lane_heals        times a slow run made the plan rate its lanes again (at most 8 over a plan's life)
 *   lane_alone_us / lane_pair_us / lane_last_us / lane_score   the figures above [us per launch of a lane]; 0 = not measured (one-lane plans)
 * `info_bytes` = sizeof(ssfm_run_info) of the caller's header (a later library fills as much as the caller knows).  For a run whose fallback is
 * resolved lazily (see ssfm_field_device_ptr) call ssfm_synchronize first. */
enum ssfm_engine {
    SSFM_ENGINE_NONE = 0,
    SSFM_ENGINE_TWO_KERNEL = 1,        /* fixed step: two launches per step (k_time, k_freq), one stream per lane */
    SSFM_ENGINE_SMALL = 2,             /* fixed step: one launch per run, a workgroup per row (<= 8192 samples) */
    SSFM_ENGINE_MEDIUM = 3,            /* fixed step: one launch per run on one XCD (2^12 ... 2^17 samples) */
    SSFM_ENGINE_ADAPT_3 = 4,           /* adaptive: three launches per step */
    SSFM_ENGINE_ADAPT_FUSED = 5,       /* adaptive: two launches per step, the step size found inside the column launch */
    SSFM_ENGINE_SMALL_ADAPT = 6,       /* adaptive: one launch per run, one workgroup */
    SSFM_ENGINE_MEDIUM_ADAPT = 7,      /* adaptive: one launch per run on one XCD */
    SSFM_ENGINE_CHIRP_SMALL = 8,       /* any length <= 2048: fixed step, one launch per run */
    SSFM_ENGINE_CHIRP_SMALL_ADAPT = 9, /* any length <= 2048: adaptive, one launch per run */
    SSFM_ENGINE_CHIRP_STEPS = 10,      /* any length: four launches per fixed step (five through the library-internal chirp step of a host-driven loop), seven adaptive */
    SSFM_ENGINE_CHIRP_MEDIUM = 11,     /* any length, 2048 < n <= 65536, complex64: fixed step, one launch per run on one XCD */
    SSFM_ENGINE_CHIRP_MEDIUM_ADAPT = 12, /* ... adaptive */
    SSFM_ENGINE_SPLIT = 13,            /* more than 2^22 samples per row, fixed step: four launches per step (csrc/ssfm_split.hpp) */
    SSFM_ENGINE_SPLIT_ADAPT = 14       /* ... adaptive: five launches per step */
};
typedef struct ssfm_run_info {
    int engine;
    int fell_back;
    int64_t fallbacks_total;
    int lanes;
    int lanes_configured;
    int lanes_share_queue;
    int lanes_remade;
    int lanes_dropped;
    int lane_heals;
    float lane_alone_us;
    float lane_pair_us;
    float lane_last_us;
    float lane_score;
    /* round 6 (appended: callers built against the shorter struct pass its size and are not written beyond it) */
    int lanes_from_pool;        /* 1: this plan's pair of lane streams came rated from the process's pool -- no probe launches at its creation */
    int pad_;
    int64_t lane_ratings_total; /* process-wide: two-lane plans whose creation rated fresh streams (about 7 ms of probe launches each) */
    int64_t lane_pairs_reused;  /* process-wide: two-lane plans that took a rated pair from the pool instead */
} ssfm_run_info;
SSFM_API int ssfm_last_run_info(ssfm_plan* plan, ssfm_run_info* info, size_t info_bytes);


/* Time of the last propagate call measured with HIP events on the plan's stream [ms], and the
 * number of kernel launches it made.  Valid after ssfm_synchronize. */
SSFM_API int ssfm_last_propagate_ms(ssfm_plan* plan, float* ms, int64_t* launches);

/* Kernel timing of ssfm_propagate_fixed with HIP events on the streams the kernels are launched on.
 *   mode 0  off
 *   mode 1  an event after EVERY launch: per-class times (index 0 = k_time begin/mid/end, 1 = k_freq), but
 *           the marker packets slow a launch-dense run noticeably (~30 % at 11 us per launch)
 *   mode 2  an event after every 64th launch of a stream: negligible overhead; each interval is split
 *           between the two classes by launch count (so both classes report the pooled average)
 *   mode 3  sampled: after every 14 launches of a stream ONE launch of each class is bracketed by two events
 *           (3 extra markers per 16 launches, the host stays ahead of the GPU); ssfm_kernel_times then reports
 *           only those single-launch intervals: launches sampled and their summed duration per class
 * After ssfm_synchronize, ssfm_kernel_times returns per class the number of launches and the summed
 * event-to-event time in ms (a dependent-launch gap is counted with the launch that follows it). */
SSFM_API int ssfm_set_profiling(ssfm_plan* plan, int mode);
/* What the plan's staging buffers hold, owned by the plan: `which` 0 labels the linear operator set with
 * ssfm_set_linear_operator, 1 / 2 the resident transfer function of slot 0 / 1.  A caller labels what it has staged
 * (any non-zero 64-bit tag, e.g. a hash of the fibre parameters) and asks later whether it is still there; EVERY entry
 * point that overwrites or reuses the buffer (a new operator, ssfm_apply_transfer, ssfm_apply_dispersion with H_out,
 * ssfm_transfer_table, ssfm_table_from_field) clears the label, so a cached operator can never be stale.  0 = unknown.
 * (The reference recomputes D~ on every call, devices.py:1137-1145; this only saves the O(N) set-up of a repeated call.) */
SSFM_API int ssfm_plan_set_tag(ssfm_plan* plan, int which, uint64_t tag);
SSFM_API int ssfm_plan_get_tag(ssfm_plan* plan, int which, uint64_t* tag);
SSFM_API int ssfm_kernel_times(ssfm_plan* plan, int64_t counts[2], double total_ms[2]);

#ifdef __cplusplus
}
#endif
#endif /* SSFM_AMD_H */
