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
 *   devices.py:1363-1368, :814-823 (LPF / BPF:     ssfm_sosfiltfilt, ssfm_sosfiltfilt_device
 *     scipy.signal.sosfiltfilt)
 *   devices.py:1512-1515 (PD: r * |x|^2, pol sum)  ssfm_square_law, ssfm_square_law_device
 *   devices.py:1521-1549, :930-936 (PD / EDFA      ssfm_device_randn (rng = "device"), _mean, _sum3, _scale_add
 *     noise currents, gain and ASE loading)
 *   numpy.fft of any N (devices.py:1178-1180)      ssfm_chirp_pre / _mid / _post on a power-of-two plan
 *   utils.py:1791-1981 (DAC: pulses, upfir)        ssfm_load_pulse / _load_padded / _load_symbols, ssfm_table_from_field,
 *                                                  ssfm_apply_table, ssfm_device_axpb / _real
 *   devices.py:480-510 (LASER), :762-778 (MZM)     ssfm_laser (+ ssfm_device_cumsum / _min), ssfm_mzm
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
int ssfm_abi_version(void);
int ssfm_device_count(int* count);
const char* ssfm_last_error(void);

/* Smallest / largest supported log2(n) for a precision (currently 8..22). */
int ssfm_supported_log2n(int precision, int* lo, int* hi);

/* Allocate every device buffer once: field, stale |A|^2, operator tables, twiddles. */
int ssfm_plan_create(ssfm_plan** out, int device, int64_t n, int batch, int precision);
int ssfm_plan_destroy(ssfm_plan* plan);

/* D~(w) [1/km], complex (precision's element type), length n, natural FFT order, HOST memory.
 * Shared by all rows.  (reference devices.py:1145) */
int ssfm_set_linear_operator(ssfm_plan* plan, const void* dtilde_host);

/* Copy a (batch, n) complex field in / out of the plan.  `is_device` != 0: the pointer is
 * device memory on the plan's device. Both calls are ordered on the plan's stream;
 * ssfm_get_field returns after the copy has completed. */
int ssfm_set_field(ssfm_plan* plan, const void* src, int is_device);
int ssfm_get_field(ssfm_plan* plan, void* dst, int is_device);
/* Device address of the plan's resident field (batch*n complex), natural time order
 * between propagate calls.
 * WHEN THE FIELD IS VALID.  Runs are asynchronous: the result is in the field buffer once ssfm_synchronize / ssfm_get_field has
 * returned SSFM_OK.  A caller that orders its OWN work behind a run on the plan's stream (ssfm_stream) instead may do so: from the
 * first call of ssfm_field_device_ptr or ssfm_stream on a plan, ssfm_propagate_fixed resolves the one kind of run whose success is only
 * known at its end -- the one-launch engine of plans of 2^12 ... 2^17 samples, whose workgroups meet inside the launch and give up
 * when the GPU does not run them side by side; the run is then repeated on the launch-per-pass engine -- BEFORE it returns (for such
 * plans the call is then synchronous).  Every other engine either needs no fallback or resolves it inside the call already.
 * ssfm_last_run_info tells which engine a run took and whether it fell back. */
void* ssfm_field_device_ptr(ssfm_plan* plan);

/* Fixed-step run (reference devices.py:1172-1196 with h given).
 *   gamma        nonlinear coefficient [1/(W km)] (rounded to the plan's real type)
 *   h_schedule   nsteps step sizes [km], HOST, float32 (C64) or float64 (C128)
 *   snapshots    NULL, or HOST buffer for (nsteps+1, batch, n) complex: the field before the
 *                first step and after every step (reference return_steps, devices.py:1184-1186)
 * Asynchronous on the plan's stream unless snapshots != NULL; see ssfm_synchronize. */
int ssfm_propagate_fixed(ssfm_plan* plan, double gamma, const void* h_schedule, int64_t nsteps,
                         void* snapshots);

/* Adaptive run (reference devices.py:1155-1156, 1193-1196): h = phi_max / max(|gamma| |A|^2),
 * maximum over all rows of the plan, clamped to length - z; z and h live on the device.
 *   single_step  the caller's evaluation of `(beta_2 == 0 and beta_3 == 0) or gamma == 0`
 *                (devices.py:1156): the first step is then the whole length
 *   max_steps    capacity of z_out / snapshots (run stops with SSFM_ERR_INVALID beyond it)
 *   steps_out    number of steps taken
 *   z_out        NULL or HOST float64[max_steps+1]: z after every step (z_out[0] = 0)
 *   snapshots    NULL or HOST (max_steps+1, batch, n) complex
 * Synchronous (the step count is only known at the end). */
int ssfm_propagate_adaptive(ssfm_plan* plan, double gamma, double length, double phi_max,
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
int ssfm_adaptive_begin(ssfm_plan* plan, double gamma, double length, double phi_max, int single_step, int64_t max_steps, int capture);
int ssfm_adaptive_run(ssfm_plan* plan, int64_t budget, void* snapshots, int64_t* steps_total, int* done);
int ssfm_adaptive_finish(ssfm_plan* plan, int64_t* steps_out, double* z_out);

/* out = ifft(fft(field) * H) on every row; H complex, length n, natural FFT order, HOST.
 * (reference DM, devices.py:1027-1029) */
int ssfm_apply_transfer(ssfm_plan* plan, const void* H_host);

/* DM with the transfer function generated on the device (reference devices.py:1025-1029):
 * H(w_k) = exp(+1j w_k^2 D/2), w_k = 2 pi fftfreq(n, dt)[k] in float64 with the reference's operation
 * order, D in s^2 (the caller has applied devices.py:1025's `D *= 1e-12**2`).  If H_out != NULL the
 * natural-order H (n complex, precision's type, HOST) is returned as well (for retH). */
int ssfm_apply_dispersion(ssfm_plan* plan, double dt_s, double D_s2, void* H_out);

/* Zero-phase IIR filtering with a cascade of second-order sections = scipy.signal.sosfiltfilt(sos, x,
 * axis=-1) (odd padding of 3*ntaps samples, steady-state initial conditions): the arithmetic of the
 * reference's LPF (devices.py:1363-1368) and BPF (devices.py:814-823).  Plan-less.
 *   sos  n_sections x 6 float64 (b0 b1 b2 a0 a1 a2, a0 == 1), HOST;  1 <= n_sections <= 4
 *   zi   n_sections x 2 float64 = scipy.signal.sosfilt_zi(sos), HOST
 *   x,y  HOST, batch x n float64 (is_complex = 0) or complex128 interleaved (is_complex = 1); y may alias x */
int ssfm_sosfiltfilt(int device, const double* sos, const double* zi, int n_sections, const void* x, void* y,
                     int64_t n, int batch, int is_complex);
/* The same with x and y in DEVICE memory of `device` (complex buffers 16-byte aligned; y may alias x), so a
 * field can be filtered where it was propagated.  sos and zi stay HOST arrays.  Synchronous. */
int ssfm_sosfiltfilt_device(int device, const double* sos, const double* zi, int n_sections, const void* x_dev, void* y_dev,
                            int64_t n, int batch, int is_complex);
/* Device time [ms] of the kernels of the last ssfm_sosfiltfilt* call on `device` (HIP events on the
 * filter's stream, transfers excluded). */
int ssfm_sosfiltfilt_last_ms(int device, float* ms);
/* Kernel launches of the last ssfm_sosfiltfilt* call on `device`: 1 when the whole forward-backward pass ran as one
 * launch (every workgroup of the call resident at once: up to about 2.3 M real or complex samples in all on an
 * MI355X, at most 512 groups of 3072 samples per row), 3 otherwise (longer calls; SSFM_SOS_ONE_LAUNCH=0; a call
 * whose workgroups did not all get to run side by side within SSFM_SOS_PATIENCE_US -- default 2000 -- is
 * repeated in this form, and after three such calls in a row the next 1000 calls skip the one-launch form).  The two
 * forms agree to rounding (the one-launch form uses fused multiply-adds), not bit for bit. */
int ssfm_sosfiltfilt_last_launches(int device, int* launches);

/* Square-law detection of the reference's PD (devices.py:1512-1515): i_ph = r * (x * x.conj()).real summed
 * over the polarisations, signal and noise kept apart as the reference's signal algebra does
 * (typing.py:1337-1344): i_sig = r * sum_p |s_p|^2,  i_noise = r * sum_p Re(s_p n_p* + n_p s_p* + n_p n_p*).
 *   sig, noise   HOST, n_pol x n complex128; noise may be NULL (then i_noise must be NULL)
 *   post   factor applied to the summed currents (1 for currents; R_load of devices.py:1547 for voltages)
 *   i_sig, i_noise   HOST, n float64 */
int ssfm_square_law(int device, const void* sig, const void* noise, int n_pol, int64_t n, double r, double post, double* i_sig, double* i_noise);

/* The same on DEVICE buffers (16-byte aligned).  Synchronous. */
int ssfm_square_law_device(int device, const void* sig, const void* noise, int n_pol, int64_t n, double r, double post, double* i_sig, double* i_noise);

/* ---- device-resident signals ----------------------------------------------------------------------
 * Raw HBM buffers that a host-side signal object can own between calls, so that a chain such as
 * FIBER -> DBP -> BPF -> PD moves nothing over PCIe until a result is looked at.  The reference keeps
 * every signal as a NumPy array (typing.py:1022-1165, :2124-2196); the host mirror's `.signal` /
 * `.noise` are materialised lazily from these.  All calls are synchronous.
 *   ssfm_device_alloc / _free   pooled per (device, size); pass the allocation size to _free
 *   ssfm_device_copy            kind 0 host->device, 1 device->host, 2 device->device
 *   ssfm_device_convert         complex64 <-> complex128, or SSFM_F64_REAL -> either (zero imaginary part); `count` elements
 *   ssfm_device_add             dst = a + b, `count` complex elements of `precision` */
int ssfm_device_alloc(int device, size_t bytes, void** out);
int ssfm_device_free(int device, void* ptr, size_t bytes);
int ssfm_device_copy(int device, void* dst, const void* src, size_t bytes, int kind);
/* Page-locked host buffers (pooled by size) for results that are read back: the destination of a device-to-host copy. */
int ssfm_host_alloc(size_t bytes, void** out);
int ssfm_host_free(void* ptr, size_t bytes);
int ssfm_device_convert(int device, const void* src, int src_precision, void* dst, int dst_precision, int64_t count);
int ssfm_device_add(int device, void* dst, const void* a, const void* b, int precision, int64_t count);
/* Device random numbers and the few float64 array operations the receiver front-end needs around them.
 * ssfm_device_randn: out[i] = mean + std * N(0,1), Philox4x32-10 keyed by `seed`, counter = (pair index, `stream`),
 * Box-Muller on two 53-bit uniforms per pair -- the documented generator behind PD / EDFA with rng="device" (the
 * reference draws from NumPy's global generator, devices.py:1521-1527, :930; the default rng="numpy" reproduces
 * those draws on the host).  sum3: out = (a + b + c + offset) * scale (a, b, c nullable); scale_add: dst = a*factor
 * (+ b); mean: mean of a (+ b).  All on `n` float64 elements in DEVICE memory, synchronous. */
int ssfm_device_randn(int device, double* out_dev, int64_t n, uint64_t seed, uint64_t stream, double mean, double std);
int ssfm_device_sum3(int device, double* out_dev, const double* a, const double* b, const double* c, double offset, double scale, int64_t n);
int ssfm_device_scale_add(int device, double* dst, const double* a, double factor, const double* b, int64_t n);
int ssfm_device_mean(int device, const double* a, const double* b, int64_t n, double* mean_out);
/* dst = running sum of src (numpy.cumsum; the laser's Wiener phase, devices.py:490) and the minimum of a, n float64 on the DEVICE. */
int ssfm_device_cumsum(int device, double* dst, const double* src, int64_t n);
int ssfm_device_min(int device, const double* a, int64_t n, double* min_out);
/* Elementwise transmitter work on DEVICE arrays (csrc/transmitter.hip), synchronous:
 * ssfm_mzm: the Mach-Zehnder transfer of devices.py:762-778 -- g = k (drive + bias) [+ k drive_noise], h = sqrt_loss (cos g +
 *   j half_eta sin g), out = in * h for signal and (nullable) noise, n_pol x n complex128; polarisation `dead_pol` of a
 *   dual-polarisation input is emptied; the drive is n float64 (drive_complex = 0) or complex128 (1);
 * ssfm_device_axpb: dst = src * alpha + beta on n float64 (is_complex = 0) or complex128 (1; beta to the real part);
 * ssfm_device_real: dst (float64) = real part of src (complex128). */
int ssfm_mzm(int device, void* out_sig, void* out_noise, const void* in_sig, const void* in_noise, int n_pol, int64_t n, const void* drive,
             const void* drive_noise, int drive_complex, double k, double bias, double sqrt_loss, double half_eta, int dead_pol);
/* LASER (devices.py:353-510) over t = linspace(0, stop, n) (t_i = i*step): out = amp [exp(j phase)] [sqrt(1 + rin)] [exp(j w t)],
 * the factors in the reference's order; `phase` (running sum of the Wiener increments) and `rin` are nullable float64 DEVICE
 * arrays drawn by the caller, `w` = 2 pi df with has_df.  `out`: n float64 when neither phase nor df is given, else n complex128. */
int ssfm_laser(int device, void* out, int64_t n, double amp, const double* phase, const double* rin, int has_df, double w, double step, double stop);
int ssfm_device_axpb(int device, void* dst, const void* src, double alpha, double beta, int64_t n, int is_complex);
int ssfm_device_real(int device, double* dst, const void* src, int64_t n);
/* Small DEVICE-array helpers that keep the transmitter / amplifier off the host, synchronous:
 * ssfm_device_mean2: numpy.mean of n float64 (out[0]) or complex128 values (out[0] + j out[1]) -- the DC level that
 *   AC coupling removes (DAC, devices.py:339-340);  ssfm_device_shift: dst = src + (re + j im);
 * ssfm_device_zero: `bytes` zero bytes (the empty y polarisation an EDFA gives a single-polarisation input, devices.py:924);
 * ssfm_device_power: mean |x|^2 of each of `rows` rows of n float64 / complex128 values -> out[rows] (HOST). */
int ssfm_device_mean2(int device, const void* src, int64_t n, int is_complex, double* out);
int ssfm_device_shift(int device, void* dst, const void* src, int64_t n, int is_complex, double re, double im);
int ssfm_device_zero(int device, void* dst, size_t bytes);
int ssfm_device_power(int device, const void* src, int rows, int64_t n, int is_complex, double* out);
/* PRBS (reference devices.py:63-182): `len` bits of the Fibonacci LFSR x^order + x^t2 + 1 (orders 7, 9, 11, 15, 20,
 * 23, 31; taps of devices.py:134-142) started from the non-zero state `seed` (the caller has applied devices.py:143-149:
 * modulo 2^order, default all ones, 0 -> 1), one uint8 0/1 per bit in DEVICE memory; the sequence and the register state
 * after `len` shifts (`final_state`, nullable HOST word: return_seed, devices.py:181) are bit for bit the reference's.
 * Every thread jumps to its 256-bit chunk with powers of the shift matrix over GF(2) and walks it as devices.py:170-175.
 * ssfm_load_bits: plan field (complex128, batch 1) <- the bits as amplitudes 0.0 / 1.0 zero-stuffed to `up` samples per
 * bit (ssfm_load_symbols for bits that are already on the device: PRBS -> DAC without a host copy).
 * ssfm_load_qpsk: plan field (complex128, `rows` rows of plan_n) <- the QPSK-like test symbols of the benchmark
 * configurations: row r, symbol k = ((2 b0 - 1) + j (2 b1 - 1)) / sqrt(2) from bits 2 (r nsym + k) and
 * + 1, at sample k sps + sps / 2, zeros elsewhere.  The two loaders are asynchronous on the plan's stream. */
int ssfm_prbs(int device, void* bits_dev, int64_t len, int order, uint32_t seed, uint32_t* final_state);
int ssfm_load_bits(ssfm_plan* plan, int64_t plan_n, const void* bits_dev, int64_t nbits, int up);
int ssfm_load_qpsk(ssfm_plan* plan, int64_t plan_n, int rows, const void* bits_dev, int64_t nsym, int sps);
/* Free / total HBM of the device and the bytes held in the library's buffer pool (nullable). */
int ssfm_device_mem_info(int device, size_t* free_bytes, size_t* total_bytes, size_t* pooled_bytes);

/* ---- lengths that are not powers of two (the reference takes any N: numpy.fft, devices.py:1178-1180) --------
 * Bluestein's identity maps a length-N transform onto the circular convolution of a power-of-two plan of length
 * M >= 2N - 1 (complex128).  ssfm_transfer_table keeps a transfer function H (HOST, M complex, the plan's type) on
 * the device in slot 0 or 1; ssfm_apply_table does x <- ifft(fft(x) * H) on the plan's field WITHOUT waiting for the
 * host; the chirp kernels move a caller-owned DEVICE field A (batch x N complex128, natural order) into and out of
 * the plan's field buffer and apply the step's elementwise operators (csrc/chirpz.hip has the algebra):
 *   ssfm_chirp_pre   F = A exp(i gamma |A|^2 hh) c, zero-padded; P (nullable) receives |A|^2
 *   ssfm_chirp_mid   F = F exp(tab h) (mode 0, tab = D~)  or  F tab (mode 1, tab = a transfer function), N entries
 *   ssfm_chirp_post  A = F conj(c) / N exp(i gamma P hh); maxbits_dev (nullable, 8 bytes) = bit pattern of max |A|^2
 * `plan_n` / `batch` are the plan's own length and batch; all asynchronous on the plan's stream. */
int ssfm_transfer_table(ssfm_plan* plan, const void* H_host, int slot);
int ssfm_apply_table(ssfm_plan* plan, int slot);
/* x <- ifft(fft(ifft(fft(x) * H0) * mul) * H1) on the plan's field: ssfm_apply_table(plan, 0), a pointwise product with the time-domain table `mul`
 * (plan length entries, DEVICE, the plan's precision, the same for every row), ssfm_apply_table(plan, 1) -- with the middle (inverse pass, product,
 * forward pass) in one launch: five launches instead of seven.  Plans in the plain layout (complex128).  Asynchronous. */
int ssfm_apply_tables_mul(ssfm_plan* plan, const void* mul_dev);
/* One whole chirp-z step in FIVE launches: ssfm_apply_tables_mul with ssfm_chirp_pre folded into its first column launch and ssfm_chirp_post into
 * its last (the caller's field A, `n` complex128 per row, is read and written directly; the plan's field buffer is not touched).
 *   A <- [ifft_n(fft_n(A exp(i gamma |A|^2 hh)) * D)] exp(i gamma |A|^2 hh)   with D = the time-domain table `mul_dev` between the two convolutions
 * P (n float64 per row) receives |A|^2 of the step's start.  h_dev (nullable, DEVICE double): hh = *h_dev / 2 instead of `hh`;  done_dev (nullable,
 * DEVICE int): the first and the last launch do nothing when it is set;  maxbits_dev (nullable, DEVICE 8 bytes): atomic maximum of the bit pattern
 * of |A|^2 after the step.  complex128 plans (SSFM_ERR_UNSUPPORTED otherwise, nothing launched).  Asynchronous on the plan's stream. */
typedef struct ssfm_chirp_io {
    void* A;
    void* P;
    const void* chirp;
    int64_t n;
    double gamma, hh;
    const void* h_dev;
    const void* done_dev;
    void* maxbits_dev;
} ssfm_chirp_io;
int ssfm_chirp_step(ssfm_plan* plan, const void* mul_dev, const ssfm_chirp_io* io);
/* A whole FIXED-step run of a field of n <= plan length / 2 samples per row (DEVICE, complex128, natural order, advanced in place) in ONE launch: a
 * workgroup per row keeps the row in registers and does the four line transforms of every step itself (k_small_chirp, csrc/ssfm_kernels.hpp).
 * chirp (n complex128, DEVICE) as ssfm_device_chirp writes it, Dt = D~ (n complex128, DEVICE, natural frequency order), hs: nsteps step sizes [km]
 * (HOST).  Complex128 plans of 256 ... 4096 samples (SSFM_ERR_UNSUPPORTED otherwise, nothing launched).  Asynchronous on the plan's stream after
 * the schedule has been copied. */
int ssfm_chirp_small(ssfm_plan* plan, void* A, const void* chirp, const void* Dt, int64_t n, double gamma, const double* hs, int64_t nsteps);
/* A whole FIXED-STEP chirp-z run on the plan's line (plans in the plain layout: complex128), four launches per step instead of ssfm_chirp_step's five: the
 * chirp products on either side of a step cancel against the neighbouring steps', so the caller takes them once -- the line holds A c (zero from `keep` =
 * the field's length up) when this is called and A c again when it returns; slots 0 and 1 hold the two convolutions' tables; mul[which[s]] (DEVICE, plan
 * length entries each, at most 256 tables) = exp(D~ h_s) / keep below `keep`, zero above; hs (HOST): the nsteps step sizes.  Asynchronous on the plan's
 * stream.  SSFM_ERR_UNSUPPORTED: a plan in the 16-byte-unit layout (nothing launched).  ssfm_chirp_propagate uses it for schedules of up to four sizes. */
int ssfm_chirp_line_run(ssfm_plan* plan, const void* const* mul, const unsigned char* which, const double* hs, int64_t nsteps, double gamma, int64_t keep);
/* The same on a complex64 plan of 2^13 ... 2^17 points whose rows fit the one-XCD engine (2^17 points in all rows, at most 64 workgroups): lengths
 * 2048 < n <= plan length / 2 in ONE launch on one XCD -- four passes per step, the chirps of neighbouring steps cancel (k_medium_chirp) -- between two
 * pointwise launches.  A, chirp, Dt: complex64, DEVICE.  At most four distinct step sizes.  Synchronous.  SSFM_ERR_UNSUPPORTED with A as it was: no such
 * plan or schedule, or the launch's workgroups did not meet within the patience (the engine is then off for this plan). */
int ssfm_chirp_medium(ssfm_plan* plan, void* A, const void* chirp, const void* Dt, int64_t n, double gamma, const double* hs, int64_t nsteps);
/* ... and its ADAPTIVE run (the rule of ssfm_chirp_small_adapt in float32 arithmetic; k_medium_chirp_adapt): z_out (HOST, nullable, max_steps + 1 entries)
 * receives z after every step, *steps_out the steps taken.  Synchronous.  SSFM_ERR_UNSUPPORTED with A as it was, as above. */
int ssfm_chirp_medium_adapt(ssfm_plan* plan, void* A, const void* chirp, const void* Dt, int64_t n, double gamma, double length, double phi_max, int64_t max_steps,
                            double* z_out, int64_t* steps_out);
/* ... and the ADAPTIVE run (h = phi_max / (|gamma| max |A|^2) over all rows, clamped to what is left of `length`; the rule of ssfm_chirp_propagate, in
 * float32 arithmetic when f32 != 0) in one launch: the rows' workgroups exchange their maxima through memory every step (k_small_chirp_adapt).  At
 * most 16 rows.  z_out (HOST, nullable, max_steps + 1 entries) receives z after every step, *steps_out the steps taken.  Synchronous.
 * SSFM_ERR_UNSUPPORTED with the field as it came: no such plan, or the rows' workgroups did not meet within the patience (SSFM_FUSED_PATIENCE_TICKS; a
 * device too busy to keep them resident together; rows that had finished by then are put back from a copy taken before the launch) --
 * ssfm_chirp_propagate then queues the run step by step. */
int ssfm_chirp_small_adapt(ssfm_plan* plan, void* A, const void* chirp, const void* Dt, int64_t n, double gamma, double length, double phi_max, int f32,
                           int64_t max_steps, double* z_out, int64_t* steps_out);
/* The chirp c_m = exp(-i pi m^2 / n) of that identity, generated on the device with its phase reduced exactly in
 * integers: ssfm_device_chirp writes c (conj = 0) or conj(c) (1), n complex128, into DEVICE memory (synchronous);
 * ssfm_load_chirp_kernel writes the convolution kernel v[m] = v[plan_n - m] = conj(c_m) (which = 0, forward transform) or
 * c_m (which = 1, inverse) into the plan's field, from where ssfm_table_from_field makes it the resident transfer
 * function of a slot -- neither a host transform nor an upload (the reference: numpy.fft, devices.py:1178-1180). */
int ssfm_device_chirp(int device, void* out_dev, int64_t n, int conj);
int ssfm_load_chirp_kernel(ssfm_plan* plan, int64_t plan_n, int64_t n, int which);
/* Pulse shaping of the reference's DAC (upfir, utils.py:1949-1981) on a complex128 plan of batch 1: ssfm_load_padded
 * writes `n_src` float64 (src_complex = 0) or complex128 (1) samples from DEVICE memory into the field, zero-padded;
 * ssfm_table_from_field makes slot <- fft(field) (the field is consumed); ssfm_load_symbols writes `nsym` float64
 * amplitudes (DEVICE; the bits as 0.0 / 1.0) zero-stuffed to `up` samples per symbol with the sample at up / 2; ssfm_apply_table then convolves.  Asynchronous. */
int ssfm_load_padded(ssfm_plan* plan, int64_t plan_n, const void* src_dev, int src_complex, int64_t n_src);
int ssfm_load_symbols(ssfm_plan* plan, int64_t plan_n, const double* sym_dev, int64_t nsym, int up);
/* The DAC's built-in pulses (utils.py:1791-1947) generated in the field instead of uploaded: `npts` samples over
 * t_i = i*step + start (t_{npts-1} = stop: numpy.linspace), zero-padded.  kind 0 nrz (params: -T/2, T/2); 1 gaussian
 * exp(-((p0 + j p1) t)^pow2m) with (p0, p1) = alpha (1 + j c); 2 raised cosine (2 beta, pi beta, value where
 * 1 - (2 beta t)^2 ~ 0); 3 root raised cosine (beta, 4 beta, 1 - beta, 1 + beta, 1/(4 beta), value at 0, value at
 * 1/(4 beta)); 4 sinc.  `params`: 7 doubles on the HOST.  Asynchronous. */
int ssfm_load_pulse(ssfm_plan* plan, int64_t plan_n, int kind, int64_t npts, double start, double step, double stop, int pow2m, const double* params);
int ssfm_table_from_field(ssfm_plan* plan, int slot);
int ssfm_chirp_pre(ssfm_plan* plan, int64_t plan_n, int batch, const void* A, void* P, const void* chirp, int64_t n, double gamma, double hh);
int ssfm_chirp_mid(ssfm_plan* plan, int64_t plan_n, int batch, const void* tab, int64_t n, double h, int mode);
int ssfm_chirp_post(ssfm_plan* plan, int64_t plan_n, int batch, void* A, const void* P, const void* chirp, int64_t n, double gamma, double hh,
                    void* maxbits_dev);
/* A whole FIBER / DBP run on a field of ANY length through the three kernels above and ssfm_apply_table, driven from C (the per-kernel entry
 * points cost a host call each, and an adaptive run driven from the host waits for every step's maximum): A (batch x n complex128, DEVICE) is
 * advanced in place, P is scratch (batch x n float64, DEVICE), chirp / Dt as for ssfm_chirp_pre / _mid; slots 0 and 1 of the plan hold the
 * chirp kernels.  hs != NULL: fixed step, `nsteps` step sizes (HOST).  hs == NULL: the adaptive rule of devices.py:1172-1196,
 * h = phi_max / (|gamma| max|A|^2) clamped to the rest of `length`, evaluated on the device in float32 (f32 != 0: the reference's arithmetic
 * in complex64 mode) or float64; z_out (HOST, max_steps + 1 doubles, nullable) receives z after every step, *steps_out the steps taken.
 * Synchronous. */
int ssfm_chirp_propagate(ssfm_plan* plan, int64_t plan_n, int batch, void* A, void* P, const void* chirp, const void* Dt, int64_t n, double gamma,
                         const double* hs, int64_t nsteps, double length, double phi_max, int f32, int64_t max_steps, double* z_out, int64_t* steps_out);

/* Forward FFT of every row into HOST `dst` (natural frequency order, unscaled) -- validation aid. */
int ssfm_debug_fft(ssfm_plan* plan, void* dst);

int ssfm_synchronize(ssfm_plan* plan);
/* The plan's hipStream_t (as void*), so callers can record events around propagate calls or order their own work behind a run (see
 * "WHEN THE FIELD IS VALID" at ssfm_field_device_ptr). */
void* ssfm_stream(ssfm_plan* plan);

/* Which engine the plan's last run took, and whether it had to be repeated on a fallback.  The engines differ in speed only (same results
 * within the stated tolerances); the single-launch ones need the GPU to run all their workgroups side by side and fall back, after a bounded
 * wait, when it does not (a shared GPU, a profiler): without this query that is invisible to the caller.
 *   engine            a value of enum ssfm_engine below: the engine that produced the result of the LAST run (after a fallback: the fallback)
 *   fell_back         1: the last run was started on a single-launch engine, gave up and was repeated
 *   fallbacks_total   such repeats over the life of the plan (a plan keeps to the fallback engine after the first)
 *   lanes             lanes the plan's fixed-step runs drive now: ssfm_num_lanes, or 1 once the plan has dropped its second lane (below)
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
    SSFM_ENGINE_CHIRP_STEPS = 10,      /* any length: four launches per fixed step (five through ssfm_chirp_step), seven adaptive */
    SSFM_ENGINE_CHIRP_MEDIUM = 11,     /* any length, 2048 < n <= 65536, complex64: fixed step, one launch per run on one XCD */
    SSFM_ENGINE_CHIRP_MEDIUM_ADAPT = 12 /* ... adaptive */
};
typedef struct ssfm_run_info {
    int engine;
    int fell_back;
    int64_t fallbacks_total;
    int lanes;
    int lanes_share_queue;
    int lanes_remade;
    int lanes_dropped;
    int lane_heals;
    float lane_alone_us;
    float lane_pair_us;
    float lane_last_us;
    float lane_score;
} ssfm_run_info;
int ssfm_last_run_info(ssfm_plan* plan, ssfm_run_info* info, size_t info_bytes);
/* Test hook of the lane health check: mode 1 makes the plan believe a four times better launch period than it has seen (the next long two-lane run looks
 * slow and the lanes are rated again), mode 2 additionally makes every rating come out bad (the repair fails; the second failure drops the plan to one
 * lane), 0 back to normal. */
int ssfm_debug_lane_fault(ssfm_plan* plan, int mode);

/* A device buffer of at least `bytes` bytes owned by the plan (slot 0 ... 3; grows on demand, freed with the plan; contents undefined between
 * calls; a growing call waits for the plan's stream).  For driver loops above this ABI that need scratch memory per call
 * (ssfm_chirp_propagate: its exp(D~ h) table, step control block and z log) without a hipMalloc / hipFree pair each time. */
int ssfm_plan_workspace(ssfm_plan* plan, int slot, size_t bytes, void** out);

/* Time of the last propagate call measured with HIP events on the plan's stream [ms], and the
 * number of kernel launches it made.  Valid after ssfm_synchronize. */
int ssfm_last_propagate_ms(ssfm_plan* plan, float* ms, int64_t* launches);

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
int ssfm_set_profiling(ssfm_plan* plan, int mode);
/* Number of row groups ("lanes") a fixed-step run drives on separate streams (env SSFM_LANES,
 * default 2, never more than the batch): one kernel launch covers batch/lanes rows. */
int ssfm_num_lanes(ssfm_plan* plan, int* lanes);
/* What the plan's staging buffers hold, owned by the plan: `which` 0 labels the linear operator set with
 * ssfm_set_linear_operator, 1 / 2 the resident transfer function of slot 0 / 1.  A caller labels what it has staged
 * (any non-zero 64-bit tag, e.g. a hash of the fibre parameters) and asks later whether it is still there; EVERY entry
 * point that overwrites or reuses the buffer (a new operator, ssfm_apply_transfer, ssfm_apply_dispersion with H_out,
 * ssfm_transfer_table, ssfm_table_from_field) clears the label, so a cached operator can never be stale.  0 = unknown.
 * (The reference recomputes D~ on every call, devices.py:1137-1145; this only saves the O(N) set-up of a repeated call.) */
int ssfm_plan_set_tag(ssfm_plan* plan, int which, uint64_t tag);
int ssfm_plan_get_tag(ssfm_plan* plan, int which, uint64_t* tag);
int ssfm_kernel_times(ssfm_plan* plan, int64_t counts[2], double total_ms[2]);

#ifdef __cplusplus
}
#endif
#endif /* SSFM_AMD_H */
