/* The C ABI of include/ssfm_amd.h from plain C: a Gaussian pulse through 20 km of fibre (fixed 0.5 km steps) and back
 * (DBP = the negated operators), printed as energies; then the adaptive step in budgeted blocks (ssfm_adaptive_begin /
 * _run / _finish), the plan's own record of what its staging buffers hold (ssfm_plan_set_tag / _get_tag) and the
 * reference's PRBS generator on the device (ssfm_prbs).  No Python, no torch -- what a binding in any language does.
 *
 *   gcc -O2 -Iinclude examples/c_abi_demo.c -o /tmp/c_abi_demo -L opticomlib_amd -l:_ssfm_amd.so -lm -Wl,-rpath,$PWD/opticomlib_amd
 *   /tmp/c_abi_demo
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "ssfm_amd.h"

#define CHECK(call)                                                        \
    do {                                                                   \
        int rc_ = (call);                                                  \
        if (rc_ != SSFM_OK) {                                              \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, ssfm_last_error()); \
            return 1;                                                      \
        }                                                                  \
    } while (0)

static double energy(const float* f, long n) {
    double e = 0;
    for (long i = 0; i < 2 * n; ++i) e += (double)f[i] * f[i];
    return e;
}

int main(void) {
    const long n = 1 << 14;
    const double dt = 1.953125e-12, pi = 3.14159265358979323846;
    const float alpha = 0.2f / 4.343f, beta2 = -21.7f, gamma = 1.3f;        /* 1/km, ps^2/km, 1/(W km) */
    const int steps = 40;
    float *field = malloc(sizeof(float) * 2 * n), *back = malloc(sizeof(float) * 2 * n), *D = malloc(sizeof(float) * 2 * n), h[40];
    int count = 0;
    CHECK(ssfm_device_count(&count));
    printf("ABI version %d, %d device(s)\n", ssfm_abi_version(), count);
    for (long i = 0; i < n; ++i) {                                          /* 10 ps Gaussian pulse, 50 mW peak */
        const double t = (i - n / 2) * dt / 10e-12;
        field[2 * i] = (float)(sqrt(0.05) * exp(-0.5 * t * t));
        field[2 * i + 1] = 0.0f;
        const long k = i < n / 2 ? i : i - n;                               /* FFT order */
        const float w = (float)(2 * pi * k / (n * dt) * 1e-12);             /* rad/ps */
        D[2 * i] = -alpha / 2;
        D[2 * i + 1] = 0.5f * beta2 * w * w;
    }
    for (int s = 0; s < steps; ++s) h[s] = 0.5f;
    ssfm_plan* plan = NULL;
    CHECK(ssfm_plan_create(&plan, 0, n, 1, SSFM_C64));
    CHECK(ssfm_set_linear_operator(plan, D));
    CHECK(ssfm_set_field(plan, field, 0));
    CHECK(ssfm_propagate_fixed(plan, gamma, h, steps, NULL));
    CHECK(ssfm_get_field(plan, back, 0));
    const double e_in = energy(field, n), e_out = energy(back, n);
    printf("energy in %.6e, after 20 km %.6e (ratio %.6f, exp(-alpha L) = %.6f)\n", e_in, e_out, e_out / e_in, exp(-(double)alpha * 20.0));
    for (long i = 0; i < n; ++i) { D[2 * i] = -D[2 * i]; D[2 * i + 1] = -D[2 * i + 1]; }     /* DBP */
    CHECK(ssfm_set_linear_operator(plan, D));
    CHECK(ssfm_propagate_fixed(plan, -gamma, h, steps, NULL));
    CHECK(ssfm_get_field(plan, back, 0));
    double err = 0, peak = 0;
    for (long i = 0; i < 2 * n; ++i) {
        const double d = fabs((double)back[i] - field[i]);
        if (d > err) err = d;
        if (fabs(field[i]) > peak) peak = fabs(field[i]);
    }
    printf("back-propagated: max deviation from the input %.3e of the peak\n", err / peak);

    /* the plan knows what it holds: a label survives until the buffer is overwritten */
    uint64_t tag = 0;
    CHECK(ssfm_plan_set_tag(plan, 0, 0x5eedULL));
    CHECK(ssfm_plan_get_tag(plan, 0, &tag));
    printf("operator label after set: %#llx", (unsigned long long)tag);
    for (long i = 0; i < n; ++i) { D[2 * i] = -D[2 * i]; D[2 * i + 1] = -D[2 * i + 1]; }     /* the forward fibre again */
    CHECK(ssfm_set_linear_operator(plan, D));
    CHECK(ssfm_plan_get_tag(plan, 0, &tag));
    printf(", after a new operator: %#llx\n", (unsigned long long)tag);

    /* adaptive step (devices.py:1155-1196 with h = None), 8 steps per call */
    CHECK(ssfm_set_field(plan, field, 0));
    CHECK(ssfm_adaptive_begin(plan, gamma, 20.0, 0.05, 0, 1 << 16, 0));
    int64_t taken = 0;
    int done = 0, calls = 0;
    while (!done) { CHECK(ssfm_adaptive_run(plan, 8, NULL, &taken, &done)); ++calls; }
    double* z = malloc(sizeof(double) * (taken + 1));
    CHECK(ssfm_adaptive_finish(plan, &taken, z));
    CHECK(ssfm_get_field(plan, back, 0));
    printf("adaptive: %lld steps in %d calls, z_end = %.6f km, energy ratio %.6f\n", (long long)taken, calls, z[taken], energy(back, n) / e_in);
    const int adaptive_ok = taken > 1 && fabs(z[taken] - 20.0) < 1e-4 && fabs(energy(back, n) / e_in - exp(-(double)alpha * 20.0)) < 1e-3;
    free(z);
    /* which engine the run took, and whether a single-launch engine had to give way to its fallback (a shared GPU, a profiler) */
    ssfm_run_info info;
    CHECK(ssfm_last_run_info(plan, &info, sizeof(info)));
    printf("last run: engine %d, fell back %d, fallbacks of this plan %lld, lanes %d (share a queue %d, remade %d)\n", info.engine, info.fell_back,
           (long long)info.fallbacks_total, info.lanes, info.lanes_share_queue, info.lanes_remade);
    CHECK(ssfm_plan_destroy(plan));

    /* PRBS-7 from the all-ones state: the reference's first 20 bits (tests/devices_test.py:52-71) */
    void* bits_dev = NULL;
    unsigned char bits[20];
    uint32_t last = 0;
    CHECK(ssfm_device_alloc(0, 20, &bits_dev));
    CHECK(ssfm_prbs(0, bits_dev, 20, 7, 0x7f, &last));
    CHECK(ssfm_device_copy(0, bits, bits_dev, 20, 1));
    CHECK(ssfm_device_free(0, bits_dev, 20));
    printf("PRBS-7: ");
    for (int i = 0; i < 20; ++i) printf("%d", bits[i]);
    printf(" (register afterwards %#x)\n", last);
    if (!adaptive_ok) return 3;
    free(field); free(back); free(D);
    return e_out > 0 && err / peak < 0.05 ? 0 : 2;
}
