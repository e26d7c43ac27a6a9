/* The C ABI of include/ssfm_amd.h from plain C: a Gaussian pulse through 20 km of fibre (fixed 0.5 km steps) and back
 * (DBP = the negated operators), printed as energies.  No Python, no torch -- what a binding in any language does.
 *
 *   gcc -O2 -Iinclude examples/c_abi_demo.c -o /tmp/c_abi_demo -L opticomlib_amd -l:_ssfm_amd.so -lm -Wl,-rpath,$PWD/opticomlib_amd
 *   /tmp/c_abi_demo
 */
#include <math.h>
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
    CHECK(ssfm_plan_destroy(plan));
    free(field); free(back); free(D);
    return e_out > 0 && err / peak < 0.05 ? 0 : 2;
}
