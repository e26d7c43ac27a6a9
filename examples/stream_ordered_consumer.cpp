// A C-ABI caller that orders its OWN work on the plan's stream behind ssfm_propagate_fixed, without ssfm_synchronize
// (include/ssfm_amd.h, "WHEN THE FIELD IS VALID").  The plan is one of those whose fixed-step run is ONE launch whose workgroups
// meet inside the kernel (2^15 x 2 complex64); with SSFM_FUSED_PATIENCE_TICKS=-1 they give up at their first meeting, as they would
// on a GPU that does not run them side by side.  The copy queued right behind the call must hold the two-kernel engine's result.
//
//   hipcc -O2 -Iinclude examples/stream_ordered_consumer.cpp -o /tmp/soc -L opticomlib_amd -l:_ssfm_amd.so -Wl,-rpath,$PWD/opticomlib_amd && /tmp/soc
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ssfm_amd.h"

#define CHECK(call)                                                                            \
    do {                                                                                       \
        int rc_ = (call);                                                                      \
        if (rc_ != 0) { std::fprintf(stderr, "%s -> %d: %s\n", #call, rc_, ssfm_last_error()); return 1; } \
    } while (0)

int main() {
    const long n = 1 << 15, rows = 2, steps = 24;
    const double dt = 1.953125e-12, pi = 3.14159265358979323846;
    std::vector<float> field(2 * n * rows), D(2 * n), want(2 * n * rows), got(2 * n * rows), h(steps, 0.25f);
    unsigned s = 12345u;
    for (auto& v : field) { s = s * 1664525u + 1013904223u; v = ((s >> 8) / 16777216.0f - 0.5f) * 0.1f; }
    for (long k = 0; k < n; ++k) {                                       // D~ = -alpha/2 + i beta2/2 w^2 (float32, devices.py:1137-1145)
        const double f = (k < n / 2 ? k : k - n) / (n * dt);
        const float w = (float)(2.0 * pi * f * 1e-12);
        D[2 * k] = -0.5f * (0.2f / 4.343f);
        D[2 * k + 1] = 0.5f * -21.7f * w * w;
    }
    // the reference: the two-kernel engine
    setenv("SSFM_MEDIUM", "0", 1);
    ssfm_plan* ref = nullptr;
    CHECK(ssfm_plan_create(&ref, 0, n, (int)rows, SSFM_C64));
    CHECK(ssfm_set_linear_operator(ref, D.data()));
    CHECK(ssfm_set_field(ref, field.data(), 0));
    CHECK(ssfm_propagate_fixed(ref, 1.3, h.data(), steps, nullptr));
    CHECK(ssfm_get_field(ref, want.data(), 0));
    CHECK(ssfm_plan_destroy(ref));
    // the one-launch engine without patience, consumed stream-ordered
    setenv("SSFM_MEDIUM", "1", 1);
    setenv("SSFM_FUSED_PATIENCE_TICKS", "-1", 1);
    ssfm_plan* plan = nullptr;
    CHECK(ssfm_plan_create(&plan, 0, n, (int)rows, SSFM_C64));
    hipStream_t stream = static_cast<hipStream_t>(ssfm_stream(plan));    // from here on the plan serves a stream-ordered consumer
    void* dev = ssfm_field_device_ptr(plan);
    CHECK(ssfm_set_linear_operator(plan, D.data()));
    CHECK(ssfm_set_field(plan, field.data(), 0));
    CHECK(ssfm_propagate_fixed(plan, 1.3, h.data(), steps, nullptr));
    float* pinned = nullptr;
    if (hipHostMalloc(reinterpret_cast<void**>(&pinned), sizeof(float) * got.size(), 0) != hipSuccess) return 2;
    if (hipMemcpyAsync(pinned, dev, sizeof(float) * got.size(), hipMemcpyDeviceToHost, stream) != hipSuccess) return 2;   // the consumer's work
    if (hipStreamSynchronize(stream) != hipSuccess) return 2;                                                            // ... and its own wait
    ssfm_run_info info;
    CHECK(ssfm_last_run_info(plan, &info, sizeof(info)));
    const int engine = info.engine, fell_back = info.fell_back;
    const int64_t fallbacks = info.fallbacks_total;
    const bool same = std::memcmp(pinned, want.data(), sizeof(float) * got.size()) == 0;
    std::printf("engine %d fell_back %d fallbacks %lld; stream-ordered copy %s the two-kernel result\n", engine, fell_back, (long long)fallbacks,
                same ? "equals" : "DIFFERS FROM");
    (void)hipHostFree(pinned);
    CHECK(ssfm_plan_destroy(plan));
    return same && engine == SSFM_ENGINE_TWO_KERNEL && fell_back == 1 && fallbacks == 1 ? 0 : 1;
}
