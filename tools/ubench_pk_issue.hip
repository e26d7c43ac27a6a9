#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k(float* out, int iters) {
    v2f a[8], w = {1.0001f, 0.9999f};
    for (int i = 0; i < 8; ++i) a[i] = v2f{(float)threadIdx.x + i, 1.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) {        // packed fma, independent chains
                asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(w));
            } else if (MODE == 1) { // two scalar fma
                asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i].x) : "v"(w.x));
                asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i].y) : "v"(w.y));
            } else if (MODE == 2) { // packed add
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(w));
            } else if (MODE == 3) { // packed mul with op_sel
                asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[1,1] op_sel_hi:[1,0]" : "+v"(a[i]) : "v"(w));
            } else {                // scalar add x2
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(w.x));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i].y) : "v"(w.y));
            }
        }
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, int threads, int blocks, float* d) {
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, 100);
    hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // instructions per wave: iters*8*(1 or 2); cycles at 2.4 GHz
    double ninstr = (double)iters * 8 * ((MODE == 1 || MODE == 4) ? 2 : 1);
    printf("%-22s threads/block %4d blocks %4d: %.3f ms -> %.2f cycles per wave-instruction (at 2.4 GHz)\n", name, threads, blocks, ms, ms * 1e-3 * 2.4e9 / ninstr);
}
int main() {
    float* d; hipMalloc(&d, 1 << 24);
    for (int threads : {256, 512, 1024}) {   // 1, 2, 4 waves per SIMD with one block per CU
        run<0>("v_pk_fma_f32", threads, 256, d);
        run<1>("2 x v_fma_f32", threads, 256, d);
        run<2>("v_pk_add_f32", threads, 256, d);
        run<3>("v_pk_mul_f32 op_sel", threads, 256, d);
        run<4>("2 x v_add_f32", threads, 256, d);
    }
    return 0;
}
