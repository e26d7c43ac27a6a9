// The two kernels of the headline configuration alone, for quick ISA inspection:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Iopticomlib_amd/csrc -mllvm -amdgpu-kernarg-preload-count=16 --cuda-device-only -S -o hot.s tools/hot_kernels.hip [-D...]
#include "ssfm_kernels.hpp"
using namespace ssfm;
#define TARGS(T) cx<T>*, T*, const cx<T>*, const cx<T>*, const cx<T>*, int, int, int, T, T, T, const TimeArgsCold<T>
#define FARGS(T) cx<T>*, const cx<T>*, const cx<T>*, const AdaptState<T>*, T, T, T, int, int, int, int
template __global__ void ssfm::k_time<float, 256, 16, 16, TM_MID, true>(TARGS(float));
template __global__ void ssfm::k_freq<float, 4096, 1, 16, FM_PHASE, true>(FARGS(float));
#ifdef HOT_ADAPT
template __global__ void ssfm::k_time<float, 256, 16, 16, TM_MID_A, true>(TARGS(float));
template __global__ void ssfm::k_freq<float, 4096, 1, 16, FM_FLY, true>(FARGS(float));
#endif
#ifdef HOT_C128
template __global__ void ssfm::k_time<double, 256, 8, 8, TM_MID, false>(TARGS(double));
template __global__ void ssfm::k_freq<double, 4096, 1, 16, FM_TABLE, false>(FARGS(double));
#endif
#ifdef HOT_C128
template __global__ void ssfm::k_freq<double, 4096, 1, 16, FM_PHASE, false>(FARGS(double));
#endif
