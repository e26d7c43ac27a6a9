// The two kernels of the headline configuration alone, for quick ISA inspection:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Iopticomlib_amd/csrc --cuda-device-only -S -o hot.s tools/hot_kernels.hip [-D...]
#include "ssfm_kernels.hpp"
using namespace ssfm;
template __global__ void ssfm::k_time<float, 256, 16, 16, TM_MID, true>(const TimeArgs<float>);
template __global__ void ssfm::k_freq<float, 4096, 1, 16, FM_PHASE, true>(const FreqArgs<float>);
#ifdef HOT_ADAPT
template __global__ void ssfm::k_time<float, 256, 16, 16, TM_MID_A, true>(const TimeArgs<float>);
template __global__ void ssfm::k_freq<float, 4096, 1, 16, FM_FLY, true>(const FreqArgs<float>);
#endif
#ifdef HOT_C128
template __global__ void ssfm::k_time<double, 256, 8, 8, TM_MID, false>(const TimeArgs<double>);
template __global__ void ssfm::k_freq<double, 4096, 1, 16, FM_TABLE, false>(const FreqArgs<double>);
#endif
