// ssfm_medium.hpp -- launcher of the single-launch engine of medium plans (ssfm_medium.hip)
#pragma once
#include "ssfm_kernels.hpp"

namespace ssfm {
bool medium_shape(int N1, int N2);
// complex64 plans in the 16-byte-unit layout, 8 points per thread, nblk = (N2 / 16) * rows workgroups (a multiple of 8, at most 64)
hipError_t launch_medium(int N1, int N2, bool phase_tables, int nblk, hipStream_t s, const MediumArgs<float>& a);
}  // namespace ssfm
