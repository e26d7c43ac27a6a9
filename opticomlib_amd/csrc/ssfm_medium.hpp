// ssfm_medium.hpp -- launcher of the single-launch engine of medium plans (ssfm_medium.hip)
#pragma once
#include "ssfm_kernels.hpp"

namespace ssfm {
bool medium_shape(int N1, int N2);
// complex64 plans in the 16-byte-unit layout, 8 points per thread, nblk = (N2 / 16) * rows workgroups (a multiple of 8, at most 64)
// (launched with `xccs` x nblk workgroups when the engine keeps to one XCD: ssfm_kernels.hpp medium_ticket)
hipError_t launch_medium(int N1, int N2, bool phase_tables, int nblk, int xccs, hipStream_t s, const MediumArgs<float>& a);
// the adaptive run of such a plan in one launch (k_medium_adapt)
hipError_t launch_medium_adapt(int N1, int N2, int nblk, int xccs, hipStream_t s, const MediumAdaptArgs<float>& a);
// the fixed-step chirp-z run on such a plan in one launch (k_medium_chirp)
hipError_t launch_medium_chirp(int N1, int N2, int nblk, int xccs, hipStream_t s, const MediumChirpArgs<float>& a);
// ... and its adaptive run (k_medium_chirp_adapt)
hipError_t launch_medium_chirp_adapt(int N1, int N2, int nblk, int xccs, hipStream_t s, const MediumChirpAdaptArgs<float>& a);
// The XCC ids (HW_REG_XCC_ID) the workgroups of a launch on `device` are dealt to, as a bit mask: 0xff on an MI355X in SPX mode, 1 when
// every XCD is a device of its own.  Found once per device by a probe launch (synchronous); 0 on error.
unsigned xcc_mask(int device);
}  // namespace ssfm
