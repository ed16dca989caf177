// b4d_passes.hpp -- the two passes of the FFT -> PSD -> autocorrelation pipeline that are built in translation units of
// their own, because each wants a different instruction scheduler of the AMDGPU back end (csrc/Makefile):
//   b4d_colpass.hip  k_col<.., COL_PSD_AC>  GCN trackers for the register-pressure bookkeeping of the scheduler
//                    (-amdgpu-use-amdgpu-trackers): same 128-VGPR budget, K2 -1.0 ... -1.6 % in shared-plan A/B runs
//   b4d_rowout.hip   k_row_c2r<.., C2R_OUT> (and its C2R_PEAK pre-pass)  max-ILP strategy (-amdgpu-sched-strategy=max-ilp):
//                    134 VGPRs, three waves per SIMD instead of four, K3 -3 ... -4 %
// Either flag on the whole of b4d_kernels.hip costs the other kernels what it wins (max-ILP: K1 +1 %, K2 +0.5 %; both together
// spill in k_col).  Outputs are bit-identical to the default scheduler's.
#pragma once
#include <vector>

#include "b4d_fft2d.hpp"

namespace b4d {
int col_psd_ac_pass(const b4d_plan* pl, const ColArgs& a, int batch, hipStream_t st);
int row_out_pass(const b4d_plan* pl, const RowOutArgs& a, int batch, hipStream_t st, std::vector<hipEvent_t>* ev);
}  // namespace b4d
