// b4d_colpass.hip -- the fused column pass (forward column FFT, PSD, inverse column FFT) of the cfg2 pipeline in a translation
// unit of its own: see b4d_passes.hpp for the scheduler it is built with.
#define B4D_UNIT_TAG 1   // kernels launched from this unit are instantiations of their own (b4d_fft2d.hpp)
#define B4D_UNIT_PASSES 2   // B4D_PASS_COL only: this unit compiles no other kernel of b4d_fft2d.hpp
#include "b4d_passes.hpp"

namespace b4d {
int col_psd_ac_pass(const b4d_plan* pl, const ColArgs& a, int batch, hipStream_t st) {
    return dispatch_col<COL_PSD_AC>(pl, a, batch, st);
}
}  // namespace b4d
