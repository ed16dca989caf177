// b4d_rowout.hip -- the inverse row pass (C2R_OUT, with its zero-lag pre-pass) of the cfg2 pipeline in a translation unit of
// its own: see b4d_passes.hpp for the scheduler it is built with.
#define B4D_UNIT_TAG 2   // kernels launched from this unit are instantiations of their own (b4d_fft2d.hpp)
#define B4D_UNIT_PASSES 8   // B4D_PASS_C2R only: this unit compiles no other kernel of b4d_fft2d.hpp
#include "b4d_passes.hpp"

namespace b4d {
int row_out_pass(const b4d_plan* pl, const RowOutArgs& a, int batch, hipStream_t st, std::vector<hipEvent_t>* ev) {
    return dispatch_c2r(pl, a, batch, st, C2R_OUT, ev);
}
}  // namespace b4d
