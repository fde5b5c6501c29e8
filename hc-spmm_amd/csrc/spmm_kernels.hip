// spmm_kernels.hip -- fp32 instantiations of the hybrid SpMM kernels (spmm_impl.h holds the device code).
#include "spmm_impl.h"

namespace hcspmm {

// vec = elements per lane access (4, 2 or 1): the caller guarantees D, ldx, ldz % vec == 0 and the alignment.
hipError_t launch_plan_f32(const PlanArgs& a, int vec, hipStream_t stream) {
  if (vec == 4) { HCSPMM_DISPATCH_L(launch_plan_LV, F32, 4, a.panel_cols, a, stream) }
  if (vec == 2) { HCSPMM_DISPATCH_L(launch_plan_LV, F32, 2, a.panel_cols, a, stream) }
  HCSPMM_DISPATCH_L(launch_plan_LV, F32, 1, a.panel_cols, a, stream)
}

hipError_t launch_window_f32(const WindowArgs& a, int vec, hipStream_t stream) {
  if (vec == 4) { HCSPMM_DISPATCH_L(launch_window_LV, F32, 4, a.D, a, stream) }
  if (vec == 2) { HCSPMM_DISPATCH_L(launch_window_LV, F32, 2, a.D, a, stream) }
  HCSPMM_DISPATCH_L(launch_window_LV, F32, 1, a.D, a, stream)
}

}  // namespace hcspmm
