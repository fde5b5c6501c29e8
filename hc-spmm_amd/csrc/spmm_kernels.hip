// spmm_kernels.hip -- fp32 instantiations of the hybrid SpMM kernels (spmm_impl.h holds the device code).
#include "spmm_impl.h"

namespace hcspmm {

// vec = elements per lane access: 4 for every width of at least 4 columns (any row stride, any base address: fp32 vectors
// are addressed with element alignment, spmm_impl.h MemF32 / lane_col); 2 and 1 serve D = 2, 3 and D = 1 only, where the
// four lanes of the smallest lane group cover the row.
hipError_t launch_plan_f32(const PlanArgs& a, int vec, hipStream_t stream) {
  if (vec == 4) { HCSPMM_DISPATCH_L(launch_plan_LV, F32, 4, a.panel_cols, a, stream) }
  if (a.D > 4 * vec) return hipErrorInvalidValue;
  if (vec == 2) return launch_plan_LV<F32, 4, 2>(a, stream);
  return launch_plan_LV<F32, 4, 1>(a, stream);
}

hipError_t launch_window_f32(const WindowArgs& a, int vec, hipStream_t stream) {
  if (vec == 4) { HCSPMM_DISPATCH_L(launch_window_LV, F32, 4, a.D, a, stream) }
  if (a.D > 4 * vec) return hipErrorInvalidValue;
  if (vec == 2) return launch_window_LV<F32, 4, 2>(a, stream);
  return launch_window_LV<F32, 4, 1>(a, stream);
}

}  // namespace hcspmm
