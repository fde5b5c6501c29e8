// spmm_kernels_h16.hip -- fp16 / bf16 feature instantiations of the hybrid SpMM kernels (spmm_impl.h):
// 16-bit X and Z, fp32 accumulation in the fp32 path's order, one rounding (RNE) per output element.
// vec = elements per lane access: 8 (16-byte loads), 4 or 1.
#include "spmm_impl.h"

namespace hcspmm {

template <typename E>
static hipError_t plan16(const PlanArgs& a, int vec, hipStream_t stream) {
  if (vec == 8) { HCSPMM_DISPATCH_L(launch_plan_LV, E, 8, a.panel_cols, a, stream) }
  if (vec == 4) { HCSPMM_DISPATCH_L(launch_plan_LV, E, 4, a.panel_cols, a, stream) }
  HCSPMM_DISPATCH_L(launch_plan_LV, E, 1, a.panel_cols, a, stream)
}

template <typename E>
static hipError_t window16(const WindowArgs& a, int vec, hipStream_t stream) {
  if (vec == 8) { HCSPMM_DISPATCH_L(launch_window_LV, E, 8, a.D, a, stream) }
  if (vec == 4) { HCSPMM_DISPATCH_L(launch_window_LV, E, 4, a.D, a, stream) }
  HCSPMM_DISPATCH_L(launch_window_LV, E, 1, a.D, a, stream)
}

hipError_t launch_plan_f16(const PlanArgs& a, int vec, hipStream_t stream) { return plan16<F16>(a, vec, stream); }
hipError_t launch_plan_bf16(const PlanArgs& a, int vec, hipStream_t stream) { return plan16<BF16>(a, vec, stream); }
hipError_t launch_window_f16(const WindowArgs& a, int vec, hipStream_t stream) { return window16<F16>(a, vec, stream); }
hipError_t launch_window_bf16(const WindowArgs& a, int vec, hipStream_t stream) { return window16<BF16>(a, vec, stream); }

}  // namespace hcspmm
