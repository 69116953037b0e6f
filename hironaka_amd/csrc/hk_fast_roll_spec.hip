// The plain rollout kernels hk::fast_kernel<HK_SPEC_M, HK_SPEC_D, kModeRollout, *> of one shape per object file (the
// other modes of the shape: hk_fast_spec.hip).
#define HK_SPEC_TU 1
#define HK_FAST_ROLL_TU 1
#include "hk_fast_kernel.h"

namespace hk {
static_assert(HK_SPEC_M * HK_SPEC_D > 0, "build with -DHK_SPEC_M=<max_points> -DHK_SPEC_D=<dim>");
template int launch_fast_roll_t<HK_SPEC_M, HK_SPEC_D>(const Params&, unsigned, hipStream_t);
}  // namespace hk
