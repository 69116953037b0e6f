// The four-lanes-per-game rollout kernels hk::quadroll_kernel<HK_SPEC_M, HK_SPEC_D, *> of one shape per object file
// (Makefile: QUAD_SPECS, the table of hk_quad_spec.hip).
#define HK_SPEC_TU 1
#include "hk_quadroll_kernel.h"

namespace hk {
static_assert(HK_SPEC_M * HK_SPEC_D > 0, "build with -DHK_SPEC_M=<max_points> -DHK_SPEC_D=<dim>");
template int launch_quadroll_t<HK_SPEC_M, HK_SPEC_D>(Params, hipStream_t);
template int launch_quadzeil_t<HK_SPEC_M, HK_SPEC_D>(Params, hipStream_t);
}  // namespace hk
