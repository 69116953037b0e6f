// The pool rollout kernels hk::pool_kernel<HK_SPEC_M, HK_SPEC_D, *> of one shape per object file (Makefile: the same
// table as hk_fast_spec.hip).
#define HK_SPEC_TU 1
#include "hk_pool_kernel.h"

namespace hk {
static_assert(HK_SPEC_M * HK_SPEC_D > 0, "build with -DHK_SPEC_M=<max_points> -DHK_SPEC_D=<dim>");
template int launch_pool_t<HK_SPEC_M, HK_SPEC_D>(Params, hipStream_t);
}  // namespace hk
