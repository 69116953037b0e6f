// The binning kernels hk::quadbin_kernel<HK_SPEC_M, HK_SPEC_D, *> of one shape per object file (Makefile: QUAD_SPECS).
#define HK_SPEC_TU 1
#include "hk_quadbin_kernel.h"

namespace hk {
static_assert(HK_SPEC_M * HK_SPEC_D > 0, "build with -DHK_SPEC_M=<max_points> -DHK_SPEC_D=<dim>");
template int launch_quadbin_t<HK_SPEC_M, HK_SPEC_D>(Params, const float*, int32_t*, int32_t*, hipStream_t);
}  // namespace hk
