// The four-lanes-per-game step kernels hk::quad_kernel<HK_SPEC_M, HK_SPEC_D, *> of one shape per object file
// (Makefile: QUAD_SPECS must list the entries of HK_QUAD_SPECS in hk_quad_kernel.h).
#define HK_SPEC_TU 1
#include "hk_quad_kernel.h"

namespace hk {
static_assert(HK_SPEC_M * HK_SPEC_D > 0, "build with -DHK_SPEC_M=<max_points> -DHK_SPEC_D=<dim>");
template int launch_quad_t<HK_SPEC_M, HK_SPEC_D>(Params, hipStream_t);
}  // namespace hk
