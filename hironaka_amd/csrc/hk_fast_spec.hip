// One register-resident specialisation hk::fast_kernel<HK_SPEC_M, HK_SPEC_D, *> per object file
// (Makefile: -DHK_SPEC_M=.. -DHK_SPEC_D=.. for every entry of HK_FAST_SPECS), so that the shapes build
// in parallel.
#define HK_SPEC_TU 1
#include "hk_fast_kernel.h"

namespace hk {
static_assert(HK_SPEC_M * HK_SPEC_D > 0, "build with -DHK_SPEC_M=<max_points> -DHK_SPEC_D=<dim>");
template int launch_fast_t<HK_SPEC_M, HK_SPEC_D>(Params, hipStream_t);
}  // namespace hk
