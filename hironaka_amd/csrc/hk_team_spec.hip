// hk::team_kernel<HK_SPEC_D, *> for one dim per object file (Makefile: -DHK_SPEC_D=2..6).
#define HK_SPEC_TU 1
#include "hk_team_kernel.h"

namespace hk {
template int launch_team_d<HK_SPEC_D>(const Params&, hipStream_t);
}  // namespace hk
