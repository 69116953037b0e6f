// C ABI of libhironaka_hip.so (include/hironaka_hip.h): argument validation, kernel selection
// and launch.  No allocation, no synchronisation, no exceptions; every entry point returns a
// status code.  gfx950 only.
#include "hk_fast_kernel.h"
#include "hk_duo_kernel.h"
#include "hk_quad_kernel.h"
#include "hk_quadroll_kernel.h"
#include "hk_quadgen_kernel.h"
#include "hk_quadbin_kernel.h"
#include "hk_team_kernel.h"
#include "hk_search.h"
#include "hk_generic_kernel.h"

using namespace hk;

namespace {

inline bool aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }
inline size_t elem_size(int dtype) { return dtype == HK_F64 ? 8 : 4; }

int check_spec(int batch, int m, int d, int dtype) {
  if (batch < 0 || m < 1 || d < 1 || d > kMaxDim) return HK_ERR_SHAPE;
  if ((int64_t)m * d > (1 << 20)) return HK_ERR_SHAPE;
  if (dtype != HK_F32 && dtype != HK_F64) return HK_ERR_UNSUPPORTED;
  return HK_OK;
}

// LDS geometry of the generic kernel: odd per-game stride, as many games per wave as fit
int plan_generic(Params& prm, int dtype) {
  const int n = prm.m * prm.d;
  int stride = n + prm.d;
  stride |= 1;
  const int64_t per_game = (int64_t)stride * (int64_t)elem_size(dtype);
  int gpb = kWave;
  while (gpb > 1 && per_game * gpb > kMaxLdsBytes) gpb >>= 1;
  if (per_game * gpb > kMaxLdsBytes) return HK_ERR_UNSUPPORTED;
  prm.lds_stride = stride;
  prm.games_per_block = gpb;
  return HK_OK;
}

template <typename T>
int launch_generic_t(const Params& prm, hipStream_t stream) {
  const size_t lds = (size_t)prm.lds_stride * prm.games_per_block * sizeof(T);
  if (lds > 64 * 1024) {
    // opting in to > 64 KiB of dynamic LDS is a per-function attribute (idempotent, cheap)
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&generic_kernel<T>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      (void)hipGetLastError();
      return HK_ERR_LAUNCH;
    }
  }
  const unsigned grid = (unsigned)(((int64_t)prm.batch + prm.games_per_block - 1) / prm.games_per_block);
  launch_prepare();
  hipLaunchKernelGGL(generic_kernel<T>, dim3(grid), dim3(kWave), lds, stream, prm);
  return launch_status();
}

int launch_generic(Params& prm, int dtype, hipStream_t stream) {
  if (prm.batch == 0) return HK_OK;
  const int st = plan_generic(prm, dtype);
  if (st != HK_OK) return st;
  return dtype == HK_F32 ? launch_generic_t<float>(prm, stream) : launch_generic_t<double>(prm, stream);
}

// kernel selection: register-resident specialisation -> team kernel (f32, dim 2..6, <= 64 rows; also the
// specialised shapes on HK_FLAG_FORCE_TEAM) -> generic kernel (anything else: f64, dim > 6, > 64 rows,
// HK_FLAG_FORCE_GENERIC, and the few mode / semantics combinations fast_supported / team_supported decline)
constexpr unsigned kHostSideFlags =
    HK_FLAG_FORCE_ONE_LANE | HK_FLAG_FORCE_TWO_LANES | HK_FLAG_FORCE_FOUR_LANES;  // kernel selection only

// hk_step on four lanes per game (hk_quad_kernel.h): where it is ahead of the two-lane kernel
static bool use_quad(const Params& prm, int dtype) {
  if (prm.flags & HK_FLAG_FORCE_FOUR_LANES) {
    Params probe = prm;
    probe.flags &= ~kHostSideFlags;
    return quad_supported(probe, dtype);
  }
  if (!quad_supported(prm, dtype)) return false;
  return quad_default(prm, device_simds());
}
// hk_generate_points on four lanes per game (hk_quadgen_kernel.h): everywhere it applies
static bool use_quadgen(const Params& prm, int dtype) {
  Params probe = prm;
  probe.flags &= ~(unsigned)HK_FLAG_FORCE_FOUR_LANES;
  return quadgen_supported(probe, dtype);
}
// plain rollouts on four lanes per game (hk_quadroll_kernel.h): forced, or where it is the default
static bool use_duo(const Params& prm) { return !(prm.flags & HK_FLAG_FORCE_ONE_LANE) && duo_wanted(prm); }
static bool use_quadroll(const Params& prm, int dtype) {
  Params probe = prm;
  probe.flags &= ~(unsigned)HK_FLAG_FORCE_FOUR_LANES;
  if (!quadroll_supported(probe, dtype)) return false;
  if (prm.flags & HK_FLAG_FORCE_FOUR_LANES) return true;
  // the small records without observations ride on the two-lane plain rollout too (hk_duo_kernel.h: flush_records):
  // where that is the faster plain kernel it takes them (28.8 -> ~23 us at (20,3) x 65 536)
  const bool small_only = !prm.obs_out && (prm.r_host_class_out || prm.r_axis_out || prm.r_done_out || prm.r_reward_out);
  const bool duo_takes = small_only && fast_supported(prm, dtype) && use_duo(prm);
  return quadroll_default(prm, device_simds(), duo_takes);
}
int launch(Params& prm, int dtype, hipStream_t stream) {
  if (prm.batch == 0) return HK_OK;
  if (use_quad(prm, dtype)) {
    prm.flags &= ~kHostSideFlags;
    return launch_quad(prm, stream);
  }
  if (use_quadroll(prm, dtype)) {
    prm.flags &= ~kHostSideFlags;  // (the compiled rollout configurations compare flags)
    return launch_quadroll(prm, stream);
  }
  if (use_quadgen(prm, dtype)) {
    prm.flags &= ~kHostSideFlags;
    return launch_quadgen(prm, stream);
  }
  // (the agent's move as an argmax of its logits is decoded by the four-lane kernel only)
  if ((prm.stages & HK_STAGE_SHIFT) && prm.axis_dtype == HK_AXIS_MASKED_LOGITS) return HK_ERR_UNSUPPORTED;
  if (fast_supported(prm, dtype)) {
    const bool duo = use_duo(prm);
    prm.flags &= ~kHostSideFlags;  // (the compiled rollout configurations compare flags)
    return duo ? launch_duo(prm, stream) : launch_fast(prm, stream);
  }
  prm.flags &= ~kHostSideFlags;
  if (team_supported(prm, dtype)) {
    const int st = launch_team(prm, stream);
    if (st != HK_ERR_UNSUPPORTED) return st;
  }
  return launch_generic(prm, dtype, stream);
}

// number of workgroups `launch` will use for this request (0: not launchable)
int64_t planned_grid(Params prm, int dtype) {
  if (prm.batch == 0) return 0;
  if (use_quadroll(prm, dtype)) return quadroll_grid(prm);
  if (fast_supported(prm, dtype)) {
    const int gpb = use_duo(prm) ? kDuoGames : fast_games_per_block(prm);
    return ((int64_t)prm.batch + gpb - 1) / gpb;
  }
  if (team_supported(prm, dtype) && plan_team(prm) == HK_OK)
    return ((int64_t)prm.batch + prm.games_per_block - 1) / prm.games_per_block;
  if (plan_generic(prm, dtype) != HK_OK) return 0;
  return ((int64_t)prm.batch + prm.games_per_block - 1) / prm.games_per_block;
}

// row length of the finished-game workspace for a geometry: an upper bound of the grids of all the kernel
// variants that may serve it (records, policies and semantics choose among them per launch)
int64_t count_slots(Params prm, int dtype) {
  if (prm.batch == 0) return 0;
  int64_t slots = planned_grid(prm, dtype);
  if (dtype == HK_F32 && prm.d >= 2 && prm.d <= 6 && prm.m <= kTeam * kTeamSlots)
    slots = std::max<int64_t>(slots, ((int64_t)prm.batch + kTeamGames - 1) / kTeamGames);
  if (has_fast_path(prm.m, prm.d, dtype))
    slots = std::max<int64_t>(slots, ((int64_t)prm.batch + kDuoGames - 1) / kDuoGames);
  Params gen = prm;
  if (plan_generic(gen, dtype) == HK_OK)
    slots = std::max<int64_t>(slots, ((int64_t)prm.batch + gen.games_per_block - 1) / gen.games_per_block);
  return slots;
}

int valid_coords_kind(int kind) {
  return (kind >= HK_F32 && kind <= HK_U8) || kind == HK_COORDS_CLASS_I32 ||
         kind == HK_COORDS_CLASS_I64 || kind == HK_COORDS_IN_RECORD;
}

size_t coords_align(int kind) {
  switch (kind) {
    case HK_F64: case HK_I64: case HK_COORDS_CLASS_I64: return 8;
    case HK_U8: return 1;
    default: return 4;
  }
}

// `internal_stages`: the feature-sort bits hk_get_features / hk_get_features_torch add on top of the public
// stage mask (callers of hk_step cannot set them: they are not in include/hironaka_hip.h)
int params_from_step(const hk_step_desc* s, Params& prm, unsigned internal_stages = 0) {
  if (!s) return HK_ERR_NULL;
  int st = check_spec(s->batch, s->max_points, s->dim, s->dtype);
  if (st != HK_OK) return st;
  if (s->batch == 0) { prm.batch = 0; return HK_OK; }
  if (!s->points_in || !s->points_out) return HK_ERR_NULL;
  const int64_t n = (int64_t)s->max_points * s->dim;
  if (s->in_stride < n || s->out_stride < n) return HK_ERR_SHAPE;
  const size_t es = elem_size(s->dtype);
  if (!aligned(s->points_in, es) || !aligned(s->points_out, es)) return HK_ERR_ALIGN;
  if (s->stages & ~(HK_STAGE_SHIFT | HK_STAGE_REPOSITION | HK_STAGE_NEWTON | HK_STAGE_RESCALE))
    return HK_ERR_UNSUPPORTED;
  if ((s->flags & HK_SEM_MASK) == HK_SEM_MASK) return HK_ERR_UNSUPPORTED;
  prm = Params{};
  if (s->stages & HK_STAGE_SHIFT) {
    if (!valid_coords_kind(s->coords_kind)) return HK_ERR_UNSUPPORTED;
    if (!s->axis) return HK_ERR_NULL;
    if (s->axis_dtype == HK_AXIS_MASKED_LOGITS) {
      if (s->coords_kind != HK_COORDS_CLASS_I32 || s->dtype != HK_F32) return HK_ERR_UNSUPPORTED;
    } else if (s->axis_dtype < HK_F32 || s->axis_dtype > HK_I64) {
      return HK_ERR_UNSUPPORTED;
    }
    if (!aligned(s->axis, (s->axis_dtype == HK_F64 || s->axis_dtype == HK_I64) ? 8 : 4)) return HK_ERR_ALIGN;
    if (s->coords_kind == HK_COORDS_IN_RECORD) {
      if (s->in_stride < n + s->dim) return HK_ERR_SHAPE;
    } else {
      if (!s->coords) return HK_ERR_NULL;
      if (!aligned(s->coords, coords_align(s->coords_kind))) return HK_ERR_ALIGN;
      if (s->coords_kind <= HK_U8 && s->coords_stride < s->dim) return HK_ERR_SHAPE;
    }
    prm.coords = s->coords;
    prm.coords_stride = s->coords_stride;
    prm.coords_kind = s->coords_kind;
    prm.axis = s->axis;
    prm.axis_dtype = s->axis_dtype;
  } else {
    prm.coords_kind = HK_COORDS_NONE;
  }
  if (s->reward_out && !aligned(s->reward_out, 4)) return HK_ERR_ALIGN;
  if (s->num_points_out && !aligned(s->num_points_out, 4)) return HK_ERR_ALIGN;
  prm.in = s->points_in;
  prm.out = s->points_out;
  prm.in_stride = s->in_stride;
  prm.out_stride = s->out_stride;
  prm.done_out = s->done_out;
  prm.prev_done_out = s->prev_done_out;
  prm.reward_out = (float*)s->reward_out;
  prm.num_points_out = s->num_points_out;
  prm.pad = s->padding_value;
  prm.reward_sign = s->reward_sign;
  prm.batch = s->batch;
  prm.m = s->max_points;
  prm.d = s->dim;
  prm.stages = s->stages | (internal_stages & kStageFeatureSorts);
  prm.flags = s->flags;
  prm.mode = kModeStep;
  return HK_OK;
}

hk_step_desc plain_desc(const void* in, void* out, int batch, int m, int d, int dtype, double pad,
                        uint32_t stages, uint32_t flags) {
  hk_step_desc s{};
  s.points_in = in;
  s.points_out = out;
  s.in_stride = s.out_stride = (int64_t)m * d;
  s.coords_kind = HK_COORDS_NONE;
  s.padding_value = pad;
  s.reward_sign = 1.0f;
  s.batch = batch;
  s.max_points = m;
  s.dim = d;
  s.dtype = dtype;
  s.stages = stages;
  s.flags = flags;
  return s;
}

}  // namespace

extern "C" {

int hk_abi_version(void) { return HK_ABI_VERSION; }

const char* hk_strerror(int status) {
  switch (status) {
    case HK_OK: return "ok";
    case HK_ERR_NULL: return "a required pointer is NULL";
    case HK_ERR_SHAPE: return "batch / max_points / dim / stride / class id out of range";
    case HK_ERR_UNSUPPORTED: return "dtype, kind or flag combination not supported";
    case HK_ERR_ALIGN: return "pointer not aligned to its element size";
    case HK_ERR_LAUNCH: return "HIP kernel launch failed";
    case HK_ERR_NO_DEVICE: return "no HIP device";
  }
  return "unknown status";
}

int hk_has_fast_path(int max_points, int dim, int dtype) { return has_fast_path(max_points, dim, dtype); }

int hk_step(const hk_step_desc* desc, void* stream) {
  Params prm{};
  const int st = params_from_step(desc, prm);
  if (st != HK_OK) return st;
  return launch(prm, desc->dtype, (hipStream_t)stream);
}

int hk_step_features(const hk_step_desc* desc, void* features_out, int scale_observation, void* stream) {
  Params prm{};
  const int st = params_from_step(desc, prm);
  if (st != HK_OK) return st;
  if (prm.batch == 0) return HK_OK;
  if (!features_out) return HK_ERR_NULL;
  if (!aligned(features_out, 16)) return HK_ERR_ALIGN;
  if (desc->dtype != HK_F32 || !(prm.stages & HK_STAGE_SHIFT)) return HK_ERR_UNSUPPORTED;
  prm.feat_out = (float*)features_out;
  prm.feat_scale = scale_observation ? 1 : 0;
  if (!use_quad(prm, desc->dtype)) return HK_ERR_UNSUPPORTED;
  prm.flags &= ~kHostSideFlags;
  return launch_quad(prm, (hipStream_t)stream);
}

int hk_shift(const void* points_in, void* points_out, const void* coords, int coords_kind,
             const void* axis, int axis_dtype, int batch, int max_points, int dim, int dtype,
             double padding_value, uint32_t flags, void* stream) {
  hk_step_desc s = plain_desc(points_in, points_out, batch, max_points, dim, dtype, padding_value,
                              HK_STAGE_SHIFT, flags);
  s.coords = coords;
  s.coords_kind = coords_kind;
  s.coords_stride = dim;
  s.axis = axis;
  s.axis_dtype = axis_dtype;
  return hk_step(&s, stream);
}

int hk_reposition(const void* points_in, void* points_out, int batch, int max_points, int dim,
                  int dtype, double padding_value, uint32_t flags, void* stream) {
  hk_step_desc s = plain_desc(points_in, points_out, batch, max_points, dim, dtype, padding_value,
                              HK_STAGE_REPOSITION, flags);
  return hk_step(&s, stream);
}

int hk_get_newton_polytope(const void* points_in, void* points_out, int batch, int max_points,
                           int dim, int dtype, double padding_value, uint32_t flags, void* stream) {
  hk_step_desc s = plain_desc(points_in, points_out, batch, max_points, dim, dtype, padding_value,
                              HK_STAGE_NEWTON, flags);
  return hk_step(&s, stream);
}

int hk_rescale(const void* points_in, void* points_out, int batch, int max_points, int dim,
               int dtype, double padding_value, uint32_t flags, void* stream) {
  hk_step_desc s = plain_desc(points_in, points_out, batch, max_points, dim, dtype, padding_value,
                              HK_STAGE_RESCALE, flags);
  return hk_step(&s, stream);
}

int hk_get_dones(const void* points, int64_t stride, uint8_t* done_out, int batch, int max_points,
                 int dim, int dtype, void* stream) {
  int st = check_spec(batch, max_points, dim, dtype);
  if (st != HK_OK) return st;
  if (batch == 0) return HK_OK;
  if (!points || !done_out) return HK_ERR_NULL;
  if (stride < (int64_t)max_points * dim) return HK_ERR_SHAPE;
  if (!aligned(points, elem_size(dtype))) return HK_ERR_ALIGN;
  return launch_counts(points, stride, done_out, nullptr, batch, max_points, dim, dtype, (hipStream_t)stream);
}

int hk_get_num_points(const void* points, int64_t stride, int32_t* num_points_out, int batch,
                      int max_points, int dim, int dtype, void* stream) {
  int st = check_spec(batch, max_points, dim, dtype);
  if (st != HK_OK) return st;
  if (batch == 0) return HK_OK;
  if (!points || !num_points_out) return HK_ERR_NULL;
  if (stride < (int64_t)max_points * dim) return HK_ERR_SHAPE;
  if (!aligned(points, elem_size(dtype)) || !aligned(num_points_out, 4)) return HK_ERR_ALIGN;
  return launch_counts(points, stride, nullptr, num_points_out, batch, max_points, dim, dtype, (hipStream_t)stream);
}

int hk_generate_points(void* points_out, int batch, int max_points, int dim, int dtype,
                       int max_value, uint64_t seed, uint64_t game_offset, uint32_t stages,
                       double padding_value, uint32_t flags, void* stream) {
  int st = check_spec(batch, max_points, dim, dtype);
  if (st != HK_OK) return st;
  if (batch == 0) return HK_OK;
  if (!points_out) return HK_ERR_NULL;
  if (max_value < 1) return HK_ERR_SHAPE;
  if (!aligned(points_out, elem_size(dtype))) return HK_ERR_ALIGN;
  if (stages & ~(HK_STAGE_REPOSITION | HK_STAGE_NEWTON | HK_STAGE_RESCALE)) return HK_ERR_UNSUPPORTED;
  if ((flags & HK_SEM_MASK) == HK_SEM_MASK) return HK_ERR_UNSUPPORTED;
  Params prm{};
  prm.out = points_out;
  prm.out_stride = (int64_t)max_points * dim;
  prm.in_stride = prm.out_stride;
  prm.coords_kind = HK_COORDS_NONE;
  prm.seed = seed;
  prm.game_offset = game_offset;
  prm.max_value = max_value;
  prm.pad = padding_value;
  prm.batch = batch;
  prm.m = max_points;
  prm.d = dim;
  prm.stages = stages;
  prm.flags = flags;
  prm.mode = kModeGenerate;
  return launch(prm, dtype, (hipStream_t)stream);
}

// ---- games binned by live rows (hk_quadbin_kernel.h) -------------------------------------------------------------------
int hk_bin_group_games(int max_points, int dim, int dtype) { return quadbin_group_games(max_points, dim, dtype); }
int hk_bin_unit_games(int max_points, int dim, int dtype) {
  return quadbin_group_games(max_points, dim, dtype) ? kQuadGames : 0;
}

static int bin_common(const void* in, void* points_out, int32_t* ids, int32_t* np_out, int batch, int m, int d, int dtype) {
  int st = check_spec(batch, m, d, dtype);
  if (st != HK_OK) return st;
  if (batch == 0) return HK_OK;
  if (!points_out || !ids) return HK_ERR_NULL;
  if (!quadbin_group_games(m, d, dtype)) return HK_ERR_UNSUPPORTED;
  if (!aligned(points_out, 16) || (in && !aligned(in, 16))) return HK_ERR_ALIGN;
  if (!aligned(ids, 4) || (np_out && !aligned(np_out, 4))) return HK_ERR_ALIGN;
  return HK_OK;
}

int hk_generate_points_binned(void* points_out, int32_t* game_ids_out, int32_t* num_points_out, int batch, int max_points,
                              int dim, int dtype, int max_value, uint64_t seed, uint64_t game_offset, uint32_t stages,
                              double padding_value, uint32_t flags, void* stream) {
  const int st = bin_common(nullptr, points_out, game_ids_out, num_points_out, batch, max_points, dim, dtype);
  if (st != HK_OK || batch == 0) return st;
  if (max_value < 1) return HK_ERR_SHAPE;
  if (stages & ~(HK_STAGE_REPOSITION | HK_STAGE_NEWTON | HK_STAGE_RESCALE)) return HK_ERR_UNSUPPORTED;
  Params prm{};
  prm.out = points_out;
  prm.out_stride = prm.in_stride = (int64_t)max_points * dim;
  prm.coords_kind = HK_COORDS_NONE;
  prm.seed = seed;
  prm.game_offset = game_offset;
  prm.max_value = max_value;
  prm.pad = padding_value;
  prm.batch = batch;
  prm.m = max_points;
  prm.d = dim;
  prm.stages = stages;
  prm.flags = flags & ~(unsigned)HK_FLAG_FORCE_FOUR_LANES;
  prm.mode = kModeGenerate;
  if (!quadgen_supported(prm, dtype)) return HK_ERR_UNSUPPORTED;  // (what the four-lane generator serves)
  return launch_quadbin(prm, nullptr, game_ids_out, num_points_out, (hipStream_t)stream);
}

int hk_bin_by_live_rows(const void* points_in, void* points_out, int32_t* game_ids_out, int32_t* num_points_out, int batch,
                        int max_points, int dim, int dtype, void* stream) {
  if (batch > 0 && !points_in) return HK_ERR_NULL;
  const int st = bin_common(points_in, points_out, game_ids_out, num_points_out, batch, max_points, dim, dtype);
  if (st != HK_OK || batch == 0) return st;
  Params prm{};
  prm.in = points_in;
  prm.out = points_out;
  prm.out_stride = prm.in_stride = (int64_t)max_points * dim;
  prm.batch = batch;
  prm.m = max_points;
  prm.d = dim;
  prm.pad = -1.0;
  return launch_quadbin(prm, (const float*)points_in, game_ids_out, num_points_out, (hipStream_t)stream);
}

static int params_from_rollout(const hk_rollout_desc* r, Params& prm) {
  if (!r) return HK_ERR_NULL;
  int st = check_spec(r->batch, r->max_points, r->dim, r->dtype);
  if (st != HK_OK) return st;
  if (r->steps < 0 || r->gen_max_value < 0 || r->episodes < 0) return HK_ERR_SHAPE;
  prm = Params{};
  prm.batch = r->batch;
  if (r->batch == 0) return HK_OK;
  const bool gen = r->gen_max_value > 0;  // the initial states are drawn inside the launch
  const int episodes = r->episodes > 1 ? r->episodes : 1;
  const bool records = r->obs_out || r->host_class_out || r->axis_out || r->done_out || r->reward_out;
  if (!gen && !r->points) return HK_ERR_NULL;
  if (gen && (r->points_in || (r->gen_stages & ~(HK_STAGE_REPOSITION | HK_STAGE_NEWTON | HK_STAGE_RESCALE))))
    return HK_ERR_UNSUPPORTED;
  // (every episode starts from the initial state again: it has to be somewhere -- points_in, or the generator)
  if (episodes > 1 && ((!gen && !r->points_in) || records)) return HK_ERR_UNSUPPORTED;
  if (r->dim < 2) return HK_ERR_SHAPE;  // the host needs a subset of >= 2 coordinates
  if (r->points && !aligned(r->points, elem_size(r->dtype))) return HK_ERR_ALIGN;
  if (r->done_count && !aligned(r->done_count, 8)) return HK_ERR_ALIGN;
  if (r->host_policy < HK_HOST_RANDOM || r->host_policy > HK_HOST_ZEILLINGER) return HK_ERR_UNSUPPORTED;
  if (r->agent_policy < HK_AGENT_RANDOM || r->agent_policy > HK_AGENT_CHOOSE_LAST) return HK_ERR_UNSUPPORTED;
  if (r->stages & ~(HK_STAGE_SHIFT | HK_STAGE_REPOSITION | HK_STAGE_NEWTON | HK_STAGE_RESCALE)) return HK_ERR_UNSUPPORTED;
  if ((r->flags & HK_SEM_MASK) == HK_SEM_MASK) return HK_ERR_UNSUPPORTED;
  if (r->points_in && !aligned(r->points_in, elem_size(r->dtype))) return HK_ERR_ALIGN;
  prm.in = gen ? nullptr : (r->points_in ? r->points_in : r->points);
  prm.out = r->points;
  prm.in_stride = prm.out_stride = (int64_t)r->max_points * r->dim;
  prm.coords_kind = HK_COORDS_NONE;
  prm.obs_out = r->obs_out;
  prm.r_host_class_out = r->host_class_out;
  prm.r_axis_out = r->axis_out;
  prm.r_done_out = r->done_out;
  prm.r_reward_out = r->reward_out;
  prm.game_length_out = r->game_length_out;
  prm.seed = r->seed;
  prm.game_offset = r->game_offset;
  if (r->game_ids && !aligned(r->game_ids, 4)) return HK_ERR_ALIGN;
  prm.game_ids = r->game_ids;
  prm.step_offset = r->step_offset;
  prm.steps = r->steps;
  prm.host_policy = r->host_policy;
  prm.agent_policy = r->agent_policy;
  prm.pad = r->padding_value;
  prm.reward_sign = r->reward_sign;
  prm.m = r->max_points;
  prm.d = r->dim;
  prm.stages = r->stages;
  prm.flags = r->flags & ~HK_FLAG_DEFER_COUNTS;  // host-side only: the kernels always add to the workspace
  prm.mode = kModeRollout;
  prm.max_value = gen ? r->gen_max_value : 0;
  prm.gen_seed = r->gen_seed;
  prm.gen_stages = r->gen_stages;
  prm.episodes = episodes;
  return HK_OK;
}

// rollouts with generated initial states and / or several episodes as ONE launch (hk_quadroll_kernel.h: GEN)
static bool use_quadroll_gen(const Params& prm, int dtype) {
  Params probe = prm;
  probe.flags &= ~(unsigned)HK_FLAG_FORCE_FOUR_LANES;
  if (prm.max_value <= 0) {
    // episodes from the states in memory: where the waves of the launch are resident all at once (four per SIMD), a wave
    // that starts its next episode early fills what the launch boundary left idle -- (20,3) x 65 536: 16.5 - 17.0 us per
    // episode against 20.8 one launch each; beyond (131 072 games: 37 against 30 us; (50,4) x 262 144: 205 against
    // 172 us, scripts/probe_persistent.py) the per-episode launches of the default kernels stay ahead
    if (prm.flags & HK_FLAG_FORCE_FOUR_LANES) return quadroll_episodes_supported(probe, dtype);
    const int64_t waves = ((int64_t)prm.batch + kQuadGames - 1) / kQuadGames;
    return prm.m <= 32 && waves <= (int64_t)4 * device_simds() && quadroll_episodes_supported(probe, dtype);
  }
  return quadroll_gen_supported(probe, dtype);
}

// the geometry of the launch(es) that will serve a rollout request (a request the fused kernel declines is served by
// hk_generate_points + a rollout per episode: the plain request's kernels)
static Params rollout_geometry(Params prm, int dtype) {
  if (!use_quadroll_gen(prm, dtype)) {
    prm.max_value = 0;
    prm.episodes = 1;
    if (!prm.in) prm.in = prm.out;
  }
  return prm;  // (the fused kernel's grid is quadroll_grid: within count_slots' bound whatever planned_grid picks)
}

uint64_t hk_rollout_workspace_bytes(const hk_rollout_desc* r) {
  Params prm{};
  if (params_from_rollout(r, prm) != HK_OK || prm.batch == 0) return 0;
  return (uint64_t)count_slots(rollout_geometry(prm, r->dtype), r->dtype) * (uint64_t)(r->steps + 1) * sizeof(uint32_t);
}

// workspace checks shared by hk_rollout and hk_rollout_reduce_counts
static int counts_workspace(const hk_rollout_desc* r, int64_t slots, uint32_t** ws) {
  if (!r->workspace) return HK_ERR_NULL;
  if (!aligned(r->workspace, 4)) return HK_ERR_ALIGN;
  if (r->workspace_bytes < (uint64_t)slots * (uint64_t)(r->steps + 1) * sizeof(uint32_t)) return HK_ERR_SHAPE;
  *ws = (uint32_t*)r->workspace;
  return HK_OK;
}

int hk_rollout(const hk_rollout_desc* r, void* stream) {
  Params prm{};
  const int st = params_from_rollout(r, prm);
  if (st != HK_OK) return st;
  if (prm.batch == 0) return HK_OK;
  const bool fused = use_quadroll_gen(prm, r->dtype);
  const bool gen = prm.max_value > 0;
  // what the fused kernel declines runs as hk_generate_points + a rollout per episode: the state then needs a buffer
  // (and a generated batch re-ordered by ids exists inside the fused kernel only)
  if (gen && !fused && (!r->points || r->game_ids)) return HK_ERR_UNSUPPORTED;
  const Params geo = rollout_geometry(prm, r->dtype);
  if (planned_grid(geo, r->dtype) == 0) return HK_ERR_UNSUPPORTED;
  const int64_t slots = count_slots(geo, r->dtype);
  const bool defer = (r->flags & HK_FLAG_DEFER_COUNTS) != 0;
  if (r->done_count || defer) {
    const int ws = counts_workspace(r, slots, &prm.count_ws);
    if (ws != HK_OK) return ws;
  }
  prm.count_stride = (uint32_t)slots;
  int ls = HK_OK;
  if (fused) {
    prm.flags &= ~kHostSideFlags;  // (the compiled rollout configurations compare flags)
    ls = launch_quadroll_gen(prm, (hipStream_t)stream);
  } else {
    const int episodes = prm.episodes;
    for (int e = 0; e < episodes && ls == HK_OK; ++e) {
      Params p = prm;
      p.max_value = 0;
      p.episodes = 1;
      p.seed = prm.seed + (uint64_t)e;
      if (gen) {
        ls = hk_generate_points(r->points, r->batch, r->max_points, r->dim, r->dtype, r->gen_max_value,
                                r->gen_seed + (uint64_t)e, r->game_offset, r->gen_stages, r->padding_value,
                                r->flags & (HK_SEM_MASK | HK_FLAG_FORCE_GENERIC | HK_FLAG_FORCE_TEAM | HK_FLAG_FORCE_ONE_LANE |
                                            HK_FLAG_FORCE_TWO_LANES | HK_FLAG_FORCE_FOUR_LANES),
                                stream);
        if (ls != HK_OK) break;
        p.in = r->points;
      }
      ls = launch(p, r->dtype, (hipStream_t)stream);
    }
  }
  if (ls != HK_OK || !r->done_count || defer) return ls;
  return launch_count_reduce(prm.count_ws, (int)slots, r->steps, (unsigned long long*)r->done_count,
                             (hipStream_t)stream);
}

int hk_rollout_reduce_counts(const hk_rollout_desc* desc, void* stream) {
  if (!desc) return HK_ERR_NULL;
  hk_rollout_desc copy = *desc;  // only the launch geometry matters here: `points` may be NULL
  if (!copy.points && copy.gen_max_value <= 0) copy.points = copy.workspace;
  const hk_rollout_desc* r = &copy;
  Params prm{};
  const int st = params_from_rollout(r, prm);
  if (st != HK_OK) return st;
  if (!r->done_count) return HK_ERR_NULL;
  if (prm.batch == 0) return HK_OK;
  const int64_t slots = count_slots(rollout_geometry(prm, r->dtype), r->dtype);
  if (slots == 0) return HK_ERR_UNSUPPORTED;
  uint32_t* ws = nullptr;
  const int wst = counts_workspace(r, slots, &ws);
  if (wst != HK_OK) return wst;
  return launch_count_reduce(ws, (int)slots, r->steps, (unsigned long long*)r->done_count, (hipStream_t)stream);
}

// ---- search tree operations --------------------------------------------------------------------------
static int search_tree_from(const hk_search_tree* t, SearchTree& s) {
  if (!t) return HK_ERR_NULL;
  if (t->batch < 0 || t->num_nodes < 1 || t->num_actions < 1 || t->num_actions > kSearchMaxActions)
    return HK_ERR_SHAPE;
  if (t->batch == 0) {
    s.batch = 0;
    return HK_OK;
  }
  if (!t->node_visits || !t->raw_values || !t->node_values || !t->parents || !t->action_from_parent ||
      !t->children_index || !t->children_prior_logits || !t->children_visits || !t->children_rewards ||
      !t->children_discounts || !t->children_values)
    return HK_ERR_NULL;
  s = SearchTree{t->node_visits, t->raw_values, t->node_values, t->parents, t->action_from_parent,
                 t->children_index, t->children_prior_logits, t->children_visits, t->children_rewards,
                 t->children_discounts, t->children_values, t->batch, t->num_nodes, t->num_actions};
  return HK_OK;
}

int hk_search_select(const hk_search_tree* tree, const float* root_gumbel, const uint8_t* root_invalid,
                     const int32_t* considered_visits, int max_num_considered_actions, int num_simulations,
                     int max_depth, int next_free_node, int32_t* parent_out, int32_t* action_out,
                     int32_t* node_out, void* stream) {
  SearchTree s{};
  const int st = search_tree_from(tree, s);
  if (st != HK_OK) return st;
  if (s.batch == 0) return HK_OK;
  if (!root_gumbel || !considered_visits || !parent_out || !action_out || !node_out) return HK_ERR_NULL;
  if (max_num_considered_actions < 1 || num_simulations < 1 || max_depth < 1 || next_free_node < 1 ||
      next_free_node >= s.num_nodes || num_simulations + 1 > s.num_nodes)
    return HK_ERR_SHAPE;
  return launch_search_select(s, root_gumbel, root_invalid, considered_visits, max_num_considered_actions,
                              num_simulations, max_depth, next_free_node, parent_out, action_out, node_out,
                              (hipStream_t)stream);
}

int hk_search_backup(const hk_search_tree* tree, const int32_t* parent, const int32_t* action,
                     const int32_t* node, const float* prior_logits, const float* value, const float* reward,
                     const float* discount, void* stream) {
  SearchTree s{};
  const int st = search_tree_from(tree, s);
  if (st != HK_OK) return st;
  if (s.batch == 0) return HK_OK;
  if (!parent || !action || !node || !prior_logits || !value || !reward || !discount) return HK_ERR_NULL;
  return launch_search_backup(s, parent, action, node, prior_logits, value, reward, discount, (hipStream_t)stream);
}

int hk_search_policy(const hk_search_tree* tree, const float* root_gumbel, const uint8_t* root_invalid,
                     int32_t* action_out, float* action_weights_out, void* stream) {
  SearchTree s{};
  const int st = search_tree_from(tree, s);
  if (st != HK_OK) return st;
  if (s.batch == 0) return HK_OK;
  if (!root_gumbel || !action_out || !action_weights_out) return HK_ERR_NULL;
  return launch_search_policy(s, root_gumbel, root_invalid, action_out, action_weights_out, (hipStream_t)stream);
}

static int expand_shape_ok(int batch, int num_nodes, int max_points, int dim) {
  if (batch < 0 || num_nodes < 1 || max_points < 1 || dim < 2 || dim > kMaxDim) return HK_ERR_SHAPE;
  if ((int64_t)max_points * dim > (1 << 20) || (int64_t)batch * num_nodes > (int64_t)1 << 40) return HK_ERR_SHAPE;
  return HK_OK;
}

int hk_search_expand_gather(const void* embeddings, const void* features, const int32_t* parent,
                            const int32_t* action, void* obs_out, void* agent_feat_out, int batch,
                            int num_nodes, int max_points, int dim, int node_major, void* stream) {
  const int st = expand_shape_ok(batch, num_nodes, max_points, dim);
  if (st != HK_OK) return st;
  const int64_t E64 = (int64_t)max_points * dim;
  const int64_t game_stride = node_major ? E64 : (int64_t)num_nodes * E64;
  const int64_t node_stride = node_major ? (int64_t)batch * E64 : E64;
  if (batch == 0) return HK_OK;
  if (!embeddings || !features || !parent || !action || !obs_out || !agent_feat_out) return HK_ERR_NULL;
  if (!aligned(embeddings, 4) || !aligned(features, 4) || !aligned(parent, 4) || !aligned(action, 4) ||
      !aligned(obs_out, 4) || !aligned(agent_feat_out, 4))
    return HK_ERR_ALIGN;
  return launch_expand_gather((const float*)embeddings, (const float*)features, parent, action, (float*)obs_out,
                              (float*)agent_feat_out, batch, num_nodes, max_points * dim, dim, game_stride, node_stride,
                              (hipStream_t)stream);
}

int hk_search_masked_argmax(const void* logits, const int32_t* action, int32_t* axis_out, int batch, int dim,
                            void* stream) {
  if (batch < 0 || dim < 2 || dim > kMaxDim) return HK_ERR_SHAPE;
  if (batch == 0) return HK_OK;
  if (!logits || !action || !axis_out) return HK_ERR_NULL;
  if (!aligned(logits, 4) || !aligned(action, 4) || !aligned(axis_out, 4)) return HK_ERR_ALIGN;
  return launch_masked_argmax((const float*)logits, action, axis_out, batch, dim, (hipStream_t)stream);
}

int hk_search_expand_scatter(const void* obs, const void* feat, const int32_t* node, void* embeddings,
                             void* features, int batch, int num_nodes, int max_points, int dim, void* stream) {
  const int st = expand_shape_ok(batch, num_nodes, max_points, dim);
  if (st != HK_OK) return st;
  if (batch == 0) return HK_OK;
  if (!obs || !feat || !node || !embeddings || !features) return HK_ERR_NULL;
  if (!aligned(obs, 4) || !aligned(feat, 4) || !aligned(node, 4) || !aligned(embeddings, 4) || !aligned(features, 4))
    return HK_ERR_ALIGN;
  return launch_expand_scatter((const float*)obs, (const float*)feat, node, (float*)embeddings, (float*)features,
                               batch, num_nodes, max_points * dim, (hipStream_t)stream);
}

int hk_search_expand_gather_agent(const void* embeddings, const int32_t* parent, void* points_out, void* coords_out,
                                  int batch, int num_nodes, int max_points, int dim, void* stream) {
  const int st = expand_shape_ok(batch, num_nodes, max_points, dim);
  if (st != HK_OK) return st;
  if (batch == 0) return HK_OK;
  if (!embeddings || !parent || !points_out || !coords_out) return HK_ERR_NULL;
  if (!aligned(embeddings, 4) || !aligned(parent, 4) || !aligned(points_out, 4) || !aligned(coords_out, 4))
    return HK_ERR_ALIGN;
  return launch_expand_gather_agent((const float*)embeddings, parent, (float*)points_out, (float*)coords_out, batch,
                                    num_nodes, max_points * dim, dim, (hipStream_t)stream);
}

int hk_search_expand_scatter_agent(const void* points, const void* feat, const void* host_logits,
                                   const int32_t* node, void* embeddings, void* features, void* agent_feat_out,
                                   int32_t* class_out, int batch, int num_nodes, int max_points, int dim,
                                   int num_classes, void* stream) {
  const int st = expand_shape_ok(batch, num_nodes, max_points, dim);
  if (st != HK_OK) return st;
  if (num_classes < 1 || (int64_t)num_classes > ((int64_t)1 << dim) - dim - 1) return HK_ERR_SHAPE;
  if (batch == 0) return HK_OK;
  if (!points || !feat || !host_logits || !node || !embeddings || !agent_feat_out) return HK_ERR_NULL;
  if (!aligned(points, 4) || !aligned(feat, 4) || !aligned(host_logits, 4) || !aligned(node, 4) ||
      !aligned(embeddings, 4) || !aligned(features, 4) || !aligned(agent_feat_out, 4) || !aligned(class_out, 4))
    return HK_ERR_ALIGN;
  return launch_expand_scatter_agent((const float*)points, (const float*)feat, (const float*)host_logits, node,
                                     (float*)embeddings, (float*)features, (float*)agent_feat_out, class_out, batch,
                                     num_nodes, max_points * dim, dim, num_classes, (hipStream_t)stream);
}

int hk_search_mask_logits(const void* logits, const int32_t* class_id, void* out, int batch, int dim, void* stream) {
  if (batch < 0 || dim < 2 || dim > kMaxDim) return HK_ERR_SHAPE;
  if (batch == 0) return HK_OK;
  if (!logits || !class_id || !out) return HK_ERR_NULL;
  if (!aligned(logits, 4) || !aligned(class_id, 4) || !aligned(out, 4)) return HK_ERR_ALIGN;
  return launch_mask_logits((const float*)logits, class_id, (float*)out, batch, dim, (hipStream_t)stream);
}

int hk_rollout_values(const void* obs, void* value_out, int batch, int steps, int obs_dim, int dim,
                      int points_offset, float discount, float reward_sign, float estimate_scale, void* stream) {
  if (batch < 0 || steps < 1 || obs_dim < 1 || dim < 1 || dim > kMaxDim || points_offset < 0 || points_offset > 1)
    return HK_ERR_SHAPE;
  if (steps > 64) return HK_ERR_UNSUPPORTED;
  if (batch == 0) return HK_OK;
  if (!obs || !value_out) return HK_ERR_NULL;
  if (!aligned(obs, 4) || !aligned(value_out, 4)) return HK_ERR_ALIGN;
  return launch_rollout_values((const float*)obs, (float*)value_out, batch, steps, obs_dim, dim, points_offset, discount,
                               reward_sign, estimate_scale, (hipStream_t)stream);
}

int hk_zeillinger(const void* points, int64_t stride, int32_t* class_out, int batch,
                  int max_points, int dim, int dtype, uint32_t flags, void* stream) {
  int st = check_spec(batch, max_points, dim, dtype);
  if (st != HK_OK) return st;
  if (batch == 0) return HK_OK;
  if (!points || !class_out) return HK_ERR_NULL;
  if (dim < 2) return HK_ERR_SHAPE;
  if (stride < (int64_t)max_points * dim) return HK_ERR_SHAPE;
  if (!aligned(points, elem_size(dtype)) || !aligned(class_out, 4)) return HK_ERR_ALIGN;
  if ((flags & HK_SEM_MASK) != HK_SEM_JAX && (flags & HK_SEM_MASK) != HK_SEM_LIST) return HK_ERR_UNSUPPORTED;
  Params prm{};
  prm.flags = flags;
  prm.in = points;
  prm.in_stride = stride;
  prm.out_stride = stride;
  prm.class_out = class_out;
  prm.coords_kind = HK_COORDS_NONE;
  prm.batch = batch;
  prm.m = max_points;
  prm.d = dim;
  prm.mode = kModeZeillinger;
  // on the register-resident / team kernels: a step launch without stages and without a state output, whose
  // only product is class_out (both variants: the semantics code travels in the flags)
  if (!(flags & HK_FLAG_FORCE_GENERIC)) {
    Params fast = prm;
    fast.mode = kModeStep;
    fast.pad = -1.0;
    // four lanes per game (hk_quadroll_kernel.h: the rollout kernel's prologue + its balanced pair loop) wherever that
    // kernel exists: the JAX variant over contiguous 16-B aligned records ((20,3) x 65 536: 21.4 us on one lane per
    // game, (20,4): 37.4)
    if (dtype == HK_F32 && (flags & HK_SEM_MASK) == HK_SEM_JAX &&
        !(flags & (HK_FLAG_FORCE_TEAM | HK_FLAG_FORCE_ONE_LANE | HK_FLAG_FORCE_TWO_LANES)) &&
        stride == (int64_t)max_points * dim && aligned(points, 16)) {
      const int qs = launch_quadzeil(prm, (hipStream_t)stream);
      if (qs != HK_ERR_UNSUPPORTED) return qs;
    }
    if (fast_supported(fast, dtype)) return launch_fast(fast, (hipStream_t)stream);
    if (team_supported(fast, dtype)) {
      const int ts = launch_team(fast, (hipStream_t)stream);
      if (ts != HK_ERR_UNSUPPORTED) return ts;
    }
  }
  return launch_generic(prm, dtype, (hipStream_t)stream);
}

int hk_get_features(const void* points_in, int64_t in_stride, void* features_out,
                    int64_t out_stride, int batch, int max_points, int dim, int dtype,
                    int scale_observation, double padding_value, void* stream) {
  hk_step_desc s = plain_desc(points_in, features_out, batch, max_points, dim, dtype, padding_value,
                              (scale_observation ? HK_STAGE_RESCALE : 0u), HK_SEM_JAX);
  s.in_stride = in_stride;
  s.out_stride = out_stride;
  Params prm{};
  const int st = params_from_step(&s, prm, kStageFeatureSort);
  if (st != HK_OK) return st;
  return launch(prm, dtype, (hipStream_t)stream);
}

int hk_get_features_torch(const void* points_in, int64_t in_stride, void* features_out, int64_t out_stride,
                          int batch, int max_points, int dim, int dtype, double padding_value, void* stream) {
  hk_step_desc s = plain_desc(points_in, features_out, batch, max_points, dim, dtype, padding_value,
                              0u, HK_SEM_TORCH);
  s.in_stride = in_stride;
  s.out_stride = out_stride;
  Params prm{};
  const int st = params_from_step(&s, prm, kStageFeatureSort0);
  if (st != HK_OK) return st;
  return launch(prm, dtype, (hipStream_t)stream);
}

int hk_decode_host_class(const int32_t* class_in, void* mask_out, int mask_dtype, int batch,
                         int dim, void* stream) {
  if (batch < 0 || dim < 2 || dim > kMaxDim) return HK_ERR_SHAPE;
  if (batch == 0) return HK_OK;
  if (!class_in || !mask_out) return HK_ERR_NULL;
  if (mask_dtype < HK_F32 || mask_dtype > HK_U8) return HK_ERR_UNSUPPORTED;
  if (!aligned(class_in, 4) || !aligned(mask_out, coords_align(mask_dtype))) return HK_ERR_ALIGN;
  return launch_decode(class_in, mask_out, mask_dtype, batch, dim, (hipStream_t)stream);
}

}  // extern "C"
