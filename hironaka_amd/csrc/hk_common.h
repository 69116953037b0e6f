// Shared device/host definitions for the HIP kernels of the batched Hironaka step.
// gfx950 (MI355X) only: wave = 64 lanes, 160 KiB LDS per CU.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hironaka_hip.h"

namespace hk {

constexpr int kWave = 64;
constexpr int kMaxLdsBytes = 160 * 1024;  // one workgroup may own the CU's whole LDS
constexpr int kMaxDim = 30;               // subsets travel as 32-bit masks

// internal stage bits (beyond the public HK_STAGE_*)
constexpr unsigned kStageFeatureSort = 1u << 8;   // jax/util.py:186-197 row ordering (last coordinate primary)
constexpr unsigned kStageFeatureSort0 = 1u << 9;  // core/tensor_points.py:72-74 (coordinate 0 alone, stable)
constexpr unsigned kStageFeatureSorts = kStageFeatureSort | kStageFeatureSort0;

// row orderings (descending) of the rank helpers
enum KeyOrder : int {
  kKeyLast = 0,   // lexicographic, last coordinate primary: the observation features' lexsort
  kKeyFirst = 1,  // lexicographic, coordinate 0 primary: the list semantics (_list_ops.py:25-41)
  kKeyCoord0 = 2  // coordinate 0 alone, ties in row order: TensorPoints.get_features
};

enum Mode : int { kModeStep = 0, kModeRollout = 1, kModeGenerate = 2, kModeZeillinger = 3 };
// template-only variant of kModeRollout: the rollout that also writes per-step observations / records
// (kept out of the plain rollout kernel, whose loop it would cost ~40 VGPRs)
constexpr int kModeRolloutRec = 4;
// template-only variant of kModeStep: the step launches whose product is not the next state -- the sorted
// observation features (hk_get_features) and Zeillinger's class (hk_zeillinger) -- kept out of the plain step
// kernel, which then carries neither their code nor their registers
constexpr int kModeStepAux = 5;

// RNG stream ids (DESIGN.md "Randomness"); the oracle uses the same two numbers.
constexpr uint32_t kStreamPolicy = 0u;
constexpr uint32_t kStreamGenerate = 1u;

// Kernel argument block (passed by value).  Union of what the modes need.
struct Params {
  const void* in;
  void* out;
  int64_t in_stride;
  int64_t out_stride;
  const void* coords;
  int64_t coords_stride;
  const void* axis;
  uint8_t* done_out;
  uint8_t* prev_done_out;
  float* reward_out;
  int32_t* num_points_out;
  int32_t* class_out;  // zeillinger operator
  // rollout
  uint32_t* count_ws;  // [steps+1][count_stride] per-workgroup finished-game counts (or NULL)
  void* obs_out;
  int32_t* r_host_class_out;
  int32_t* r_axis_out;
  uint8_t* r_done_out;
  float* r_reward_out;
  int32_t* game_length_out;
  uint64_t seed;
  uint64_t game_offset;
  uint32_t step_offset;
  int32_t steps;
  int32_t host_policy;
  int32_t agent_policy;
  int32_t max_value;  // generate
  double pad;
  float reward_sign;
  int32_t batch;
  int32_t m;
  int32_t d;
  int32_t coords_kind;
  int32_t axis_dtype;
  uint32_t stages;
  uint32_t flags;
  int32_t lds_stride;       // elements per game in LDS (generic kernel)
  int32_t games_per_block;  // <= 64
  int32_t mode;
  // row length of count_ws: the same for every kernel variant that may serve a geometry (>= any of their
  // grids), so that deferred counts of different launches meet in one workspace
  uint32_t count_stride;
  float pad_f32;  // (float)pad, filled in by launchers whose kernels would otherwise convert per wave
  // hk_step_features (four-lane kernel): the observation features of the step's RESULT as a second output
  float* feat_out;     // [batch, m * d] or NULL
  int32_t feat_scale;  // rescale before the sort (scale_observation)
  // rollouts: the policy stream's game index of position g is game_offset + game_ids[g] (NULL: game_offset + g)
  const int32_t* game_ids;
  // rollouts from initial states drawn inside the launch (hk_rollout_desc.gen_max_value > 0: `max_value` holds it, `in`
  // is NULL and `out` may be) and episodes back to back
  uint64_t gen_seed;
  uint32_t gen_stages;
  int32_t episodes;  // >= 1
};

// hipGetLastError() is sticky per host thread and other users of the runtime in this process
// (torch) leave benign errors behind: clear it right before a launch, read it right after.
inline void launch_prepare() { (void)hipGetLastError(); }
inline int launch_status() { return hipGetLastError() == hipSuccess ? HK_OK : HK_ERR_LAUNCH; }

// ---- Philox4x32-10 ------------------------------------------------------------------------
struct U4 {
  uint32_t x, y, z, w;
};

__host__ __device__ inline U4 philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                         uint64_t seed) {
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0;
    c1 = n1;
    c2 = n2;
    c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return U4{c0, c1, c2, c3};
}

// The two draws (host, agent) of policy step `step` of game `gg` as 32-bit words for mulhi32 (DESIGN.md "Randomness").
// Round 3: while the host's classes fit 16-bit draws comfortably (dim <= kPolicyShortDim: at most 247 classes) a Philox
// block (gg_lo, gg_hi, step >> 2, kStreamPolicy) serves FOUR consecutive steps -- word step & 3, its high half the
// host's draw, its low half the agent's, each handed on as the high half of a 32-bit word (mulhi32(h << 16, n) =
// (h * n) >> 16) -- : half the Philox work of the two-steps-per-block form (a block is ~115 instructions, 20 of them
// quarter-rate multiplies; it was 14 % of a SIMD's time in the fused rollouts).  Beyond that dimension: block
// (.., step >> 1, ..), words (x, y) for the even step, (z, w) for the odd one, as before.
constexpr int kPolicyShortDim = 8;

struct PolicyCache {
  U4 r;
  uint32_t block = 0xFFFFFFFFu;  // wave-uniform: which block `r` holds
};

__host__ __device__ inline uint32_t u4_word(const U4& r, uint32_t i) {
  return i == 0 ? r.x : (i == 1 ? r.y : (i == 2 ? r.z : r.w));
}

__host__ __device__ inline void policy_words(uint64_t gg, uint32_t step, uint64_t seed, PolicyCache& cache, int d,
                                             uint32_t& host_word, uint32_t& agent_word) {
  const bool short_form = d <= kPolicyShortDim;
  const uint32_t block = short_form ? step >> 2 : step >> 1;
  if (cache.block != block) {
    cache.r = philox4x32((uint32_t)gg, (uint32_t)(gg >> 32), block, kStreamPolicy, seed);
    cache.block = block;
  }
  if (short_form) {
    const uint32_t w = u4_word(cache.r, step & 3u);
    host_word = w & 0xFFFF0000u;
    agent_word = w << 16;
  } else {
    host_word = (step & 1u) ? cache.r.z : cache.r.x;
    agent_word = (step & 1u) ? cache.r.w : cache.r.y;
  }
}

// floor(r * n / 2^32): 32-bit word -> [0, n)
__host__ __device__ inline uint32_t mulhi32(uint32_t r, uint32_t n) {
  return (uint32_t)(((uint64_t)r * n) >> 32);
}

// ---- the generator stream (DESIGN.md "Randomness"): element e = row * dim + coordinate of game gg ----------------------
//   max_value <= kGenShortMax (round 4, ABI 4): Philox block (gg_lo, gg_hi, e >> 3, kStreamGenerate), word (e >> 1) & 3,
//     its LOW half for even e, its HIGH half for odd e, value = (half * max_value) >> 16 -- eight elements per block: the
//     draws of a fresh game are half of what generating it costs (a block is ~70 instructions, 20 of them slow 32 x 32 ->
//     64 multiplies).  A 16-bit draw's bias is max_value / 2^16 < 0.1 % at the cap (0.03 % at the trainers' max_value 20);
//   beyond: block e >> 2, word e & 3, value = mulhi32(word, max_value) -- four elements per block, as in ABI <= 3.
constexpr int kGenShortMax = 64;
__host__ __device__ inline bool gen_short(uint32_t max_value) { return max_value <= (uint32_t)kGenShortMax; }
// the values of the elements a block serves, in element order: v[0..8) short, v[0..4) otherwise
__host__ __device__ inline void gen_block_values(const U4& r, uint32_t max_value, bool short_form, uint32_t (&v)[8]) {
  const uint32_t w[4] = {r.x, r.y, r.z, r.w};
  if (short_form) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      v[2 * k] = ((w[k] & 0xFFFFu) * max_value) >> 16;
      v[2 * k + 1] = ((w[k] >> 16) * max_value) >> 16;
    }
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      v[k] = mulhi32(w[k], max_value);
      v[4 + k] = 0u;
    }
  }
}

// ---- host action codec (jax/host_action_preprocess.py:8-24,78-87) -------------------------
// class id -> bitmask (bit j <-> coordinate j): the cls-th integer >= 3 that is not a power of
// two.  Integers with top bit L hold classes [2^L - L - 1, 2^(L+1) - L - 3].
__host__ __device__ inline uint32_t decode_class(int cls, int d) {
  for (int L = 1; L < d; ++L) {
    const int hi = (int)(2u << L) - L - 3;
    if (cls <= hi) return (uint32_t)(cls + L + 2);
  }
  return 3u;
}

__host__ __device__ inline int encode_mask(uint32_t v) {
  int lg = 0;
  while ((v >> (lg + 1)) != 0) ++lg;
  return (int)v - lg - 2;
}

__device__ inline double load_scalar(const void* base, int dtype, size_t idx) {
  switch (dtype) {
    case HK_F32: return (double)((const float*)base)[idx];
    case HK_F64: return ((const double*)base)[idx];
    case HK_I32: return (double)((const int32_t*)base)[idx];
    case HK_I64: return (double)((const long long*)base)[idx];
    case HK_U8: return (double)((const uint8_t*)base)[idx];
  }
  return 0.0;
}

// A workgroup's finished-game count of one step goes to its own slot of the caller's workspace, ADDED
// (an atomic nobody waits for; the slot is private, so there is no contention): the slots start at zero,
// accumulate over as many launches as the caller defers the reduction for (HK_FLAG_DEFER_COUNTS), and are
// zeroed again by the reduce kernel that sums them into done_count.
__device__ __forceinline__ void count_add(uint32_t* slot, uint32_t v) {
  __hip_atomic_fetch_add(slot, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The per-step finished-game counts of a wave from its games' FIRST finished steps (a finished game stays finished: rows
// are only ever removed): step s counts the games with 0 <= length <= s.  One ballot per step, the popcount parked in
// lane s - base, and ONE atomic instruction per 64 steps (lane s adds to slot[s * stride]) -- as a loop of ballot + branch
// + address + atomic per step this was ~1 us at the end of every wave's life.  `counted`: the lane speaks for a game
// (one lane per game); steps first .. nsteps.
__device__ __forceinline__ void add_length_counts(uint32_t* slot, uint32_t stride, int first, int nsteps, bool counted,
                                                  int length, int lane) {
  const unsigned ulen = counted ? (unsigned)length : 0xFFFFFFFFu;  // (-1 = never finished: above every step)
  for (int base = first; base <= nsteps; base += kWave) {
    uint32_t mine = 0;
    const int top = nsteps < base + kWave - 1 ? nsteps : base + kWave - 1;
#pragma nounroll
    for (int s = base; s <= top; ++s) {
      const unsigned long long b = __ballot(ulen <= (unsigned)s);
      mine = (lane == s - base) ? (uint32_t)__popcll(b) : mine;
    }
    const int s = base + lane;
    if (s <= nsteps && mine) count_add(slot + (size_t)s * stride, mine);
  }
}

// The same fetch split in two, branch-free in the part that touches memory: `fetch_raw` requests the
// aligned dword(s) holding element idx (the upper request repeats the lower one for elements narrower
// than 8 bytes, so every address is inside the caller's array's own words), and nothing looks at the
// bits until `scalar_from_raw`.  Fetches of any mix of dtypes therefore stay in flight together with
// whatever is issued next -- a dtype switch around the loads would end every case in a wait.
struct RawScalar {
  uint32_t lo, hi, shift;
};

__device__ inline int dtype_size(int dtype) {
  return (dtype == HK_F64 || dtype == HK_I64) ? 8 : (dtype == HK_U8 ? 1 : 4);
}

__device__ inline RawScalar fetch_raw(const void* base, int dtype, size_t idx) {
  const int esz = dtype_size(dtype);
  // pointer arithmetic only (an integer round trip would turn these into flat loads)
  const char* p = static_cast<const char*>(base) + idx * (size_t)esz;
  const uint32_t off = (uint32_t)(reinterpret_cast<uintptr_t>(p) & 3u);
  const uint32_t* lo = reinterpret_cast<const uint32_t*>(p - off);
  const uint32_t* hi = lo + (esz == 8 ? 1 : 0);
  RawScalar r;
  r.lo = *lo;
  r.hi = *hi;
  r.shift = off * 8u;
  return r;
}

__device__ inline double scalar_from_raw(const RawScalar& r, int dtype) {
  switch (dtype) {
    case HK_F32: return (double)__uint_as_float(r.lo);
    case HK_F64: return __longlong_as_double((long long)(((uint64_t)r.hi << 32) | r.lo));
    case HK_I32: return (double)(int32_t)r.lo;
    case HK_I64: return (double)(long long)(((uint64_t)r.hi << 32) | r.lo);
    case HK_U8: return (double)((r.lo >> r.shift) & 0xFFu);
  }
  return 0.0;
}

// `arange(d) == axis` (_jax_ops.py:79): non-integral / out-of-range values match nothing.
__device__ inline int axis_index(double a, int d) {
  if (!(a >= 0.0) || a >= (double)d || a != floor(a)) return -1;
  return (int)a;
}

// min / max of two floats that are known not to be NaN where the result matters (finite values and the
// +inf holes of the register-resident kernels).  fminf/fmaxf make the compiler canonicalise every operand
// it cannot prove quiet (an extra v_max x, x per operand); v_med3 against -inf / +inf is the same
// function on non-NaN inputs in ONE instruction.
__device__ __forceinline__ float hk_fmin(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, -INFINITY); }
__device__ __forceinline__ float hk_fmax(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, INFINITY); }

// the f32-typed padding of the torch sibling (see oracle/hko_impl.inc torch_pad)
template <typename T>
__device__ inline T torch_pad(T pad) {
  return (T)(float)pad;
}

}  // namespace hk
