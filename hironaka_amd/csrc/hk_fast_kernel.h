// Register-resident specialisations of the Hironaka step for compile-time (max_points, dim),
// float32, in-place-order semantics (JAX / torch).  One lane per game.
//
// Data movement.  A workgroup (one wave) owns 64 consecutive games = one contiguous slab of HBM.
// The slab is copied with fully coalesced wave requests (lane l moves bytes [16l, 16l+16) of each
// KiB; W = 4/2/1 dwords by the divisibility of M*D) into an LDS *image* whose per-game stride is
// conflict-free for W-wide ds_read/ds_write (stride/W odd).  Results leave the same way.  HBM
// sees exactly one read and one write of the state per launch, whatever the number of steps.
//
// Sparsity.  After a Newton-polytope pass a game of max_points = 20 typically keeps ~5 points;
// the other rows are padding.  Each lane therefore scans its image once (live-row bitmask +
// a check that every row is either fully available and finite or exactly the padding row),
// GATHERS its live rows into registers q[0..n) with compile-time register indices, and a
// transition -- shift, reposition, the O(n^2 D) domination test, rescale -- runs as branch-free
// code for NB rows, the smallest of 1..8, 10, 12, ... that covers the wave-uniform nmax = max over
// the 64 games of n (hk_fast_rows.h: one instruction of any kind per four cycles is all a lone wave
// per SIMD gets, so loop control must not cost as much as the work).
// Rows beyond a lane's own n are +inf "holes": they cannot dominate, and whatever is computed for
// them is never published.  The image is rebuilt (pad fill + scatter of live rows to their
// original slots, so rows keep their positions as the reference's in-place semantics requires)
// only when the state has to leave the registers: at the end of the launch or for a per-step
// observation.  In a fused rollout the rows are re-gathered whenever nmax would shrink.
//
// Domination test on the gathered rows, per unordered pair i<j:
//     t = max_k(q_i - q_j),  u = min_k(q_i - q_j)          (one set of differences)
//     row j is removed by i   iff t <= 0                    (P_i <= P_j; ties go to the lower index)
//     row i is removed by j   iff u >= 0 and t > 0          (P_j <= P_i and not equal)
// accumulated as ONE running minimum per row in a VGPR (no lane-mask SGPR pressure).  The sign of a float difference is exact, so this equals the reference's
// `diff >= 0` test (_jax_ops.py:55-56); keeping the first of equal rows equals its
// remove_repeated (_jax_ops.py:24-40).
//
// Exactness guard.  The shortcut is bit-exact iff every row is all->=0-and-finite or uniformly equal
// to the duplicate-fill value AND that fill value equals the padding value (always under torch
// semantics; under JAX semantics iff padding_value == -1, _jax_ops.py:65).  Anything else (mixed
// sign rows, irregular padding, non-finite values, exotic padding values) is detected per wave and
// that wave runs the exact generic routines on its LDS image instead.
#pragma once

#include <type_traits>

#include "hk_fast_rows.h"
#include "hk_generic_kernel.h"

namespace hk {

template <int M, int D>
struct FastGeom {
  static constexpr int N = M * D;
  static constexpr int W = (N % 4 == 0) ? 4 : ((N % 2 == 0) ? 2 : 1);
  static constexpr int Q = N / W;                     // W-chunks per game
  static constexpr int S = (Q % 2 == 1) ? N : N + W;  // LDS stride in floats, S/W odd
  // register capacity: live rows a lane can hold.  Beyond 32 rows the state would not fit the VGPR
  // budget at a useful occupancy; a wave whose widest game has more live rows than C runs the exact
  // generic routines on its LDS image instead (Newton-reduced states of 50 slots hold ~12 rows).
  static constexpr int C = (M <= 32) ? M : 32;
};

// native vector types (usable as inline-asm register operands)
typedef float vf4 __attribute__((ext_vector_type(4)));
typedef float vf2 __attribute__((ext_vector_type(2)));
template <int W> struct VecOf;
template <> struct VecOf<4> { using type = vf4; };
template <> struct VecOf<2> { using type = vf2; };
template <> struct VecOf<1> { using type = float; };

// wave-wide maximum of a small int in [0, hi], returned in an SGPR: a downward search with one
// ballot per candidate (scalar compares only) instead of a 6-deep chain of cross-lane shuffles whose
// LDS-crossbar latency a lone wave per SIMD cannot hide.  From a previous maximum it ends after a
// step or two.
__device__ inline int wave_max(int v, int hi) {
  int m = hi;
#pragma nounroll
  while (m > 0 && !__any(v >= m)) --m;
  return m;
}

// ---- slab I/O ---------------------------------------------------------------------------------------
// A lone wave per SIMD hides no latency by itself, so the whole slab is requested before the first
// dependent instruction: Q wide loads per lane in flight together (one HBM round trip per launch, not
// one per request).  Chunks past a partial slab re-read its last chunk (always a valid address) and
// are simply not written to the image.
// element offsets of chunk q (W floats) of a slab: in HBM (from the slab's first record) and in the LDS
// image.  Contiguous records (stride == M*D, the usual case) and an unpadded image need no division.
template <int M, int D, bool CONTIG>
__device__ __forceinline__ int64_t slab_chunk_global(int q, int64_t stride) {
  using G = FastGeom<M, D>;
  if (CONTIG) return (int64_t)q * G::W;
  const int g = q / G::Q, c = q - g * G::Q;
  return g * stride + c * G::W;
}
template <int M, int D>
__device__ __forceinline__ int slab_chunk_lds(int q) {
  using G = FastGeom<M, D>;
  if (G::S == G::N) return q * G::W;
  const int g = q / G::Q, c = q - g * G::Q;
  return g * G::S + c * G::W;
}

// the slab's requests (issue) and their way into the image (commit) are separate calls so that a kernel can
// put other loads between them
template <int M, int D>
struct SlabRegs {
  typename VecOf<FastGeom<M, D>::W>::type v[FastGeom<M, D>::Q];
};

template <int M, int D, bool CONTIG>
__device__ __forceinline__ void fast_slab_issue_impl(SlabRegs<M, D>& r, const float* base, int64_t in_stride,
                                                     int ngames, int lane) {
  using G = FastGeom<M, D>;
  using V = typename VecOf<G::W>::type;
  const int total = ngames * G::Q;
#pragma unroll
  for (int it = 0; it < G::Q; ++it) {
    int q = lane + it * kWave;
    q = q < total ? q : total - 1;
    r.v[it] = *reinterpret_cast<const V*>(base + slab_chunk_global<M, D, CONTIG>(q, in_stride));
  }
}

template <int M, int D>
__device__ __forceinline__ void fast_slab_issue(SlabRegs<M, D>& r, const float* in, int64_t in_stride, int64_t g0,
                                                int ngames, int lane) {
  const float* base = in + g0 * in_stride;
  if (in_stride == FastGeom<M, D>::N) fast_slab_issue_impl<M, D, true>(r, base, in_stride, ngames, lane);
  else fast_slab_issue_impl<M, D, false>(r, base, in_stride, ngames, lane);
}

template <int M, int D>
__device__ __forceinline__ void fast_slab_commit(SlabRegs<M, D>& r, float* lds, int ngames, int lane) {
  using G = FastGeom<M, D>;
  using V = typename VecOf<G::W>::type;
  const int total = ngames * G::Q;
  // an opaque use of every chunk right here: otherwise the compiler sinks each load into the
  // conditional block of its store and waits for it there, one round trip per chunk
#pragma unroll
  for (int it = 0; it < G::Q; ++it) asm volatile("" : "+v"(r.v[it]));
#pragma unroll
  for (int it = 0; it < G::Q; ++it) {
    const int q = lane + it * kWave;
    if (q < total) *reinterpret_cast<V*>(lds + slab_chunk_lds<M, D>(q)) = r.v[it];
  }
}

// Stores need no such care (nothing waits for them); the image is read a few chunks at a time so that a
// store inside the rollout loop (per-step observations) adds little to the loop's register pressure.
template <int M, int D, bool CONTIG, int kBatch, bool NT = false>
__device__ __forceinline__ void fast_store_slab_impl(const float* lds, float* base, int64_t out_stride, int ngames,
                                                     int lane) {
  using G = FastGeom<M, D>;
  using V = typename VecOf<G::W>::type;
  const int total = ngames * G::Q;
#pragma unroll
  for (int i0 = 0; i0 < G::Q; i0 += kBatch) {
    V v[kBatch];
#pragma unroll
    for (int u = 0; u < kBatch; ++u) {  // the image holds kWave games whatever ngames is
      const int q = lane + (i0 + u < G::Q ? i0 + u : G::Q - 1) * kWave;
      v[u] = *reinterpret_cast<const V*>(lds + slab_chunk_lds<M, D>(q));
    }
#pragma unroll
    for (int u = 0; u < kBatch; ++u) asm volatile("" : "+v"(v[u]));
#pragma unroll
    for (int u = 0; u < kBatch; ++u) {
      const int q = lane + (i0 + u) * kWave;
      if (i0 + u < G::Q && q < total) {
        V* dst = reinterpret_cast<V*>(base + slab_chunk_global<M, D, CONTIG>(q, out_stride));
        if constexpr (NT) __builtin_nontemporal_store(v[u], dst);
        else *dst = v[u];
      }
    }
  }
}

// kBatch: image chunks read per round (4 inside the rollout loop; everything at once for the final store,
// when the registers are free and the LDS latency would otherwise be paid once per round)
// (NT: non-temporal stores for a rollout's final state, see duo_store_slab)
template <int M, int D, int kBatch = 4, bool NT = false>
__device__ inline void fast_store_slab(const float* lds, float* out, int64_t out_stride, int64_t g0,
                                       int ngames, int lane) {
  float* base = out + g0 * out_stride;
  if (out_stride == FastGeom<M, D>::N) fast_store_slab_impl<M, D, true, kBatch, NT>(lds, base, out_stride, ngames, lane);
  else fast_store_slab_impl<M, D, false, kBatch, NT>(lds, base, out_stride, ngames, lane);
}

// ---- image <-> registers --------------------------------------------------------------------------
// one pass over the lane's image: bitmask of the fully available rows + representability
template <int M, int D>
__device__ inline void scan_image(const float* mine, float fill, MaskT<M>& live, bool& ok) {
  using G = FastGeom<M, D>;
  using V = typename VecOf<G::W>::type;
  float p[M * D];
#pragma unroll
  for (int c = 0; c < G::Q; ++c) {
    const V v = *reinterpret_cast<const V*>(mine + c * G::W);
    const float* f = reinterpret_cast<const float*>(&v);
#pragma unroll
    for (int w = 0; w < G::W; ++w) p[c * G::W + w] = f[w];
  }
  live = 0;
  ok = true;
  const uint32_t fill_bits = __float_as_uint(fill);
#pragma unroll
  for (int i = 0; i < M; ++i) {
    // on the bit patterns: a row is available iff its largest pattern is below +inf's ([+0, +inf): no
    // negative, -0, inf or NaN), and it is the padding row iff smallest == largest == the fill value's
    uint32_t hi = __float_as_uint(p[i * D]), lo = hi;
#pragma unroll
    for (int k = 1; k < D; ++k) {
      const uint32_t u = __float_as_uint(p[i * D + k]);
      hi = u > hi ? u : hi;
      lo = u < lo ? u : lo;
    }
    const bool ge = hi < 0x7F800000u;
    const bool fl = (lo == fill_bits) && (hi == fill_bits);
    ok &= (ge | fl);
    live |= ge ? ((MaskT<M>)1 << i) : (MaskT<M>)0;
  }
}

template <int M, int D>
__device__ inline void fill_image(float* mine, float pad) {
  using G = FastGeom<M, D>;
  using V = typename VecOf<G::W>::type;
  V v;
  float* f = reinterpret_cast<float*>(&v);
#pragma unroll
  for (int w = 0; w < G::W; ++w) f[w] = pad;
#pragma unroll
  for (int c = 0; c < G::Q; ++c) *reinterpret_cast<V*>(mine + c * G::W) = v;
}

// The caller's actions of one game, as fetched (hk_step): raw bits, converted only after the slab
// requests have been issued so that all of it shares one HBM round trip.
template <int D>
struct RawActions {
  RawScalar c[D];
  RawScalar axis;
};

__device__ inline int coords_dtype(int kind) {
  return kind == HK_COORDS_CLASS_I32 ? HK_I32
         : (kind == HK_COORDS_CLASS_I64 ? HK_I64 : (kind == HK_COORDS_IN_RECORD ? HK_F32 : kind));
}

template <int D>
__device__ inline void fast_fetch_actions(const Params& prm, int64_t g, int m, RawActions<D>& r) {
  const int kind = prm.coords_kind;
  const bool is_class = kind == HK_COORDS_CLASS_I32 || kind == HK_COORDS_CLASS_I64;
  const bool in_record = kind == HK_COORDS_IN_RECORD;
  const void* base = in_record ? prm.in : prm.coords;
  // element index of the first value; class ids: one value per game (fetched D times, same address)
  const size_t first = is_class ? (size_t)g
                                : (size_t)(in_record ? g * prm.in_stride + (int64_t)m * D : g * prm.coords_stride);
  const size_t step = is_class ? 0 : 1;
  const int dt = coords_dtype(kind);
#pragma unroll
  for (int k = 0; k < D; ++k) r.c[k] = fetch_raw(base, dt, first + step * k);
  r.axis = fetch_raw(prm.axis, prm.axis_dtype, (size_t)g);
}

template <int D>
__device__ inline void fast_decode_actions(const Params& prm, const RawActions<D>& r, float (&c)[D], int& axis) {
  const int kind = prm.coords_kind;
  if (kind == HK_COORDS_CLASS_I32 || kind == HK_COORDS_CLASS_I64) {
    long long cls = (kind == HK_COORDS_CLASS_I32) ? (long long)(int32_t)r.c[0].lo
                                                  : (long long)(((uint64_t)r.c[0].hi << 32) | r.c[0].lo);
    constexpr long long ncls = (1ll << D) - D - 1;
    cls = cls < 0 ? 0 : (cls >= ncls ? ncls - 1 : cls);
    const uint32_t v = decode_class((int)cls, D);
#pragma unroll
    for (int k = 0; k < D; ++k) c[k] = (float)((v >> k) & 1u);
  } else {
    const int dt = coords_dtype(kind);
#pragma unroll
    for (int k = 0; k < D; ++k) c[k] = (float)scalar_from_raw(r.c[k], dt);
  }
  axis = axis_index(scalar_from_raw(r.axis, prm.axis_dtype), D);
}

// random / fixed policies of jax/players.py (not Zeillinger: that one runs on the generic kernel)
// `cache` holds the Philox block of steps {2b, 2b+1}: the fused loop calls Philox every other step.
// `zeillinger_cls`: the class Zeillinger's host picks on the current state (computed by the caller, who holds
// the rows), used when host_policy == HK_HOST_ZEILLINGER
template <int D>
__device__ inline void policy_from_words(uint32_t ra, uint32_t rb, int host_policy, int agent_policy, int& cls,
                                         int& axis, uint32_t& mask, int zeillinger_cls = 0) {
  constexpr uint32_t ncls = (1u << D) - (uint32_t)D - 1u;
  cls = (host_policy == HK_HOST_RANDOM) ? (int)mulhi32(ra, ncls)
                                        : (host_policy == HK_HOST_ZEILLINGER ? zeillinger_cls : (int)ncls - 1);
  mask = decode_class(cls, D);
  if (agent_policy == HK_AGENT_RANDOM) {
    axis = (int)mulhi32(rb, (uint32_t)D);
  } else if (agent_policy == HK_AGENT_RANDOM_LEGAL) {
    const int pick = (int)mulhi32(rb, (uint32_t)__popc(mask));
    int seen = 0;
    axis = 0;
#pragma unroll
    for (int k = 0; k < D; ++k)
      if ((mask >> k) & 1u) {
        if (seen == pick) axis = k;
        ++seen;
      }
  } else if (agent_policy == HK_AGENT_CHOOSE_FIRST) {
    axis = __ffs(mask) - 1;
  } else {
    axis = 31 - __clz(mask);
  }
}

template <int D>
__device__ inline void fast_policy(uint64_t seed, int host_policy, int agent_policy, uint64_t gg, uint32_t step,
                                   PolicyCache& cache, int& cls, int& axis, uint32_t& mask,
                                   int zeillinger_cls = 0) {
  uint32_t ra, rb;
  policy_words(gg, step, seed, cache, D, ra, rb);
  policy_from_words<D>(ra, rb, host_policy, agent_policy, cls, axis, mask, zeillinger_cls);
}

template <int D>
__device__ inline void fast_policy(const Params& prm, uint64_t gg, uint32_t step, PolicyCache& cache,
                                   int& cls, int& axis, uint32_t& mask, int zeillinger_cls = 0) {
  fast_policy<D>(prm.seed, prm.host_policy, prm.agent_policy, gg, step, cache, cls, axis, mask, zeillinger_cls);
}

// ---- the policy stream off the critical path (plain rollouts; see hk_duo_kernel.h) ------------------------------------
// A window of kFastPreBlocks Philox blocks per game (24 steps: four per block) is computed before the first step -- while the wave
// would otherwise only wait for its slab --, two independent chains at a time, DECODED (subset mask | axis << 5) and
// parked in LDS, one byte per game and step; a step reads its byte.  Episodes longer than a window refill it between
// two passes over the staircase.
constexpr int kFastPreBlocks = 6;  // (a block serves four steps: hk_common.h policy_words)

template <int D>
__device__ __forceinline__ void fast_policy_fill(uint8_t* act, uint64_t gg, uint32_t wb0, int nb, uint64_t seed,
                                                 int host_policy, int agent_policy, int lane) {
  static_assert(D <= 5 && D <= kPolicyShortDim, "an action travels as a byte; four steps per Philox block");
#pragma nounroll
  for (int i = 0; i < kFastPreBlocks; i += 2) {
    if (i >= nb) break;  // wave-uniform
    const U4 r0 = philox4x32((uint32_t)gg, (uint32_t)(gg >> 32), wb0 + (uint32_t)i, kStreamPolicy, seed);
    const U4 r1 = philox4x32((uint32_t)gg, (uint32_t)(gg >> 32), wb0 + (uint32_t)i + 1u, kStreamPolicy, seed);
    int cls, axis;
    uint32_t mask;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t w0 = u4_word(r0, k), w1 = u4_word(r1, k);
      policy_from_words<D>(w0 & 0xFFFF0000u, w0 << 16, host_policy, agent_policy, cls, axis, mask, 0);
      act[(4 * i + k) * kWave + lane] = (uint8_t)(mask | ((uint32_t)axis << 5));
      policy_from_words<D>(w1 & 0xFFFF0000u, w1 << 16, host_policy, agent_policy, cls, axis, mask, 0);
      act[(4 * i + 4 + k) * kWave + lane] = (uint8_t)(mask | ((uint32_t)axis << 5));
    }
  }
}

// ---- the kernel: MODE is one of kModeStep / kModeRollout / kModeRolloutRec / kModeGenerate -------
// HOT: a rollout configuration as compile-time constants.  kHotJax: the JAX trainer's rollouts (shift +
// reposition + Newton polytope, JAX semantics without behaviour flags, uniformly random host and agent);
// kHotTorch: the same stages under the torch sibling's semantics (illegal axis / finished game not shifted)
// with the agent drawing among the host's coordinates -- SURVEY 8(d)'s two protocols.  A lone wave per SIMD
// pays a fetch bubble for every taken branch, and the per-step tests of stages / flags / policies are ~15 %
// of a fused rollout (measured: 43.5 -> 38.1 us per 20-step episode of 65 536 games).
constexpr int kHotNone = 0, kHotJax = 1, kHotTorch = 2;
constexpr int kHotList = 4;  // hk::quadroll_kernel: run-time configured, the state published sorted + compacted
constexpr unsigned kHotTorchFlags = HK_SEM_TORCH | HK_FLAG_AXIS_NOOP_IF_INVALID | HK_FLAG_IGNORE_ENDED;

template <int M, int D, int MODE, int HOT = kHotNone>
__global__ __launch_bounds__(kWave, ((MODE == kModeRolloutRec || M * D > 64) ? 1 : 2)) void fast_kernel(const float* in0, int64_t in_stride0, int batch0, int gpb0, const Params prm) {
  // in0 / in_stride0 / batch0 / gpb0 repeat prm.in / in_stride / batch / games_per_block as leading scalar
  // arguments: those are PRELOADED into SGPRs at wave launch (-mllvm -amdgpu-kernarg-preload-count), so the
  // slab's loads can be issued without waiting for the first scalar load of the argument block
  using G = FastGeom<M, D>;
  __shared__ __align__(16) float lds[kWave * G::S];
  __shared__ float cbuf[kWave * D];  // slow path only: subset mask / row scratch per lane
  // plain rollouts on the shapes a byte per action serves: the decoded actions of a window of steps (fast_policy_fill)
  constexpr bool kWindow = MODE == kModeRollout && D <= 5;
  __shared__ __align__(16) uint8_t pol[kWindow ? 4 * kFastPreBlocks * kWave : 16];
  constexpr bool kRec = MODE == kModeRolloutRec;            // rollout + per-step observations / records
  constexpr bool kRoll = MODE == kModeRollout || kRec;
  constexpr bool kStep = MODE == kModeStep || MODE == kModeStepAux;  // Aux: features / Zeillinger's class
  const int lane = threadIdx.x;
  // games per wave: 64 (fewer only through the tuning hook of fast_games_per_block())
  const int gpb = gpb0;
  const int64_t g0 = (int64_t)blockIdx.x * gpb;
  const int64_t left = (int64_t)batch0 - g0;
  const int ngames = (int)(left < gpb ? left : gpb);
  const bool active = lane < ngames;
  const int64_t g = g0 + lane;
  // (the game id's load before the slab's, its arithmetic behind them: hk_duo_kernel.h)
  const bool has_ids = kRoll && prm.game_ids != nullptr;
  uint32_t raw_id = 0;
  if (has_ids && active) raw_id = (uint32_t)prm.game_ids[g];
  __builtin_amdgcn_sched_barrier(0);
  SlabRegs<M, D> slab;
  if (MODE != kModeGenerate) fast_slab_issue<M, D>(slab, in0, in_stride0, g0, ngames, lane);
  __builtin_amdgcn_sched_barrier(0);
  const uint64_t gg = prm.game_offset + (has_ids ? (uint64_t)raw_id : (uint64_t)g);
  // plain rollouts: the first window of decoded actions, computed while the slab is in flight (Zeillinger's host reads
  // the state: its rollouts keep the policies in the loop and never look at the window)
  uint32_t pol_b0 = prm.step_offset >> 2;  // first Philox block of the window (wave-uniform)
  const uint32_t pol_last = (prm.steps > 0) ? (prm.step_offset + (uint32_t)prm.steps - 1u) >> 2 : pol_b0;
  if constexpr (kWindow) {
    if (HOT || prm.host_policy != HK_HOST_ZEILLINGER) {
      const uint32_t nb = pol_last - pol_b0 + 1u;
      fast_policy_fill<D>(pol, gg, pol_b0, (int)(nb < (uint32_t)kFastPreBlocks ? nb : (uint32_t)kFastPreBlocks), prm.seed,
                          HOT ? (int)HK_HOST_RANDOM : prm.host_policy,
                          (HOT == kHotJax) ? (int)HK_AGENT_RANDOM
                                           : (HOT == kHotTorch) ? (int)HK_AGENT_RANDOM_LEGAL : prm.agent_policy, lane);
    }
  }
  float* mine = lds + lane * G::S;
  const float pad = (float)prm.pad;
  const unsigned flags = (HOT == kHotJax) ? (unsigned)HK_SEM_JAX : (HOT == kHotTorch) ? kHotTorchFlags : prm.flags;
  const unsigned stages = HOT ? (unsigned)(HK_STAGE_SHIFT | HK_STAGE_REPOSITION | HK_STAGE_NEWTON)
                              : (MODE == kModeGenerate) ? (prm.stages & ~HK_STAGE_SHIFT) : prm.stages;
  const float fill = ((flags & HK_SEM_MASK) == HK_SEM_JAX) ? -1.0f : pad;  // _jax_ops.py:65 does not forward pad
  const int nsteps = (kRoll) ? prm.steps : 1;
  // list semantics / COMPACT_SORTED: every Newton stage leaves the state sorted (descending lexicographic)
  // and compacted.  Only the variants that are off the hot paths carry the code.
  constexpr bool kSortedCapable = MODE == kModeStepAux || MODE == kModeRolloutRec;
  const bool sorted_out = kSortedCapable && (stages & HK_STAGE_NEWTON) &&
                          ((flags & HK_SEM_MASK) == HK_SEM_LIST || (flags & HK_FLAG_COMPACT_SORTED));
  // A PLAIN rollout (no per-step records) under list semantics sorts once, at the end: between two Newton stages
  // the order of the rows changes nothing (the stage removes duplicates and dominated rows whatever their order;
  // the policies and counts do not look at it), so only the state that leaves the kernel has to be sorted +
  // compacted -- ranked right after the last Newton stage, before a rescale could round two keys together.
  constexpr bool kEndSort = MODE == kModeRollout && HOT == kHotNone;
  const bool end_sort = kEndSort && nsteps > 0 && (stages & HK_STAGE_NEWTON) &&
                        ((flags & HK_SEM_MASK) == HK_SEM_LIST || (flags & HK_FLAG_COMPACT_SORTED));
  PolicyCache pcache;

  // step mode: the action loads join the slab's requests in flight
  float c[D];
  int axis_in = -1;
#pragma unroll
  for (int k = 0; k < D; ++k) c[k] = 0.0f;
  RawActions<D> raw;
  const bool fetch_actions = kStep && (stages & HK_STAGE_SHIFT) && active;
  if (fetch_actions) fast_fetch_actions<D>(prm, g, M, raw);

  // ---- 1. the image --------------------------------------------------------------------------------
  if (MODE == kModeGenerate) {
    // (hk_common.h: eight elements per Philox block for small max_value, four otherwise)
    if (gen_short((uint32_t)prm.max_value)) {
#pragma unroll
      for (int e = 0; e < M * D; e += 8) {
        const U4 r = philox4x32((uint32_t)gg, (uint32_t)(gg >> 32), (uint32_t)(e >> 3), kStreamGenerate, prm.seed);
        uint32_t v[8];
        gen_block_values(r, (uint32_t)prm.max_value, true, v);
#pragma unroll
        for (int qd = 0; qd < 8; ++qd)
          if (e + qd < M * D) mine[e + qd] = (float)v[qd];
      }
    } else {
#pragma unroll
      for (int e = 0; e < M * D; e += 4) {
        const U4 r = philox4x32((uint32_t)gg, (uint32_t)(gg >> 32), (uint32_t)(e >> 2), kStreamGenerate, prm.seed);
        uint32_t v[8];
        gen_block_values(r, (uint32_t)prm.max_value, false, v);
#pragma unroll
        for (int qd = 0; qd < 4; ++qd)
          if (e + qd < M * D) mine[e + qd] = (float)v[qd];
      }
    }
  } else {
    fast_slab_commit<M, D>(slab, lds, ngames, lane);
  }
  if (fetch_actions) fast_decode_actions<D>(prm, raw, c, axis_in);
  __syncthreads();

  // ---- 2. live rows, exactness guard ------------------------------------------------------------
  MaskT<M> gmask;
  bool ok;
  scan_image<M, D>(mine, fill, gmask, ok);
  if (!active) {  // lanes past the batch read stale LDS: give them an empty, harmless game
    gmask = 0;
    ok = true;
  }
  int np = mask_pop(gmask);
  int nmax = wave_max(np, M);
  const bool exact = (fill == pad) && __all(ok) && nmax <= G::C;

  if (!exact) {
    // ---- slow path (whole wave): the exact generic routines on the image ------------------------
    float* cs = cbuf + lane * D;
    if (MODE == kModeStepAux && prm.class_out) {  // hk_zeillinger: the class is the only output
      if (active)
        prm.class_out[g] = ((flags & HK_SEM_MASK) == HK_SEM_LIST) ? zeillinger_list_game<float>(mine, prm.m, prm.d)
                                                                  : zeillinger_game<float>(mine, prm.m, prm.d);
      return;
    }
    np = active ? num_points<float>(mine, M, D) : 2;
    int length = (np < 2) ? 0 : -1;
    if (kRoll && prm.count_ws) {
      const unsigned long long b0 = __ballot(active && np < 2);
      if (lane == 0) count_add(prm.count_ws + blockIdx.x, (uint32_t)__popcll(b0));
    }
    for (int t = 0; t < nsteps; ++t) {
      int axis = -1, cls = 0;
      if (kRoll) {
        if (prm.obs_out) {
          __syncthreads();
          fast_store_slab<M, D>(lds, (float*)prm.obs_out + (int64_t)t * prm.batch * G::N, (int64_t)G::N,
                                g0, ngames, lane);
          __syncthreads();
        }
        uint32_t mask;
        const int zc = (prm.host_policy == HK_HOST_ZEILLINGER && active) ? zeillinger_game<float>(mine, prm.m, prm.d) : 0;
        fast_policy<D>(prm, gg, prm.step_offset + (uint32_t)t, pcache, cls, axis, mask, zc);
        for (int k = 0; k < prm.d; ++k) cs[k] = (float)((mask >> k) & 1u);
      } else if (kStep && (stages & HK_STAGE_SHIFT) && active) {
        load_coords<float>(prm, g, cs);
        axis = axis_in;
      }
      const bool prev_done = np < 2;
      if (active) stages_game<float>(mine, prm.m, prm.d, cs, axis, pad, stages, flags);  // runtime bounds: rolled
      np = active ? num_points<float>(mine, prm.m, prm.d) : 2;
      const bool done = np < 2;
      if (done && length < 0) length = t + 1;
      if (kRoll) {
        if (active) {
          const int64_t at = (int64_t)t * prm.batch + g;
          if (prm.r_host_class_out) prm.r_host_class_out[at] = cls;
          if (prm.r_axis_out) prm.r_axis_out[at] = axis;
          if (prm.r_done_out) prm.r_done_out[at] = done;
          if (prm.r_reward_out) prm.r_reward_out[at] = prm.reward_sign * (float)(done && !prev_done);
        }
        if (prm.count_ws) {
          const unsigned long long bd = __ballot(active && done);
          if (lane == 0) count_add(prm.count_ws + (size_t)(t + 1) * prm.count_stride + blockIdx.x, (uint32_t)__popcll(bd));
        }
      } else if (kStep && active) {
        if (prm.done_out) prm.done_out[g] = done;
        if (prm.prev_done_out) prm.prev_done_out[g] = prev_done;
        if (prm.reward_out) prm.reward_out[g] = prm.reward_sign * (float)(done && !prev_done);
        if (prm.num_points_out) prm.num_points_out[g] = np;
      }
    }
    if (kRoll && active && prm.game_length_out) prm.game_length_out[g] = length;
    __syncthreads();
    fast_store_slab<M, D>(lds, (float*)prm.out, prm.out_stride, g0, ngames, lane);
    return;
  }

  // ---- 3. gather the live rows ------------------------------------------------------------------------
  float q[G::C * D];
#pragma unroll
  for (int e = 0; e < G::C * D; ++e) q[e] = INFINITY;  // rows past nmax are holes in the straight-line bodies
  gather_rows<M, G::C, D>(q, mine, gmask, nmax);
  if (MODE == kModeStepAux && prm.class_out) {  // hk_zeillinger: the class is the only output
    if (active)
      prm.class_out[g] = c_zeillinger<G::C, D, true>(q, nmax, (flags & HK_SEM_MASK) == HK_SEM_LIST);
    return;
  }
  if (!active) np = 2;  // never "done", never counted
  int length = (np < 2) ? 0 : -1;
  if (kRoll && prm.count_ws) {
    const unsigned long long b0 = __ballot(active && np < 2);
    if (lane == 0) count_add(prm.count_ws + blockIdx.x, (uint32_t)__popcll(b0));
  }

  // ---- 4. the transitions --------------------------------------------------------------------------
  // Wave-uniform facts about the optional outputs are computed ONCE: tested per step through the
  // kernel-argument struct, SGPR pressure makes the compiler re-issue the 64-byte s_load of the pointer
  // block (and a full s_waitcnt) several times per step.
  const bool want_obs = kRec && prm.obs_out != nullptr;
  const bool want_records = kRec && (prm.r_host_class_out || prm.r_axis_out ||
                                                       prm.r_done_out || prm.r_reward_out);
  uint32_t* count_slot = (kRoll && prm.count_ws) ? prm.count_ws + blockIdx.x : nullptr;
  uint32_t count_stride = prm.count_stride;
  uint32_t step0 = prm.step_offset;
  uint64_t seed = prm.seed;
  int host_policy = HOT ? (int)HK_HOST_RANDOM : prm.host_policy;
  int agent_policy = (HOT == kHotJax) ? (int)HK_AGENT_RANDOM
                                      : (HOT == kHotTorch) ? (int)HK_AGENT_RANDOM_LEGAL : prm.agent_policy;
  // opaque to the optimiser: the values now "come from" the asm, so they stay in SGPRs (or a VGPR lane)
  // instead of being re-loaded from the kernel-argument / dispatch memory inside the loop
  if (HOT)
    asm volatile("" : "+s"(count_slot), "+s"(count_stride), "+s"(step0), "+s"(seed));
  else
    asm volatile("" : "+s"(count_slot), "+s"(count_stride), "+s"(step0), "+s"(seed), "+s"(host_policy),
                 "+s"(agent_policy));
  bool rescale_pending = false;
  // (Zeillinger's host scans all pairs of rows before every step: its rollouts keep the single loop below)
  const bool staircase = MODE == kModeRollout && (HOT || host_policy != HK_HOST_ZEILLINGER);
  if (MODE == kModeRollout && staircase) {
    // ---- plain rollouts: a STAIRCASE of loops, one per bucket of rows, entered from the top down (nmax never
    // grows): each loop is straight-line for its bucket -- no per-step dispatch over the buckets, and only the rows
    // the bucket covers are loop-carried registers (see hk_duo_kernel.h) --------------------------------------------
    static_assert(G::C <= 8 || G::C % 2 == 0, "bucket ladder: 1..8, then even numbers");
    int t = 0;
    bool stop = false;
    while (t < nsteps && !stop) {  // one pass per window of actions (episodes of up to 24 steps: one pass)
      int tw = nsteps;
      if constexpr (kWindow) {
        if ((uint32_t)((step0 + (uint32_t)t) >> 2) - pol_b0 >= (uint32_t)kFastPreBlocks) {
          __syncthreads();
          pol_b0 = (step0 + (uint32_t)t) >> 2;
          const uint32_t nb = pol_last - pol_b0 + 1u;
          fast_policy_fill<D>(pol, gg, pol_b0, (int)(nb < (uint32_t)kFastPreBlocks ? nb : (uint32_t)kFastPreBlocks), seed,
                              host_policy, agent_policy, lane);
          __syncthreads();
        }
        const uint32_t wend_abs = (pol_b0 + (uint32_t)kFastPreBlocks) << 2;  // last step (exclusive) the window covers
        tw = (wend_abs - step0 < (uint32_t)nsteps) ? (int)(wend_abs - step0) : nsteps;
      }
      RowLevels<G::C>::run([&](auto nbc, auto loc) {
        constexpr int NB = decltype(nbc)::value, LO = decltype(loc)::value;
        while (t < tw && (nmax > LO || LO == 0) && !stop) {  // (a wave of empty games has nmax 0: the last loop's)
          int axis, cls;
          uint32_t mask;
          if constexpr (kWindow) {
            const uint32_t a = pol[(int)(step0 + (uint32_t)t - (pol_b0 << 2)) * kWave + lane];
            mask = a & 31u;
            axis = (int)(a >> 5);
          } else {
            fast_policy<D>(seed, host_policy, agent_policy, gg, step0 + (uint32_t)t, pcache, cls, axis, mask, 0);
          }
          // (end_sort: the last step's rescale waits until the rows are ranked, after the loop)
          const unsigned st = (end_sort && t + 1 == nsteps) ? (stages & ~(unsigned)HK_STAGE_RESCALE) : stages;
          rescale_pending = end_sort && t + 1 == nsteps && (stages & HK_STAGE_RESCALE);
          np = b_stages<G::C, D, NB, true>(q, c, axis, np, flags, st, mask);
          if (!active) np = 2;
          const bool done = np < 2;
          if (done && length < 0) length = t + 1;
          // (the finished-game counts: ballots over the games' first finished steps after the loop -- a finished game
          // stays finished --, not an atomic per step in here)
          if constexpr (NB == 1) {
            // Fixed point (see hk_duo_kernel.h): once every game of the wave is down to one point at the origin (or
            // none) nothing changes any more
            if (t + 1 < nsteps && !__any(active && !done)) {
              bool still = true;
#pragma unroll
              for (int k = 0; k < D; ++k) still &= (q[k] == 0.0f);
              still |= !(q[0] < INFINITY);
              if (!__any(active && !still)) stop = true;
            }
          } else {
            // re-gather when the widest game of the wave got narrower (removed rows are holes until then)
            if (t + 1 < nsteps && !__any(active && np >= nmax)) {
              gmask = scatter_rows<M, G::C, D, NB>(q, mine, gmask, nmax);
              const int nprev = nmax;
              nmax = wave_max(active ? np : 0, nmax - 1);
              gather_rows<M, G::C, D, NB>(q, mine, gmask, nprev);  // rows [nmax, nprev) become holes again
            }
          }
          ++t;
        }
      });
    }
    // games whose first finished step is <= s, for every step s >= 1 (s = 0 was counted at entry)
    if (count_slot) add_length_counts(count_slot, count_stride, 1, nsteps, active, length, lane);
  }
  for (int t = 0; !staircase && t < nsteps; ++t) {  // single steps, recording rollouts, the aux step modes, Zeillinger
    int axis = -1, cls = 0;
    uint32_t mask = 0;  // rollouts: the policy's subset as a 0/1 mask (the shift as selects, b_shift_mask)
    if (kRoll) {
      if (want_obs) {  // state before the step: rebuild the image, store it coalesced
        __syncthreads();
        fill_image<M, D>(mine, pad);
        scatter_rows<M, G::C, D>(q, mine, gmask, nmax);
        __syncthreads();
        fast_store_slab<M, D>(lds, (float*)prm.obs_out + (int64_t)t * prm.batch * G::N, (int64_t)G::N,
                              g0, ngames, lane);
      }
      int zc = 0;
      if (!HOT && host_policy == HK_HOST_ZEILLINGER) zc = c_zeillinger<G::C, D>(q, nmax);
      fast_policy<D>(seed, host_policy, agent_policy, gg, step0 + (uint32_t)t, pcache, cls, axis, mask, zc);
    } else if (kStep) {
      axis = axis_in;  // c[] and the axis were fetched at kernel entry
    }
    const bool prev_done = np < 2;

    if (sorted_out) {
      // list semantics: right after the Newton stage (before a rescale could round two keys together) the
      // survivors are sorted descending-lexicographically and packed to the front -- physically: rows to
      // their rank in the image, slots 0..n-1 become the game's live slots, registers re-gathered in that order
      np = run_stages<G::C, D, kRoll>(q, nmax, c, axis, np, flags, stages & ~(unsigned)HK_STAGE_RESCALE, mask);
      int rank[G::C];
      feature_ranks<G::C, D, kKeyFirst>(q, nmax, rank);
      __syncthreads();
      scatter_ranked<G::C, D>(q, mine, rank, nmax);
      __syncthreads();
      gmask = (np >= (int)(8 * sizeof(MaskT<M>))) ? ~(MaskT<M>)0 : (((MaskT<M>)1 << np) - 1);
      const int nprev = nmax;
      nmax = wave_max(active ? np : 0, nmax);
      gather_rows<M, G::C, D>(q, mine, gmask, nprev);
      if (stages & HK_STAGE_RESCALE) np = run_stages<G::C, D>(q, nmax, c, axis, np, flags, HK_STAGE_RESCALE);
    } else {
      // (end_sort: the last step's rescale waits until the rows are ranked, after the loop)
      const unsigned st = (end_sort && t + 1 == nsteps) ? (stages & ~(unsigned)HK_STAGE_RESCALE) : stages;
      np = run_stages<G::C, D, kRoll>(q, nmax, c, axis, np, flags, st, mask);  // branch-free body for >= nmax rows
      rescale_pending = end_sort && t + 1 == nsteps && (stages & HK_STAGE_RESCALE);
    }
    if (!active) np = 2;
    const bool done = np < 2;
    if (done && length < 0) length = t + 1;
    if (kRoll) {
      if (want_records && active) {
        const int64_t at = (int64_t)t * prm.batch + g;
        if (prm.r_host_class_out) prm.r_host_class_out[at] = cls;
        if (prm.r_axis_out) prm.r_axis_out[at] = axis;
        if (prm.r_done_out) prm.r_done_out[at] = done;
        if (prm.r_reward_out) prm.r_reward_out[at] = prm.reward_sign * (float)(done && !prev_done);
      }
      const unsigned long long bd = __ballot(active && done);
      if (count_slot && lane == 0) count_add(count_slot + (size_t)(t + 1) * count_stride, (uint32_t)__popcll(bd));
      // Fixed point (see hk_duo_kernel.h): once every game of the wave is down to one point at the origin (or none)
      // nothing changes any more; the rest of the episode is the finished-game counts, in closed form.
      if (MODE == kModeRollout && nmax == 1 && bd == __ballot(active) && t + 1 < nsteps) {
        bool still = true;
#pragma unroll
        for (int k = 0; k < D; ++k) still &= (q[k] == 0.0f);
        still |= !(q[0] < INFINITY);
        if (!__any(active && !still)) {
          if (count_slot && lane == 0)
            for (int tt = t + 1; tt < nsteps; ++tt) count_add(count_slot + (size_t)(tt + 1) * count_stride, (uint32_t)__popcll(bd));
          break;
        }
      }
      // re-gather when the widest game of the wave got narrower (removed rows are holes until then)
      // (one ballot per step; the cross-lane maximum only when some game still fills all nmax rows)
      if (t + 1 < nsteps && !__any(active && np >= nmax)) {
        gmask = scatter_rows<M, G::C, D>(q, mine, gmask, nmax);
        const int nprev = nmax;
        nmax = wave_max(active ? np : 0, nmax - 1);
        gather_rows<M, G::C, D>(q, mine, gmask, nprev);  // rows [nmax, nprev) become holes again
      }
    } else if (kStep && active) {
      if (prm.done_out) prm.done_out[g] = done;
      if (prm.prev_done_out) prm.prev_done_out[g] = prev_done;
      if (prm.reward_out) prm.reward_out[g] = prm.reward_sign * (float)(done && !prev_done);
      if (prm.num_points_out) prm.num_points_out[g] = np;
    }
  }
  if (kRoll && active && prm.game_length_out) prm.game_length_out[g] = length;

  // ---- 5. publish: pad everywhere, live rows back in their slots (or, for the observation features,
  // at their rank in descending key order: padding rows are all equal and end up behind them) -------
  __syncthreads();
  fill_image<M, D>(mine, pad);
  if (MODE == kModeStepAux && (stages & kStageFeatureSorts)) {
    int rank[G::C];
    feature_ranks<G::C, D, kKeyLast>(q, nmax, rank, (stages & kStageFeatureSort0) != 0);
    scatter_ranked<G::C, D>(q, mine, rank, nmax);
  } else if (kEndSort && end_sort) {
    int rank[G::C];
    feature_ranks<G::C, D, kKeyFirst>(q, nmax, rank);
    if (rescale_pending) c_rescale<G::C, D>(q, nmax, flags);
    scatter_ranked<G::C, D>(q, mine, rank, nmax);
  } else {
    scatter_rows<M, G::C, D>(q, mine, gmask, nmax);
  }
  __syncthreads();
  fast_store_slab<M, D, G::Q, (MODE == kModeRollout)>(lds, (float*)prm.out, prm.out_stride, g0, ngames, lane);
}

// ---- the specialisation table ------------------------------------------------------------------
// (max_points, dim): BASELINE configs (10,3) (20,3) (50,4) plus the small shapes the reference's
// tests and YAMLs use.
#ifndef HK_FAST_SPECS  // (a build experiment may narrow the table)
#define HK_FAST_SPECS(X) X(4, 3) X(5, 3) X(10, 3) X(16, 3) X(20, 3) X(8, 4) X(20, 4)
#endif

inline int has_fast_path(int m, int d, int dtype) {
  if (dtype != HK_F32) return 0;
#define HK_X(M_, D_) if (m == M_ && d == D_) return 1;
  HK_FAST_SPECS(HK_X)
#undef HK_X
  return 0;
}

// vector slab I/O needs W-aligned records
template <int M, int D>
bool fast_aligned_t(const Params& prm) {
  using G = FastGeom<M, D>;
  const size_t vec_bytes = G::W * 4;
  if ((prm.in && (reinterpret_cast<uintptr_t>(prm.in) % vec_bytes)) ||
      (reinterpret_cast<uintptr_t>(prm.out) % vec_bytes) || (prm.in_stride % G::W) ||
      (prm.out_stride % G::W))
    return false;
  if (prm.obs_out && (reinterpret_cast<uintptr_t>(prm.obs_out) % vec_bytes)) return false;
  return true;
}

// One lane per game means one INSTRUCTION STREAM per 64 games; a wave's stream is latency-bound on its
// own (~7.7 cycles per instruction measured, whether or not it shares its SIMD), so 65 536 games = 1024
// waves leave the 1024 SIMDs half idle and the same kernels run ~2x more games per second at >= 262 144
// games.  Putting fewer games in a wave (idle lanes, more waves) does not buy that back: measured at
// 65 536 games, 64/32/16 games per wave give 50.4/50.7/84.8 us per 20-step rollout and 11.6/12.3/18.5 us
// per hk_step, so a wave always takes 64 games.
inline int fast_games_per_block(const Params&) { return kWave; }

inline int fast_hot_config(const Params& prm) {
  if (prm.stages != (HK_STAGE_SHIFT | HK_STAGE_REPOSITION | HK_STAGE_NEWTON) || prm.host_policy != HK_HOST_RANDOM)
    return kHotNone;
  if (prm.flags == HK_SEM_JAX && prm.agent_policy == HK_AGENT_RANDOM) return kHotJax;
  if (prm.flags == kHotTorchFlags && prm.agent_policy == HK_AGENT_RANDOM_LEGAL) return kHotTorch;
  return kHotNone;
}

// The plain rollouts of a shape (three kernels: run-time configured, kHotJax, kHotTorch -- the staircase makes them the
// longest to compile) live in a translation unit of their own (hk_fast_roll_spec.hip), the other modes in
// hk_fast_spec.hip: the build's critical path halves.
template <int M, int D>
int launch_fast_roll_t(const Params& prm, unsigned grid, hipStream_t stream) {
  const int hot = fast_hot_config(prm);
  if (hot == kHotJax)
    hipLaunchKernelGGL((fast_kernel<M, D, kModeRollout, kHotJax>), dim3(grid), dim3(kWave), 0, stream, (const float*)prm.in,
                       prm.in_stride, prm.batch, prm.games_per_block, prm);
  else if (hot == kHotTorch)
    hipLaunchKernelGGL((fast_kernel<M, D, kModeRollout, kHotTorch>), dim3(grid), dim3(kWave), 0, stream, (const float*)prm.in,
                       prm.in_stride, prm.batch, prm.games_per_block, prm);
  else
    hipLaunchKernelGGL((fast_kernel<M, D, kModeRollout>), dim3(grid), dim3(kWave), 0, stream, (const float*)prm.in,
                       prm.in_stride, prm.batch, prm.games_per_block, prm);
  return launch_status();
}

#ifndef HK_FAST_ROLL_TU
#define HK_X(M_, D_) extern template int launch_fast_roll_t<M_, D_>(const Params&, unsigned, hipStream_t);
HK_FAST_SPECS(HK_X)
#undef HK_X
#endif

template <int M, int D>
int launch_fast_t(Params prm, hipStream_t stream) {
  prm.games_per_block = fast_games_per_block(prm);
  const unsigned grid = (unsigned)(((int64_t)prm.batch + prm.games_per_block - 1) / prm.games_per_block);
  launch_prepare();
  const bool sorted_out = (prm.stages & HK_STAGE_NEWTON) &&
                          ((prm.flags & HK_SEM_MASK) == HK_SEM_LIST || (prm.flags & HK_FLAG_COMPACT_SORTED));
  if (prm.mode == kModeStep && (prm.class_out || (prm.stages & kStageFeatureSorts) || sorted_out))
    hipLaunchKernelGGL((fast_kernel<M, D, kModeStepAux>), dim3(grid), dim3(kWave), 0, stream, (const float*)prm.in,
                       prm.in_stride, prm.batch, prm.games_per_block, prm);
  else if (prm.mode == kModeStep)
    hipLaunchKernelGGL((fast_kernel<M, D, kModeStep>), dim3(grid), dim3(kWave), 0, stream, (const float*)prm.in,
                       prm.in_stride, prm.batch, prm.games_per_block, prm);
  else if (prm.mode == kModeRollout && (prm.obs_out || prm.r_host_class_out || prm.r_axis_out ||
                                        prm.r_done_out || prm.r_reward_out))
    hipLaunchKernelGGL((fast_kernel<M, D, kModeRolloutRec>), dim3(grid), dim3(kWave), 0, stream, (const float*)prm.in,
                       prm.in_stride, prm.batch, prm.games_per_block, prm);
  else if (prm.mode == kModeRollout)
    return launch_fast_roll_t<M, D>(prm, grid, stream);
  else
    hipLaunchKernelGGL((fast_kernel<M, D, kModeGenerate>), dim3(grid), dim3(kWave), 0, stream, (const float*)prm.in,
                       prm.in_stride, prm.batch, prm.games_per_block, prm);
  return launch_status();
}

// does this request run on a register-resident specialisation? (else: generic kernel)
inline bool fast_supported(const Params& prm, int dtype) {
  if (dtype != HK_F32) return false;
  // sorted + compacted output (list semantics): not from the generator, and not under Zeillinger's host,
  // whose tie-breaks follow the physical row order
  const bool sorted_out = (prm.stages & HK_STAGE_NEWTON) &&
                          ((prm.flags & HK_SEM_MASK) == HK_SEM_LIST || (prm.flags & HK_FLAG_COMPACT_SORTED));
  if (sorted_out && (prm.mode == kModeGenerate ||
                     (prm.mode == kModeRollout && prm.host_policy == HK_HOST_ZEILLINGER)))
    return false;
  if (prm.flags & (HK_FLAG_FORCE_GENERIC | HK_FLAG_FORCE_TEAM)) return false;
  if ((prm.stages & kStageFeatureSorts) && prm.mode != kModeStep) return false;
  if (prm.mode == kModeZeillinger) return false;
#define HK_X(M_, D_) if (prm.m == M_ && prm.d == D_) return fast_aligned_t<M_, D_>(prm);
  HK_FAST_SPECS(HK_X)
#undef HK_X
  return false;
}

// Each (max_points, dim) specialisation is compiled in its own translation unit (hk_fast_spec.hip, built
// once per entry of the table by the Makefile, in parallel); the dispatcher below exists only in the main
// unit, where the specialisations are declared but not instantiated.
#ifndef HK_SPEC_TU
#define HK_X(M_, D_) extern template int launch_fast_t<M_, D_>(Params, hipStream_t);
HK_FAST_SPECS(HK_X)
#undef HK_X

inline int launch_fast(const Params& prm, hipStream_t stream) {
#define HK_X(M_, D_) if (prm.m == M_ && prm.d == D_) return launch_fast_t<M_, D_>(prm, stream);
  HK_FAST_SPECS(HK_X)
#undef HK_X
  return HK_ERR_UNSUPPORTED;
}
#endif

#ifndef HK_SPEC_TU  // the per-shape translation units hold only their own specialisations
// ---- finished-game counters: per-workgroup partials -> done_count ---------------------------------
// block (t, seg) sums a segment of count_ws[t][0..nblocks) -- all of its loads in flight together -- adds it
// to done_count[t] (a handful of atomics per step instead of nblocks on one cache line) and leaves the
// partials zeroed for the next launches.
constexpr int kCountSeg = 2048;  // entries per block: 8 per thread

// (round 3: a wave's sum by DPP row shifts + broadcasts instead of six shuffles through the LDS crossbar, one 64-bit
// atomic per wave, no barrier.  An episode + its reduction did not get shorter by it -- 25.4 us either way at
// 65 536 games: the ~4.5 us a reduction costs behind a rollout are the dependent launch, not its instructions.  The
// alternative -- short rollouts adding their counts straight into the caller's 64-bit accumulator, one atomic
// instruction per wave -- serialises 4096 atomics on two addresses: one step of 65 536 games 37 us instead of 10.9)
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
  // inclusive scan within rows of 16 lanes, then the rows' totals carried over: lane 63 holds the wave's sum
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);  // row_shr:1
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);  // row_shr:2
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);  // row_shr:4
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);  // row_shr:8
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);  // row_bcast:15 -> rows 1, 3
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);  // row_bcast:31 -> rows 2, 3
  return (uint32_t)__builtin_amdgcn_readlane((int)v, kWave - 1);
}

__global__ __launch_bounds__(256) void count_reduce_kernel(uint32_t* ws, int nblocks,
                                                           unsigned long long* done_count) {
  uint32_t* row = ws + (size_t)blockIdx.x * nblocks;
  const int i0 = blockIdx.y * kCountSeg + threadIdx.x;
  uint32_t v[kCountSeg / 256];
#pragma unroll
  for (int k = 0; k < kCountSeg / 256; ++k) {
    const int i = i0 + k * 256;
    v[k] = (i < nblocks) ? row[i] : 0u;
  }
  uint32_t s = 0;  // (a slot counts games of one workgroup: a thread's eight slots stay far below 2^32)
#pragma unroll
  for (int k = 0; k < kCountSeg / 256; ++k) {
    const int i = i0 + k * 256;
    if (v[k]) row[i] = 0;
    s += v[k];
  }
  const uint32_t tot = wave_sum_u32(s);
  if ((threadIdx.x & (kWave - 1)) == 0 && tot) atomicAdd(&done_count[blockIdx.x], (unsigned long long)tot);
}

inline int launch_count_reduce(uint32_t* ws, int nblocks, int steps, unsigned long long* done_count,
                               hipStream_t stream) {
  launch_prepare();
  hipLaunchKernelGGL(count_reduce_kernel, dim3(steps + 1, (nblocks + kCountSeg - 1) / kCountSeg), dim3(256), 0,
                     stream, ws, nblocks, done_count);
  return launch_status();
}

// ---- small utility kernels ------------------------------------------------------------------------

template <typename T>
__global__ void counts_kernel(const T* points, int64_t stride, uint8_t* done_out, int32_t* num_out,
                              int batch, int m, int d) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= batch) return;
  const T* p = points + g * stride;
  int n = 0;
  for (int i = 0; i < m; ++i) n += (p[(int64_t)i * d] >= (T)0) ? 1 : 0;
  if (done_out) done_out[g] = n < 2;
  if (num_out) num_out[g] = n;
}

inline int launch_counts(const void* points, int64_t stride, uint8_t* done_out, int32_t* num_out,
                         int batch, int m, int d, int dtype, hipStream_t stream) {
  const unsigned grid = (unsigned)((batch + 255) / 256);
  launch_prepare();
  if (dtype == HK_F32)
    hipLaunchKernelGGL(counts_kernel<float>, dim3(grid), dim3(256), 0, stream, (const float*)points,
                       stride, done_out, num_out, batch, m, d);
  else
    hipLaunchKernelGGL(counts_kernel<double>, dim3(grid), dim3(256), 0, stream, (const double*)points,
                       stride, done_out, num_out, batch, m, d);
  return launch_status();
}

__global__ void decode_kernel(const int32_t* cls, void* mask_out, int mask_dtype, int batch, int d) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)batch * d) return;
  const int g = (int)(idx / d), k = (int)(idx % d);
  const int ncls = (int)((1ll << d) - d - 1);
  int c = cls[g];
  c = c < 0 ? 0 : (c >= ncls ? ncls - 1 : c);
  const int bit = (decode_class(c, d) >> k) & 1u;
  switch (mask_dtype) {
    case HK_F32: ((float*)mask_out)[idx] = (float)bit; break;
    case HK_F64: ((double*)mask_out)[idx] = (double)bit; break;
    case HK_I32: ((int32_t*)mask_out)[idx] = bit; break;
    case HK_I64: ((long long*)mask_out)[idx] = bit; break;
    default: ((uint8_t*)mask_out)[idx] = (uint8_t)bit; break;
  }
}

inline int launch_decode(const int32_t* cls, void* mask_out, int mask_dtype, int batch, int d,
                         hipStream_t stream) {
  const int64_t total = (int64_t)batch * d;
  const unsigned grid = (unsigned)((total + 255) / 256);
  launch_prepare();
  hipLaunchKernelGGL(decode_kernel, dim3(grid), dim3(256), 0, stream, cls, mask_out, mask_dtype, batch, d);
  return launch_status();
}

#endif  // HK_SPEC_TU

}  // namespace hk
