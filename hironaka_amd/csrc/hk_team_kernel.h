// Team kernel: FOUR lanes per game, for the shapes without a register-resident specialisation
// (float32, in-place-order semantics, dim 2..6, max_points <= 64 -- BASELINE's (50, 4) above all).
//
// Why not one lane per game here.  A (50, 4) game is 800 B; 64 of them fill 51 KiB of LDS, so a CU
// holds three one-lane-per-game waves -- less than one per SIMD -- and each of them walks its rows
// through LDS with nothing to hide the latency behind (the first version of this path did exactly that:
// 392 us per step at 262 144 games, 13 % of the HBM roofline; this kernel: 111 us, 48 %).  Giving a game
// to a QUAD of lanes divides the LDS footprint and the per-wave instruction stream by four (16 games,
// 13 KiB per wave -> ~11 waves per CU) and puts the rows back in registers:
//
//   * live rows are compacted (original order kept); compact row r lives in lane r % 4 of the team,
//     register slot r / 4 (<= 16 slots = 64 rows).  Rows [n, 4*smax) are +inf holes; smax = the
//     wave-uniform number of slots in use, every slot loop leaves on it with a scalar branch.
//   * shift / reposition / rescale work on the registers; column minima / the game maximum are
//     finished with two DPP quad_perm exchanges (no LDS, no ballots).
//   * the domination test: the team mirrors its rows into the game's LDS region, then every lane
//     tests its rows i against the rows j > i read from LDS (one broadcast ds_read per j for the whole
//     team), each unordered pair once, ~n^2/8 row pairs per lane:
//         t = max_k(P_j - P_i), u = min_k(P_j - P_i);  j removed iff u >= 0;  i removed iff t <= 0 and u < 0
//     -- _jax_ops.py:15-73 in one pass (ties keep the lower index); the sign of a float difference is
//     exact, so this is the reference's `diff >= 0` test.
//   * the bitmask of live ORIGINAL slots is kept team-uniform; removed rows become holes and the
//     rows are squeezed again only when the widest game of the wave got narrower.
//   * publish: pad everywhere, live rows back at their original slots, coalesced slab store.
//
// The exactness guard and the whole-wave slow path (exact generic routines on the image, one lane
// per game) are those of the other kernels.
#pragma once

#include "hk_rows_io.h"

namespace hk {

constexpr int kTeam = 4;
constexpr int kTeamGames = kWave / kTeam;
constexpr int kTeamSlots = 16;  // register slots per lane: 4 * 16 = 64 rows

// ---- exchanges inside a quad: DPP quad_perm [1,0,3,2] and [2,3,0,1] ------------------------------------
template <int CTRL>
__device__ __forceinline__ int quad_perm_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
constexpr int kQuadXor1 = 0xB1, kQuadXor2 = 0x4E;

__device__ __forceinline__ float quad_min(float v) {
  v = hk_fmin(v, __int_as_float(quad_perm_i<kQuadXor1>(__float_as_int(v))));
  return hk_fmin(v, __int_as_float(quad_perm_i<kQuadXor2>(__float_as_int(v))));
}
__device__ __forceinline__ float quad_max(float v) {
  v = hk_fmax(v, __int_as_float(quad_perm_i<kQuadXor1>(__float_as_int(v))));
  return hk_fmax(v, __int_as_float(quad_perm_i<kQuadXor2>(__float_as_int(v))));
}
__device__ __forceinline__ uint32_t quad_or(uint32_t v) {
  v |= (uint32_t)quad_perm_i<kQuadXor1>((int)v);
  return v | (uint32_t)quad_perm_i<kQuadXor2>((int)v);
}
__device__ __forceinline__ Mask64 quad_or64(Mask64 v) {
  return ((Mask64)quad_or((uint32_t)(v >> 32)) << 32) | quad_or((uint32_t)v);
}

// ---- rows <-> registers --------------------------------------------------------------------------------
template <int S_, int W>
__device__ __forceinline__ int orig_slot(const uint32_t (&orig)[W]) {
  return (int)((orig[S_ / 4] >> (8 * (S_ % 4))) & 0xFFu);
}

// Lane `tl` takes the live rows number tl, tl+4, ... (in original order) of `mask`; orig[s] = the row's
// original slot.  from_compact: the row values sit at their RANK in the region (after a squeeze),
// otherwise at their original slot (the image).
// Slots [0, sfill) are written: those past the lane's rows become holes (sfill = the previous smax, so
// that no stale row survives a squeeze).
template <int D, int C>
__device__ __forceinline__ void team_gather(float (&q)[C * D], uint32_t (&orig)[C / 4], const float* mine, Mask64 mask,
                                            int tl, int sfill, bool from_compact) {
  Mask64 mk = mask;
  if (tl > 0) mk &= mk - 1;
  if (tl > 1) mk &= mk - 1;
  if (tl > 2) mk &= mk - 1;
#pragma unroll
  for (int w = 0; w < C / 4; ++w) orig[w] = 0;
  unrolled_while<0, C>([&](auto sc) {
    constexpr int s = decltype(sc)::value;
    if (s >= sfill) return false;
    const bool has = mk != 0;
    const int o = has ? (__ffsll(mk) - 1) : 0;
    orig[s / 4] |= (uint32_t)o << (8 * (s % 4));
    float v[D];
    row_load<D>(mine + (from_compact ? (has ? kTeam * s + tl : 0) : o) * D, v);
#pragma unroll
    for (int k = 0; k < D; ++k) q[s * D + k] = has ? v[k] : INFINITY;
    mk &= mk - 1;
    mk &= mk - 1;
    mk &= mk - 1;
    mk &= mk - 1;
    return true;
  });
}

// registers -> rows 4s+tl of the region (holes included: they must read as +inf in the test below)
template <int D, int C>
__device__ __forceinline__ void team_mirror(const float (&q)[C * D], float* mine, int tl, int smax) {
  unrolled_while<0, C>([&](auto sc) {
    constexpr int s = decltype(sc)::value;
    if (s >= smax) return false;
    float v[D];
#pragma unroll
    for (int k = 0; k < D; ++k) v[k] = q[s * D + k];
    row_store<D>(mine + (kTeam * s + tl) * D, v);
    return true;
  });
}

// _jax_ops.py:114-123 / _torch_ops.py:113-133
template <int D, int C>
__device__ __forceinline__ void team_reposition(float (&q)[C * D], int smax, unsigned flags) {
  const bool jax_sem = (flags & HK_SEM_MASK) == HK_SEM_JAX;
  float mn[D];
#pragma unroll
  for (int k = 0; k < D; ++k) mn[k] = INFINITY;
  unrolled_while<0, C>([&](auto sc) {
    constexpr int s = decltype(sc)::value;
    if (s >= smax) return false;
#pragma unroll
    for (int k = 0; k < D; ++k) mn[k] = hk_fmin(mn[k], q[s * D + k]);
    return true;
  });
  float sub[D];
#pragma unroll
  for (int k = 0; k < D; ++k) {
    mn[k] = quad_min(mn[k]);
    sub[k] = (mn[k] < INFINITY && (!jax_sem || mn[k] > 0.0f)) ? mn[k] : 0.0f;
  }
  unrolled_while<0, C>([&](auto sc) {
    constexpr int s = decltype(sc)::value;
    if (s >= smax) return false;
#pragma unroll
    for (int k = 0; k < D; ++k) q[s * D + k] = q[s * D + k] - sub[k];  // inf - sub = inf: holes stay
    return true;
  });
}

// _jax_ops.py:93-111 / _torch_ops.py:136-146
template <int D, int C>
__device__ __forceinline__ void team_rescale(float (&q)[C * D], int smax, unsigned flags) {
  const bool jax_sem = (flags & HK_SEM_MASK) == HK_SEM_JAX;
  float mx = -1.0f;
  unrolled_while<0, C>([&](auto sc) {
    constexpr int s = decltype(sc)::value;
    if (s >= smax) return false;
    const bool live = q[s * D] < INFINITY;
#pragma unroll
    for (int k = 0; k < D; ++k) mx = hk_fmax(mx, live ? q[s * D + k] : -1.0f);
    return true;
  });
  mx = quad_max(mx);
  const bool skip = jax_sem ? (mx <= 1e-8f) : (mx < 0.0f);
  const float div = (skip || mx == 0.0f) ? 1.0f : mx;
  unrolled_while<0, C>([&](auto sc) {
    constexpr int s = decltype(sc)::value;
    if (s >= smax) return false;
    const bool live = q[s * D] < INFINITY;
#pragma unroll
    for (int k = 0; k < D; ++k) q[s * D + k] = live ? q[s * D + k] / div : INFINITY;
    return true;
  });
}

// The pair loop, each unordered pair ONCE.  Rows j = 4(c-1) .. 4c-1 of the game (one broadcast read per j)
// are tested against the lane's slots 0 .. c-1, i.e. against rows i < j only (in the last slot, c-1, the
// lanes whose row is not below j sit the test out).  One set of differences serves both directions:
//     t = max_k(P_j - P_i), u = min_k(P_j - P_i)
//     j is removed by i  iff  u >= 0           (P_i <= P_j; equal rows: the later one goes)
//     i is removed by j  iff  t <= 0 and u < 0 (P_j <= P_i and not equal)
// The verdicts on the lane's own rows i go to acc[s]; those on row j -- which another lane of the team
// owns -- are collected as bits of jmask and OR-ed over the team afterwards.  Holes are +inf: a hole j
// gets its bit set (harmless), a hole i never sets one (u = -inf).  The segment for c slots is
// straight-line in the slots; segments are entered while rows remain (scalar test once per four rows).
template <int D, int C, int CSEG>
__device__ __forceinline__ void team_pair_segment(const float (&q)[C * D], float (&acc)[C], uint32_t (&jmask)[2],
                                                  const float* mine, int tl, int nmax) {
  constexpr int j0 = kTeam * (CSEG - 1);
  const int j1 = nmax < kTeam * CSEG ? nmax : kTeam * CSEG;
  for (int j = j0; j < j1; ++j) {
    float pj[D];
    row_load<D>(mine + j * D, pj);
    bool jdead = false;
#pragma unroll
    for (int s = 0; s < CSEG; ++s) {
      float t = pj[0] - q[s * D], u = t;
#pragma unroll
      for (int k = 1; k < D; ++k) {
        const float dk = pj[k] - q[s * D + k];
        t = hk_fmax(t, dk);
        u = hk_fmin(u, dk);
      }
      const bool below = (s < CSEG - 1) || (j0 + tl < j);  // this lane's row i = 4s + tl is below j
      jdead |= below && (u >= 0.0f);
      acc[s] = hk_fmin(acc[s], (below && u < 0.0f) ? t : 1.0f);
    }
    const uint32_t bit = jdead ? (1u << (j & 31)) : 0u;
    if (j < 32) jmask[0] |= bit;
    else jmask[1] |= bit;
  }
}

template <int D, int C, int CSEG>
struct TeamPairs {
  static __device__ __forceinline__ void run(const float (&q)[C * D], float (&acc)[C], uint32_t (&jmask)[2],
                                             const float* mine, int tl, int nmax) {
    if (nmax <= kTeam * (CSEG - 1)) return;
    team_pair_segment<D, C, CSEG>(q, acc, jmask, mine, tl, nmax);
    if constexpr (CSEG < C) TeamPairs<D, C, CSEG + 1>::run(q, acc, jmask, mine, tl, nmax);
  }
};

// _jax_ops.py:15-73.  The region must hold the mirror of the team's rows.  Removed rows become holes;
// returns the team-uniform bitmask of the original slots that were removed.
template <int D, int C>
__device__ __forceinline__ Mask64 team_newton(float (&q)[C * D], const uint32_t (&orig)[C / 4], const float* mine, int tl,
                                              int nmax, int smax) {
  float acc[C];
#pragma unroll
  for (int s = 0; s < C; ++s) acc[s] = INFINITY;
  uint32_t jmask[2] = {0u, 0u};
  TeamPairs<D, C, 1>::run(q, acc, jmask, mine, tl, nmax);
  jmask[0] = quad_or(jmask[0]);
  jmask[1] = quad_or(jmask[1]);
  const Mask64 jm = ((Mask64)jmask[1] << 32) | jmask[0];
  Mask64 dead = 0;
  unrolled_while<0, C>([&](auto sc) {
    constexpr int s = decltype(sc)::value;
    if (s >= smax) return false;
    const bool by_lower = (jm >> (kTeam * s + tl)) & 1ull;
    const bool removed = ((acc[s] <= 0.0f) || by_lower) && (q[s * D] < INFINITY);
    dead |= removed ? ((Mask64)1 << orig_slot<s>(orig)) : (Mask64)0;
#pragma unroll
    for (int k = 0; k < D; ++k) q[s * D + k] = removed ? INFINITY : q[s * D + k];
    return true;
  });
  return quad_or64(dead);
}

// pad everywhere, then the live rows at their original slots
template <int D, int C>
__device__ __forceinline__ void team_publish(const float (&q)[C * D], const uint32_t (&orig)[C / 4], float* mine, int m,
                                             float pad, int tl, int smax, bool active) {
  float pv[D];
#pragma unroll
  for (int k = 0; k < D; ++k) pv[k] = pad;
  if (active)
    for (int i = tl; i < m; i += kTeam) row_store<D>(mine + i * D, pv);
  unrolled_while<0, C>([&](auto sc) {
    constexpr int s = decltype(sc)::value;
    if (s >= smax) return false;
    if (q[s * D] < INFINITY) {
      float v[D];
#pragma unroll
      for (int k = 0; k < D; ++k) v[k] = q[s * D + k];
      row_store<D>(mine + orig_slot<s>(orig) * D, v);
    }
    return true;
  });
}

// register budget: three waves per SIMD up to dim 4 (<= 168 VGPRs), two beyond
// Zeillinger's host for a team (jax/players.py:55-109; see c_zeillinger in hk_fast_rows.h): every lane scans
// the pairs (its rows i, rows j > i from the mirror), keeps its best (L, S, position) -- the position
// i*64 + j makes "first minimum in row-major order" independent of the order in which the lanes visit the
// pairs -- and two DPP exchanges leave the team's best pair, with its difference vector, in every lane.
// The region must hold the mirror of the team's rows.
// BOTH: jax or list variant by the runtime flag (see c_zeillinger)
template <int D, int C, bool BOTH = false>
__device__ __forceinline__ int team_zeillinger(const float (&q)[C * D], const float* mine, int tl, int nmax, int smax,
                                               bool list_rt = false) {
  const bool LIST = BOTH && list_rt;
  float bestL = INFINITY, bestS = INFINITY;
  int bestP = 0x7FFFFFFF;
  float bd[D];
#pragma unroll
  for (int k = 0; k < D; ++k) bd[k] = 0.0f;
  for (int j = 1; j < nmax; ++j) {
    float pj[D];
    row_load<D>(mine + j * D, pj);
    const bool live_j = pj[0] < INFINITY;
    unrolled_while<0, C>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
      if (s >= smax || kTeam * s >= j) return false;  // slots whose rows are all >= j: nothing below j left
      const int i = kTeam * s + tl;
      float v[D];
#pragma unroll
      for (int k = 0; k < D; ++k) v[k] = q[s * D + k] - pj[k];
      float mx = v[0], mn = v[0];
#pragma unroll
      for (int k = 1; k < D; ++k) {
        mx = (v[k] > mx) ? v[k] : mx;
        mn = (v[k] < mn) ? v[k] : mn;
      }
      const bool close = fabsf(mx - mn) <= 1e-8f + 1e-5f * fabsf(mn);  // jnp.isclose
      float cnt = 0.0f;
#pragma unroll
      for (int k = 0; k < D; ++k) cnt += (float)((v[k] == mx) + (v[k] == mn));
      const float L = mx - mn;
      const int pos = i * 64 + j;
      const bool valid = (i < j) && live_j && (q[s * D] < INFINITY) && (LIST || !close);
      const bool better = valid && (L < bestL || (L == bestL && (cnt < bestS || (cnt == bestS && pos < bestP))));
      bestL = better ? L : bestL;
      bestS = better ? cnt : bestS;
      bestP = better ? pos : bestP;
#pragma unroll
      for (int k = 0; k < D; ++k) bd[k] = better ? v[k] : bd[k];
      return true;
    });
  }
  // the team's best: exchange with the lane 1 away, then 2 away
#define HK_TEAM_MERGE(CTRL)                                                                              \
  {                                                                                                      \
    const float oL = __int_as_float(quad_perm_i<CTRL>(__float_as_int(bestL)));                           \
    const float oS = __int_as_float(quad_perm_i<CTRL>(__float_as_int(bestS)));                           \
    const int oP = quad_perm_i<CTRL>(bestP);                                                             \
    const bool take = oL < bestL || (oL == bestL && (oS < bestS || (oS == bestS && oP < bestP)));        \
    _Pragma("unroll") for (int k = 0; k < D; ++k) {                                                      \
      const float ov = __int_as_float(quad_perm_i<CTRL>(__float_as_int(bd[k])));                         \
      bd[k] = take ? ov : bd[k];                                                                         \
    }                                                                                                    \
    bestL = take ? oL : bestL;                                                                           \
    bestS = take ? oS : bestS;                                                                           \
    bestP = take ? oP : bestP;                                                                           \
  }
  HK_TEAM_MERGE(kQuadXor1)
  HK_TEAM_MERGE(kQuadXor2)
#undef HK_TEAM_MERGE
  int lo = 0, hi = 0;
  float vlo = bd[0], vhi = bd[0];
#pragma unroll
  for (int k = 1; k < D; ++k) {
    if (bd[k] < vlo) { vlo = bd[k]; lo = k; }
    if (bd[k] > vhi) { vhi = bd[k]; hi = k; }
  }
  if (LIST) {
    if (bestP == 0x7FFFFFFF) return -1;
    return (lo == hi) ? encode_mask(3u) : encode_mask((1u << lo) | (1u << hi));
  }
  if (bestP == 0x7FFFFFFF || lo == hi) return 0;
  return encode_mask((1u << lo) | (1u << hi));
}

// rank of each of the lane's rows among the team's live rows in descending key order (KEY: a KeyOrder; among
// equal keys the lower index first).  The region must hold the mirror of the rows.
template <int D, int C, int KEY>
__device__ __forceinline__ void team_ranks(const float (&q)[C * D], const float* mine, int tl, int nmax, int smax,
                                           int (&rank)[C]) {
#pragma unroll
  for (int s = 0; s < C; ++s) rank[s] = 0;
  for (int j = 0; j < nmax; ++j) {
    float pj[D];
    row_load<D>(mine + j * D, pj);
    const bool live_j = pj[0] < INFINITY;
    unrolled_while<0, C>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
      if (s >= smax) return false;
      const bool j_greater = key_gt<D, KEY>(pj, &q[s * D]);
      const bool i_greater = key_gt<D, KEY>(&q[s * D], pj);
      const bool before = live_j && (j_greater || (!i_greater && j < kTeam * s + tl));
      rank[s] += before ? 1 : 0;
      return true;
    });
  }
}

// the lane's live rows to their rank in the region
template <int D, int C>
__device__ __forceinline__ void team_scatter_ranked(const float (&q)[C * D], float* mine, const int (&rank)[C],
                                                    int smax) {
  unrolled_while<0, C>([&](auto sc) {
    constexpr int s = decltype(sc)::value;
    if (s >= smax) return false;
    if (q[s * D] < INFINITY) {
      float v[D];
#pragma unroll
      for (int k = 0; k < D; ++k) v[k] = q[s * D + k];
      row_store<D>(mine + rank[s] * D, v);
    }
    return true;
  });
}

// observation features (jax/util.py:186-197; coord0: core/tensor_points.py:72-74): pad everywhere, the live
// rows at their rank
template <int D, int C>
__device__ __forceinline__ void team_publish_ranked(const float (&q)[C * D], float* mine, int m, float pad, int tl,
                                                    int nmax, int smax, bool active, bool coord0) {
  int rank[C];
  if (coord0) team_ranks<D, C, kKeyCoord0>(q, mine, tl, nmax, smax, rank);  // (a rolled loop over the mirror: two
  else team_ranks<D, C, kKeyLast>(q, mine, tl, nmax, smax, rank);           //  copies are cheap here)
  __syncthreads();
  float pv[D];
#pragma unroll
  for (int k = 0; k < D; ++k) pv[k] = pad;
  if (active)
    for (int i = tl; i < m; i += kTeam) row_store<D>(mine + i * D, pv);
  team_scatter_ranked<D, C>(q, mine, rank, smax);
}

// ZEIL: the rollout variant whose host is Zeillinger's (its pair scan would otherwise sit in every rollout's
// register budget)
template <int D, int MODE, bool ZEIL = false>
__global__ __launch_bounds__(kWave, (D <= 4 ? 3 : 2)) void team_kernel(const Params prm) {
  extern __shared__ __align__(16) unsigned char hk_smem[];
  float* lds = reinterpret_cast<float*>(hk_smem);
  constexpr int C = kTeamSlots;
  constexpr bool kStep = MODE == kModeStep || MODE == kModeStepAux;  // Aux: features / Zeillinger's class
  const int lane = threadIdx.x;
  const int tl = lane & (kTeam - 1), tg = lane >> 2;
  const int m = prm.m, n_el = m * D, S = prm.lds_stride;
  float* cbuf = lds + kTeamGames * S;  // slow path only: D floats per game
  const int64_t g0 = (int64_t)blockIdx.x * kTeamGames;
  const int64_t left = (int64_t)prm.batch - g0;
  const int ngames = (int)(left < kTeamGames ? left : kTeamGames);
  const bool active = tg < ngames;
  const bool leader = active && tl == 0;
  const int64_t g = g0 + tg;
  const uint64_t gg =
      prm.game_offset + ((prm.mode == kModeRollout && prm.game_ids && active) ? (uint64_t)(uint32_t)prm.game_ids[g] : (uint64_t)g);
  float* mine = lds + tg * S;
  const float pad = (float)prm.pad;
  const unsigned flags = prm.flags;
  const unsigned stages = (MODE == kModeGenerate) ? (prm.stages & ~HK_STAGE_SHIFT) : prm.stages;
  const float fill = ((flags & HK_SEM_MASK) == HK_SEM_JAX) ? -1.0f : pad;  // _jax_ops.py:65 does not forward pad
  const int nsteps = (MODE == kModeRollout) ? prm.steps : 1;
  const bool vec_in = (n_el % 4 == 0) && (prm.in_stride % 4 == 0) && prm.in &&
                      (reinterpret_cast<uintptr_t>(prm.in) % 16 == 0);
  const bool vec_out = (n_el % 4 == 0) && (prm.out_stride % 4 == 0) &&
                       (reinterpret_cast<uintptr_t>(prm.out) % 16 == 0);
  const bool vec_obs = (n_el % 4 == 0) && (reinterpret_cast<uintptr_t>(prm.obs_out) % 16 == 0);
  PolicyCache pcache;

  float c[D];
  int axis_in = -1;
#pragma unroll
  for (int k = 0; k < D; ++k) c[k] = 0.0f;
  RawActions<D> raw;
  const bool fetch_actions = kStep && (stages & HK_STAGE_SHIFT) && active;
  if (fetch_actions) fast_fetch_actions<D>(prm, g, m, raw);  // converted after the slab is requested

  // ---- 1. the image --------------------------------------------------------------------------------
  if (MODE == kModeGenerate) {
    if (active) {
      // (hk_common.h: eight elements per Philox block for small max_value, four otherwise; block b -> lane b % 4)
      const bool sh = gen_short((uint32_t)prm.max_value);
      const int per = sh ? 8 : 4;
      for (int e = tl * per; e < n_el; e += per * kTeam) {
        const U4 r = philox4x32((uint32_t)gg, (uint32_t)(gg >> 32), (uint32_t)(e / per), kStreamGenerate, prm.seed);
        uint32_t v[8];
        gen_block_values(r, (uint32_t)prm.max_value, sh, v);
        for (int qd = 0; qd < per && e + qd < n_el; ++qd) mine[e + qd] = (float)v[qd];
      }
    }
  } else {
    rows_copy_slab<true>(lds, const_cast<float*>((const float*)prm.in), prm.in_stride, n_el, S, g0, ngames, lane,
                        vec_in);
  }
  if (fetch_actions) fast_decode_actions<D>(prm, raw, c, axis_in);
  __syncthreads();

  // ---- 2. live rows, exactness guard (each lane scans every fourth row of its game) -----------------
  Mask64 part = 0;
  bool ok = true;
  if (active)
    for (int i = tl; i < m; i += kTeam) {
      float v[D];
      row_load<D>(mine + i * D, v);
      bool ge = true, fl = true;
#pragma unroll
      for (int k = 0; k < D; ++k) {
        ge &= (__float_as_uint(v[k]) < 0x7F800000u);  // [+0, +inf)
        fl &= (v[k] == fill);
      }
      ok &= (ge | fl);
      part |= ge ? ((Mask64)1 << i) : (Mask64)0;
    }
  Mask64 gmask = quad_or64(part);
  ok = quad_or(ok ? 0u : 1u) == 0u;
  int np = __popcll(gmask);
  int nmax = wave_max(np, m);
  int smax = (nmax + kTeam - 1) / kTeam;
  const bool exact = (fill == pad) && __all(ok);

  if (!exact) {
    // ---- slow path (whole wave): the exact generic routines on the image, one lane per game ------
    float* cs = cbuf + tg * D;
    if (MODE == kModeStepAux && prm.class_out) {  // hk_zeillinger: the class is the only output
      if (leader)
        prm.class_out[g] = ((flags & HK_SEM_MASK) == HK_SEM_LIST) ? zeillinger_list_game<float>(mine, m, prm.d)
                                                                  : zeillinger_game<float>(mine, m, prm.d);
      return;
    }
    np = leader ? num_points<float>(mine, m, D) : 2;
    int length = (np < 2) ? 0 : -1;
    if (MODE == kModeRollout && prm.count_ws) {
      const unsigned long long b0 = __ballot(leader && np < 2);
      if (lane == 0) count_add(prm.count_ws + blockIdx.x, (uint32_t)__popcll(b0));
    }
    for (int t = 0; t < nsteps; ++t) {
      int axis = -1, cls = 0;
      if (MODE == kModeRollout) {
        if (prm.obs_out) {
          __syncthreads();
          rows_copy_slab<false>(lds, (float*)prm.obs_out + (int64_t)t * prm.batch * n_el, (int64_t)n_el, n_el, S,
                               g0, ngames, lane, vec_obs);
          __syncthreads();
        }
        uint32_t mask;
        const int zc = (ZEIL && prm.host_policy == HK_HOST_ZEILLINGER && leader) ? zeillinger_game<float>(mine, m, prm.d) : 0;
        fast_policy<D>(prm, gg, prm.step_offset + (uint32_t)t, pcache, cls, axis, mask, zc);
        if (leader)
          for (int k = 0; k < D; ++k) cs[k] = (float)((mask >> k) & 1u);
      } else if (kStep && (stages & HK_STAGE_SHIFT) && leader) {
        load_coords<float>(prm, g, cs);
        axis = axis_in;
      }
      const bool prev_done = np < 2;
      if (leader) stages_game<float>(mine, m, prm.d, cs, axis, pad, stages, flags);
      np = leader ? num_points<float>(mine, m, prm.d) : 2;
      const bool done = np < 2;
      if (done && length < 0) length = t + 1;
      if (MODE == kModeRollout) {
        if (leader) {
          const int64_t at = (int64_t)t * prm.batch + g;
          if (prm.r_host_class_out) prm.r_host_class_out[at] = cls;
          if (prm.r_axis_out) prm.r_axis_out[at] = axis;
          if (prm.r_done_out) prm.r_done_out[at] = done;
          if (prm.r_reward_out) prm.r_reward_out[at] = prm.reward_sign * (float)(done && !prev_done);
        }
        if (prm.count_ws) {
          const unsigned long long bd = __ballot(leader && done);
          if (lane == 0) count_add(prm.count_ws + (size_t)(t + 1) * prm.count_stride + blockIdx.x, (uint32_t)__popcll(bd));
        }
      } else if (kStep && leader) {
        if (prm.done_out) prm.done_out[g] = done;
        if (prm.prev_done_out) prm.prev_done_out[g] = prev_done;
        if (prm.reward_out) prm.reward_out[g] = prm.reward_sign * (float)(done && !prev_done);
        if (prm.num_points_out) prm.num_points_out[g] = np;
      }
    }
    if (MODE == kModeRollout && leader && prm.game_length_out) prm.game_length_out[g] = length;
    __syncthreads();
    rows_copy_slab<false>(lds, (float*)prm.out, prm.out_stride, n_el, S, g0, ngames, lane, vec_out);
    return;
  }

  // ---- 3. the live rows into registers ---------------------------------------------------------------
  float q[C * D];
  uint32_t orig[C / 4];  // original slot of each register row, one byte each
#pragma unroll
  for (int e = 0; e < C * D; ++e) q[e] = INFINITY;  // slots past smax are holes in the pair loop
  if (!active) {  // teams past the batch: an empty, harmless game
    gmask = 0;
    np = 2;  // never "done", never counted
  }
  team_gather<D, C>(q, orig, mine, gmask, tl, smax, false);
  if (MODE == kModeStepAux && prm.class_out) {  // hk_zeillinger: the class is the only output
    __syncthreads();
    team_mirror<D, C>(q, mine, tl, smax);
    __syncthreads();
    const int zc = team_zeillinger<D, C, true>(q, mine, tl, nmax, smax, (flags & HK_SEM_MASK) == HK_SEM_LIST);
    if (leader) prm.class_out[g] = zc;
    return;
  }
  int length = (np < 2) ? 0 : -1;
  if (MODE == kModeRollout && prm.count_ws) {
    const unsigned long long b0 = __ballot(leader && np < 2);
    if (lane == 0) count_add(prm.count_ws + blockIdx.x, (uint32_t)__popcll(b0));
  }

  // ---- 4. the transitions --------------------------------------------------------------------------
  for (int t = 0; t < nsteps; ++t) {
    int axis = -1, cls = 0;
    if (MODE == kModeRollout) {
      if (prm.obs_out) {  // state before the step: rebuild the image, store it coalesced
        __syncthreads();
        team_publish<D, C>(q, orig, mine, m, pad, tl, smax, active);
        __syncthreads();
        rows_copy_slab<false, kCopyBatchInLoop>(lds, (float*)prm.obs_out + (int64_t)t * prm.batch * n_el,
                                               (int64_t)n_el, n_el, S, g0, ngames, lane, vec_obs);
        __syncthreads();
      }
      uint32_t mask;
      int zc = 0;
      if (ZEIL && prm.host_policy == HK_HOST_ZEILLINGER) {
        __syncthreads();
        team_mirror<D, C>(q, mine, tl, smax);
        __syncthreads();
        zc = team_zeillinger<D, C>(q, mine, tl, nmax, smax);
      }
      fast_policy<D>(prm, gg, prm.step_offset + (uint32_t)t, pcache, cls, axis, mask, zc);
#pragma unroll
      for (int k = 0; k < D; ++k) c[k] = (float)((mask >> k) & 1u);
    } else if (kStep) {
      axis = axis_in;
    }
    const bool prev_done = np < 2;

    if (stages & HK_STAGE_SHIFT) c_shift<C, D>(q, smax, c, axis, np, flags);
    if (stages & HK_STAGE_REPOSITION) team_reposition<D, C>(q, smax, flags);
    if (stages & HK_STAGE_NEWTON) {
      team_mirror<D, C>(q, mine, tl, smax);
      __syncthreads();
      const Mask64 dead = team_newton<D, C>(q, orig, mine, tl, nmax, smax);
      gmask &= ~dead;
      if (active) np = __popcll(gmask);
      if (MODE == kModeStepAux && ((flags & HK_SEM_MASK) == HK_SEM_LIST || (flags & HK_FLAG_COMPACT_SORTED))) {
        // list semantics: right after the Newton stage (before a rescale could round two keys together) the
        // survivors are sorted descending-lexicographically and packed to the front -- physically: rows to
        // their rank in the region, slots 0..n-1 become the game's live slots, registers re-gathered
        __syncthreads();
        team_mirror<D, C>(q, mine, tl, smax);
        __syncthreads();
        int rank[C];
        team_ranks<D, C, kKeyFirst>(q, mine, tl, nmax, smax, rank);
        __syncthreads();
        team_scatter_ranked<D, C>(q, mine, rank, smax);
        __syncthreads();
        const int live = active ? np : 0;
        gmask = live >= 64 ? ~(Mask64)0 : (((Mask64)1 << live) - 1);
        const int sprev = smax;
        nmax = wave_max(live, nmax);
        smax = (nmax + kTeam - 1) / kTeam;
        team_gather<D, C>(q, orig, mine, gmask, tl, sprev, true);
      }
    }
    if (stages & HK_STAGE_RESCALE) team_rescale<D, C>(q, smax, flags);

    const bool done = np < 2;
    if (done && length < 0) length = t + 1;
    if (MODE == kModeRollout) {
      if (leader) {
        const int64_t at = (int64_t)t * prm.batch + g;
        if (prm.r_host_class_out) prm.r_host_class_out[at] = cls;
        if (prm.r_axis_out) prm.r_axis_out[at] = axis;
        if (prm.r_done_out) prm.r_done_out[at] = done;
        if (prm.r_reward_out) prm.r_reward_out[at] = prm.reward_sign * (float)(done && !prev_done);
      }
      if (prm.count_ws) {
        const unsigned long long bd = __ballot(leader && done);
        if (lane == 0) count_add(prm.count_ws + (size_t)(t + 1) * prm.count_stride + blockIdx.x, (uint32_t)__popcll(bd));
      }
      // squeeze when the widest game of the wave got narrower: live rows to their new ranks in the
      // region, then every lane takes back rows 4s+tl
      if (t + 1 < nsteps && !__any(active && np >= nmax)) {
        __syncthreads();
        unrolled_while<0, C>([&](auto sc) {
          constexpr int s = decltype(sc)::value;
          if (s >= smax) return false;
          if (q[s * D] < INFINITY) {
            const int rank = __popcll(gmask & (((Mask64)1 << orig_slot<s>(orig)) - 1));
            float v[D];
#pragma unroll
            for (int k = 0; k < D; ++k) v[k] = q[s * D + k];
            row_store<D>(mine + rank * D, v);
          }
          return true;
        });
        __syncthreads();
        nmax = wave_max(active ? np : 0, nmax - 1);
        const int sprev = smax;
        smax = (nmax + kTeam - 1) / kTeam;
        team_gather<D, C>(q, orig, mine, gmask, tl, sprev, true);
      }
    } else if (kStep && leader) {
      if (prm.done_out) prm.done_out[g] = done;
      if (prm.prev_done_out) prm.prev_done_out[g] = prev_done;
      if (prm.reward_out) prm.reward_out[g] = prm.reward_sign * (float)(done && !prev_done);
      if (prm.num_points_out) prm.num_points_out[g] = np;
    }
  }
  if (MODE == kModeRollout && leader && prm.game_length_out) prm.game_length_out[g] = length;

  // ---- 5. publish ----------------------------------------------------------------------------------
  __syncthreads();
  if (MODE == kModeStepAux && (stages & kStageFeatureSorts)) {
    team_mirror<D, C>(q, mine, tl, smax);
    __syncthreads();
    team_publish_ranked<D, C>(q, mine, m, pad, tl, nmax, smax, active, (stages & kStageFeatureSort0) != 0);
  } else {
    team_publish<D, C>(q, orig, mine, m, pad, tl, smax, active);
  }
  __syncthreads();
  rows_copy_slab<false>(lds, (float*)prm.out, prm.out_stride, n_el, S, g0, ngames, lane, vec_out);
}

// ---- host side ---------------------------------------------------------------------------------------
// HK_FLAG_FORCE_TEAM puts the shapes that have a register-resident specialisation on this kernel too
inline bool team_supported(const Params& prm, int dtype) {
  if (dtype != HK_F32) return false;
  // sorted + compacted output (list semantics): single steps only
  if ((prm.stages & HK_STAGE_NEWTON) && prm.mode != kModeStep &&
      ((prm.flags & HK_SEM_MASK) == HK_SEM_LIST || (prm.flags & HK_FLAG_COMPACT_SORTED)))
    return false;
  if (prm.flags & HK_FLAG_FORCE_GENERIC) return false;
  if ((prm.stages & kStageFeatureSorts) && prm.mode != kModeStep) return false;
  if (prm.mode == kModeZeillinger) return false;
  if (prm.d < 2 || prm.d > 6 || prm.m > kTeam * kTeamSlots) return false;
  return true;
}

// LDS geometry: 16 regions of `stride` floats (>= the rows rounded up to a multiple of 4, stride / row
// width odd so that the 16 teams' broadcast reads fall on different banks) + D floats per game of
// slow-path scratch
inline int plan_team(Params& prm) {
  const int w = prm.d == 4 ? 4 : (prm.d == 2 ? 2 : 1);
  int stride = ((prm.m + kTeam - 1) / kTeam) * kTeam * prm.d;
  stride = (stride + w - 1) / w * w;
  if (((stride / w) & 1) == 0) stride += w;
  if ((int64_t)(stride + prm.d) * 4 * kTeamGames > kMaxLdsBytes) return HK_ERR_UNSUPPORTED;
  prm.lds_stride = stride;
  prm.games_per_block = kTeamGames;
  return HK_OK;
}

template <int D, int MODE, bool ZEIL = false>
int launch_team_t(const Params& prm, hipStream_t stream) {
  const size_t lds = (size_t)(prm.lds_stride + prm.d) * kTeamGames * sizeof(float);
  const unsigned grid = (unsigned)(((int64_t)prm.batch + kTeamGames - 1) / kTeamGames);
  launch_prepare();
  hipLaunchKernelGGL((team_kernel<D, MODE, ZEIL>), dim3(grid), dim3(kWave), lds, stream, prm);
  return launch_status();
}

template <int D>
int launch_team_d(const Params& prm, hipStream_t stream) {
  const bool sorted_out = (prm.stages & HK_STAGE_NEWTON) &&
                          ((prm.flags & HK_SEM_MASK) == HK_SEM_LIST || (prm.flags & HK_FLAG_COMPACT_SORTED));
  if (prm.mode == kModeStep && (prm.class_out || (prm.stages & kStageFeatureSorts) || sorted_out))
    return launch_team_t<D, kModeStepAux>(prm, stream);
  if (prm.mode == kModeStep) return launch_team_t<D, kModeStep>(prm, stream);
  if (prm.mode == kModeRollout && prm.host_policy == HK_HOST_ZEILLINGER)
    return launch_team_t<D, kModeRollout, true>(prm, stream);
  if (prm.mode == kModeRollout) return launch_team_t<D, kModeRollout>(prm, stream);
  return launch_team_t<D, kModeGenerate>(prm, stream);
}

// one translation unit per dim (hk_team_spec.hip), as for the register-resident specialisations
#ifndef HK_SPEC_TU
extern template int launch_team_d<2>(const Params&, hipStream_t);
extern template int launch_team_d<3>(const Params&, hipStream_t);
extern template int launch_team_d<4>(const Params&, hipStream_t);
extern template int launch_team_d<5>(const Params&, hipStream_t);
extern template int launch_team_d<6>(const Params&, hipStream_t);

inline int launch_team(Params& prm, hipStream_t stream) {
  const int st = plan_team(prm);
  if (st != HK_OK) return st;
  switch (prm.d) {
    case 2: return launch_team_d<2>(prm, stream);
    case 3: return launch_team_d<3>(prm, stream);
    case 4: return launch_team_d<4>(prm, stream);
    case 5: return launch_team_d<5>(prm, stream);
    case 6: return launch_team_d<6>(prm, stream);
  }
  return HK_ERR_UNSUPPORTED;
}
#endif

}  // namespace hk
