// The stages of the Hironaka step on the live rows a lane holds in registers (see hk_fast_kernel.h).
//
// q[C*D]: up to C rows with COMPILE-TIME indices; rows [n, nmax) of a lane are +inf holes; nmax is
// wave-uniform.  Every row loop must (a) be fully unrolled -- a rolled loop would index q dynamically
// and push it into scratch memory -- and (b) leave early on `r >= nmax` with a scalar branch.  LLVM only
// unrolls such "upper-bound" loops up to 8 iterations by default, so the unrolling is done by
// construction: `unrolled_while<0, C>(f)` calls f(integral_constant<r>) for r = 0, 1, ... and stops at
// the first `false`, as nested ifs the optimiser cannot re-roll.
#pragma once

#include <type_traits>

#include "hk_common.h"

namespace hk {

template <int I, int N>
struct UnrolledWhile {
  template <typename F>
  static __device__ __forceinline__ void run(F&& f) {
    if (f(std::integral_constant<int, I>{})) UnrolledWhile<I + 1, N>::run(f);
  }
};
template <int N>
struct UnrolledWhile<N, N> {
  template <typename F>
  static __device__ __forceinline__ void run(F&&) {}
};
template <int I0, int N, typename F>
__device__ __forceinline__ void unrolled_while(F&& f) {
  UnrolledWhile<(I0 < N ? I0 : N), N>::run(f);
}

template <int M>
using MaskT = typename std::conditional<(M <= 32), uint32_t, unsigned long long>::type;

__device__ inline int mask_pop(uint32_t m) { return __popc(m); }
__device__ inline int mask_pop(unsigned long long m) { return __popcll(m); }
__device__ inline int mask_first(uint32_t m) { return __ffs(m) - 1; }
__device__ inline int mask_first(unsigned long long m) { return __ffsll(m) - 1; }

// rows of the set bits of `mask`, ascending, into q[0..n); q[n..nmax) := +inf
// (SB: compile-time bound of the rows touched -- the staircase of rollout loops knows it per level)
template <int M, int C, int D, int SB = C>
__device__ __forceinline__ void gather_rows(float (&q)[C * D], const float* mine, MaskT<M> mask, int nmax) {
  unrolled_while<0, SB>([&](auto rc) {
    constexpr int r = decltype(rc)::value;
    if (r >= nmax) return false;
    const bool has = mask != 0;
    const int s = has ? mask_first(mask) : 0;
    mask &= mask - 1;
    const float* row = mine + s * D;
#pragma unroll
    for (int k = 0; k < D; ++k) {
      const float v = row[k];
      q[r * D + k] = has ? v : INFINITY;
    }
    return true;
  });
}

// live rows of q back to their slots; returns the mask of the slots still alive
template <int M, int C, int D, int SB = C>
__device__ __forceinline__ MaskT<M> scatter_rows(const float (&q)[C * D], float* mine, MaskT<M> mask, int nmax) {
  MaskT<M> alive = 0;
  unrolled_while<0, SB>([&](auto rc) {
    constexpr int r = decltype(rc)::value;
    if (r >= nmax) return false;
    const bool has = mask != 0;
    const int s = has ? mask_first(mask) : 0;
    mask &= mask - 1;
    if (has && q[r * D] < INFINITY) {
      alive |= (MaskT<M>)1 << s;
      float* row = mine + s * D;
#pragma unroll
      for (int k = 0; k < D; ++k) row[k] = q[r * D + k];
    }
    return true;
  });
  return alive;
}

template <int C, int D>
__device__ __forceinline__ int count_live(const float (&q)[C * D], int nmax) {
  int n = 0;
  unrolled_while<0, C>([&](auto rc) {
    constexpr int r = decltype(rc)::value;
    if (r >= nmax) return false;
    n += (q[r * D] < INFINITY) ? 1 : 0;
    return true;
  });
  return n;
}

// _jax_ops.py:76-90 / _torch_ops.py:46-110
template <int C, int D>
__device__ __forceinline__ void c_shift(float (&q)[C * D], int nmax, const float (&c)[D], int axis, int np,
                                        unsigned flags) {
  bool apply = true;
  if (flags & HK_FLAG_AXIS_NOOP_IF_INVALID) {
#pragma unroll
    for (int k = 0; k < D; ++k) {
      const float onehot = (k == axis) ? 1.0f : 0.0f;
      if (!(onehot - c[k] <= 0.0f)) apply = false;
    }
  }
  if ((flags & HK_FLAG_IGNORE_ENDED) && np < 2) apply = false;
  bool isax[D];
#pragma unroll
  for (int k = 0; k < D; ++k) isax[k] = apply && (k == axis);
  unrolled_while<0, C>([&](auto rc) {
    constexpr int r = decltype(rc)::value;
    if (r >= nmax) return false;
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < D; ++k) s = s + q[r * D + k] * c[k];  // order 0..D-1, no contraction
    const bool live = q[r * D] < INFINITY;                   // holes would give inf*0 = NaN
#pragma unroll
    for (int k = 0; k < D; ++k) q[r * D + k] = (live && isax[k]) ? s : q[r * D + k];
    return true;
  });
}

// _jax_ops.py:114-123 / _torch_ops.py:113-133 (padding rows are not in q)
template <int C, int D>
__device__ __forceinline__ void c_reposition(float (&q)[C * D], int nmax, unsigned flags) {
  const bool jax_sem = (flags & HK_SEM_MASK) == HK_SEM_JAX;
  float mn[D];
#pragma unroll
  for (int k = 0; k < D; ++k) mn[k] = INFINITY;
  unrolled_while<0, C>([&](auto rc) {
    constexpr int r = decltype(rc)::value;
    if (r >= nmax) return false;
#pragma unroll
    for (int k = 0; k < D; ++k) mn[k] = hk_fmin(mn[k], q[r * D + k]);
    return true;
  });
  float sub[D];
#pragma unroll
  for (int k = 0; k < D; ++k)  // JAX leaves a column whose minimum is <= 0 untouched: subtract 0
    sub[k] = (mn[k] < INFINITY && (!jax_sem || mn[k] > 0.0f)) ? mn[k] : 0.0f;
  unrolled_while<0, C>([&](auto rc) {
    constexpr int r = decltype(rc)::value;
    if (r >= nmax) return false;
#pragma unroll
    for (int k = 0; k < D; ++k) q[r * D + k] = q[r * D + k] - sub[k];  // inf - sub = inf: holes stay
    return true;
  });
}

// t = max_k(a - b), u = min_k(a - b).  The differences are results of a subtraction (canonical), so plain fmaxf /
// fminf chains become v_max3_f32 / v_min3_f32 (one instruction per three values) without the canonicalising
// v_max x, x that values of unknown origin get; hk_fmax / hk_fmin (v_med3) stay for those.
template <int D>
__device__ __forceinline__ void diff_extrema(const float* a, const float* b, float& t, float& u) {
  float dk[D];
#pragma unroll
  for (int k = 0; k < D; ++k) dk[k] = a[k] - b[k];
  t = dk[0];
  u = dk[0];
#pragma unroll
  for (int k = 1; k < D; ++k) {
    t = __builtin_fmaxf(t, dk[k]);
    u = __builtin_fminf(u, dk[k]);
  }
}

// _jax_ops.py:15-73: removed rows become holes.  Pair i<j:  t = max_k(q_i-q_j), u = min_k(q_i-q_j);
// j removed iff t <= 0 (ties go to the lower index), i removed iff u >= 0 and t > 0.
template <int C, int D>
__device__ __forceinline__ void c_newton(float (&q)[C * D], int nmax) {
  float acc[C];
#pragma unroll
  for (int r = 0; r < C; ++r) acc[r] = INFINITY;
  unrolled_while<0, C - 1>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    if (i + 1 >= nmax) return false;
    unrolled_while<i + 1, C>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      if (j >= nmax) return false;
      float t, u;
      diff_extrema<D>(&q[i * D], &q[j * D], t, u);
      acc[j] = hk_fmin(acc[j], t);
      acc[i] = hk_fmin(acc[i], (t > 0.0f) ? -u : 1.0f);
      return true;
    });
    return true;
  });
  unrolled_while<0, C>([&](auto rc) {
    constexpr int r = decltype(rc)::value;
    if (r >= nmax) return false;
    const bool removed = acc[r] <= 0.0f;
#pragma unroll
    for (int k = 0; k < D; ++k) q[r * D + k] = removed ? INFINITY : q[r * D + k];
    return true;
  });
}

// _jax_ops.py:93-111 / _torch_ops.py:136-146 on the live rows (the maximum over a game that has a
// live row is attained on a live row; a game without one is all padding and does not change)
template <int C, int D>
__device__ __forceinline__ void c_rescale(float (&q)[C * D], int nmax, unsigned flags) {
  const bool jax_sem = (flags & HK_SEM_MASK) == HK_SEM_JAX;
  float mx = -1.0f;
  unrolled_while<0, C>([&](auto rc) {
    constexpr int r = decltype(rc)::value;
    if (r >= nmax) return false;
    const bool live = q[r * D] < INFINITY;
#pragma unroll
    for (int k = 0; k < D; ++k) mx = hk_fmax(mx, live ? q[r * D + k] : -1.0f);
    return true;
  });
  const bool skip = jax_sem ? (mx <= 1e-8f) : (mx < 0.0f);
  const float div = (skip || mx == 0.0f) ? 1.0f : mx;
  unrolled_while<0, C>([&](auto rc) {
    constexpr int r = decltype(rc)::value;
    if (r >= nmax) return false;
    const bool live = q[r * D] < INFINITY;
#pragma unroll
    for (int k = 0; k < D; ++k) q[r * D + k] = live ? q[r * D + k] / div : INFINITY;
    return true;
  });
}

// ---- straight-line bodies ------------------------------------------------------------------------------
// A lone wave per SIMD issues ONE instruction (of any kind) per four cycles, so the scalar compare +
// branch that ends each row loop early costs as much as the vector work it guards, and every taken
// branch adds a fetch bubble.  The whole step therefore also exists as branch-free code for NB rows,
// NB = 1..8, 10, 12, ... C, and the wave runs the smallest body that covers nmax (rows in [nmax, NB) are
// +inf holes: they cost a few idle vector instructions and change nothing).
template <int C, int D, int NB>
__device__ __forceinline__ void b_shift(float (&q)[C * D], const float (&c)[D], int axis, int np, unsigned flags) {
  bool apply = true;
  if (flags & HK_FLAG_AXIS_NOOP_IF_INVALID) {
#pragma unroll
    for (int k = 0; k < D; ++k) {
      const float onehot = (k == axis) ? 1.0f : 0.0f;
      if (!(onehot - c[k] <= 0.0f)) apply = false;
    }
  }
  if ((flags & HK_FLAG_IGNORE_ENDED) && np < 2) apply = false;
  bool isax[D];
#pragma unroll
  for (int k = 0; k < D; ++k) isax[k] = apply && (k == axis);
#pragma unroll
  for (int r = 0; r < NB; ++r) {
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < D; ++k) s = s + q[r * D + k] * c[k];  // order 0..D-1, no contraction
    const bool live = q[r * D] < INFINITY;                   // holes would give inf*0 = NaN
#pragma unroll
    for (int k = 0; k < D; ++k) q[r * D + k] = (live && isax[k]) ? s : q[r * D + k];
  }
}

// The same shift when the subset is known to be a 0/1 mask (the in-kernel policies: bits of `cmask`): the sum of the
// chosen coordinates as selects instead of products.  Bit-identical on the rows this code sees (finite, >= +0:
// x*1 = x, x*0 = +0 and 0 + a = a), two instructions per row shorter, and a hole (all +inf) stays all +inf by itself
// -- every subset has two coordinates or more, inf + 0 = inf and there is no inf*0 -- so no per-row `live` test.
template <int C, int D, int NB>
__device__ __forceinline__ void b_shift_mask(float (&q)[C * D], uint32_t cmask, int axis, int np, unsigned flags) {
  bool apply = true;
  if (flags & HK_FLAG_AXIS_NOOP_IF_INVALID) apply = axis >= 0 && ((cmask >> axis) & 1u);
  if ((flags & HK_FLAG_IGNORE_ENDED) && np < 2) apply = false;
  bool isax[D], in[D];
#pragma unroll
  for (int k = 0; k < D; ++k) {
    isax[k] = apply && (k == axis);
    in[k] = (cmask >> k) & 1u;
  }
#pragma unroll
  for (int r = 0; r < NB; ++r) {
    float s = in[0] ? q[r * D] : 0.0f;
#pragma unroll
    for (int k = 1; k < D; ++k) s = s + (in[k] ? q[r * D + k] : 0.0f);  // order 0..D-1
#pragma unroll
    for (int k = 0; k < D; ++k) q[r * D + k] = isax[k] ? s : q[r * D + k];
  }
}

template <int C, int D, int NB, bool BIN = false>
__device__ __forceinline__ void b_reposition(float (&q)[C * D], unsigned flags) {
  const bool jax_sem = (flags & HK_SEM_MASK) == HK_SEM_JAX;
  if constexpr (BIN) {
    // rollouts with in-kernel 0/1 subsets: coordinates are >= +0 or the +inf of a hole, so the float order is the
    // unsigned order of the bit patterns (v_min_u32 / v_min3_u32, no canonicalising v_max x, x; see d_reposition)
    uint32_t mb[D];
#pragma unroll
    for (int k = 0; k < D; ++k) mb[k] = __float_as_uint(q[k]);
#pragma unroll
    for (int r = 1; r < NB; ++r)
#pragma unroll
      for (int k = 0; k < D; ++k) {
        const uint32_t w = __float_as_uint(q[r * D + k]);
        mb[k] = w < mb[k] ? w : mb[k];
      }
#pragma unroll
    for (int k = 0; k < D; ++k) {
      // (no live row in the column: subtract the largest finite float -- +inf stays +inf; see hk_duo_kernel.h)
      const float sub = __uint_as_float(mb[k] < 0x7F7FFFFFu ? mb[k] : 0x7F7FFFFFu);
#pragma unroll
      for (int r = 0; r < NB; ++r) q[r * D + k] = q[r * D + k] - sub;
    }
    return;
  }
  float mn[D];
#pragma unroll
  for (int k = 0; k < D; ++k) mn[k] = INFINITY;
#pragma unroll
  for (int r = 0; r < NB; ++r)
#pragma unroll
    for (int k = 0; k < D; ++k) mn[k] = hk_fmin(mn[k], q[r * D + k]);
  float sub[D];
#pragma unroll
  for (int k = 0; k < D; ++k)  // JAX leaves a column whose minimum is <= 0 untouched: subtract 0
    sub[k] = (mn[k] < INFINITY && (!jax_sem || mn[k] > 0.0f)) ? mn[k] : 0.0f;
#pragma unroll
  for (int r = 0; r < NB; ++r)
#pragma unroll
    for (int k = 0; k < D; ++k) q[r * D + k] = q[r * D + k] - sub[k];  // inf - sub = inf: holes stay
}

template <int C, int D, int NB>
__device__ __forceinline__ void b_newton(float (&q)[C * D]) {
  if constexpr (NB < 2) return;  // (one row: nothing to compare)
  // (every accumulator takes its first contribution by assignment -- known at compile time -- instead of starting at
  // +inf: no min(+inf, x))
  float acc[NB];
#pragma unroll
  for (int i = 0; i + 1 < NB; ++i) {
#pragma unroll
    for (int j = i + 1; j < NB; ++j) {
      float t, u;
      diff_extrema<D>(&q[i * D], &q[j * D], t, u);
      const float vi = (t > 0.0f) ? -u : 1.0f;
      acc[j] = (i == 0) ? t : hk_fmin(acc[j], t);
      acc[i] = (i == 0 && j == 1) ? vi : hk_fmin(acc[i], vi);
    }
    // the pairs are all independent; left alone the scheduler interleaves hundreds of them and the
    // differences in flight overflow the register file.  One row of pairs at a time is plenty of ILP.
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int r = 0; r < NB; ++r) {
    const bool removed = acc[r] <= 0.0f;
#pragma unroll
    for (int k = 0; k < D; ++k) q[r * D + k] = removed ? INFINITY : q[r * D + k];
  }
}

template <int C, int D, int NB>
__device__ __forceinline__ void b_rescale(float (&q)[C * D], unsigned flags) {
  const bool jax_sem = (flags & HK_SEM_MASK) == HK_SEM_JAX;
  float mx = -1.0f;
#pragma unroll
  for (int r = 0; r < NB; ++r) {
    const bool live = q[r * D] < INFINITY;
#pragma unroll
    for (int k = 0; k < D; ++k) mx = hk_fmax(mx, live ? q[r * D + k] : -1.0f);
  }
  const bool skip = jax_sem ? (mx <= 1e-8f) : (mx < 0.0f);
  const float div = (skip || mx == 0.0f) ? 1.0f : mx;
#pragma unroll
  for (int r = 0; r < NB; ++r) {
    const bool live = q[r * D] < INFINITY;
#pragma unroll
    for (int k = 0; k < D; ++k) q[r * D + k] = live ? q[r * D + k] / div : INFINITY;
  }
}

// one transition on rows [0, NB); returns the number of live rows.  BIN: the subset is the 0/1 mask `cmask`
template <int C, int D, int NB, bool BIN = false>
__device__ __forceinline__ int b_stages(float (&q)[C * D], const float (&c)[D], int axis, int np, unsigned flags,
                                        unsigned stages, uint32_t cmask = 0) {
  if (stages & HK_STAGE_SHIFT) {
    if constexpr (BIN) b_shift_mask<C, D, NB>(q, cmask, axis, np, flags);
    else b_shift<C, D, NB>(q, c, axis, np, flags);
  }
  if (stages & HK_STAGE_REPOSITION) b_reposition<C, D, NB, BIN>(q, flags);
  if (stages & HK_STAGE_NEWTON) b_newton<C, D, NB>(q);
  if (stages & HK_STAGE_RESCALE) b_rescale<C, D, NB>(q, flags);
  int n = 0;
#pragma unroll
  for (int r = 0; r < NB; ++r) n += (q[r * D] < INFINITY) ? 1 : 0;
  return n;
}

// the smallest body that covers nmax
template <int C, int D, int NB, bool BIN = false>
struct StagesFor {
  static constexpr int kNext = (NB < 8) ? NB + 1 : NB + 2;
  static __device__ __forceinline__ int run(float (&q)[C * D], int nmax, const float (&c)[D], int axis, int np,
                                            unsigned flags, unsigned stages, uint32_t cmask = 0) {
    if constexpr (NB >= C) {
      return b_stages<C, D, C, BIN>(q, c, axis, np, flags, stages, cmask);
    } else {
      if (nmax <= NB) return b_stages<C, D, NB, BIN>(q, c, axis, np, flags, stages, cmask);
      return StagesFor<C, D, kNext, BIN>::run(q, nmax, c, axis, np, flags, stages, cmask);
    }
  }
};

// The buckets of rows (1..8, 10, 12, ...) from the top down: f(NB, LO) for every bucket NB with the next smaller one
// LO (0 below the first) -- the staircase of rollout loops (hk_fast_kernel.h, hk_duo_kernel.h)
template <int NB>
struct RowLevels {
  static constexpr int kLo = (NB <= 8) ? NB - 1 : NB - 2;
  template <typename F>
  static __device__ __forceinline__ void run(F&& f) {
    f(std::integral_constant<int, NB>{}, std::integral_constant<int, kLo>{});
    if constexpr (kLo >= 1) RowLevels<kLo>::run(f);
  }
};

template <int C, int D, bool BIN = false>
__device__ __forceinline__ int run_stages(float (&q)[C * D], int nmax, const float (&c)[D], int axis, int np,
                                          unsigned flags, unsigned stages, uint32_t cmask = 0) {
  return StagesFor<C, D, 1, BIN>::run(q, nmax, c, axis, np, flags, stages, cmask);
}

// ---- observation features (jax/util.py:186-197): rows in descending order, LAST coordinate primary ------
// a's key strictly greater than b's, in the order KEY (KeyOrder); `coord0` (wave-uniform, looked at under
// kKeyLast only) switches that instance to coordinate 0 alone at run time: the register-resident kernel
// serves both feature orders with ONE copy of its unrolled pair loop (two copies spill)
template <int D, int KEY = kKeyLast>
__device__ __forceinline__ bool key_gt(const float* a, const float* b, bool coord0 = false) {
  if (KEY == kKeyCoord0) return a[0] > b[0];
  bool gt = false, eq = true;
#pragma unroll
  for (int kk = 0; kk < D; ++kk) {
    const int k = (KEY == kKeyFirst) ? kk : D - 1 - kk;
    gt |= eq && (a[k] > b[k]);
    eq &= (a[k] == b[k]);
  }
  return (KEY == kKeyLast && coord0) ? (a[0] > b[0]) : gt;
}

// rank[r] = position of live row r in the sorted order = number of live rows that come before it (greater
// key; among equal keys the lower index first).  Holes count for nothing.
template <int C, int D, int KEY = kKeyLast>
__device__ __forceinline__ void feature_ranks(const float (&q)[C * D], int nmax, int (&rank)[C], bool coord0 = false) {
#pragma unroll
  for (int r = 0; r < C; ++r) rank[r] = 0;
  unrolled_while<0, C - 1>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    if (i + 1 >= nmax) return false;
    const bool live_i = q[i * D] < INFINITY;
    unrolled_while<i + 1, C>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      if (j >= nmax) return false;
      const bool live_j = q[j * D] < INFINITY;
      const bool j_first = key_gt<D, KEY>(&q[j * D], &q[i * D], coord0);  // otherwise i (the lower index) comes first
      rank[j] += (!j_first && live_i) ? 1 : 0;
      rank[i] += (j_first && live_j) ? 1 : 0;
      return true;
    });
    return true;
  });
}

// live rows to their ranks in the (pad-filled) image
template <int C, int D>
__device__ __forceinline__ void scatter_ranked(const float (&q)[C * D], float* mine, const int (&rank)[C], int nmax) {
  unrolled_while<0, C>([&](auto rc) {
    constexpr int r = decltype(rc)::value;
    if (r >= nmax) return false;
    if (q[r * D] < INFINITY) {
      float* row = mine + rank[r] * D;
#pragma unroll
      for (int k = 0; k < D; ++k) row[k] = q[r * D + k];
    }
    return true;
  });
}

// ---- Zeillinger's host on the live rows (jax/players.py:55-109) ---------------------------------------------
// Over the pairs of live rows: characteristic vector (L, S) = (max - min, #max + #min) of the difference
// P_i - P_j; pairs whose difference is constant (jnp.isclose(max, min)) do not count; the first minimum of
// (L, S) in row-major order wins -- among ordered pairs that is always an (i, j) with i < j, and the
// compaction keeps the rows' order, so the triangle i < j in compact order finds the same pair.  The
// subset is {argmin, argmax} of that difference; no valid pair -> class 0 (players.py:96-109).
// LIST: Zeillinger._select_coord (host.py:70-95) instead -- every pair of available rows counts (no isclose
// filter), the rows are in the state's physical order (which the gather keeps), a game with fewer than two
// rows gives -1 and coinciding argmin / argmax the subset {0, 1}.  BOTH: the variant is the wave-uniform
// runtime flag `list` (hk_zeillinger's kernels: one copy of the pair loop for the two variants).
template <int C, int D, bool BOTH = false>
__device__ __forceinline__ int c_zeillinger(const float (&q)[C * D], int nmax, bool list_rt = false) {
  const bool LIST = BOTH && list_rt;
  float bestL = INFINITY, bestS = INFINITY;
  float bd[D];
#pragma unroll
  for (int k = 0; k < D; ++k) bd[k] = 0.0f;
  bool have = false;
  unrolled_while<0, C - 1>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    if (i + 1 >= nmax) return false;
    const bool live_i = q[i * D] < INFINITY;
    unrolled_while<i + 1, C>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      if (j >= nmax) return false;
      float v[D];
#pragma unroll
      for (int k = 0; k < D; ++k) v[k] = q[i * D + k] - q[j * D + k];
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
      // (bitwise on purpose: && / || become branches on the exec mask, one per term and pair)
      const bool valid = live_i & (q[j * D] < INFINITY) & (LIST | !close);
      const bool better = valid & ((LIST & !have) | (L < bestL) | ((L == bestL) & (cnt < bestS)));
      bestL = better ? L : bestL;
      bestS = better ? cnt : bestS;
#pragma unroll
      for (int k = 0; k < D; ++k) bd[k] = better ? v[k] : bd[k];
      have |= better;
      return true;
    });
    return true;
  });
  int lo = 0, hi = 0;
  float vlo = bd[0], vhi = bd[0];
#pragma unroll
  for (int k = 1; k < D; ++k) {
    if (bd[k] < vlo) { vlo = bd[k]; lo = k; }
    if (bd[k] > vhi) { vhi = bd[k]; hi = k; }
  }
  if (LIST) {
    if (!have) return -1;
    return (lo == hi) ? encode_mask(3u) : encode_mask((1u << lo) | (1u << hi));
  }
  if (!have || lo == hi) return 0;
  return encode_mask((1u << lo) | (1u << hi));
}

// ---- multi-word unsigned compares as borrow chains (v_subb_co, the borrow travelling in a scalar register pair), their
// results added to counters (v_addc_co) or selecting (v_cndmask) without a detour through scalar and/or logic ----------
// Inline asm, because the compiler turns every C++ spelling of a borrow chain (__builtin_subc, 128-bit subtraction with
// overflow) back into compares -- and the 64-bit compares it makes of it are slower here than what they replace
// (measured: hk_get_features at (50,4) 137 us against 85, dense states 926 against 177).  gfx950 wants two wait states
// between a vector instruction that WRITES a scalar register and a vector instruction that reads it, and the
// compiler's hazard recogniser does not look into inline asm: every instruction below that reads a lane mask waits
// for itself (s_nop 1), whatever the scheduler put before it -- or sits in a block whose order provides the distance
// (kb_rank3 / kb_rank2 / zeil_pair2).  Their VGPR results are never read through DPP without a compiler-generated
// instruction in between (zeil_merge is plain C++ for that reason); scripts/check_sgpr_hazards.py scans a listing.
using LaneMask = uint64_t;
__device__ __forceinline__ LaneMask kb_subb(uint32_t mine, uint32_t other, LaneMask borrow_in) {
  uint32_t diff;
  LaneMask borrow;
  asm("s_nop 1\n\tv_subb_co_u32_e64 %0, %1, %2, %3, %4" : "=v"(diff), "=s"(borrow) : "v"(mine), "v"(other), "s"(borrow_in));
  return borrow;
}
__device__ __forceinline__ void kb_count(uint32_t& n, LaneMask c) {
  LaneMask carry;
  asm("s_nop 1\n\tv_addc_co_u32_e64 %0, %1, 0, %0, %2" : "+v"(n), "=s"(carry) : "s"(c));
}
// lanes where `other` comes first: other > mine, or other == mine on the lanes of `tie` (words: least significant first)
template <int W>
__device__ __forceinline__ LaneMask key_other_first(const uint32_t (&mine)[W], const uint32_t (&other)[W], LaneMask tie) {
  LaneMask c = kb_subb(mine[0], other[0], tie);
#pragma unroll
  for (int i = 1; i < W; ++i) c = kb_subb(mine[i], other[i], c);
  return c;
}
__device__ __forceinline__ uint32_t kb_select(uint32_t keep, uint32_t take, LaneMask c) {
  uint32_t r;
  asm("s_nop 1\n\tv_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(keep), "v"(take), "s"(c));
  return r;
}
__device__ __forceinline__ float kb_select(float keep, float take, LaneMask c) {
  return __uint_as_float(kb_select(__float_as_uint(keep), __float_as_uint(take), c));
}
// Three (two) chains against the SAME other row, interleaved link by link, and their counts, as one block: a link reads
// the borrow its chain wrote three (two + s_nop 0) instructions earlier, a count the borrow of a chain that ended two
// instructions before it -- the two wait states are in the order of the instructions, no s_nop per instruction.
// r?: += (other comes before m?);  w (SHARED): += all of them (what the other row "won").
template <int W, bool SHARED>
__device__ __forceinline__ void kb_rank3(const uint32_t (&o)[W], const uint32_t (&mA)[W], const uint32_t (&mB)[W],
                                         const uint32_t (&mC)[W], LaneMask tA, LaneMask tB, LaneMask tC, uint32_t& rA,
                                         uint32_t& rB, uint32_t& rC, uint32_t& w) {
  uint32_t j;
  LaneMask x, cA, cB, cC;
  static_assert(W == 1 || W == 3 || W == 4, "keys of one, three or four words");
  if constexpr (W == 1 && SHARED)
    asm("s_nop 1\n\t"
        "v_subb_co_u32_e64 %[j], %[cA], %[mA0], %[o0], %[tA]\n\t"
        "v_subb_co_u32_e64 %[j], %[cB], %[mB0], %[o0], %[tB]\n\t"
        "v_subb_co_u32_e64 %[j], %[cC], %[mC0], %[o0], %[tC]\n\t"
        "v_addc_co_u32_e64 %[rA], %[x], 0, %[rA], %[cA]\n\t"
        "v_addc_co_u32_e64 %[rB], %[x], 0, %[rB], %[cB]\n\t"
        "v_addc_co_u32_e64 %[rC], %[x], 0, %[rC], %[cC]\n\t"
        "v_addc_co_u32_e64 %[w], %[x], 0, %[w], %[cA]\n\t"
        "v_addc_co_u32_e64 %[w], %[x], 0, %[w], %[cB]\n\t"
        "v_addc_co_u32_e64 %[w], %[x], 0, %[w], %[cC]\n\t"
        : [rA] "+v"(rA), [rB] "+v"(rB), [rC] "+v"(rC), [w] "+v"(w), [j] "=&v"(j), [x] "=&s"(x), [cA] "=&s"(cA), [cB] "=&s"(cB), [cC] "=&s"(cC)
        : [o0] "v"(o[0]), [mA0] "v"(mA[0]), [mB0] "v"(mB[0]), [mC0] "v"(mC[0]), [tA] "s"(tA), [tB] "s"(tB), [tC] "s"(tC));
  else if constexpr (W == 1 && !SHARED)
    asm("s_nop 1\n\t"
        "v_subb_co_u32_e64 %[j], %[cA], %[mA0], %[o0], %[tA]\n\t"
        "v_subb_co_u32_e64 %[j], %[cB], %[mB0], %[o0], %[tB]\n\t"
        "v_subb_co_u32_e64 %[j], %[cC], %[mC0], %[o0], %[tC]\n\t"
        "v_addc_co_u32_e64 %[rA], %[x], 0, %[rA], %[cA]\n\t"
        "v_addc_co_u32_e64 %[rB], %[x], 0, %[rB], %[cB]\n\t"
        "v_addc_co_u32_e64 %[rC], %[x], 0, %[rC], %[cC]\n\t"
        : [rA] "+v"(rA), [rB] "+v"(rB), [rC] "+v"(rC), [j] "=&v"(j), [x] "=&s"(x), [cA] "=&s"(cA), [cB] "=&s"(cB), [cC] "=&s"(cC)
        : [o0] "v"(o[0]), [mA0] "v"(mA[0]), [mB0] "v"(mB[0]), [mC0] "v"(mC[0]), [tA] "s"(tA), [tB] "s"(tB), [tC] "s"(tC));
  else if constexpr (W == 3 && SHARED)
    asm("s_nop 1\n\t"
        "v_subb_co_u32_e64 %[j], %[cA], %[mA0], %[o0], %[tA]\n\t"
        "v_subb_co_u32_e64 %[j], %[cB], %[mB0], %[o0], %[tB]\n\t"
        "v_subb_co_u32_e64 %[j], %[cC], %[mC0], %[o0], %[tC]\n\t"
        "v_subb_co_u32_e64 %[j], %[cA], %[mA1], %[o1], %[cA]\n\t"
        "v_subb_co_u32_e64 %[j], %[cB], %[mB1], %[o1], %[cB]\n\t"
        "v_subb_co_u32_e64 %[j], %[cC], %[mC1], %[o1], %[cC]\n\t"
        "v_subb_co_u32_e64 %[j], %[cA], %[mA2], %[o2], %[cA]\n\t"
        "v_subb_co_u32_e64 %[j], %[cB], %[mB2], %[o2], %[cB]\n\t"
        "v_subb_co_u32_e64 %[j], %[cC], %[mC2], %[o2], %[cC]\n\t"
        "v_addc_co_u32_e64 %[rA], %[x], 0, %[rA], %[cA]\n\t"
        "v_addc_co_u32_e64 %[rB], %[x], 0, %[rB], %[cB]\n\t"
        "v_addc_co_u32_e64 %[rC], %[x], 0, %[rC], %[cC]\n\t"
        "v_addc_co_u32_e64 %[w], %[x], 0, %[w], %[cA]\n\t"
        "v_addc_co_u32_e64 %[w], %[x], 0, %[w], %[cB]\n\t"
        "v_addc_co_u32_e64 %[w], %[x], 0, %[w], %[cC]\n\t"
        : [rA] "+v"(rA), [rB] "+v"(rB), [rC] "+v"(rC), [w] "+v"(w), [j] "=&v"(j), [x] "=&s"(x), [cA] "=&s"(cA), [cB] "=&s"(cB), [cC] "=&s"(cC)
        : [o0] "v"(o[0]), [o1] "v"(o[1]), [o2] "v"(o[2]), [mA0] "v"(mA[0]), [mA1] "v"(mA[1]), [mA2] "v"(mA[2]), [mB0] "v"(mB[0]), [mB1] "v"(mB[1]), [mB2] "v"(mB[2]), [mC0] "v"(mC[0]), [mC1] "v"(mC[1]), [mC2] "v"(mC[2]), [tA] "s"(tA), [tB] "s"(tB), [tC] "s"(tC));
  else if constexpr (W == 3 && !SHARED)
    asm("s_nop 1\n\t"
        "v_subb_co_u32_e64 %[j], %[cA], %[mA0], %[o0], %[tA]\n\t"
        "v_subb_co_u32_e64 %[j], %[cB], %[mB0], %[o0], %[tB]\n\t"
        "v_subb_co_u32_e64 %[j], %[cC], %[mC0], %[o0], %[tC]\n\t"
        "v_subb_co_u32_e64 %[j], %[cA], %[mA1], %[o1], %[cA]\n\t"
        "v_subb_co_u32_e64 %[j], %[cB], %[mB1], %[o1], %[cB]\n\t"
        "v_subb_co_u32_e64 %[j], %[cC], %[mC1], %[o1], %[cC]\n\t"
        "v_subb_co_u32_e64 %[j], %[cA], %[mA2], %[o2], %[cA]\n\t"
        "v_subb_co_u32_e64 %[j], %[cB], %[mB2], %[o2], %[cB]\n\t"
        "v_subb_co_u32_e64 %[j], %[cC], %[mC2], %[o2], %[cC]\n\t"
        "v_addc_co_u32_e64 %[rA], %[x], 0, %[rA], %[cA]\n\t"
        "v_addc_co_u32_e64 %[rB], %[x], 0, %[rB], %[cB]\n\t"
        "v_addc_co_u32_e64 %[rC], %[x], 0, %[rC], %[cC]\n\t"
        : [rA] "+v"(rA), [rB] "+v"(rB), [rC] "+v"(rC), [j] "=&v"(j), [x] "=&s"(x), [cA] "=&s"(cA), [cB] "=&s"(cB), [cC] "=&s"(cC)
        : [o0] "v"(o[0]), [o1] "v"(o[1]), [o2] "v"(o[2]), [mA0] "v"(mA[0]), [mA1] "v"(mA[1]), [mA2] "v"(mA[2]), [mB0] "v"(mB[0]), [mB1] "v"(mB[1]), [mB2] "v"(mB[2]), [mC0] "v"(mC[0]), [mC1] "v"(mC[1]), [mC2] "v"(mC[2]), [tA] "s"(tA), [tB] "s"(tB), [tC] "s"(tC));
  else if constexpr (W == 4 && SHARED)
    asm("s_nop 1\n\t"
        "v_subb_co_u32_e64 %[j], %[cA], %[mA0], %[o0], %[tA]\n\t"
        "v_subb_co_u32_e64 %[j], %[cB], %[mB0], %[o0], %[tB]\n\t"
        "v_subb_co_u32_e64 %[j], %[cC], %[mC0], %[o0], %[tC]\n\t"
        "v_subb_co_u32_e64 %[j], %[cA], %[mA1], %[o1], %[cA]\n\t"
        "v_subb_co_u32_e64 %[j], %[cB], %[mB1], %[o1], %[cB]\n\t"
        "v_subb_co_u32_e64 %[j], %[cC], %[mC1], %[o1], %[cC]\n\t"
        "v_subb_co_u32_e64 %[j], %[cA], %[mA2], %[o2], %[cA]\n\t"
        "v_subb_co_u32_e64 %[j], %[cB], %[mB2], %[o2], %[cB]\n\t"
        "v_subb_co_u32_e64 %[j], %[cC], %[mC2], %[o2], %[cC]\n\t"
        "v_subb_co_u32_e64 %[j], %[cA], %[mA3], %[o3], %[cA]\n\t"
        "v_subb_co_u32_e64 %[j], %[cB], %[mB3], %[o3], %[cB]\n\t"
        "v_subb_co_u32_e64 %[j], %[cC], %[mC3], %[o3], %[cC]\n\t"
        "v_addc_co_u32_e64 %[rA], %[x], 0, %[rA], %[cA]\n\t"
        "v_addc_co_u32_e64 %[rB], %[x], 0, %[rB], %[cB]\n\t"
        "v_addc_co_u32_e64 %[rC], %[x], 0, %[rC], %[cC]\n\t"
        "v_addc_co_u32_e64 %[w], %[x], 0, %[w], %[cA]\n\t"
        "v_addc_co_u32_e64 %[w], %[x], 0, %[w], %[cB]\n\t"
        "v_addc_co_u32_e64 %[w], %[x], 0, %[w], %[cC]\n\t"
        : [rA] "+v"(rA), [rB] "+v"(rB), [rC] "+v"(rC), [w] "+v"(w), [j] "=&v"(j), [x] "=&s"(x), [cA] "=&s"(cA), [cB] "=&s"(cB), [cC] "=&s"(cC)
        : [o0] "v"(o[0]), [o1] "v"(o[1]), [o2] "v"(o[2]), [o3] "v"(o[3]), [mA0] "v"(mA[0]), [mA1] "v"(mA[1]), [mA2] "v"(mA[2]), [mA3] "v"(mA[3]), [mB0] "v"(mB[0]), [mB1] "v"(mB[1]), [mB2] "v"(mB[2]), [mB3] "v"(mB[3]), [mC0] "v"(mC[0]), [mC1] "v"(mC[1]), [mC2] "v"(mC[2]), [mC3] "v"(mC[3]), [tA] "s"(tA), [tB] "s"(tB), [tC] "s"(tC));
  else if constexpr (W == 4 && !SHARED)
    asm("s_nop 1\n\t"
        "v_subb_co_u32_e64 %[j], %[cA], %[mA0], %[o0], %[tA]\n\t"
        "v_subb_co_u32_e64 %[j], %[cB], %[mB0], %[o0], %[tB]\n\t"
        "v_subb_co_u32_e64 %[j], %[cC], %[mC0], %[o0], %[tC]\n\t"
        "v_subb_co_u32_e64 %[j], %[cA], %[mA1], %[o1], %[cA]\n\t"
        "v_subb_co_u32_e64 %[j], %[cB], %[mB1], %[o1], %[cB]\n\t"
        "v_subb_co_u32_e64 %[j], %[cC], %[mC1], %[o1], %[cC]\n\t"
        "v_subb_co_u32_e64 %[j], %[cA], %[mA2], %[o2], %[cA]\n\t"
        "v_subb_co_u32_e64 %[j], %[cB], %[mB2], %[o2], %[cB]\n\t"
        "v_subb_co_u32_e64 %[j], %[cC], %[mC2], %[o2], %[cC]\n\t"
        "v_subb_co_u32_e64 %[j], %[cA], %[mA3], %[o3], %[cA]\n\t"
        "v_subb_co_u32_e64 %[j], %[cB], %[mB3], %[o3], %[cB]\n\t"
        "v_subb_co_u32_e64 %[j], %[cC], %[mC3], %[o3], %[cC]\n\t"
        "v_addc_co_u32_e64 %[rA], %[x], 0, %[rA], %[cA]\n\t"
        "v_addc_co_u32_e64 %[rB], %[x], 0, %[rB], %[cB]\n\t"
        "v_addc_co_u32_e64 %[rC], %[x], 0, %[rC], %[cC]\n\t"
        : [rA] "+v"(rA), [rB] "+v"(rB), [rC] "+v"(rC), [j] "=&v"(j), [x] "=&s"(x), [cA] "=&s"(cA), [cB] "=&s"(cB), [cC] "=&s"(cC)
        : [o0] "v"(o[0]), [o1] "v"(o[1]), [o2] "v"(o[2]), [o3] "v"(o[3]), [mA0] "v"(mA[0]), [mA1] "v"(mA[1]), [mA2] "v"(mA[2]), [mA3] "v"(mA[3]), [mB0] "v"(mB[0]), [mB1] "v"(mB[1]), [mB2] "v"(mB[2]), [mB3] "v"(mB[3]), [mC0] "v"(mC[0]), [mC1] "v"(mC[1]), [mC2] "v"(mC[2]), [mC3] "v"(mC[3]), [tA] "s"(tA), [tB] "s"(tB), [tC] "s"(tC));
}
template <int W, bool SHARED>
__device__ __forceinline__ void kb_rank2(const uint32_t (&o)[W], const uint32_t (&mA)[W], const uint32_t (&mB)[W],
                                         LaneMask tA, LaneMask tB, uint32_t& rA, uint32_t& rB, uint32_t& w) {
  uint32_t j;
  LaneMask x, cA, cB;
  static_assert(W == 1 || W == 3 || W == 4, "keys of one, three or four words");
  if constexpr (W == 1 && SHARED)
    asm("s_nop 1\n\t"
        "v_subb_co_u32_e64 %[j], %[cA], %[mA0], %[o0], %[tA]\n\t"
        "v_subb_co_u32_e64 %[j], %[cB], %[mB0], %[o0], %[tB]\n\t"
        "s_nop 0\n\t"
        "v_addc_co_u32_e64 %[rA], %[x], 0, %[rA], %[cA]\n\t"
        "v_addc_co_u32_e64 %[rB], %[x], 0, %[rB], %[cB]\n\t"
        "v_addc_co_u32_e64 %[w], %[x], 0, %[w], %[cA]\n\t"
        "v_addc_co_u32_e64 %[w], %[x], 0, %[w], %[cB]\n\t"
        : [rA] "+v"(rA), [rB] "+v"(rB), [w] "+v"(w), [j] "=&v"(j), [x] "=&s"(x), [cA] "=&s"(cA), [cB] "=&s"(cB)
        : [o0] "v"(o[0]), [mA0] "v"(mA[0]), [mB0] "v"(mB[0]), [tA] "s"(tA), [tB] "s"(tB));
  else if constexpr (W == 1 && !SHARED)
    asm("s_nop 1\n\t"
        "v_subb_co_u32_e64 %[j], %[cA], %[mA0], %[o0], %[tA]\n\t"
        "v_subb_co_u32_e64 %[j], %[cB], %[mB0], %[o0], %[tB]\n\t"
        "s_nop 0\n\t"
        "v_addc_co_u32_e64 %[rA], %[x], 0, %[rA], %[cA]\n\t"
        "v_addc_co_u32_e64 %[rB], %[x], 0, %[rB], %[cB]\n\t"
        : [rA] "+v"(rA), [rB] "+v"(rB), [j] "=&v"(j), [x] "=&s"(x), [cA] "=&s"(cA), [cB] "=&s"(cB)
        : [o0] "v"(o[0]), [mA0] "v"(mA[0]), [mB0] "v"(mB[0]), [tA] "s"(tA), [tB] "s"(tB));
  else if constexpr (W == 3 && SHARED)
    asm("s_nop 1\n\t"
        "v_subb_co_u32_e64 %[j], %[cA], %[mA0], %[o0], %[tA]\n\t"
        "v_subb_co_u32_e64 %[j], %[cB], %[mB0], %[o0], %[tB]\n\t"
        "s_nop 0\n\t"
        "v_subb_co_u32_e64 %[j], %[cA], %[mA1], %[o1], %[cA]\n\t"
        "v_subb_co_u32_e64 %[j], %[cB], %[mB1], %[o1], %[cB]\n\t"
        "s_nop 0\n\t"
        "v_subb_co_u32_e64 %[j], %[cA], %[mA2], %[o2], %[cA]\n\t"
        "v_subb_co_u32_e64 %[j], %[cB], %[mB2], %[o2], %[cB]\n\t"
        "s_nop 0\n\t"
        "v_addc_co_u32_e64 %[rA], %[x], 0, %[rA], %[cA]\n\t"
        "v_addc_co_u32_e64 %[rB], %[x], 0, %[rB], %[cB]\n\t"
        "v_addc_co_u32_e64 %[w], %[x], 0, %[w], %[cA]\n\t"
        "v_addc_co_u32_e64 %[w], %[x], 0, %[w], %[cB]\n\t"
        : [rA] "+v"(rA), [rB] "+v"(rB), [w] "+v"(w), [j] "=&v"(j), [x] "=&s"(x), [cA] "=&s"(cA), [cB] "=&s"(cB)
        : [o0] "v"(o[0]), [o1] "v"(o[1]), [o2] "v"(o[2]), [mA0] "v"(mA[0]), [mA1] "v"(mA[1]), [mA2] "v"(mA[2]), [mB0] "v"(mB[0]), [mB1] "v"(mB[1]), [mB2] "v"(mB[2]), [tA] "s"(tA), [tB] "s"(tB));
  else if constexpr (W == 3 && !SHARED)
    asm("s_nop 1\n\t"
        "v_subb_co_u32_e64 %[j], %[cA], %[mA0], %[o0], %[tA]\n\t"
        "v_subb_co_u32_e64 %[j], %[cB], %[mB0], %[o0], %[tB]\n\t"
        "s_nop 0\n\t"
        "v_subb_co_u32_e64 %[j], %[cA], %[mA1], %[o1], %[cA]\n\t"
        "v_subb_co_u32_e64 %[j], %[cB], %[mB1], %[o1], %[cB]\n\t"
        "s_nop 0\n\t"
        "v_subb_co_u32_e64 %[j], %[cA], %[mA2], %[o2], %[cA]\n\t"
        "v_subb_co_u32_e64 %[j], %[cB], %[mB2], %[o2], %[cB]\n\t"
        "s_nop 0\n\t"
        "v_addc_co_u32_e64 %[rA], %[x], 0, %[rA], %[cA]\n\t"
        "v_addc_co_u32_e64 %[rB], %[x], 0, %[rB], %[cB]\n\t"
        : [rA] "+v"(rA), [rB] "+v"(rB), [j] "=&v"(j), [x] "=&s"(x), [cA] "=&s"(cA), [cB] "=&s"(cB)
        : [o0] "v"(o[0]), [o1] "v"(o[1]), [o2] "v"(o[2]), [mA0] "v"(mA[0]), [mA1] "v"(mA[1]), [mA2] "v"(mA[2]), [mB0] "v"(mB[0]), [mB1] "v"(mB[1]), [mB2] "v"(mB[2]), [tA] "s"(tA), [tB] "s"(tB));
  else if constexpr (W == 4 && SHARED)
    asm("s_nop 1\n\t"
        "v_subb_co_u32_e64 %[j], %[cA], %[mA0], %[o0], %[tA]\n\t"
        "v_subb_co_u32_e64 %[j], %[cB], %[mB0], %[o0], %[tB]\n\t"
        "s_nop 0\n\t"
        "v_subb_co_u32_e64 %[j], %[cA], %[mA1], %[o1], %[cA]\n\t"
        "v_subb_co_u32_e64 %[j], %[cB], %[mB1], %[o1], %[cB]\n\t"
        "s_nop 0\n\t"
        "v_subb_co_u32_e64 %[j], %[cA], %[mA2], %[o2], %[cA]\n\t"
        "v_subb_co_u32_e64 %[j], %[cB], %[mB2], %[o2], %[cB]\n\t"
        "s_nop 0\n\t"
        "v_subb_co_u32_e64 %[j], %[cA], %[mA3], %[o3], %[cA]\n\t"
        "v_subb_co_u32_e64 %[j], %[cB], %[mB3], %[o3], %[cB]\n\t"
        "s_nop 0\n\t"
        "v_addc_co_u32_e64 %[rA], %[x], 0, %[rA], %[cA]\n\t"
        "v_addc_co_u32_e64 %[rB], %[x], 0, %[rB], %[cB]\n\t"
        "v_addc_co_u32_e64 %[w], %[x], 0, %[w], %[cA]\n\t"
        "v_addc_co_u32_e64 %[w], %[x], 0, %[w], %[cB]\n\t"
        : [rA] "+v"(rA), [rB] "+v"(rB), [w] "+v"(w), [j] "=&v"(j), [x] "=&s"(x), [cA] "=&s"(cA), [cB] "=&s"(cB)
        : [o0] "v"(o[0]), [o1] "v"(o[1]), [o2] "v"(o[2]), [o3] "v"(o[3]), [mA0] "v"(mA[0]), [mA1] "v"(mA[1]), [mA2] "v"(mA[2]), [mA3] "v"(mA[3]), [mB0] "v"(mB[0]), [mB1] "v"(mB[1]), [mB2] "v"(mB[2]), [mB3] "v"(mB[3]), [tA] "s"(tA), [tB] "s"(tB));
  else if constexpr (W == 4 && !SHARED)
    asm("s_nop 1\n\t"
        "v_subb_co_u32_e64 %[j], %[cA], %[mA0], %[o0], %[tA]\n\t"
        "v_subb_co_u32_e64 %[j], %[cB], %[mB0], %[o0], %[tB]\n\t"
        "s_nop 0\n\t"
        "v_subb_co_u32_e64 %[j], %[cA], %[mA1], %[o1], %[cA]\n\t"
        "v_subb_co_u32_e64 %[j], %[cB], %[mB1], %[o1], %[cB]\n\t"
        "s_nop 0\n\t"
        "v_subb_co_u32_e64 %[j], %[cA], %[mA2], %[o2], %[cA]\n\t"
        "v_subb_co_u32_e64 %[j], %[cB], %[mB2], %[o2], %[cB]\n\t"
        "s_nop 0\n\t"
        "v_subb_co_u32_e64 %[j], %[cA], %[mA3], %[o3], %[cA]\n\t"
        "v_subb_co_u32_e64 %[j], %[cB], %[mB3], %[o3], %[cB]\n\t"
        "s_nop 0\n\t"
        "v_addc_co_u32_e64 %[rA], %[x], 0, %[rA], %[cA]\n\t"
        "v_addc_co_u32_e64 %[rB], %[x], 0, %[rB], %[cB]\n\t"
        : [rA] "+v"(rA), [rB] "+v"(rB), [j] "=&v"(j), [x] "=&s"(x), [cA] "=&s"(cA), [cB] "=&s"(cB)
        : [o0] "v"(o[0]), [o1] "v"(o[1]), [o2] "v"(o[2]), [o3] "v"(o[3]), [mA0] "v"(mA[0]), [mA1] "v"(mA[1]), [mA2] "v"(mA[2]), [mA3] "v"(mA[3]), [mB0] "v"(mB[0]), [mB1] "v"(mB[1]), [mB2] "v"(mB[2]), [mB3] "v"(mB[3]), [tA] "s"(tA), [tB] "s"(tB));
}
// N rows mine[0 .. N) against one other row: blocks of three, then two, then the single chain
template <int W, int N, bool SHARED, typename MineAt, typename TieAt, typename RankAt>
__device__ __forceinline__ void kb_rank_rows(const uint32_t (&o)[W], MineAt&& mine_at, TieAt&& tie_at, RankAt&& rank_at,
                                             uint32_t& w) {
  unrolled_while<0, (N + 2) / 3>([&](auto gc) {
    constexpr int a = 3 * decltype(gc)::value;
    if constexpr (a + 3 <= N) {
      kb_rank3<W, SHARED>(o, mine_at(std::integral_constant<int, a>{}), mine_at(std::integral_constant<int, a + 1>{}),
                          mine_at(std::integral_constant<int, a + 2>{}), tie_at(std::integral_constant<int, a>{}),
                          tie_at(std::integral_constant<int, a + 1>{}), tie_at(std::integral_constant<int, a + 2>{}),
                          rank_at(std::integral_constant<int, a>{}), rank_at(std::integral_constant<int, a + 1>{}),
                          rank_at(std::integral_constant<int, a + 2>{}), w);
    } else if constexpr (a + 2 == N) {
      kb_rank2<W, SHARED>(o, mine_at(std::integral_constant<int, a>{}), mine_at(std::integral_constant<int, a + 1>{}),
                          tie_at(std::integral_constant<int, a>{}), tie_at(std::integral_constant<int, a + 1>{}),
                          rank_at(std::integral_constant<int, a>{}), rank_at(std::integral_constant<int, a + 1>{}), w);
    } else if constexpr (a + 1 == N) {
      const LaneMask c = key_other_first<W>(mine_at(std::integral_constant<int, a>{}), o, tie_at(std::integral_constant<int, a>{}));
      kb_count(rank_at(std::integral_constant<int, a>{}), c);
      if constexpr (SHARED) kb_count(w, c);
    }
    return true;
  });
}


// ---- Zeillinger's pair test for the kernels that split a game over lanes (hk_duo_kernel.h, hk_quadroll_kernel.h) ------
// The best pair so far as ONE comparable key: hi = the bits of L (a non-negative finite float: its bit pattern orders
// like its value), lo = S << 16 | 64 i + j -- "smaller (L, S), then the earlier pair" is an unsigned compare of (hi, lo).
template <int D>
struct ZeilBest {
  // none yet: just above the pattern of +inf -- below every NaN, which is what L comes out as when either row is a hole
  // (every difference +-inf or NaN): holes need no test of their own
  static constexpr uint32_t kNone = 0x7F800001u;
  uint32_t hi = kNone, lo = 0xFFFFFFFFu;
  float bd[D];  // the pair's difference (KEEP; else re-read from the parked rows)
  __device__ __forceinline__ bool have() const { return hi < kNone; }
};

// one pair (mine = the earlier row i, other = row j): its characteristic vector against the best so far.  max / min /
// median of three are single instructions, #max + #min = 2 + (median == max) + (median == min) in dimension 3; in
// dimension 4 two rounds of a sorting network leave minimum, maximum and the two middle values, #max + #min = 2 + the
// middle values' matches.  "better" is the borrow of key - best over the two words, selecting directly.
// OK = false: the caller has no pairs to leave out (`ok` is not looked at).
// the pair's key: hi = the pattern of L (all ones: the pair does not count), lo = S << 16 | idx
template <int D, bool OK>
__device__ __forceinline__ void zeil_key(const float* mine, const float* other, bool ok, int idx, uint32_t& khi,
                                         uint32_t& klo, float (&v)[D]) {
#pragma unroll
  for (int k = 0; k < D; ++k) v[k] = mine[k] - other[k];
  float mx, mn;
  uint32_t cnt;
  if constexpr (D == 3) {
    mx = __builtin_fmaxf(__builtin_fmaxf(v[0], v[1]), v[2]);
    mn = __builtin_fminf(__builtin_fminf(v[0], v[1]), v[2]);
    const float md = __builtin_amdgcn_fmed3f(v[0], v[1], v[2]);
    cnt = 2u + (uint32_t)(md == mx) + (uint32_t)(md == mn);
  } else if constexpr (D == 4) {
    const float lo1 = __builtin_fminf(v[0], v[1]), hi1 = __builtin_fmaxf(v[0], v[1]);
    const float lo2 = __builtin_fminf(v[2], v[3]), hi2 = __builtin_fmaxf(v[2], v[3]);
    mn = __builtin_fminf(lo1, lo2);
    mx = __builtin_fmaxf(hi1, hi2);
    const float m1 = __builtin_fmaxf(lo1, lo2), m2 = __builtin_fminf(hi1, hi2);
    // (all four equal would count 6, not 8: that pair is `close` and does not count at all)
    cnt = 2u + (uint32_t)(m1 == mn) + (uint32_t)(m2 == mn) + (uint32_t)(m1 == mx) + (uint32_t)(m2 == mx);
  } else {
    mx = v[0];
    mn = v[0];
#pragma unroll
    for (int k = 1; k < D; ++k) {
      mx = __builtin_fmaxf(mx, v[k]);
      mn = __builtin_fminf(mn, v[k]);
    }
    cnt = 0;
#pragma unroll
    for (int k = 0; k < D; ++k) cnt += (uint32_t)(v[k] == mx) + (uint32_t)(v[k] == mn);
  }
  const float L = mx - mn;
  const bool close = fabsf(L) <= 1e-8f + 1e-5f * fabsf(mn);  // jnp.isclose(max, min)
  const bool out = OK ? (close | !ok) : close;
  khi = out ? 0xFFFFFFFFu : __float_as_uint(L);
  klo = (cnt << 16) | (uint32_t)idx;
}

template <int D, bool KEEP, bool OK = true>
__device__ __forceinline__ void zeil_pair(ZeilBest<D>& best, const float* mine, const float* other, bool ok, int idx) {
  float v[D];
  uint32_t khi, klo;
  zeil_key<D, OK>(mine, other, ok, idx, khi, klo, v);
  const LaneMask better = kb_subb(khi, best.hi, kb_subb(klo, best.lo, 0ull));  // best > key
  best.hi = kb_select(best.hi, khi, better);
  best.lo = kb_select(best.lo, klo, better);
  if constexpr (KEEP) {
#pragma unroll
    for (int k = 0; k < D; ++k) best.bd[k] = kb_select(best.bd[k], v[k], better);
  }
}

// the same with the update as plain compares and selects (the two-lane kernel's unrolled DPP variant: it keeps the
// difference too, and there the compiler's own sequence measured ahead of the chain: 37 against 42 us per episode)
template <int D, bool KEEP, bool OK = true>
__device__ __forceinline__ void zeil_pair_cmp(ZeilBest<D>& best, const float* mine, const float* other, bool ok, int idx) {
  float v[D];
#pragma unroll
  for (int k = 0; k < D; ++k) v[k] = mine[k] - other[k];
  float mx, mn;
  uint32_t cnt;
  if constexpr (D == 3) {
    mx = __builtin_fmaxf(__builtin_fmaxf(v[0], v[1]), v[2]);
    mn = __builtin_fminf(__builtin_fminf(v[0], v[1]), v[2]);
    const float md = __builtin_amdgcn_fmed3f(v[0], v[1], v[2]);
    cnt = 2u + (uint32_t)(md == mx) + (uint32_t)(md == mn);
  } else {
    mx = v[0];
    mn = v[0];
#pragma unroll
    for (int k = 1; k < D; ++k) {
      mx = __builtin_fmaxf(mx, v[k]);
      mn = __builtin_fminf(mn, v[k]);
    }
    cnt = 0;
#pragma unroll
    for (int k = 0; k < D; ++k) cnt += (uint32_t)(v[k] == mx) + (uint32_t)(v[k] == mn);
  }
  const float L = mx - mn;
  const bool close = fabsf(L) <= 1e-8f + 1e-5f * fabsf(mn);  // jnp.isclose(max, min)
  // (every term bitwise: with && / || the compiler made each one a branch on the exec mask)
  const bool valid = (OK ? ok : true) & (mine[0] < INFINITY) & (other[0] < INFINITY) & !close;
  const uint32_t khi = __float_as_uint(L), klo = (cnt << 16) | (uint32_t)idx;
  const bool better = valid & ((khi < best.hi) | ((khi == best.hi) & (klo < best.lo)));
  best.hi = better ? khi : best.hi;
  best.lo = better ? klo : best.lo;
  if constexpr (KEEP) {
#pragma unroll
    for (int k = 0; k < D; ++k) best.bd[k] = better ? v[k] : best.bd[k];
  }
}

// two pairs, each against a best of its own: the two compare-and-select sequences interleaved in one block, so that every
// instruction reads a borrow written two instructions (or one and an s_nop) earlier -- no wait per instruction
template <int D>
__device__ __forceinline__ void zeil_pair2(ZeilBest<D>& ba, ZeilBest<D>& bb, const float* mine_a, const float* other_a,
                                           int idx_a, const float* mine_b, const float* other_b, int idx_b) {
  float va[D], vb[D];
  uint32_t ha, la, hb, lb, j;
  LaneMask ca, cb;
  zeil_key<D, false>(mine_a, other_a, true, idx_a, ha, la, va);
  zeil_key<D, false>(mine_b, other_b, true, idx_b, hb, lb, vb);
  asm("v_sub_co_u32_e64 %[j], %[ca], %[la], %[bal]\n\t"
      "v_sub_co_u32_e64 %[j], %[cb], %[lb], %[bbl]\n\t"
      "s_nop 0\n\t"
      "v_subb_co_u32_e64 %[j], %[ca], %[ha], %[bah], %[ca]\n\t"
      "v_subb_co_u32_e64 %[j], %[cb], %[hb], %[bbh], %[cb]\n\t"
      "s_nop 0\n\t"
      "v_cndmask_b32_e64 %[bah], %[bah], %[ha], %[ca]\n\t"
      "v_cndmask_b32_e64 %[bal], %[bal], %[la], %[ca]\n\t"
      "v_cndmask_b32_e64 %[bbh], %[bbh], %[hb], %[cb]\n\t"
      "v_cndmask_b32_e64 %[bbl], %[bbl], %[lb], %[cb]"
      : [bah] "+v"(ba.hi), [bal] "+v"(ba.lo), [bbh] "+v"(bb.hi), [bbl] "+v"(bb.lo), [j] "=&v"(j), [ca] "=&s"(ca),
        [cb] "=&s"(cb)
      : [ha] "v"(ha), [la] "v"(la), [hb] "v"(hb), [lb] "v"(lb));
}

// (plain compares and selects: once per call, and its results travel through DPP next -- instructions the compiler
// spaces itself)
template <int D, bool KEEP>
__device__ __forceinline__ void zeil_merge(ZeilBest<D>& best, const ZeilBest<D>& o) {
  const bool take = (o.hi < best.hi) | ((o.hi == best.hi) & (o.lo < best.lo));
  best.hi = take ? o.hi : best.hi;
  best.lo = take ? o.lo : best.lo;
  if constexpr (KEEP) {
#pragma unroll
    for (int k = 0; k < D; ++k) best.bd[k] = take ? o.bd[k] : best.bd[k];
  }
}

}  // namespace hk
