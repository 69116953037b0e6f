// Four lanes per game, rank-addressed compaction: hk_step at batches that do not fill the device, and (50, 4).
//
// What bounds one hk_step launch of 65 536 (20,3)-games is not the 31 MB it moves but the length of each wave's
// dependent chain between "slab landed" and "slab stored" (hk_duo_kernel.h: every lane of a pair scans all 20 rows,
// finds its live rows with a serial find-first-set chain, and rebuilds the image with a pad fill + scatter) with two
// waves per SIMD to hide it behind.  This kernel cuts the chain and doubles the streams:
//   * a QUAD of lanes owns a game (16 games per wave): twice the waves of the two-lane kernel, each about half as
//     long, four to eight per SIMD;
//   * each lane scans only ITS quarter of the rows (live bit + exactness guard), the quad ORs the bitmasks (two DPP
//     exchanges), and every live row goes -- one LDS write -- to the slot of its RANK among the live rows (popcount of
//     the mask below it) in a compact image, tagged with its original index; lane j then reads ranks j, j+4, ...
//     with compile-time slot indices: no serial bit scan, no gather loop;
//   * shift / reposition / rescale as in the other kernels, column reductions through two DPP exchanges;
//   * the domination test never touches LDS: a lane tests its own slots' triangle, ALL pairs (mine a, b of the lane
//     one up) and the pairs a <= b with the lane two up -- the lane one down and the lane two up do the mirror
//     image, so every pair of the game is visited once (the "two up" diagonal twice, consistently): 2 S^2 tests per
//     lane for S slots; the partner's rows arrive through DPP quad_perm, and what a lane learns about a partner's
//     rows travels back the same way;
//   * the result is written over the ORIGINAL image in place: a row that was live goes back to its own slot (new
//     coordinates, or padding if it was removed); rows that were padding stay untouched -- no fill, no scatter loop.
// Row order (which of two equal rows survives, _jax_ops.py:15-40) is the compact rank = the physical row order.
// The exactness guard and its whole-wave fallback on the generic routines are those of the other kernels.
#pragma once

#include "hk_fast_kernel.h"

namespace hk {

constexpr int kQuad = 4;
constexpr int kQuadGames = kWave / kQuad;

// DPP quad_perm controls: the value of the lane 1 up / 2 up / 3 up (= 1 down) inside the quad
constexpr int kQuadUp1 = 0x39, kQuadUp2 = 0x4E, kQuadUp3 = 0x93, kQuadSwap1 = 0xB1;
template <int CTRL>
__device__ __forceinline__ int qperm_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
template <int CTRL>
__device__ __forceinline__ float qperm(float v) { return __int_as_float(qperm_i<CTRL>(__float_as_int(v))); }

__device__ __forceinline__ float q_min(float v) {
  v = hk_fmin(v, qperm<kQuadSwap1>(v));
  return hk_fmin(v, qperm<kQuadUp2>(v));
}
__device__ __forceinline__ float q_max(float v) {
  v = hk_fmax(v, qperm<kQuadSwap1>(v));
  return hk_fmax(v, qperm<kQuadUp2>(v));
}
__device__ __forceinline__ uint32_t q_or(uint32_t v) {
  v |= (uint32_t)qperm_i<kQuadSwap1>((int)v);
  return v | (uint32_t)qperm_i<kQuadUp2>((int)v);
}
__device__ __forceinline__ int q_sum(int v) {
  v += qperm_i<kQuadSwap1>(v);
  return v + qperm_i<kQuadUp2>(v);
}

template <int M, int D>
struct QuadGeom {
  static constexpr int N = M * D;
  static constexpr int W = (N % 4 == 0) ? 4 : ((N % 2 == 0) ? 2 : 1);  // slab chunk width (floats)
  static constexpr int Q = N / W;                                      // chunks per game
  static constexpr int QL = (kQuadGames * Q + kWave - 1) / kWave;      // slab chunks per lane
  static constexpr int R = (M + kQuad - 1) / kQuad;                    // rows a lane scans = slots per lane
  static constexpr int CW = (D <= 3) ? 4 : D + 1;                      // compact row: D coordinates + original index
  static constexpr int kImage = kQuadGames * N;                        // floats
  static constexpr int kCompact = kQuadGames * M * CW;                 // floats
  // buckets of straight-line bodies (slots per lane)
  static constexpr int next_bucket(int nb) { return nb < 6 ? nb + 1 : (nb < 10 ? nb + 2 : nb + 3); }
};

// ---- slab I/O: 16 consecutive games = one contiguous piece of HBM (contiguous records only: the dispatcher sends
// strided records to the other kernels) ----------------------------------------------------------------------------
template <int M, int D>
struct QuadSlab {
  typename VecOf<QuadGeom<M, D>::W>::type v[QuadGeom<M, D>::QL];
};

template <int M, int D>
__device__ __forceinline__ void quad_slab_issue(QuadSlab<M, D>& r, const float* base, int ngames, int lane) {
  using G = QuadGeom<M, D>;
  using V = typename VecOf<G::W>::type;
  const int total = ngames * G::Q;
#pragma unroll
  for (int it = 0; it < G::QL; ++it) {
    int q = lane + it * kWave;
    q = q < total ? q : total - 1;
    r.v[it] = *reinterpret_cast<const V*>(base + (int64_t)q * G::W);
  }
}

template <int M, int D>
__device__ __forceinline__ void quad_slab_commit(QuadSlab<M, D>& r, float* image, int ngames, int lane) {
  using G = QuadGeom<M, D>;
  using V = typename VecOf<G::W>::type;
  const int total = ngames * G::Q;
#pragma unroll
  for (int it = 0; it < G::QL; ++it) asm volatile("" : "+v"(r.v[it]));
#pragma unroll
  for (int it = 0; it < G::QL; ++it) {
    const int q = lane + it * kWave;
    if (q < total) *reinterpret_cast<V*>(image + q * G::W) = r.v[it];
  }
}

template <int M, int D>
__device__ __forceinline__ void quad_slab_store(const float* image, float* base, int ngames, int lane) {
  using G = QuadGeom<M, D>;
  using V = typename VecOf<G::W>::type;
  const int total = ngames * G::Q;
  V v[G::QL];
#pragma unroll
  for (int it = 0; it < G::QL; ++it) {
    int q = lane + it * kWave;
    q = q < kQuadGames * G::Q ? q : kQuadGames * G::Q - 1;
    v[it] = *reinterpret_cast<const V*>(image + q * G::W);
  }
#pragma unroll
  for (int it = 0; it < G::QL; ++it) asm volatile("" : "+v"(v[it]));
#pragma unroll
  for (int it = 0; it < G::QL; ++it) {
    const int q = lane + it * kWave;
    if (q < total) *reinterpret_cast<V*>(base + (int64_t)q * G::W) = v[it];
  }
}

// LDS traffic of ONE wave is ordered by the hardware; this keeps the compiler from moving a lane's reads above
// the writes of the other lanes they depend on (no s_barrier: waves of a workgroup never share a region)
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- the stages on NB slots per lane --------------------------------------------------------------------------------
template <int R, int D, int NB>
__device__ __forceinline__ void qd_reposition(float (&q)[R * D], unsigned flags) {
  const bool jax_sem = (flags & HK_SEM_MASK) == HK_SEM_JAX;
  float mn[D];
#pragma unroll
  for (int k = 0; k < D; ++k) mn[k] = INFINITY;
#pragma unroll
  for (int r = 0; r < NB; ++r)
#pragma unroll
    for (int k = 0; k < D; ++k) mn[k] = hk_fmin(mn[k], q[r * D + k]);
  float sub[D];
#pragma unroll
  for (int k = 0; k < D; ++k) {
    mn[k] = q_min(mn[k]);
    sub[k] = (mn[k] < INFINITY && (!jax_sem || mn[k] > 0.0f)) ? mn[k] : 0.0f;  // see b_reposition
  }
#pragma unroll
  for (int r = 0; r < NB; ++r)
#pragma unroll
    for (int k = 0; k < D; ++k) q[r * D + k] = q[r * D + k] - sub[k];
}

template <int R, int D, int NB>
__device__ __forceinline__ void qd_rescale(float (&q)[R * D], unsigned flags) {
  const bool jax_sem = (flags & HK_SEM_MASK) == HK_SEM_JAX;
  float mx = -1.0f;
#pragma unroll
  for (int r = 0; r < NB; ++r) {
    const bool live = q[r * D] < INFINITY;
#pragma unroll
    for (int k = 0; k < D; ++k) mx = hk_fmax(mx, live ? q[r * D + k] : -1.0f);
  }
  mx = q_max(mx);
  const bool skip = jax_sem ? (mx <= 1e-8f) : (mx < 0.0f);
  const float div = (skip || mx == 0.0f) ? 1.0f : mx;
#pragma unroll
  for (int r = 0; r < NB; ++r) {
    const bool live = q[r * D] < INFINITY;
#pragma unroll
    for (int k = 0; k < D; ++k) q[r * D + k] = live ? q[r * D + k] / div : INFINITY;
  }
}

// One pair (mine, other) with t = max_k(mine - other), u = min_k(mine - other) (see d_newton in hk_duo_kernel.h):
//   mine earlier:  other removed iff t <= 0;            mine removed iff u >= 0 and t > 0
//   other earlier: other removed iff t <= 0 and u < 0;  mine removed iff u >= 0
// `acc` / `oth` are running minima; <= 0 means removed.  ORDER: +1 mine earlier, -1 other earlier, 0: `late` decides
// at run time (true: the other row is the earlier one).
template <int D, int ORDER>
__device__ __forceinline__ void qd_pair(const float* mine, const float* other, float& acc, float& oth, bool late) {
  float t, u;
  diff_extrema<D>(mine, other, t, u);
  if (ORDER > 0) {
    oth = hk_fmin(oth, t);
    acc = hk_fmin(acc, (t > 0.0f) ? -u : 1.0f);
  } else if (ORDER < 0) {
    oth = hk_fmin(oth, (u < 0.0f) ? t : 1.0f);
    acc = hk_fmin(acc, -u);
  } else {
    oth = hk_fmin(oth, (u < 0.0f || !late) ? t : 1.0f);
    acc = hk_fmin(acc, (t > 0.0f || late) ? -u : 1.0f);
  }
}

// _jax_ops.py:15-73 across the quad.  Rank of (slot s, lane j) = 4 s + j.
//   own slots a < b:                 a earlier
//   lane one up, all (a, b):         my rank 4a + j, theirs 4b + (j+1)%4: mine earlier iff a < b, or a == b and j != 3
//   lane two up, a <= b:             mine earlier iff a < b, or a == b and j < 2
template <int R, int D, int NB>
__device__ __forceinline__ void qd_newton(float (&q)[R * D], int j) {
  float acc[NB], o1[NB], o2[NB];
#pragma unroll
  for (int r = 0; r < NB; ++r) acc[r] = o1[r] = o2[r] = INFINITY;
#pragma unroll
  for (int a = 0; a + 1 < NB; ++a) {
#pragma unroll
    for (int b = a + 1; b < NB; ++b) {
      float t, u;
      diff_extrema<D>(&q[a * D], &q[b * D], t, u);
      acc[b] = hk_fmin(acc[b], t);
      acc[a] = hk_fmin(acc[a], (t > 0.0f) ? -u : 1.0f);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  const bool late1 = j == 3, late2 = j >= 2;
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    float p1[D], p2[D];
#pragma unroll
    for (int k = 0; k < D; ++k) {
      p1[k] = qperm<kQuadUp1>(q[b * D + k]);
      p2[k] = qperm<kQuadUp2>(q[b * D + k]);
    }
#pragma unroll
    for (int a = 0; a < NB; ++a) {
      if (a < b) {
        qd_pair<D, +1>(&q[a * D], p1, acc[a], o1[b], false);
        qd_pair<D, +1>(&q[a * D], p2, acc[a], o2[b], false);
      } else if (a == b) {
        qd_pair<D, 0>(&q[a * D], p1, acc[a], o1[b], late1);
        qd_pair<D, 0>(&q[a * D], p2, acc[a], o2[b], late2);
      } else {
        qd_pair<D, -1>(&q[a * D], p1, acc[a], o1[b], true);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int r = 0; r < NB; ++r) {
    // what the lane one DOWN found out about my slot r (its "one up" is me), and the lane two up
    const float v = hk_fmin(acc[r], hk_fmin(qperm<kQuadUp3>(o1[r]), qperm<kQuadUp2>(o2[r])));
    const bool removed = v <= 0.0f;
#pragma unroll
    for (int k = 0; k < D; ++k) q[r * D + k] = removed ? INFINITY : q[r * D + k];
  }
}

// one transition on slots [0, NB) of the four lanes; returns the GAME's number of live rows
template <int R, int D, int NB>
__device__ __forceinline__ int qd_stages(float (&q)[R * D], const float (&c)[D], int axis, int np, int j,
                                         unsigned flags, unsigned stages) {
  if (stages & HK_STAGE_SHIFT) b_shift<R, D, NB>(q, c, axis, np, flags);
  if (stages & HK_STAGE_REPOSITION) qd_reposition<R, D, NB>(q, flags);
  if (stages & HK_STAGE_NEWTON) qd_newton<R, D, NB>(q, j);
  if (stages & HK_STAGE_RESCALE) qd_rescale<R, D, NB>(q, flags);
  int n = 0;
#pragma unroll
  for (int r = 0; r < NB; ++r) n += (q[r * D] < INFINITY) ? 1 : 0;
  return q_sum(n);
}

template <int M, int D, int NB>
struct QuadStagesFor {
  using G = QuadGeom<M, D>;
  static __device__ __forceinline__ int run(float (&q)[G::R * D], int smax, const float (&c)[D], int axis, int np,
                                            int j, unsigned flags, unsigned stages) {
    if constexpr (NB >= G::R) {
      return qd_stages<G::R, D, G::R>(q, c, axis, np, j, flags, stages);
    } else {
      if (smax <= NB) return qd_stages<G::R, D, NB>(q, c, axis, np, j, flags, stages);
      return QuadStagesFor<M, D, G::next_bucket(NB)>::run(q, smax, c, axis, np, j, flags, stages);
    }
  }
};

// ---- the kernel: single steps with the caller's actions (hk_step) -----------------------------------------------
// HOT: kHotJax = the JAX trainer's take_actions (shift + reposition + Newton polytope, JAX semantics) compiled in.
template <int M, int D, int HOT, int WPB>
__global__ __launch_bounds__(kWave * WPB) void quad_kernel(const float* in0, int64_t in_stride0, int batch0,
                                                           const Params prm) {
  using G = QuadGeom<M, D>;
  constexpr int R = G::R;
  using MaskM = MaskT<M>;
  __shared__ __align__(16) float lds_all[WPB * (G::kImage + G::kCompact)];
  __shared__ float cbuf_all[WPB * kQuadGames * D];  // slow path only
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & (kWave - 1);
  float* image = lds_all + wave * (G::kImage + G::kCompact);
  float* compact = image + G::kImage;
  const int j = lane & 3, gi = lane >> 2;
  const int64_t g0 = ((int64_t)blockIdx.x * WPB + wave) * kQuadGames;
  const int64_t left = (int64_t)batch0 - g0;
  if (left <= 0) return;
  const int ngames = (int)(left < kQuadGames ? left : kQuadGames);
  const bool active = gi < ngames;
  const bool leader = active && j == 0;
  const int64_t g = g0 + gi;
  QuadSlab<M, D> slab;
  quad_slab_issue<M, D>(slab, in0 + g0 * G::N, ngames, lane);
  float* mine = image + gi * G::N;
  const float pad = (float)prm.pad;
  const unsigned flags = (HOT == kHotJax) ? (unsigned)HK_SEM_JAX : prm.flags;
  const unsigned stages = HOT ? (unsigned)(HK_STAGE_SHIFT | HK_STAGE_REPOSITION | HK_STAGE_NEWTON) : prm.stages;
  const float fill = ((flags & HK_SEM_MASK) == HK_SEM_JAX) ? -1.0f : pad;
  float c[D];
  int axis_in = -1;
#pragma unroll
  for (int k = 0; k < D; ++k) c[k] = 0.0f;
  RawActions<D> raw;
  const bool fetch_actions = (stages & HK_STAGE_SHIFT) && active;
  if (fetch_actions) fast_fetch_actions<D>(prm, g, M, raw);
  quad_slab_commit<M, D>(slab, image, ngames, lane);
  if (fetch_actions) fast_decode_actions<D>(prm, raw, c, axis_in);
  wave_lds_fence();

  // ---- live rows + exactness guard: lane j looks at rows j*R .. j*R + R - 1 -----------------------------------------
  float rows[R * D];
  const int i0 = j * R;
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int k = 0; k < D; ++k) rows[r * D + k] = (i0 + r < M) ? mine[(i0 + r) * D + k] : fill;
  const uint32_t fill_bits = __float_as_uint(fill);
  uint32_t lmask = 0, bad = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    uint32_t hi = __float_as_uint(rows[r * D]), lo = hi;
#pragma unroll
    for (int k = 1; k < D; ++k) {
      const uint32_t w = __float_as_uint(rows[r * D + k]);
      hi = w > hi ? w : hi;
      lo = w < lo ? w : lo;
    }
    const bool ge = hi < 0x7F800000u;  // every coordinate in [+0, +inf)
    const bool fl = (lo == fill_bits) && (hi == fill_bits);
    lmask |= (ge && i0 + r < M) ? (1u << r) : 0u;
    bad |= (ge || fl) ? 0u : 1u;
  }
  if (!active) {
    lmask = 0;
    bad = 0;
  }
  // the game's mask of live rows (row i = bit i); ranks of my rows
  MaskM gmask;
  int below;  // live rows of the game below my first row
  if constexpr (M <= 32) {
    gmask = q_or(lmask << i0);
    below = __popc(gmask & ((1u << i0) - 1u));
  } else {
    const unsigned long long mm = (unsigned long long)lmask << i0;
    gmask = ((unsigned long long)q_or((uint32_t)(mm >> 32)) << 32) | q_or((uint32_t)mm);
    below = __popcll(gmask & ((1ull << i0) - 1ull));
  }
  int np = mask_pop(gmask);
  const bool exact = (fill == pad) && !__any(bad != 0);

  if (!exact) {
    // ---- slow path (whole wave): the quad's first lane runs the exact generic routines on the image -----------------
    float* cs = cbuf_all + (wave * kQuadGames + gi) * D;
    if ((stages & HK_STAGE_SHIFT) && leader) load_coords<float>(prm, g, cs);
    np = leader ? num_points<float>(mine, M, D) : 2;
    const bool prev_done = np < 2;
    if (leader) {
      stages_game<float>(mine, prm.m, prm.d, cs, axis_in, pad, stages, flags);
      np = num_points<float>(mine, prm.m, prm.d);
      const bool done = np < 2;
      if (prm.done_out) prm.done_out[g] = done;
      if (prm.prev_done_out) prm.prev_done_out[g] = prev_done;
      if (prm.reward_out) prm.reward_out[g] = prm.reward_sign * (float)(done && !prev_done);
      if (prm.num_points_out) prm.num_points_out[g] = np;
    }
    wave_lds_fence();
    quad_slab_store<M, D>(image, (float*)prm.out + g0 * G::N, ngames, lane);
    return;
  }

  // ---- every live row to the slot of its rank in the compact image, tagged with its original index ---------------
  float* cmine = compact + gi * (M * G::CW);
  {
    int rank = below;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const bool live = (lmask >> r) & 1u;
      if (live) {
        float* dst = cmine + rank * G::CW;
        if constexpr (D <= 3) {
          vf4 v;
          v.x = rows[r * D];
          v.y = D > 1 ? rows[r * D + (D > 1 ? 1 : 0)] : 0.0f;
          v.z = D > 2 ? rows[r * D + (D > 2 ? 2 : 0)] : 0.0f;
          v.w = __int_as_float(i0 + r);
          *reinterpret_cast<vf4*>(dst) = v;
        } else {
#pragma unroll
          for (int k = 0; k < D; ++k) dst[k] = rows[r * D + k];
          dst[D] = __int_as_float(i0 + r);
        }
      }
      rank += live ? 1 : 0;
    }
  }
  // slots per lane in use: the wave-uniform maximum of ceil(np / 4) (a downward search: a few ballots)
  int smax = R;
#pragma nounroll
  while (smax > 1 && !__any(np > kQuad * (smax - 1))) --smax;
  wave_lds_fence();

  // ---- my slots: ranks j, j + 4, ... ----------------------------------------------------------------------------------
  float q[R * D];
  int orig[R];
#pragma unroll
  for (int s = 0; s < R; ++s) {
    orig[s] = 0;
#pragma unroll
    for (int k = 0; k < D; ++k) q[s * D + k] = INFINITY;
  }
  unrolled_while<0, R>([&](auto sc) {
    constexpr int s = decltype(sc)::value;
    if (s >= smax) return false;
    const bool has = kQuad * s + j < np;
    const float* src = cmine + (has ? kQuad * s + j : 0) * G::CW;
    if constexpr (D <= 3) {
      const vf4 v = *reinterpret_cast<const vf4*>(src);
      q[s * D] = has ? v.x : INFINITY;
      if (D > 1) q[s * D + (D > 1 ? 1 : 0)] = has ? v.y : INFINITY;
      if (D > 2) q[s * D + (D > 2 ? 2 : 0)] = has ? v.z : INFINITY;
      orig[s] = has ? __float_as_int(v.w) : -1;
    } else {
#pragma unroll
      for (int k = 0; k < D; ++k) {
        const float v = src[k];
        q[s * D + k] = has ? v : INFINITY;
      }
      const int o = __float_as_int(src[D]);
      orig[s] = has ? o : -1;
    }
    return true;
  });

  // ---- the transition -------------------------------------------------------------------------------------------------
  const bool prev_done = np < 2;
  np = QuadStagesFor<M, D, 1>::run(q, smax, c, axis_in, np, j, flags, stages);
  const bool done = np < 2;
  if (leader) {
    if (prm.done_out) prm.done_out[g] = done;
    if (prm.prev_done_out) prm.prev_done_out[g] = prev_done;
    if (prm.reward_out) prm.reward_out[g] = prm.reward_sign * (float)(done && !prev_done);
    if (prm.num_points_out) prm.num_points_out[g] = np;
  }

  // ---- in place: every row that was live goes back to its slot, new coordinates or padding ---------------------------
  unrolled_while<0, R>([&](auto sc) {
    constexpr int s = decltype(sc)::value;
    if (s >= smax) return false;
    if (orig[s] >= 0) {
      const bool removed = !(q[s * D] < INFINITY);
      float* dst = mine + orig[s] * D;
#pragma unroll
      for (int k = 0; k < D; ++k) dst[k] = removed ? pad : q[s * D + k];
    }
    return true;
  });
  wave_lds_fence();
  quad_slab_store<M, D>(image, (float*)prm.out + g0 * G::N, ngames, lane);
}

// ---- host side -------------------------------------------------------------------------------------------------------
template <int M, int D>
constexpr int quad_waves_per_block() {
  // LDS per wave = image + compact image; keep a workgroup at or below 64 KiB of static LDS
  using G = QuadGeom<M, D>;
  return ((G::kImage + G::kCompact) * 4 * 4 <= 64 * 1024) ? 4 : 1;
}

template <int M, int D>
int launch_quad_t(Params prm, hipStream_t stream) {
  constexpr int WPB = quad_waves_per_block<M, D>();
  const int64_t waves = ((int64_t)prm.batch + kQuadGames - 1) / kQuadGames;
  const unsigned grid = (unsigned)((waves + WPB - 1) / WPB);
  prm.games_per_block = kQuadGames * WPB;
  launch_prepare();
  if (prm.flags == HK_SEM_JAX && prm.stages == (HK_STAGE_SHIFT | HK_STAGE_REPOSITION | HK_STAGE_NEWTON))
    hipLaunchKernelGGL((quad_kernel<M, D, kHotJax, WPB>), dim3(grid), dim3(kWave * WPB), 0, stream,
                       (const float*)prm.in, prm.in_stride, prm.batch, prm);
  else
    hipLaunchKernelGGL((quad_kernel<M, D, kHotNone, WPB>), dim3(grid), dim3(kWave * WPB), 0, stream,
                       (const float*)prm.in, prm.in_stride, prm.batch, prm);
  return launch_status();
}

// (max_points, dim) with a four-lane step kernel
#ifndef HK_QUAD_SPECS
#define HK_QUAD_SPECS(X) X(10, 3) X(20, 3) X(20, 4) X(50, 4)
#endif

// hk_step requests this kernel serves: plain steps (no class / feature outputs), JAX or torch semantics (the sorted
// output of the list semantics stays with the other kernels), float32, contiguous W-aligned records
template <int M, int D>
bool quad_ok_t(const Params& prm) {
  using G = QuadGeom<M, D>;
  const size_t vec_bytes = G::W * 4;
  return prm.in_stride == G::N && prm.out_stride == G::N && reinterpret_cast<uintptr_t>(prm.in) % vec_bytes == 0 &&
         reinterpret_cast<uintptr_t>(prm.out) % vec_bytes == 0;
}

inline bool quad_supported(const Params& prm, int dtype) {
  if (dtype != HK_F32 || prm.mode != kModeStep) return false;
  if (prm.class_out || (prm.stages & kStageFeatureSorts)) return false;
  if (prm.coords_kind == HK_COORDS_IN_RECORD) return false;
  if (prm.flags & (HK_FLAG_FORCE_GENERIC | HK_FLAG_FORCE_TEAM | HK_FLAG_FORCE_ONE_LANE | HK_FLAG_FORCE_TWO_LANES))
    return false;
  if ((prm.stages & HK_STAGE_NEWTON) &&
      ((prm.flags & HK_SEM_MASK) == HK_SEM_LIST || (prm.flags & HK_FLAG_COMPACT_SORTED)))
    return false;
#define HK_X(M_, D_) if (prm.m == M_ && prm.d == D_) return quad_ok_t<M_, D_>(prm);
  HK_QUAD_SPECS(HK_X)
#undef HK_X
  return false;
}

#ifndef HK_SPEC_TU
#define HK_X(M_, D_) extern template int launch_quad_t<M_, D_>(Params, hipStream_t);
HK_QUAD_SPECS(HK_X)
#undef HK_X

inline int launch_quad(const Params& prm, hipStream_t stream) {
#define HK_X(M_, D_) if (prm.m == M_ && prm.d == D_) return launch_quad_t<M_, D_>(prm, stream);
  HK_QUAD_SPECS(HK_X)
#undef HK_X
  return HK_ERR_UNSUPPORTED;
}
#endif

}  // namespace hk
