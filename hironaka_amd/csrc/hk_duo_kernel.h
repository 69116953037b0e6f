// Two lanes per game: steps and rollouts for batches that do not fill the device.
//
// hk::fast_kernel holds a game in ONE lane, so 65 536 games are 1024 instruction streams for the 1024 SIMDs of
// an MI355X, and a lone wave per SIMD issues an instruction only every ~7.7 cycles: the
// vector ALUs idle half of the time and nothing overlaps the load / store bursts.  hk::duo_kernel deals the
// live rows of a game alternately to a PAIR of lanes (compact rank r -> lane r & 1, slot r >> 1; 32 games per
// wave), so the same batch is twice the waves, each with about half the instructions:
//   * shift is per row; reposition / rescale reduce over a lane's own rows and meet the partner's result
//     through one DPP exchange (quad_perm [1,0,3,2]) per coordinate;
//   * the domination test: a lane runs its own rows' triangle, and of the cross pairs (mine a, partner's b)
//     only those with a <= b, in BOTH directions -- the partner does the same from its side, so every cross
//     pair is visited once (the diagonal twice, consistently); what a lane finds out about the partner's rows
//     travels back through one exchange per slot;
//   * everything else (policies, finished-game counts, the exactness guard and its whole-wave fallback on the
//     generic routines, bucketed straight-line bodies, re-gathering when the widest game narrows) follows
//     hk_fast_kernel.h.
// Row order (which of two equal rows survives, _jax_ops.py:15-40) is the compact rank, i.e. the physical row
// order, exactly as in the one-lane kernel.
#pragma once

#include "hk_fast_kernel.h"

namespace hk {

constexpr int kDuoGames = kWave / 2;

// the partner lane's value: DPP quad_perm [1, 0, 3, 2]
__device__ __forceinline__ int duo_other_i(int v) { return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true); }
__device__ __forceinline__ float duo_other(float v) { return __int_as_float(duo_other_i(__float_as_int(v))); }

template <int M, int D>
struct DuoGeom {
  using G = FastGeom<M, D>;
  static constexpr int CH = (G::C + 1) / 2;                       // slots per lane
  static constexpr int QH = (kDuoGames * G::Q + kWave - 1) / kWave;  // slab chunks per lane
};

// ---- slab I/O for 32 games per wave (see hk_fast_kernel.h: all requests in flight before the first use) --
template <int M, int D>
struct DuoSlabRegs {
  typename VecOf<FastGeom<M, D>::W>::type v[DuoGeom<M, D>::QH];
};

template <int M, int D, bool CONTIG>
__device__ __forceinline__ void duo_slab_issue_impl(DuoSlabRegs<M, D>& r, const float* base, int64_t in_stride,
                                                    int ngames, int lane) {
  using G = FastGeom<M, D>;
  using V = typename VecOf<G::W>::type;
  const int total = ngames * G::Q;
#pragma unroll
  for (int it = 0; it < DuoGeom<M, D>::QH; ++it) {
    int q = lane + it * kWave;
    q = q < total ? q : total - 1;
    r.v[it] = *reinterpret_cast<const V*>(base + slab_chunk_global<M, D, CONTIG>(q, in_stride));
  }
}

template <int M, int D>
__device__ __forceinline__ void duo_slab_issue(DuoSlabRegs<M, D>& r, const float* in, int64_t in_stride, int64_t g0,
                                               int ngames, int lane) {
  const float* base = in + g0 * in_stride;
  if (in_stride == FastGeom<M, D>::N) duo_slab_issue_impl<M, D, true>(r, base, in_stride, ngames, lane);
  else duo_slab_issue_impl<M, D, false>(r, base, in_stride, ngames, lane);
}

template <int M, int D>
__device__ __forceinline__ void duo_slab_commit(DuoSlabRegs<M, D>& r, float* lds, int ngames, int lane) {
  using G = FastGeom<M, D>;
  using V = typename VecOf<G::W>::type;
  const int total = ngames * G::Q;
#pragma unroll
  for (int it = 0; it < DuoGeom<M, D>::QH; ++it) asm volatile("" : "+v"(r.v[it]));
#pragma unroll
  for (int it = 0; it < DuoGeom<M, D>::QH; ++it) {
    const int q = lane + it * kWave;
    if (q < total) *reinterpret_cast<V*>(lds + slab_chunk_lds<M, D>(q)) = r.v[it];
  }
}

// kBatch: image chunks read per round (everything at once for the final store; fewer inside the recording loop,
// whose register budget the rows own)
template <int M, int D, bool CONTIG, int kBatch, bool NT = false>
__device__ __forceinline__ void duo_store_slab_impl(const float* lds, float* base, int64_t out_stride, int ngames,
                                                    int lane) {
  using G = FastGeom<M, D>;
  using V = typename VecOf<G::W>::type;
  constexpr int QH = DuoGeom<M, D>::QH;
  const int total = ngames * G::Q;
#pragma unroll
  for (int i0 = 0; i0 < QH; i0 += kBatch) {
    V v[kBatch];
#pragma unroll
    for (int u = 0; u < kBatch; ++u) {  // the image holds kDuoGames games whatever ngames is
      int q = lane + (i0 + u) * kWave;
      q = q < kDuoGames * G::Q ? q : kDuoGames * G::Q - 1;
      v[u] = *reinterpret_cast<const V*>(lds + slab_chunk_lds<M, D>(q));
    }
#pragma unroll
    for (int u = 0; u < kBatch; ++u) asm volatile("" : "+v"(v[u]));
#pragma unroll
    for (int u = 0; u < kBatch; ++u) {
      const int q = lane + (i0 + u) * kWave;
      if (i0 + u < QH && q < total) {
        V* dst = reinterpret_cast<V*>(base + slab_chunk_global<M, D, CONTIG>(q, out_stride));
        if constexpr (NT) __builtin_nontemporal_store(v[u], dst);
        else *dst = v[u];
      }
    }
  }
}

// NT: non-temporal stores -- a rollout's final state is written once and not read again by the launch: kept out of the
// XCD's L2 it leaves the NEXT episode's initial states there (an episode restart re-reads them)
template <int M, int D, int kBatch = DuoGeom<M, D>::QH, bool NT = false>
__device__ inline void duo_store_slab(const float* lds, float* out, int64_t out_stride, int64_t g0, int ngames,
                                      int lane) {
  float* base = out + g0 * out_stride;
  if (out_stride == FastGeom<M, D>::N) duo_store_slab_impl<M, D, true, kBatch, NT>(lds, base, out_stride, ngames, lane);
  else duo_store_slab_impl<M, D, false, kBatch, NT>(lds, base, out_stride, ngames, lane);
}

// ---- image <-> registers ----------------------------------------------------------------------------------
// the set bits of `mask` in ascending order are the game's live rows; lane h takes the ranks h, h + 2, ...
// (SB: compile-time bound of the slots touched -- the staircase of rollout loops below knows it per level)
template <int M, int CH, int D, int SB = CH>
__device__ __forceinline__ void duo_gather(float (&q)[CH * D], const float* mine, uint32_t mask, int smax, int h) {
  unrolled_while<0, SB>([&](auto sc) {
    constexpr int s = decltype(sc)::value;
    if (s >= smax) return false;
    const bool has0 = mask != 0;
    const int b0 = has0 ? mask_first(mask) : 0;
    mask &= mask - 1;
    const bool has1 = mask != 0;
    const int b1 = has1 ? mask_first(mask) : 0;
    mask &= mask - 1;
    const bool has = h ? has1 : has0;
    const float* row = mine + (h ? b1 : b0) * D;
#pragma unroll
    for (int k = 0; k < D; ++k) {
      const float v = row[k];
      q[s * D + k] = has ? v : INFINITY;
    }
    return true;
  });
}

// the lane's live rows back to their slots; returns the mask of the GAME's slots still alive
template <int M, int CH, int D, int SB = CH>
__device__ __forceinline__ uint32_t duo_scatter(const float (&q)[CH * D], float* mine, uint32_t mask, int smax, int h) {
  uint32_t alive = 0;
  unrolled_while<0, SB>([&](auto sc) {
    constexpr int s = decltype(sc)::value;
    if (s >= smax) return false;
    const bool has0 = mask != 0;
    const int b0 = has0 ? mask_first(mask) : 0;
    mask &= mask - 1;
    const bool has1 = mask != 0;
    const int b1 = has1 ? mask_first(mask) : 0;
    mask &= mask - 1;
    const bool has = h ? has1 : has0;
    const int slot = h ? b1 : b0;
    if (has && q[s * D] < INFINITY) {
      alive |= 1u << slot;
      float* row = mine + slot * D;
#pragma unroll
      for (int k = 0; k < D; ++k) row[k] = q[s * D + k];
    }
    return true;
  });
  return alive | (uint32_t)duo_other_i((int)alive);
}

// one pass over the lane's HALF of the game's image (rows h * ceil(M / 2) ...): bitmask of the fully available rows of
// the whole game (the partner's half through DPP) + representability of the lane's half (the caller ANDs over the wave)
template <int M, int D>
__device__ __forceinline__ void duo_scan_half(const float* mine, float fill, int h, uint32_t& live, bool& ok) {
  static_assert(M <= 32, "the game's mask is one word");
  constexpr int C = (M + 1) / 2;
  const int i0 = h * C;
  const float* base = mine + i0 * D;
  uint32_t mask = 0;
  ok = true;
  const uint32_t fill_bits = __float_as_uint(fill);
#pragma unroll
  for (int r = 0; r < C; ++r) {
    // (M odd: the second lane's last row lies past the game -- its read stays inside the wave's image and is not counted)
    const bool valid = (M % 2 == 0) || r + 1 < C || h == 0;
    uint32_t hi = __float_as_uint(base[r * D]), lo = hi;
#pragma unroll
    for (int k = 1; k < D; ++k) {
      const uint32_t u = __float_as_uint(base[r * D + k]);
      hi = u > hi ? u : hi;
      lo = u < lo ? u : lo;
    }
    const bool ge = hi < 0x7F800000u;
    const bool fl = (lo == fill_bits) && (hi == fill_bits);
    ok &= (ge | fl | !valid);
    mask |= (ge && valid) ? (1u << r) : 0u;
  }
  mask <<= i0;
  live = mask | (uint32_t)duo_other_i((int)mask);
}

// ---- the stages on NB slots per lane ----------------------------------------------------------------------
template <int CH, int D, int NB, bool BIN = false>
__device__ __forceinline__ void d_reposition(float (&q)[CH * D], unsigned flags) {
  const bool jax_sem = (flags & HK_SEM_MASK) == HK_SEM_JAX;
  if constexpr (BIN) {
    // Rollouts with in-kernel 0/1 subsets: every coordinate is >= +0 or the +inf of a hole (guarded at entry; sums,
    // differences against the column minimum and quotients keep it so, and x - x is +0), so the float order is the
    // unsigned order of the bit patterns: v_min_u32 with the partner's value as a DPP operand -- the float minimum of
    // values of unknown origin costs a canonicalising v_max x, x per operand (5 instructions per column at one slot
    // per lane, now 1).  A column without live rows subtracts 0; a minimum of 0 subtracts itself (JAX and torch
    // semantics agree there).
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
      const uint32_t o = (uint32_t)duo_other_i((int)mb[k]);
      mb[k] = o < mb[k] ? o : mb[k];
      // (a column without live rows -- an empty game -- subtracts the largest finite float: +inf stays +inf, one
      // v_min_u32 where the test for +inf took a compare and a select)
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
  for (int k = 0; k < D; ++k) {
    mn[k] = hk_fmin(mn[k], duo_other(mn[k]));
    sub[k] = (mn[k] < INFINITY && (!jax_sem || mn[k] > 0.0f)) ? mn[k] : 0.0f;  // see b_reposition
  }
#pragma unroll
  for (int r = 0; r < NB; ++r)
#pragma unroll
    for (int k = 0; k < D; ++k) q[r * D + k] = q[r * D + k] - sub[k];
}

template <int CH, int D, int NB>
__device__ __forceinline__ void d_rescale(float (&q)[CH * D], unsigned flags) {
  const bool jax_sem = (flags & HK_SEM_MASK) == HK_SEM_JAX;
  float mx = -1.0f;
#pragma unroll
  for (int r = 0; r < NB; ++r) {
    const bool live = q[r * D] < INFINITY;
#pragma unroll
    for (int k = 0; k < D; ++k) mx = hk_fmax(mx, live ? q[r * D + k] : -1.0f);
  }
  mx = hk_fmax(mx, duo_other(mx));
  const bool skip = jax_sem ? (mx <= 1e-8f) : (mx < 0.0f);
  const float div = (skip || mx == 0.0f) ? 1.0f : mx;
#pragma unroll
  for (int r = 0; r < NB; ++r) {
    const bool live = q[r * D] < INFINITY;
#pragma unroll
    for (int k = 0; k < D; ++k) q[r * D + k] = live ? q[r * D + k] / div : INFINITY;
  }
}

// _jax_ops.py:15-73 across the pair.  Own rows i < j as in b_newton (rank 2i+h < 2j+h).  Cross pairs (mine a,
// partner's b) with a <= b: my rank 2a+h, the partner's 2b+1-h, so my row is the earlier one unless a == b and
// h == 1.  With t = max_k(mine - other), u = min_k(mine - other):
//   mine earlier:  other removed iff t <= 0;            mine removed iff u >= 0 and t > 0
//   other earlier: other removed iff t <= 0 and u < 0;  mine removed iff u >= 0
// acc[] <= 0 marks my row removed; oth[] <= 0 marks the partner's slot removed (sent back at the end).
template <int CH, int D, int NB>
__device__ __forceinline__ void d_newton(float (&q)[CH * D], int h) {
  // (the accumulators take their FIRST contribution by assignment -- which one that is, is known at compile time --
  // instead of starting at +inf: min(+inf, x) was an instruction per slot, two with the canonicalising v_max)
  float acc[NB], oth[NB];
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
    __builtin_amdgcn_sched_barrier(0);
  }
  const bool late = h != 0;  // on the diagonal the partner's row is the earlier one
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    float o[D];
#pragma unroll
    for (int k = 0; k < D; ++k) o[k] = duo_other(q[b * D + k]);
    {  // diagonal
      float t, u;
      diff_extrema<D>(&q[b * D], o, t, u);
      const float va = (t > 0.0f || late) ? -u : 1.0f;
      acc[b] = (NB == 1) ? va : hk_fmin(acc[b], va);
      oth[b] = (u < 0.0f || !late) ? t : 1.0f;
    }
#pragma unroll
    for (int a = 0; a < b; ++a) {
      float t, u;
      diff_extrema<D>(&q[a * D], o, t, u);
      oth[b] = hk_fmin(oth[b], t);
      acc[a] = hk_fmin(acc[a], (t > 0.0f) ? -u : 1.0f);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int r = 0; r < NB; ++r) {
    // (two compares, not a float minimum of a value that came through DPP: that one is canonicalised first)
    const bool r0 = acc[r] <= 0.0f, r1 = duo_other(oth[r]) <= 0.0f;
    const bool removed = r0 || r1;
#pragma unroll
    for (int k = 0; k < D; ++k) q[r * D + k] = removed ? INFINITY : q[r * D + k];
  }
}

// one transition on slots [0, NB) of both lanes; returns the GAME's number of live rows
template <int CH, int D, int NB, bool BIN = false>
__device__ __forceinline__ int d_stages(float (&q)[CH * D], const float (&c)[D], int axis, int np, int h,
                                        unsigned flags, unsigned stages, uint32_t cmask = 0) {
  if (stages & HK_STAGE_SHIFT) {
    if constexpr (BIN) b_shift_mask<CH, D, NB>(q, cmask, axis, np, flags);
    else b_shift<CH, D, NB>(q, c, axis, np, flags);
  }
  if (stages & HK_STAGE_REPOSITION) d_reposition<CH, D, NB, BIN>(q, flags);
  if (stages & HK_STAGE_NEWTON) d_newton<CH, D, NB>(q, h);
  if (stages & HK_STAGE_RESCALE) d_rescale<CH, D, NB>(q, flags);
  int n = 0;
#pragma unroll
  for (int r = 0; r < NB; ++r) n += (q[r * D] < INFINITY) ? 1 : 0;
  return n + duo_other_i(n);
}

template <int CH, int D, int NB, bool BIN = false>
struct DuoStagesFor {
  static constexpr int kNext = (NB < 6) ? NB + 1 : NB + 2;
  static __device__ __forceinline__ int run(float (&q)[CH * D], int smax, const float (&c)[D], int axis, int np,
                                            int h, unsigned flags, unsigned stages, uint32_t cmask = 0) {
    if constexpr (NB >= CH) {
      return d_stages<CH, D, CH, BIN>(q, c, axis, np, h, flags, stages, cmask);
    } else {
      if (smax <= NB) return d_stages<CH, D, NB, BIN>(q, c, axis, np, h, flags, stages, cmask);
      return DuoStagesFor<CH, D, kNext, BIN>::run(q, smax, c, axis, np, h, flags, stages, cmask);
    }
  }
};

// ---- list semantics for plain rollouts: the pair's rows ranked in descending lexicographic order (coordinate 0
// first; rows are distinct after a Newton stage) and written to their rank, once, at the end of the launch -------------
template <int CH, int D>
__device__ __forceinline__ void duo_ranks_first(const float (&q)[CH * D], int smax, int (&rank)[CH]) {
#pragma unroll
  for (int s = 0; s < CH; ++s) rank[s] = 0;
  unrolled_while<0, CH>([&](auto bc) {
    constexpr int b = decltype(bc)::value;
    if (b >= smax) return false;
    float o[D];
#pragma unroll
    for (int k = 0; k < D; ++k) o[k] = duo_other(q[b * D + k]);
    const bool live_o = o[0] < INFINITY, live_b = q[b * D] < INFINITY;
    unrolled_while<0, CH>([&](auto ac) {
      constexpr int a = decltype(ac)::value;
      if (a >= smax) return false;
      rank[a] += (live_o && key_gt<D, kKeyFirst>(o, &q[a * D])) ? 1 : 0;                      // the partner's row b
      if (a != b) rank[a] += (live_b && key_gt<D, kKeyFirst>(&q[b * D], &q[a * D])) ? 1 : 0;  // my own row b
      return true;
    });
    return true;
  });
}

template <int CH, int D>
__device__ __forceinline__ void duo_scatter_ranked(const float (&q)[CH * D], float* mine, const int (&rank)[CH], int smax) {
  unrolled_while<0, CH>([&](auto sc) {
    constexpr int s = decltype(sc)::value;
    if (s >= smax) return false;
    if (q[s * D] < INFINITY) {
      float* row = mine + rank[s] * D;
#pragma unroll
      for (int k = 0; k < D; ++k) row[k] = q[s * D + k];
    }
    return true;
  });
}

// The buckets of slots per lane (1..6, 8, 10, ...) from the top down: f(NB, LO) for every bucket NB with the next
// smaller one LO (0 below the first)
template <int NB>
struct DuoLevels {
  static constexpr int kLo = (NB <= 6) ? NB - 1 : NB - 2;
  template <typename F>
  static __device__ __forceinline__ void run(F&& f) {
    f(std::integral_constant<int, NB>{}, std::integral_constant<int, kLo>{});
    if constexpr (kLo >= 1) DuoLevels<kLo>::run(f);
  }
};

// The policy stream of a pair: Philox block b (steps 4b .. 4b + 3, hk_common.h policy_words) is computed by the lane
// with (b & 1) == h only -- one Philox per lane per EIGHT steps -- and its word reaches the partner through DPP.
struct DuoPolicyCache {
  U4 r;
  uint32_t pair = 0xFFFFFFFFu;  // wave-uniform: `r` is block 2 * pair + h
};

__device__ inline void duo_policy_words(uint64_t gg, uint32_t step, uint64_t seed, DuoPolicyCache& cache, int h,
                                        uint32_t& host_word, uint32_t& agent_word) {
  const uint32_t block = step >> 2, pair = block >> 1;
  if (cache.pair != pair) {
    cache.r = philox4x32((uint32_t)gg, (uint32_t)(gg >> 32), (pair << 1) | (uint32_t)h, kStreamPolicy, seed);
    cache.pair = pair;
  }
  const uint32_t mine = u4_word(cache.r, step & 3u);
  const bool own = (int)(block & 1u) == h;
  // (the exchange as a statement of its own, pinned by an empty asm: written inside the select the compiler ran the
  // DPP move under the select's exec mask -- the partner lane inactive, its value read as zero)
  uint32_t theirs = (uint32_t)duo_other_i((int)mine);
  asm volatile("" : "+v"(theirs));
  const uint32_t w = own ? mine : theirs;
  host_word = w & 0xFFFF0000u;
  agent_word = w << 16;
}

// ---- Zeillinger's host on the pair's rows (jax/players.py:55-109; hk_fast_rows.h: c_zeillinger is the one-lane twin) --
// Over the pairs i < j of live rows (row-major order = rank order: the deals keep it): the characteristic vector
// (L, S) = (max - min, #max + #min) of P_i - P_j, differences that are constant (jnp.isclose) do not count, the first
// minimum wins, the subset is {argmin, argmax} of that difference; no pair: class 0.  Rank of (slot s, lane h) = 2 s + h.
// A lane takes its own slots a < b and, against the partner's rows (DPP), the slots b >= a (lane 1: b > a) -- every
// pair once, "mine" always the earlier row: NB^2 tests per lane where the one-lane kernel runs 2 NB^2 on half the
// waves.  "First" is explicit -- the pair's index 64 i + j breaks ties --, so the two lanes' bests merge by the same
// comparison.
// Buckets of more than kDuoZeilDpp slots per lane: the same pairs as a ROLLED loop over the game's rows parked by
// rank in its image (scratch between two deals) -- unrolled, 64 pair tests on 8 slots per lane spilled 692 B per lane
// and a first step took 51 us; lane h takes the rows i = h, h + 2, ... against every later row.  (Measured per
// 20-step episode of 65 536 games: DPP up to 4 slots 47.5 us, up to 6 slots 44.3 us; 242 VGPRs, no scratch.)
constexpr int kDuoZeilDpp = 6;

template <int CH, int D, int NB>
__device__ __forceinline__ int duo_zeillinger(const float (&q)[CH * D], int h, float* mine, int smax) {
  constexpr bool KEEP = NB <= kDuoZeilDpp;
  ZeilBest<D> best;
#pragma unroll
  for (int k = 0; k < D; ++k) best.bd[k] = 0.0f;
  if constexpr (KEEP) {
    float p[NB * D];
#pragma unroll
    for (int e = 0; e < NB * D; ++e) p[e] = duo_other(q[e]);
#pragma unroll
    for (int a = 0; a < NB; ++a) {
#pragma unroll
      for (int b = a + 1; b < NB; ++b)
        zeil_pair_cmp<D, true, false>(best, &q[a * D], &q[b * D], true, (128 * a + 2 * b) + 65 * h);
#pragma unroll
      for (int b = a; b < NB; ++b)
        zeil_pair_cmp<D, true>(best, &q[a * D], &p[b * D], (b > a) | (h == 0), (128 * a + 2 * b + 1) + 63 * h);
      if constexpr (NB > 3) __builtin_amdgcn_sched_barrier(0);
    }
  } else {
    __syncthreads();  // (a workgroup is one wave)
#pragma unroll
    for (int s = 0; s < NB; ++s)
#pragma unroll
      for (int k = 0; k < D; ++k) mine[(2 * s + h) * D + k] = q[s * D + k];
    __syncthreads();
    const int n = 2 * smax;  // ranks in use (wave-uniform); holes are +inf
    // two pairs per pass with a best of their own each (two independent chains; the order of the merges does not
    // matter: the pair index is part of the key)
    ZeilBest<D> second;
#pragma nounroll
    for (int i = h; i + 1 < n; i += 2) {
      float pi[D];
#pragma unroll
      for (int k = 0; k < D; ++k) pi[k] = mine[i * D + k];
#pragma nounroll
      for (int j = i + 1; j < n; j += 2) {
        float pj[D], pk[D];
        const int j2 = (j + 1 < n) ? j + 1 : j;
#pragma unroll
        for (int k = 0; k < D; ++k) {
          pj[k] = mine[j * D + k];
          pk[k] = mine[j2 * D + k];
        }
        zeil_pair2<D>(best, second, pi, pj, 64 * i + j, pi, pk, 64 * i + j + 1);  // (past the end: the last pair again, later)
      }
    }
    zeil_merge<D, false>(best, second);
  }
  {  // the partner's best
    ZeilBest<D> o;
    o.hi = (uint32_t)duo_other_i((int)best.hi);
    o.lo = (uint32_t)duo_other_i((int)best.lo);
    if constexpr (KEEP) {
#pragma unroll
      for (int k = 0; k < D; ++k) o.bd[k] = duo_other(best.bd[k]);
    }
    zeil_merge<D, KEEP>(best, o);
  }
  const bool have = best.have();
  if constexpr (!KEEP) {  // the chosen pair's difference from the parked rows
    const int i = have ? (int)((best.lo & 0xFFFFu) >> 6) : 0, j = have ? (int)(best.lo & 63u) : 0;
#pragma unroll
    for (int k = 0; k < D; ++k) best.bd[k] = mine[i * D + k] - mine[j * D + k];
    __syncthreads();  // (the image is written again by the next deal)
  }
  int lo = 0, hi = 0;
  float vlo = best.bd[0], vhi = best.bd[0];
#pragma unroll
  for (int k = 1; k < D; ++k) {
    if (best.bd[k] < vlo) { vlo = best.bd[k]; lo = k; }
    if (best.bd[k] > vhi) { vhi = best.bd[k]; hi = k; }
  }
  if (!have || lo == hi) return 0;
  return encode_mask((1u << lo) | (1u << hi));
}

// ---- the policy stream off the critical path (plain rollouts) ------------------------------------------------------
// A Philox block is ~115 instructions, 20 of them quarter-rate 32 x 32 -> 64 multiplies in a dependent chain of ten
// rounds: computed inside the step loop (one block per lane every four steps) it was a fifth of the loop's
// instructions and most of a late step's latency (a step on one slot per lane is ~40 instructions without it).  The
// words depend on (game, step) only, not on the state: a WINDOW of kDuoPreBlocks blocks per game (24 steps: four per block) is
// computed before the first step -- while the wave would otherwise only wait for its slab -- two independent chains at
// a time, DECODED (subset mask, axis: the decode was another ~15 instructions of every step) and parked in LDS, one byte
// per game and step; a step reads its byte (both lanes of a pair the same address).  Episodes longer than a window
// refill it between two passes over the staircase.
constexpr int kDuoPreBlocks = 6;  // (a block serves four steps)

// blocks [wb0, wb0 + nb) of the wave's games: lane (gi, h) computes blocks wb0 + h, wb0 + h + 2, ... and stores the
// DECODED actions of their steps, one byte per game and step: subset mask (D bits) | axis << 5
template <int D>
__device__ __forceinline__ void duo_policy_fill(uint8_t* act, uint64_t gg, uint32_t wb0, int nb, uint64_t seed,
                                                int host_policy, int agent_policy, int gi, int h) {
  static_assert(D <= 5 && D <= kPolicyShortDim, "an action travels as a byte; four steps per Philox block");
  auto put = [&](const U4& r, int b) {  // the four steps of block b
    int cls, axis;
    uint32_t mask;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t w = u4_word(r, k);
      policy_from_words<D>(w & 0xFFFF0000u, w << 16, host_policy, agent_policy, cls, axis, mask, 0);
      act[(4 * b + k) * kDuoGames + gi] = (uint8_t)(mask | ((uint32_t)axis << 5));
    }
  };
#pragma nounroll
  for (int i = 0; i < kDuoPreBlocks; i += 4) {
    if (i >= nb) break;  // wave-uniform
    const int b0 = i + h, b1 = b0 + 2;
    if (i + 2 < nb) {  // (wave-uniform) two independent chains
      const U4 r0 = philox4x32((uint32_t)gg, (uint32_t)(gg >> 32), wb0 + (uint32_t)b0, kStreamPolicy, seed);
      const U4 r1 = philox4x32((uint32_t)gg, (uint32_t)(gg >> 32), wb0 + (uint32_t)b1, kStreamPolicy, seed);
      put(r0, b0);
      if (b1 < kDuoPreBlocks) put(r1, b1);
    } else {  // the window's last one or two blocks
      const U4 r0 = philox4x32((uint32_t)gg, (uint32_t)(gg >> 32), wb0 + (uint32_t)b0, kStreamPolicy, seed);
      if (b0 < kDuoPreBlocks) put(r0, b0);
    }
  }
}

// ---- the kernel: fused rollouts (MODE kModeRollout; kModeRolloutRec: with per-step observations / records) and
// single steps with the caller's actions (kModeStep: hk_step) -------------------------------------------------
// ACTS (plain rollouts): the kernel also writes the small per-step records (flush_records below) -- its own
// instantiation, the headline kernel carries none of it (measured: +0.4 us per 65 536-game episode as a run-time branch)
// ZEIL (plain rollouts): Zeillinger's host -- its choice depends on the state, so the policies stay inside the loop
// (duo_zeillinger on the rows in registers + the pair's Philox words), no action window; its own instantiation too.
template <int M, int D, int MODE, int HOT = kHotNone, bool ACTS = false, bool ZEIL = false>
__global__ __launch_bounds__(kWave, 2) void duo_kernel(const float* in0, int64_t in_stride0, int batch0,
                                                       const Params prm) {
  static_assert(!ACTS || MODE == kModeRollout, "the small records ride on the plain rollout");
  static_assert(!ZEIL || (MODE == kModeRollout && HOT == kHotNone && !ACTS), "Zeillinger's host: plain rollouts");
  constexpr bool kRec = MODE == kModeRolloutRec;
  constexpr bool kRoll = MODE == kModeRollout || kRec;
  using G = FastGeom<M, D>;
  constexpr int CH = DuoGeom<M, D>::CH;
  static_assert(M <= 32, "the live mask of a game travels as 32 bits");
  __shared__ __align__(16) float lds[kDuoGames * G::S];
  __shared__ float cbuf[kDuoGames * D];  // slow path only
  // plain rollouts: the policy words of a window of steps (duo_policy_fill)
  // (one row more than the window: the step loop requests the NEXT step's byte while it works on this one)
  __shared__ __align__(16) uint8_t pol[(MODE == kModeRollout) ? (4 * kDuoPreBlocks + 1) * kDuoGames : 16];
  const int lane = threadIdx.x;
  const int h = lane & 1, gi = lane >> 1;
  const int64_t g0 = (int64_t)blockIdx.x * kDuoGames;
  const int64_t left = (int64_t)batch0 - g0;
  const int ngames = (int)(left < kDuoGames ? left : kDuoGames);
  const bool active = gi < ngames;
  const bool leader = active && h == 0;
  const int64_t g = g0 + gi;
#ifdef HK_DUO_PROBE  // dev builds (scripts/build_probe.sh): a time line per wave, written over game_length_out
  __shared__ int32_t probe_buf[24];
  if (lane < 24) probe_buf[lane] = 0;
  const long long probe_t0 = wall_clock64();
  long long probe_t1 = 0, probe_t2 = 0;
  int probe_steps = 0, probe_smax = 0;
#endif
  // (the game id's load goes out BEFORE the slab's: loads return in order, and the action window -- filled while the
  // slab is in flight -- needs the id; behind the slab requests it would arrive with them: +0.9 us per episode;
  // the slab requests follow it at once; nothing touches the id before they are out)
  const bool has_ids = kRoll && prm.game_ids != nullptr;
  uint32_t raw_id = 0;  // (positions: non-negative; zero-extended below -- a sign extension would wait for the load here)
  if (has_ids && active) raw_id = (uint32_t)prm.game_ids[g];
  __builtin_amdgcn_sched_barrier(0);
  DuoSlabRegs<M, D> slab;
  duo_slab_issue<M, D>(slab, in0, in_stride0, g0, ngames, lane);
  __builtin_amdgcn_sched_barrier(0);
  const uint64_t gg = prm.game_offset + (has_ids ? (uint64_t)raw_id : (uint64_t)g);
  // plain rollouts: the first window of policy words, computed while the slab is in flight
  uint32_t pol_b0 = prm.step_offset >> 2;  // first block of the window (wave-uniform)
  const uint32_t pol_last = (prm.steps > 0) ? (prm.step_offset + (uint32_t)prm.steps - 1u) >> 2 : pol_b0;
  if constexpr (MODE == kModeRollout && !ZEIL) {
    const uint32_t nb = pol_last - pol_b0 + 1u;
    duo_policy_fill<D>(pol, gg, pol_b0, (int)(nb < (uint32_t)kDuoPreBlocks ? nb : (uint32_t)kDuoPreBlocks), prm.seed,
                       HOT ? (int)HK_HOST_RANDOM : prm.host_policy,
                       (HOT == kHotJax) ? (int)HK_AGENT_RANDOM
                                        : (HOT == kHotTorch) ? (int)HK_AGENT_RANDOM_LEGAL : prm.agent_policy, gi, h);
  }
#ifdef HK_DUO_PROBE
  if (lane == 0) probe_buf[21] = (int32_t)wall_clock64();  // the action window is filled
#endif
  float* mine = lds + gi * G::S;
  const float pad = (float)prm.pad;
  const unsigned flags = (HOT == kHotJax) ? (unsigned)HK_SEM_JAX : (HOT == kHotTorch) ? kHotTorchFlags : prm.flags;
  const unsigned stages = HOT ? (unsigned)(HK_STAGE_SHIFT | HK_STAGE_REPOSITION | HK_STAGE_NEWTON) : prm.stages;
  const float fill = ((flags & HK_SEM_MASK) == HK_SEM_JAX) ? -1.0f : pad;
  const int nsteps = kRoll ? prm.steps : 1;
  // plain rollouts under list semantics sort once, at the end (see hk_fast_kernel.h)
  constexpr bool kEndSort = MODE == kModeRollout && HOT == kHotNone;
  const bool end_sort = kEndSort && nsteps > 0 && (stages & HK_STAGE_NEWTON) &&
                        ((flags & HK_SEM_MASK) == HK_SEM_LIST || (flags & HK_FLAG_COMPACT_SORTED));
  bool rescale_pending = false;
  PolicyCache pcache;      // slow path (one lane per game computes)
  DuoPolicyCache dcache;
  float c[D];
  int axis_in = -1;
#pragma unroll
  for (int k = 0; k < D; ++k) c[k] = 0.0f;
  // step mode: the action loads join the slab's requests in flight (both lanes of a pair fetch them)
  RawActions<D> raw;
  const bool fetch_actions = !kRoll && (stages & HK_STAGE_SHIFT) && active;
  if (fetch_actions) fast_fetch_actions<D>(prm, g, M, raw);
  duo_slab_commit<M, D>(slab, lds, ngames, lane);
  if (fetch_actions) fast_decode_actions<D>(prm, raw, c, axis_in);
  __syncthreads();
#ifdef HK_DUO_PROBE
  if (lane == 0) probe_buf[22] = (int32_t)wall_clock64();  // the slab is in LDS
#endif

  // ---- live rows, exactness guard: each lane of a pair scans its half of the game's rows, the masks meet through DPP
  // (rounds 2-4 had both lanes scan the whole game: 1.3 us of a wave's 15.7, scripts/probe_timeline.py) -----------
  uint32_t gmask;
  bool ok;
  duo_scan_half<M, D>(mine, fill, h, gmask, ok);
  if (!active) {
    gmask = 0;
    ok = true;
  }
  int np = mask_pop(gmask);
  int nmax = wave_max(np, M);
  const bool exact = (fill == pad) && __all(ok) && nmax <= G::C;

  if (!exact) {
    // ---- slow path (whole wave): the pair's first lane runs the exact generic routines on the image ----------
    float* cs = cbuf + gi * D;
    np = leader ? num_points<float>(mine, M, D) : 2;
    int length = (np < 2) ? 0 : -1;
    if (kRoll && prm.count_ws) {
      const unsigned long long b0 = __ballot(leader && np < 2);
      if (lane == 0) count_add(prm.count_ws + blockIdx.x, (uint32_t)__popcll(b0));
    }
    for (int t = 0; t < nsteps; ++t) {
      int axis = axis_in, cls = 0;
      if (kRec && prm.obs_out) {
        __syncthreads();
        duo_store_slab<M, D, 2>(lds, (float*)prm.obs_out + (int64_t)t * prm.batch * G::N, (int64_t)G::N, g0, ngames, lane);
        __syncthreads();
      }
      if (kRoll) {
        uint32_t mask;
        const int zc = (prm.host_policy == HK_HOST_ZEILLINGER && leader) ? zeillinger_game<float>(mine, prm.m, prm.d) : 0;
        fast_policy<D>(prm, gg, prm.step_offset + (uint32_t)t, pcache, cls, axis, mask, zc);
        if (leader)
          for (int k = 0; k < prm.d; ++k) cs[k] = (float)((mask >> k) & 1u);
      } else if ((stages & HK_STAGE_SHIFT) && leader) {
        load_coords<float>(prm, g, cs);
      }
      const bool prev_done = np < 2;
      if (leader) {
        stages_game<float>(mine, prm.m, prm.d, cs, axis, pad, stages, flags);
        np = num_points<float>(mine, prm.m, prm.d);
      }
      const bool done = np < 2;
      if (done && length < 0) length = t + 1;
      if (kRoll && prm.count_ws) {
        const unsigned long long bd = __ballot(leader && done);
        if (lane == 0) count_add(prm.count_ws + (size_t)(t + 1) * prm.count_stride + blockIdx.x, (uint32_t)__popcll(bd));
      }
      if ((kRec || ACTS) && leader) {  // (ACTS: plain rollouts with the small records, flush_records below)
        const int64_t at = (int64_t)t * prm.batch + g;
        if (prm.r_host_class_out) prm.r_host_class_out[at] = cls;
        if (prm.r_axis_out) prm.r_axis_out[at] = axis;
        if (prm.r_done_out) prm.r_done_out[at] = done;
        if (prm.r_reward_out) prm.r_reward_out[at] = prm.reward_sign * (float)(done && !prev_done);
      }
      if (!kRoll && leader) {
        if (prm.done_out) prm.done_out[g] = done;
        if (prm.prev_done_out) prm.prev_done_out[g] = prev_done;
        if (prm.reward_out) prm.reward_out[g] = prm.reward_sign * (float)(done && !prev_done);
        if (prm.num_points_out) prm.num_points_out[g] = np;
      }
    }
    if (kRoll && leader && prm.game_length_out) prm.game_length_out[g] = length;
    __syncthreads();
    duo_store_slab<M, D>(lds, (float*)prm.out, prm.out_stride, g0, ngames, lane);
    return;
  }

  // ---- the pair's rows ----------------------------------------------------------------------------------
  int smax = (nmax + 1) >> 1;
#ifdef HK_DUO_PROBE
  probe_t1 = wall_clock64();
  probe_smax = smax;
#endif
  float q[CH * D];
#pragma unroll
  for (int e = 0; e < CH * D; ++e) q[e] = INFINITY;
  duo_gather<M, CH, D>(q, mine, gmask, smax, h);
#ifdef HK_DUO_PROBE
  {
    float sink = 0.0f;
#pragma unroll
    for (int e = 0; e < CH * D; ++e) sink += (q[e] < INFINITY) ? q[e] : 0.0f;
    asm volatile("" : "+v"(sink));
    if (lane == 0) probe_buf[23] = (int32_t)wall_clock64();  // the rows are in registers
  }
#endif
  if (!active) np = 2;
  int length = (np < 2) ? 0 : -1;
  if (kRoll && prm.count_ws) {
    const unsigned long long b0 = __ballot(leader && np < 2);
    if (lane == 0) count_add(prm.count_ws + blockIdx.x, (uint32_t)__popcll(b0));
  }
  uint32_t* count_slot = (kRoll && prm.count_ws) ? prm.count_ws + blockIdx.x : nullptr;
  uint32_t count_stride = prm.count_stride;
  uint32_t step0 = prm.step_offset;
  uint64_t seed = prm.seed;
  int host_policy = HOT ? (int)HK_HOST_RANDOM : prm.host_policy;
  int agent_policy = (HOT == kHotJax) ? (int)HK_AGENT_RANDOM
                                      : (HOT == kHotTorch) ? (int)HK_AGENT_RANDOM_LEGAL : prm.agent_policy;
  if (HOT)
    asm volatile("" : "+s"(count_slot), "+s"(count_stride), "+s"(step0), "+s"(seed));
  else
    asm volatile("" : "+s"(count_slot), "+s"(count_stride), "+s"(step0), "+s"(seed), "+s"(host_policy),
                 "+s"(agent_policy));
  // Plain rollouts write the small records (host class, axis, done, reward) too, when asked: they are a function of the
  // window's action bytes and of the game's first finished step, so no step stores them -- they go out per window
  // (before it moves, and after the last step), lane = (step within a pair, game of the wave): 128-B runs per step and
  // field (hk_quadroll_kernel.h does the same on 16 games).  Observations stay with kModeRolloutRec / the four-lane kernel.
  static_assert(kDuoGames == 32 && kWave == 64, "flush_records: two steps of the wave's games per instruction");
  int32_t* rec_cls = ACTS ? prm.r_host_class_out : nullptr;
  int32_t* rec_axis = ACTS ? prm.r_axis_out : nullptr;
  uint8_t* rec_done = ACTS ? prm.r_done_out : nullptr;
  float* rec_reward = ACTS ? prm.r_reward_out : nullptr;
  int64_t rec_batch = prm.batch;
  float rec_sign = prm.reward_sign;
  if constexpr (ACTS)  // (pinned: left to the compiler the stores' address arithmetic ends in an illegal VGPR -> SGPR copy)
    asm volatile("" : "+s"(rec_cls), "+s"(rec_axis), "+s"(rec_done), "+s"(rec_reward), "+s"(rec_batch), "+s"(rec_sign));
  constexpr bool want_small = ACTS;
  int t_rec = 0;  // the records of steps < t_rec are written
  auto flush_records = [&](int t_end) {  // (t_end: wave-uniform)
    if (t_rec < t_end) {
      const int gsel = lane & (kDuoGames - 1), sub = lane >> 5;
      const int lg = __shfl(length, gsel * 2);  // the game's first finished step (0: from the start, -1: not yet)
      const bool gok = gsel < ngames;
      const int64_t off = (int64_t)sub * rec_batch + gsel;
      const uint8_t* w = pol + (int)(step0 + (uint32_t)t_rec - (pol_b0 << 2)) * kDuoGames + lane;
      for (int ta = t_rec; ta < t_end; ta += 2, w += 2 * kDuoGames) {
        const int tt = ta + sub;
        if (gok && tt < t_end) {
          const uint32_t a = *w;
          const int64_t row = (int64_t)ta * rec_batch + g0;  // (scalar)
          if (rec_cls) (rec_cls + row)[off] = encode_mask(a & 31u);
          if (rec_axis) (rec_axis + row)[off] = (int32_t)(a >> 5);
          if (rec_done) (rec_done + row)[off] = lg >= 0 && tt + 1 >= lg;
          if (rec_reward) (rec_reward + row)[off] = rec_sign * (float)(lg >= 1 && tt + 1 == lg);
        }
      }
    }
    t_rec = t_end;
  };
  if constexpr (MODE == kModeRollout) {
    // ---- plain rollouts: a STAIRCASE of loops, one per bucket of slots per lane, entered from the top down.  smax
    // never grows, so the wave walks down the stairs once; each loop is straight-line for its own bucket -- no
    // per-step dispatch over the buckets, and only the slots the bucket covers are loop-carried registers (a single
    // loop over all buckets moves the whole row array at every back edge: 16 v_mov_b64 per step at (20,3)). -------
    static_assert(CH <= 6 || CH % 2 == 0, "bucket ladder: 1..6, then even numbers");
    int t = 0;
    bool stop = false;
    while (t < nsteps && !stop) {  // one pass per window of policy words (episodes of up to 24 steps: one pass)
    if (!ZEIL && (uint32_t)((step0 + (uint32_t)t) >> 2) - pol_b0 >= (uint32_t)kDuoPreBlocks) {
      if constexpr (want_small) flush_records(__builtin_amdgcn_readfirstlane(t));
      __syncthreads();
      pol_b0 = (step0 + (uint32_t)t) >> 2;
      const uint32_t nb = pol_last - pol_b0 + 1u;
      duo_policy_fill<D>(pol, gg, pol_b0, (int)(nb < (uint32_t)kDuoPreBlocks ? nb : (uint32_t)kDuoPreBlocks), seed,
                         host_policy, agent_policy, gi, h);
      __syncthreads();
    }
    // last step (exclusive) the window covers
    const uint32_t wend_abs = (pol_b0 + (uint32_t)kDuoPreBlocks) << 2;
    const int tw = (!ZEIL && wend_abs - step0 < (uint32_t)nsteps) ? (int)(wend_abs - step0) : nsteps;
    // the action byte of step t, requested one step ahead: the LDS round trip (it opened every step: ds_read, wait)
    // runs behind the previous step's arithmetic
    uint32_t a_next = ZEIL ? 0u : pol[(int)(step0 + (uint32_t)t - (pol_b0 << 2)) * kDuoGames + gi];
    DuoLevels<CH>::run([&](auto nbc, auto loc) {
      constexpr int NB = decltype(nbc)::value, LO = decltype(loc)::value;
#ifndef HK_NO_SETPRIO
      // The launch ends with its slowest waves -- the ones that start in a wide bucket or run many steps -- and the two
      // waves of a SIMD share its issue slots by priority, then age: a wave in a wide bucket outranks its neighbour
      // (who is ahead anyway), and so does one that is still running late in the episode.
      if ((smax > LO || LO == 0) && t < tw && !stop) {
        if (t >= 10) __builtin_amdgcn_s_setprio(3);
        else if constexpr (NB >= 6) __builtin_amdgcn_s_setprio(3);
        else if constexpr (NB >= 4) __builtin_amdgcn_s_setprio(2);
        else if constexpr (NB >= 2) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
      }
#endif
      while (t < tw && (smax > LO || LO == 0) && !stop) {  // (a wave of empty games has smax 0: the last loop's)
#ifndef HK_NO_SETPRIO
        if constexpr (NB == 1) {
          if (t == 10) __builtin_amdgcn_s_setprio(3);
        }
#endif
        uint32_t mask;
        int axis;
        if constexpr (ZEIL) {
          // (a game with fewer than two rows has no pair: class 0 -- a wave of finished games skips the test)
          static_assert(NB <= kDuoZeilDpp || 2 * NB <= M, "the rolled pair loop parks ranks 0 .. 2 NB - 1 in the game's M rows");
          const int zc = __any(active && np >= 2) ? duo_zeillinger<CH, D, NB>(q, h, mine, smax) : 0;
          uint32_t ra, rb;
          int cls;
          duo_policy_words(gg, step0 + (uint32_t)t, seed, dcache, h, ra, rb);
          policy_from_words<D>(ra, rb, host_policy, agent_policy, cls, axis, mask, zc);
        } else {
          const uint32_t a = a_next;
          a_next = pol[(int)(step0 + (uint32_t)t + 1u - (pol_b0 << 2)) * kDuoGames + gi];  // (past the window: not used)
          mask = a & 31u;
          axis = (int)(a >> 5);
        }
        const unsigned st = (end_sort && t + 1 == nsteps) ? (stages & ~(unsigned)HK_STAGE_RESCALE) : stages;
        rescale_pending = end_sort && t + 1 == nsteps && (stages & HK_STAGE_RESCALE);
        np = d_stages<CH, D, NB, true>(q, c, axis, np, h, flags, st, mask);
#ifdef HK_DUO_PROBE
        if (lane == 0 && t == 0) probe_buf[20] = (int32_t)wall_clock64();  // step 0: the stages are done
#endif
        if (!active) np = 2;
        const bool done = np < 2;
        if (done && length < 0) length = t + 1;
        // (the finished-game counts: ballots over the games' first finished steps after the loop -- a finished game
        // stays finished --, not an atomic per step in here)
        if constexpr (NB == 1) {
          // Fixed point: a game that is down to ONE point sitting at the origin (or to none) does not change any
          // more -- whatever the subset and the axis: the shift adds zeros, reposition / rescale find nothing to move,
          // the Newton stage has nothing to compare.  Once every game of the wave is there (a game reaches it one
          // step after it ends when reposition is on; the mean game lasts 5 steps, the longest of 32 about 13), the
          // rest of the episode changes nothing.
          if (t + 1 < nsteps && !__any(active && !done)) {
            bool still = true;  // (one slot per lane: it holds the game's point, a hole, or nothing)
#pragma unroll
            for (int k = 0; k < D; ++k) still &= (q[k] == 0.0f);
            still |= !(q[0] < INFINITY);
            if (!__any(active && !still)) stop = true;
          }
        } else {
          // re-deal the rows when the widest game of the wave fits fewer slots per lane
          if (t + 1 < nsteps && !__any(active && ((np + 1) >> 1) >= smax)) {
            __syncthreads();
            gmask = duo_scatter<M, CH, D, NB>(q, mine, gmask, smax, h);
            __syncthreads();
            const int sprev = smax;
            nmax = wave_max(active ? np : 0, 2 * smax - 2);
            smax = (nmax + 1) >> 1;
            duo_gather<M, CH, D, NB>(q, mine, gmask, sprev, h);  // slots [smax, sprev) become holes again
          }
        }
#ifdef HK_DUO_PROBE  // per-step stamps: the clock and the slots per lane after the step
        if (lane == 0 && t < 24) probe_buf[t] = (int32_t)((((uint32_t)wall_clock64()) << 4) | (uint32_t)(smax & 15));
#endif
        ++t;
      }
    });
    }
#ifdef HK_DUO_PROBE
    probe_steps = t;
    probe_t2 = wall_clock64();
#endif
    // games whose first finished step is <= s, for every step s >= 1 (s = 0 was counted at entry)
    if (count_slot) add_length_counts(count_slot, count_stride, 1, nsteps, leader, length, lane);
    if constexpr (want_small) {  // window by window up to the last step (the loop may have left at a fixed point)
      for (;;) {
        const uint32_t wend = ((pol_b0 + (uint32_t)kDuoPreBlocks) << 2) - step0;  // steps (from 0) the window reaches
        const int te = __builtin_amdgcn_readfirstlane((wend < (uint32_t)nsteps) ? (int)wend : nsteps);
        flush_records(te);
        if (te >= nsteps) break;
        __syncthreads();
        pol_b0 = (step0 + (uint32_t)te) >> 2;
        const uint32_t nb = pol_last - pol_b0 + 1u;
        duo_policy_fill<D>(pol, gg, pol_b0, (int)(nb < (uint32_t)kDuoPreBlocks ? nb : (uint32_t)kDuoPreBlocks), seed,
                           host_policy, agent_policy, gi, h);
        __syncthreads();
      }
    }
  }
  const bool want_obs = kRec && prm.obs_out != nullptr;
  const bool want_records = kRec && (prm.r_host_class_out || prm.r_axis_out || prm.r_done_out || prm.r_reward_out);
  for (int t = 0; MODE != kModeRollout && t < nsteps; ++t) {  // single steps and recording rollouts
    int axis = axis_in, cls = 0;
    if (want_obs) {  // state before the step: rebuild the image, store it coalesced
      __syncthreads();
      if (h == 0) fill_image<M, D>(mine, pad);
      __syncthreads();
      duo_scatter<M, CH, D>(q, mine, gmask, smax, h);
      __syncthreads();
      duo_store_slab<M, D, 2>(lds, (float*)prm.obs_out + (int64_t)t * prm.batch * G::N, (int64_t)G::N, g0, ngames, lane);
    }
    uint32_t mask = 0;
    if (kRoll) {
      uint32_t ra, rb;
      duo_policy_words(gg, step0 + (uint32_t)t, seed, dcache, h, ra, rb);
      policy_from_words<D>(ra, rb, host_policy, agent_policy, cls, axis, mask, 0);
    }
    const bool prev_done = np < 2;
    // (rollouts: the subset is the policy's 0/1 mask -- the shift as selects, see b_shift_mask)
    const unsigned st = (end_sort && t + 1 == nsteps) ? (stages & ~(unsigned)HK_STAGE_RESCALE) : stages;
    rescale_pending = end_sort && t + 1 == nsteps && (stages & HK_STAGE_RESCALE);
    np = DuoStagesFor<CH, D, 1, kRoll>::run(q, smax, c, axis, np, h, flags, st, mask);
    if (!active) np = 2;
    const bool done = np < 2;
    if (done && length < 0) length = t + 1;
    if (!kRoll && leader) {
      if (prm.done_out) prm.done_out[g] = done;
      if (prm.prev_done_out) prm.prev_done_out[g] = prev_done;
      if (prm.reward_out) prm.reward_out[g] = prm.reward_sign * (float)(done && !prev_done);
      if (prm.num_points_out) prm.num_points_out[g] = np;
    }
    if (want_records && leader) {
      const int64_t at = (int64_t)t * prm.batch + g;
      if (prm.r_host_class_out) prm.r_host_class_out[at] = cls;
      if (prm.r_axis_out) prm.r_axis_out[at] = axis;
      if (prm.r_done_out) prm.r_done_out[at] = done;
      if (prm.r_reward_out) prm.r_reward_out[at] = prm.reward_sign * (float)(done && !prev_done);
    }
    const unsigned long long bd = __ballot(leader && done);
    if (count_slot && lane == 0) count_add(count_slot + (size_t)(t + 1) * count_stride, (uint32_t)__popcll(bd));
    // Fixed point: a game that is down to ONE point sitting at the origin (or to none) does not change any more --
    // whatever the subset and the axis: the shift adds zeros, reposition / rescale find nothing to move, the Newton
    // stage has nothing to compare.  Once every game of the wave is there (a game reaches it one step after it ends
    // when reposition is on; the mean game lasts 5 steps, the longest of 32 about 13), the rest of the episode is
    // the finished-game counts, added in closed form.
    if (MODE == kModeRollout && smax == 1 && bd == __ballot(leader) && t + 1 < nsteps) {
      // (one slot per lane left: a lane holds the game's point, a hole, or nothing)
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
    // re-deal the rows when the widest game of the wave fits fewer slots per lane
    if (t + 1 < nsteps && !__any(active && ((np + 1) >> 1) >= smax)) {
      __syncthreads();
      gmask = duo_scatter<M, CH, D>(q, mine, gmask, smax, h);
      __syncthreads();
      const int sprev = smax;
      nmax = wave_max(active ? np : 0, 2 * smax - 2);
      smax = (nmax + 1) >> 1;
      duo_gather<M, CH, D>(q, mine, gmask, sprev, h);  // slots [smax, sprev) become holes again
    }
  }
  if (kRoll && leader && prm.game_length_out) prm.game_length_out[g] = length;

  // ---- publish: pad everywhere, live rows back in their slots --------------------------------------------------
  __syncthreads();
  if (h == 0) fill_image<M, D>(mine, pad);
  __syncthreads();
  if (kEndSort && end_sort) {
    int rank[CH];
    duo_ranks_first<CH, D>(q, smax, rank);
    if (rescale_pending) d_rescale<CH, D, CH>(q, flags);
    duo_scatter_ranked<CH, D>(q, mine, rank, smax);
  } else {
    duo_scatter<M, CH, D>(q, mine, gmask, smax, h);
  }
  __syncthreads();
  duo_store_slab<M, D, DuoGeom<M, D>::QH, (MODE == kModeRollout)>(lds, (float*)prm.out, prm.out_stride, g0, ngames, lane);
#ifdef HK_DUO_PROBE
  if (kRoll && lane == 0 && prm.game_length_out && ngames >= 8) {
    int32_t* w = prm.game_length_out + g0;
    w[0] = (int32_t)probe_t0;
    w[1] = (int32_t)probe_t1;
    w[2] = (int32_t)probe_t2;
    w[3] = (int32_t)wall_clock64();
    w[4] = probe_steps;
    w[5] = probe_smax;
    w[6] = (int32_t)blockIdx.x;
    w[7] = (int32_t)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));  // HW_ID
    if (ngames >= 32)
      for (int i = 0; i < 24; ++i) w[8 + i] = probe_buf[i];
  }
#endif
}

// ---- host side -----------------------------------------------------------------------------------------------
// steps and rollouts on a shape with a register-resident specialisation, when the batch leaves SIMDs
// short of a second wave under the one-lane kernel
// SIMDs of the current device (4 per CU); queried once per device
inline int device_simds() {
  static int cached[16] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
  if (!cached[dev]) {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    (void)hipGetLastError();
    cached[dev] = 4 * cus;
  }
  return cached[dev];
}

inline bool duo_wanted(const Params& prm) {
  if ((prm.mode != kModeRollout && prm.mode != kModeStep) || prm.m > 32) return false;
  if (prm.mode == kModeStep && (prm.class_out || (prm.stages & kStageFeatureSorts))) return false;
  const bool any_records = prm.obs_out || prm.r_host_class_out || prm.r_axis_out || prm.r_done_out || prm.r_reward_out;
  // Zeillinger's host: plain rollouts (duo_kernel<..., ZEIL>); with records the one-lane kernel
  if (prm.mode == kModeRollout && prm.host_policy == HK_HOST_ZEILLINGER && any_records) return false;
  if ((prm.stages & HK_STAGE_NEWTON) &&
      ((prm.flags & HK_SEM_MASK) == HK_SEM_LIST || (prm.flags & HK_FLAG_COMPACT_SORTED))) {
    // sorted + compacted output: plain rollouts only (they sort once, at the end)
    const bool records = prm.obs_out || prm.r_host_class_out || prm.r_axis_out || prm.r_done_out || prm.r_reward_out;
    if (prm.mode != kModeRollout || records) return false;
  }
  if (prm.flags & HK_FLAG_FORCE_TWO_LANES) return true;
  // measured (scripts/probe_duo.py): ahead while the one-lane kernel (64 games per wave) leaves SIMDs short of
  // a second wave -- up to 1.5 waves per SIMD, 98 304 games on the 1024 SIMDs of an MI355X -- and at any size
  // for the shapes whose one-lane kernel runs one wave per SIMD
  return (int64_t)prm.batch * 2 <= (int64_t)3 * kWave * device_simds() || prm.m * prm.d > 64;
}

template <int M, int D>
int launch_duo_t(Params prm, hipStream_t stream) {
  const unsigned grid = (unsigned)(((int64_t)prm.batch + kDuoGames - 1) / kDuoGames);
  prm.games_per_block = kDuoGames;
  launch_prepare();
  // (the small records alone ride on the plain rollout kernels: ACTS)
  const bool records = prm.obs_out != nullptr;
  const bool acts = !records && (prm.r_host_class_out || prm.r_axis_out || prm.r_done_out || prm.r_reward_out);
  const int hot = (prm.mode == kModeRollout && !records) ? fast_hot_config(prm) : kHotNone;
  if (prm.mode == kModeRollout && records)
    hipLaunchKernelGGL((duo_kernel<M, D, kModeRolloutRec>), dim3(grid), dim3(kWave), 0, stream, (const float*)prm.in,
                       prm.in_stride, prm.batch, prm);
  else if (prm.mode == kModeRollout && prm.host_policy == HK_HOST_ZEILLINGER)
    hipLaunchKernelGGL((duo_kernel<M, D, kModeRollout, kHotNone, false, true>), dim3(grid), dim3(kWave), 0, stream,
                       (const float*)prm.in, prm.in_stride, prm.batch, prm);
  else if (prm.mode == kModeRollout && acts && hot == kHotJax)
    hipLaunchKernelGGL((duo_kernel<M, D, kModeRollout, kHotJax, true>), dim3(grid), dim3(kWave), 0, stream,
                       (const float*)prm.in, prm.in_stride, prm.batch, prm);
  else if (prm.mode == kModeRollout && acts)
    hipLaunchKernelGGL((duo_kernel<M, D, kModeRollout, kHotNone, true>), dim3(grid), dim3(kWave), 0, stream,
                       (const float*)prm.in, prm.in_stride, prm.batch, prm);
  else if (prm.mode == kModeStep && prm.flags == HK_SEM_JAX &&
           prm.stages == (HK_STAGE_SHIFT | HK_STAGE_REPOSITION | HK_STAGE_NEWTON))
    // take_actions as the JAX trainer configures it (jax/util.py:83-125 with reposition, without rescale)
    hipLaunchKernelGGL((duo_kernel<M, D, kModeStep, kHotJax>), dim3(grid), dim3(kWave), 0, stream,
                       (const float*)prm.in, prm.in_stride, prm.batch, prm);
  else if (prm.mode == kModeStep)
    hipLaunchKernelGGL((duo_kernel<M, D, kModeStep>), dim3(grid), dim3(kWave), 0, stream, (const float*)prm.in,
                       prm.in_stride, prm.batch, prm);
  else if (hot == kHotJax)
    hipLaunchKernelGGL((duo_kernel<M, D, kModeRollout, kHotJax>), dim3(grid), dim3(kWave), 0, stream,
                       (const float*)prm.in, prm.in_stride, prm.batch, prm);
  else if (hot == kHotTorch)
    hipLaunchKernelGGL((duo_kernel<M, D, kModeRollout, kHotTorch>), dim3(grid), dim3(kWave), 0, stream,
                       (const float*)prm.in, prm.in_stride, prm.batch, prm);
  else
    hipLaunchKernelGGL((duo_kernel<M, D, kModeRollout>), dim3(grid), dim3(kWave), 0, stream, (const float*)prm.in,
                       prm.in_stride, prm.batch, prm);
  return launch_status();
}

// instantiated next to the one-lane specialisations (hk_fast_spec.hip); the dispatcher lives in the main unit
#ifndef HK_SPEC_TU
#define HK_X(M_, D_) extern template int launch_duo_t<M_, D_>(Params, hipStream_t);
HK_FAST_SPECS(HK_X)
#undef HK_X

inline int launch_duo(const Params& prm, hipStream_t stream) {
#define HK_X(M_, D_) if (prm.m == M_ && prm.d == D_) return launch_duo_t<M_, D_>(prm, stream);
  HK_FAST_SPECS(HK_X)
#undef HK_X
  return HK_ERR_UNSUPPORTED;
}
#endif

}  // namespace hk
