// Four lanes per game, fused rollouts: hk_rollout on the four-lane scheme of hk_quad_kernel.h.
//
// hk::team_kernel served the rollouts of the large games ((50,4): 800 B per game) with four lanes per game too, but
// mirrored every row through LDS for the pair test (one broadcast ds_read per row and slot, 38 % of its LDS cycles bank
// conflicts) inside ONE loop over the steps with a dispatch over the buckets in it: 378 us per 20-step episode of
// 262 144 games against the ~90 us it takes to move the state once each way.  This kernel is the rollout loop of
// hk_duo_kernel.h on the quad's data flow:
//   * 16 games per wave; slab in by LDS-DMA, each lane scans its quarter of the rows, every live row goes to the slot
//     of its rank in a compact image laid OVER the slab image (rows of 16 B; the row's index in the game travels in a
//     byte array next to it), lane j reads ranks j, j + 4, ...: rows (and their indices) live in registers from then on;
//   * the domination test by DPP quad exchanges (qd_newton: own slots' triangle + the lane one up + half of the lane
//     two up, 2 S^2 tests per lane for S slots), beyond 8 slots per lane the segmented loop over the parked rows
//     (qd_newton_lds); column minima / maxima through two DPP exchanges; the shift as selects (0/1 subsets);
//   * a STAIRCASE of step loops, one per bucket of slots per lane, entered from the top down; when the widest game of the
//     wave fits fewer slots the quads re-deal their rows through the compact image (rank = popcounts of the four lanes'
//     live masks, no serial scan);
//   * the policy stream off the critical path: the wave computes the DECODED actions (subset mask, axis) of a window of
//     24 steps per game before the first step -- while its slab is in flight -- and parks them in LDS, one byte per
//     game and step; a step reads its byte (all four lanes of a quad the same address).  Philox is keyed by the
//     global game index and the step as everywhere else: bit-identical results;
//   * a game's first finished step is kept per game; the per-step finished-game counts are ballots over those at the
//     end of the launch (a finished game stays finished), not an atomic per step inside the loop;
//   * fixed-point exit as in the other rollout kernels (every game of the wave without a point, or with one point at the
//     origin);
//   * at the end the image is rebuilt (padding everywhere, survivors at their own rows) and stored as one slab.
// Plain and recording rollouts (REC: per-step observations / records); no Zeillinger host -- its choice depends on the
// state --, no sorted output; exactness guard and whole-wave fallback on the generic routines as in the other kernels.
#pragma once

#include "hk_quad_kernel.h"
#include "hk_quadgen_kernel.h"

namespace hk {

constexpr int kQrBlocks = 6;               // Philox blocks per game in a window of actions (24 steps: four per block)
constexpr int kQrSteps = 4 * kQrBlocks;

template <int M, int D>
struct QuadRollGeom {
  using Q = QuadGeom<M, D>;
  static constexpr int R = Q::R;
  static constexpr int CW = (D <= 4) ? 4 : D;                  // floats per compact row
  static constexpr int kCompact = kQuadGames * M * CW;
  static constexpr int kRegion = Q::kImage > kCompact ? Q::kImage : kCompact;  // floats: image and compact image alias
  static constexpr int kTags = (kQuadGames * M + 15) / 16 * 16;                 // bytes
  static constexpr int kActs = (kQrSteps + 1) * kQuadGames;  // bytes (>= the slow path's scratch); one row more than the
                                                              // window: the step loop requests the next step's byte ahead
  static constexpr int kWaveBytes = kRegion * 4 + kTags + kActs;
  static constexpr bool kBig = kWaveBytes > 12 * 1024;
  // waves per workgroup: four; one for the large games, whose 160 KB of LDS per CU then hold 11 waves (three waves
  // per SIMD: <= 168 VGPRs) instead of 8
  static constexpr int kWpb = kBig ? 1 : 4;
  static constexpr int kWavesPerSimd = kBig ? 3 : 4;
  static_assert(D <= 5, "an action travels as a byte: subset mask (D bits) | axis << 5");
  static_assert(kActs >= kQuadGames * D * 4, "the slow path's subset scratch lies over the action window");
  // the bucket below nb on the ladder 1..6, 8, 10, 13, 16, ... (QuadGeom::next_bucket), the top bucket being R
  static constexpr int prev_bucket(int nb) {
    int p = 0, b = 1;
    while (b < nb) {
      p = b;
      b = Q::next_bucket(b);
    }
    return p;
  }
};

// f(NB, LO) for every bucket NB from R down, LO the next smaller one (0 below the first)
template <int M, int D, int NB>
struct QuadLevels {
  static constexpr int kLo = QuadRollGeom<M, D>::prev_bucket(NB);
  template <typename F>
  static __device__ __forceinline__ void run(F&& f) {
    f(std::integral_constant<int, NB>{}, std::integral_constant<int, kLo>{});
    if constexpr (kLo >= 1) QuadLevels<M, D, kLo>::run(f);
  }
};

// reposition when every coordinate is >= +0 or the +inf of a hole (in-kernel 0/1 subsets; see d_reposition): unsigned
// minima of the bit patterns, the quad's through two DPP exchanges
template <int R, int D, int NB>
__device__ __forceinline__ void qr_reposition_bin(float (&q)[R * D]) {
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
    uint32_t o = (uint32_t)qperm_i<kQuadSwap1>((int)mb[k]);
    mb[k] = o < mb[k] ? o : mb[k];
    o = (uint32_t)qperm_i<kQuadUp2>((int)mb[k]);
    mb[k] = o < mb[k] ? o : mb[k];
    // (a column without live rows -- an empty game -- subtracts the largest finite float: +inf stays +inf, one
    // v_min_u32 where the test for +inf took a compare and a select)
    const float sub = __uint_as_float(mb[k] < 0x7F7FFFFFu ? mb[k] : 0x7F7FFFFFu);
#pragma unroll
    for (int r = 0; r < NB; ++r) q[r * D + k] = q[r * D + k] - sub;
  }
}

// one transition on slots [0, NB) of the four lanes with the policy's 0/1 subset `cmask`; returns the GAME's live rows
template <int M, int CW, int R, int D, int NB>
__device__ __forceinline__ int qr_stages(float (&q)[R * D], uint32_t cmask, int axis, int np, int j, unsigned flags,
                                         unsigned stages, float* cmine, int smax) {
  if (stages & HK_STAGE_SHIFT) b_shift_mask<R, D, NB>(q, cmask, axis, np, flags);
  if (stages & HK_STAGE_REPOSITION) qr_reposition_bin<R, D, NB>(q);
  if (stages & HK_STAGE_NEWTON) {
    if constexpr (NB > kQuadDppSlots) {
      const int slots_end = kQuad * smax < M ? kQuad * smax : M;
      qd_newton_lds<M, CW, R, D, NB, true>(q, cmine, j, slots_end);
    } else {
      qd_newton<R, D, NB>(q, j);
    }
  }
  if (stages & HK_STAGE_RESCALE) qd_rescale<R, D, NB>(q, flags);
  int n = 0;
#pragma unroll
  for (int r = 0; r < NB; ++r) n += (q[r * D] < INFINITY) ? 1 : 0;
  return q_sum(n);
}

// the decoded actions of blocks [wb0, wb0 + nb) of the wave's 16 games: lane l serves game l & 15, blocks (l >> 4) + 4 i.
// One byte per game and step: subset mask | axis << 5.
template <int D>
__device__ __forceinline__ void qr_policy_fill(uint8_t* act, uint64_t gg, uint32_t wb0, int nb, uint64_t seed,
                                               int host_policy, int agent_policy, int lane) {
  // gg: the policy stream's index of game (lane & 15) of the wave -- the lane's FILL game, not the game it plays
  static_assert(D <= kPolicyShortDim, "four steps per Philox block (hk_common.h policy_words)");
  const int game = lane & (kQuadGames - 1), sub = lane >> 4;
#pragma nounroll
  for (int i = 0; i < kQrBlocks; i += 4) {
    if (i >= nb) break;  // wave-uniform
    const int b = i + sub;
    if (b < kQrBlocks) {
      const U4 r = philox4x32((uint32_t)gg, (uint32_t)(gg >> 32), wb0 + (uint32_t)b, kStreamPolicy, seed);
      int cls, axis;
      uint32_t mask;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint32_t w = u4_word(r, k);
        policy_from_words<D>(w & 0xFFFF0000u, w << 16, host_policy, agent_policy, cls, axis, mask, 0);
        act[(4 * b + k) * kQuadGames + game] = (uint8_t)(mask | ((uint32_t)axis << 5));
      }
    }
  }
}

// the quad's rows to their ranks in the compact image (rank order = (slot, lane) order = the order they had), the rows'
// indices next to them; then slots [0, snew) back: lane j takes ranks j, j + 4, ...  The index of the row in slot s of
// lane j lives in tmine[4 s + j] between two deals, NOT in a register: it is looked at by the next deal and by the final
// image only, and thirteen more loop-carried registers were part of what the (50,4) kernels spilled.
template <int M, int CW, int R, int D, int NB>
__device__ __forceinline__ void qr_redeal(float (&q)[R * D], float* cmine, uint8_t* tmine, int j, int np, int snew) {
  // (opaque: the masks below depend on the lane only, and hoisted out of the step loops -- 3 per slot, live through the
  // whole staircase -- they were most of the (50,4) kernels' 212 - 332 B of scratch per lane)
  asm volatile("" : "+v"(j));
  uint32_t lm = 0;
#pragma unroll
  for (int s = 0; s < NB; ++s) lm |= (q[s * D] < INFINITY) ? (1u << s) : 0u;
  const uint32_t l1 = (uint32_t)qperm_i<kQuadUp1>((int)lm), l2 = (uint32_t)qperm_i<kQuadUp2>((int)lm),
                 l3 = (uint32_t)qperm_i<kQuadUp3>((int)lm);
  // the lane k up is lane (j + k) & 3 of the quad: its slot s comes before mine iff that lane index is below j
  const bool b1 = ((j + 1) & 3) < j, b2 = ((j + 2) & 3) < j, b3 = ((j + 3) & 3) < j;
  uint32_t tg[NB];  // my rows' indices, read before any lane writes a new one
#pragma unroll
  for (int s = 0; s < NB; ++s) tg[s] = tmine[kQuad * s + j < M ? kQuad * s + j : 0];
#pragma unroll
  for (int s = 0; s < NB; ++s) asm volatile("" : "+v"(tg[s]));
  wave_lds_fence();
#pragma unroll
  for (int s = 0; s < NB; ++s) {
    const uint32_t below = (1u << s) - 1u, at = 1u << s;
    const int rank = __popc(lm & below) + __popc(l1 & (below | (b1 ? at : 0u))) + __popc(l2 & (below | (b2 ? at : 0u))) +
                     __popc(l3 & (below | (b3 ? at : 0u)));
    if ((lm >> s) & 1u) {
      float* dst = cmine + rank * CW;
      if constexpr (D == 4) {
        *reinterpret_cast<vf4*>(dst) = vf4{q[s * D], q[s * D + 1], q[s * D + 2], q[s * D + 3]};
      } else if constexpr (D == 3) {
        *reinterpret_cast<vf4*>(dst) = vf4{q[s * D], q[s * D + 1], q[s * D + 2], 0.0f};
      } else {
#pragma unroll
        for (int k = 0; k < D; ++k) dst[k] = q[s * D + k];
      }
      tmine[rank] = (uint8_t)tg[s];
    }
  }
  wave_lds_fence();
  unrolled_while<0, NB>([&](auto sc) {
    constexpr int s = decltype(sc)::value;
    if (s >= snew) {  // slots past the new bucket are holes again (no live row can sit there: np <= 4 snew)
#pragma unroll
      for (int k = 0; k < D; ++k) q[s * D + k] = INFINITY;
      return true;
    }
    const bool has = kQuad * s + j < np;
    const int r = kQuad * s + j < M ? kQuad * s + j : 0;
    const float* src = cmine + r * CW;
    if constexpr (D == 4 || D == 3) {
      const vf4 v = *reinterpret_cast<const vf4*>(src);
      q[s * D] = has ? v.x : INFINITY;
      q[s * D + 1] = has ? v.y : INFINITY;
      q[s * D + 2] = has ? v.z : INFINITY;
      if constexpr (D == 4) q[s * D + 3] = has ? v.w : INFINITY;
    } else {
#pragma unroll
      for (int k = 0; k < D; ++k) {
        const float v = src[k];
        q[s * D + k] = has ? v : INFINITY;
      }
    }
    return true;
  });
  wave_lds_fence();  // (the compact image is scratch again: qd_newton_lds parks rows there)
}

// the wave's image from the rows in registers: padding everywhere (16-B pieces), then every live row at its own place
// (SB: compile-time bound of the slots in use -- the level of the staircase the caller is on)
template <int M, int D, int R, int SB = R>
__device__ __forceinline__ void qr_build_image(const float (&q)[R * D], const uint8_t* tags, float* region, int smax,
                                               float pad, int j, int lane) {
  using G = QuadGeom<M, D>;
  // (the game's addresses from the lane, here: as loop invariants they lived in registers through the whole staircase)
  int gi = lane >> 2;
  asm volatile("" : "+v"(gi));
  const uint8_t* tmine = tags + gi * M;
  float* mine = region + gi * G::N;
  wave_lds_fence();
#pragma unroll
  for (int it = 0; it < G::QL; ++it) {
    const int qq = lane + it * kWave;
    if (qq < kQuadGames * G::Q) {
      if constexpr (G::W == 4) *reinterpret_cast<vf4*>(region + qq * 4) = vf4{pad, pad, pad, pad};
      else if constexpr (G::W == 2) *reinterpret_cast<vf2*>(region + qq * 2) = vf2{pad, pad};
      else region[qq] = pad;
    }
  }
  wave_lds_fence();
  unrolled_while<0, SB>([&](auto sc) {
    constexpr int s = decltype(sc)::value;
    if (s >= smax) return false;
    if (q[s * D] < INFINITY) {
      float* dst = mine + (int)tmine[kQuad * s + j < M ? kQuad * s + j : 0] * D;  // (the row's index in the game)
      if constexpr (D == 4) {
        *reinterpret_cast<vf4*>(dst) = vf4{q[s * D], q[s * D + 1], q[s * D + 2], q[s * D + 3]};
      } else {
#pragma unroll
        for (int k = 0; k < D; ++k) dst[k] = q[s * D + k];
      }
    }
    return true;
  });
  wave_lds_fence();
}

// list semantics (_list_ops.py:25-41) for PLAIN rollouts: the state leaves the kernel sorted (descending lexicographic,
// coordinate 0 first) and compacted -- ranked once, when it is published (between two Newton stages the order of the rows
// changes nothing: hk_fast_kernel.h), before a pending rescale could round two keys together.  Rows are distinct after
// a Newton stage.  Up to 8 slots per lane the ranks come through DPP (qd_ranks_first), beyond it from a rolled loop over
// the rows parked by rank.
template <int M, int CW, int R, int D, int NB>
__device__ __forceinline__ void qr_build_sorted(float (&q)[R * D], float* region, float* cmine, int smax, float pad,
                                                int j, int lane, bool rescale_pending, unsigned flags) {
  using G = QuadGeom<M, D>;
  int rank[R];
  if constexpr (NB <= kQuadDppSlots) {
    qd_ranks_first<R, D, NB>(q, rank);
  } else {
    wave_lds_fence();
    qd_ranks_lds_t<M, CW, R, D, NB, kKeyFirst, false>(q, cmine, j, kQuad * smax < M ? kQuad * smax : M, rank);
  }
  if (rescale_pending) qd_rescale<R, D, NB>(q, flags);
  int gi = lane >> 2;
  asm volatile("" : "+v"(gi));
  float* mine = region + gi * G::N;
  wave_lds_fence();
#pragma unroll
  for (int it = 0; it < G::QL; ++it) {
    const int qq = lane + it * kWave;
    if (qq < kQuadGames * G::Q) {
      if constexpr (G::W == 4) *reinterpret_cast<vf4*>(region + qq * 4) = vf4{pad, pad, pad, pad};
      else if constexpr (G::W == 2) *reinterpret_cast<vf2*>(region + qq * 2) = vf2{pad, pad};
      else region[qq] = pad;
    }
  }
  wave_lds_fence();
  unrolled_while<0, NB>([&](auto sc) {
    constexpr int s = decltype(sc)::value;
    if (s >= smax) return false;
    if (q[s * D] < INFINITY) {
      float* dst = mine + rank[s] * D;
#pragma unroll
      for (int k = 0; k < D; ++k) dst[k] = q[s * D + k];
    }
    return true;
  });
  wave_lds_fence();
}

// ZEIL: Zeillinger's host (jax/players.py:55-109) on four lanes per game -- the quad parks its rows by rank (4 s + j)
// in the compact image (scratch between the stages), lane j takes the rows i = j, j + 4, ... against every later row as
// a rolled loop, two pairs per pass (hk_fast_rows.h: zeil_pair, one packed key (L, S, pair index)); the four bests
// merge through two DPP rotations; the chosen pair's difference is re-read from the parked rows.  hk_duo_kernel.h has
// the two-lane twin.
template <int M, int CW, int R, int D>
__device__ __forceinline__ int qr_zeillinger(const float (&q)[R * D], float* cmine, int j, int smax, int np) {
  wave_lds_fence();
  unrolled_while<0, R>([&](auto sc) {
    constexpr int s = decltype(sc)::value;
    if (s >= smax) return false;
    if (kQuad * s + j < M) {  // (ranks past the game's M rows do not exist: the image holds M rows per game)
      float* dst = cmine + (kQuad * s + j) * CW;
      if constexpr (D == 4) {
        *reinterpret_cast<vf4*>(dst) = vf4{q[s * D], q[s * D + 1], q[s * D + 2], q[s * D + 3]};
      } else {
#pragma unroll
        for (int k = 0; k < D; ++k) dst[k] = q[s * D + k];
      }
    }
    return true;
  });
  wave_lds_fence();
  // ranks in use (wave-uniform; holes are +inf): the bucket's 4 smax, or -- `np`: a bound on the ranks in use of the
  // lane's game, where the caller has one (freshly dealt rows: the live rows ARE the first np ranks; between two deals
  // the survivors keep their slots and the caller passes M) -- the wave's largest game, at most three ballots below
  // (the pair loop is quadratic in it)
  int n = (kQuad * smax < M) ? kQuad * smax : M;
#pragma nounroll
  for (int t = 0; t < kQuad - 1 && n > 2 && !__any(np >= n); ++t) --n;
  // The pairs i < jj < n in COLUMN-major order (position jj (jj - 1) / 2 + i), lane j taking every fourth position from
  // j on -- every lane the same number of pairs, whatever n (row-wise, lane j took the rows i = j, j + 4, ... against all
  // later rows: the inner loops ran in lockstep for the longest of the four, 55 passes of pairs where 47.5 are due at
  // n = 20).  A step of four positions wraps at most once from column 4 on; the six pairs of the columns 1 .. 3 go
  // first.  Two positions per pass, a best of its own each (two independent chains; the pair's index is part of the
  // key, so the order of the merges does not matter).
  ZeilBest<D> best, second;
  auto pair = [&](ZeilBest<D>& bst, int i, int jj, auto checked, bool ok) {
    float pi[D], pj[D];
#pragma unroll
    for (int k = 0; k < D; ++k) {
      pi[k] = cmine[i * CW + k];
      pj[k] = cmine[jj * CW + k];
    }
    zeil_pair<D, false, decltype(checked)::value>(bst, pi, pj, ok, 64 * i + jj);
  };
  {
    const int ia = (j == 2) ? 1 : 0, ja = (j == 0) ? 1 : ((j == 3) ? 3 : 2);  // (0,1) (0,2) (1,2) (0,3)
    const bool oka = ja < n;
    pair(best, ia, oka ? ja : 0, std::true_type{}, oka);
    const bool okb = j < 2 && 3 < n;                                           // (1,3) (2,3)
    pair(second, okb ? j + 1 : 0, okb ? 3 : 0, std::true_type{}, okb);
  }
  const int rest = n >= 5 ? n * (n - 1) / 2 - 6 : 0;  // the pairs of the columns >= 4
  const int passes = (rest + 2 * kQuad - 1) / (2 * kQuad);
  int i = j, jj = 4;
  auto advance = [&]() {
    i += kQuad;
    const bool wrap = i >= jj;
    i = wrap ? i - jj : i;
    jj += wrap ? 1 : 0;
  };
#pragma nounroll
  for (int t = 0; t + 1 < passes; ++t) {  // (all but the last pass: every position exists)
    float pa[D], qa[D], pb[D], qb[D];
    const int ia = i, ja = jj;
    advance();
#pragma unroll
    for (int k = 0; k < D; ++k) {
      pa[k] = cmine[ia * CW + k];
      qa[k] = cmine[ja * CW + k];
      pb[k] = cmine[i * CW + k];
      qb[k] = cmine[jj * CW + k];
    }
    zeil_pair2<D>(best, second, pa, qa, 64 * ia + ja, pb, qb, 64 * i + jj);
    advance();
  }
  if (passes > 0) {
    const bool ok1 = jj < n;
    pair(best, ok1 ? i : 0, ok1 ? jj : 0, std::true_type{}, ok1);
    advance();
    const bool ok2 = jj < n;
    pair(second, ok2 ? i : 0, ok2 ? jj : 0, std::true_type{}, ok2);
  }
  zeil_merge<D, false>(best, second);
  {
    ZeilBest<D> o;
    o.hi = (uint32_t)qperm_i<kQuadUp1>((int)best.hi);
    o.lo = (uint32_t)qperm_i<kQuadUp1>((int)best.lo);
    zeil_merge<D, false>(best, o);
    o.hi = (uint32_t)qperm_i<kQuadUp2>((int)best.hi);
    o.lo = (uint32_t)qperm_i<kQuadUp2>((int)best.lo);
    zeil_merge<D, false>(best, o);
  }
  const bool have = best.have();
  const int bi = have ? (int)((best.lo & 0xFFFFu) >> 6) : 0, bj = have ? (int)(best.lo & 63u) : 0;
  float bd[D];
#pragma unroll
  for (int k = 0; k < D; ++k) bd[k] = cmine[bi * CW + k] - cmine[bj * CW + k];
  wave_lds_fence();  // (the stages park rows here again)
  int lo = 0, hi = 0;
  float vlo = bd[0], vhi = bd[0];
#pragma unroll
  for (int k = 1; k < D; ++k) {
    if (bd[k] < vlo) { vlo = bd[k]; lo = k; }
    if (bd[k] > vhi) { vhi = bd[k]; hi = k; }
  }
  if (!have || lo == hi) return 0;
  return encode_mask((1u << lo) | (1u << hi));
}

// REC: the recording rollout -- per step the observation (the state before the step, rebuilt from the rows in
// registers and stored as one slab) and / or the small records (host class, axis, done, reward) -- what the
// simulate-shaped consumers read (hironaka/jax/simulation_fn.py:196-211).  Once every game of the wave is at its
// fixed point only the stores go on (the same image, the policies' draws, done = 1, reward = 0).
// GEN: the initial states are drawn inside the launch (hk_rollout_desc.gen_max_value: hk_quadgen_kernel.h's rows ->
// the generator's stages on all slots -> one re-deal into the wave's bucket) and `episodes` of them run back to back --
// episode e with seed + e and gen_seed + e, the counts accumulating; prm.out may be NULL (no final state is stored):
// the whole loop of JAXTrainer.compute_rho (jax_trainer.py:502-555) without a byte of state traffic.
// EPI (without GEN): `episodes` of them from the states in memory -- every episode reads its slab again (hk_rollout_desc.
// episodes with points_in); its own instantiation, so that the one-episode kernels carry no loop.
template <int M, int D, int HOT, int WPB, bool REC = false, bool ZEIL = false, bool GEN = false, bool EPI = GEN>
__global__ __launch_bounds__(kWave * WPB, (QuadRollGeom<M, D>::kWavesPerSimd)) void quadroll_kernel(
    const float* in0, int64_t in_stride0, int batch0, const Params prm) {
  static_assert(!GEN || EPI, "generated initial states come with the episode loop");
  static_assert(!EPI || !REC, "episodes back to back: plain rollouts");
  static_assert(!ZEIL || (!REC && HOT == kHotNone), "Zeillinger's host: plain rollouts, policies inside the loop");
  static_assert(!GEN || !REC, "generated initial states: plain rollouts");
  static_assert(!GEN || QuadRollGeom<M, D>::kRegion >= QuadGenGeom<M, D>::kRegion, "the generator's staging image lies in the region");
  using G = QuadGeom<M, D>;
  using RG = QuadRollGeom<M, D>;
  constexpr int R = RG::R, CW = RG::CW;
  using MaskM = MaskT<M>;
  static_assert(M <= 255, "a row's index travels as a byte");
  __shared__ __align__(16) float lds_all[WPB * RG::kRegion];
  __shared__ __align__(16) uint8_t tag_all[WPB * RG::kTags];
  __shared__ __align__(16) uint8_t act_all[WPB * RG::kActs];  // (the slow path's subset scratch lies over it)
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & (kWave - 1);
  float* region = lds_all + wave * RG::kRegion;
  uint8_t* tags = tag_all + wave * RG::kTags;
  uint8_t* act = act_all + wave * RG::kActs;
  const int j = lane & 3, gi = lane >> 2;
  const int64_t g0 = ((int64_t)blockIdx.x * WPB + wave) * kQuadGames;
  const int64_t left = (int64_t)batch0 - g0;
  if (left <= 0) return;  // (waves of a workgroup never meet at a barrier)
  const int ngames = (int)(left < kQuadGames ? left : kQuadGames);
  const bool active = gi < ngames;
  const bool leader = active && j == 0;
  const int64_t g = g0 + gi;
  // hk_rollout_desc.game_ids (a re-ordered batch keeps every game's policy stream): the id of the lane's FILL game
  // (lane & 15: the game whose action bytes it computes), requested before the slab and not touched until the slab's
  // requests are out (hk_duo_kernel.h); the id of the game the lane plays comes from that lane when it is needed
  const bool has_ids = prm.game_ids != nullptr;
  const int fill_game = lane & (kQuadGames - 1);
  uint32_t raw_id = 0;
  if (has_ids && fill_game < ngames) raw_id = (uint32_t)prm.game_ids[g0 + fill_game];
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (!GEN) quad_slab_load<M, D>(in0 + g0 * G::N, region, ngames, lane);
  __builtin_amdgcn_sched_barrier(0);
  const float pad = prm.pad_f32;
  const unsigned flags = (HOT == kHotJax) ? (unsigned)HK_SEM_JAX : (HOT == kHotTorch) ? kHotTorchFlags : prm.flags;
  const unsigned stages = (HOT == kHotJax || HOT == kHotTorch) ? (unsigned)(HK_STAGE_SHIFT | HK_STAGE_REPOSITION | HK_STAGE_NEWTON)
                                                              : prm.stages;
  const float fill = ((flags & HK_SEM_MASK) == HK_SEM_JAX) ? -1.0f : pad;
  const int nsteps = prm.steps;
  uint32_t step0 = prm.step_offset;
  int host_policy = prm.host_policy, agent_policy = prm.agent_policy;
  asm volatile("" : "+s"(step0), "+s"(host_policy), "+s"(agent_policy));
  const uint64_t gg_fill = prm.game_offset + (has_ids ? (uint64_t)raw_id : (uint64_t)(g0 + fill_game));
  const uint32_t wb_last = nsteps > 0 ? (step0 + (uint32_t)nsteps - 1u) >> 2 : step0 >> 2;
  float* mine = region + gi * G::N;
  const int episodes = EPI ? prm.episodes : 1;
  for (int ep = 0;; ++ep) {  // (EPI: `episodes` of them; the loop's body is not indented)
  const bool last_episode = !EPI || ep + 1 >= episodes;
  if constexpr (EPI && !GEN) {
    if (ep > 0) quad_slab_load<M, D>(in0 + g0 * G::N, region, ngames, lane);  // (the first one went out above)
  }
  uint64_t seed = prm.seed + (uint64_t)ep;
  asm volatile("" : "+s"(seed));
  // the first window of decoded actions, computed while the slab is in flight
  uint32_t wb0 = step0 >> 2;  // first Philox block of the window (wave-uniform)
  if constexpr (!ZEIL) {
    const uint32_t nb = wb_last - wb0 + 1u;
    qr_policy_fill<D>(act, gg_fill, wb0, (int)(nb < (uint32_t)kQrBlocks ? nb : (uint32_t)kQrBlocks), seed, host_policy,
                      agent_policy, lane);
  }
  float q[R * D];
  int np, smax;
  float* cmine = region + gi * (M * CW);
  uint8_t* tmine = tags + gi * M;
  if constexpr (GEN) {
    // ---- a fresh game per quad: slot s of lane j = row 4 s + j, every row live; the generator's stages on all R slots;
    // then the quads re-deal into the wave's bucket (rank = popcounts, as between two steps) -------------------------------
    wave_lds_fence();
    const uint64_t gg_gen = prm.game_offset + (has_ids ? (uint64_t)(uint32_t)__shfl((int)raw_id, gi) : (uint64_t)g);
    qg_rows<M, D>(q, region, gg_gen, prm.gen_seed + (uint64_t)ep, (uint32_t)prm.max_value, j, gi);
#pragma unroll
    for (int s = 0; s < R; ++s)
      if ((kQuad * s + kQuad <= M) || kQuad * s + j < M) tmine[kQuad * s + j] = (uint8_t)(kQuad * s + j);
    np = qg_stages<M, D, R * D, false>(q, j, flags, prm.gen_stages, cmine, nullptr, prm.max_value, lane);
    smax = R;
#pragma nounroll
    while (smax > 1 && !__any(active && np > kQuad * (smax - 1))) --smax;
    if (smax < R) qr_redeal<M, CW, R, D, R>(q, cmine, tmine, j, active ? np : 0, smax);
  } else {
  wait_vmem_all();
  wave_lds_fence();

  // ---- live rows + exactness guard: lane j looks at rows j*R .. j*R + R - 1 (hk_quad_kernel.h) ------------------------
  float rows[R * D];
  const int i0 = j * R;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if constexpr (D == 4) {
      const vf4 v = *reinterpret_cast<const vf4*>(mine + (i0 + r < M ? i0 + r : M - 1) * D);
      rows[r * D] = (i0 + r < M) ? v.x : fill;
      rows[r * D + 1] = (i0 + r < M) ? v.y : fill;
      rows[r * D + 2] = (i0 + r < M) ? v.z : fill;
      rows[r * D + 3] = (i0 + r < M) ? v.w : fill;
    } else {
#pragma unroll
      for (int k = 0; k < D; ++k) rows[r * D + k] = (i0 + r < M) ? mine[(i0 + r) * D + k] : fill;
    }
  }
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
  np = mask_pop(gmask);
  const bool exact = (fill == pad) && !__any(bad != 0);

  if (!exact) {
    // ---- slow path (whole wave): the quad's first lane runs the exact generic routines on the image -----------------
    wave_lds_fence();
    float* cs = reinterpret_cast<float*>(act) + gi * D;
    const uint64_t gg = prm.game_offset + (has_ids ? (uint64_t)(uint32_t)__shfl((int)raw_id, gi) : (uint64_t)g);
    PolicyCache pcache;
    if constexpr (ZEIL) {
      if (prm.class_out) {  // hk_zeillinger: the class is the launch's only product
        if (leader) (prm.class_out + g0)[(unsigned)gi] = zeillinger_game<float>(mine, prm.m, prm.d);
        return;
      }
    }
    np = leader ? num_points<float>(mine, M, D) : 2;
    int length = (np < 2) ? 0 : -1;
    if (prm.count_ws) {
      const unsigned long long b0 = __ballot(leader && np < 2);
      if (lane == 0 && b0) count_add(prm.count_ws + blockIdx.x, (uint32_t)__popcll(b0));
    }
    for (int t = 0; t < nsteps; ++t) {
      int axis = -1, cls = 0;
      uint32_t mask;
      if constexpr (REC) {
        if (prm.obs_out) {  // the state before the step
          wave_lds_fence();
          quad_slab_store<M, D>(region, (float*)prm.obs_out + ((int64_t)t * prm.batch + g0) * G::N, ngames, lane);
          wave_lds_fence();
        }
      }
      const int zc = (host_policy == HK_HOST_ZEILLINGER && leader) ? zeillinger_game<float>(mine, prm.m, prm.d) : 0;
      fast_policy<D>(seed, host_policy, agent_policy, gg, step0 + (uint32_t)t, pcache, cls, axis, mask, zc);
      const bool prev_done = np < 2;
      if (leader) {
        for (int k = 0; k < prm.d; ++k) cs[k] = (float)((mask >> k) & 1u);
        stages_game<float>(mine, prm.m, prm.d, cs, axis, pad, stages, flags);
        np = num_points<float>(mine, prm.m, prm.d);
      }
      const bool done = np < 2;
      if (done && length < 0) length = t + 1;
      if constexpr (REC) {
        if (leader) {
          const int64_t at = (int64_t)t * prm.batch + g;
          if (prm.r_host_class_out) prm.r_host_class_out[at] = cls;
          if (prm.r_axis_out) prm.r_axis_out[at] = axis;
          if (prm.r_done_out) prm.r_done_out[at] = done;
          if (prm.r_reward_out) prm.r_reward_out[at] = prm.reward_sign * (float)(done && !prev_done);
        }
      }
      if (prm.count_ws) {
        const unsigned long long bd = __ballot(leader && done);
        if (lane == 0 && bd)
          count_add(prm.count_ws + (size_t)(t + 1) * prm.count_stride + blockIdx.x, (uint32_t)__popcll(bd));
      }
    }
    if (leader && last_episode && prm.game_length_out) prm.game_length_out[g] = length;
    wave_lds_fence();
    if (last_episode) {
      quad_slab_store<M, D>(region, (float*)prm.out + g0 * G::N, ngames, lane);
      return;
    }
    wave_lds_fence();
    continue;  // (EPI: the next episode reads its slab again)
  }

  // ---- compaction: every live row to the slot of its rank (the compact image lies over the slab image: every lane
  // holds its rows in registers by now), its index in the game next to it -------------------------------------------------
  {
    int rank = below;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const bool live = (lmask >> r) & 1u;
      if (live) {
        float* dst = cmine + rank * CW;
        if constexpr (D == 4) {
          *reinterpret_cast<vf4*>(dst) = vf4{rows[r * D], rows[r * D + 1], rows[r * D + 2], rows[r * D + 3]};
        } else if constexpr (D == 3) {
          *reinterpret_cast<vf4*>(dst) = vf4{rows[r * D], rows[r * D + 1], rows[r * D + 2], 0.0f};
        } else {
#pragma unroll
          for (int k = 0; k < D; ++k) dst[k] = rows[r * D + k];
        }
        tmine[rank] = (uint8_t)(i0 + r);
      }
      rank += live ? 1 : 0;
    }
  }
  smax = R;
#pragma nounroll
  while (smax > 1 && !__any(np > kQuad * (smax - 1))) --smax;
  wave_lds_fence();

  // ---- my slots: ranks j, j + 4, ... up to the wave's smax; slots past the game's live rows are holes -------------------
#pragma unroll
  for (int e = 0; e < R * D; ++e) q[e] = INFINITY;
  unrolled_while<0, R>([&](auto sc) {
    constexpr int s = decltype(sc)::value;
    if (s >= smax) return false;
    const bool has = kQuad * s + j < np;
    const int r = kQuad * s + j < M ? kQuad * s + j : 0;
    const float* src = cmine + r * CW;
    if constexpr (D == 4 || D == 3) {
      const vf4 v = *reinterpret_cast<const vf4*>(src);
      q[s * D] = has ? v.x : INFINITY;
      q[s * D + 1] = has ? v.y : INFINITY;
      q[s * D + 2] = has ? v.z : INFINITY;
      if constexpr (D == 4) q[s * D + 3] = has ? v.w : INFINITY;
    } else {
#pragma unroll
      for (int k = 0; k < D; ++k) {
        const float v = src[k];
        q[s * D + k] = has ? v : INFINITY;
      }
    }
    return true;
  });
  wave_lds_fence();
  }  // (!GEN)
  if constexpr (ZEIL && !GEN) {
    if (prm.class_out) {
      // hk_zeillinger (jax/players.py:55-109) as its own operator: Zeillinger's class of every game of the batch, the
      // launch's only product -- the rollout kernel's prologue (slab in, live rows to their slots) and its pair loop
      const int zc = qr_zeillinger<M, CW, R, D>(q, cmine, j, smax, np);
      if (leader) (prm.class_out + g0)[(unsigned)gi] = zc;
      return;
    }
  }
  if (!active) np = 2;  // never finished, never counted
  int length = (np < 2) ? 0 : -1;

  // ---- the steps: a staircase of loops, one per bucket of slots per lane ---------------------------------------------
  const bool want_obs = REC && prm.obs_out != nullptr;
  const bool want_records = REC && (prm.r_host_class_out || prm.r_axis_out || prm.r_done_out || prm.r_reward_out);
  // The small records are a function of the window of action bytes and of the game's first finished step, so no step
  // stores them: they are written per window (before it moves, and at the end) -- lane = (step within a group of four,
  // game of the wave), one byte of the window per lane, 64-B runs per step and field instead of one 16-lane store per
  // step, field and wave.  (Round 2 stored them inside the step: 80 partial stores per wave and episode.)
  int32_t* rec_cls = REC ? prm.r_host_class_out : nullptr;
  int32_t* rec_axis = REC ? prm.r_axis_out : nullptr;
  uint8_t* rec_done = REC ? prm.r_done_out : nullptr;
  float* rec_reward = REC ? prm.r_reward_out : nullptr;
  float* rec_obs = REC ? (float*)prm.obs_out : nullptr;
  int64_t rec_batch = prm.batch;
  float rec_sign = prm.reward_sign;
  if constexpr (REC) asm volatile("" : "+s"(rec_cls), "+s"(rec_axis), "+s"(rec_done), "+s"(rec_reward), "+s"(rec_obs),
                                  "+s"(rec_batch), "+s"(rec_sign));
  int t_rec = 0;  // the records of steps < t_rec are written
  auto flush_records = [&](int t_end, int len_now) {
    if (want_records && t_rec < t_end) {
      const int gsel = lane & (kQuadGames - 1), sub = lane >> 4;
      const int lg = __shfl(len_now, gsel * kQuad);  // the game's first finished step (0: from the start, -1: not yet)
      const bool gok = gsel < ngames;
      const int64_t off = (int64_t)sub * rec_batch + gsel;
      const uint8_t* w = act + (int)(step0 + (uint32_t)t_rec - (wb0 << 2)) * kQuadGames + lane;
      for (int ta = t_rec; ta < t_end; ta += 4, w += 4 * kQuadGames) {
        const int tt = ta + sub;
        if (gok && tt < t_end) {
          const uint32_t a = *w;
          const int64_t row = (int64_t)ta * rec_batch + g0;  // (scalar)
          const bool done = lg >= 0 && tt + 1 >= lg;
          if (rec_cls) (rec_cls + row)[off] = encode_mask(a & 31u);
          if (rec_axis) (rec_axis + row)[off] = (int32_t)(a >> 5);
          if (rec_done) (rec_done + row)[off] = done;
          if (rec_reward) (rec_reward + row)[off] = rec_sign * (float)(lg >= 1 && tt + 1 == lg);
        }
      }
    }
    t_rec = t_end;
  };
  int t = 0;
  bool stop = false;
  // (EPI: the last episode's final state; GEN: if anybody asks)
  const bool publish = !EPI || (last_episode && prm.out != nullptr);
  bool published = !publish;
  // list semantics / COMPACT_SORTED (a run-time configured plain rollout of its own, kHotList -- the ranks cost the
  // widest level registers the other configurations do not have to spare): sorted + compacted when published
  constexpr bool kEndSort = HOT == kHotList;
  static_assert(!kEndSort || (!REC && !ZEIL && !GEN && !EPI), "sorted output: plain rollouts");
  const bool end_sort = kEndSort && nsteps > 0 && (stages & HK_STAGE_NEWTON) &&
                        ((flags & HK_SEM_MASK) == HK_SEM_LIST || (flags & HK_FLAG_COMPACT_SORTED));
  bool rescale_pending = false;
  PolicyCache zcache;  // (ZEIL: the lane's own Philox block, one per four steps)
  const uint64_t gg_game =  // (ZEIL) the policy stream's index of the game this lane plays
      ZEIL ? prm.game_offset + (has_ids ? (uint64_t)(uint32_t)__shfl((int)raw_id, gi) : (uint64_t)g) : 0;
  if (nsteps <= 0 && !published) {  // (no step: no level of the staircase is entered)
    qr_build_image<M, D, R, R>(q, tags, region, smax, pad, j, lane);
    published = true;
  }
  while (t < nsteps && !stop) {  // one pass per window of actions (episodes of up to 24 steps: one pass)
    if (!ZEIL && (uint32_t)((step0 + (uint32_t)t) >> 2) - wb0 >= (uint32_t)kQrBlocks) {
      if constexpr (REC) flush_records(t, length);
      wave_lds_fence();
      wb0 = (step0 + (uint32_t)t) >> 2;
      asm volatile("" : "+s"(wb0));  // (the refill's Philox rounds stay in here: their invariant parts were hoisted and spilled)
      const uint32_t nb = wb_last - wb0 + 1u;
      // (the fill game's stream index again, from the lane: kept in registers from the first fill it was spilled)
      int lane_r = lane;
      asm volatile("" : "+v"(lane_r));
      const int fill_r = lane_r & (kQuadGames - 1);
      uint32_t id_r = 0;  // (the id is loaded again too: refills are rare, a register through all the loops is not)
      if (has_ids && fill_r < ngames) id_r = (uint32_t)prm.game_ids[g0 + fill_r];
      const uint64_t gg_refill = prm.game_offset + (has_ids ? (uint64_t)id_r : (uint64_t)(g0 + fill_r));
      qr_policy_fill<D>(act, gg_refill, wb0, (int)(nb < (uint32_t)kQrBlocks ? nb : (uint32_t)kQrBlocks), seed, host_policy,
                        agent_policy, lane_r);
      wave_lds_fence();
    }
    const uint32_t wstep0 = wb0 << 2;                       // first step of the window
    const uint32_t wleft = wstep0 + (uint32_t)kQrSteps - step0;  // steps (from 0) the window reaches
    const int tw = (!ZEIL && wleft < (uint32_t)nsteps) ? (int)wleft : nsteps;
    const uint8_t* arow = act + gi;
    // the action byte of step t, requested one step ahead (the LDS round trip opened every step: hk_duo_kernel.h)
    uint32_t a_next = ZEIL ? 0u : arow[(int)(step0 + (uint32_t)t - wstep0) * kQuadGames];
    QuadLevels<M, D, R>::run([&](auto nbc, auto loc) {
      constexpr int NB = decltype(nbc)::value, LO = decltype(loc)::value;
      // (no s_setprio by bucket here: with three or four waves per SIMD it starves the others -- measured 24.4 -> 26.5 us
      // at (20,3) x 65 536, 192 -> 202 us at (50,4) x 262 144; hk_duo_kernel.h, two waves per SIMD, gains 3 % from it)
      while (t < tw && (smax > LO || LO == 0) && !stop) {  // (a wave of empty games has smax 1: the last loop's)
        uint32_t cmask;
        int axis;
        if constexpr (ZEIL) {
          // (a game with fewer than two rows has no pair: class 0 -- a wave of finished games skips the test)
          const int zc = __any(active && np >= 2) ? qr_zeillinger<M, CW, R, D>(q, cmine, j, smax, M) : 0;
          uint32_t ra, rb;
          int cls;
          policy_words(gg_game, step0 + (uint32_t)t, seed, zcache, D, ra, rb);
          policy_from_words<D>(ra, rb, host_policy, agent_policy, cls, axis, cmask, zc);
        } else {
          const uint32_t a = a_next;
          a_next = arow[(int)(step0 + (uint32_t)t + 1u - wstep0) * kQuadGames];  // (past the window: not used)
          cmask = a & 31u;
          axis = (int)(a >> 5);
        }
        if constexpr (REC) {
          if (want_obs) {  // the state before the step
            qr_build_image<M, D, R, NB>(q, tags, region, smax, pad, j, lane);
#if !defined(HK_QR_EXP) || HK_QR_EXP != 2
            // (ordinary stores: 315 MB of non-temporal stores in this pattern take 61.6 us where ordinary ones take
            // 55.5 and a plain fill 47 -- scripts/probe_obs_pattern.py; the L2 write-back combines, the stream does not)
            quad_slab_store<M, D, false>(region, rec_obs + ((int64_t)t * rec_batch + g0) * G::N, ngames, lane);
#endif
            wave_lds_fence();
          }
        }
#if defined(HK_QR_EXP) && HK_QR_EXP == 1
        if (!REC)
#endif
        // (end_sort: the last step's rescale waits until the rows are ranked, at the publish)
        const unsigned st = (kEndSort && end_sort && t + 1 == nsteps) ? (stages & ~(unsigned)HK_STAGE_RESCALE) : stages;
        if constexpr (kEndSort) rescale_pending = end_sort && t + 1 == nsteps && (stages & HK_STAGE_RESCALE);
        np = qr_stages<M, CW, R, D, NB>(q, cmask, axis, np, j, flags, st, cmine, smax);
        if (!active) np = 2;
        const bool done = np < 2;
        if (done && length < 0) length = t + 1;
        if constexpr (NB == 1) {
          // Fixed point (hk_duo_kernel.h): once every game of the wave is down to one point at the origin (or none)
          // nothing changes any more
          if (t + 1 < nsteps && !__any(active && !done)) {
            bool still = true;  // (one slot per lane: a row of the game's, a hole, or nothing)
#pragma unroll
            for (int k = 0; k < D; ++k) still &= (q[k] == 0.0f);
            still |= !(q[0] < INFINITY);
            if (!__any(active && !still)) stop = true;
          }
        } else {
          // re-deal the rows when the widest game of the wave fits fewer slots per lane
          if (t + 1 < nsteps && !__any(active && np > kQuad * (smax - 1))) {
            int snew = smax - 1;
#pragma nounroll
            while (snew > 1 && !__any(active && np > kQuad * (snew - 1))) --snew;
            qr_redeal<M, CW, R, D, NB>(q, cmine, tmine, j, active ? np : 0, snew);
            smax = snew;
          }
        }
        ++t;
      }
      // ---- publish, on the level the episode ends on: padding everywhere, the survivors at their own rows.  (After the
      // staircase it would read every slot of every level: the whole row array stayed live through all the loops --
      // at (50,4), 52 + 13 registers under a 168-register budget: every lane spilled 212 - 332 B.) -------------------------
      if (!published && (t >= nsteps || stop)) {
        if (kEndSort && end_sort) qr_build_sorted<M, CW, R, D, NB>(q, region, cmine, smax, pad, j, lane, rescale_pending, flags);
        else qr_build_image<M, D, R, NB>(q, tags, region, smax, pad, j, lane);
        published = true;
      }
    });
  }
  // (scalar base + the lane's game index: the 64-bit index g need not live through the loops)
  if (leader && last_episode && prm.game_length_out) (prm.game_length_out + g0)[(unsigned)gi] = length;
  if constexpr (REC) {
    // every game of the wave at its fixed point before the last step: the state does not change any more, the
    // observations go on (the same image); the records of a window are written before it moves
    for (; t < nsteps; ++t) {
      if ((uint32_t)((step0 + (uint32_t)t) >> 2) - wb0 >= (uint32_t)kQrBlocks) {
        flush_records(t, length);
        wave_lds_fence();
        wb0 = (step0 + (uint32_t)t) >> 2;
        const uint32_t nb = wb_last - wb0 + 1u;
        qr_policy_fill<D>(act, gg_fill, wb0, (int)(nb < (uint32_t)kQrBlocks ? nb : (uint32_t)kQrBlocks), seed, host_policy,
                          agent_policy, lane);
        wave_lds_fence();
      }
      if (want_obs)
        quad_slab_store<M, D, false>(region, rec_obs + ((int64_t)t * rec_batch + g0) * G::N, ngames, lane);
    }
    flush_records(nsteps, length);
  }
  // (non-temporal: the final state is written once and not read again by the launch -- kept out of the XCD's L2 it
  // leaves the next episode's initial states there, hk_duo_kernel.h)
  if (publish) {
    float* outp = (float*)prm.out;
    if constexpr (EPI) asm volatile("" : "+s"(outp));  // (inside the episode loop: the chunks' 64-bit addresses are not hoisted out of it)
    quad_slab_store<M, D, !REC>(region, outp + g0 * G::N, ngames, lane);
  }
  // the finished-game counts: games whose first finished step is <= s, for every s (a finished game stays finished)
  if (prm.count_ws) add_length_counts(prm.count_ws + blockIdx.x, prm.count_stride, 0, nsteps, leader, length, lane);
  if (last_episode) break;
  wave_lds_fence();  // (the next episode's action window and staging image)
  }  // episodes
}

// ---- host side -------------------------------------------------------------------------------------------------------
inline bool quadroll_sorted_output(const Params& prm) {
  return (prm.stages & HK_STAGE_NEWTON) &&
         ((prm.flags & HK_SEM_MASK) == HK_SEM_LIST || (prm.flags & HK_FLAG_COMPACT_SORTED));
}

// rollouts of float32, contiguous, W-aligned records
inline bool quadroll_request_ok(const Params& prm) {
  if (prm.mode != kModeRollout || prm.m > 255) return false;
  // Zeillinger's host: plain rollouts (quadroll_kernel<..., ZEIL>)
  if (prm.host_policy == HK_HOST_ZEILLINGER &&
      (prm.obs_out || prm.r_host_class_out || prm.r_axis_out || prm.r_done_out || prm.r_reward_out))
    return false;
  // sorted + compacted output (list semantics): plain rollouts only -- ranked once, at the publish --, not under
  // Zeillinger's host, whose tie-breaks follow the physical row order
  if (quadroll_sorted_output(prm) &&
      (prm.max_value > 0 || prm.episodes > 1 || prm.host_policy == HK_HOST_ZEILLINGER || prm.obs_out ||
       prm.r_host_class_out || prm.r_axis_out || prm.r_done_out || prm.r_reward_out))
    return false;
  if (prm.flags & (HK_FLAG_FORCE_GENERIC | HK_FLAG_FORCE_TEAM | HK_FLAG_FORCE_ONE_LANE | HK_FLAG_FORCE_TWO_LANES))
    return false;
  return true;
}

template <int M, int D>
int launch_quadroll_t(Params prm, hipStream_t stream) {
  constexpr int WPB = QuadRollGeom<M, D>::kWpb;
  const int64_t waves = ((int64_t)prm.batch + kQuadGames - 1) / kQuadGames;
  const unsigned grid = (unsigned)((waves + WPB - 1) / WPB);
  prm.games_per_block = kQuadGames * WPB;
  prm.pad_f32 = (float)prm.pad;
  launch_prepare();
  const int hot = fast_hot_config(prm);
  if (prm.obs_out || prm.r_host_class_out || prm.r_axis_out || prm.r_done_out || prm.r_reward_out) {
    if (prm.obs_out && reinterpret_cast<uintptr_t>(prm.obs_out) % (QuadGeom<M, D>::W * 4)) return HK_ERR_ALIGN;
    if (hot == kHotJax)
      hipLaunchKernelGGL((quadroll_kernel<M, D, kHotJax, WPB, true>), dim3(grid), dim3(kWave * WPB), 0, stream,
                         (const float*)prm.in, prm.in_stride, prm.batch, prm);
    else
      hipLaunchKernelGGL((quadroll_kernel<M, D, kHotNone, WPB, true>), dim3(grid), dim3(kWave * WPB), 0, stream,
                         (const float*)prm.in, prm.in_stride, prm.batch, prm);
    return launch_status();
  }
  if (prm.host_policy == HK_HOST_ZEILLINGER)
    hipLaunchKernelGGL((quadroll_kernel<M, D, kHotNone, WPB, false, true>), dim3(grid), dim3(kWave * WPB), 0, stream,
                       (const float*)prm.in, prm.in_stride, prm.batch, prm);
  else if (hot == kHotJax)
    hipLaunchKernelGGL((quadroll_kernel<M, D, kHotJax, WPB>), dim3(grid), dim3(kWave * WPB), 0, stream,
                       (const float*)prm.in, prm.in_stride, prm.batch, prm);
  else if (hot == kHotTorch)
    hipLaunchKernelGGL((quadroll_kernel<M, D, kHotTorch, WPB>), dim3(grid), dim3(kWave * WPB), 0, stream,
                       (const float*)prm.in, prm.in_stride, prm.batch, prm);
  else if (quadroll_sorted_output(prm))
    hipLaunchKernelGGL((quadroll_kernel<M, D, kHotList, WPB>), dim3(grid), dim3(kWave * WPB), 0, stream,
                       (const float*)prm.in, prm.in_stride, prm.batch, prm);
  else
    hipLaunchKernelGGL((quadroll_kernel<M, D, kHotNone, WPB>), dim3(grid), dim3(kWave * WPB), 0, stream,
                       (const float*)prm.in, prm.in_stride, prm.batch, prm);
  return launch_status();
}

// hk_zeillinger on the rollout kernel's prologue + pair loop (JAX variant, contiguous records, the large games: the
// small shapes' one-lane kernel is ahead there)
template <int M, int D>
int launch_quadzeil_t(Params prm, hipStream_t stream) {
  constexpr int WPB = QuadRollGeom<M, D>::kWpb;
  const int64_t waves = ((int64_t)prm.batch + kQuadGames - 1) / kQuadGames;
  const unsigned grid = (unsigned)((waves + WPB - 1) / WPB);
  prm.pad_f32 = -1.0f;
  prm.pad = -1.0;
  prm.mode = kModeRollout;
  prm.steps = 0;
  prm.host_policy = HK_HOST_ZEILLINGER;
  prm.flags = HK_SEM_JAX;
  prm.stages = 0;
  prm.count_ws = nullptr;
  prm.episodes = 1;
  launch_prepare();
  hipLaunchKernelGGL((quadroll_kernel<M, D, kHotNone, WPB, false, true>), dim3(grid), dim3(kWave * WPB), 0, stream,
                     (const float*)prm.in, prm.in_stride, prm.batch, prm);
  return launch_status();
}

inline bool quadroll_supported(const Params& prm, int dtype) {
  if (dtype != HK_F32 || !quadroll_request_ok(prm)) return false;
#define HK_X(M_, D_) if (prm.m == M_ && prm.d == D_) return quad_ok_t<M_, D_>(prm);
  HK_QUAD_SPECS(HK_X)
#undef HK_X
  return false;
}

// ---- rollouts from initial states drawn inside the launch (hk_rollout_desc.gen_max_value) ------------------------------
// plain rollouts (no records), in-place order, a duplicate fill equal to the padding value (there is no exactness guard
// to send anything else down the generic routines: the draws are canonical by construction); Zeillinger's host where
// this kernel is its rollout kernel anyway (the large games)
template <int M, int D>
int launch_quadroll_gen_t(Params prm, hipStream_t stream) {
  constexpr int WPB = QuadRollGeom<M, D>::kWpb;
  const int64_t waves = ((int64_t)prm.batch + kQuadGames - 1) / kQuadGames;
  const unsigned grid = (unsigned)((waves + WPB - 1) / WPB);
  prm.games_per_block = kQuadGames * WPB;
  prm.pad_f32 = (float)prm.pad;
  launch_prepare();
  const int hot = fast_hot_config(prm);
  if (prm.max_value <= 0) {  // episodes back to back from the states in memory (EPI without GEN)
    if (prm.host_policy == HK_HOST_ZEILLINGER) return HK_ERR_UNSUPPORTED;
    if (hot == kHotJax)
      hipLaunchKernelGGL((quadroll_kernel<M, D, kHotJax, WPB, false, false, false, true>), dim3(grid), dim3(kWave * WPB), 0,
                         stream, (const float*)prm.in, prm.in_stride, prm.batch, prm);
    else if (hot == kHotTorch)
      hipLaunchKernelGGL((quadroll_kernel<M, D, kHotTorch, WPB, false, false, false, true>), dim3(grid), dim3(kWave * WPB), 0,
                         stream, (const float*)prm.in, prm.in_stride, prm.batch, prm);
    else
      hipLaunchKernelGGL((quadroll_kernel<M, D, kHotNone, WPB, false, false, false, true>), dim3(grid), dim3(kWave * WPB), 0,
                         stream, (const float*)prm.in, prm.in_stride, prm.batch, prm);
    return launch_status();
  }
  if (prm.host_policy == HK_HOST_ZEILLINGER) {
    if constexpr (M > 32)
      hipLaunchKernelGGL((quadroll_kernel<M, D, kHotNone, WPB, false, true, true>), dim3(grid), dim3(kWave * WPB), 0, stream,
                         (const float*)nullptr, prm.in_stride, prm.batch, prm);
    else
      return HK_ERR_UNSUPPORTED;
  } else if (hot == kHotJax) {
    hipLaunchKernelGGL((quadroll_kernel<M, D, kHotJax, WPB, false, false, true>), dim3(grid), dim3(kWave * WPB), 0, stream,
                       (const float*)nullptr, prm.in_stride, prm.batch, prm);
  } else if (hot == kHotTorch) {
    hipLaunchKernelGGL((quadroll_kernel<M, D, kHotTorch, WPB, false, false, true>), dim3(grid), dim3(kWave * WPB), 0, stream,
                       (const float*)nullptr, prm.in_stride, prm.batch, prm);
  } else {
    hipLaunchKernelGGL((quadroll_kernel<M, D, kHotNone, WPB, false, false, true>), dim3(grid), dim3(kWave * WPB), 0, stream,
                       (const float*)nullptr, prm.in_stride, prm.batch, prm);
  }
  return launch_status();
}

// episodes back to back from the states in memory: plain rollouts the four-lane kernel serves, not Zeillinger's host
inline bool quadroll_episodes_supported(const Params& prm, int dtype) {
  if (prm.max_value > 0 || prm.episodes <= 1 || !prm.in || !prm.out) return false;
  if (prm.obs_out || prm.r_host_class_out || prm.r_axis_out || prm.r_done_out || prm.r_reward_out) return false;
  if (prm.host_policy == HK_HOST_ZEILLINGER) return false;
  return quadroll_supported(prm, dtype);
}

inline bool quadroll_gen_supported(const Params& prm, int dtype) {
  if (dtype != HK_F32 || prm.max_value <= 0 || !quadroll_request_ok(prm)) return false;
  if (prm.obs_out || prm.r_host_class_out || prm.r_axis_out || prm.r_done_out || prm.r_reward_out) return false;
  if (prm.host_policy == HK_HOST_ZEILLINGER && prm.m <= 32) return false;
  const float pad = (float)prm.pad;
  if (!(pad < 0.0f) || ((prm.flags & HK_SEM_MASK) == HK_SEM_JAX && pad != -1.0f)) return false;
#define HK_X(M_, D_) \
  if (prm.m == M_ && prm.d == D_) return reinterpret_cast<uintptr_t>(prm.out) % (QuadGeom<M_, D_>::W * 4) == 0;
  HK_QUAD_SPECS(HK_X)
#undef HK_X
  return false;
}

// where this kernel is the default: the shapes without a two-lane kernel ((50,4): hk::team_kernel's rollouts before)
// ... and, on the small shapes, batches of up to two of its waves per SIMD (32 768 games on an MI355X): measured
// (scripts/probe_rollout_families.py, (20,3)): 14.0 / 15.0 / 17.4 us per 20-step episode at 4 096 / 16 384 / 32 768 games against
// 16.9 / 18.3 / 18.8 on two lanes per game, 24.5 against 22.2 at 65 536
// Recording rollouts: at every size (scripts/probe_records.py, (20,3), per 20-step episode incl. the counter reduce:
// 39.8 against 48.6 us with the small records and 69.5 against 82.1 us with the observations at 65 536 games, 90.6 / 240
// against 114 / 279 us at 262 144).
inline bool quadroll_default(const Params& prm, int simds, bool small_records_elsewhere = false) {
  if (prm.m > 32) return true;
  // Zeillinger's host: the two-lane kernel is ahead at every size it serves (scripts/probe_zeillinger.py, (20,3):
  // 40.9 against 44.3 us at 32 768 games, 37.4 against 38.5 at 8 192, 44.2 against 57.0 at 65 536)
  if (prm.host_policy == HK_HOST_ZEILLINGER) return false;
  if (prm.obs_out) return true;
  if ((prm.r_host_class_out || prm.r_axis_out || prm.r_done_out || prm.r_reward_out) && !small_records_elsewhere) return true;
  const int64_t waves = ((int64_t)prm.batch + kQuadGames - 1) / kQuadGames;
  // short rollouts (measured at (20,3) x 65 536, scripts/probe_short_rollouts.py: 1 / 2 / 4 / 6 steps 8.2 / 10.0 / 12.3 /
  // 14.4 us against the two-lane kernel's 9.6 / 11.5 / 13.6 / 15.0; level at 8 steps): the wave's shorter chain
  // counts while the wide first steps are most of the launch
  if (prm.steps <= 6 && waves <= (int64_t)4 * simds) return true;
  return waves <= (int64_t)2 * simds;
}

// workgroups of a launch (the finished-game workspace has one slot per workgroup and step)
inline int64_t quadroll_grid(const Params& prm) {
#define HK_X(M_, D_)                                                             \
  if (prm.m == M_ && prm.d == D_) {                                               \
    constexpr int WPB = QuadRollGeom<M_, D_>::kWpb;                               \
    return (((int64_t)prm.batch + kQuadGames - 1) / kQuadGames + WPB - 1) / WPB;  \
  }
  HK_QUAD_SPECS(HK_X)
#undef HK_X
  return 0;
}

#ifndef HK_SPEC_TU
#define HK_X(M_, D_) extern template int launch_quadroll_t<M_, D_>(Params, hipStream_t); \
  extern template int launch_quadroll_gen_t<M_, D_>(Params, hipStream_t);           \
  extern template int launch_quadzeil_t<M_, D_>(Params, hipStream_t);
HK_QUAD_SPECS(HK_X)
#undef HK_X

inline int launch_quadzeil(const Params& prm, hipStream_t stream) {
#define HK_X(M_, D_) if (prm.m == M_ && prm.d == D_) return launch_quadzeil_t<M_, D_>(prm, stream);
  HK_QUAD_SPECS(HK_X)
#undef HK_X
  return HK_ERR_UNSUPPORTED;
}

inline int launch_quadroll_gen(const Params& prm, hipStream_t stream) {
#define HK_X(M_, D_) if (prm.m == M_ && prm.d == D_) return launch_quadroll_gen_t<M_, D_>(prm, stream);
  HK_QUAD_SPECS(HK_X)
#undef HK_X
  return HK_ERR_UNSUPPORTED;
}

inline int launch_quadroll(const Params& prm, hipStream_t stream) {
#define HK_X(M_, D_) if (prm.m == M_ && prm.d == D_) return launch_quadroll_t<M_, D_>(prm, stream);
  HK_QUAD_SPECS(HK_X)
#undef HK_X
  return HK_ERR_UNSUPPORTED;
}
#endif

}  // namespace hk
