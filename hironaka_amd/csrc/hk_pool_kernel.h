// Pool rollouts: a workgroup keeps 256 games in LDS and re-deals the LIVE ones to its waves between steps.
//
// hk::duo_kernel gives every wave 32 games for the whole episode.  The mean game of BASELINE configs[1] lasts five
// steps and holds six rows at the start, but a wave runs until its LONGEST game is over (~13 of 20 steps) in the bucket
// of its WIDEST one (36 pair tests per lane where the average game needs 16): of the 6.8 M wave instructions of a
// 65 536-game episode about 0.8 M are needed by a schedule that spends nothing on finished games and runs every game in
// its own bucket (tests/model in DESIGN.md section 5).  The state cannot move between lanes while it lives in
// registers, so this kernel parks it in LDS between ROUNDS:
//   * a workgroup = 8 waves = 256 games (61 KB of images at (20,3)); every wave loads, scans and finally stores its 32
//     "home" games exactly like hk::duo_kernel (one contiguous slab, all requests in flight at once);
//   * a round = a few steps of the two-lane staircase (hk_duo_kernel.h: d_stages, re-deals inside the wave when its
//     widest game narrows) on the 32 games a wave currently holds; at the end of the round the rows go back to the
//     game's image (removed rows as padding: the image is a valid state at every round boundary);
//   * between two rounds the workgroup sorts its live games -- a game that reached its fixed point (no point, or one
//     point at the origin: no subset / axis changes it any more) is dropped for good -- by bucket, widest first, with
//     one ballot per bucket class, one LDS atomic per wave and class and two barriers; wave w takes positions
//     [32 w, 32 w + 32) of that order: waves whose range is empty skip the round, the others run games of one bucket;
//   * rounds end after steps 1, 2, 3, 5, 8, 12 (the buckets change fastest at the start); once 32 or fewer games
//     are alive one wave runs them to the end of the episode without further barriers;
//   * the finished-game counts: a game's FIRST finished step goes into an LDS histogram whose prefix sums are added
//     to the caller's workspace once per workgroup (a finished game stays finished: rows are only ever removed);
//   * Philox stays keyed by the GLOBAL game index and the step, so results do not depend on which lane ran a game:
//     bit-identical to every other kernel family (tests force this one with HK_FLAG_FORCE_POOL).
// The exactness guard is the two-lane kernel's; a workgroup with a non-canonical game runs the exact generic
// routines wave by wave on the home games (no pool).
#pragma once

#include "hk_duo_kernel.h"

namespace hk {

constexpr int kPoolWaves = 8;
constexpr int kPoolGames = kPoolWaves * kDuoGames;  // 256
constexpr int kPoolThreads = kPoolWaves * kWave;    // 512
constexpr int kPoolMaxSteps = 255;                  // first-finished histogram in LDS
constexpr int kPoolClasses = 6;                     // bucket classes of the sort: slots per lane 1..5, >= 6

// LDS traffic between lanes of ONE wave needs no s_barrier (a wave's DS operations execute in order); the compiler
// must not move memory operations across the hand-over, though
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// duo_scatter that also writes the padding row over the slots of rows removed since the gather: the image stays a
// valid state.  Returns the mask of the GAME's slots still alive.
template <int M, int CH, int D, int SB = CH>
__device__ __forceinline__ uint32_t pool_scatter(const float (&q)[CH * D], float* mine, uint32_t mask, int smax, int h,
                                                 float pad) {
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
    if (has) {
      const bool live = q[s * D] < INFINITY;
      alive |= live ? (1u << slot) : 0u;
      float* row = mine + slot * D;
#pragma unroll
      for (int k = 0; k < D; ++k) row[k] = live ? q[s * D + k] : pad;
    }
    return true;
  });
  return alive | (uint32_t)duo_other_i((int)alive);
}

// end of the round that starts at step t: block-uniform.  HK_POOL_SCHEDULE 0: one sort, before the first step (round 0
// runs no step), then every wave keeps its games; 1: rounds end after steps 1, 2, 3, 5, 8, 12 while more than a
// wave's worth of games is alive
#ifndef HK_POOL_SCHEDULE
#define HK_POOL_SCHEDULE 0
#endif
__device__ __forceinline__ int pool_round_end(int t, int nsteps, int live, int round) {
  int e = nsteps;
#if HK_POOL_SCHEDULE == 0
  if (round == 0) e = 0;
#else
  if (live > kDuoGames) e = (t < 3) ? t + 1 : (t < 5) ? 5 : (t < 8) ? 8 : (t < 12) ? 12 : nsteps;
#endif
  return e < nsteps ? e : nsteps;
}

template <int M, int D, int HOT>
__global__ __launch_bounds__(kPoolThreads, 2) void pool_kernel(const float* in0, int64_t in_stride0, int batch0,
                                                               const Params prm) {
  using G = FastGeom<M, D>;
  constexpr int CH = DuoGeom<M, D>::CH;
  static_assert(M <= 32, "the live mask of a game travels as 32 bits");
  __shared__ __align__(16) float lds[kPoolGames * G::S];
  __shared__ __align__(16) uint32_t m_phil[kPoolGames * 2 * 4];  // the pair's Philox blocks between rounds
  __shared__ uint32_t m_mask[kPoolGames];                        // live slots of a game's image
  __shared__ int32_t m_len[kPoolGames];                          // first finished step (-1: not yet)
  __shared__ int32_t order[kPoolGames];                          // live games, widest bucket first
  __shared__ uint32_t ctr[2][8];                                 // games per bucket class (double-buffered)
  __shared__ uint32_t hist[kPoolMaxSteps + 1];                   // games first finished after step s - 1 (0: at entry)
  __shared__ float cbuf[kPoolGames * D];                         // slow path only
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & (kWave - 1);
  const int h = lane & 1, gi = lane >> 1;
  const int64_t gb = (int64_t)blockIdx.x * kPoolGames;
  const int64_t g0 = gb + (int64_t)wave * kDuoGames;  // the wave's home slab
  const int64_t left = (int64_t)batch0 - g0;
  const int ngames = (int)(left < 0 ? 0 : (left < kDuoGames ? left : kDuoGames));
  const bool active = gi < ngames;
  float* wimg = lds + wave * kDuoGames * G::S;
#ifdef HK_POOL_PROBE  // dev builds (scripts/build_probe.sh): a time line per wave, written over game_length_out
  int probe_n = 0;
  int32_t probe_v[32];
#define HK_POOL_STAMP(tag) do { if (probe_n < 31) { probe_v[probe_n++] = (int32_t)wall_clock64(); probe_v[probe_n++] = (tag); } } while (0)
  HK_POOL_STAMP(0);
#else
#define HK_POOL_STAMP(tag) do { } while (0)
#endif
  DuoSlabRegs<M, D> slab;
  if (ngames > 0) duo_slab_issue<M, D>(slab, in0, in_stride0, g0, ngames, lane);
  const float pad = (float)prm.pad;
  const unsigned flags = (HOT == kHotJax) ? (unsigned)HK_SEM_JAX : (HOT == kHotTorch) ? kHotTorchFlags : prm.flags;
  const unsigned stages = HOT ? (unsigned)(HK_STAGE_SHIFT | HK_STAGE_REPOSITION | HK_STAGE_NEWTON) : prm.stages;
  const float fill = ((flags & HK_SEM_MASK) == HK_SEM_JAX) ? -1.0f : pad;
  const int nsteps = prm.steps;
  for (int s = tid; s <= kPoolMaxSteps; s += kPoolThreads) hist[s] = 0;
  if (tid < 16) (&ctr[0][0])[tid] = 0;
  if (ngames > 0) duo_slab_commit<M, D>(slab, wimg, ngames, lane);
  wave_sync();

  // ---- live rows, exactness guard (both lanes of a pair scan the whole home game) ------------------------------
  const int home = wave * kDuoGames + gi;
  uint32_t gmask;
  bool ok;
  scan_image<M, D>(lds + home * G::S, fill, gmask, ok);
  if (!active) {
    gmask = 0;
    ok = true;
  }
  int np = mask_pop(gmask);
  int nmax = wave_max(np, M);
  const bool exact = __syncthreads_and((fill == pad) && __all(ok) && nmax <= G::C) != 0;  // (hist / ctr are zero now)

  if (!exact) {
    // ---- slow path (whole workgroup, wave by wave on the home games): the pair's first lane runs the exact generic
    // routines on the image (hk_duo_kernel.h) -------------------------------------------------------------------
    float* mine = lds + home * G::S;
    float* cs = cbuf + home * D;
    const bool leader = active && h == 0;
    const uint64_t gg = prm.game_offset + (uint64_t)(gb + home);
    PolicyCache pcache;
    np = leader ? num_points<float>(mine, M, D) : 2;
    int length = (np < 2) ? 0 : -1;
    if (prm.count_ws) {
      const unsigned long long b0 = __ballot(leader && np < 2);
      if (lane == 0 && b0) count_add(prm.count_ws + blockIdx.x, (uint32_t)__popcll(b0));
    }
    for (int t = 0; t < nsteps; ++t) {
      int axis = -1, cls = 0;
      uint32_t mask;
      const int zc = (prm.host_policy == HK_HOST_ZEILLINGER && leader) ? zeillinger_game<float>(mine, prm.m, prm.d) : 0;
      fast_policy<D>(prm, gg, prm.step_offset + (uint32_t)t, pcache, cls, axis, mask, zc);
      if (leader) {
        for (int k = 0; k < prm.d; ++k) cs[k] = (float)((mask >> k) & 1u);
        stages_game<float>(mine, prm.m, prm.d, cs, axis, pad, stages, flags);
        np = num_points<float>(mine, prm.m, prm.d);
      }
      const bool done = np < 2;
      if (done && length < 0) length = t + 1;
      if (prm.count_ws) {
        const unsigned long long bd = __ballot(leader && done);
        if (lane == 0 && bd)
          count_add(prm.count_ws + (size_t)(t + 1) * prm.count_stride + blockIdx.x, (uint32_t)__popcll(bd));
      }
    }
    if (leader && prm.game_length_out) prm.game_length_out[gb + home] = length;
    wave_sync();
    if (ngames > 0) duo_store_slab<M, D>(wimg, (float*)prm.out, prm.out_stride, g0, ngames, lane);
    return;
  }

  HK_POOL_STAMP(1);
  // ---- the rounds ---------------------------------------------------------------------------------------------
  const bool counting = prm.count_ws != nullptr;
  uint32_t step0 = prm.step_offset;
  uint64_t seed = prm.seed;
  int host_policy = HOT ? (int)HK_HOST_RANDOM : prm.host_policy;
  int agent_policy = (HOT == kHotJax) ? (int)HK_AGENT_RANDOM
                                      : (HOT == kHotTorch) ? (int)HK_AGENT_RANDOM_LEGAL : prm.agent_policy;
  if (HOT)
    asm volatile("" : "+s"(step0), "+s"(seed));
  else
    asm volatile("" : "+s"(step0), "+s"(seed), "+s"(host_policy), "+s"(agent_policy));
  int j = home;          // the game (index within the workgroup) this pair holds in the current round
  bool have = active;    // ... if it holds one
  int length = (have && np < 2) ? 0 : -1;
  if (counting) {
    const unsigned long long b0 = __ballot(have && h == 0 && np < 2);
    if (lane == 0 && b0) atomicAdd(&hist[0], (uint32_t)__popcll(b0));
  }
  DuoPolicyCache dcache;
  float c[D];
#pragma unroll
  for (int k = 0; k < D; ++k) c[k] = 0.0f;
  int t0 = 0;               // first step of the current round (block-uniform)
  int live = kPoolGames;    // live games of the workgroup (block-uniform; an upper bound in round 0)
  for (int round = 0;; ++round) {
    const int tend = pool_round_end(t0, nsteps, live, round);
    bool still = true;  // the pair's game is at its fixed point (or the pair holds none)
    if (tend == t0) {
      // a round without steps (the sort before the first step): the images are as loaded
      still = !have || np == 0;
      if (have && h == 0) {
        m_mask[j] = gmask;
        m_len[j] = length;
      }
    } else if (__any(have)) {
      const bool leader = have && h == 0;
      const uint64_t gg = prm.game_offset + (uint64_t)(gb + j);
      float* mine = lds + j * G::S;
      np = have ? mask_pop(gmask) : 0;
      nmax = wave_max(np, M);
      int smax = (nmax + 1) >> 1;
      float q[CH * D];
#pragma unroll
      for (int e = 0; e < CH * D; ++e) q[e] = INFINITY;
      duo_gather<M, CH, D>(q, mine, gmask, smax, h);
      if (!have) np = 2;  // never finished, never counted
      int t = t0;
      bool stop = false;
      static_assert(CH <= 6 || CH % 2 == 0, "bucket ladder: 1..6, then even numbers");
      DuoLevels<CH>::run([&](auto nbc, auto loc) {
        constexpr int NB = decltype(nbc)::value, LO = decltype(loc)::value;
        while (t < tend && (smax > LO || LO == 0) && !stop) {
          int axis, cls;
          uint32_t mask, ra, rb;
          duo_policy_words(gg, step0 + (uint32_t)t, seed, dcache, h, ra, rb);
          policy_from_words<D>(ra, rb, host_policy, agent_policy, cls, axis, mask, 0);
          np = d_stages<CH, D, NB, true>(q, c, axis, np, h, flags, stages, mask);
          if (!have) np = 2;
          const bool done = np < 2;
          const bool first = done && length < 0;
          if (first) length = t + 1;
          if (counting) {
            const unsigned long long bf = __ballot(leader && first);
            if (lane == 0 && bf) atomicAdd(&hist[t + 1], (uint32_t)__popcll(bf));
          }
          if constexpr (NB == 1) {
            // every game of the wave at its fixed point: nothing left to do in this round (or any later one)
            if (t + 1 < tend && !__any(have && !done)) {
              bool fixed = true;  // (one slot per lane: it holds the game's point, a hole, or nothing)
#pragma unroll
              for (int k = 0; k < D; ++k) fixed &= (q[k] == 0.0f);
              fixed |= !(q[0] < INFINITY);
              if (!__any(have && !fixed)) stop = true;
            }
          } else {
            // re-deal the rows when the widest game of the wave fits fewer slots per lane
            if (t + 1 < tend && !__any(have && ((np + 1) >> 1) >= smax)) {
              wave_sync();
              gmask = pool_scatter<M, CH, D, NB>(q, mine, gmask, smax, h, pad);
              wave_sync();
              const int sprev = smax;
              nmax = wave_max(have ? np : 0, 2 * smax - 2);
              smax = (nmax + 1) >> 1;
              duo_gather<M, CH, D, NB>(q, mine, gmask, sprev, h);  // slots [smax, sprev) become holes again
            }
          }
          ++t;
        }
      });
      HK_POOL_STAMP(0x100 | (smax << 4) | round);
      // fixed point: no live row, or one live row at the origin (nothing changes it any more)
      bool mine_still = true;
      unrolled_while<0, CH>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        if (s >= smax) return false;
        bool zero = true;
#pragma unroll
        for (int k = 0; k < D; ++k) zero &= (q[s * D + k] == 0.0f);
        mine_still &= zero || !(q[s * D] < INFINITY);
        return true;
      });
      still = !have || (np < 2 && mine_still && duo_other_i((int)mine_still) != 0);
      // rows back into the game's image, the rest of the game's state next to it
      wave_sync();
      gmask = pool_scatter<M, CH, D>(q, mine, gmask, smax, h, pad);
      if (have) {
        if (h == 0) {
          m_mask[j] = gmask;
          m_len[j] = length;
        }
        uint32_t* ph = m_phil + (j * 2 + h) * 4;
        ph[0] = dcache.r.x;
        ph[1] = dcache.r.y;
        ph[2] = dcache.r.z;
        ph[3] = dcache.r.w;
      }
    }
    t0 = tend;
    HK_POOL_STAMP(0x200 | round);
    if (t0 >= nsteps && nsteps > 0) break;
    if (nsteps == 0) break;

    // ---- sort the live games by bucket class, widest first ------------------------------------------------------
    int key = 0;
    if (have && h == 0 && !still) {
      const int s = (np + 1) >> 1;
      key = s < 1 ? 1 : (s > kPoolClasses ? kPoolClasses : s);
    }
    int rank_in = 0;
    uint32_t mycnt = 0;
    const unsigned long long lt = ((unsigned long long)1 << lane) - 1;
#pragma unroll
    for (int cc = kPoolClasses; cc >= 1; --cc) {
      const unsigned long long b = __ballot(key == cc);
      rank_in = (key == cc) ? __popcll(b & lt) : rank_in;
      mycnt = (lane == cc) ? (uint32_t)__popcll(b) : mycnt;
    }
    uint32_t* cn = ctr[round & 1];
    uint32_t wbase = 0;
    if (lane >= 1 && lane <= kPoolClasses && mycnt) wbase = atomicAdd(&cn[lane], mycnt);
    __syncthreads();
    uint32_t run = 0, off = 0;
#pragma unroll
    for (int cc = kPoolClasses; cc >= 1; --cc) {
      const uint32_t wb = (uint32_t)__builtin_amdgcn_readlane((int)wbase, cc);
      off = (key == cc) ? run + wb : off;
      run += cn[cc];
    }
    live = (int)run;
    if (key) order[off + (uint32_t)rank_in] = j;
    __syncthreads();
    if (tid < 8) cn[tid] = 0;  // (next used two sorts from now: two barriers in between)
    if (live == 0) break;
    // ---- the new deal: the waves take 32 consecutive positions each; waves w and w + 4 share a SIMD (a workgroup's
    // waves go round the four SIMDs), so the widest games' wave is paired with the narrowest games' one ----------------
    const int p = (wave < kPoolWaves / 2 ? wave : kPoolWaves + kPoolWaves / 2 - 1 - wave) * kDuoGames + gi;
    have = p < live;
    j = have ? order[p] : 0;
    gmask = have ? m_mask[j] : 0u;
    length = have ? m_len[j] : 0;
    if (have) {
      const uint32_t* ph = m_phil + (j * 2 + h) * 4;
      dcache.r.x = ph[0];
      dcache.r.y = ph[1];
      dcache.r.z = ph[2];
      dcache.r.w = ph[3];
    }
    // every live game ran step t0 - 1 in the round before (if there was one)
    dcache.pair = (t0 > 0) ? (step0 + (uint32_t)(t0 - 1)) >> 3 : 0xFFFFFFFFu;  // (a pair of blocks: eight steps)
    HK_POOL_STAMP(0x300 | (live << 12) | round);
  }
  HK_POOL_STAMP(0x400);

  // ---- publish: the images are the final states -------------------------------------------------------------------
  __syncthreads();
  if (ngames > 0) duo_store_slab<M, D>(wimg, (float*)prm.out, prm.out_stride, g0, ngames, lane);
  if (active && h == 0 && prm.game_length_out) prm.game_length_out[gb + home] = m_len[home];
  if (counting) {
    for (int s = tid; s <= nsteps; s += kPoolThreads) {
      uint32_t cum = 0;
      for (int l = 0; l <= s; ++l) cum += hist[l];
      if (cum) count_add(prm.count_ws + (size_t)s * prm.count_stride + blockIdx.x, cum);
    }
  }
#ifdef HK_POOL_PROBE
  HK_POOL_STAMP(0x500);
  if (prm.game_length_out && ngames == kDuoGames) {
    __syncthreads();
    if (lane == 0) {
      int32_t* w = prm.game_length_out + g0;
      for (int i = 0; i < 32; ++i) w[i] = i < probe_n ? probe_v[i] : -1;
    }
  }
#endif
}

// ---- host side -----------------------------------------------------------------------------------------------
// plain rollouts (no per-step records, no Zeillinger host, no sorted output) on a shape with a two-lane kernel
inline bool pool_supported(const Params& prm) {
  if (prm.mode != kModeRollout || prm.m > 32 || prm.steps > kPoolMaxSteps) return false;
  if (prm.host_policy == HK_HOST_ZEILLINGER || prm.game_ids) return false;
  if (prm.obs_out || prm.r_host_class_out || prm.r_axis_out || prm.r_done_out || prm.r_reward_out) return false;
  if ((prm.stages & HK_STAGE_NEWTON) &&
      ((prm.flags & HK_SEM_MASK) == HK_SEM_LIST || (prm.flags & HK_FLAG_COMPACT_SORTED)))
    return false;  // (sorted + compacted output: the other families sort once at the end)
  return true;
}

// where the pool is the default (measured: scripts/probe_pool.py)
inline bool pool_default(const Params& prm, int simds) {
  (void)simds;
  return false;
}

template <int M, int D>
int launch_pool_t(Params prm, hipStream_t stream) {
  const unsigned grid = (unsigned)(((int64_t)prm.batch + kPoolGames - 1) / kPoolGames);
  prm.games_per_block = kPoolGames;
  launch_prepare();
  const int hot = fast_hot_config(prm);
  if (hot == kHotJax)
    hipLaunchKernelGGL((pool_kernel<M, D, kHotJax>), dim3(grid), dim3(kPoolThreads), 0, stream, (const float*)prm.in,
                       prm.in_stride, prm.batch, prm);
  else if (hot == kHotTorch)
    hipLaunchKernelGGL((pool_kernel<M, D, kHotTorch>), dim3(grid), dim3(kPoolThreads), 0, stream, (const float*)prm.in,
                       prm.in_stride, prm.batch, prm);
  else
    hipLaunchKernelGGL((pool_kernel<M, D, kHotNone>), dim3(grid), dim3(kPoolThreads), 0, stream, (const float*)prm.in,
                       prm.in_stride, prm.batch, prm);
  return launch_status();
}

#ifndef HK_SPEC_TU
#define HK_X(M_, D_) extern template int launch_pool_t<M_, D_>(Params, hipStream_t);
HK_FAST_SPECS(HK_X)
#undef HK_X

inline int launch_pool(const Params& prm, hipStream_t stream) {
#define HK_X(M_, D_) if (prm.m == M_ && prm.d == D_) return launch_pool_t<M_, D_>(prm, stream);
  HK_FAST_SPECS(HK_X)
#undef HK_X
  return HK_ERR_UNSUPPORTED;
}
#endif

}  // namespace hk
