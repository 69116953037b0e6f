// Games binned by live rows, on the device: hk_generate_points_binned / hk_bin_by_live_rows.
//
// A wave of the rollout kernels runs the straight-line body of its WIDEST game (hk_duo_kernel.h, hk_quadroll_kernel.h),
// so a batch whose neighbouring games have about the same number of live rows rolls out faster -- (50,4) x 262 144:
// 144 us against 172 us -- and hk_rollout_desc.game_ids keeps every game's policy stream, so the re-ordered batch plays,
// game by game, exactly what the generated order plays.  Round 3 binned with the tensor library (hk_get_num_points +
// a stable argsort + a gather: 85 us per batch of 65 536 games -- seventy times what an episode gains).  Here the order
// is LOCAL to a workgroup's group of games (256, or 64 of the large ones): groups of 64 already give 70 % of what a
// global sort gives (mean squared slots per lane at (50,4): 38.6 unsorted, 24.9 / 22.5 / 20.9 in groups of 64 / 128 / 256,
// 19.2 globally), nothing crosses a workgroup, and it is part of the launch that produces the states:
//   * every quad has its game in registers (the generator's rows after the stages -- hk_quadgen_kernel.h -- or, for a
//     batch that exists already, its quarter of the game's 16-B chunks) and knows its live rows;
//   * a stable counting sort over the group, widest first: per-wave counts per bin (LDS atomics), a game's rank among the
//     equal games of its own wave by sixteen scalar compares, the waves' offsets per bin and one scan over the 64 bins;
//   * the quads write their games into the DESTINATION wave's image, every wave stores its image as one slab, and the
//     game's index in the batch goes to game_ids_out[position];
//   * the k-th sixteen games of ALL the groups lie together in the output, the widest stratum first (group G's k-th
//     wave stores its slab at stratum k, place G): the rollout's workgroups then start with the heavy waves, and every XCD
//     gets every weight.  Measured at (50,4) x 262 144 (scripts/probe_orders.py): 173 us per episode in the generated
//     order, 144 us globally sorted, 147 - 150 us in this layout -- and 193 - 204 us with the groups sorted where they
//     lie (workgroups go round the eight XCDs: every group's heaviest wave lands on the same one).
// The order is a function of the states alone (no atomics decide it): a batch binned twice gives the same permutation.
// No counterpart in the reference: its batches carry no order (jax/util.py:385-392 draws them at random).
#pragma once

#include "hk_quadgen_kernel.h"

namespace hk {

template <int M, int D>
struct QuadBinGeom {
  using G = QuadGeom<M, D>;
  using GG = QuadGenGeom<M, D>;
  // waves of a workgroup = of a group: 16, or 4 for the large games (12.8 KB of image per wave: with 8 waves and 102 KB
  // a CU holds ONE workgroup at a time, and the generator that also bins took 140 us at (50,4) x 262 144 where the
  // plain one took 106; with 4 waves 93 -- the episode on groups of 64 / 128 / all games: 153 / 149 / 144 us)
  static constexpr int kWaves = (GG::kRegion * 4 * 16 <= 100 * 1024) ? 16 : 4;
  static constexpr int kGames = kWaves * kQuadGames;
  static constexpr int kBins = 64;                // bin b = M - live rows (0 = widest); M + 1 = no game
  static constexpr int kPay = (G::Q + kQuad - 1) / kQuad;  // 16-B (W-float) chunks of a game per lane
  static_assert(M + 2 <= kBins, "a bin per number of live rows");
};

// GEN: the states are drawn here (hk_generate_points' arguments in prm); otherwise they are read from in0
template <int M, int D, bool GEN>
__global__ __launch_bounds__((kWave * QuadBinGeom<M, D>::kWaves)) void quadbin_kernel(const float* in0, float* out0,
                                                                                   int32_t* ids_out, int32_t* np_out,
                                                                                   int batch0, const Params prm) {
  using G = QuadGeom<M, D>;
  using GG = QuadGenGeom<M, D>;
  using BG = QuadBinGeom<M, D>;
  using V = typename VecOf<G::W>::type;
  constexpr int R = G::R, W = BG::kWaves;
  extern __shared__ __align__(16) float lds_all[];  // W * GG::kRegion floats (the large games' 100 KB: opted in at launch)
  __shared__ uint32_t wcount[W * BG::kBins];  // games per (wave, bin)
  __shared__ uint32_t woff[W * BG::kBins];    // ... of the earlier waves
  __shared__ uint32_t binbase[BG::kBins];     // games in the bins before; first: the bins' totals
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & (kWave - 1);
  float* region = lds_all + wave * GG::kRegion;
  const int j = lane & 3, gi = lane >> 2;
  const int64_t gbase = (int64_t)blockIdx.x * BG::kGames;
  const int64_t gleft = (int64_t)batch0 - gbase;  // (>= 1: the grid covers the batch)
  const int ngroup = (int)(gleft < BG::kGames ? gleft : BG::kGames);
  const int64_t g0 = gbase + wave * kQuadGames;
  int ngames = ngroup - wave * kQuadGames;
  ngames = ngames < 0 ? 0 : (ngames > kQuadGames ? kQuadGames : ngames);
  const bool active = gi < ngames;
  // (every wave reaches every barrier: no early exit)
  wcount[wave * BG::kBins + lane] = 0u;
  float* mine = region + gi * G::N;
  const float pad = prm.pad_f32;
  float q[GEN ? R * D : 1];
  V pay[GEN ? 1 : BG::kPay];
  int np;
  if constexpr (GEN) {
    const uint64_t gg = prm.game_offset + (uint64_t)(g0 + gi);  // (quads past the batch draw games nobody stores)
    qg_rows<M, D>(q, region, gg, prm.seed, (uint32_t)prm.max_value, j, gi);
    np = qg_stages<M, D, R * D, false>(q, j, prm.flags, prm.stages, region + gi * G::kGameStride, nullptr, prm.max_value, lane);
  } else {
    if (ngames > 0) quad_slab_load<M, D>(in0 + g0 * G::N, region, ngames, lane);
    wait_vmem_all();
    wave_lds_fence();
    int n = 0;  // live rows (x_0 >= 0, hk_get_num_points): lane j looks at rows j, j + 4, ...
#pragma unroll
    for (int s = 0; s < R; ++s)
      if ((kQuad * s + kQuad <= M) || kQuad * s + j < M) n += (mine[(kQuad * s + j) * D] >= 0.0f) ? 1 : 0;
    np = q_sum(n);
#pragma unroll
    for (int i = 0; i < BG::kPay; ++i) {
      const int c = kQuad * i + j;
      pay[i] = *reinterpret_cast<const V*>(mine + ((kQuad * i + kQuad <= G::Q) || c < G::Q ? c : 0) * G::W);
    }
#pragma unroll
    for (int i = 0; i < BG::kPay; ++i) asm volatile("" : "+v"(pay[i]));
  }
  // ---- the game's place in its group: bins from the widest down, games of a bin in their order ---------------------------
  const int bin = active ? M - np : M + 1;
  int within = 0;  // equal games of my wave before mine
#pragma unroll
  for (int o = 0; o < kQuadGames; ++o) {
    const int bo = __builtin_amdgcn_readlane(bin, kQuad * o);
    within += (bo == bin && o < gi) ? 1 : 0;
  }
  wave_lds_fence();
  if (j == 0) atomicAdd(&wcount[wave * BG::kBins + bin], 1u);
  __syncthreads();
  {  // thread (wave, lane = bin): the earlier waves' games of the bin; the last wave also knows the bin's total
    uint32_t before = 0;
#pragma nounroll
    for (int w = 0; w < wave; ++w) before += wcount[w * BG::kBins + lane];
    woff[wave * BG::kBins + lane] = before;
    if (wave == W - 1) binbase[lane] = before + wcount[wave * BG::kBins + lane];
  }
  __syncthreads();
  if (wave == 0) {  // exclusive scan over the bins
    const uint32_t tot = binbase[lane];
    uint32_t x = tot;
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
      const uint32_t y = (uint32_t)__shfl_up((int)x, o);
      x += (lane >= o) ? y : 0u;
    }
    binbase[lane] = x - tot;
  }
  __syncthreads();
  const int pos = (int)(binbase[bin] + woff[wave * BG::kBins + bin]) + within;
  // where the sixteen games of the group's k-th wave go: stratum k, place = the group's index, for the full groups; a
  // partial last group stays where it lies (behind all the strata)
  const int64_t nfull = (int64_t)batch0 / BG::kGames;
  const bool full = ngroup == BG::kGames;
  auto stratum_base = [&](int k) -> int64_t {
    return full ? ((int64_t)k * nfull + blockIdx.x) * kQuadGames : gbase + (int64_t)k * kQuadGames;
  };
  float* dst = lds_all + (pos >> 4) * GG::kRegion + (pos & (kQuadGames - 1)) * G::N;
  // (every wave is past its own use of its region: the generator's staging / scratch, or the chunks it took out)
  if constexpr (GEN) {
#pragma unroll
    for (int s = 0; s < R; ++s) {
      if ((kQuad * s + kQuad <= M) || kQuad * s + j < M) {
        const bool removed = !(q[s * D] < INFINITY);
        float* row = dst + (kQuad * s + j) * D;
        if constexpr (D == 4) {
          *reinterpret_cast<vf4*>(row) = vf4{removed ? pad : q[s * D], removed ? pad : q[s * D + 1],
                                             removed ? pad : q[s * D + 2], removed ? pad : q[s * D + 3]};
        } else {
#pragma unroll
          for (int k = 0; k < D; ++k) row[k] = removed ? pad : q[s * D + k];
        }
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < BG::kPay; ++i) {
      const int c = kQuad * i + j;
      if ((kQuad * i + kQuad <= G::Q) || c < G::Q) *reinterpret_cast<V*>(dst + c * G::W) = pay[i];
    }
  }
  if (active && j == 0) {
    const int64_t at = stratum_base(pos >> 4) + (pos & (kQuadGames - 1));
    ids_out[at] = (int32_t)(g0 + gi);
    if (np_out) np_out[at] = np;
  }
  __syncthreads();
  if (ngames > 0) quad_slab_store<M, D>(region, out0 + stratum_base(wave) * G::N, ngames, lane);
}

// ---- host side -------------------------------------------------------------------------------------------------------
template <int M, int D>
int launch_quadbin_t(Params prm, const float* in, int32_t* ids_out, int32_t* np_out, hipStream_t stream) {
  using BG = QuadBinGeom<M, D>;
  const unsigned grid = (unsigned)(((int64_t)prm.batch + BG::kGames - 1) / BG::kGames);
  constexpr size_t lds = (size_t)BG::kWaves * QuadGenGeom<M, D>::kRegion * sizeof(float);
  prm.pad_f32 = (float)prm.pad;
  if (lds > 48 * 1024) {  // (static + dynamic LDS beyond 64 KiB is a per-function opt-in: idempotent, cheap)
    const void* fn = in ? reinterpret_cast<const void*>(&quadbin_kernel<M, D, false>)
                        : reinterpret_cast<const void*>(&quadbin_kernel<M, D, true>);
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      (void)hipGetLastError();
      return HK_ERR_LAUNCH;
    }
  }
  launch_prepare();
  if (in)
    hipLaunchKernelGGL((quadbin_kernel<M, D, false>), dim3(grid), dim3(kWave * BG::kWaves), lds, stream, in, (float*)prm.out,
                       ids_out, np_out, prm.batch, prm);
  else
    hipLaunchKernelGGL((quadbin_kernel<M, D, true>), dim3(grid), dim3(kWave * BG::kWaves), lds, stream, (const float*)nullptr,
                       (float*)prm.out, ids_out, np_out, prm.batch, prm);
  return launch_status();
}

// games per group of a shape (0: the shape has no binning kernel)
inline int quadbin_group_games(int m, int d, int dtype) {
  if (dtype != HK_F32) return 0;
#define HK_X(M_, D_) if (m == M_ && d == D_) return QuadBinGeom<M_, D_>::kGames;
  HK_QUAD_SPECS(HK_X)
#undef HK_X
  return 0;
}

#ifndef HK_SPEC_TU
#define HK_X(M_, D_) extern template int launch_quadbin_t<M_, D_>(Params, const float*, int32_t*, int32_t*, hipStream_t);
HK_QUAD_SPECS(HK_X)
#undef HK_X

inline int launch_quadbin(const Params& prm, const float* in, int32_t* ids_out, int32_t* np_out, hipStream_t stream) {
#define HK_X(M_, D_) if (prm.m == M_ && prm.d == D_) return launch_quadbin_t<M_, D_>(prm, in, ids_out, np_out, stream);
  HK_QUAD_SPECS(HK_X)
#undef HK_X
  return HK_ERR_UNSUPPORTED;
}
#endif

}  // namespace hk
