// Four lanes per game, state generation: hk_generate_points on the four-lane data flow of hk_quad_kernel.h.
//
// generate_pts (hironaka/jax/util.py:385-392): randint[0, max_value) -> float32 -> Newton polytope -> [reposition] ->
// [rescale].  The generator stream (DESIGN.md "Randomness"; hk_common.h: gen_block_values) is Philox4x32-10 keyed by the
// seed with counter (game_lo, game_hi, block, kStreamGenerate): eight elements per block (16-bit draws) while max_value
// is small, four otherwise.
//
// Rounds 1 - 3 ran this on hk::fast_kernel: ONE lane per game computes the game's M * D / 4 Philox blocks, scans its
// image and walks the whole triangle of row pairs -- 21.5 us per 65 536 (20,3)-games, one wave per SIMD, longer than the
// 19.4 us rollout it feeds.  Here a QUAD of lanes owns a game (16 games per wave, four waves per SIMD):
//   * the Philox blocks of a game are dealt over its four lanes: block b goes to lane b & 3, its eight (or four) values as
//     16-B LDS writes at their place in the game's staging image -- the concatenation of the blocks IS the game --, and
//     lane j reads rows j, j + 4, ... back;
//   * every row of a fresh game is live (the draws are non-negative), so a row's rank is its index: no scan, no
//     compaction, rank of (slot s, lane j) = 4 s + j as qd_newton / qd_newton_two_level expect;
//   * the stages run once, straight-line, for all R slots: the DPP pair test up to 8 slots per lane, the two-level test
//     through the parked rows beyond ((50,4): 13 slots per lane);
//   * every lane writes its own rows (or padding) to their places in the wave's image: no fill pass; one slab store.
// The exactness guard has nothing to check here: the draws are finite and >= +0.  What the guard's other half covers --
// a duplicate-fill value (-1.0 under JAX semantics, _jax_ops.py:65) different from the padding value -- and the sorted
// output of the list semantics stay with the other kernels (quadgen_supported declines).
#pragma once

#include "hk_quad_kernel.h"

namespace hk {

template <int M, int D>
struct QuadGenGeom {
  using G = QuadGeom<M, D>;
  static constexpr int R = G::R;
  // Philox blocks per game: eight elements per block for small max_value, four otherwise (hk_common.h)
  static constexpr int kBlocksShort = (M * D + 7) / 8, kBlocksLong = (M * D + 3) / 4;
  // floats per game of the staging image: whole blocks, 16-B aligned
  static constexpr int kStage = (kBlocksShort * 8 > kBlocksLong * 4) ? kBlocksShort * 8 : kBlocksLong * 4;
  static constexpr int kRegion =  // floats per wave: staging and image alias
      (kQuadGames * kStage > G::kImage) ? kQuadGames * kStage : G::kImage;
  static_assert(G::kGameStride == G::N, "the parked rows of the pair loops lie in the game's part of the image");
  static constexpr bool kTwoLevel = R > kQuadDppSlots && M <= 64;  // qd_newton_two_level's scratch: 128 B per game
};

// The rows of game `gg` in the four-lane layout: slot s of lane j = row 4 s + j; slots past the game's M rows are holes.
// Block b is computed by lane b & 3 of the quad and written -- 32 or 16 bytes -- at its place in the game's staging image
// (the concatenation of the blocks IS the game); lane j then reads rows j, j + 4, ....  `stage`: the wave's LDS region,
// free again on return.
template <int M, int D, int RD>
__device__ __forceinline__ void qg_rows(float (&q)[RD], float* stage, uint64_t gg, uint64_t seed, uint32_t max_value,
                                        int j, int gi) {
  using GG = QuadGenGeom<M, D>;
  constexpr int R = GG::R;
  static_assert(RD == R * D, "slots per lane x dim");
  // (opaque: inside a rollout's episode loop the first Philox round of every block -- it depends on the lane alone -- and
  // the staging addresses were hoisted out of the loop and kept in registers through every episode: spills)
  asm volatile("" : "+v"(j), "+v"(gi));
  float* mine = stage + gi * GG::kStage;
  if (gen_short(max_value)) {  // (wave-uniform)
#pragma unroll
    for (int i = 0; i < (GG::kBlocksShort + kQuad - 1) / kQuad; ++i) {
      const int b = kQuad * i + j;
      const U4 r = philox4x32((uint32_t)gg, (uint32_t)(gg >> 32), (uint32_t)b, kStreamGenerate, seed);
      uint32_t v[8];
      gen_block_values(r, max_value, true, v);
      if ((kQuad * i + kQuad <= GG::kBlocksShort) || b < GG::kBlocksShort) {
        *reinterpret_cast<vf4*>(mine + 8 * b) = vf4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
        *reinterpret_cast<vf4*>(mine + 8 * b + 4) = vf4{(float)v[4], (float)v[5], (float)v[6], (float)v[7]};
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < (GG::kBlocksLong + kQuad - 1) / kQuad; ++i) {
      const int b = kQuad * i + j;
      const U4 r = philox4x32((uint32_t)gg, (uint32_t)(gg >> 32), (uint32_t)b, kStreamGenerate, seed);
      uint32_t v[8];
      gen_block_values(r, max_value, false, v);
      if ((kQuad * i + kQuad <= GG::kBlocksLong) || b < GG::kBlocksLong)
        *reinterpret_cast<vf4*>(mine + 4 * b) = vf4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    }
  }
  wave_lds_fence();
#pragma unroll
  for (int s = 0; s < R; ++s) {
    const bool has = (kQuad * s + kQuad <= M) || kQuad * s + j < M;
    const float* src = mine + (has ? kQuad * s + j : 0) * D;
    if constexpr (D == 4) {
      const vf4 v = *reinterpret_cast<const vf4*>(src);
      q[s * D] = has ? v.x : INFINITY;
      q[s * D + 1] = has ? v.y : INFINITY;
      q[s * D + 2] = has ? v.z : INFINITY;
      q[s * D + 3] = has ? v.w : INFINITY;
    } else {
#pragma unroll
      for (int k = 0; k < D; ++k) {
        const float v = src[k];
        q[s * D + k] = has ? v : INFINITY;
      }
    }
  }
  wave_lds_fence();  // (the region becomes the output image)
}

// (the domination test on packed rows -- qg_newton_packed -- lives in hk_quad_kernel.h: hk_step's dense states use it too)

// the generator's stages on all R slots (every row live: ranks = row indices); returns the GAME's number of live rows
// TWO: the float fallback of many slots per lane may use qd_newton_two_level (`tsc`: its 128 B of scratch per game)
template <int M, int D, int RD, bool TWO = true>
__device__ __forceinline__ int qg_stages(float (&q)[RD], int j, unsigned flags, unsigned stages, float* cmine,
                                         uint8_t* tsc, int max_value, int lane) {
  using G = QuadGeom<M, D>;
  constexpr int R = G::R;
  static_assert(RD == R * D, "slots per lane x dim");
  // many slots per lane, small draws: the Newton stage on packed rows, the minima subtracted afterwards
  const bool packed = R > kQuadDppSlots && D <= 4 && (stages & HK_STAGE_NEWTON) && max_value <= kPackMaxValue;
  if constexpr (R > kQuadDppSlots && D <= 4) {
    if (packed) qg_newton_packed<M, D, R>(q, reinterpret_cast<uint32_t*>(cmine), j, lane);
  }
  if (stages & HK_STAGE_REPOSITION) qd_reposition<R, D, R>(q, flags);
  if ((stages & HK_STAGE_NEWTON) && !packed) {
    if constexpr (R > kQuadDppSlots) {
      if constexpr (TWO && QuadGenGeom<M, D>::kTwoLevel) qd_newton_two_level<M, G::CW, R, D, R>(q, cmine, tsc, j, M);
      else qd_newton_lds<M, G::CW, R, D, R, true>(q, cmine, j, M);
    } else {
      qd_newton<R, D, R>(q, j);
    }
  }
  if (stages & HK_STAGE_RESCALE) qd_rescale<R, D, R>(q, flags);
  int n = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) n += (q[r * D] < INFINITY) ? 1 : 0;
  return q_sum(n);
}

template <int M, int D, int WPB>
__global__ __launch_bounds__(kWave * WPB, (QuadGeom<M, D>::kWavesPerSimd)) void quadgen_kernel(float* out0, int batch0,
                                                                                              const Params prm) {
  using G = QuadGeom<M, D>;
  using GG = QuadGenGeom<M, D>;
  constexpr int R = G::R;
  __shared__ __align__(16) float lds_all[WPB * GG::kRegion];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & (kWave - 1);
  float* region = lds_all + wave * GG::kRegion;
  const int j = lane & 3, gi = lane >> 2;
  const int64_t g0 = ((int64_t)blockIdx.x * WPB + wave) * kQuadGames;
  const int64_t left = (int64_t)batch0 - g0;
  if (left <= 0) return;  // (waves of a workgroup never meet at a barrier)
  const int ngames = (int)(left < kQuadGames ? left : kQuadGames);
  const uint64_t gg = prm.game_offset + (uint64_t)(g0 + gi);  // (quads past the batch draw games nobody stores)
  float q[R * D];
  qg_rows<M, D>(q, region, gg, prm.seed, (uint32_t)prm.max_value, j, gi);
  float* mine = region + gi * G::N;
  // (the parked rows of the big games' pair loops: the game's part of the region -- G::kGameStride == N there.  The
  // float fallback of many slots per lane -- draws of 127 and more -- is the one-level loop: with qd_newton_two_level
  // compiled in, the kernel needed 188 registers, two waves per SIMD, where the packed path gets by with 134 and three:
  // 106 -> 93 us at (50,4) x 262 144)
  (void)qg_stages<M, D, R * D, false>(q, j, prm.flags, prm.stages, region + gi * G::kGameStride, nullptr, prm.max_value,
                                      lane);
  const float pad = prm.pad_f32;
#pragma unroll
  for (int s = 0; s < R; ++s) {
    if ((kQuad * s + kQuad <= M) || kQuad * s + j < M) {
      const bool removed = !(q[s * D] < INFINITY);
      float* dst = mine + (kQuad * s + j) * D;
      if constexpr (D == 4) {
        *reinterpret_cast<vf4*>(dst) = vf4{removed ? pad : q[s * D], removed ? pad : q[s * D + 1],
                                           removed ? pad : q[s * D + 2], removed ? pad : q[s * D + 3]};
      } else {
#pragma unroll
        for (int k = 0; k < D; ++k) dst[k] = removed ? pad : q[s * D + k];
      }
    }
  }
  wave_lds_fence();
  quad_slab_store<M, D>(region, out0 + g0 * G::N, ngames, lane);
}

// ---- host side -------------------------------------------------------------------------------------------------------
template <int M, int D>
int launch_quadgen_t(Params prm, hipStream_t stream) {
  constexpr int WPB = quad_waves_per_block<M, D>();
  const int64_t waves = ((int64_t)prm.batch + kQuadGames - 1) / kQuadGames;
  const unsigned grid = (unsigned)((waves + WPB - 1) / WPB);
  prm.pad_f32 = (float)prm.pad;
  launch_prepare();
  hipLaunchKernelGGL((quadgen_kernel<M, D, WPB>), dim3(grid), dim3(kWave * WPB), 0, stream, (float*)prm.out, prm.batch,
                     prm);
  return launch_status();
}

// hk_generate_points requests this kernel serves: float32, contiguous W-aligned records, in-place order, a duplicate
// fill equal to the padding value (the other kernels' exactness guard sends anything else down the generic routines)
inline bool quadgen_supported(const Params& prm, int dtype) {
  if (dtype != HK_F32 || prm.mode != kModeGenerate) return false;
  if (prm.flags & (HK_FLAG_FORCE_GENERIC | HK_FLAG_FORCE_TEAM | HK_FLAG_FORCE_ONE_LANE | HK_FLAG_FORCE_TWO_LANES))
    return false;
  if ((prm.flags & HK_SEM_MASK) == HK_SEM_LIST || (prm.flags & HK_FLAG_COMPACT_SORTED)) return false;
  if ((prm.flags & HK_SEM_MASK) == HK_SEM_JAX && (prm.stages & HK_STAGE_NEWTON) && (float)prm.pad != -1.0f) return false;
  if (!((float)prm.pad < 0.0f)) return false;  // (a removed row must read as padding)
  if (prm.out_stride != (int64_t)prm.m * prm.d) return false;
#define HK_X(M_, D_) \
  if (prm.m == M_ && prm.d == D_) return reinterpret_cast<uintptr_t>(prm.out) % (QuadGeom<M_, D_>::W * 4) == 0;
  HK_QUAD_SPECS(HK_X)
#undef HK_X
  return false;
}

#ifndef HK_SPEC_TU
#define HK_X(M_, D_) extern template int launch_quadgen_t<M_, D_>(Params, hipStream_t);
HK_QUAD_SPECS(HK_X)
#undef HK_X

inline int launch_quadgen(const Params& prm, hipStream_t stream) {
#define HK_X(M_, D_) if (prm.m == M_ && prm.d == D_) return launch_quadgen_t<M_, D_>(prm, stream);
  HK_QUAD_SPECS(HK_X)
#undef HK_X
  return HK_ERR_UNSUPPORTED;
}
#endif

}  // namespace hk
