// Generic kernel: one lane per game, the game's points resident in LDS for the whole
// launch, runtime (max_points, dim), T = float | double.  Serves every shape that has no
// register-resident specialisation, the list/torch semantics, the operators that are not on
// the bench path (zeillinger, features) and doubles as an independent GPU implementation the
// specialised kernels are cross-checked against.
//
// Data movement: a workgroup (one wave) owns `games_per_block` consecutive games.  Their
// records are contiguous in HBM ([batch, max_points, dim] row-major), so the wave copies the
// slab with consecutive lanes on consecutive addresses (fully coalesced 256 B wave-requests)
// into an LDS image with an ODD per-game stride; afterwards lane g walks its own game with
// ds_read/ds_write_b32 at stride `lds_stride` dwords -- conflict-free because the stride is odd
// (bank = (lane*stride + e) mod 32 is a bijection over 32 lanes).  The result leaves the same
// way.  HBM traffic is exactly one read and one write of the state.
#pragma once

#include "hk_game_generic.h"

namespace hk {

// cooperative, coalesced HBM -> LDS copy of `ngames` records of `n` elements
// (rolled loop, n is a run-time value: batched by hand so that kSlabBatch independent requests per lane
// are in flight before the first dependent access -- one per iteration would expose a full HBM latency)
constexpr int kSlabBatch = 8;

template <typename T, bool TO_LDS>
__device__ inline void copy_slab(T* lds, T* glob, int64_t gstride, int n, int S, int64_t g0, int ngames,
                                 int lane) {
  int g = lane / n, e = lane % n;
  const int dg = kWave / n, de = kWave % n;
  const int total = ngames * n;
  for (int c0 = lane; c0 < total; c0 += kWave * kSlabBatch) {
    T v[kSlabBatch];
    int lo[kSlabBatch];
    int64_t go[kSlabBatch];
#pragma unroll
    for (int u = 0; u < kSlabBatch; ++u) {
      lo[u] = g * S + e;
      go[u] = (g0 + g) * gstride + e;
      g += dg;
      e += de;
      if (e >= n) {
        e -= n;
        ++g;
      }
    }
    // requests past the end repeat the batch's first (valid) one, so every load is unconditional and
    // the whole batch is in flight before the opaque use below; a load left inside its `if` is sunk
    // next to its store by the compiler and waited for there, one round trip per element
#pragma unroll
    for (int u = 1; u < kSlabBatch; ++u)
      if (c0 + u * kWave >= total) {
        lo[u] = lo[0];
        go[u] = go[0];
      }
#pragma unroll
    for (int u = 0; u < kSlabBatch; ++u) v[u] = TO_LDS ? glob[go[u]] : lds[lo[u]];
#pragma unroll
    for (int u = 0; u < kSlabBatch; ++u) asm volatile("" : "+v"(v[u]));
#pragma unroll
    for (int u = 0; u < kSlabBatch; ++u)
      if (c0 + u * kWave < total) {
        if (TO_LDS) lds[lo[u]] = v[u];
        else glob[go[u]] = v[u];
      }
  }
}

template <typename T>
__device__ inline void load_slab(T* lds, const T* in, int64_t in_stride, int n, int S, int64_t g0,
                                 int ngames, int lane) {
  copy_slab<T, true>(lds, const_cast<T*>(in), in_stride, n, S, g0, ngames, lane);
}

template <typename T>
__device__ inline void store_slab(const T* lds, T* out, int64_t out_stride, int n, int S,
                                  int64_t g0, int ngames, int lane) {
  copy_slab<T, false>(const_cast<T*>(lds), out, out_stride, n, S, g0, ngames, lane);
}

// host subset of game `g` -> c[0..d) (as T) ; returns the bitmask of entries == 1
template <typename T>
__device__ inline void load_coords(const Params& prm, int64_t g, T* c) {
  const int d = prm.d, kind = prm.coords_kind;
  if (kind == HK_COORDS_CLASS_I32 || kind == HK_COORDS_CLASS_I64) {
    long long cls = (kind == HK_COORDS_CLASS_I32) ? (long long)((const int32_t*)prm.coords)[g]
                                                  : ((const long long*)prm.coords)[g];
    const long long ncls = (1ll << d) - d - 1;
    cls = cls < 0 ? 0 : (cls >= ncls ? ncls - 1 : cls);  // ids are validated by the caller
    const uint32_t v = decode_class((int)cls, d);
    for (int k = 0; k < d; ++k) c[k] = (T)((v >> k) & 1u);
  } else if (kind == HK_COORDS_IN_RECORD) {
    const T* rec = (const T*)prm.in + g * prm.in_stride + (int64_t)prm.m * d;
    for (int k = 0; k < d; ++k) c[k] = rec[k];
  } else if (kind >= HK_F32 && kind <= HK_U8) {
    for (int k = 0; k < d; ++k)
      c[k] = (T)load_scalar(prm.coords, kind, (size_t)(g * prm.coords_stride + k));
  } else {
    for (int k = 0; k < d; ++k) c[k] = (T)0;
  }
}

// fixed policies of jax/players.py for one game (DESIGN.md "Randomness")
template <typename T>
__device__ inline void choose_actions(const T* p, const Params& prm, uint64_t gg, uint32_t step,
                                      int& cls, int& axis, uint32_t& mask) {
  const int d = prm.d;
  const uint32_t ncls = (1u << d) - (uint32_t)d - 1u;
  PolicyCache cache;
  uint32_t ra, rb;
  policy_words(gg, step, prm.seed, cache, prm.d, ra, rb);
  if (prm.host_policy == HK_HOST_RANDOM) cls = (int)mulhi32(ra, ncls);
  else if (prm.host_policy == HK_HOST_ALL_COORD) cls = (int)ncls - 1;
  else cls = zeillinger_game(p, prm.m, d);
  mask = decode_class(cls, d);
  if (prm.agent_policy == HK_AGENT_RANDOM) {
    axis = (int)mulhi32(rb, (uint32_t)d);
  } else if (prm.agent_policy == HK_AGENT_RANDOM_LEGAL) {
    const int pick = (int)mulhi32(rb, (uint32_t)__popc(mask));
    int seen = 0;
    axis = 0;
    for (int k = 0; k < d; ++k)
      if ((mask >> k) & 1u) {
        if (seen == pick) axis = k;
        ++seen;
      }
  } else if (prm.agent_policy == HK_AGENT_CHOOSE_FIRST) {
    axis = __ffs(mask) - 1;
  } else {
    axis = 31 - __clz(mask);
  }
}

template <typename T>
__global__ __launch_bounds__(kWave) void generic_kernel(const Params prm) {
  extern __shared__ __align__(16) unsigned char hk_smem[];
  T* lds = reinterpret_cast<T*>(hk_smem);
  const int lane = threadIdx.x;
  const int m = prm.m, d = prm.d, n = m * d, S = prm.lds_stride;
  const int gpb = prm.games_per_block;
  const int64_t g0 = (int64_t)blockIdx.x * gpb;
  const int64_t left = (int64_t)prm.batch - g0;
  const int ngames = (int)(left < gpb ? left : gpb);
  const bool active = lane < ngames;
  const int64_t g = g0 + lane;
  T* p = lds + lane * S;
  T* c = p + n;  // d scratch elements behind the points: subset mask, later a row buffer
  const T pad = (T)prm.pad;

  if (prm.mode == kModeGenerate) {
    if (active) {
      const uint64_t gg = prm.game_offset + (uint64_t)g;
      // (hk_common.h: eight elements per Philox block for small max_value, four otherwise)
      const bool sh = gen_short((uint32_t)prm.max_value);
      const int per = sh ? 8 : 4;
      for (int e = 0; e < n; e += per) {
        const U4 r = philox4x32((uint32_t)gg, (uint32_t)(gg >> 32), (uint32_t)(e / per), kStreamGenerate, prm.seed);
        uint32_t v[8];
        gen_block_values(r, (uint32_t)prm.max_value, sh, v);
        for (int q = 0; q < per && e + q < n; ++q) p[e + q] = (T)v[q];
      }
      stages_game(p, m, d, c, -1, pad, prm.stages & ~HK_STAGE_SHIFT, prm.flags);
    }
    __syncthreads();
    store_slab(lds, (T*)prm.out, prm.out_stride, n, S, g0, ngames, lane);
    return;
  }

  load_slab(lds, (const T*)prm.in, prm.in_stride, n, S, g0, ngames, lane);
  __syncthreads();

  if (prm.mode == kModeZeillinger) {
    if (active)
      prm.class_out[g] = ((prm.flags & HK_SEM_MASK) == HK_SEM_LIST) ? zeillinger_list_game(p, m, d)
                                                                     : zeillinger_game(p, m, d);
    return;
  }

  if (prm.mode == kModeStep) {
    if (active) {
      const int before = num_points(p, m, d);
      int axis = -1;
      if (prm.stages & HK_STAGE_SHIFT) {
        load_coords(prm, g, c);
        axis = axis_index(load_scalar(prm.axis, prm.axis_dtype, (size_t)g), d);
      }
      stages_game(p, m, d, c, axis, pad, prm.stages, prm.flags);
      const int after = num_points(p, m, d);
      const bool prev_done = before < 2, done = after < 2;
      if (prm.done_out) prm.done_out[g] = done;
      if (prm.prev_done_out) prm.prev_done_out[g] = prev_done;
      if (prm.reward_out) prm.reward_out[g] = prm.reward_sign * (float)(done && !prev_done);
      if (prm.num_points_out) prm.num_points_out[g] = after;
    }
    __syncthreads();
    store_slab(lds, (T*)prm.out, prm.out_stride, n, S, g0, ngames, lane);
    return;
  }

  // ---- kModeRollout: T fused steps, state never leaves LDS ---------------------------------
  const uint64_t gg = prm.game_offset + ((prm.game_ids && active) ? (uint64_t)(uint32_t)prm.game_ids[g] : (uint64_t)g);  // (ids: unsigned 32-bit)
  int np = active ? num_points(p, m, d) : 2;
  int length = (np < 2) ? 0 : -1;
  if (prm.count_ws) {
    const unsigned long long b0 = __ballot(active && np < 2);
    if (lane == 0) count_add(prm.count_ws + blockIdx.x, (uint32_t)__popcll(b0));
  }
  for (int t = 0; t < prm.steps; ++t) {
    if (prm.obs_out) {  // state before the step, coalesced
      store_slab(lds, (T*)prm.obs_out + (int64_t)t * prm.batch * n, (int64_t)n, n, S, g0, ngames, lane);
      __syncthreads();
    }
    bool done = false;
    if (active) {
      const bool prev_done = np < 2;
      int cls, axis;
      uint32_t mask;
      choose_actions(p, prm, gg, prm.step_offset + (uint32_t)t, cls, axis, mask);
      for (int k = 0; k < d; ++k) c[k] = (T)((mask >> k) & 1u);
      stages_game(p, m, d, c, axis, pad, prm.stages, prm.flags);
      np = num_points(p, m, d);
      done = np < 2;
      if (done && length < 0) length = t + 1;
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
    __syncthreads();
  }
  if (active && prm.game_length_out) prm.game_length_out[g] = length;
  store_slab(lds, (T*)prm.out, prm.out_stride, n, S, g0, ngames, lane);
}

}  // namespace hk
