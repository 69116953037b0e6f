// mid_kernel<D, MODE>: float32, JAX / torch in-place semantics, compile-time dim D in 2..6, RUN-TIME
// max_points <= 64.  The kernel for big games (BASELINE config 3: 50 points, dim 4) and for every f32
// shape without a register-resident specialisation.
//
// Why it exists: holding a game's rows in registers needs fully unrolled loops; at 32+ rows the
// unrolled O(n^2) pair code (~100 KB) no longer fits the 64 KB instruction cache and the VGPR file
// caps the occupancy.  Here the rows stay in LDS -- "the domination test staged in LDS" -- and the
// loops are ordinary rolled loops:
//   * I/O as in the other kernels: the wave copies its contiguous slab with coalesced (16-B where the
//     record allows) requests into an LDS image, per-game stride conflict-free for the row width.
//   * one scan builds the live-row bitmask (+ the exactness guard of hk_fast_kernel.h), then the live
//     rows are COMPACTED IN PLACE to the front of the lane's image (write index <= read index), and
//     every stage loops over the lane's own n rows (lanes with fewer rows idle; no hole handling).
//   * the running minima of the domination test (one float per row, see hk_fast_rows.h) live in the
//     FREE tail of the lane's image -- n rows + n floats must fit m*D floats, which a Newton-reduced
//     game always satisfies; denser games take the exact generic path -- so LDS use stays at one image.
//   * after the test the survivors are compacted again (and their slot bits kept), so a fused
//     rollout never carries dead rows; the image is rebuilt (rows back to their slots, padding
//     elsewhere) only when the state leaves the chip.
#pragma once

#include "hk_fast_kernel.h"

namespace hk {

using Mask64 = unsigned long long;

template <int D>
__device__ __forceinline__ void row_load(const float* p, float (&v)[D]) {
  if constexpr (D == 4) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  } else if constexpr (D == 2) {
    const float2 t = *reinterpret_cast<const float2*>(p);
    v[0] = t.x; v[1] = t.y;
  } else {
#pragma unroll
    for (int k = 0; k < D; ++k) v[k] = p[k];
  }
}

template <int D>
__device__ __forceinline__ void row_store(float* p, const float (&v)[D]) {
  if constexpr (D == 4) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  } else if constexpr (D == 2) {
    *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]);
  } else {
#pragma unroll
    for (int k = 0; k < D; ++k) p[k] = v[k];
  }
}

// row width in floats the LDS stride must respect (ds_read_b128 / b64 / b32)
template <int D>
constexpr int mid_row_width() { return D == 4 ? 4 : (D == 2 ? 2 : 1); }

// coalesced HBM <-> LDS slab copy, run-time record length n; 16-B requests when the records allow
// The loop is rolled (n is a run-time value), so it is batched by hand: kCopyBatch independent
// requests per lane are issued before the first dependent LDS access, otherwise every iteration would
// expose a full HBM latency (measured: 273 us instead of ~80 us for the (50,4) slab copy).
constexpr int kCopyBatchDefault = 16;
constexpr int kCopyBatchInLoop = 4;  // stores issued inside the rollout loop: keep the loop's register pressure low

template <bool TO_LDS, int kCopyBatch = kCopyBatchDefault>
__device__ inline void mid_copy_slab(float* lds, float* glob, int64_t gstride, int n, int S, int64_t g0,
                                     int ngames, int lane, bool vec4) {
  if (vec4) {
    const int Q = n >> 2;
    const bool lds4 = (S & 3) == 0;  // 16-B aligned LDS rows: ds_read/write_b128
    int g = lane / Q, c = lane % Q;
    const int dg = kWave / Q, dc = kWave % Q;
    const int total = ngames * Q;
    for (int q0 = lane; q0 < total; q0 += kWave * kCopyBatch) {
      vf4 v[kCopyBatch];
      int lo[kCopyBatch];
      int64_t go[kCopyBatch];
#pragma unroll
      for (int u = 0; u < kCopyBatch; ++u) {
        lo[u] = g * S + c * 4;
        go[u] = (g0 + g) * gstride + c * 4;
        g += dg;
        c += dc;
        if (c >= Q) { c -= Q; ++g; }
      }
      // unconditional requests (see copy_slab): past the end, repeat the batch's first one
#pragma unroll
      for (int u = 1; u < kCopyBatch; ++u)
        if (q0 + u * kWave >= total) {
          lo[u] = lo[0];
          go[u] = go[0];
        }
#pragma unroll
      for (int u = 0; u < kCopyBatch; ++u) {
        if (TO_LDS) {
          v[u] = *reinterpret_cast<const vf4*>(glob + go[u]);
        } else if (lds4) {
          v[u] = *reinterpret_cast<const vf4*>(lds + lo[u]);
        } else {
          const float* l = lds + lo[u];
          v[u] = vf4{l[0], l[1], l[2], l[3]};
        }
      }
#pragma unroll
      for (int u = 0; u < kCopyBatch; ++u) asm volatile("" : "+v"(v[u]));
#pragma unroll
      for (int u = 0; u < kCopyBatch; ++u) {
        if (q0 + u * kWave < total) {
          if (!TO_LDS) {
            *reinterpret_cast<vf4*>(glob + go[u]) = v[u];
          } else if (lds4) {
            *reinterpret_cast<vf4*>(lds + lo[u]) = v[u];
          } else {
            float* l = lds + lo[u];
            l[0] = v[u].x; l[1] = v[u].y; l[2] = v[u].z; l[3] = v[u].w;
          }
        }
      }
    }
  } else {
    copy_slab<float, TO_LDS>(lds, glob, gstride, n, S, g0, ngames, lane);
  }
}

// live rows of `mask` (ascending) to the front of the image; returns their number
template <int D>
__device__ inline int mid_compact(float* mine, Mask64 mask) {
  int r = 0;
  while (mask) {
    const int s = __ffsll(mask) - 1;
    mask &= mask - 1;
    if (s != r) {
      float v[D];
      row_load<D>(mine + s * D, v);
      row_store<D>(mine + r * D, v);
    }
    ++r;
  }
  return r;
}

// inverse: compact rows [0, n) back to the slots of `mask` (highest first), padding everywhere else
template <int D>
__device__ inline void mid_expand(float* mine, Mask64 mask, int n, int m, float pad) {
  Mask64 mk = mask;
  for (int r = n - 1; r >= 0; --r) {
    const int s = 63 - __clzll(mk);
    mk &= ~((Mask64)1 << s);
    if (s != r) {
      float v[D];
      row_load<D>(mine + r * D, v);
      row_store<D>(mine + s * D, v);
    }
  }
  float pv[D];
#pragma unroll
  for (int k = 0; k < D; ++k) pv[k] = pad;
  for (int i = 0; i < m; ++i)
    if (!((mask >> i) & 1ull)) row_store<D>(mine + i * D, pv);
}

// _jax_ops.py:76-90 / _torch_ops.py:46-110 on n compact rows
template <int D>
__device__ inline void mid_shift(float* mine, int n, const float (&c)[D], int axis, unsigned flags) {
  bool apply = axis >= 0;
  if (flags & HK_FLAG_AXIS_NOOP_IF_INVALID) {
#pragma unroll
    for (int k = 0; k < D; ++k) {
      const float onehot = (k == axis) ? 1.0f : 0.0f;
      if (!(onehot - c[k] <= 0.0f)) apply = false;
    }
  }
  if ((flags & HK_FLAG_IGNORE_ENDED) && n < 2) apply = false;
  if (!apply) return;
  for (int r = 0; r < n; ++r) {
    float v[D];
    row_load<D>(mine + r * D, v);
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < D; ++k) s = s + v[k] * c[k];  // order 0..D-1, no contraction
    mine[r * D + axis] = s;
  }
}

// _jax_ops.py:114-123 / _torch_ops.py:113-133
template <int D>
__device__ inline void mid_reposition(float* mine, int n, unsigned flags) {
  const bool jax_sem = (flags & HK_SEM_MASK) == HK_SEM_JAX;
  float mn[D];
#pragma unroll
  for (int k = 0; k < D; ++k) mn[k] = INFINITY;
  for (int r = 0; r < n; ++r) {
    float v[D];
    row_load<D>(mine + r * D, v);
#pragma unroll
    for (int k = 0; k < D; ++k) mn[k] = hk_fmin(mn[k], v[k]);
  }
  float sub[D];
  bool any = false;
#pragma unroll
  for (int k = 0; k < D; ++k) {
    sub[k] = (mn[k] < INFINITY && (!jax_sem || mn[k] > 0.0f)) ? mn[k] : 0.0f;
    any |= sub[k] != 0.0f;
  }
  if (!any) return;
  for (int r = 0; r < n; ++r) {
    float v[D];
    row_load<D>(mine + r * D, v);
#pragma unroll
    for (int k = 0; k < D; ++k) v[k] = v[k] - sub[k];
    row_store<D>(mine + r * D, v);
  }
}

// _jax_ops.py:15-73 on n compact rows; acc[r] lives at acc_top[-r].  Survivors are compacted again;
// returns their number and clears the slot bits of the removed rows in `mask`.
template <int D>
__device__ inline int mid_newton(float* mine, int n, float* acc_top, Mask64& mask) {
  for (int r = 0; r < n; ++r) acc_top[-r] = INFINITY;
  for (int i = 0; i + 1 < n; ++i) {
    float qi[D];
    row_load<D>(mine + i * D, qi);
    float ai = acc_top[-i];
#pragma unroll 4
    for (int j = i + 1; j < n; ++j) {
      float qj[D];
      row_load<D>(mine + j * D, qj);
      float t = qi[0] - qj[0], u = t;
#pragma unroll
      for (int k = 1; k < D; ++k) {
        const float dk = qi[k] - qj[k];
        t = hk_fmax(t, dk);
        u = hk_fmin(u, dk);
      }
      acc_top[-j] = hk_fmin(acc_top[-j], t);        // j removed by i iff t <= 0
      ai = hk_fmin(ai, (t > 0.0f) ? -u : 1.0f);     // i removed by j iff u >= 0 and t > 0
    }
    acc_top[-i] = ai;
  }
  Mask64 mk = mask, alive = 0;
  int w = 0;
  for (int r = 0; r < n; ++r) {
    const int s = __ffsll(mk) - 1;
    mk &= mk - 1;
    if (acc_top[-r] > 0.0f) {
      alive |= (Mask64)1 << s;
      if (w != r) {
        float v[D];
        row_load<D>(mine + r * D, v);
        row_store<D>(mine + w * D, v);
      }
      ++w;
    }
  }
  mask = alive;
  return w;
}

// _jax_ops.py:93-111 / _torch_ops.py:136-146
template <int D>
__device__ inline void mid_rescale(float* mine, int n, unsigned flags) {
  const bool jax_sem = (flags & HK_SEM_MASK) == HK_SEM_JAX;
  float mx = -1.0f;
  for (int r = 0; r < n; ++r) {
    float v[D];
    row_load<D>(mine + r * D, v);
#pragma unroll
    for (int k = 0; k < D; ++k) mx = hk_fmax(mx, v[k]);
  }
  const bool skip = jax_sem ? (mx <= 1e-8f) : (mx < 0.0f);
  if (skip || mx == 0.0f) return;
  for (int r = 0; r < n; ++r) {
    float v[D];
    row_load<D>(mine + r * D, v);
#pragma unroll
    for (int k = 0; k < D; ++k) v[k] = v[k] / mx;
    row_store<D>(mine + r * D, v);
  }
}

template <int D, int MODE>
__global__ __launch_bounds__(kWave) void mid_kernel(const Params prm) {
  extern __shared__ __align__(16) unsigned char hk_smem[];
  float* lds = reinterpret_cast<float*>(hk_smem);
  const int lane = threadIdx.x;
  const int m = prm.m, n_el = m * D, S = prm.lds_stride, gpb = prm.games_per_block;
  float* cbuf = lds + gpb * S;  // slow path: subset mask / row scratch, D floats per lane
  const int64_t g0 = (int64_t)blockIdx.x * gpb;
  const int64_t left = (int64_t)prm.batch - g0;
  const int ngames = (int)(left < gpb ? left : gpb);
  const bool active = lane < ngames;
  const int64_t g = g0 + lane;
  const uint64_t gg = prm.game_offset + (uint64_t)g;
  float* mine = lds + (lane < gpb ? lane : 0) * S;
  const float pad = (float)prm.pad;
  const unsigned flags = prm.flags;
  const unsigned stages = (MODE == kModeGenerate) ? (prm.stages & ~HK_STAGE_SHIFT) : prm.stages;
  const float fill = ((flags & HK_SEM_MASK) == HK_SEM_TORCH) ? pad : -1.0f;
  const int nsteps = (MODE == kModeRollout) ? prm.steps : 1;
  const bool vec_in = (n_el % 4 == 0) && (prm.in_stride % 4 == 0) && prm.in &&
                      (reinterpret_cast<uintptr_t>(prm.in) % 16 == 0);
  const bool vec_out = (n_el % 4 == 0) && (prm.out_stride % 4 == 0) &&
                       (reinterpret_cast<uintptr_t>(prm.out) % 16 == 0);
  PolicyCache pcache;

  float c[D];
  int axis_in = -1;
#pragma unroll
  for (int k = 0; k < D; ++k) c[k] = 0.0f;
  RawActions<D> raw;
  const bool fetch_actions = MODE == kModeStep && (stages & HK_STAGE_SHIFT) && active;
  if (fetch_actions) fast_fetch_actions<D>(prm, g, m, raw);  // converted after the slab is requested

  // ---- 1. the image --------------------------------------------------------------------------------
  if (MODE == kModeGenerate) {
    if (active)
      for (int e = 0; e < n_el; e += 4) {
        const U4 r = philox4x32((uint32_t)gg, (uint32_t)(gg >> 32), (uint32_t)(e >> 2), kStreamGenerate,
                                prm.seed);
        const uint32_t w[4] = {r.x, r.y, r.z, r.w};
        for (int qd = 0; qd < 4 && e + qd < n_el; ++qd)
          mine[e + qd] = (float)mulhi32(w[qd], (uint32_t)prm.max_value);
      }
  } else {
    mid_copy_slab<true>(lds, const_cast<float*>((const float*)prm.in), prm.in_stride, n_el, S, g0, ngames, lane,
                        vec_in);
  }
  if (fetch_actions) fast_decode_actions<D>(prm, raw, c, axis_in);
  __syncthreads();

  // ---- 2. live rows, exactness guard ------------------------------------------------------------
  Mask64 gmask = 0;
  bool ok = true;
  if (active)
    for (int i = 0; i < m; ++i) {
      float v[D];
      row_load<D>(mine + i * D, v);
      bool ge = true, fl = true;
#pragma unroll
      for (int k = 0; k < D; ++k) {
        ge &= (__float_as_uint(v[k]) < 0x7F800000u);
        fl &= (v[k] == fill);
      }
      ok &= (ge | fl);
      gmask |= ge ? ((Mask64)1 << i) : 0ull;
    }
  int np = __popcll(gmask);
  const int nmax0 = wave_max(np, m);
  // the running minima need n floats behind the n compact rows: the lane's region is S floats (S >=
  // m*D + m when LDS is plentiful, else ~m*D and only Newton-reduced games fit -- plan_mid)
  const bool exact = (fill == pad) && __all(ok) && nmax0 * (D + 1) <= S;

  if (!exact) {
    // ---- slow path (whole wave): the exact generic routines on the image ------------------------
    float* cs = cbuf + lane * D;
    np = active ? num_points<float>(mine, m, D) : 2;
    int length = (np < 2) ? 0 : -1;
    if (MODE == kModeRollout && prm.count_ws) {
      const unsigned long long b0 = __ballot(active && np < 2);
      if (lane == 0) prm.count_ws[blockIdx.x] = (uint32_t)__popcll(b0);
    }
    for (int t = 0; t < nsteps; ++t) {
      int axis = -1, cls = 0;
      if (MODE == kModeRollout) {
        if (prm.obs_out) {
          __syncthreads();
          mid_copy_slab<false>(lds, (float*)prm.obs_out + (int64_t)t * prm.batch * n_el, (int64_t)n_el, n_el, S,
                               g0, ngames, lane, (n_el % 4 == 0) && (reinterpret_cast<uintptr_t>(prm.obs_out) % 16 == 0));
          __syncthreads();
        }
        uint32_t mask;
        fast_policy<D>(prm, gg, prm.step_offset + (uint32_t)t, pcache, cls, axis, mask);
        if (active)
          for (int k = 0; k < D; ++k) cs[k] = (float)((mask >> k) & 1u);
      } else if (MODE == kModeStep && (stages & HK_STAGE_SHIFT) && active) {
        load_coords<float>(prm, g, cs);
        axis = axis_in;
      }
      const bool prev_done = np < 2;
      if (active) stages_game<float>(mine, m, prm.d, cs, axis, pad, stages, flags);
      np = active ? num_points<float>(mine, m, prm.d) : 2;
      const bool done = np < 2;
      if (done && length < 0) length = t + 1;
      if (MODE == kModeRollout) {
        if (active) {
          const int64_t at = (int64_t)t * prm.batch + g;
          if (prm.r_host_class_out) prm.r_host_class_out[at] = cls;
          if (prm.r_axis_out) prm.r_axis_out[at] = axis;
          if (prm.r_done_out) prm.r_done_out[at] = done;
          if (prm.r_reward_out) prm.r_reward_out[at] = prm.reward_sign * (float)(done && !prev_done);
        }
        if (prm.count_ws) {
          const unsigned long long bd = __ballot(active && done);
          if (lane == 0) prm.count_ws[(size_t)(t + 1) * gridDim.x + blockIdx.x] = (uint32_t)__popcll(bd);
        }
      } else if (MODE == kModeStep && active) {
        if (prm.done_out) prm.done_out[g] = done;
        if (prm.prev_done_out) prm.prev_done_out[g] = prev_done;
        if (prm.reward_out) prm.reward_out[g] = prm.reward_sign * (float)(done && !prev_done);
        if (prm.num_points_out) prm.num_points_out[g] = np;
      }
    }
    if (MODE == kModeRollout && active && prm.game_length_out) prm.game_length_out[g] = length;
    __syncthreads();
    mid_copy_slab<false>(lds, (float*)prm.out, prm.out_stride, n_el, S, g0, ngames, lane, vec_out);
    return;
  }

  // ---- 3. compact the live rows in place ------------------------------------------------------------
  if (active) mid_compact<D>(mine, gmask);
  float* acc_top = mine + S - 1;
  if (!active) np = 2;  // never "done", never counted
  int nrows = active ? __popcll(gmask) : 0;
  int length = (np < 2) ? 0 : -1;
  if (MODE == kModeRollout && prm.count_ws) {
    const unsigned long long b0 = __ballot(active && np < 2);
    if (lane == 0) prm.count_ws[blockIdx.x] = (uint32_t)__popcll(b0);
  }

  // ---- 4. the transitions --------------------------------------------------------------------------
  for (int t = 0; t < nsteps; ++t) {
    int axis = -1, cls = 0;
    if (MODE == kModeRollout) {
      if (prm.obs_out) {  // state before the step: rebuild, store coalesced, compact again
        __syncthreads();
        if (active) mid_expand<D>(mine, gmask, nrows, m, pad);
        __syncthreads();
        mid_copy_slab<false>(lds, (float*)prm.obs_out + (int64_t)t * prm.batch * n_el, (int64_t)n_el, n_el, S, g0,
                             ngames, lane, (n_el % 4 == 0) && (reinterpret_cast<uintptr_t>(prm.obs_out) % 16 == 0));
        __syncthreads();
        if (active) mid_compact<D>(mine, gmask);
      }
      uint32_t mask;
      fast_policy<D>(prm, gg, prm.step_offset + (uint32_t)t, pcache, cls, axis, mask);
#pragma unroll
      for (int k = 0; k < D; ++k) c[k] = (float)((mask >> k) & 1u);
    } else if (MODE == kModeStep) {
      axis = axis_in;
    }
    const bool prev_done = np < 2;

    if (stages & HK_STAGE_SHIFT) mid_shift<D>(mine, nrows, c, axis, flags);
    if (stages & HK_STAGE_REPOSITION) mid_reposition<D>(mine, nrows, flags);
    if (stages & HK_STAGE_NEWTON) nrows = mid_newton<D>(mine, nrows, acc_top, gmask);
    if (stages & HK_STAGE_RESCALE) mid_rescale<D>(mine, nrows, flags);

    np = active ? nrows : 2;
    const bool done = np < 2;
    if (done && length < 0) length = t + 1;
    if (MODE == kModeRollout) {
      if (active) {
        const int64_t at = (int64_t)t * prm.batch + g;
        if (prm.r_host_class_out) prm.r_host_class_out[at] = cls;
        if (prm.r_axis_out) prm.r_axis_out[at] = axis;
        if (prm.r_done_out) prm.r_done_out[at] = done;
        if (prm.r_reward_out) prm.r_reward_out[at] = prm.reward_sign * (float)(done && !prev_done);
      }
      if (prm.count_ws) {
        const unsigned long long bd = __ballot(active && done);
        if (lane == 0) prm.count_ws[(size_t)(t + 1) * gridDim.x + blockIdx.x] = (uint32_t)__popcll(bd);
      }
    } else if (MODE == kModeStep && active) {
      if (prm.done_out) prm.done_out[g] = done;
      if (prm.prev_done_out) prm.prev_done_out[g] = prev_done;
      if (prm.reward_out) prm.reward_out[g] = prm.reward_sign * (float)(done && !prev_done);
      if (prm.num_points_out) prm.num_points_out[g] = np;
    }
  }
  if (MODE == kModeRollout && active && prm.game_length_out) prm.game_length_out[g] = length;

  // ---- 5. publish ----------------------------------------------------------------------------------
  __syncthreads();
  if (active) mid_expand<D>(mine, gmask, nrows, m, pad);
  __syncthreads();
  mid_copy_slab<false>(lds, (float*)prm.out, prm.out_stride, n_el, S, g0, ngames, lane, vec_out);
}

// ---- host side ---------------------------------------------------------------------------------------
inline bool mid_supported(const Params& prm, int dtype) {
  if (dtype != HK_F32) return false;
  if ((prm.flags & HK_SEM_MASK) == HK_SEM_LIST || (prm.flags & HK_FLAG_COMPACT_SORTED)) return false;
  if (prm.flags & HK_FLAG_FORCE_GENERIC) return false;
  if (prm.stages & kStageFeatureSort) return false;
  if (prm.mode == kModeZeillinger) return false;
  if (prm.mode == kModeRollout && prm.host_policy == HK_HOST_ZEILLINGER) return false;
  if (prm.d < 2 || prm.d > 6 || prm.m > 64) return false;
  return true;
}

// LDS geometry: stride a multiple of the row width with stride/width odd; + D floats per lane of scratch
inline int plan_mid(Params& prm) {
  const int w = prm.d == 4 ? 4 : (prm.d == 2 ? 2 : 1);
  const int n = prm.m * prm.d;
  // tuning hook (scripts/probe_stages.py): HK_MID_GAMES_PER_WAVE=8|16|32|64
  static const int forced = [] {
    const char* e = getenv("HK_MID_GAMES_PER_WAVE");
    const int v = e ? atoi(e) : 0;
    return (v == 8 || v == 16 || v == 32 || v == 64) ? v : 0;
  }();
  int gpb = forced ? forced : kWave;
  // room for one running minimum per row behind the image when the block then still fits
  // 64 KiB (>= 2 blocks per CU); big games keep the bare image and use its free tail instead
  int stride = n + prm.m;
  if ((int64_t)(stride + w + prm.d) * 4 * gpb > 64 * 1024) stride = n;
  stride = (stride + w - 1) / w * w;
  if (((stride / w) & 1) == 0) stride += w;
  while (gpb > 1 && (int64_t)(stride + prm.d) * 4 * gpb > kMaxLdsBytes) gpb >>= 1;
  if ((int64_t)(stride + prm.d) * 4 * gpb > kMaxLdsBytes) return HK_ERR_UNSUPPORTED;
  prm.lds_stride = stride;
  prm.games_per_block = gpb;
  return HK_OK;
}

template <int D, int MODE>
int launch_mid_t(const Params& prm, hipStream_t stream) {
  const size_t lds = (size_t)(prm.lds_stride + prm.d) * prm.games_per_block * sizeof(float);
  if (lds > 64 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&mid_kernel<D, MODE>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      (void)hipGetLastError();
      return HK_ERR_LAUNCH;
    }
  }
  const unsigned grid = (unsigned)(((int64_t)prm.batch + prm.games_per_block - 1) / prm.games_per_block);
  launch_prepare();
  hipLaunchKernelGGL((mid_kernel<D, MODE>), dim3(grid), dim3(kWave), lds, stream, prm);
  return launch_status();
}

template <int D>
int launch_mid_d(const Params& prm, hipStream_t stream) {
  if (prm.mode == kModeStep) return launch_mid_t<D, kModeStep>(prm, stream);
  if (prm.mode == kModeRollout) return launch_mid_t<D, kModeRollout>(prm, stream);
  return launch_mid_t<D, kModeGenerate>(prm, stream);
}

inline int launch_mid(Params& prm, hipStream_t stream) {
  const int st = plan_mid(prm);
  if (st != HK_OK) return st;
  switch (prm.d) {
    case 2: return launch_mid_d<2>(prm, stream);
    case 3: return launch_mid_d<3>(prm, stream);
    case 4: return launch_mid_d<4>(prm, stream);
    case 5: return launch_mid_d<5>(prm, stream);
    case 6: return launch_mid_d<6>(prm, stream);
  }
  return HK_ERR_UNSUPPORTED;
}

}  // namespace hk
