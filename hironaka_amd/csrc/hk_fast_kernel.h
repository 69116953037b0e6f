// Register-resident specialisations of the Hironaka step for compile-time (max_points, dim),
// float32, in-place-order semantics (JAX / torch).  One lane per game, the game's M*D floats in
// VGPRs with compile-time indices for the whole launch (all T steps of a rollout), every loop
// fully unrolled.
//
// HBM <-> registers goes through an LDS transpose so that HBM only ever sees fully coalesced
// wave requests (lane l moves bytes [16l, 16l+16) of each KiB of the wave's contiguous slab;
// W = 4/2/1 dwords chosen from the divisibility of M*D) and each lane then pulls its own game out
// of LDS with W-wide ds_reads at a per-game stride chosen conflict-free for that width:
//   W=4  ds_read_b128 / ds_write_b128 : stride/4 odd   (pad one 16-B slot if M*D/4 is even)
//   W=2  ds_read_b64                  : stride/2 odd   (always true when M*D = 2 mod 4)
//   W=1  ds_read_b32                  : stride odd
//
// The Newton-polytope test is the O(M^2 D) part.  Per unordered pair i<j on the rows with
// unavailable rows mapped to +inf:   t = max_k(q_i - q_j),  w = max_k(q_j - q_i)
//   row j is removed by i   iff t <= 0                    (P_i <= P_j; ties go to the lower index)
//   row i is removed by j   iff w <= 0 and t > 0          (P_j <= P_i and not equal)
// accumulated as one running minimum per row in a VGPR (no lane-mask SGPR pressure, so the same
// code serves M = 50): D subs + 2 max3 + cmp + cndmask + 2 min per pair.  The sign of a float
// difference is exact, so this equals the reference's `diff >= 0` test (_jax_ops.py:55-56).
// Inputs the shortcut cannot represent exactly (rows that are neither fully >= 0 nor uniformly
// equal to the duplicate-fill value, or a padding value != the fill value under JAX semantics)
// are detected per wave and routed through the exact generic routine on the LDS image.
#pragma once

#include "hk_generic_kernel.h"

namespace hk {

template <int M, int D>
struct FastGeom {
  static constexpr int N = M * D;
  static constexpr int W = (N % 4 == 0) ? 4 : ((N % 2 == 0) ? 2 : 1);
  static constexpr int Q = N / W;                       // W-chunks per game
  static constexpr int S = (Q % 2 == 1) ? N : N + W;    // LDS stride in floats, S/W odd
  static constexpr int kLdsBytes = kWave * S * 4;
};

template <int W> struct VecOf;
template <> struct VecOf<4> { using type = float4; };
template <> struct VecOf<2> { using type = float2; };
template <> struct VecOf<1> { using type = float; };

// coalesced slab copy HBM -> LDS image (per-game stride S), vector width W
template <int M, int D>
__device__ inline void fast_load_slab(float* lds, const float* in, int64_t in_stride, int64_t g0,
                                      int ngames, int lane) {
  using G = FastGeom<M, D>;
  using V = typename VecOf<G::W>::type;
  const int total = ngames * G::Q;
#pragma unroll
  for (int it = 0; it < G::Q; ++it) {
    const int q = lane + it * kWave;
    if (q < total) {
      const int g = q / G::Q, c = q - g * G::Q;
      const V v = *reinterpret_cast<const V*>(in + (g0 + g) * in_stride + c * G::W);
      *reinterpret_cast<V*>(lds + g * G::S + c * G::W) = v;
    }
  }
}

template <int M, int D>
__device__ inline void fast_store_slab(const float* lds, float* out, int64_t out_stride, int64_t g0,
                                       int ngames, int lane) {
  using G = FastGeom<M, D>;
  using V = typename VecOf<G::W>::type;
  const int total = ngames * G::Q;
#pragma unroll
  for (int it = 0; it < G::Q; ++it) {
    const int q = lane + it * kWave;
    if (q < total) {
      const int g = q / G::Q, c = q - g * G::Q;
      const V v = *reinterpret_cast<const V*>(lds + g * G::S + c * G::W);
      *reinterpret_cast<V*>(out + (g0 + g) * out_stride + c * G::W) = v;
    }
  }
}

template <int M, int D>
__device__ inline void regs_from_lds(float (&p)[M * D], const float* mine) {
  using G = FastGeom<M, D>;
  using V = typename VecOf<G::W>::type;
#pragma unroll
  for (int c = 0; c < G::Q; ++c) {
    const V v = *reinterpret_cast<const V*>(mine + c * G::W);
    const float* f = reinterpret_cast<const float*>(&v);
#pragma unroll
    for (int w = 0; w < G::W; ++w) p[c * G::W + w] = f[w];
  }
}

template <int M, int D>
__device__ inline void regs_to_lds(const float (&p)[M * D], float* mine) {
  using G = FastGeom<M, D>;
  using V = typename VecOf<G::W>::type;
#pragma unroll
  for (int c = 0; c < G::Q; ++c) {
    V v;
    float* f = reinterpret_cast<float*>(&v);
#pragma unroll
    for (int w = 0; w < G::W; ++w) f[w] = p[c * G::W + w];
    *reinterpret_cast<V*>(mine + c * G::W) = v;
  }
}

template <int M, int D>
__device__ inline int fast_num_points(const float (&p)[M * D]) {
  int n = 0;
#pragma unroll
  for (int i = 0; i < M; ++i) n += (p[i * D] >= 0.0f) ? 1 : 0;
  return n;
}

// _jax_ops.py:76-90 / _torch_ops.py:46-110 in registers
template <int M, int D>
__device__ inline void fast_shift(float (&p)[M * D], const float (&c)[D], int axis, float pad,
                                  unsigned flags) {
  const bool torch_sem = (flags & HK_SEM_MASK) == HK_SEM_TORCH;
  if (torch_sem) pad = torch_pad(pad);
  bool apply = true;
  if (flags & HK_FLAG_AXIS_NOOP_IF_INVALID) {
#pragma unroll
    for (int k = 0; k < D; ++k) {
      const float onehot = (k == axis) ? 1.0f : 0.0f;
      if (!(onehot - c[k] <= 0.0f)) apply = false;
    }
  }
  if (flags & HK_FLAG_IGNORE_ENDED) {
    if (fast_num_points<M, D>(p) < 2) apply = false;
  }
  const int eff_axis = apply ? axis : -1;
#pragma unroll
  for (int i = 0; i < M; ++i) {
    float s = 0.0f;
    bool any = false;
#pragma unroll
    for (int k = 0; k < D; ++k) {
      s = s + p[i * D + k] * c[k];
      any |= (p[i * D + k] >= 0.0f);
    }
#pragma unroll
    for (int k = 0; k < D; ++k) {
      const float v = p[i * D + k];
      const bool avail = torch_sem ? (v >= 0.0f) : any;
      const float moved = (k == eff_axis) ? s : v;
      p[i * D + k] = avail ? moved : pad;
    }
  }
}

// _jax_ops.py:114-123 / _torch_ops.py:113-133
template <int M, int D>
__device__ inline void fast_reposition(float (&p)[M * D], float pad, unsigned flags) {
  const bool jax_sem = (flags & HK_SEM_MASK) == HK_SEM_JAX;
#pragma unroll
  for (int k = 0; k < D; ++k) {
    float mn = INFINITY;  // minimum over the entries >= 0 (+inf: none)
#pragma unroll
    for (int i = 0; i < M; ++i) {
      const float v = p[i * D + k];
      mn = fminf(mn, (v >= 0.0f) ? v : INFINITY);
    }
    // JAX: untouched when no entry is available or the minimum is <= 0 (:121)
    const bool touch = jax_sem ? (mn > 0.0f && mn < INFINITY) : true;
    const float sub = (mn < INFINITY) ? mn : 0.0f;
#pragma unroll
    for (int i = 0; i < M; ++i) {
      const float v = p[i * D + k];
      const float moved = (v >= 0.0f) ? v - sub : pad;
      p[i * D + k] = touch ? moved : v;
    }
  }
}

// _jax_ops.py:93-111 / _torch_ops.py:136-146
template <int M, int D>
__device__ inline void fast_rescale(float (&p)[M * D], float pad, unsigned flags) {
  const bool jax_sem = (flags & HK_SEM_MASK) == HK_SEM_JAX;
  float mx = p[0];
#pragma unroll
  for (int e = 1; e < M * D; ++e) mx = fmaxf(mx, p[e]);
  if (jax_sem) {
    const bool skip = (mx <= 1e-8f);
    const float div = skip ? 1.0f : mx;
#pragma unroll
    for (int i = 0; i < M; ++i) {
      bool any = false;
#pragma unroll
      for (int k = 0; k < D; ++k) any |= (p[i * D + k] >= 0.0f);
#pragma unroll
      for (int k = 0; k < D; ++k) {
        const float v = p[i * D + k];
        p[i * D + k] = any ? (skip ? v : v / div) : pad;
      }
    }
  } else {
    pad = torch_pad(pad);
    const float div = (mx == 0.0f) ? 1.0f : mx;
#pragma unroll
    for (int e = 0; e < M * D; ++e) {
      const float v = p[e];
      p[e] = (v >= 0.0f) ? v / div : pad;
    }
  }
}

// t = max_k(a_k - b_k), u = min_k(a_k - b_k) from ONE set of differences
template <int D>
__device__ inline void diff_extrema(const float* a, const float* b, float& t, float& u) {
  const float d0 = a[0] - b[0];
  t = d0;
  u = d0;
#pragma unroll
  for (int k = 1; k < D; ++k) {
    const float dk = a[k] - b[k];
    t = fmaxf(t, dk);
    u = fminf(u, dk);
  }
}

// true iff every row of this lane's game is either fully available and finite, or uniformly equal
// to the duplicate-fill value: exactly the inputs fast_newton reproduces bit for bit
template <int M, int D>
__device__ inline bool fast_newton_representable(const float (&p)[M * D], float fill) {
  bool ok = true;
#pragma unroll
  for (int i = 0; i < M; ++i) {
    bool all_ge = true, all_fill = true;
#pragma unroll
    for (int k = 0; k < D; ++k) {
      all_ge &= (p[i * D + k] >= 0.0f) & (p[i * D + k] < INFINITY);
      all_fill &= (p[i * D + k] == fill);
    }
    ok &= (all_ge | all_fill);
  }
  return ok;
}

// Newton polytope in registers on a representable game (fill == padw < 0).  Unavailable rows are
// mapped to +inf in place (they can neither dominate nor be the reason a finite row is kept) and
// restored to `fill` at the end.
template <int M, int D>
__device__ inline void fast_newton(float (&p)[M * D], float padw) {
  float acc[M];
#pragma unroll
  for (int i = 0; i < M; ++i) {
    const bool avail = (p[i * D] >= 0.0f);
    acc[i] = INFINITY;
#pragma unroll
    for (int k = 0; k < D; ++k) p[i * D + k] = avail ? p[i * D + k] : INFINITY;
  }
#pragma unroll
  for (int i = 0; i < M - 1; ++i) {
#pragma unroll
    for (int j = i + 1; j < M; ++j) {
      float t, u;  // extrema of P_i - P_j: t <= 0 <=> P_i <= P_j ; u >= 0 <=> P_j <= P_i
      diff_extrema<D>(&p[i * D], &p[j * D], t, u);
      acc[j] = fminf(acc[j], t);
      acc[i] = fminf(acc[i], (t > 0.0f) ? -u : 1.0f);
    }
  }
#pragma unroll
  for (int i = 0; i < M; ++i) {
    const bool keep = (p[i * D] < INFINITY) & !(acc[i] <= 0.0f);
#pragma unroll
    for (int k = 0; k < D; ++k) p[i * D + k] = keep ? p[i * D + k] : padw;
  }
}

// stages on the register state; `mine` is this lane's LDS image (slow-path scratch)
template <int M, int D>
__device__ inline void fast_stages(float (&p)[M * D], const float (&c)[D], int axis, float pad,
                                   unsigned stages, unsigned flags, float* mine, int m_rt, int d_rt) {
  if (stages & HK_STAGE_SHIFT) fast_shift<M, D>(p, c, axis, pad, flags);
  if (stages & HK_STAGE_REPOSITION) fast_reposition<M, D>(p, pad, flags);
  if (stages & HK_STAGE_NEWTON) {
    const bool torch_sem = (flags & HK_SEM_MASK) == HK_SEM_TORCH;
    const float padw = torch_sem ? torch_pad(pad) : pad;
    const float fill = torch_sem ? padw : -1.0f;
    const bool ok = (fill == padw) && fast_newton_representable<M, D>(p, fill);
    if (__all(ok)) {
      fast_newton<M, D>(p, padw);
    } else {  // rare: exact generic routine on the LDS image, whole wave
      regs_to_lds<M, D>(p, mine);
      newton_game<float>(mine, m_rt, d_rt, pad, flags);  // runtime bounds: keep it rolled
      regs_from_lds<M, D>(p, mine);
    }
  }
  if (stages & HK_STAGE_RESCALE) fast_rescale<M, D>(p, pad, flags);
}

template <int D>
__device__ inline void fast_load_coords(const Params& prm, int64_t g, int m, float (&c)[D]) {
  const int kind = prm.coords_kind;
  if (kind == HK_COORDS_CLASS_I32 || kind == HK_COORDS_CLASS_I64) {
    long long cls = (kind == HK_COORDS_CLASS_I32) ? (long long)((const int32_t*)prm.coords)[g]
                                                  : ((const long long*)prm.coords)[g];
    constexpr long long ncls = (1ll << D) - D - 1;
    cls = cls < 0 ? 0 : (cls >= ncls ? ncls - 1 : cls);
    const uint32_t v = decode_class((int)cls, D);
#pragma unroll
    for (int k = 0; k < D; ++k) c[k] = (float)((v >> k) & 1u);
  } else if (kind == HK_COORDS_IN_RECORD) {
    const float* rec = (const float*)prm.in + g * prm.in_stride + (int64_t)m * D;
#pragma unroll
    for (int k = 0; k < D; ++k) c[k] = rec[k];
  } else {
#pragma unroll
    for (int k = 0; k < D; ++k)
      c[k] = (float)load_scalar(prm.coords, kind, (size_t)(g * prm.coords_stride + k));
  }
}

template <int M, int D>
__global__ __launch_bounds__(kWave) void fast_kernel(const Params prm) {
  using G = FastGeom<M, D>;
  __shared__ __align__(16) float lds[kWave * G::S];
  const int lane = threadIdx.x;
  const int64_t g0 = (int64_t)blockIdx.x * kWave;
  const int64_t left = (int64_t)prm.batch - g0;
  const int ngames = (int)(left < kWave ? left : kWave);
  const bool active = lane < ngames;
  const int64_t g = g0 + lane;
  const uint64_t gg = prm.game_offset + (uint64_t)g;
  float* mine = lds + lane * G::S;
  const float pad = (float)prm.pad;
  const int mode = prm.mode;
  constexpr uint32_t ncls = (1u << D) - (uint32_t)D - 1u;
  float p[M * D];
  float c[D];
#pragma unroll
  for (int k = 0; k < D; ++k) c[k] = 0.0f;

  // ---- bring the state into registers ---------------------------------------------------------
  if (mode == kModeGenerate) {
#pragma unroll
    for (int e = 0; e < M * D; e += 4) {
      const U4 r = philox4x32((uint32_t)gg, (uint32_t)(gg >> 32), (uint32_t)(e >> 2),
                              kStreamGenerate, prm.seed);
      const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
      for (int qd = 0; qd < 4; ++qd)
        if (e + qd < M * D) p[e + qd] = (float)mulhi32(w[qd], (uint32_t)prm.max_value);
    }
  } else {
    fast_load_slab<M, D>(lds, (const float*)prm.in, prm.in_stride, g0, ngames, lane);
    __syncthreads();
    // inactive lanes of the last wave read stale LDS; they compute but never publish
    regs_from_lds<M, D>(p, mine);
  }

  // ---- one (step / generate) or `steps` (rollout) transitions, a single inlined call site -----
  const int nsteps = (mode == kModeRollout) ? prm.steps : 1;
  unsigned stages = prm.stages;
  if (mode == kModeGenerate) stages &= ~HK_STAGE_SHIFT;
  int np = active ? fast_num_points<M, D>(p) : 2;
  int length = (np < 2) ? 0 : -1;
  if (mode == kModeRollout && prm.count_ws) {
    const unsigned long long b0 = __ballot(active && np < 2);
    if (lane == 0) prm.count_ws[blockIdx.x] = (uint32_t)__popcll(b0);
  }
  for (int t = 0; t < nsteps; ++t) {
    int axis = -1, cls = 0;
    if (mode == kModeRollout) {
      if (prm.obs_out) {  // state before the step, coalesced through the LDS image
        __syncthreads();
        regs_to_lds<M, D>(p, mine);
        __syncthreads();
        fast_store_slab<M, D>(lds, (float*)prm.obs_out + (int64_t)t * prm.batch * G::N,
                              (int64_t)G::N, g0, ngames, lane);
      }
      const U4 r = philox4x32((uint32_t)gg, (uint32_t)(gg >> 32), prm.step_offset + (uint32_t)t,
                              kStreamPolicy, prm.seed);
      cls = (prm.host_policy == HK_HOST_RANDOM) ? (int)mulhi32(r.x, ncls) : (int)ncls - 1;
      const uint32_t mask = decode_class(cls, D);
      if (prm.agent_policy == HK_AGENT_RANDOM) {
        axis = (int)mulhi32(r.y, (uint32_t)D);
      } else if (prm.agent_policy == HK_AGENT_RANDOM_LEGAL) {
        const int pick = (int)mulhi32(r.y, (uint32_t)__popc(mask));
        int seen = 0;
        axis = 0;
#pragma unroll
        for (int k = 0; k < D; ++k)
          if ((mask >> k) & 1u) {
            if (seen == pick) axis = k;
            ++seen;
          }
      } else if (prm.agent_policy == HK_AGENT_CHOOSE_FIRST) {
        axis = __ffs(mask) - 1;
      } else {
        axis = 31 - __clz(mask);
      }
#pragma unroll
      for (int k = 0; k < D; ++k) c[k] = (float)((mask >> k) & 1u);
    } else if (mode == kModeStep && (stages & HK_STAGE_SHIFT) && active) {
      fast_load_coords<D>(prm, g, M, c);
      axis = axis_index(load_scalar(prm.axis, prm.axis_dtype, (size_t)g), D);
    }
    const bool prev_done = np < 2;

    fast_stages<M, D>(p, c, axis, pad, stages, prm.flags, mine, prm.m, prm.d);

    np = active ? fast_num_points<M, D>(p) : 2;
    const bool done = np < 2;
    if (done && length < 0) length = t + 1;
    if (mode == kModeRollout) {
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
    } else if (mode == kModeStep && active) {
      if (prm.done_out) prm.done_out[g] = done;
      if (prm.prev_done_out) prm.prev_done_out[g] = prev_done;
      if (prm.reward_out) prm.reward_out[g] = prm.reward_sign * (float)(done && !prev_done);
      if (prm.num_points_out) prm.num_points_out[g] = np;
    }
  }
  if (mode == kModeRollout && active && prm.game_length_out) prm.game_length_out[g] = length;

  // ---- publish the state ------------------------------------------------------------------------
  __syncthreads();
  regs_to_lds<M, D>(p, mine);
  __syncthreads();
  fast_store_slab<M, D>(lds, (float*)prm.out, prm.out_stride, g0, ngames, lane);
}

// ---- the specialisation table ------------------------------------------------------------------
// (max_points, dim): BASELINE configs (10,3) (20,3) (50,4) plus the small shapes the reference's
// tests and YAMLs use.
#define HK_FAST_SPECS(X) X(4, 3) X(4, 4) X(5, 3) X(6, 3) X(8, 3) X(10, 3) X(16, 3) X(20, 3) X(8, 4) X(20, 4) X(50, 4)

inline int has_fast_path(int m, int d, int dtype) {
  if (dtype != HK_F32) return 0;
#define HK_X(M_, D_) if (m == M_ && d == D_) return 1;
  HK_FAST_SPECS(HK_X)
#undef HK_X
  return 0;
}

// vector slab I/O needs W-aligned records
template <int M, int D>
bool fast_aligned_t(const Params& prm) {
  using G = FastGeom<M, D>;
  const size_t vec_bytes = G::W * 4;
  if ((prm.in && (reinterpret_cast<uintptr_t>(prm.in) % vec_bytes)) ||
      (reinterpret_cast<uintptr_t>(prm.out) % vec_bytes) || (prm.in_stride % G::W) ||
      (prm.out_stride % G::W))
    return false;
  if (prm.obs_out && (reinterpret_cast<uintptr_t>(prm.obs_out) % vec_bytes)) return false;
  return true;
}

template <int M, int D>
int launch_fast_t(const Params& prm, hipStream_t stream) {
  const unsigned grid = (unsigned)(((int64_t)prm.batch + kWave - 1) / kWave);
  launch_prepare();
  hipLaunchKernelGGL((fast_kernel<M, D>), dim3(grid), dim3(kWave), 0, stream, prm);
  return launch_status();
}

// does this request run on a register-resident specialisation? (else: generic kernel)
inline bool fast_supported(const Params& prm, int dtype) {
  if (dtype != HK_F32) return false;
  if ((prm.flags & HK_SEM_MASK) == HK_SEM_LIST || (prm.flags & HK_FLAG_COMPACT_SORTED)) return false;
  if (prm.flags & HK_FLAG_FORCE_GENERIC) return false;
  if (prm.stages & kStageFeatureSort) return false;
  if (prm.mode == kModeZeillinger) return false;
  if (prm.mode == kModeRollout && prm.host_policy == HK_HOST_ZEILLINGER) return false;
#define HK_X(M_, D_) if (prm.m == M_ && prm.d == D_) return fast_aligned_t<M_, D_>(prm);
  HK_FAST_SPECS(HK_X)
#undef HK_X
  return false;
}

inline int launch_fast(const Params& prm, hipStream_t stream) {
#define HK_X(M_, D_) if (prm.m == M_ && prm.d == D_) return launch_fast_t<M_, D_>(prm, stream);
  HK_FAST_SPECS(HK_X)
#undef HK_X
  return HK_ERR_UNSUPPORTED;
}

// ---- finished-game counters: per-workgroup partials -> done_count ---------------------------------
// block t sums count_ws[t][0..nblocks) and adds it to done_count[t]: steps+1 atomics in total
// instead of (steps+1) * nblocks on one cache line.
__global__ __launch_bounds__(256) void count_reduce_kernel(const uint32_t* ws, int nblocks,
                                                           unsigned long long* done_count) {
  __shared__ unsigned long long part[256 / kWave];
  const uint32_t* row = ws + (size_t)blockIdx.x * nblocks;
  unsigned long long s = 0;
  for (int i = threadIdx.x; i < nblocks; i += 256) s += row[i];
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) s += __shfl_down(s, off, kWave);
  if ((threadIdx.x & (kWave - 1)) == 0) part[threadIdx.x / kWave] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long tot = 0;
    for (int w = 0; w < 256 / kWave; ++w) tot += part[w];
    if (tot) atomicAdd(&done_count[blockIdx.x], tot);
  }
}

inline int launch_count_reduce(const uint32_t* ws, int nblocks, int steps, unsigned long long* done_count,
                               hipStream_t stream) {
  launch_prepare();
  hipLaunchKernelGGL(count_reduce_kernel, dim3(steps + 1), dim3(256), 0, stream, ws, nblocks, done_count);
  return launch_status();
}

// ---- small utility kernels ------------------------------------------------------------------------

template <typename T>
__global__ void counts_kernel(const T* points, int64_t stride, uint8_t* done_out, int32_t* num_out,
                              int batch, int m, int d) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= batch) return;
  const T* p = points + g * stride;
  int n = 0;
  for (int i = 0; i < m; ++i) n += (p[(int64_t)i * d] >= (T)0) ? 1 : 0;
  if (done_out) done_out[g] = n < 2;
  if (num_out) num_out[g] = n;
}

inline int launch_counts(const void* points, int64_t stride, uint8_t* done_out, int32_t* num_out,
                         int batch, int m, int d, int dtype, hipStream_t stream) {
  const unsigned grid = (unsigned)((batch + 255) / 256);
  launch_prepare();
  if (dtype == HK_F32)
    hipLaunchKernelGGL(counts_kernel<float>, dim3(grid), dim3(256), 0, stream, (const float*)points,
                       stride, done_out, num_out, batch, m, d);
  else
    hipLaunchKernelGGL(counts_kernel<double>, dim3(grid), dim3(256), 0, stream, (const double*)points,
                       stride, done_out, num_out, batch, m, d);
  return launch_status();
}

__global__ void decode_kernel(const int32_t* cls, void* mask_out, int mask_dtype, int batch, int d) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)batch * d) return;
  const int g = (int)(idx / d), k = (int)(idx % d);
  const int ncls = (int)((1ll << d) - d - 1);
  int c = cls[g];
  c = c < 0 ? 0 : (c >= ncls ? ncls - 1 : c);
  const int bit = (decode_class(c, d) >> k) & 1u;
  switch (mask_dtype) {
    case HK_F32: ((float*)mask_out)[idx] = (float)bit; break;
    case HK_F64: ((double*)mask_out)[idx] = (double)bit; break;
    case HK_I32: ((int32_t*)mask_out)[idx] = bit; break;
    case HK_I64: ((long long*)mask_out)[idx] = bit; break;
    default: ((uint8_t*)mask_out)[idx] = (uint8_t)bit; break;
  }
}

inline int launch_decode(const int32_t* cls, void* mask_out, int mask_dtype, int batch, int d,
                         hipStream_t stream) {
  const int64_t total = (int64_t)batch * d;
  const unsigned grid = (unsigned)((total + 255) / 256);
  launch_prepare();
  hipLaunchKernelGGL(decode_kernel, dim3(grid), dim3(256), 0, stream, cls, mask_out, mask_dtype, batch, d);
  return launch_status();
}

}  // namespace hk
