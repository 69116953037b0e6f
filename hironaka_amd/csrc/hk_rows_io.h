// Row and slab I/O shared by the kernels whose shape is a run-time value (hk_team_kernel.h):
// D-wide LDS row accesses and the coalesced HBM <-> LDS slab copy.
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

// coalesced HBM <-> LDS slab copy, run-time record length n; 16-B requests when the records allow
// The loop is rolled (n is a run-time value), so it is batched by hand: kCopyBatch independent
// requests per lane are issued before the first dependent LDS access, otherwise every iteration would
// expose a full HBM latency (measured on the (50,4) slab copy: 273 us one request at a time, ~80 us batched).
constexpr int kCopyBatchDefault = 16;
constexpr int kCopyBatchInLoop = 4;  // stores issued inside the rollout loop: keep the loop's register pressure low

template <bool TO_LDS, int kCopyBatch = kCopyBatchDefault>
__device__ inline void rows_copy_slab(float* lds, float* glob, int64_t gstride, int n, int S, int64_t g0,
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

}  // namespace hk
