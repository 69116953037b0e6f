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
  // Large games ((50,4): 12.8 KB of image per wave) keep ONE region per wave: the compact image lies over the
  // slab image (every lane has read its rows by then), without the index tag, and the result returns through the
  // compact slots to the lane that owns each row, which rebuilds its part of the image.  Small games keep the two
  // images apart and write the result back in place (fewer LDS operations).
  static constexpr bool kBig = N * kQuadGames * 4 > 8 * 1024;
  static constexpr int CW = kBig ? ((D <= 4) ? 4 : D) : ((D <= 3) ? 4 : D + 1);  // compact row (+ index tag when small)
  static constexpr int kImage = kQuadGames * N;                        // floats
  static constexpr int kCompact = kBig ? 0 : kQuadGames * M * CW;      // floats (kBig: aliased over the image)
  static constexpr int kRegion = kBig ? kQuadGames * (N > M * CW ? N : M * CW) : kImage + kCompact;
  static constexpr int kGameStride = kBig ? (N > M * CW ? N : M * CW) : N;  // floats between two games' images
  static constexpr int kStoreBatch = QL < 12 ? QL : 10;                // slab chunks per store round
  // waves per SIMD the register budget is set for: what the LDS footprint lets a CU hold (160 KB; four waves per
  // workgroup, one per SIMD), at most four; two for the large games
  static constexpr int kLdsPerGroup = 4 * (kRegion * 4 + kQuadGames * D * 4);
  static constexpr int kWavesPerSimd = kBig ? 2 : ((160 * 1024 / kLdsPerGroup) < 4 ? (160 * 1024 / kLdsPerGroup) : 4);
  // buckets of straight-line bodies (slots per lane)
  static constexpr int next_bucket(int nb) { return nb < 6 ? nb + 1 : (nb < 10 ? nb + 2 : nb + 3); }
  static_assert(!kBig || kGameStride == N, "the aliased layout assumes the compact rows fit the image");
};

// ---- slab I/O: 16 consecutive games = one contiguous piece of HBM (contiguous records only: the dispatcher sends
// strided records to the other kernels) ----------------------------------------------------------------------------
// in: LDS-DMA (global_load_lds_dwordx4: lane l of request `it` moves 16 B to image + (it * 64 + l) * 16 -- the image
// IS the slab, no registers, no ds_write); chunks past a partial slab re-read its last chunk into image space that
// nobody looks at.  Completion is the wave's vmcnt.
// Addressing: `base` and `image` are wave-uniform (the wave index comes through readfirstlane), the lane adds an
// unsigned 32-bit byte offset, and the requests of a lane differ by a compile-time constant: one VGPR offset for the
// whole slab, the global address in SGPRs (saddr form), the rest in the instructions' immediate offsets -- the
// per-request 64-bit address arithmetic was a third of the VALU instructions of the I/O skeleton, and at four waves
// per SIMD every VALU instruction of the wave program is 16 cycles of the launch.
// the instruction's immediate offset is 12 bits unsigned here (it moves the LDS address too); what exceeds it goes
// into both base pointers
constexpr int kDmaImm = 4096;
template <int BYTES, int OFFSET>
__device__ __forceinline__ void lds_dma(const float* src, float* dst) {
  static_assert(BYTES == 16 || BYTES == 4, "request size");
  if constexpr (BYTES == 16) __builtin_amdgcn_global_load_lds(src, dst, 16, OFFSET, 0);
  else __builtin_amdgcn_global_load_lds(src, dst, 4, OFFSET, 0);
}

template <int M, int D>
__device__ __forceinline__ void quad_slab_load(const float* base, float* image, int ngames, int lane) {
  using G = QuadGeom<M, D>;
  constexpr int CH = (G::W == 4) ? 4 : 1;                    // floats per request (16 B, or dwords when 16 does not divide a record)
  constexpr int FULL = kQuadGames * G::N / CH;               // requests of a full slab
  constexpr int QF = (FULL + kWave - 1) / kWave;             // ... per lane
  const float* src = base + (unsigned)lane * CH;             // request `it`: + it * 64 * CH floats, in the immediate
  const unsigned total = (unsigned)ngames * (G::N / CH);
  // the immediate offset moves the global AND the LDS address: request `it` lands at image + (it * 64 + lane) * CH
  if (ngames == kQuadGames) {  // (wave-uniform) every request but the last one whole
    unrolled_while<0, QF>([&](auto ic) {
      constexpr int it = decltype(ic)::value;
      if ((it + 1) * kWave <= FULL || lane < FULL - it * kWave)
        lds_dma<CH * 4, (it * kWave * CH * 4) % kDmaImm>(src + (it * kWave * CH * 4) / kDmaImm * (kDmaImm / 4),
                                                        image + (it * kWave * CH * 4) / kDmaImm * (kDmaImm / 4));
      return true;
    });
  } else {  // the batch's last slab: lanes past its end sit out
    unrolled_while<0, QF>([&](auto ic) {
      constexpr int it = decltype(ic)::value;
      if ((unsigned)lane + it * kWave < total)
        lds_dma<CH * 4, (it * kWave * CH * 4) % kDmaImm>(src + (it * kWave * CH * 4) / kDmaImm * (kDmaImm / 4),
                                                        image + (it * kWave * CH * 4) / kDmaImm * (kDmaImm / 4));
      return true;
    });
  }
}

__device__ __forceinline__ void wait_vmem_all() { __builtin_amdgcn_s_waitcnt(0x0F70); }  // vmcnt(0)

// NT: non-temporal stores (write-once streams: the per-step observations of a recording rollout)
template <int M, int D, bool NT = false>
__device__ __forceinline__ void quad_slab_store(const float* image, float* base, int ngames, int lane) {
  using G = QuadGeom<M, D>;
  using V = typename VecOf<G::W>::type;
  constexpr int B = G::kStoreBatch;
  constexpr int FULL = kQuadGames * G::Q;
  const unsigned total = (unsigned)ngames * G::Q;
  const float* src = image + (unsigned)lane * G::W;  // chunk `it` of the lane: + it * 64 * W floats, an immediate
  float* dst = base + (unsigned)lane * G::W;
  const bool whole = ngames == kQuadGames;           // (wave-uniform)
#pragma unroll
  for (int i0 = 0; i0 < G::QL; i0 += B) {
    V v[B];
#pragma unroll
    for (int u = 0; u < B; ++u) {
      // (chunks past the image: the read lands in LDS nobody owns or past the allocation -- harmless, never stored)
      if (i0 + u < G::QL) v[u] = *reinterpret_cast<const V*>(src + (i0 + u) * kWave * G::W);
    }
#pragma unroll
    for (int u = 0; u < B; ++u) asm volatile("" : "+v"(v[u]));
    if (whole) {
#pragma unroll
      for (int u = 0; u < B; ++u) {
        const int it = i0 + u;
        if (it < G::QL && ((it + 1) * kWave <= FULL || lane < FULL - it * kWave)) {
          if constexpr (NT) __builtin_nontemporal_store(v[u], reinterpret_cast<V*>(dst + it * kWave * G::W));
          else *reinterpret_cast<V*>(dst + it * kWave * G::W) = v[u];
        }
      }
    } else {
#pragma unroll
      for (int u = 0; u < B; ++u) {
        const int it = i0 + u;
        if (it < G::QL && (unsigned)lane + it * kWave < total) {
          if constexpr (NT) __builtin_nontemporal_store(v[u], reinterpret_cast<V*>(dst + it * kWave * G::W));
          else *reinterpret_cast<V*>(dst + it * kWave * G::W) = v[u];
        }
      }
    }
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
    // a correctly rounded division is ~10 instructions on ~8 registers: one row at a time (interleaved over all the
    // slots they pushed the run-time configured kernels over their 128 registers, into scratch)
    __builtin_amdgcn_sched_barrier(0);
  }
}

// t = max_k(a - b), u = min_k(a - b).  The differences are results of a subtraction (canonical), so the plain
// fmaxf / fminf chains become v_max3_f32 / v_min3_f32 without the canonicalising v_max x, x that loaded values get.
template <int D>
__device__ __forceinline__ void qd_extrema(const float* a, const float* b, float& t, float& u) {
  float dk[D];
#pragma unroll
  for (int k = 0; k < D; ++k) dk[k] = a[k] - b[k];
  t = dk[0];
  u = dk[0];
#pragma unroll
  for (int k = 1; k < D; ++k) {
    t = __builtin_fmaxf(t, dk[k]);
    u = __builtin_fminf(u, dk[k]);
  }
}

// One pair (mine, other) with t = max_k(mine - other), u = min_k(mine - other) (see d_newton in hk_duo_kernel.h):
//   mine earlier:  other removed iff t <= 0;            mine removed iff u >= 0 and t > 0
//   other earlier: other removed iff t <= 0 and u < 0;  mine removed iff u >= 0
// `acc` / `oth` are running minima; <= 0 means removed.  ORDER: +1 mine earlier, -1 other earlier, 0: `late` decides
// at run time (true: the other row is the earlier one).
template <int D, int ORDER, bool ACC_FIRST = false, bool OTH_FIRST = false>
__device__ __forceinline__ void qd_pair(const float* mine, const float* other, float& acc, float& oth, bool late) {
  float t, u;
  qd_extrema<D>(mine, other, t, u);
  float vo, va;
  if (ORDER > 0) {
    vo = t;
    va = (t > 0.0f) ? -u : 1.0f;
  } else if (ORDER < 0) {
    vo = (u < 0.0f) ? t : 1.0f;
    va = -u;
  } else {
    vo = (u < 0.0f || !late) ? t : 1.0f;
    va = (t > 0.0f || late) ? -u : 1.0f;
  }
  // (*_FIRST: the accumulator's first contribution is an assignment -- no +inf start, no min(+inf, x))
  oth = OTH_FIRST ? vo : hk_fmin(oth, vo);
  acc = ACC_FIRST ? va : hk_fmin(acc, va);
}

// _jax_ops.py:15-73 across the quad.  Rank of (slot s, lane j) = 4 s + j.
//   own slots a < b:                 a earlier
//   lane one up, all (a, b):         my rank 4a + j, theirs 4b + (j+1)%4: mine earlier iff a < b, or a == b and j != 3
//   lane two up, a <= b:             mine earlier iff a < b, or a == b and j < 2
template <int R, int D, int NB>
__device__ __forceinline__ void qd_newton(float (&q)[R * D], int j) {
  // (accumulators: first contribution by assignment, known at compile time; see d_newton)
  float acc[NB], o1[NB], o2[NB];
#pragma unroll
  for (int a = 0; a + 1 < NB; ++a) {
#pragma unroll
    for (int b = a + 1; b < NB; ++b) {
      float t, u;
      qd_extrema<D>(&q[a * D], &q[b * D], t, u);
      const float va = (t > 0.0f) ? -u : 1.0f;
      acc[b] = (a == 0) ? t : hk_fmin(acc[b], t);
      acc[a] = (a == 0 && b == 1) ? va : hk_fmin(acc[a], va);
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
    unrolled_while<0, NB>([&](auto ac) {  // (compile-time a: which contribution is an accumulator's first)
      constexpr int a = decltype(ac)::value;
      constexpr bool of = a == 0;        // o1[b], o2[b]: slot a = 0 comes first (0 <= b)
      constexpr bool af = NB == 1;       // acc[a]: the own-slot triangle came first, unless there is none
      // (b is a run-time loop index of an unrolled loop; the three cases are resolved when it is unrolled)
      if (a < b) {
        qd_pair<D, +1, false, of>(&q[a * D], p1, acc[a], o1[b], false);
        qd_pair<D, +1, false, of>(&q[a * D], p2, acc[a], o2[b], false);
      } else if (a == b) {
        qd_pair<D, 0, af, of>(&q[a * D], p1, acc[a], o1[b], late1);
        qd_pair<D, 0, false, of>(&q[a * D], p2, acc[a], o2[b], late2);
      } else {
        qd_pair<D, -1, false, of>(&q[a * D], p1, acc[a], o1[b], true);
      }
      // many registers per lane: one pair of tests at a time (the scheduler would interleave the whole row of them)
      if constexpr (R * D > 32) __builtin_amdgcn_sched_barrier(0);
      return true;
    });
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int r = 0; r < NB; ++r) {
    // what the lane one DOWN found out about my slot r (its "one up" is me), and the lane two up; compares, not a
    // float minimum of values that came through DPP (those are canonicalised first)
    const bool r0 = acc[r] <= 0.0f, r1 = qperm<kQuadUp3>(o1[r]) <= 0.0f, r2 = qperm<kQuadUp2>(o2[r]) <= 0.0f;
    const bool removed = r0 || r1 || r2;
#pragma unroll
    for (int k = 0; k < D; ++k) q[r * D + k] = removed ? INFINITY : q[r * D + k];
  }
}

// The same test for MANY slots per lane (more than kQuadDppSlots: states that no Newton pass has thinned yet), as a
// rolled loop over the game's rows through the compact image instead of 2 NB^2 unrolled pair tests: the quad parks
// its rows at their ranks, then rows j = 4(c-1) .. 4c-1 (one broadcast read per j) meet the lane's slots 0 .. c-1,
// i.e. the rows i < j only -- each unordered pair once, one set of differences for both directions:
//     t = max_k(P_j - P_i), u = min_k(P_j - P_i);   j is removed by i iff u >= 0;   i by j iff t <= 0 and u < 0
// (of two equal rows the later one goes).  Verdicts on the lane's own rows go to acc[]; those on row j, which another
// lane owns, are bits of jmask, OR-ed over the quad afterwards.  The segment for c slots is straight-line in the slots.
constexpr int kQuadDppSlots = 8;

template <int CW, int R, int D, int NB, int CSEG>
struct QuadLdsPairs {
  static __device__ __forceinline__ void run(const float (&q)[R * D], float (&acc)[NB], uint32_t (&jmask)[2],
                                             const float* cmine, int j, int rows_end) {
    constexpr int r0 = kQuad * (CSEG - 1);
    if (rows_end <= r0) return;
    const int r1 = rows_end < kQuad * CSEG ? rows_end : kQuad * CSEG;
#pragma nounroll
    for (int row = r0; row < r1; ++row) {
      float pj[D];
      if constexpr (D == 4) {
        const vf4 v = *reinterpret_cast<const vf4*>(cmine + row * CW);
        pj[0] = v.x; pj[1] = v.y; pj[2] = v.z; pj[3] = v.w;
      } else {
#pragma unroll
        for (int k = 0; k < D; ++k) pj[k] = cmine[row * CW + k];
      }
      bool jdead = false;
#pragma unroll
      for (int s = 0; s < CSEG; ++s) {
        float t, u;
        qd_extrema<D>(pj, &q[s * D], t, u);
        const bool below = (s < CSEG - 1) || (r0 + j < row);  // my row 4s + j lies below row j
        jdead |= below && (u >= 0.0f);
        acc[s] = hk_fmin(acc[s], (below && u < 0.0f) ? t : 1.0f);
      }
      const uint32_t bit = jdead ? (1u << (row & 31)) : 0u;
      if (row < 32) jmask[0] |= bit;
      else jmask[1] |= bit;
    }
    if constexpr (CSEG < NB) QuadLdsPairs<CW, R, D, NB, CSEG + 1>::run(q, acc, jmask, cmine, j, rows_end);
  }
};

// PIN (callers inside a step / episode loop): the parked rows' address is made opaque here -- hoisted out of a rollout's
// step loops, the segments' start addresses (one register each, live through the whole staircase) were part of what the
// (50,4) rollout kernels spilled
template <int M, int CW, int R, int D, int NB, bool PIN = false>
__device__ __forceinline__ void qd_newton_lds(float (&q)[R * D], float* cmine, int j, int rows_end) {
  static_assert(kQuad * NB <= 64, "verdicts on row j travel as 64 bits");
  if constexpr (PIN) asm volatile("" : "+v"(cmine));
#pragma unroll
  for (int s = 0; s < NB; ++s) {
    if (kQuad * s + j < M) {
      float* dst = cmine + (kQuad * s + j) * CW;
      if constexpr (D == 4) {
        *reinterpret_cast<vf4*>(dst) = vf4{q[s * D], q[s * D + 1], q[s * D + 2], q[s * D + 3]};
      } else {
#pragma unroll
        for (int k = 0; k < D; ++k) dst[k] = q[s * D + k];
      }
    }
  }
  wave_lds_fence();
  float acc[NB];
#pragma unroll
  for (int s = 0; s < NB; ++s) acc[s] = INFINITY;
  uint32_t jmask[2] = {0u, 0u};
  QuadLdsPairs<CW, R, D, NB, 1>::run(q, acc, jmask, cmine, j, rows_end);
  jmask[0] = q_or(jmask[0]);
  jmask[1] = q_or(jmask[1]);
#pragma unroll
  for (int s = 0; s < NB; ++s) {
    const int i = kQuad * s + j;
    const bool by_lower = ((i < 32 ? jmask[0] : jmask[1]) >> (i & 31)) & 1u;
    const bool removed = acc[s] <= 0.0f || by_lower;  // (a hole stays a hole either way)
#pragma unroll
    for (int k = 0; k < D; ++k) q[s * D + k] = removed ? INFINITY : q[s * D + k];
  }
  wave_lds_fence();  // the compact image is written again after the stages
}

// The same for many slots per lane in TWO LEVELS (round 3).  Domination is transitive and the result of the test is the set
// of minimal rows (of equal rows the one with the lowest rank), so a row may be dropped as soon as ANY row dominates it --
// whatever happens to that row later: its own dominator dominates both.  Level 1: every lane tests the triangle of its
// OWN slots (no communication: NB (NB - 1) / 2 tests); a quarter of a game's rows, drawn across the game, already
// removes most dominated rows (50 random rows in dimension 4: ~25 of them survive level 1, ~15 the whole test).  The
// survivors are parked at their rank among the survivors (popcounts of the four lanes' masks: rank order = row order,
// which decides between equal rows) with their slot's index next to them, re-dealt four ways, and level 2 is the
// segmented loop above over the SURVIVORS only (in a bucket of 4, 6, 8, 10 or NB slots per lane: the widest game of
// the wave decides); its verdicts return through a byte per row.  For 50 dense rows: 78 + ~75 pair tests per lane
// instead of 306.  `tsc`: 2 x 64 bytes of scratch per game (slot indices, verdicts).
template <int M, int CW, int R, int D, int NB>
__device__ __forceinline__ void qd_newton_two_level(float (&q)[R * D], float* cmine, uint8_t* tsc, int j, int) {
  static_assert(kQuad * NB <= 64 + kQuad, "a slot's index travels as a byte below 64");
  float acc[NB];
#pragma unroll
  for (int s = 0; s < NB; ++s) acc[s] = INFINITY;
#pragma unroll
  for (int a = 0; a + 1 < NB; ++a) {
#pragma unroll
    for (int b = a + 1; b < NB; ++b) {
      float t, u;
      qd_extrema<D>(&q[a * D], &q[b * D], t, u);
      acc[b] = hk_fmin(acc[b], t);
      acc[a] = hk_fmin(acc[a], (t > 0.0f) ? -u : 1.0f);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  uint32_t lm = 0;  // my slots that are live and survived level 1
#pragma unroll
  for (int s = 0; s < NB; ++s) lm |= (q[s * D] < INFINITY && acc[s] > 0.0f) ? (1u << s) : 0u;
  const int np1 = q_sum(__popc(lm));
  int s1 = NB;  // slots per lane at level 2: the wave-uniform maximum of ceil(np1 / 4)
#pragma nounroll
  while (s1 > 1 && !__any(np1 > kQuad * (s1 - 1))) --s1;
  const uint32_t l1 = (uint32_t)qperm_i<kQuadUp1>((int)lm), l2 = (uint32_t)qperm_i<kQuadUp2>((int)lm),
                 l3 = (uint32_t)qperm_i<kQuadUp3>((int)lm);
  const bool b1 = ((j + 1) & 3) < j, b2 = ((j + 2) & 3) < j, b3 = ((j + 3) & 3) < j;
  uint8_t* tags = tsc;
  uint8_t* verdict = tsc + 64;
#pragma unroll
  for (int s = 0; s < NB; ++s) {
    const uint32_t below = (1u << s) - 1u, at = 1u << s;
    const int rank = __popc(lm & below) + __popc(l1 & (below | (b1 ? at : 0u))) + __popc(l2 & (below | (b2 ? at : 0u))) +
                     __popc(l3 & (below | (b3 ? at : 0u)));
    if ((lm >> s) & 1u) {
      float* dst = cmine + rank * CW;
      if constexpr (D == 4) {
        *reinterpret_cast<vf4*>(dst) = vf4{q[s * D], q[s * D + 1], q[s * D + 2], q[s * D + 3]};
      } else {
#pragma unroll
        for (int k = 0; k < D; ++k) dst[k] = q[s * D + k];
      }
      tags[rank] = (uint8_t)(kQuad * s + j);
    }
  }
  wave_lds_fence();
  auto level2 = [&](auto nbc) __attribute__((always_inline)) {
    constexpr int NB2 = decltype(nbc)::value;
    float q2[NB2 * D];
    int tag2[NB2];
#pragma unroll
    for (int s = 0; s < NB2; ++s) {
      const bool has = kQuad * s + j < np1;
      const int r = has ? kQuad * s + j : 0;
      const float* src = cmine + r * CW;
      if constexpr (D == 4) {
        const vf4 v = *reinterpret_cast<const vf4*>(src);
        q2[s * D] = has ? v.x : INFINITY;
        q2[s * D + 1] = has ? v.y : INFINITY;
        q2[s * D + 2] = has ? v.z : INFINITY;
        q2[s * D + 3] = has ? v.w : INFINITY;
      } else {
#pragma unroll
        for (int k = 0; k < D; ++k) {
          const float v = src[k];
          q2[s * D + k] = has ? v : INFINITY;
        }
      }
      tag2[s] = has ? (int)tags[r] : -1;
    }
    // (the rolled loop over the parked survivors -- they lie at their ranks already --, not the unrolled DPP test:
    // inside this body that one spilled 1.4 KB per lane at (50,4))
    qd_newton_lds<M, CW, NB2, D, NB2>(q2, cmine, j, np1);
#pragma unroll
    for (int s = 0; s < NB2; ++s)
      if (tag2[s] >= 0) verdict[tag2[s]] = (q2[s * D] < INFINITY) ? (uint8_t)0 : (uint8_t)1;
  };
  if (s1 <= 4) level2(std::integral_constant<int, (NB < 4 ? NB : 4)>{});
  else if (s1 <= 6) level2(std::integral_constant<int, (NB < 6 ? NB : 6)>{});
  else if (s1 <= 8) level2(std::integral_constant<int, (NB < 8 ? NB : 8)>{});
  else if (s1 <= 10) level2(std::integral_constant<int, (NB < 10 ? NB : 10)>{});
  else level2(std::integral_constant<int, NB>{});
  wave_lds_fence();
  // my slots back from where they are parked (the registers that held them were free during level 2: with them live
  // across it the (50,4) kernel spilled 1.2 KB per lane)
#pragma unroll
  for (int s = 0; s < NB; ++s) {
    const uint32_t below = (1u << s) - 1u, at = 1u << s;
    const int rank = __popc(lm & below) + __popc(l1 & (below | (b1 ? at : 0u))) + __popc(l2 & (below | (b2 ? at : 0u))) +
                     __popc(l3 & (below | (b3 ? at : 0u)));
    const bool kept = ((lm >> s) & 1u) && verdict[(kQuad * s + j) & 63] == 0;
    int off = (kept ? rank : 0) * CW;
    // (opaque: otherwise the compiler forwards the values stored at this very address before level 2 and keeps the
    // whole row array live across it)
    asm volatile("" : "+v"(off));
    const float* src = cmine + off;
    if constexpr (D == 4) {
      const vf4 v = *reinterpret_cast<const vf4*>(src);
      q[s * D] = kept ? v.x : INFINITY;
      q[s * D + 1] = kept ? v.y : INFINITY;
      q[s * D + 2] = kept ? v.z : INFINITY;
      q[s * D + 3] = kept ? v.w : INFINITY;
    } else {
#pragma unroll
      for (int k = 0; k < D; ++k) {
        const float v = src[k];
        q[s * D + k] = kept ? v : INFINITY;
      }
    }
  }
  wave_lds_fence();  // the compact image is written again after the stages
}

// observation features (jax/util.py:186-197, last coordinate primary; `coord0`: core/tensor_points.py:72-74, coordinate
// 0 alone): position of each of the lane's rows in descending key order, rows with EQUAL keys in row order (their
// compact rank: 4 s + lane).  Every pair of the game once -- own slots' triangle, ALL pairs with the lane one up, the
// pairs a <= b with the lane two up, as in qd_newton -- and what a lane counts for a partner's rows travels back.
template <int D>
__device__ __forceinline__ void qd_key_cmp(const float* a, const float* b, bool coord0, bool& gt, bool& eq) {
  gt = false;
  eq = true;
#pragma unroll
  for (int kk = 0; kk < D; ++kk) {
    const int k = D - 1 - kk;
    gt |= eq && (a[k] > b[k]);
    eq &= (a[k] == b[k]);
  }
  if (coord0) {
    gt = a[0] > b[0];
    eq = a[0] == b[0];
  }
}

// ---- key comparisons as borrow chains.  On the exact path every coordinate of a live row lies in [+0, +inf) by its bit
// pattern (the guard), so float order is the order of the patterns as unsigned integers and a key of W coordinates is
// ONE W-word number: `other > mine` is the borrow out of mine - other (v_sub_co / v_subb_co, W instructions, the borrow
// travelling in a scalar register pair), and `other > mine, or equal and tie` the same chain with `tie` as its borrow in.
// The borrow is added to a counter as it is (v_addc_co).  Before: W greater-than + W equal compares whose results met
// in scalar and/or chains -- ~14 vector + ~10 scalar instructions per pair, now W + 2 and none.
// Liveness rides in the key: the most significant word is pattern + 1 for a live row and 0 for a hole, so a hole is
// below every live row and never counted as coming first.  (kb_subb / kb_count / key_other_first: hk_fast_rows.h)
// the key words of slot s (least significant first).  KEY: kKeyLast (observation features, jax/util.py:186-197: last
// coordinate primary), kKeyFirst (list semantics, coordinate 0 primary), kKeyCoord0 (core/tensor_points.py:72-74:
// coordinate 0 alone)
template <int D, int KEY>
constexpr int key_words() { return KEY == kKeyCoord0 ? 1 : D; }
template <int D, int KEY>
__device__ __forceinline__ void key_of(const float* row, uint32_t (&w)[key_words<D, KEY>()]) {
  constexpr int W = key_words<D, KEY>();
  const bool live = row[0] < INFINITY;
  const int top = (KEY == kKeyLast) ? D - 1 : 0;
#pragma unroll
  for (int i = 0; i + 1 < W; ++i) w[i] = __float_as_uint(row[KEY == kKeyLast ? i : D - 1 - i]);
  w[W - 1] = live ? __float_as_uint(row[top]) + 1u : 0u;
}

// Every pair of the game once -- own slots' triangle, ALL pairs with the lane one up, the pairs a <= b with the lane two
// up, as in qd_newton -- and what a lane counts for a partner's rows travels back.  A partner's slot b receives `the
// number of my LIVE rows that come before it`: my holes lose against every live row, so that is (pairs counted) -
// (pairs the partner's row won) without a look at my liveness.
template <int R, int D, int NB, int KEY>
__device__ __forceinline__ void qd_ranks_stable_t(const float (&q)[R * D], int j, int (&rank)[R]) {
  constexpr int W = key_words<D, KEY>();
  uint32_t kw[NB][W];
  uint32_t rk[NB], won1[NB], won2[NB], diag2[NB];  // won: pairs the partner's slot b won against my rows
#pragma unroll
  for (int s = 0; s < NB; ++s) {
    key_of<D, KEY>(&q[s * D], kw[s]);
    rk[s] = won1[s] = won2[s] = diag2[s] = 0u;
  }
  const LaneMask none = 0ull;
  const LaneMask all = ~0ull, late1 = 0x8888888888888888ull, early2 = 0x3333333333333333ull;  // lane 3 of a quad; lanes 0, 1
  uint32_t ownlost[NB];
#pragma unroll
  for (int s = 0; s < NB; ++s) ownlost[s] = 0u;
  auto mine_at = [&](auto ac) -> const uint32_t(&)[W] { return kw[decltype(ac)::value]; };
  auto rank_at = [&](auto ac) -> uint32_t& { return rk[decltype(ac)::value]; };
  auto no_tie = [&](auto) { return none; };
  // (the chains run in blocks of three rows of mine against one other row: kb_rank_rows)
  unrolled_while<0, NB>([&](auto bc) {
    constexpr int b = decltype(bc)::value;
    // own slots a < b: a is the earlier row, b first iff its key is strictly greater
    kb_rank_rows<W, b, true>(kw[b], mine_at, no_tie, rank_at, ownlost[b]);
    uint32_t p1[W], p2[W];
#pragma unroll
    for (int i = 0; i < W; ++i) {
      p1[i] = (uint32_t)qperm_i<kQuadUp1>((int)kw[b][i]);
      p2[i] = (uint32_t)qperm_i<kQuadUp2>((int)kw[b][i]);
    }
    // the lane one up: all pairs; on equal keys its row is the earlier one iff b < a, or b == a on lane 3
    kb_rank_rows<W, NB, true>(p1, mine_at, [&](auto ac) {
      constexpr int a = decltype(ac)::value;
      return b < a ? all : (b == a ? late1 : none);
    }, rank_at, won1[b]);
    // the lane two up: pairs a < b (it does the mirror image); equal keys: my row is the earlier one
    kb_rank_rows<W, b, true>(p2, mine_at, no_tie, rank_at, won2[b]);
    {  // ... and the diagonal, counted ONCE, by the lower lane of the two
      const LaneMask c = key_other_first<W>(kw[b], p2, none);
      LaneMask live_b;
      asm("v_cmp_ne_u32_e64 %0, 0, %1" : "=s"(live_b) : "v"(kw[b][W - 1]));
      kb_count(rk[b], c & early2);
      kb_count(diag2[b], ~c & early2 & live_b);
    }
    return true;
  });
#pragma unroll
  for (int b = 1; b < NB; ++b) rk[b] += (uint32_t)b - ownlost[b];  // (my holes a < b lose against a live b)
#pragma unroll
  for (int s = 0; s < R; ++s) rank[s] = 0;
#pragma unroll
  for (int s = 0; s < NB; ++s) {  // what the lane one DOWN and the lane two up counted for my slot s
    const uint32_t o1 = (uint32_t)NB - won1[s], o2 = (uint32_t)s - won2[s] + diag2[s];
    rank[s] = (int)(rk[s] + (uint32_t)qperm_i<kQuadUp3>((int)o1) + (uint32_t)qperm_i<kQuadUp2>((int)o2));
  }
}
// list semantics (_list_ops.py:25-41): descending lexicographic order, coordinate 0 first.  Rows are distinct after the
// Newton stage, so the tie rules above never apply and the same pair-once walk serves.
template <int R, int D, int NB>
__device__ __forceinline__ void qd_ranks_first(const float (&q)[R * D], int (&rank)[R]) {
  qd_ranks_stable_t<R, D, NB, kKeyFirst>(q, 0, rank);
}
template <int R, int D, int NB>
__device__ __forceinline__ void qd_ranks_stable(const float (&q)[R * D], int j, bool coord0, int (&rank)[R]) {
  if (coord0) qd_ranks_stable_t<R, D, NB, kKeyCoord0>(q, j, rank);
  else qd_ranks_stable_t<R, D, NB, kKeyLast>(q, j, rank);
}

// The same ranks for MANY slots per lane (more than kQuadDppSlots: states no Newton pass has thinned, (50,4) only): the
// quad parks the KEYS of its rows at their compact ranks (4 s + lane) and every lane walks ALL the game's rows -- one
// broadcast read per row -- against its own slots: a row comes first iff its key is greater, or (TIES) equal with the
// lower compact rank -- that condition, a compare of the loop counter, is the chain's borrow in.
template <int M, int CW, int R, int D, int NB, int KEY, bool TIES>
__device__ __forceinline__ void qd_ranks_lds_t(const float (&q)[R * D], float* cmine, int j, int rows_end, int (&rank)[R]) {
  constexpr int W = key_words<D, KEY>();
  static_assert(CW >= W, "a parked key fits a parked row");
  uint32_t kw[NB][W], rk[NB];
#pragma unroll
  for (int s = 0; s < NB; ++s) {
    key_of<D, KEY>(&q[s * D], kw[s]);
    rk[s] = 0u;
    if (kQuad * s + j < M) {
      uint32_t* dst = reinterpret_cast<uint32_t*>(cmine) + (kQuad * s + j) * CW;
      if constexpr (W == 4) {
        *reinterpret_cast<uint4*>(dst) = uint4{kw[s][0], kw[s][1], kw[s][2], kw[s][3]};
      } else {
#pragma unroll
        for (int i = 0; i < W; ++i) dst[i] = kw[s][i];
      }
    }
  }
  wave_lds_fence();
#pragma nounroll
  for (int row = 0; row < rows_end; ++row) {
    uint32_t pj[W];
    const uint32_t* src = reinterpret_cast<const uint32_t*>(cmine) + row * CW;
    if constexpr (W == 4) {
      const uint4 v = *reinterpret_cast<const uint4*>(src);
      pj[0] = v.x; pj[1] = v.y; pj[2] = v.z; pj[3] = v.w;
    } else {
#pragma unroll
      for (int i = 0; i < W; ++i) pj[i] = src[i];
    }
    const int dj = row - j;  // (row < 4 s + j  <=>  dj < 4 s)
    uint32_t unused = 0u;
    kb_rank_rows<W, NB, false>(pj, [&](auto sc) -> const uint32_t(&)[W] { return kw[decltype(sc)::value]; },
                               [&](auto sc) { return TIES ? __ballot(dj < kQuad * decltype(sc)::value) : 0ull; },
                               [&](auto sc) -> uint32_t& { return rk[decltype(sc)::value]; }, unused);
  }
#pragma unroll
  for (int s = 0; s < R; ++s) rank[s] = 0;
#pragma unroll
  for (int s = 0; s < NB; ++s) rank[s] = (int)rk[s];
  wave_lds_fence();  // (the compact image is written again afterwards)
}
template <int M, int CW, int R, int D, int NB>
__device__ __forceinline__ void qd_ranks_lds(const float (&q)[R * D], float* cmine, int j, int rows_end, bool coord0,
                                             int (&rank)[R]) {
  if (coord0) qd_ranks_lds_t<M, CW, R, D, NB, kKeyCoord0, true>(q, cmine, j, rows_end, rank);
  else qd_ranks_lds_t<M, CW, R, D, NB, kKeyLast, true>(q, cmine, j, rows_end, rank);
}

// ---- the domination test of a FRESH game on packed rows (many slots per lane: (50,4)) ----------------------------------
// A fresh draw below 127 fits seven bits, so a row of up to four coordinates is ONE dword with a guard bit above every
// field, and "a <= b in every coordinate" is one subtraction: no field of (b | H) - a borrows iff every b_k >= a_k, i.e.
// iff all guard bits H = 0x80808080 survive -- v_sub, v_and, v_cmp where the float test takes ten instructions on four
// registers per row.  A hole is 0x7F7F7F7F: above every row, dominated by all of them, dominating none.
//   level 0: every lane's row of least coordinate sum (v_sad_u8) is a "champion"; the quad's four champions (DPP) are
//            tested against all slots -- strict domination only, so a champion never removes itself or its twins.
//            Domination is transitive and the result is the set of minimal rows (of equal rows the one of lowest rank),
//            so a row may go as soon as ANY row dominates it.  Of 50 uniform rows in dimension 4 about 20 survive (30 in
//            the widest of a wave's 16 games; the own-slot triangles of qd_newton_two_level leave 29 / 36);
//   level 1: the survivors are parked at their rank among the survivors (rank order = row order: it decides between equal
//            rows), re-dealt four ways, and every lane tests its slots against ALL of them (one broadcast read per
//            row): removed iff dominated by a different row, or by an equal one of lower rank.  The verdicts return
//            through a byte per survivor.
// `sc`: the game's LDS scratch, 16-B aligned, M + 4 dwords + M bytes.  Equal to _jax_ops.py:15-73 on integer rows; `reposition` commutes with
// it there (a column's minimum over the survivors is its minimum over all rows: a row holding it can only be dominated by
// a row that holds it too), so the caller subtracts the minima afterwards, on the survivors' floats.
constexpr uint32_t kPackGuard = 0x80808080u, kPackHole = 0x7F7F7F7Fu;
constexpr int kPackMaxValue = 127;  // draws are < max_value: at most 126

// rows [4 (C - 1), 4 C) of the parked survivors against the lane's slots, for C = 1 .. NB2: against such a row, slot
// s < C - 1 is the EARLIER row (it goes only if the row is below it and different), slot s > C - 1 the later one (an
// equal row removes it too), slot C - 1 is decided per row.  The verdicts accumulate as lane masks in scalar registers
// (ballots: no vector instruction; as a bool per slot the compiler packed them into bytes of vector registers, ~30
// instructions per row).
typedef uint32_t vu4 __attribute__((ext_vector_type(4)));

template <int NB2, int C>
struct QgPackedSeg {
  // `cur`: the segment's four rows (one 16-B read); the next segment's are requested before this one's tests.  Slot by
  // slot: the four rows' verdicts on a slot meet in scalar lane masks and are folded into the slot's counter at once
  // (one vector instruction per slot and segment; a lane mask per slot kept over the whole level overflows the scalar
  // registers: 2 000 spills in the build that tried).
  // The test itself: row <= slot in every field  <=>  sum |slot_k - row_k| + sum row_k == sum slot_k (v_sad_u8 with the
  // row's field sum as its addend, one compare: two instructions where the guard-bit subtraction takes three; the
  // row's sum is one v_sad_u8 per row, the slot's one per slot).
  static __device__ __forceinline__ void run(const uint32_t (&w2)[NB2], const uint32_t (&g2)[NB2], uint32_t (&cnt)[NB2],
                                             const uint32_t* sv, int np1, int rows, int j, vu4 cur) {
    constexpr int r0 = kQuad * (C - 1);
    if (rows <= r0) return;  // (wave-uniform)
    vu4 nxt = cur;
    if constexpr (C < NB2) nxt = *reinterpret_cast<const vu4*>(sv + kQuad * C);
    uint32_t c[kQuad], cs[kQuad];
#pragma unroll
    for (int rr = 0; rr < kQuad; ++rr) {
      c[rr] = (r0 + rr < np1) ? cur[rr] : kPackHole;  // (past np1: holes, dominating nothing)
      cs[rr] = __builtin_amdgcn_sad_u8(c[rr], 0u, 0u);
    }
#pragma unroll
    for (int s = 0; s < NB2; ++s) {
      bool hit = false;
#pragma unroll
      for (int rr = 0; rr < kQuad; ++rr) {
        const bool dom = __builtin_amdgcn_sad_u8(w2[s], c[rr], cs[rr]) == g2[s];  // row <= my slot s, coordinate by coordinate
        if (s > C - 1) hit |= dom;                                      // my slot is the later row: equal rows count
        else if (s < C - 1) hit |= dom & (w2[s] != c[rr]);              // ... the earlier row: different rows only
        else hit |= dom & ((w2[s] != c[rr]) | (rr < j));                // ... decided by the ranks r0 + rr and r0 + j
      }
      cnt[s] += hit ? 1u : 0u;
    }
    if constexpr (C < NB2) QgPackedSeg<NB2, C + 1>::run(w2, g2, cnt, sv, np1, rows, j, nxt);
  }
};

template <int NB2>
__device__ __forceinline__ void qg_packed_level1(const uint32_t* sv, uint8_t* vd, int np1, int rows, int j, int lane) {
  uint32_t w2[NB2], g2[NB2], cnt[NB2];
  const vu4 first = *reinterpret_cast<const vu4*>(sv);
#pragma unroll
  for (int s = 0; s < NB2; ++s) {
    const int r = kQuad * s + j;
    const uint32_t v = sv[r < rows ? r : 0];
    w2[s] = (r < np1) ? v : kPackHole;
    g2[s] = __builtin_amdgcn_sad_u8(w2[s], 0u, 0u);  // (the slot's field sum)
    cnt[s] = 0u;
  }
  QgPackedSeg<NB2, 1>::run(w2, g2, cnt, sv, np1, rows, j, first);  // (rows: wave-uniform, >= every game's np1)
#pragma unroll
  for (int s = 0; s < NB2; ++s) {
    const int r = kQuad * s + j;
    if (r < np1) vd[r] = cnt[s] ? (uint8_t)1 : (uint8_t)0;
  }
}

// w[s]: the packed row of slot s < NB (a hole: kPackHole); q[] receives the survivors' floats (slots >= NB are not touched)
template <int M, int D, int R, int NB>
__device__ __forceinline__ void qg_newton_on_packed(float (&q)[R * D], uint32_t (&w)[NB], uint32_t* sc, int j, int lane) {
  static_assert(D <= 4, "a row travels as four 7-bit fields");
  // level 0: the quad's four champions against every slot
  {
    uint32_t bs = __builtin_amdgcn_sad_u8(w[0], 0u, 0u), bw = w[0];
#pragma unroll
    for (int s = 1; s < NB; ++s) {
      const uint32_t ss = __builtin_amdgcn_sad_u8(w[s], 0u, 0u);
      const bool lt = ss < bs;
      bs = lt ? ss : bs;
      bw = lt ? w[s] : bw;
    }
    const uint32_t c0 = bw, c1 = (uint32_t)qperm_i<kQuadUp1>((int)bw), c2 = (uint32_t)qperm_i<kQuadUp2>((int)bw),
                   c3 = (uint32_t)qperm_i<kQuadUp3>((int)bw);
    const uint32_t s0 = __builtin_amdgcn_sad_u8(c0, 0u, 0u), s1 = __builtin_amdgcn_sad_u8(c1, 0u, 0u),
                   s2 = __builtin_amdgcn_sad_u8(c2, 0u, 0u), s3 = __builtin_amdgcn_sad_u8(c3, 0u, 0u);
#pragma unroll
    for (int s = 0; s < NB; ++s) {
      const uint32_t g = __builtin_amdgcn_sad_u8(w[s], 0u, 0u);  // (champion <= slot in every field: see QgPackedSeg)
      const bool r0 = (__builtin_amdgcn_sad_u8(w[s], c0, s0) == g) & (w[s] != c0);
      const bool r1 = (__builtin_amdgcn_sad_u8(w[s], c1, s1) == g) & (w[s] != c1);
      const bool r2 = (__builtin_amdgcn_sad_u8(w[s], c2, s2) == g) & (w[s] != c2);
      const bool r3 = (__builtin_amdgcn_sad_u8(w[s], c3, s3) == g) & (w[s] != c3);
      w[s] = (r0 | r1 | r2 | r3) ? kPackHole : w[s];
    }
  }
  // the survivors to their ranks (popcounts of the four lanes' masks, hk_quadroll_kernel.h: qr_redeal)
  uint32_t lm = 0;
#pragma unroll
  for (int s = 0; s < NB; ++s) lm |= (w[s] != kPackHole) ? (1u << s) : 0u;
  const int np1 = q_sum(__popc(lm));
  int s1 = NB;  // slots per lane at level 1: the wave-uniform maximum of ceil(np1 / 4)
#pragma nounroll
  while (s1 > 1 && !__any(np1 > kQuad * (s1 - 1))) --s1;
  const uint32_t l1 = (uint32_t)qperm_i<kQuadUp1>((int)lm), l2 = (uint32_t)qperm_i<kQuadUp2>((int)lm),
                 l3 = (uint32_t)qperm_i<kQuadUp3>((int)lm);
  const bool b1 = ((j + 1) & 3) < j, b2 = ((j + 2) & 3) < j, b3 = ((j + 3) & 3) < j;
  uint8_t* vd = reinterpret_cast<uint8_t*>(sc + M + 4);  // (segment reads run up to 3 dwords past the M rows)
  int rk[NB];
#pragma unroll
  for (int s = 0; s < NB; ++s) {
    const uint32_t below = (1u << s) - 1u, at = 1u << s;
    rk[s] = __popc(lm & below) + __popc(l1 & (below | (b1 ? at : 0u))) + __popc(l2 & (below | (b2 ? at : 0u))) +
            __popc(l3 & (below | (b3 ? at : 0u)));
    if ((lm >> s) & 1u) sc[rk[s]] = w[s];
  }
  wave_lds_fence();
  const int rows = (kQuad * s1 < M) ? kQuad * s1 : M;
  // (buckets of slots per lane; 50 uniform rows in dimension 4: the widest of a wave's 16 games keeps 26 - 36 rows after
  // level 0, 7 - 9 slots per lane)
  if (s1 <= 2) qg_packed_level1<(NB < 2 ? NB : 2)>(sc, vd, np1, rows, j, lane);
  else if (s1 <= 4) qg_packed_level1<(NB < 4 ? NB : 4)>(sc, vd, np1, rows, j, lane);
  else if (s1 <= 6) qg_packed_level1<(NB < 6 ? NB : 6)>(sc, vd, np1, rows, j, lane);
  else if (s1 <= 7) qg_packed_level1<(NB < 7 ? NB : 7)>(sc, vd, np1, rows, j, lane);
  else if (s1 <= 8) qg_packed_level1<(NB < 8 ? NB : 8)>(sc, vd, np1, rows, j, lane);
  else if (s1 <= 9) qg_packed_level1<(NB < 9 ? NB : 9)>(sc, vd, np1, rows, j, lane);
  else if (s1 <= 10) qg_packed_level1<(NB < 10 ? NB : 10)>(sc, vd, np1, rows, j, lane);
  else qg_packed_level1<NB>(sc, vd, np1, rows, j, lane);
  wave_lds_fence();
  // the survivors' floats from their packed words (v_cvt_f32_ubyteN: exact): q[] is written here and nowhere read since
  // the packing -- its R * D registers are free during the whole test (inside a rollout kernel's 168-register budget
  // they were what spilled)
#pragma unroll
  for (int s = 0; s < NB; ++s) {
    const bool kept = ((lm >> s) & 1u) && vd[((lm >> s) & 1u) ? rk[s] : 0] == 0;
#pragma unroll
    for (int k = 0; k < D; ++k) q[s * D + k] = kept ? (float)((w[s] >> (8 * k)) & 0xFFu) : INFINITY;
  }
  wave_lds_fence();  // (the scratch is the game's image again)
}

// a FRESH game (the generator: draws below 127, every value integral): pack, test
template <int M, int D, int R>
__device__ __forceinline__ void qg_newton_packed(float (&q)[R * D], uint32_t* sc, int j, int lane) {
  uint32_t w[R];
#pragma unroll
  for (int s = 0; s < R; ++s) {
    uint32_t v = (uint32_t)q[s * D];
#pragma unroll
    for (int k = 1; k < D; ++k) v |= (uint32_t)q[s * D + k] << (8 * k);
    w[s] = (q[s * D] < INFINITY) ? v : kPackHole;
  }
  qg_newton_on_packed<M, D, R, R>(q, w, sc, j, lane);
}

// ANY state (hk_step on many slots per lane): w[s] = the packed row of slot s, true (wave-uniform) iff every live coordinate
// of the wave is integral and at most 126 -- fract() of all of them is zero (a hole's is NaN or 0: the maximum skips
// it), no saturating conversion reaches a guard bit.
template <int R, int D, int NB>
__device__ __forceinline__ bool qd_pack_rows(const float (&q)[R * D], uint32_t (&w)[NB]) {
  static_assert(D <= 4, "four fields");
  uint32_t bad = 0u;
  float fr = 0.0f;
#pragma unroll
  for (int s = 0; s < NB; ++s) {
    uint32_t raw = 0u;
#pragma unroll
    for (int k = 0; k < D; ++k) {
      raw = __builtin_amdgcn_cvt_pk_u8_f32(q[s * D + k], k, raw);
      fr = __builtin_fmaxf(fr, __builtin_amdgcn_fractf(q[s * D + k]));
    }
    const bool live = q[s * D] < INFINITY;
    bad |= live ? (raw | (raw + 0x01010101u)) : 0u;
    w[s] = live ? raw : kPackHole;
  }
  return !__any(((bad & kPackGuard) != 0u) | (fr > 0.0f));
}

// one transition on slots [0, NB) of the four lanes; returns the GAME's number of live rows.  `sorted` (list
// semantics): the survivors are ranked right after the Newton stage, before a rescale could round two keys together.
// AUX: the kernel may be asked for the sorted observation features of many slots per lane (the run-time configured
// kernels; the compiled configurations carry none of that code -- with it, even dead, the (50,4) step kernels went from
// 236 registers to 256 and 372 B of scratch)
template <int M, int CW, int R, int D, int NB, bool AUX = true>
__device__ __forceinline__ int qd_stages(float (&q)[R * D], const float (&c)[D], int axis, int np, int j,
                                         unsigned flags, unsigned stages, float* cmine, int slots_end, bool sorted,
                                         int (&rank)[R], uint8_t* tsc = nullptr) {
  if (stages & HK_STAGE_SHIFT) b_shift<R, D, NB>(q, c, axis, np, flags);
  if (stages & HK_STAGE_REPOSITION) qd_reposition<R, D, NB>(q, flags);
  if (stages & HK_STAGE_NEWTON) {
    if constexpr (NB > kQuadDppSlots) {
      // many slots per lane (states no Newton pass has thinned): integral rows of small coordinates -- what a generator's
      // raw draws are after a shift and the reposition -- take the test on packed rows (the game's parked-row region is
      // its scratch), anything else the float tests
      bool packed = false;
      if constexpr (D <= 4) {
        uint32_t w[NB];
        packed = qd_pack_rows<R, D, NB>(q, w);
        if (packed) qg_newton_on_packed<M, D, R, NB>(q, w, reinterpret_cast<uint32_t*>(cmine), j, 0);
      }
      if (!packed) {
        if (tsc) qd_newton_two_level<M, CW, R, D, NB>(q, cmine, tsc, j, slots_end);
        else qd_newton_lds<M, CW, R, D, NB>(q, cmine, j, slots_end);
      }
    } else {
      qd_newton<R, D, NB>(q, j);
    }
    if constexpr (NB <= kQuadDppSlots)
      if (sorted) qd_ranks_first<R, D, NB>(q, rank);
  }
  if (stages & HK_STAGE_RESCALE) qd_rescale<R, D, NB>(q, flags);
  if constexpr (NB <= kQuadDppSlots) {  // the observation features are sorted AFTER their rescale
    if (stages & kStageFeatureSorts) qd_ranks_stable<R, D, NB>(q, j, (stages & kStageFeatureSort0) != 0, rank);
  } else if constexpr (AUX) {
    if (stages & kStageFeatureSorts)
      qd_ranks_lds<M, CW, R, D, NB>(q, cmine, j, slots_end, (stages & kStageFeatureSort0) != 0, rank);
  }
  int n = 0;
#pragma unroll
  for (int r = 0; r < NB; ++r) n += (q[r * D] < INFINITY) ? 1 : 0;
  return q_sum(n);
}

template <int M, int D, int NB, bool AUX = true>
struct QuadStagesFor {
  using G = QuadGeom<M, D>;
  static __device__ __forceinline__ int run(float (&q)[G::R * D], int smax, const float (&c)[D], int axis, int np,
                                            int j, unsigned flags, unsigned stages, float* cmine, bool sorted,
                                            int (&rank)[G::R], uint8_t* tsc = nullptr) {
    const int slots_end = kQuad * smax < M ? kQuad * smax : M;
    if constexpr (NB >= G::R) {
      return qd_stages<M, G::CW, G::R, D, G::R, AUX>(q, c, axis, np, j, flags, stages, cmine, slots_end, sorted, rank, tsc);
    } else {
      if (smax <= NB)
        return qd_stages<M, G::CW, G::R, D, NB, AUX>(q, c, axis, np, j, flags, stages, cmine, slots_end, sorted, rank, tsc);
      return QuadStagesFor<M, D, G::next_bucket(NB), AUX>::run(q, smax, c, axis, np, j, flags, stages, cmine, sorted, rank,
                                                               tsc);
    }
  }
};

// ---- the caller's actions ----------------------------------------------------------------------------------------
// ACT: the layout of (coords, axis) as a compile-time fact, for the layouts the trainers use.  The run-time version
// (kActAny: fetch_raw / scalar_from_raw, any dtype through double) costs ~0.9 us of a 7.5 us launch at 65 536 games:
// 64-bit address arithmetic per value and a dtype switch per value in front of the slab's commit.
enum QuadAct : int {
  kActAny = 0,
  kActMaskF32AxisI32 = 1,  // [B, d] float32 multi-binary mask (contiguous) + int32 axis: bench.py, get_take_actions
  kActMaskF32AxisI64 = 2,  // ... + int64 axis: torch.argmax of the opponent's one-hot (recurrent_fn.py)
  kActMaskF32AxisF32 = 3,  // ... + float32 axis: the JAX trainer's arrays (jax_trainer.py:528)
  kActClassI32AxisI32 = 4,  // [B] int32 class ids + int32 axis: the gym / list surface
  kActClassI32Logits = 5    // [B] int32 class ids + [B, d] float32 agent logits (HK_AXIS_MASKED_LOGITS): the search's expansion
};

inline int quad_act_of(const Params& prm) {
  if (!(prm.stages & HK_STAGE_SHIFT)) return kActAny;
  if (prm.coords_kind == HK_F32 && prm.coords_stride == prm.d) {
    if (prm.axis_dtype == HK_I32) return kActMaskF32AxisI32;
    if (prm.axis_dtype == HK_I64) return kActMaskF32AxisI64;
    if (prm.axis_dtype == HK_F32) return kActMaskF32AxisF32;
  }
  if (prm.coords_kind == HK_COORDS_CLASS_I32 && prm.axis_dtype == HK_I32) return kActClassI32AxisI32;
  if (prm.coords_kind == HK_COORDS_CLASS_I32 && prm.axis_dtype == HK_AXIS_MASKED_LOGITS) return kActClassI32Logits;
  return kActAny;
}

// One load per lane and value: lane j of the quad fetches coordinate min(j, D-1) of its game's mask (the four lanes of
// a quad read consecutive dwords: the wave's request is dense), every lane the game's axis; the quad then hands the
// coordinates round with DPP broadcasts.  Issued right behind the slab's requests, consumed after its commit.
template <int D, int ACT>
struct QuadActions {
  uint32_t cword, aword, aword_hi;
  RawActions<D> raw;  // kActAny only
};

// (g0: the wave's first game, wave-uniform; gi: the lane's game inside the slab -- scalar base + 32-bit lane offset)
template <int D, int ACT>
__device__ __forceinline__ void quad_actions_issue(QuadActions<D, ACT>& a, const Params& prm, int64_t g0, unsigned gi,
                                                   int m, int j) {
  static_assert(D <= kQuad || ACT == kActAny || ACT == kActClassI32AxisI32, "mask broadcast needs dim <= 4");
  if constexpr (ACT == kActAny) {
    fast_fetch_actions<D>(prm, g0 + gi, m, a.raw);
  } else {
    if constexpr (ACT == kActClassI32AxisI32 || ACT == kActClassI32Logits) a.cword = ((const uint32_t*)prm.coords + g0)[gi];
    else a.cword = ((const uint32_t*)prm.coords + g0 * D)[gi * D + (unsigned)(j < D ? j : D - 1)];
    if constexpr (ACT == kActClassI32Logits) {  // lane k of the quad fetches logit k
      a.aword = ((const uint32_t*)prm.axis + g0 * D)[gi * D + (unsigned)(j < D ? j : D - 1)];
    } else if constexpr (ACT == kActMaskF32AxisI64) {
      a.aword = ((const uint32_t*)prm.axis + 2 * g0)[2 * gi];
      a.aword_hi = ((const uint32_t*)prm.axis + 2 * g0)[2 * gi + 1];
    } else {
      a.aword = ((const uint32_t*)prm.axis + g0)[gi];
    }
  }
}

// Both requests are in flight before anything waits: the words pass through one empty asm statement, so no use of
// either can be scheduled between the two loads.  (Without it the compiler issues the axis request, waits for it AND
// the slab -- vmcnt counts in order -- and only then asks for the mask word: a second HBM round trip per launch.)
template <int D, int ACT>
__device__ __forceinline__ void quad_actions_commit(QuadActions<D, ACT>& a) {
  if constexpr (ACT == kActMaskF32AxisI64) asm volatile("" : "+v"(a.cword), "+v"(a.aword), "+v"(a.aword_hi));
  else if constexpr (ACT != kActAny) asm volatile("" : "+v"(a.cword), "+v"(a.aword));
}

template <int D, int ACT>
__device__ __forceinline__ void quad_actions_decode(const QuadActions<D, ACT>& a, const Params& prm, float (&c)[D],
                                                    int& axis) {
  if constexpr (ACT == kActAny) {
    fast_decode_actions<D>(prm, a.raw, c, axis);
  } else {
    uint32_t subset = 0;
    if constexpr (ACT == kActClassI32AxisI32 || ACT == kActClassI32Logits) {
      constexpr int ncls = (1 << D) - D - 1;
      int cls = (int)a.cword;
      cls = cls < 0 ? 0 : (cls >= ncls ? ncls - 1 : cls);
      subset = decode_class(cls, D);
#pragma unroll
      for (int k = 0; k < D; ++k) c[k] = (float)((subset >> k) & 1u);
    } else {
      // quad_perm [k, k, k, k]: coordinate k sits in lane k of the quad
      c[0] = __int_as_float(qperm_i<0x00>((int)a.cword));
      if constexpr (D > 1) c[D > 1 ? 1 : 0] = __int_as_float(qperm_i<0x55>((int)a.cword));
      if constexpr (D > 2) c[D > 2 ? 2 : 0] = __int_as_float(qperm_i<0xAA>((int)a.cword));
      if constexpr (D > 3) c[D > 3 ? 3 : 0] = __int_as_float(qperm_i<0xFF>((int)a.cword));
    }
    if constexpr (ACT == kActClassI32Logits) {
      // the agent's move: argmax of its logits over the subset's coordinates (jax/util.py:287-327; first maximum, a
      // NaN beats every number) -- hk_search_masked_argmax in the step's own action decode
      float lg[D];
      lg[0] = __int_as_float(qperm_i<0x00>((int)a.aword));
      if constexpr (D > 1) lg[D > 1 ? 1 : 0] = __int_as_float(qperm_i<0x55>((int)a.aword));
      if constexpr (D > 2) lg[D > 2 ? 2 : 0] = __int_as_float(qperm_i<0xAA>((int)a.aword));
      if constexpr (D > 3) lg[D > 3 ? 3 : 0] = __int_as_float(qperm_i<0xFF>((int)a.aword));
      int best = 0;
      float bv = (subset & 1u) ? lg[0] : -INFINITY;
#pragma unroll
      for (int k = 1; k < D; ++k) {
        const float v = ((subset >> k) & 1u) ? lg[k] : -INFINITY;
        const bool better = (v > bv) || (v != v && bv == bv);
        best = better ? k : best;
        bv = better ? v : bv;
      }
      axis = best;
    } else if constexpr (ACT == kActMaskF32AxisF32) {
      const float f = __uint_as_float(a.aword);  // `arange(d) == axis`: non-integral / out-of-range match nothing
      const int i = (int)f;
      axis = (f >= 0.0f && f < (float)D && (float)i == f) ? i : -1;
    } else if constexpr (ACT == kActMaskF32AxisI64) {
      axis = (a.aword_hi == 0u && a.aword < (uint32_t)D) ? (int)a.aword : -1;
    } else {
      axis = (a.aword < (uint32_t)D) ? (int)a.aword : -1;
    }
  }
}

// ---- the kernel: single steps with the caller's actions (hk_step) -----------------------------------------------
// HOT: kHotJax = the JAX trainer's take_actions (shift + reposition + Newton polytope, JAX semantics) compiled in.
// register budget: four waves per SIMD (<= 128 VGPRs), three for the large games (<= 168; their LDS allows no more)
// FEAT: hk_step_features -- the observation features of the result (jax/util.py:172-214: [rescale] + rows in descending
// key order, equal keys in row order) as a second output of the same launch: the bucket body has the result's rows in
// registers, ranks them (qd_ranks_stable) and builds the sorted image in the compact region, which is free by then.
// kHotSort (large games): the launches whose product is a sorted image and nothing else (hk_get_features /
// hk_get_features_torch: [rescale] + the ranks) -- no step stages compiled in, so the kernel fits three waves per SIMD (it
// waits for its slab most of its life: PMC at two waves, 2 190 VALU instructions inside 8 400 wave cycles) and needs no
// scratch for the two-level test.
constexpr int kHotSort = 3;
template <int M, int D, int HOT>
constexpr int quad_waves_per_simd() {
  return (HOT == kHotSort && QuadGeom<M, D>::kBig) ? 3 : QuadGeom<M, D>::kWavesPerSimd;
}
template <int M, int D, int HOT, int WPB, int ACT, bool FEAT = false>
__global__ __launch_bounds__(kWave * WPB, (quad_waves_per_simd<M, D, HOT>())) void quad_kernel(const float* in0, int64_t in_stride0, int batch0,
                                                           const Params prm) {
  using G = QuadGeom<M, D>;
  constexpr int R = G::R;
  using MaskM = MaskT<M>;
  __shared__ __align__(16) float lds_all[WPB * G::kRegion];
  __shared__ float cbuf_all[WPB * kQuadGames * D];  // slow path only
  // large games: scratch of the two-level domination test (slot indices + verdicts, 2 x 64 bytes per game)
  __shared__ __align__(16) uint8_t tsc_all[(G::kBig && M <= 64 && HOT != kHotSort) ? WPB * kQuadGames * 128 : 16];
  // the wave index as a scalar: the slab's addresses, the LDS region and the game count stay in SGPRs
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & (kWave - 1);
  float* image = lds_all + wave * G::kRegion;
  float* compact = G::kBig ? image : image + G::kImage;
  const int j = lane & 3, gi = lane >> 2;
  const int64_t g0 = ((int64_t)blockIdx.x * WPB + wave) * kQuadGames;
  const int64_t left = (int64_t)batch0 - g0;
  if (left <= 0) return;
  const int ngames = (int)(left < kQuadGames ? left : kQuadGames);
  const bool active = gi < ngames;
  const bool leader = active && j == 0;
  const int64_t g = g0 + gi;
#ifdef HK_QUAD_PROBE  // (HK_QUAD_CUT=9: a time line per wave, written over num_points_out: scripts/probe_step_timeline.py)
  const long long tl0 = wall_clock64();
  long long tl1 = 0, tl2 = 0;
#endif
  quad_slab_load<M, D>(in0 + g0 * G::N, image, ngames, lane);
  float* mine = image + gi * G::N;
  const float pad = prm.pad_f32;
  const unsigned flags = (HOT == kHotJax) ? (unsigned)HK_SEM_JAX : prm.flags;
  const unsigned stages = (HOT == kHotSort) ? (prm.stages & ((unsigned)HK_STAGE_RESCALE | kStageFeatureSorts))
                          : HOT ? (unsigned)(HK_STAGE_SHIFT | HK_STAGE_REPOSITION | HK_STAGE_NEWTON) : prm.stages;
  const float fill = ((flags & HK_SEM_MASK) == HK_SEM_JAX) ? -1.0f : pad;
  float c[D];
  int axis_in = -1;
#pragma unroll
  for (int k = 0; k < D; ++k) c[k] = 0.0f;
  QuadActions<D, ACT> actions;
  const bool fetch_actions = (stages & HK_STAGE_SHIFT) && active;
  if (fetch_actions) quad_actions_issue<D, ACT>(actions, prm, g0, (unsigned)gi, M, j);
  wait_vmem_all();
  if (fetch_actions) {
    quad_actions_commit<D, ACT>(actions);
    quad_actions_decode<D, ACT>(actions, prm, c, axis_in);
  }
  wave_lds_fence();
#ifdef HK_QUAD_PROBE  // dev builds only (scripts/build_probe.sh): stop after a phase to see what each one costs
  tl1 = wall_clock64();
  const int cut = prm.lds_stride;
  if (cut == 1) {
    quad_slab_store<M, D>(image, (float*)prm.out + g0 * G::N, ngames, lane);
    return;
  }
#endif

  // ---- live rows + exactness guard: lane j looks at rows j*R .. j*R + R - 1 -----------------------------------------
  float rows[R * D];
  const int i0 = j * R;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if constexpr (D == 4) {  // a row is one aligned 16-B read
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
  bool lv[R];  // (the compaction selects on these: compare results stay in scalar registers, no bit tests on lmask)
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
    lv[r] = ge && i0 + r < M && active;
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
#ifdef HK_QUAD_PROBE
  if (cut == 2) {
    if (leader && prm.num_points_out) prm.num_points_out[g] = np + below + (exact ? 1 : 0);
    quad_slab_store<M, D>(image, (float*)prm.out + g0 * G::N, ngames, lane);
    return;
  }
#endif

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
      if constexpr (FEAT) {
        // hk_step_features: the observation features of the result by the generic routines too (jax/util.py:172-214:
        // [rescale] + rows in descending key order), built in the game's part of the compact region
        float* f = compact + gi * G::N;
        for (int e = 0; e < G::N; ++e) f[e] = mine[e];
        stages_game<float>(f, M, D, cs, -1, pad, (prm.feat_scale ? (unsigned)HK_STAGE_RESCALE : 0u) | kStageFeatureSort,
                           (unsigned)HK_SEM_JAX);
      }
    }
    wave_lds_fence();
    quad_slab_store<M, D>(image, (float*)prm.out + g0 * G::N, ngames, lane);
    if constexpr (FEAT) quad_slab_store<M, D>(compact, prm.feat_out + g0 * G::N, ngames, lane);
    return;
  }

  // ---- compaction.  Small games: EVERY row to a slot of the compact image, tagged with its original index -- a live
  // row to its RANK among the live rows, a dead one behind them (np + its rank among the dead): a permutation, so
  // every write is unconditional.  Large games (the compact image lies over the slab image, which every lane has
  // read by now): the live rows only, no tag.
  float* cmine = compact + gi * (G::kBig ? G::kGameStride : M * G::CW);
  {
    int rank = below;
    // small games: slot of row r = live ? below + (my live rows before r) : (np + i0 - below) + (my dead rows before
    // r) -- two popcounts with an addend, no chain through the rows
    const uint32_t dmask = ~lmask;
    const int dbase = np + i0 - below;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int i = i0 + r;
      const bool live = G::kBig ? (bool)((lmask >> r) & 1u) : lv[r];
      if constexpr (G::kBig) {
        if (live) {
          float* dst = cmine + rank * G::CW;
          if constexpr (D == 4) {
            *reinterpret_cast<vf4*>(dst) = vf4{rows[r * D], rows[r * D + 1], rows[r * D + 2], rows[r * D + 3]};
          } else {
#pragma unroll
            for (int k = 0; k < D; ++k) dst[k] = rows[r * D + k];
          }
        }
      } else {
        const uint32_t before = (1u << r) - 1u;
        const uint32_t msel = live ? lmask : dmask;  // (select the operands, not the results: one popcount per row)
        const int bsel = live ? below : dbase;
        const int slot = bsel + __popc(msel & before);
        if (M % kQuad == 0 || i < M) {  // (rows past M exist only in the last lane's tail when 4 does not divide M)
          float* dst = cmine + slot * G::CW;
          if constexpr (D <= 3) {
            vf4 v;
            v.x = rows[r * D];
            v.y = D > 1 ? rows[r * D + (D > 1 ? 1 : 0)] : 0.0f;
            v.z = D > 2 ? rows[r * D + (D > 2 ? 2 : 0)] : 0.0f;
            v.w = __int_as_float(i * D);  // the tag: where the row lives in the image (floats)
            *reinterpret_cast<vf4*>(dst) = v;
          } else {
#pragma unroll
            for (int k = 0; k < D; ++k) dst[k] = rows[r * D + k];
            dst[D] = __int_as_float(i * D);
          }
        }
      }
      if constexpr (G::kBig) rank += live ? 1 : 0;
    }
  }
  // slots per lane in use: the wave-uniform maximum of ceil(np / 4) (a downward search: a few ballots)
  int smax = R;
#pragma nounroll
  while (smax > 1 && !__any(np > kQuad * (smax - 1))) --smax;
  wave_lds_fence();

  const bool prev_done = np < 2;
  const int np_before = np;
  if constexpr (!G::kBig) {
    // ---- small games: ONE dispatch on the wave's bucket (slots per lane in use), and inside it, straight-line for
    // that many slots: my slots in (ranks j, j + 4, ...; a slot past the game's live rows is a hole), the transition,
    // the rows back.  (Round 2 first read all slots, then dispatched the stages, then walked the slots again: three
    // ladders, and the slots past the bucket materialised as +inf for nobody.) --------------------------------------
    // list semantics / COMPACT_SORTED (run-time configured kernels only): sorted + compacted output
    const bool list_sorted = HOT == kHotNone && (stages & HK_STAGE_NEWTON) &&
                             ((flags & HK_SEM_MASK) == HK_SEM_LIST || (flags & HK_FLAG_COMPACT_SORTED));
    // ... or the observation features: rows in descending key order (hk_get_features / hk_get_features_torch)
    const bool sorted = list_sorted || ((HOT == kHotNone || HOT == kHotSort) && (stages & kStageFeatureSorts));
#ifdef HK_QUAD_PROBE
    float probe_acc = 0.0f;
#endif
    unrolled_while<1, R + 1>([&](auto nbc) {
      constexpr int NB = decltype(nbc)::value;
      if (NB < R && smax > NB) return true;
      float q[R * D];
      int orig[R], rank[R];
      const int npj = np - j;
      const float* cj = cmine + j * G::CW;
#pragma unroll
      for (int s = 0; s < NB; ++s) {
        // (slot s of lane j is rank 4 s + j: one lane base, the slot in the read's immediate; np - j against a constant)
        const bool has = kQuad * s < npj;
        const float* src = (kQuad * s + kQuad <= M) ? cj + kQuad * s * G::CW
                                                    : cmine + (kQuad * s + j < M ? kQuad * s + j : 0) * G::CW;
        if constexpr (D <= 3) {
          const vf4 v = *reinterpret_cast<const vf4*>(src);
          q[s * D] = has ? v.x : INFINITY;
          if (D > 1) q[s * D + (D > 1 ? 1 : 0)] = has ? v.y : INFINITY;
          if (D > 2) q[s * D + (D > 2 ? 2 : 0)] = has ? v.z : INFINITY;
          orig[s] = __float_as_int(v.w);
        } else {
#pragma unroll
          for (int k = 0; k < D; ++k) {
            const float v = src[k];
            q[s * D + k] = has ? v : INFINITY;
          }
          orig[s] = __float_as_int(src[D]);
        }
      }
#ifdef HK_QUAD_PROBE
      if (cut == 3 || cut == 4) {
        if (cut == 4) qd_stages<M, G::CW, R, D, NB>(q, c, axis_in, np, j, flags, stages, cmine, M, list_sorted, rank);
#pragma unroll
        for (int e = 0; e < NB * D; ++e) probe_acc += (q[e] < INFINITY) ? q[e] : 0.0f;
        return false;
      }
#endif
      np = qd_stages<M, G::CW, R, D, NB>(q, c, axis_in, np, j, flags, stages, cmine, M, list_sorted, rank);
      if constexpr (FEAT) {
        static_assert(G::kCompact >= G::kImage, "the features image is built in the compact region");
        // (every lane has read its slots: the compact region is free)  padding everywhere, then the survivors --
        // rescaled if asked -- at their rank
        float fq[R * D];
        int frank[R];
#pragma unroll
        for (int e = 0; e < NB * D; ++e) fq[e] = q[e];
        if (prm.feat_scale) qd_rescale<R, D, NB>(fq, (unsigned)HK_SEM_JAX);
        qd_ranks_stable<R, D, NB>(fq, j, false, frank);
#pragma unroll
        for (int it = 0; it < G::QL; ++it) {
          const int qq = lane + it * kWave;
          if (qq < kQuadGames * G::Q) {
            if constexpr (G::W == 4) *reinterpret_cast<vf4*>(compact + qq * 4) = vf4{pad, pad, pad, pad};
            else if constexpr (G::W == 2) *reinterpret_cast<vf2*>(compact + qq * 2) = vf2{pad, pad};
            else compact[qq] = pad;
          }
        }
        wave_lds_fence();
#pragma unroll
        for (int s = 0; s < NB; ++s) {
          if (fq[s * D] < INFINITY) {
            float* dst = compact + gi * G::N + frank[s] * D;
#pragma unroll
            for (int k = 0; k < D; ++k) dst[k] = fq[s * D + k];
          }
        }
      }
      if (sorted) {
        // list semantics: padding everywhere (the quads fill the wave's image in 16-B pieces), then every survivor at
        // its rank
#pragma unroll
        for (int it = 0; it < G::QL; ++it) {
          const int qq = lane + it * kWave;
          if (qq < kQuadGames * G::Q) {
            if constexpr (G::W == 4) *reinterpret_cast<vf4*>(image + qq * 4) = vf4{pad, pad, pad, pad};
            else if constexpr (G::W == 2) *reinterpret_cast<vf2*>(image + qq * 2) = vf2{pad, pad};
            else image[qq] = pad;
          }
        }
        wave_lds_fence();
#pragma unroll
        for (int s = 0; s < NB; ++s) {
          if (q[s * D] < INFINITY) {
            float* dst = mine + rank[s] * D;
#pragma unroll
            for (int k = 0; k < D; ++k) dst[k] = q[s * D + k];
          }
        }
      } else {
        // in place: the row of every slot in use goes back to its own place in the image -- new coordinates if it
        // survived, padding if it was removed; the dead rows that sit in the slots past the live ones are padding
        // already and are rewritten as such (unconditional writes)
#pragma unroll
        for (int s = 0; s < NB; ++s) {
          if (M % kQuad == 0 || kQuad * s + j < M) {
            const bool removed = !(q[s * D] < INFINITY);
            float* dst = mine + orig[s];
#pragma unroll
            for (int k = 0; k < D; ++k) dst[k] = removed ? pad : q[s * D + k];
          }
        }
      }
      return false;
    });
#ifdef HK_QUAD_PROBE
    if (cut == 3 || cut == 4) {
      if (leader && prm.reward_out) prm.reward_out[g] = probe_acc;
      quad_slab_store<M, D>(image, (float*)prm.out + g0 * G::N, ngames, lane);
      return;
    }
#endif
  }

  // ---- large games: my slots (ranks j, j + 4, ... up to the wave's smax; slots past the game's live rows are holes)
  float q[G::kBig ? R * D : 1];
  int rank[G::kBig ? R : 1];  // (the sorted observation features: a row's place in descending key order)
  if constexpr (G::kBig) {
#pragma unroll
    for (int e = 0; e < R * D; ++e) q[e] = INFINITY;
    unrolled_while<0, R>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
      if (s >= smax) return false;
      const bool has = kQuad * s + j < np;
      const float* src = cmine + (kQuad * s + j < M ? kQuad * s + j : 0) * G::CW;
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
      return true;
    });
#ifdef HK_QUAD_PROBE
    if (cut == 3) {
      float acc = 0.0f;
#pragma unroll
      for (int e = 0; e < R * D; ++e) acc += (q[e] < INFINITY) ? q[e] : 0.0f;
      if (leader && prm.reward_out) prm.reward_out[g] = acc;
      quad_slab_store<M, D>(image, (float*)prm.out + g0 * G::N, ngames, lane);
      return;
    }
#endif
    uint8_t* tsc = (G::kBig && M <= 64 && HOT != kHotSort) ? tsc_all + (wave * kQuadGames + gi) * 128 : nullptr;
    np = QuadStagesFor<M, D, 1, HOT == kHotNone || HOT == kHotSort>::run(q, smax, c, axis_in, np, j, flags, stages, cmine,
                                                                         false, rank, tsc);
  }
  const bool done = np < 2;
  if (leader) {  // (scalar base + the lane's game index: no 64-bit address arithmetic per output)
    const unsigned ug = (unsigned)gi;
    if (prm.done_out) (prm.done_out + g0)[ug] = done;
    if (prm.prev_done_out) (prm.prev_done_out + g0)[ug] = prev_done;
    if (prm.reward_out) (prm.reward_out + g0)[ug] = prm.reward_sign * (float)(done && !prev_done);
    if (prm.num_points_out) (prm.num_points_out + g0)[ug] = np;
  }
  if constexpr (G::kBig) {
#ifdef HK_QUAD_PROBE
    if (cut == 4) {
      float acc = 0.0f;
#pragma unroll
      for (int e = 0; e < R * D; ++e) acc += (q[e] < INFINITY) ? q[e] : 0.0f;
      if (leader && prm.reward_out) prm.reward_out[g] = acc;
      quad_slab_store<M, D>(image, (float*)prm.out + g0 * G::N, ngames, lane);
      return;
    }
#endif
    if ((HOT == kHotNone || HOT == kHotSort) && (stages & kStageFeatureSorts)) {
      // ---- the observation features (hk_get_features / hk_get_features_torch): padding everywhere (every lane holds
      // its rows in registers: the region is free), then every live row at its rank in descending key order ----------
      wave_lds_fence();
#pragma unroll
      for (int it = 0; it < G::QL; ++it) {
        const int qq = lane + it * kWave;
        if (qq < kQuadGames * G::Q) {
          if constexpr (G::W == 4) *reinterpret_cast<vf4*>(image + qq * 4) = vf4{pad, pad, pad, pad};
          else if constexpr (G::W == 2) *reinterpret_cast<vf2*>(image + qq * 2) = vf2{pad, pad};
          else image[qq] = pad;
        }
      }
      wave_lds_fence();
      unrolled_while<0, R>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        if (s >= smax) return false;
        if (q[s * D] < INFINITY) {
          float* dst = mine + rank[s] * D;
          if constexpr (D == 4) {
            *reinterpret_cast<vf4*>(dst) = vf4{q[s * D], q[s * D + 1], q[s * D + 2], q[s * D + 3]};
          } else {
#pragma unroll
            for (int k = 0; k < D; ++k) dst[k] = q[s * D + k];
          }
        }
        return true;
      });
    } else {
    // ---- large games: every slot in use returns its row (padding if it was removed) to its compact slot; the lane
    // that OWNS row i (it scanned it, it knows its rank) picks it up there and rebuilds its rows of the image: a dead
    // row is padding, as it was.  All reads of the compact image precede the first write of the slab image in the
    // wave's program order, which is the order LDS serves them in. ----------------------------------------------------
    unrolled_while<0, R>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
      if (s >= smax) return false;
      if (kQuad * s + j < np_before) {
        const bool removed = !(q[s * D] < INFINITY);
        float* dst = cmine + (kQuad * s + j) * G::CW;
        if constexpr (D == 4) {
          *reinterpret_cast<vf4*>(dst) = vf4{removed ? pad : q[s * D], removed ? pad : q[s * D + 1],
                                             removed ? pad : q[s * D + 2], removed ? pad : q[s * D + 3]};
        } else {
#pragma unroll
          for (int k = 0; k < D; ++k) dst[k] = removed ? pad : q[s * D + k];
        }
      }
      return true;
    });
    wave_lds_fence();
    float fin[R * D];
    {
      int rank = below;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const bool live = (lmask >> r) & 1u;
        const float* src = cmine + (live ? rank : 0) * G::CW;
        if constexpr (D == 4) {
          const vf4 v = *reinterpret_cast<const vf4*>(src);
          fin[r * D] = live ? v.x : pad;
          fin[r * D + 1] = live ? v.y : pad;
          fin[r * D + 2] = live ? v.z : pad;
          fin[r * D + 3] = live ? v.w : pad;
        } else {
#pragma unroll
          for (int k = 0; k < D; ++k) {
            const float v = src[k];
            fin[r * D + k] = live ? v : pad;
          }
        }
        rank += live ? 1 : 0;
      }
    }
#pragma unroll
    for (int e = 0; e < R * D; ++e) asm volatile("" : "+v"(fin[e]));  // every read is issued before the first write
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (M % kQuad == 0 || i0 + r < M) {
        float* dst = mine + (i0 + r) * D;
        if constexpr (D == 4) {
          *reinterpret_cast<vf4*>(dst) = vf4{fin[r * D], fin[r * D + 1], fin[r * D + 2], fin[r * D + 3]};
        } else {
#pragma unroll
          for (int k = 0; k < D; ++k) dst[k] = fin[r * D + k];
        }
      }
    }
    }  // (not the sorted features)
  }
  wave_lds_fence();
#ifdef HK_QUAD_PROBE
  tl2 = wall_clock64();
#endif
  quad_slab_store<M, D>(image, (float*)prm.out + g0 * G::N, ngames, lane);
  if constexpr (FEAT) quad_slab_store<M, D>(compact, prm.feat_out + g0 * G::N, ngames, lane);
#ifdef HK_QUAD_PROBE
  if (cut == 9 && prm.num_points_out && ngames == kQuadGames) {
    wait_vmem_all();  // the slab's stores have left the wave
    const long long tl3 = wall_clock64();
    if (lane == 0) {
      int32_t* w = prm.num_points_out + g0;
      w[0] = (int32_t)tl0;
      w[1] = (int32_t)tl1;
      w[2] = (int32_t)tl2;
      w[3] = (int32_t)tl3;
      w[4] = smax;
      w[5] = (int32_t)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));  // HW_ID
      w[6] = (int32_t)blockIdx.x;
      w[7] = wave;
    }
  }
#endif
}

// ---- host side -------------------------------------------------------------------------------------------------------
template <int M, int D>
constexpr int quad_waves_per_block() {
  // LDS per wave = image + compact image; keep a workgroup at or below 64 KiB of static LDS
  using G = QuadGeom<M, D>;
  return (G::kRegion * 4 * 4 + 4 * kQuadGames * D * 4 <= 64 * 1024) ? 4 : 1;
}

template <int M, int D, int WPB, int HOT, int ACT, bool FEAT = false>
void launch_quad_k(const Params& prm, unsigned grid, hipStream_t stream) {
  size_t dynamic_lds = 0;
#ifdef HK_QUAD_PROBE  // unused dynamic LDS: caps the workgroups per CU (how much does a second round of waves buy?)
  const char* e = getenv("HK_QUAD_DLDS");
  dynamic_lds = e ? (size_t)atoi(e) : 0;
#endif
  hipLaunchKernelGGL((quad_kernel<M, D, HOT, WPB, ACT, FEAT>), dim3(grid), dim3(kWave * WPB), dynamic_lds, stream,
                     (const float*)prm.in, prm.in_stride, prm.batch, prm);
}

template <int M, int D, int WPB>
int launch_quad_w(Params prm, hipStream_t stream) {
  const int64_t waves = ((int64_t)prm.batch + kQuadGames - 1) / kQuadGames;
  const unsigned grid = (unsigned)((waves + WPB - 1) / WPB);
  prm.games_per_block = kQuadGames * WPB;
  prm.pad_f32 = (float)prm.pad;
  launch_prepare();
  const bool hot = prm.flags == HK_SEM_JAX && prm.stages == (HK_STAGE_SHIFT | HK_STAGE_REPOSITION | HK_STAGE_NEWTON);
  const int act = quad_act_of(prm);
  if (prm.feat_out) {
    // the step + features launch of the search's expansion: class-id subsets with an int32 axis or the agent's logits
    // (or, in the JAX trainer's configuration, a float mask with an int32 axis), no sorted output, small games
    if constexpr (D <= kQuad && !QuadGeom<M, D>::kBig) {
      const bool plain = !(prm.stages & kStageFeatureSorts) && (prm.flags & HK_SEM_MASK) != HK_SEM_LIST &&
                         !(prm.flags & HK_FLAG_COMPACT_SORTED);
      if (!plain) return HK_ERR_UNSUPPORTED;
      if (act == kActClassI32Logits) {
        if (hot) launch_quad_k<M, D, WPB, kHotJax, kActClassI32Logits, true>(prm, grid, stream);
        else launch_quad_k<M, D, WPB, kHotNone, kActClassI32Logits, true>(prm, grid, stream);
      } else if (act == kActClassI32AxisI32) {
        if (hot) launch_quad_k<M, D, WPB, kHotJax, kActClassI32AxisI32, true>(prm, grid, stream);
        else launch_quad_k<M, D, WPB, kHotNone, kActClassI32AxisI32, true>(prm, grid, stream);
      } else if (act == kActMaskF32AxisI32 && hot) {  // (the agent-role tree: float mask from the embedding)
        launch_quad_k<M, D, WPB, kHotJax, kActMaskF32AxisI32, true>(prm, grid, stream);
      } else {
        return HK_ERR_UNSUPPORTED;
      }
      return launch_status();
    } else {
      return HK_ERR_UNSUPPORTED;
    }
  }
  if constexpr (QuadGeom<M, D>::kBig) {
    // the sorted observation features of the large games: the instantiation without the step stages (three waves per SIMD)
    if ((prm.stages & kStageFeatureSorts) && !(prm.stages & (HK_STAGE_SHIFT | HK_STAGE_REPOSITION | HK_STAGE_NEWTON))) {
      launch_quad_k<M, D, WPB, kHotSort, kActAny>(prm, grid, stream);
      return launch_status();
    }
  }
  if (hot) {  // the JAX trainer's take_actions, with the trainers' action layouts compiled in
    switch (act) {
      case kActMaskF32AxisI32: launch_quad_k<M, D, WPB, kHotJax, kActMaskF32AxisI32>(prm, grid, stream); break;
      case kActMaskF32AxisI64: launch_quad_k<M, D, WPB, kHotJax, kActMaskF32AxisI64>(prm, grid, stream); break;
      case kActMaskF32AxisF32: launch_quad_k<M, D, WPB, kHotJax, kActMaskF32AxisF32>(prm, grid, stream); break;
      case kActClassI32AxisI32: launch_quad_k<M, D, WPB, kHotJax, kActClassI32AxisI32>(prm, grid, stream); break;
      case kActClassI32Logits:
        if constexpr (D <= kQuad && !QuadGeom<M, D>::kBig) {
          launch_quad_k<M, D, WPB, kHotJax, kActClassI32Logits>(prm, grid, stream);
          break;
        } else {
          return HK_ERR_UNSUPPORTED;
        }
      default: launch_quad_k<M, D, WPB, kHotJax, kActAny>(prm, grid, stream);
    }
  } else if (act == kActClassI32Logits) {
    if constexpr (D <= kQuad && !QuadGeom<M, D>::kBig) launch_quad_k<M, D, WPB, kHotNone, kActClassI32Logits>(prm, grid, stream);
    else return HK_ERR_UNSUPPORTED;
  } else if (act == kActClassI32AxisI32) {
    launch_quad_k<M, D, WPB, kHotNone, kActClassI32AxisI32>(prm, grid, stream);
  } else if (act == kActMaskF32AxisI64) {
    launch_quad_k<M, D, WPB, kHotNone, kActMaskF32AxisI64>(prm, grid, stream);
  } else {
    launch_quad_k<M, D, WPB, kHotNone, kActAny>(prm, grid, stream);
  }
  return launch_status();
}

template <int M, int D>
int launch_quad_t(Params prm, hipStream_t stream) {
#ifdef HK_QUAD_PROBE
  const char* e = getenv("HK_QUAD_CUT");
  prm.lds_stride = e ? atoi(e) : 0;
  const char* w = getenv("HK_QUAD_WPB");
  const int wpb = w ? atoi(w) : 0;
  if constexpr (quad_waves_per_block<M, D>() > 1) {
    if (wpb == 1) return launch_quad_w<M, D, 1>(prm, stream);
    if (wpb == 2) return launch_quad_w<M, D, 2>(prm, stream);
    if (wpb == 8) return launch_quad_w<M, D, 8>(prm, stream);
  }
#endif
  return launch_quad_w<M, D, quad_waves_per_block<M, D>()>(prm, stream);
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
  if (prm.class_out) return false;
  if (prm.coords_kind == HK_COORDS_IN_RECORD) return false;
  if (prm.flags & (HK_FLAG_FORCE_GENERIC | HK_FLAG_FORCE_TEAM | HK_FLAG_FORCE_ONE_LANE | HK_FLAG_FORCE_TWO_LANES))
    return false;
  const bool sorted = (prm.stages & HK_STAGE_NEWTON) &&
                      ((prm.flags & HK_SEM_MASK) == HK_SEM_LIST || (prm.flags & HK_FLAG_COMPACT_SORTED));
  if (sorted && prm.m * prm.d > 128) return false;  // (the large games' sorted output stays with the team kernel)
#define HK_X(M_, D_) if (prm.m == M_ && prm.d == D_) return quad_ok_t<M_, D_>(prm);
  HK_QUAD_SPECS(HK_X)
#undef HK_X
  return false;
}

// where the four-lane kernel is the default choice: everywhere it applies (scripts/probe_crossover.py: ahead of the
// one-lane kernel from 32 768 to 524 288 games on (10,3), (20,3), (20,4) -- e.g. (20,3): 6.0 vs 11.3 us at 32 768,
// 44.9 vs 52.7 us at 524 288 -- and of the team kernel on (50,4): 94 vs 113 us at 262 144)
inline bool quad_default(const Params&, int) { return true; }

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
