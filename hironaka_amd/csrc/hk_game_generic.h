// Per-game algorithms of the Hironaka step on an LDS-resident game `p` (row-major
// p[i*d + k], runtime m and d), one lane per game.  These back the generic kernel that
// serves every (max_points, dim, dtype) without a register-resident specialisation.
//
// Reference semantics followed (paths relative to the reference root):
//   shift       hironaka/src/_jax_ops.py:76-90, _torch_ops.py:46-110, _list_ops.py:76-101
//   reposition  _jax_ops.py:114-123, _torch_ops.py:113-133, _list_ops.py:104-133
//   newton      _jax_ops.py:15-73, _torch_ops.py:8-43 + _fn.py:192-213, _list_ops.py:9-45
//   rescale     _jax_ops.py:93-111, _torch_ops.py:136-146, _fn.py:133-153
//   zeillinger  hironaka/jax/players.py:55-109
//   features    hironaka/jax/util.py:186-197
#pragma once

#include "hk_common.h"

namespace hk {

template <typename T>
__device__ inline int num_points(const T* p, int m, int d) {
  int n = 0;
  for (int i = 0; i < m; ++i) n += (p[i * d] >= (T)0) ? 1 : 0;
  return n;
}

// x_axis <- sum_k x_k * c_k on every available row; unavailable rows/entries := pad.
template <typename T>
__device__ inline void shift_game(T* p, int m, int d, const T* c, int axis, T pad,
                                  unsigned flags) {
  const unsigned sem = flags & HK_SEM_MASK;
  if (sem == HK_SEM_TORCH) pad = torch_pad(pad);
  bool apply = true;
  if (flags & HK_FLAG_AXIS_NOOP_IF_INVALID) {
    for (int k = 0; k < d; ++k) {
      const T onehot = (k == axis) ? (T)1 : (T)0;
      if (!(onehot - c[k] <= (T)0)) apply = false;
    }
  }
  if (flags & HK_FLAG_IGNORE_ENDED) {
    if (num_points(p, m, d) < 2) apply = false;
  }
  for (int i = 0; i < m; ++i) {
    T* r = p + i * d;
    T s = (T)0;
    bool any = false;
    for (int k = 0; k < d; ++k) {
      s = s + r[k] * c[k];  // coordinate order 0..d-1, no contraction (-ffp-contract=off)
      any |= (r[k] >= (T)0);
    }
    for (int k = 0; k < d; ++k) {
      const T v = r[k];
      const bool avail = (sem == HK_SEM_TORCH) ? (v >= (T)0) : any;
      const T moved = (apply && k == axis) ? s : v;
      r[k] = avail ? moved : pad;
    }
  }
}

// (operation for operation oracle/hko_impl.inc: reposition_game -- comparisons included, so that non-finite entries, which
// only this exact path ever sees, come out as the restatement's do: under JAX semantics an unavailable entry stands in as
// the column's maximum, a NaN minimum compares false with 0 and is subtracted, _jax_ops.py:114-120)
template <typename T>
__device__ inline void reposition_game(T* p, int m, int d, T pad, unsigned flags) {
  const unsigned sem = flags & HK_SEM_MASK;
  for (int k = 0; k < d; ++k) {
    if (sem == HK_SEM_JAX) {
      T colmax = p[k];
      for (int i = 1; i < m; ++i)
        if (p[i * d + k] > colmax) colmax = p[i * d + k];
      T mn = (p[k] >= (T)0) ? p[k] : colmax;
      for (int i = 1; i < m; ++i) {
        const T v = (p[i * d + k] >= (T)0) ? p[i * d + k] : colmax;
        if (v < mn) mn = v;
      }
      if (mn <= (T)0) continue;  // column untouched (_jax_ops.py:121)
      for (int i = 0; i < m; ++i) {
        const T v = p[i * d + k];
        p[i * d + k] = (v >= (T)0) ? v - mn : pad;
      }
    } else {
      bool have = false;
      T mn = (T)0;
      for (int i = 0; i < m; ++i) {
        const T v = p[i * d + k];
        if (v >= (T)0 && (!have || v < mn)) {
          mn = v;
          have = true;
        }
      }
      for (int i = 0; i < m; ++i) {
        const T v = p[i * d + k];
        p[i * d + k] = (v >= (T)0) ? v - mn : pad;
      }
    }
  }
}

// Duplicate removal + domination test, in place, rows keep their positions.
// Sequential application is equivalent to the reference's simultaneous masks: a duplicate is
// decided against the first (never modified) occurrence, and "P_j <= P_i componentwise" is a
// strict partial order on the de-duplicated rows, so every removed row is also dominated by a
// surviving (minimal) row and dropping it early changes nothing.  The test is the reference's SUBTRACTION
// (P_i - P_j >= 0, _jax_ops.py:55-56), not a comparison: inf - inf is NaN, so two rows that are both infinite in a
// coordinate do not dominate each other -- on such rows the relation is still transitive (a row dominated through an
// infinite coordinate is dominated by finite ones only), so the sequential form keeps equalling the simultaneous one.
template <typename T>
__device__ inline void newton_game(T* p, int m, int d, T pad, unsigned flags) {
  const unsigned sem = flags & HK_SEM_MASK;
  if (sem == HK_SEM_TORCH) pad = torch_pad(pad);
  const T fill = (sem == HK_SEM_JAX) ? (T)-1 : pad;  // _jax_ops.py:65 does not forward pad
  for (int i = 1; i < m; ++i) {
    bool rep = false;
    for (int j = 0; j < i && !rep; ++j) {
      bool same = true;
      for (int k = 0; k < d; ++k) same &= (p[i * d + k] == p[j * d + k]);
      rep = same;
    }
    if (rep)
      for (int k = 0; k < d; ++k) p[i * d + k] = fill;
  }
  for (int i = 0; i < m; ++i) {
    bool ai = true;
    for (int k = 0; k < d; ++k) ai &= (p[i * d + k] >= (T)0);
    if (!ai) continue;
    bool removed = false;
    for (int j = 0; j < m && !removed; ++j) {
      if (j == i) continue;
      bool dom = true;  // row j available and P_j <= P_i
      for (int k = 0; k < d; ++k) {
        const T vj = p[j * d + k];
        dom &= (vj >= (T)0) & (p[i * d + k] - vj >= (T)0);
      }
      removed = dom;
    }
    if (removed)
      for (int k = 0; k < d; ++k) p[i * d + k] = pad;
  }
}

// list semantics (_list_ops.py:25-41): survivors sorted descending-lexicographically and
// packed to the front.  `row` is scratch for one row (d elements).
template <typename T>
__device__ inline void sort_compact_game(T* p, int m, int d, T pad, T* row) {
  int n = 0;
  for (int i = 0; i < m; ++i) {
    bool all = true;
    for (int k = 0; k < d; ++k) all &= (p[i * d + k] >= (T)0);
    if (!all) continue;
    if (n != i)
      for (int k = 0; k < d; ++k) p[n * d + k] = p[i * d + k];
    ++n;
  }
  for (int e = n * d; e < m * d; ++e) p[e] = pad;
  for (int i = 1; i < n; ++i) {
    for (int k = 0; k < d; ++k) row[k] = p[i * d + k];
    int pos = i;
    while (pos > 0) {
      bool before = false;  // row > p[pos-1] lexicographically, coordinate 0 first
      for (int k = 0; k < d; ++k) {
        const T a = row[k], b = p[(pos - 1) * d + k];
        if (a > b) { before = true; break; }
        if (a < b) break;
      }
      if (!before) break;
      for (int k = 0; k < d; ++k) p[pos * d + k] = p[(pos - 1) * d + k];
      --pos;
    }
    for (int k = 0; k < d; ++k) p[pos * d + k] = row[k];
  }
}

template <typename T>
__device__ inline void rescale_game(T* p, int m, int d, T pad, unsigned flags) {
  const unsigned sem = flags & HK_SEM_MASK;
  if (sem == HK_SEM_LIST) {
    T mx = (T)0;
    for (int i = 0; i < m; ++i) {
      bool all = true;
      for (int k = 0; k < d; ++k) all &= (p[i * d + k] >= (T)0);
      if (!all) continue;
      for (int k = 0; k < d; ++k) mx = (p[i * d + k] > mx) ? p[i * d + k] : mx;
    }
    for (int i = 0; i < m; ++i) {
      bool all = true;
      for (int k = 0; k < d; ++k) all &= (p[i * d + k] >= (T)0);
      for (int k = 0; k < d; ++k) {
        const T v = p[i * d + k];
        p[i * d + k] = all ? ((mx == (T)0) ? v : v / mx) : pad;
      }
    }
    return;
  }
  T mx = p[0];
  for (int e = 1; e < m * d; ++e) mx = (p[e] > mx) ? p[e] : mx;
  if (sem == HK_SEM_JAX) {
    const bool skip = (mx <= (T)1e-8);
    for (int i = 0; i < m; ++i) {
      bool any = false;
      for (int k = 0; k < d; ++k) any |= (p[i * d + k] >= (T)0);
      for (int k = 0; k < d; ++k) {
        const T v = p[i * d + k];
        p[i * d + k] = any ? (skip ? v : v / mx) : pad;
      }
    }
  } else {  // torch
    pad = torch_pad(pad);
    if (mx == (T)0) mx = (T)1;
    for (int e = 0; e < m * d; ++e) {
      const T v = p[e];
      p[e] = (v >= (T)0) ? v / mx : pad;
    }
  }
}

// zeillinger_fn_slice: class id of the host's choice
template <typename T>
__device__ inline int zeillinger_game(const T* p, int m, int d) {
  float bestL = INFINITY, bestS = INFINITY;
  int bi = 0, bj = 0;
  bool have = false;
  for (int i = 0; i < m; ++i) {
    bool negi = false;
    for (int k = 0; k < d; ++k) negi |= (p[i * d + k] < (T)0);
    for (int j = 0; j < m; ++j) {
      float L = INFINITY, S = INFINITY;
      bool neg = negi;
      float mx = (float)(p[i * d] - p[j * d]), mn = mx;
      for (int k = 0; k < d; ++k) {
        neg |= (p[j * d + k] < (T)0);
        const float v = (float)(p[i * d + k] - p[j * d + k]);
        mx = (v > mx) ? v : mx;
        mn = (v < mn) ? v : mn;
      }
      const bool close = fabsf(mx - mn) <= 1e-8f + 1e-5f * fabsf(mn);  // jnp.isclose
      if (!neg && !close) {
        int cnt = 0;
        for (int k = 0; k < d; ++k) {
          const float v = (float)(p[i * d + k] - p[j * d + k]);
          cnt += (v == mx) + (v == mn);
        }
        L = mx - mn;
        S = (float)cnt;
      }
      if (!have || L < bestL || (L == bestL && S < bestS)) {
        bestL = L;
        bestS = S;
        bi = i;
        bj = j;
        have = true;
      }
    }
  }
  int lo = 0, hi = 0;
  float vlo = (float)(p[bi * d] - p[bj * d]), vhi = vlo;
  for (int k = 1; k < d; ++k) {
    const float v = (float)(p[bi * d + k] - p[bj * d + k]);
    if (v < vlo) { vlo = v; lo = k; }
    if (v > vhi) { vhi = v; hi = k; }
  }
  if (lo == hi) return 0;
  return encode_mask((1u << lo) | (1u << hi));
}

// Zeillinger._select_coord (host.py:70-95) on padded rows: pairs i<j of the available rows in row
// order, key (L, S) = (max-min, #max + #min) of P_i - P_j, first minimum wins; the subset is
// {argmin, argmax} of that difference, or {0, 1} if they coincide.  -1: fewer than 2 rows.
template <typename T>
__device__ inline int zeillinger_list_game(const T* p, int m, int d) {
  T bestL = (T)0;
  int bestS = 0, bi = -1, bj = -1;
  for (int i = 0; i < m; ++i) {
    if (!(p[i * d] >= (T)0)) continue;
    for (int j = i + 1; j < m; ++j) {
      if (!(p[j * d] >= (T)0)) continue;
      T mx = p[i * d] - p[j * d], mn = mx;
      for (int k = 1; k < d; ++k) {
        const T v = p[i * d + k] - p[j * d + k];
        mx = (v > mx) ? v : mx;
        mn = (v < mn) ? v : mn;
      }
      int cnt = 0;
      for (int k = 0; k < d; ++k) {
        const T v = p[i * d + k] - p[j * d + k];
        cnt += (v == mx) + (v == mn);
      }
      const T L = mx - mn;
      if (bi < 0 || L < bestL || (L == bestL && cnt < bestS)) {
        bestL = L;
        bestS = cnt;
        bi = i;
        bj = j;
      }
    }
  }
  if (bi < 0) return -1;
  int lo = 0, hi = 0;
  T vlo = p[bi * d] - p[bj * d], vhi = vlo;
  for (int k = 1; k < d; ++k) {
    const T v = p[bi * d + k] - p[bj * d + k];
    if (v < vlo) { vlo = v; lo = k; }
    if (v > vhi) { vhi = v; hi = k; }
  }
  if (lo == hi) return encode_mask(3u);  // [0, 1]
  return encode_mask((1u << lo) | (1u << hi));
}

// stable in-place insertion sort, descending, LAST coordinate primary (lexsort(-x^T))
template <typename T>
__device__ inline void feature_sort_game(T* p, int m, int d, T* row, bool coord0) {
  for (int i = 1; i < m; ++i) {
    for (int k = 0; k < d; ++k) row[k] = p[i * d + k];
    int pos = i;
    while (pos > 0) {
      bool before = coord0 && row[0] > p[(pos - 1) * d];
      for (int k = d - 1; k >= 0 && !coord0; --k) {
        const T a = row[k], b = p[(pos - 1) * d + k];
        if (a > b) { before = true; break; }
        if (a < b) break;
      }
      if (!before) break;
      for (int k = 0; k < d; ++k) p[pos * d + k] = p[(pos - 1) * d + k];
      --pos;
    }
    for (int k = 0; k < d; ++k) p[pos * d + k] = row[k];
  }
}

// take_actions' stage order (jax/util.py:117-123); `c`/`row` live right behind the points
template <typename T>
__device__ inline void stages_game(T* p, int m, int d, T* c, int axis, T pad, unsigned stages,
                                   unsigned flags) {
  if (stages & HK_STAGE_SHIFT) shift_game(p, m, d, c, axis, pad, flags);
  if (stages & HK_STAGE_REPOSITION) reposition_game(p, m, d, pad, flags);
  if (stages & HK_STAGE_NEWTON) {
    newton_game(p, m, d, pad, flags);
    if ((flags & HK_SEM_MASK) == HK_SEM_LIST || (flags & HK_FLAG_COMPACT_SORTED))
      sort_compact_game(p, m, d, pad, c);
  }
  if (stages & HK_STAGE_RESCALE) rescale_game(p, m, d, pad, flags);
  if (stages & kStageFeatureSorts) feature_sort_game(p, m, d, c, (stages & kStageFeatureSort0) != 0);
}

}  // namespace hk
