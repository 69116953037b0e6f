// Tree operations of the batched Gumbel-MuZero search: what the reference gets from
// `mctx.gumbel_muzero_policy` + `qtransform_completed_by_mix_value(use_mixed_value=True)`
// (hironaka/jax/simulation_fn.py:85-117).  mctx is a third-party package that is neither vendored nor
// installed: the algorithm is the published one (Danihelka et al., "Policy improvement by planning with
// Gumbel", ICLR 2022) and these kernels are pinned to oracle/search_oracle.py -- PARITY UNPINNED against
// mctx itself.
//
// One lane per game walks ITS tree (batch-first arrays, as mctx lays them out and as
// simulation_fn.py:176-186 reads them back).  A tree of 33 nodes x 4 actions is ~3 KB per game, 25 MB for
// 8192 games: resident in L2 / MALL, so the lane-strided accesses are cache hits; the environment step
// and the network evaluation between two tree operations dominate a simulation.
//
// Arithmetic contract (shared with the oracle): statistics are float32 and the backward pass is plain
// float32; whatever feeds an argmax or the final softmax is float64 computed from those statistics.
#pragma once

#include "hk_common.h"

namespace hk {

constexpr int kSearchMaxActions = 32;

struct SearchTree {
  int32_t* node_visits;
  float* raw_values;
  float* node_values;
  int32_t* parents;
  int32_t* action_from_parent;
  int32_t* children_index;
  float* children_prior_logits;
  int32_t* children_visits;
  float* children_rewards;
  float* children_discounts;
  float* children_values;
  int32_t batch, num_nodes, num_actions;
};

// completed Q-values of node `n` of game `b` (appendix D of the paper; mctx defaults value_scale = 0.1,
// maxvisit_init = 50, rescale_values, epsilon = 1e-8) into cq[0..A).  AMAX = compile-time bound of the
// action count: the per-action arrays live in registers and every exp() is evaluated once.
template <int AMAX>
__device__ inline void search_completed_q(const SearchTree& t, int64_t b, int n, double (&cq)[AMAX]) {
  const int A = t.num_actions;
  const int64_t e0 = (b * t.num_nodes + n) * A;
  double logit[AMAX], p[AMAX];
  int visits[AMAX];
  double mx = -INFINITY;
#pragma unroll
  for (int a = 0; a < AMAX; ++a)
    if (a < A) {
      logit[a] = (double)t.children_prior_logits[e0 + a];
      visits[a] = t.children_visits[e0 + a];
      cq[a] = (double)t.children_rewards[e0 + a] +
              (double)t.children_discounts[e0 + a] * (double)t.children_values[e0 + a];
      mx = fmax(mx, logit[a]);
    }
  double den = 0.0;
#pragma unroll
  for (int a = 0; a < AMAX; ++a)
    if (a < A) {
      p[a] = exp(logit[a] - mx);
      den += p[a];
    }
  double sum_probs = 0.0, sum_visits = 0.0;
  int maxvisit = 0;
#pragma unroll
  for (int a = 0; a < AMAX; ++a)
    if (a < A) {
      p[a] = fmax(1.1754943508222875e-38, p[a] / den);
      if (visits[a] > 0) sum_probs += p[a];
      sum_visits += (double)visits[a];
      maxvisit = visits[a] > maxvisit ? visits[a] : maxvisit;
    }
  double weighted_q = 0.0;
#pragma unroll
  for (int a = 0; a < AMAX; ++a)
    if (a < A && visits[a] > 0) weighted_q += p[a] * cq[a] / sum_probs;
  const double raw = (double)t.raw_values[b * t.num_nodes + n];
  const double value = (raw + sum_visits * weighted_q) / (sum_visits + 1.0);
  double lo = INFINITY, hi = -INFINITY;
#pragma unroll
  for (int a = 0; a < AMAX; ++a)
    if (a < A) {
      if (!(visits[a] > 0)) cq[a] = value;
      lo = fmin(lo, cq[a]);
      hi = fmax(hi, cq[a]);
    }
  const double scale = (50.0 + (double)maxvisit) * 0.1;
  const double span = fmax(hi - lo, 1e-8);
#pragma unroll
  for (int a = 0; a < AMAX; ++a)
    if (a < A) cq[a] = scale * ((cq[a] - lo) / span);
}

// Gumbel + logits + completed Q of the root actions whose visit count equals `considered_visit`
// (sequential halving), -inf for the others; first maximum
template <int AMAX>
__device__ inline int search_root_argmax(const SearchTree& t, int64_t b, const double (&cq)[AMAX], const float* gumbel,
                                         const uint8_t* invalid, int considered_visit) {
  const int A = t.num_actions;
  const int64_t e0 = b * t.num_nodes * A;
  double mx = -INFINITY;
#pragma unroll
  for (int a = 0; a < AMAX; ++a)
    if (a < A) mx = fmax(mx, (double)t.children_prior_logits[e0 + a]);
  int best = 0;
  double best_s = -INFINITY;
#pragma unroll
  for (int a = 0; a < AMAX; ++a)
    if (a < A) {
      double s = fmax(-1e9, (double)gumbel[b * A + a] + ((double)t.children_prior_logits[e0 + a] - mx) + cq[a]);
      if (t.children_visits[e0 + a] != considered_visit) s = -INFINITY;
      if (invalid && invalid[b * A + a]) s = -INFINITY;
      if (s > best_s) {
        best_s = s;
        best = a;
      }
    }
  return best;
}

// ---- the descent with ONE LANE PER ACTION (AMAX lanes per game).  With one lane per game every level of the walk
// was 2 A double-precision exp() calls one after the other behind a dependent load -- 16 us per simulation for 8192
// games, the largest HIP kernel of a search; here the A exponentials of a level run side by side and a level's loads
// are one coalesced request per array.  Sums over the actions keep the sequential order of the one-lane code (every
// lane adds the group's values in order 0 .. A-1: bit-identical results); maxima are order-free.
template <int AMAX>
__device__ inline double grp_get(double v, int src) { return __shfl(v, src, AMAX); }
template <int AMAX>
__device__ inline double grp_max(double v) {
#pragma unroll
  for (int off = AMAX / 2; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, AMAX));
  return v;
}
template <int AMAX>
__device__ inline double grp_min(double v) {
#pragma unroll
  for (int off = AMAX / 2; off > 0; off >>= 1) v = fmin(v, __shfl_xor(v, off, AMAX));
  return v;
}
template <int AMAX>
__device__ inline int grp_isum(int v) {
#pragma unroll
  for (int off = AMAX / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, AMAX);
  return v;
}
template <int AMAX>
__device__ inline int grp_imax(int v) {
#pragma unroll
  for (int off = AMAX / 2; off > 0; off >>= 1) {
    const int o = __shfl_xor(v, off, AMAX);
    v = o > v ? o : v;
  }
  return v;
}
template <int AMAX>
__device__ inline double grp_seq_sum(double v, int A) {  // v_0 + v_1 + ... in that order (lanes >= A hold 0)
  double s = 0.0;
  for (int k = 0; k < A; ++k) s += grp_get<AMAX>(v, k);
  return s;
}
template <int AMAX>
__device__ inline int grp_first_argmax(double v, int A) {  // the one-lane loop: `if (s > best_s)` over a = 0 .. A-1
  int best = 0;
  double best_s = -INFINITY;
  for (int k = 0; k < A; ++k) {
    const double sk = grp_get<AMAX>(v, k);
    if (sk > best_s) {
      best_s = sk;
      best = k;
    }
  }
  return best;
}

// completed Q-value of MY action at node n (search_completed_q, one lane per action); also returns the lane's prior
// logit, its visit count, the group's maximum logit and total visit count
template <int AMAX>
__device__ inline double lane_completed_q(const SearchTree& t, int64_t b, int n, int a, double& logit_out,
                                          int& visits_out, double& mx_out, double& sum_visits_out) {
  const int A = t.num_actions;
  const bool on = a < A;
  const int64_t e = (b * t.num_nodes + n) * A + (on ? a : 0);
  const double logit = on ? (double)t.children_prior_logits[e] : -INFINITY;
  const int visits = on ? t.children_visits[e] : 0;
  double cq = on ? (double)t.children_rewards[e] + (double)t.children_discounts[e] * (double)t.children_values[e] : 0.0;
  const double raw = (double)t.raw_values[b * t.num_nodes + n];
  const double mx = grp_max<AMAX>(logit);
  double p = on ? exp(logit - mx) : 0.0;
  const double den = grp_seq_sum<AMAX>(p, A);
  p = on ? fmax(1.1754943508222875e-38, p / den) : 0.0;
  const double sum_probs = grp_seq_sum<AMAX>(visits > 0 ? p : 0.0, A);
  const double sum_visits = (double)grp_isum<AMAX>(visits);
  const int maxvisit = grp_imax<AMAX>(visits);
  const double weighted_q = grp_seq_sum<AMAX>(visits > 0 ? p * cq / sum_probs : 0.0, A);
  const double value = (raw + sum_visits * weighted_q) / (sum_visits + 1.0);
  if (!(visits > 0)) cq = value;
  const double lo = grp_min<AMAX>(on ? cq : INFINITY), hi = grp_max<AMAX>(on ? cq : -INFINITY);
  const double scale = (50.0 + (double)maxvisit) * 0.1;
  const double span = fmax(hi - lo, 1e-8);
  logit_out = logit;
  visits_out = visits;
  mx_out = mx;
  sum_visits_out = sum_visits;
  return scale * ((cq - lo) / span);
}

// one simulation's descent: the edge (parent, action) to expand and the node index the expansion writes
// (the existing child at the depth limit, else `next_free`)
template <int AMAX>
__device__ inline void search_descend(const SearchTree& t, int64_t b, int a, const float* gumbel,
                                      const uint8_t* invalid, const int32_t* table, int max_considered,
                                      int num_simulations, int max_depth, int next_free, int32_t* parent_out,
                                      int32_t* action_out, int32_t* node_out) {
  const int A = t.num_actions;
  const bool on = a < A;
  // root: sequential halving over the Gumbel-top-k actions
  double logit, mx, sum_visits;
  int visits;
  double cq = lane_completed_q<AMAX>(t, b, 0, a, logit, visits, mx, sum_visits);
  const bool bad = on && invalid && invalid[b * A + a];
  const int num_valid = A - grp_isum<AMAX>(bad ? 1 : 0);
  const int sim_index = grp_isum<AMAX>(visits);
  const int num_considered = max_considered < num_valid ? max_considered : num_valid;
  const int considered_visit = table[(int64_t)num_considered * num_simulations + sim_index];
  double s = -INFINITY;
  if (on) {
    s = fmax(-1e9, (double)gumbel[b * A + a] + (logit - mx) + cq);
    if (visits != considered_visit) s = -INFINITY;
    if (bad) s = -INFINITY;
  }
  int node = 0;
  int action = grp_first_argmax<AMAX>(s, A);
  int next = t.children_index[(b * t.num_nodes + node) * A + action];
  int depth = 0;
  while (next != -1 && depth + 1 < max_depth) {  // (uniform inside a game's group of lanes)
    node = next;
    ++depth;
    // interior: argmax(softmax(logits + completed Q) - N / (1 + sum N))
    cq = lane_completed_q<AMAX>(t, b, node, a, logit, visits, mx, sum_visits);
    const double z = on ? cq + logit : -INFINITY;
    const double mz = grp_max<AMAX>(z);
    const double ez = on ? exp(z - mz) : 0.0;
    const double den = grp_seq_sum<AMAX>(ez, A);
    s = on ? ez / den - (double)visits / (1.0 + sum_visits) : -INFINITY;
    action = grp_first_argmax<AMAX>(s, A);
    next = t.children_index[(b * t.num_nodes + node) * A + action];
  }
  if (a == 0) {
    parent_out[b] = node;
    action_out[b] = action;
    node_out[b] = next == -1 ? next_free : next;
  }
}

template <int AMAX>
__global__ __launch_bounds__(256) void search_select_kernel(SearchTree t, const float* gumbel, const uint8_t* invalid,
                                                            const int32_t* table, int max_considered,
                                                            int num_simulations, int max_depth, int next_free,
                                                            int32_t* parent_out, int32_t* action_out,
                                                            int32_t* node_out) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int a = (int)(tid % AMAX);
  const int64_t game = tid / AMAX;
  if (game >= t.batch) return;  // (whole groups: a game's lanes stay together)
  search_descend<AMAX>(t, game, a, gumbel, invalid, table, max_considered, num_simulations, max_depth, next_free,
                       parent_out, action_out, node_out);
}

// expansion (the new node's statistics and its edge) + the backward pass to the root
__device__ inline void search_backup_game(const SearchTree& t, int64_t b, const int32_t* parent_in,
                                          const int32_t* action_in, const int32_t* node_in, const float* prior_logits,
                                          const float* value, const float* reward, const float* discount) {
  const int A = t.num_actions, N = t.num_nodes;
  const int parent = parent_in[b], action = action_in[b], node = node_in[b];
  for (int a = 0; a < A; ++a) t.children_prior_logits[(b * N + node) * A + a] = prior_logits[b * A + a];
  t.raw_values[b * N + node] = value[b];
  t.node_values[b * N + node] = value[b];
  t.node_visits[b * N + node] += 1;
  t.children_index[(b * N + parent) * A + action] = node;
  t.children_rewards[(b * N + parent) * A + action] = reward[b];
  t.children_discounts[(b * N + parent) * A + action] = discount[b];
  t.parents[b * N + node] = parent;
  t.action_from_parent[b * N + node] = action;
  int index = node;
  float leaf_value = value[b];
  while (index != 0) {
    const int p = t.parents[b * N + index];
    const int act = t.action_from_parent[b * N + index];
    const int64_t e = (b * N + p) * A + act;
    const float count = (float)t.node_visits[b * N + p];
    leaf_value = t.children_rewards[e] + t.children_discounts[e] * leaf_value;
    const float parent_value = (t.node_values[b * N + p] * count + leaf_value) / (count + 1.0f);
    t.children_values[e] = t.node_values[b * N + index];
    t.children_visits[e] += 1;
    t.node_values[b * N + p] = parent_value;
    t.node_visits[b * N + p] += 1;
    index = p;
  }
}

__global__ void search_backup_kernel(SearchTree t, const int32_t* parent_in, const int32_t* action_in,
                                     const int32_t* node_in, const float* prior_logits, const float* value,
                                     const float* reward, const float* discount) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= t.batch) return;
  search_backup_game(t, b, parent_in, action_in, node_in, prior_logits, value, reward, discount);
}

// the improved policy at the root after the last simulation: action (Gumbel argmax among the most
// visited) and action_weights = softmax(logits + completed Q)
template <int AMAX>
__global__ __launch_bounds__(64) void search_policy_kernel(SearchTree t, const float* gumbel, const uint8_t* invalid,
                                                           int32_t* action_out, float* weights_out) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= t.batch) return;
  const int A = t.num_actions;
  const int64_t e0 = b * t.num_nodes * A;
  double cq[AMAX];
  search_completed_q<AMAX>(t, b, 0, cq);
  int maxvisit = 0;
  for (int a = 0; a < A; ++a) maxvisit = t.children_visits[e0 + a] > maxvisit ? t.children_visits[e0 + a] : maxvisit;
  action_out[b] = search_root_argmax<AMAX>(t, b, cq, gumbel, invalid, maxvisit);
  double mx = -INFINITY;
#pragma unroll
  for (int a = 0; a < AMAX; ++a)
    if (a < A) {
      cq[a] += (double)t.children_prior_logits[e0 + a];
      mx = fmax(mx, cq[a]);
    }
  if (invalid) {
#pragma unroll
    for (int a = 0; a < AMAX; ++a)
      if (a < A) cq[a] = invalid[b * A + a] ? -3.4028234663852886e38 : cq[a] - mx;
    mx = -INFINITY;
#pragma unroll
    for (int a = 0; a < AMAX; ++a)
      if (a < A) mx = fmax(mx, cq[a]);
  }
  double den = 0.0;
#pragma unroll
  for (int a = 0; a < AMAX; ++a)
    if (a < A) {
      cq[a] = exp(cq[a] - mx);
      den += cq[a];
    }
#pragma unroll
  for (int a = 0; a < AMAX; ++a)
    if (a < A) weights_out[b * A + a] = (float)(cq[a] / den);
}

// AMAX = the smallest of 4 / 8 / 16 / 32 that covers the action count
#define HK_SEARCH_DISPATCH(A_, CALL)          \
  do {                                        \
    if ((A_) <= 4) { CALL(4); }               \
    else if ((A_) <= 8) { CALL(8); }          \
    else if ((A_) <= 16) { CALL(16); }        \
    else { CALL(32); }                        \
  } while (0)

inline int launch_search_select(const SearchTree& t, const float* gumbel, const uint8_t* invalid,
                                const int32_t* table, int max_considered, int num_simulations, int max_depth,
                                int next_free, int32_t* parent_out, int32_t* action_out, int32_t* node_out,
                                hipStream_t stream) {
  launch_prepare();
#define HK_CALL(AM)                                                                                               \
  hipLaunchKernelGGL(search_select_kernel<AM>, dim3((unsigned)(((int64_t)t.batch * AM + 255) / 256)), dim3(256), 0,  \
                     stream, t, gumbel, invalid, table, max_considered, num_simulations, max_depth, next_free,     \
                     parent_out, action_out, node_out)
  HK_SEARCH_DISPATCH(t.num_actions, HK_CALL);
#undef HK_CALL
  return launch_status();
}

inline int launch_search_backup(const SearchTree& t, const int32_t* parent, const int32_t* action, const int32_t* node,
                                const float* prior_logits, const float* value, const float* reward,
                                const float* discount, hipStream_t stream) {
  launch_prepare();
  hipLaunchKernelGGL(search_backup_kernel, dim3((t.batch + 63) / 64), dim3(64), 0, stream, t, parent, action, node,
                     prior_logits, value, reward, discount);
  return launch_status();
}

inline int launch_search_policy(const SearchTree& t, const float* gumbel, const uint8_t* invalid, int32_t* action_out,
                                float* weights_out, hipStream_t stream) {
  launch_prepare();
#define HK_CALL(AM)                                                                                               \
  hipLaunchKernelGGL(search_policy_kernel<AM>, dim3((t.batch + 63) / 64), dim3(64), 0, stream, t, gumbel, invalid, \
                     action_out, weights_out)
  HK_SEARCH_DISPATCH(t.num_actions, HK_CALL);
#undef HK_CALL
  return launch_status();
}

// ---- expansion glue of a host-role tree (hironaka/jax/recurrent_fn.py:84-104 between the search's select and its
// backward pass).  The tree keeps, per node, the points (embeddings [B, N, E], E = max_points * dim) and their
// observation features (features [B, N, E]: what hk_get_features makes of the points -- computed ONCE, when the node
// is created, for the host network; the agent network of a later expansion reads the same rows).  Three small
// kernels replace the ~9 tensor-library launches of an expansion (gather, decode, two concatenations, the feature
// sort of the strided agent observation, mask fill / compare / where / argmax, index_put). ----------------------------

// obs_out [B, E] = embeddings[b, parent[b]];  agent_feat_out [B, E + d] = features[b, parent[b]] ++ the 0/1 subset of
// the host's class id (host_action_preprocess: class -> mask, out-of-range ids clamped like hk_decode_host_class)
// (game_stride / node_stride, in elements: [B, N, E] tables pass N * E and E, node-major [N, B, E] ones E and B * E)
__global__ void expand_gather_kernel(const float* emb, const float* feat, const int32_t* parent, const int32_t* action,
                                     float* obs_out, float* agent_feat_out, int batch, int nodes, int E, int d,
                                     int64_t game_stride, int64_t node_stride) {
  // (blockIdx.x = the game, blockIdx.y = chunk of its row: no division of a 64-bit linear index by a run-time length)
  const int per = 2 * E + d;
  const int g = blockIdx.x, e = blockIdx.y * blockDim.x + threadIdx.x;
  if (e >= per) return;
  int p = parent[g];
  p = p < 0 ? 0 : (p >= nodes ? nodes - 1 : p);
  const int64_t row = (int64_t)g * game_stride + (int64_t)p * node_stride;
  if (e < E) {
    obs_out[(int64_t)g * E + e] = emb[row + e];
  } else if (e < 2 * E) {
    agent_feat_out[(int64_t)g * (E + d) + (e - E)] = feat[row + (e - E)];
  } else {
    const int k = e - 2 * E;
    const int ncls = (int)((1ll << d) - d - 1);
    int c = action[g];
    c = c < 0 ? 0 : (c >= ncls ? ncls - 1 : c);
    agent_feat_out[(int64_t)g * (E + d) + E + k] = (float)((decode_class(c, d) >> k) & 1u);
  }
}

// the agent's answer: argmax of its logits over the coordinates of the host's subset (jax/util.py:287-327: logits
// outside the subset count as -inf; the first maximum wins; a NaN beats every number, as in the tensor libraries)
__global__ void masked_argmax_kernel(const float* logits, const int32_t* action, int32_t* axis_out, int batch, int d) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= batch) return;
  const int ncls = (int)((1ll << d) - d - 1);
  int c = action[g];
  c = c < 0 ? 0 : (c >= ncls ? ncls - 1 : c);
  const uint32_t mask = decode_class(c, d);
  int best = 0;
  float bv = ((mask >> 0) & 1u) ? logits[(int64_t)g * d] : -INFINITY;
  for (int k = 1; k < d; ++k) {
    const float v = ((mask >> k) & 1u) ? logits[(int64_t)g * d + k] : -INFINITY;
    if ((v > bv) || (v != v && bv == bv)) {
      best = k;
      bv = v;
    }
  }
  axis_out[g] = best;
}

// embeddings[b, node[b]] = obs[b];  features[b, node[b]] = feat[b]
__global__ void expand_scatter_kernel(const float* obs, const float* feat_in, const int32_t* node, float* emb,
                                      float* feat, int batch, int nodes, int E) {
  const int g = blockIdx.x, e = blockIdx.y * blockDim.x + threadIdx.x;
  if (e >= 2 * E) return;
  const int n = node[g];
  if (n < 0 || n >= nodes) return;
  const int64_t row = ((int64_t)g * nodes + n) * E;
  if (e < E) emb[row + e] = obs[(int64_t)g * E + e];
  else feat[row + (e - E)] = feat_in[(int64_t)g * E + (e - E)];
}

// ---- the same for an AGENT-role tree (recurrent_fn.py:105-121): a node's embedding is the agent observation, points
// followed by the host's 0/1 subset ([B, N, E + d]); the features table holds the features of the points ([B, N, E]).
// points_out [B, E], coords_out [B, d] = the two parts of embeddings[b, parent[b]] (what the step consumes)
__global__ void expand_gather_agent_kernel(const float* emb, const int32_t* parent, float* points_out,
                                           float* coords_out, int batch, int nodes, int E, int d) {
  const int per = E + d;
  const int g = blockIdx.x, e = blockIdx.y * blockDim.x + threadIdx.x;
  if (e >= per) return;
  int p = parent[g];
  p = p < 0 ? 0 : (p >= nodes ? nodes - 1 : p);
  const float v = emb[((int64_t)g * nodes + p) * per + e];
  if (e < E) points_out[(int64_t)g * E + e] = v;
  else coords_out[(int64_t)g * d + (e - E)] = v;
}

// the host's answer on the new points and the new node: class = argmax of host_logits [B, C] (first maximum, NaN
// beats every number), mask = its subset;  embeddings[b, node[b]] = points ++ mask;  features[b, node[b]] = feat;
// agent_feat_out [B, E + d] = feat ++ mask (the agent network's input);  class_out [B] = the class
__global__ void expand_scatter_agent_kernel(const float* points, const float* feat_in, const float* host_logits,
                                            const int32_t* node, float* emb, float* feat, float* agent_feat_out,
                                            int32_t* class_out, int batch, int nodes, int E, int d, int C) {
  const int per = 2 * E + d;  // E points, E features, d mask entries
  const int g = blockIdx.x, e = blockIdx.y * blockDim.x + threadIdx.x;
  if (e >= per) return;
  const int n = node[g];
  const bool inside = n >= 0 && n < nodes;
  const int64_t erow = ((int64_t)g * nodes + (inside ? n : 0)) * (E + d);
  const int64_t frow = ((int64_t)g * nodes + (inside ? n : 0)) * E;
  if (e < E) {
    if (inside) emb[erow + e] = points[(int64_t)g * E + e];
  } else if (e < 2 * E) {
    const float v = feat_in[(int64_t)g * E + (e - E)];
    if (inside && feat) feat[frow + (e - E)] = v;  // (feat == NULL: the caller keeps no per-node features)
    agent_feat_out[(int64_t)g * (E + d) + (e - E)] = v;
  } else {
    const int k = e - 2 * E;
    int best = 0;
    float bv = host_logits[(int64_t)g * C];
    for (int c = 1; c < C; ++c) {
      const float v = host_logits[(int64_t)g * C + c];
      if ((v > bv) || (v != v && bv == bv)) {
        best = c;
        bv = v;
      }
    }
    const int ncls = (int)((1ll << d) - d - 1);
    const int cc = best >= ncls ? ncls - 1 : best;
    const float bit = (float)((decode_class(cc, d) >> k) & 1u);
    if (inside) emb[erow + E + k] = bit;
    agent_feat_out[(int64_t)g * (E + d) + E + k] = bit;
    if (k == 0 && class_out) class_out[g] = cc;
  }
}

// the agent's action mask on its logits (jax/util.py:287-305, NaN-free form): out[b, k] = logits[b, k] if coordinate k
// is in the subset of class_id[b], else -inf
__global__ void mask_logits_kernel(const float* logits, const int32_t* class_id, float* out, int batch, int d) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)batch * d) return;
  const int g = (int)(idx / d), k = (int)(idx % d);
  const int ncls = (int)((1ll << d) - d - 1);
  int c = class_id[g];
  c = c < 0 ? 0 : (c >= ncls ? ncls - 1 : c);
  out[idx] = ((decode_class(c, d) >> k) & 1u) ? logits[idx] : -INFINITY;
}

// the glue kernels' grid: x = the games, y = 128-element chunks of one game's row of work
constexpr int kExpandBlock = 128;
inline dim3 expand_grid(int per, int batch) { return dim3((unsigned)batch, (unsigned)((per + kExpandBlock - 1) / kExpandBlock)); }

inline int launch_expand_gather_agent(const float* emb, const int32_t* parent, float* points_out, float* coords_out,
                                      int batch, int nodes, int E, int d, hipStream_t stream) {
  launch_prepare();
  hipLaunchKernelGGL(expand_gather_agent_kernel, expand_grid(E + d, batch), dim3(kExpandBlock), 0, stream, emb,
                     parent, points_out, coords_out, batch, nodes, E, d);
  return launch_status();
}

inline int launch_expand_scatter_agent(const float* points, const float* feat_in, const float* host_logits,
                                       const int32_t* node, float* emb, float* feat, float* agent_feat_out,
                                       int32_t* class_out, int batch, int nodes, int E, int d, int C,
                                       hipStream_t stream) {
  launch_prepare();
  hipLaunchKernelGGL(expand_scatter_agent_kernel, expand_grid(2 * E + d, batch), dim3(kExpandBlock), 0, stream, points,
                     feat_in, host_logits, node, emb, feat, agent_feat_out, class_out, batch, nodes, E, d, C);
  return launch_status();
}

inline int launch_mask_logits(const float* logits, const int32_t* class_id, float* out, int batch, int d,
                              hipStream_t stream) {
  launch_prepare();
  const int64_t total = (int64_t)batch * d;
  hipLaunchKernelGGL(mask_logits_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, logits, class_id,
                     out, batch, d);
  return launch_status();
}

inline int launch_expand_gather(const float* emb, const float* feat, const int32_t* parent, const int32_t* action,
                                float* obs_out, float* agent_feat_out, int batch, int nodes, int E, int d,
                                int64_t game_stride, int64_t node_stride, hipStream_t stream) {
  launch_prepare();
  hipLaunchKernelGGL(expand_gather_kernel, expand_grid(2 * E + d, batch), dim3(kExpandBlock), 0, stream, emb, feat,
                     parent, action, obs_out, agent_feat_out, batch, nodes, E, d, game_stride, node_stride);
  return launch_status();
}

inline int launch_masked_argmax(const float* logits, const int32_t* action, int32_t* axis_out, int batch, int d,
                                hipStream_t stream) {
  launch_prepare();
  hipLaunchKernelGGL(masked_argmax_kernel, dim3((batch + 255) / 256), dim3(256), 0, stream, logits, action, axis_out,
                     batch, d);
  return launch_status();
}

inline int launch_expand_scatter(const float* obs, const float* feat_in, const int32_t* node, float* emb, float* feat,
                                 int batch, int nodes, int E, hipStream_t stream) {
  launch_prepare();
  hipLaunchKernelGGL(expand_scatter_kernel, expand_grid(2 * E, batch), dim3(kExpandBlock), 0, stream, obs, feat_in,
                     node, emb, feat, batch, nodes, E);
  return launch_status();
}

// ---- value targets of a finished self-play rollout: JAXTrainer.rollout_postprocess (jax_trainer.py:558-592) ->
// calculate_value_using_reward_fn (jax/util.py:261-284).  obs [B, T, obs_dim]: the observation before each move;
// num_points[t] = #(entries >= 0) / dim - offset; done = num_points <= 1; a move that finishes the game earns
// rew_sign; value[t] = sum_s rew[s] * clip(disc^(s - t), -1, 1)  (the discounted reward of the finishing move; constant
// after the end through the clip)  +  [game unfinished at T-1] est_scale / max(num_points[T-1], 1) * disc^(T-1-t).
// One wave per game, lane t = move t (T <= 64).
__global__ __launch_bounds__(64) void rollout_values_kernel(const float* obs, float* value_out, int batch, int T,
                                                            int obs_dim, int dim, int offset, float disc,
                                                            float rew_sign, float est_scale) {
  const int b = blockIdx.x;
  const int t = threadIdx.x;
  if (b >= batch) return;
  int np = 2;
  if (t < T) {
    const float* row = obs + ((int64_t)b * T + t) * obs_dim;
    int count = 0;
    for (int e = 0; e < obs_dim; ++e) count += (row[e] >= 0.0f) ? 1 : 0;
    np = count / dim - offset;
  }
  const bool done = np <= 1;
  const bool next_done = __shfl_down(done ? 1 : 0, 1) != 0 && t + 1 < T;
  const bool rew = t < T && next_done && !done;
  unsigned long long finishing = __ballot(rew);
  const int np_last = __shfl(np, T - 1);
  float acc = 0.0f;
  while (finishing) {  // (ascending s: the order of a row-times-table product's non-zero terms)
    const int s = __ffsll((long long)finishing) - 1;
    finishing &= finishing - 1;
    acc += rew_sign * fminf(fmaxf(powf(disc, (float)(s - t)), -1.0f), 1.0f);
  }
  const float unfinished = (np_last <= 1) ? 0.0f
                                          : est_scale / (float)(np_last < 1 ? 1 : np_last) * powf(disc, (float)(T - 1 - t));
  if (t < T) value_out[(int64_t)b * T + t] = acc + unfinished;
}

inline int launch_rollout_values(const float* obs, float* value_out, int batch, int T, int obs_dim, int dim, int offset,
                                 float disc, float rew_sign, float est_scale, hipStream_t stream) {
  launch_prepare();
  hipLaunchKernelGGL(rollout_values_kernel, dim3(batch), dim3(64), 0, stream, obs, value_out, batch, T, obs_dim, dim,
                     offset, disc, rew_sign, est_scale);
  return launch_status();
}

}  // namespace hk
