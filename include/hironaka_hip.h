/*
 * hironaka_hip.h -- C ABI of the MI355X-native batched Hironaka-game environment.
 *
 * This is the drop-in boundary for the one hot path of honglu2875/hironaka: the
 * vectorised state transition  shift -> [reposition] -> Newton polytope -> [rescale]
 * -> done / reward  over point tensors laid out [batch, max_points, dim].
 *
 * Every entry point is `extern "C"`, takes plain pointers and sizes, never allocates,
 * never synchronises, never throws, and returns an `int` status (0 = ok, < 0 = error).
 * All data pointers are DEVICE pointers (hipMalloc'd / torch-ROCm `tensor.data_ptr()`);
 * `stream` is a `hipStream_t` passed as `void*` (NULL = the null stream).  Calls are
 * re-entrant per stream and capturable into a hipGraph.
 *
 * What each entry point replaces in the reference (paths relative to the reference root):
 *
 *   hk_step                  hironaka/jax/util.py:83-125      get_take_actions/take_actions
 *                            (+ util.py:34-35 get_dones, util.py:129-149 get_reward_fn,
 *                             recurrent_fn.py:84-121 order of operations around the step)
 *                            hironaka/trainer/fused_game.py:150-163  (torch caller)
 *                            hironaka/agent.py:69-72                  (list caller)
 *   hk_shift                 hironaka/src/_jax_ops.py:76-90   shift_jax
 *                            hironaka/src/_torch_ops.py:46-110 shift_torch
 *                            hironaka/src/_list_ops.py:76-101  shift_lst
 *   hk_reposition            _jax_ops.py:114-123 / _torch_ops.py:113-133 / _list_ops.py:104-133
 *   hk_get_newton_polytope   _jax_ops.py:32-73 (remove_repeated + get_interior)
 *                            _torch_ops.py:8-43 + _fn.py:192-213 / _list_ops.py:9-45
 *                            (native precedent: hironaka/cpp/cppUtil.cpp:58-61
 *                             getNewtonPolytope_approx, loaded by src/_np_ops.py:6-15)
 *   hk_rescale               _jax_ops.py:93-111 / _torch_ops.py:136-146 / _fn.py:133-153
 *   hk_get_dones             jax/util.py:34-35, core/tensor_points.py:118-120
 *   hk_get_num_points        core/tensor_points.py:65-70
 *   hk_generate_points       jax/util.py:385-392 generate_pts, trainer/trainer.py:592-600
 *   hk_generate_points_binned / hk_bin_by_live_rows   (no counterpart: an MI355X-side ordering of a batch)
 *   hk_rollout               jax/jax_trainer.py:502-555 compute_rho inner loop with
 *                            jax/players.py:28-39,142-212 fixed policies fused in
 *   hk_zeillinger            jax/players.py:55-109 zeillinger_fn, host.py:54-95 Zeillinger
 *   hk_get_features          jax/util.py:172-214 get_feature_fn (order_and_rescale)
 *   hk_get_features_torch    core/tensor_points.py:72-74 TensorPoints.get_features
 *   hk_decode_host_class     jax/host_action_preprocess.py:8-65, src/_fn.py:241-325
 *   hk_step_features         hk_step + hk_get_features of its result in one launch (recurrent_fn.py:84-104 + the
 *                            policy's feature function)
 *   hk_rollout_values        jax/jax_trainer.py:558-592 rollout_postprocess, jax/util.py:261-284
 *   hk_search_select / _backup / _policy
 *                            the calls into mctx at jax/simulation_fn.py:85-117
 *   hk_search_expand_gather / _masked_argmax / _expand_scatter
 *                            jax/recurrent_fn.py:84-104 (host-role tree: class id -> subset,
 *                            agent observation, the agent's masked argmax, the new embedding)
 *   hk_search_expand_gather_agent / _expand_scatter_agent / _mask_logits
 *                            jax/recurrent_fn.py:105-121 (agent-role tree) and the agent's
 *                            action mask jax/util.py:287-305
 */
#ifndef HIRONAKA_HIP_H
#define HIRONAKA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HK_ABI_VERSION 4 /* 2: + HK_AXIS_MASKED_LOGITS, hk_step_features, hk_rollout_values, hk_search_expand_* / _masked_argmax / _mask_logits; 3: + hk_rollout_desc.game_ids (and the policy stream of dim <= 8 became four steps per Philox block with 16-bit draws: seeds are not comparable with ABI 2); 4: + hk_rollout_desc.gen_max_value / gen_stages / gen_seed / episodes (initial states drawn inside the launch, `points` may be NULL), - HK_FLAG_FORCE_POOL */

/* ---- status codes -------------------------------------------------------------------- */
#define HK_OK 0
#define HK_ERR_NULL (-1)        /* a required pointer is NULL                              */
#define HK_ERR_SHAPE (-2)       /* batch/max_points/dim/stride out of range                */
#define HK_ERR_UNSUPPORTED (-3) /* dtype / kind / flag combination not implemented        */
#define HK_ERR_ALIGN (-4)       /* pointer not aligned to its element size                 */
#define HK_ERR_LAUNCH (-5)      /* hipLaunchKernel reported an error                       */
#define HK_ERR_NO_DEVICE (-6)   /* no HIP device / wrong architecture                      */

/* ---- scalar dtypes --------------------------------------------------------------------- */
#define HK_F32 0
#define HK_F64 1
#define HK_I32 2
#define HK_I64 3
#define HK_U8 4

/* ---- how the host's coordinate subset is handed over ---------------------------------- */
/* a [batch, dim] multi-binary mask of dtype HK_F32/HK_F64/HK_I32/HK_I64/HK_U8 uses the
 * dtype code itself as `coords_kind`; the remaining kinds are:                            */
#define HK_COORDS_CLASS_I32 16 /* [batch] compressed class id (host_action_preprocess.py)  */
#define HK_COORDS_CLASS_I64 17
#define HK_COORDS_IN_RECORD 18 /* mask is the `dim` elements after the points in the input
                                  record (agent observation, jax/util.py:66-74)            */
/* ---- `axis_dtype` beyond the scalar dtypes ------------------------------------------------ */
#define HK_AXIS_MASKED_LOGITS 32 /* `axis` is [batch, dim] float32 logits of the agent; its move is the argmax over
                                    the coordinates of the host's subset (jax/util.py:287-327: action mask + argmax;
                                    first maximum, NaN beats every number).  Needs HK_COORDS_CLASS_I32 and a shape
                                    with a four-lane step kernel (float32, contiguous records of (10,3) (20,3) (20,4)):
                                    anything else returns HK_ERR_UNSUPPORTED                 */
#define HK_COORDS_NONE 19      /* no shift stage                                           */

/* ---- pipeline stages (bitmask) --------------------------------------------------------- */
#define HK_STAGE_SHIFT 1u
#define HK_STAGE_REPOSITION 2u
#define HK_STAGE_NEWTON 4u
#define HK_STAGE_RESCALE 8u

/* ---- semantics: which sibling implementation of the reference is reproduced ----------- */
#define HK_SEM_JAX 0u   /* hironaka/src/_jax_ops.py   (the path the JAX trainer runs)       */
#define HK_SEM_TORCH 1u /* hironaka/src/_torch_ops.py (TensorPoints / FusedGame)           */
#define HK_SEM_LIST 2u  /* hironaka/src/_list_ops.py  (ListPoints / gym envs)              */
#define HK_SEM_MASK 3u

/* ---- behaviour flags (OR-ed with the semantics code) ---------------------------------- */
#define HK_FLAG_AXIS_NOOP_IF_INVALID 4u /* axis not in subset => no shift (torch/list)     */
#define HK_FLAG_IGNORE_ENDED 8u         /* games with < 2 points are not shifted (torch)   */
#define HK_FLAG_COMPACT_SORTED 16u      /* Newton output sorted descending-lex + compacted
                                           (list semantics, _list_ops.py:25-41)            */
#define HK_FLAG_FORCE_GENERIC 32u       /* testing: bypass the specialised kernels          */
#define HK_FLAG_FORCE_TEAM 64u          /* testing: bypass only the register-resident
                                           specialisations (use the four-lanes-per-game kernel) */
#define HK_FLAG_FORCE_ONE_LANE 256u     /* testing: one lane per game (hk::fast_kernel) where the launch would
                                         * otherwise deal a game's rows to two lanes (hk::duo_kernel)     */
#define HK_FLAG_FORCE_TWO_LANES 512u    /* testing / tuning: two lanes per game (hk::duo_kernel) at any batch size */
#define HK_FLAG_FORCE_FOUR_LANES 1024u  /* testing / tuning: four lanes per game (hk::quad_kernel) for every hk_step
                                         * it can serve, at any batch size                                       */
/* (2048u: HK_FLAG_FORCE_POOL of ABI 3 -- the block-level re-deal experiment left the library in ABI 4; reserved) */
#define HK_FLAG_DEFER_COUNTS 128u       /* hk_rollout: leave the finished-game counts as partial
                                           sums in `workspace` (they accumulate over launches);
                                           hk_rollout_reduce_counts adds them to done_count     */

/* ---- fixed policies fused into hk_rollout (jax/players.py) ----------------------------- */
#define HK_HOST_RANDOM 0    /* players.py:28-39   uniform class id                          */
#define HK_HOST_ALL_COORD 1 /* players.py:42-52   all coordinates                          */
#define HK_HOST_ZEILLINGER 2 /* players.py:84-109                                           */
#define HK_AGENT_RANDOM 0       /* players.py:142-153 uniform over all `dim` axes (JAX)    */
#define HK_AGENT_RANDOM_LEGAL 1 /* trainer/player_modules/modules.py:48-52, agent.py:85-90 */
#define HK_AGENT_CHOOSE_FIRST 2 /* players.py:156-183                                      */
#define HK_AGENT_CHOOSE_LAST 3  /* players.py:186-212                                      */

/* One fused state transition over a batch of games.  A *record* is one game's row of the
 * input/output matrices: `max_points*dim` point coordinates, optionally followed by other
 * columns (the agent observation carries the `dim` coordinate mask there).               */
typedef struct hk_step_desc {
  const void* points_in; /* [batch, in_stride]                                            */
  void* points_out;      /* [batch, out_stride]; may equal points_in if strides are equal */
  int64_t in_stride;     /* elements between consecutive games, >= max_points*dim          */
  int64_t out_stride;    /* only the first max_points*dim elements of a record are written */
  const void* coords;    /* see coords_kind; ignored for IN_RECORD / NONE                  */
  int64_t coords_stride; /* elements between games for mask kinds (>= dim)                 */
  const void* axis;      /* [batch], dtype axis_dtype (HK_I32/I64/F32/F64); NULL w/o shift */
  uint8_t* done_out;      /* [batch] or NULL: (#rows with x_0 >= 0) < 2 after the step     */
  uint8_t* prev_done_out; /* [batch] or NULL: the same before the step                     */
  void* reward_out;       /* [batch] f32 or NULL: reward_sign * (done && !prev_done)       */
  int32_t* num_points_out; /* [batch] or NULL: #rows with x_0 >= 0 after the step          */
  double padding_value;   /* value written into removed rows (negative; -1.0 by default)  */
  float reward_sign;      /* +1 host, -1 agent (jax/util.py:138-144)                       */
  int32_t batch;
  int32_t max_points;
  int32_t dim;
  int32_t dtype;       /* HK_F32 or HK_F64: element type of points_in / points_out        */
  int32_t coords_kind; /* dtype code of a mask, or HK_COORDS_*                             */
  int32_t axis_dtype;
  uint32_t stages; /* HK_STAGE_* bitmask, applied in the order shift, reposition, newton,
                      rescale                                                              */
  uint32_t flags;  /* HK_SEM_* | HK_FLAG_*                                                  */
} hk_step_desc;

/* T fused steps with the host and agent policies evaluated inside the kernel; the state
 * never leaves the chip between steps.  Randomness is Philox4x32-10 keyed by `seed`, with
 * counter (global game index, step index, stream id) -- defined in DESIGN.md so that the
 * CPU oracle reproduces every action bit for bit and a sharded run equals the unsharded
 * one.                                                                                    */
typedef struct hk_rollout_desc {
  void* points;           /* [batch, max_points*dim] final state (and, unless points_in is
                             given, the initial state: updated in place)                   */
  const void* points_in;  /* NULL, or [batch, max_points*dim] initial state, left untouched:
                             an episode restart then costs no device-to-device copy        */
  uint64_t* done_count;   /* [steps+1] or NULL; += #finished games before step 0 and after
                             each step (caller zeroes; accumulates across shards).  Needs
                             `workspace`: per-workgroup partial counts are added there and
                             summed by a second tiny kernel -- 1024 waves hitting one counter
                             with atomics would serialise at ~10 ns each.  With
                             HK_FLAG_DEFER_COUNTS that second kernel is not launched (and
                             done_count may be NULL): call hk_rollout_reduce_counts once after
                             any number of launches that shared the workspace                */
  void* workspace;        /* device memory of >= hk_rollout_workspace_bytes(desc) bytes, or
                             NULL when no counts are wanted.  Must be ZERO before its first
                             use; every reduction leaves it zero again                       */
  uint64_t workspace_bytes;
  void* obs_out;          /* [steps, batch, max_points*dim] or NULL: state before each step */
  int32_t* host_class_out; /* [steps, batch] or NULL: class id chosen by the host          */
  int32_t* axis_out;       /* [steps, batch] or NULL: axis chosen by the agent             */
  uint8_t* done_out;       /* [steps, batch] or NULL: done after each step                 */
  float* reward_out;       /* [steps, batch] or NULL                                       */
  int32_t* game_length_out; /* [batch] or NULL: number of steps after which the game was
                               first done (0 = done at entry, -1 = not within this call)   */
  uint64_t seed;
  uint64_t game_offset; /* global index of this shard's first game                         */
  uint32_t step_offset; /* global index of the first step of this call                     */
  double padding_value;
  float reward_sign;
  int32_t batch;
  int32_t max_points;
  int32_t dim;
  int32_t dtype;
  int32_t steps;
  int32_t host_policy;
  int32_t agent_policy;
  uint32_t stages;
  uint32_t flags;
  const int32_t* game_ids; /* [batch] or NULL: the policy stream of the game at position g is keyed
                              by game_offset + game_ids[g] instead of game_offset + g -- a batch
                              whose games were re-ordered (binned by live rows at generate time so
                              that a wave holds games of one size) rolls out exactly as the
                              original order would, game by game.  The reference's batches carry
                              no order (jax/util.py:385-392 draws them at random).  Non-negative
                              (read as unsigned 32-bit).                                         */
  /* ---- ABI 4: the whole loop of compute_rho (jax_trainer.py:502-555) in one launch.  The reference draws a
   * fresh batch per loop (jax/util.py:385-392) and reads nothing back but a histogram; with gen_max_value > 0 the
   * initial state of game g is what hk_generate_points(max_value = gen_max_value, seed = gen_seed, game_offset + g
   * [or + game_ids[g]], stages = gen_stages, padding_value, the semantics bits of flags) would have written --
   * drawn inside the launch, never stored.  points_in must then be NULL, and `points` MAY be NULL ("counts only":
   * no final state is written either; done_count / game_length_out are the products).                          */
  int32_t gen_max_value; /* 0: the initial state comes from memory (points_in / points)                          */
  uint32_t gen_stages;   /* HK_STAGE_NEWTON | HK_STAGE_REPOSITION | HK_STAGE_RESCALE of the generator            */
  uint64_t gen_seed;
  int32_t episodes;      /* 0 or 1: one episode.  E > 1: E independent episodes back to back inside the launch
                            (a wave starts its next episode when its own games are finished: no launch boundary,
                            nothing waits for the slowest wave) -- episode e starts from the initial state again
                            (points_in, or generated with gen_seed + e) and plays with seed + e; done_count
                            accumulates over the episodes; `points` and game_length_out, if given, receive the LAST
                            episode's; needs points_in or gen_max_value, no per-step records                   */
  int32_t reserved_;     /* 0                                                                                    */
} hk_rollout_desc;

/* ---- library ---------------------------------------------------------------------------- */
int hk_abi_version(void);
const char* hk_strerror(int status);
/* 1 if (max_points, dim, dtype) has a register-resident specialised kernel, else 0.       */
int hk_has_fast_path(int max_points, int dim, int dtype);

/* ---- the fused step --------------------------------------------------------------------- */
int hk_step(const hk_step_desc* desc, void* stream);

/* ---- the reference's individual operators (thin wrappers over the same kernel) -------- */
int hk_shift(const void* points_in, void* points_out, const void* coords, int coords_kind,
             const void* axis, int axis_dtype, int batch, int max_points, int dim, int dtype,
             double padding_value, uint32_t flags, void* stream);
int hk_reposition(const void* points_in, void* points_out, int batch, int max_points, int dim,
                  int dtype, double padding_value, uint32_t flags, void* stream);
int hk_get_newton_polytope(const void* points_in, void* points_out, int batch, int max_points,
                           int dim, int dtype, double padding_value, uint32_t flags,
                           void* stream);
int hk_rescale(const void* points_in, void* points_out, int batch, int max_points, int dim,
               int dtype, double padding_value, uint32_t flags, void* stream);
int hk_get_dones(const void* points, int64_t stride, uint8_t* done_out, int batch,
                 int max_points, int dim, int dtype, void* stream);
int hk_get_num_points(const void* points, int64_t stride, int32_t* num_points_out, int batch,
                      int max_points, int dim, int dtype, void* stream);

/* ---- state generation: randint[0, max_value) -> stages (newton / reposition / rescale) -- */
int hk_generate_points(void* points_out, int batch, int max_points, int dim, int dtype,
                       int max_value, uint64_t seed, uint64_t game_offset, uint32_t stages,
                       double padding_value, uint32_t flags, void* stream);

/* ---- games binned by live rows (ABI 4; no reference counterpart: its batches carry no order, jax/util.py:385-392) ----
 * A wave of the rollout kernels runs the body of its widest game, so neighbours of one size roll out faster; with the
 * permutation as hk_rollout_desc.game_ids every game keeps its policy stream: the re-ordered batch plays, game by game,
 * what the original order plays.  The order is local to groups of G = hk_bin_group_games(...) consecutive games (one
 * workgroup each): inside a group the games are ranked by live rows (x_0 >= 0, hk_get_num_points), widest first, equal
 * ones in their original order.  With U = hk_bin_unit_games(...) (16) and F = batch / G full groups, the game of rank p in
 * full group k goes to position (p / U) * F * U + k * U + p % U: the k-th units of all the groups lie together, the
 * widest stratum first -- the rollout kernels' workgroups start with the heavy waves and every XCD gets every weight.
 * A partial last group is ranked in place, behind the strata.
 * points_out [batch, max_points*dim], game_ids_out [batch] (position -> index in the batch), num_points_out [batch] or NULL
 * (live rows of the game at each position).  float32 and the shapes with a four-lane kernel; else HK_ERR_UNSUPPORTED.
 * hk_generate_points_binned = hk_generate_points with that order applied before the states are stored (one launch: the
 * generator has every game's rows in registers); hk_bin_by_live_rows re-orders an existing batch (NOT in place: a
 * workgroup's games leave for every stratum).                                                                        */
int hk_bin_group_games(int max_points, int dim, int dtype); /* 0: no binning kernel for the shape */
int hk_bin_unit_games(int max_points, int dim, int dtype);
int hk_generate_points_binned(void* points_out, int32_t* game_ids_out, int32_t* num_points_out, int batch, int max_points,
                              int dim, int dtype, int max_value, uint64_t seed, uint64_t game_offset, uint32_t stages,
                              double padding_value, uint32_t flags, void* stream);
int hk_bin_by_live_rows(const void* points_in, void* points_out, int32_t* game_ids_out, int32_t* num_points_out, int batch,
                        int max_points, int dim, int dtype, void* stream);

/* ---- fused T-step rollout with in-kernel fixed policies -------------------------------- */
int hk_rollout(const hk_rollout_desc* desc, void* stream);
/* bytes of `workspace` hk_rollout needs for this descriptor (0 if the descriptor is invalid);
 * depends only on batch, steps, max_points, dim, dtype, flags.                               */
uint64_t hk_rollout_workspace_bytes(const hk_rollout_desc* desc);
/* done_count[0..steps] += the partial counts that hk_rollout launches with HK_FLAG_DEFER_COUNTS
 * left in desc->workspace (same batch / steps / max_points / dim / dtype / flags as those
 * launches); the workspace is zero afterwards.  The counterpart of summing the per-loop
 * histograms on the host (jax_trainer.py:513,533-534), once instead of after every rollout. */
int hk_rollout_reduce_counts(const hk_rollout_desc* desc, void* stream);

/* ---- tree operations of the batched Gumbel-MuZero search --------------------
 * Replaces the calls into the third-party `mctx` package at hironaka/jax/simulation_fn.py:85-117
 * (`mctx.gumbel_muzero_policy`, qtransform_completed_by_mix_value(use_mixed_value=True)); the arrays
 * are mctx's `Tree` fields with the same names, batch-first, which simulation_fn.py:176-186 reads back
 * (node_values[:, 0], children_index[:, 0, :], embeddings).  Embeddings stay with the caller.
 * num_nodes = num_simulations + 1; node 0 is the root; children_index == -1 means unvisited.
 * One search = initialise the arrays (root statistics at node 0, node_visits[:, 0] = 1), then
 * num_simulations x { hk_search_select -> caller runs recurrent_fn on (parent embedding, action) ->
 * hk_search_backup }, then hk_search_policy.                                                      */
typedef struct hk_search_tree {
  int32_t* node_visits;         /* [B, N]    */
  float* raw_values;            /* [B, N]    */
  float* node_values;           /* [B, N]    */
  int32_t* parents;             /* [B, N]    */
  int32_t* action_from_parent;  /* [B, N]    */
  int32_t* children_index;      /* [B, N, A] */
  float* children_prior_logits; /* [B, N, A] */
  int32_t* children_visits;     /* [B, N, A] */
  float* children_rewards;      /* [B, N, A] */
  float* children_discounts;    /* [B, N, A] */
  float* children_values;       /* [B, N, A] */
  int32_t batch, num_nodes, num_actions; /* num_actions <= 32 */
} hk_search_tree;

/* One simulation's descent (root: Gumbel + sequential halving; below: argmax(pi' - N/(1+sum N))):
 * parent_out/action_out [B] = the edge to expand, node_out [B] = the node index the expansion writes
 * (`next_free_node` for an unvisited edge, the existing child when max_depth stopped the descent).
 * root_gumbel [B, A] = gumbel_scale * Gumbel(0,1) noise (caller draws it); root_invalid [B, A] u8 or
 * NULL; considered_visits [max_num_considered_actions + 1, num_simulations] i32 = the sequential
 * halving schedule (hironaka_amd.search.get_table_of_considered_visits).                          */
int hk_search_select(const hk_search_tree* tree, const float* root_gumbel, const uint8_t* root_invalid,
                     const int32_t* considered_visits, int max_num_considered_actions, int num_simulations,
                     int max_depth, int next_free_node, int32_t* parent_out, int32_t* action_out,
                     int32_t* node_out, void* stream);
/* Expansion + backward pass: writes the new node's statistics (prior_logits [B, A], value [B]) and
 * its edge (reward, discount [B]) and updates values / visit counts up to the root.              */
int hk_search_backup(const hk_search_tree* tree, const int32_t* parent, const int32_t* action,
                     const int32_t* node, const float* prior_logits, const float* value,
                     const float* reward, const float* discount, void* stream);
/* After the last simulation: action_out [B] (the Gumbel argmax among the most visited root actions)
 * and action_weights_out [B, A] = softmax(root logits + completed Q-values).                      */
int hk_search_policy(const hk_search_tree* tree, const float* root_gumbel, const uint8_t* root_invalid,
                     int32_t* action_out, float* action_weights_out, void* stream);

/* ---- expansion glue of a HOST-role tree (hironaka/jax/recurrent_fn.py:84-104: host class id -> subset, the
 * agent observation, the opponent's masked argmax, the step, the new embedding) ---------------------------
 * The caller keeps two tables per tree: embeddings [B, N, E] (the points of every node, E = max_points*dim,
 * float32) and features [B, N, E] (hk_get_features of those points, written once when a node is created).
 * One expansion = hk_search_expand_gather -> agent network on agent_feat_out -> hk_search_masked_argmax ->
 * hk_step (class-id coords, int32 axis) -> hk_get_features -> hk_search_expand_scatter -> host network.
 * node_major != 0: the two tables are laid out [N, B, E] instead -- simulation s creates node s + 1 for every game
 * that expands an unvisited edge, so hk_step / hk_get_features can write the new rows straight into slice s + 1 of
 * the tables (a game whose descent stopped on an existing child re-derives that child's rows, and its slot s + 1 is
 * never referenced) and hk_search_expand_scatter is not needed.                                               */
/* obs_out [B, E] = embeddings[b, parent[b]]; agent_feat_out [B, E + dim] = features[b, parent[b]] followed by
 * the 0/1 subset of the host's class id action[b] (clamped into range like hk_decode_host_class)           */
int hk_search_expand_gather(const void* embeddings, const void* features, const int32_t* parent,
                            const int32_t* action, void* obs_out, void* agent_feat_out, int batch,
                            int num_nodes, int max_points, int dim, int node_major, void* stream);
/* axis_out[b] = argmax_k of logits[b, k] over the coordinates k of the subset of class id action[b]
 * (jax/util.py:287-327: the agent's action mask + argmax; first maximum, NaN beats every number)       */
int hk_search_masked_argmax(const void* logits, const int32_t* action, int32_t* axis_out, int batch, int dim,
                            void* stream);
/* embeddings[b, node[b]] = obs[b]; features[b, node[b]] = feat[b]   (obs, feat: [B, E] float32)          */
int hk_search_expand_scatter(const void* obs, const void* feat, const int32_t* node, void* embeddings,
                             void* features, int batch, int num_nodes, int max_points, int dim, void* stream);

/* The same for an AGENT-role tree (recurrent_fn.py:105-121): embeddings [B, N, E + dim] hold the agent observation
 * (points followed by the host's 0/1 subset), features [B, N, E] the features of the points.  One expansion =
 * hk_search_expand_gather_agent -> hk_step (float mask, the agent's axis) -> hk_get_features -> host network ->
 * hk_search_expand_scatter_agent -> agent network -> hk_search_mask_logits.                                   */
/* points_out [B, E], coords_out [B, dim] = the two parts of embeddings[b, parent[b]]                          */
int hk_search_expand_gather_agent(const void* embeddings, const int32_t* parent, void* points_out, void* coords_out,
                                  int batch, int num_nodes, int max_points, int dim, void* stream);
/* class = argmax_c host_logits[b, c] (first maximum, NaN beats every number; num_classes <= 2^dim - dim - 1),
 * mask = its subset; embeddings[b, node[b]] = points[b] ++ mask; features[b, node[b]] = feat[b] (features may be
 * NULL: nothing in an agent-role expansion reads a node's features again -- the agent network's input is agent_feat_out);
 * agent_feat_out [B, E + dim] = feat[b] ++ mask; class_out [B] (or NULL) = class                              */
int hk_search_expand_scatter_agent(const void* points, const void* feat, const void* host_logits,
                                   const int32_t* node, void* embeddings, void* features, void* agent_feat_out,
                                   int32_t* class_out, int batch, int num_nodes, int max_points, int dim,
                                   int num_classes, void* stream);
/* out[b, k] = logits[b, k] if coordinate k belongs to the subset of class_id[b], else -inf (the agent's action
 * mask, jax/util.py:287-305 in its NaN-free form); out may equal logits                                       */
int hk_search_mask_logits(const void* logits, const int32_t* class_id, void* out, int batch, int dim, void* stream);

/* hk_step with the observation features of its RESULT (hk_get_features of points_out) as a second output of the
 * same launch: features_out [batch, max_points*dim] contiguous float32.  The search's expansion (recurrent_fn.py:84-104
 * followed by the policy network's feature function, jax/util.py:172-214).  Four-lane step kernel only: float32,
 * contiguous records of (10,3) (20,3) (20,4), HK_COORDS_CLASS_I32 with an int32 axis or HK_AXIS_MASKED_LOGITS (JAX or
 * torch semantics), or a float32 mask with an int32 axis in the JAX trainer's configuration (shift + reposition + Newton); anything else returns HK_ERR_UNSUPPORTED (call hk_step and hk_get_features instead).          */
int hk_step_features(const hk_step_desc* desc, void* features_out, int scale_observation, void* stream);

/* ---- value targets of a self-play rollout: JAXTrainer.rollout_postprocess (jax_trainer.py:558-592) with
 * calculate_value_using_reward_fn (jax/util.py:261-284) ----------------------------------------------------------
 * obs [B, T, obs_dim] float32 (the observation before each of the T moves; T <= 64), value_out [B, T] float32.
 * num_points = #(entries >= 0) / dim - points_offset (1 when the observation carries the subset tail: agent role /
 * unified tree); discount is signed (the unified tree passes -discount); reward_sign = +1 host, -1 agent / unified;
 * estimate_scale = the sign of the "1 / remaining points" estimate of an unfinished game (+1 host, -1 agent,
 * times (-1)^(T+1) on a unified tree).                                                                            */
int hk_rollout_values(const void* obs, void* value_out, int batch, int steps, int obs_dim, int dim,
                      int points_offset, float discount, float reward_sign, float estimate_scale, void* stream);

/* ---- fixed host policy as its own operator: class id per game -------------------------- */
/* flags: HK_SEM_JAX (default; all ordered pairs, isclose-degenerate pairs skipped, degenerate
 * game -> class 0) or HK_SEM_LIST (host.py:70-95: pairs i<j of the available rows in row order,
 * no isclose; a game with fewer than 2 available rows -> -1 = "no subset")                    */
int hk_zeillinger(const void* points, int64_t stride, int32_t* class_out, int batch,
                  int max_points, int dim, int dtype, uint32_t flags, void* stream);

/* ---- observation transform: [rescale] + rows sorted descending, last coordinate primary */
int hk_get_features(const void* points_in, int64_t in_stride, void* features_out,
                    int64_t out_stride, int batch, int max_points, int dim, int dtype,
                    int scale_observation, double padding_value, void* stream);

/* ---- the torch container's observation (core/tensor_points.py:72-74): rows ordered by coordinate 0,
 * descending -- unavailable rows last -- values untouched.  The reference leaves the order among equal
 * coordinates 0 to torch.argsort; here such rows keep their order (stable).                     */
int hk_get_features_torch(const void* points_in, int64_t in_stride, void* features_out,
                          int64_t out_stride, int batch, int max_points, int dim, int dtype,
                          double padding_value, void* stream);

/* ---- host action codec ------------------------------------------------------------------ */
int hk_decode_host_class(const int32_t* class_in, void* mask_out, int mask_dtype, int batch,
                         int dim, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HIRONAKA_HIP_H */
