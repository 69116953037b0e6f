"""The environment half of the MCTS recurrent function -- the counterpart of
``hironaka/jax/recurrent_fn.py`` with the same factory names, arguments and order of operations
around the step (recurrent_fn.py:84-121, 161-197):

    prev_dones -> (agent role: step first) -> opponent acts -> (host role: step) -> dones -> reward
    -> role_fn(next_observations)

The search driver that calls it in the reference (third-party ``mctx``) is out of scope; any search
loop over torch tensors can call these functions.  ``mctx.RecurrentFnOutput`` is replaced by a
NamedTuple with the same fields.
"""
from __future__ import annotations

from typing import Callable, NamedTuple, Tuple

import torch

from . import _abi as A
from ._lib import HironakaHipError
from .functional import flatten, get_dones, get_preprocess_fns, get_take_actions, make_agent_obs
from .host_action_preprocess import get_batch_decode, get_batch_decode_from_one_hot, num_classes


class RecurrentFnOutput(NamedTuple):
    reward: torch.Tensor
    discount: torch.Tensor
    prior_logits: torch.Tensor
    value: torch.Tensor


class HostExpander:
    """One expansion of a HOST-role tree as fused device work (recurrent_fn.py:84-104 around the two networks):

        hk_search_expand_gather  (parent points + parent features ++ subset of the host's class id)
        -> agent network -> hk_search_masked_argmax -> hk_step (class ids, int32 axis; writes the new node's points)
        -> hk_get_features (writes the new node's features) -> host network

    instead of gather / decode / concat / feature sort of the strided agent observation / concat / mask fill /
    compare / where / argmax / step / feature sort / index_put.  It needs the two networks as
    ``model(features, params) -> (logits, value)`` behind the STANDARD feature functions (functional.get_feature_fn:
    the observation features of a node are computed once, when the node is created, and kept in a second table
    next to the embeddings), so it is what ``HipTrainer`` builds for its own policies; arbitrary callables keep
    the generic path below.  Results are identical to the generic path (tests/test_gpu_trainer.py)."""

    def __init__(self, host_model: Callable, agent_model: Callable, spec: Tuple[int, int], discount: float,
                 scale_observation: bool, reposition: bool, rescale_points: bool = False, reward_sign: float = 1.0):
        from . import ops
        self.host_model, self.agent_model = host_model, agent_model
        self.m, self.d = spec
        self.discount = discount
        self.scale_observation = scale_observation
        self.reward_sign = reward_sign
        self.stages = ops.make_stages(shift=True, reposition=reposition, newton=True, rescale=rescale_points)
        self.logits_in_step = None  # unknown until the first expansion
        self.features_in_step = None

    def accepts(self, root_embedding: torch.Tensor) -> bool:
        return (root_embedding.is_cuda and root_embedding.dtype == torch.float32 and root_embedding.dim() == 2
                and root_embedding.shape[1] == self.m * self.d)

    def begin(self, tree, root_embedding: torch.Tensor):
        """per search: the two node-major tables [N, B, E] (points and features; root rows filled) and the scratch
        arrays of an expansion.  Simulation s creates node s + 1, so the step and the feature transform write the new
        rows straight into slice s + 1 -- no scatter.  (A game whose descent stopped on an existing child, at the depth
        limit, re-derives that child's rows: identical values, and its slot s + 1 is never referenced.)
        `state["tree"]` is the tree with `embeddings` replaced by the [B, N, E] view of the points table."""
        from . import ops
        b, n, e = tree.embeddings.shape
        dev = tree.embeddings.device
        points = torch.zeros((n, b, e), dtype=torch.float32, device=dev)
        features = torch.zeros((n, b, e), dtype=torch.float32, device=dev)
        points[0] = root_embedding
        ops.get_features(root_embedding.contiguous(), self.scale_observation, spec=(self.m, self.d), out=features[0])
        return {"tree": tree._replace(embeddings=points.permute(1, 0, 2)), "points": points, "features": features,
                "sim": 0,
                "obs": torch.empty((b, e), dtype=torch.float32, device=dev),
                "agent_feat": torch.empty((b, e + self.d), dtype=torch.float32, device=dev),
                "axis": torch.empty(b, dtype=torch.int32, device=dev),
                "discount": torch.full((b,), self.discount, dtype=torch.float32, device=dev)}

    def expand(self, params, key, tree, state, parent: torch.Tensor, action: torch.Tensor, node: torch.Tensor):
        import ctypes as C

        from . import ops
        from ._lib import check, lib
        (host_params, *_), (agent_params, *_) = params
        points, features = state["points"], state["features"]
        n, b, e = points.shape
        m, d = self.m, self.d
        L = lib()
        stream = C.c_void_p(torch.cuda.current_stream(points.device).cuda_stream)
        obs, agent_feat, axis = state["obs"], state["agent_feat"], state["axis"]
        slot = state["sim"] + 1  # the node this simulation creates (hk_search_select's next_free_node)
        state["sim"] += 1
        check(L.hk_search_expand_gather(points.data_ptr(), features.data_ptr(), parent.data_ptr(), action.data_ptr(),
                                        obs.data_ptr(), agent_feat.data_ptr(), b, n, m, d, 1, stream),
              "hk_search_expand_gather")
        logits, _ = self.agent_model(agent_feat, agent_params)
        logits = logits.to(torch.float32).contiguous()
        want = ("done", "prev_done", "reward")
        if self.features_in_step is not False:
            # everything between the two networks in ONE launch: the agent's masked argmax, the step, the features of
            # the new node (hk_step_features); found out once, on the first expansion (eager, before any capture)
            try:
                res = ops.step(obs, action, logits, stages=self.stages, spec=(m, d), out=points[slot], want=want,
                               reward_sign=self.reward_sign, features_out=features[slot],
                               scale_observation=self.scale_observation)
                self.features_in_step = True
                prior, value = self.host_model(features[slot], host_params)
                return RecurrentFnOutput(reward=res["reward"], discount=state["discount"], prior_logits=prior,
                                         value=value.reshape(b))
            except HironakaHipError as err:
                if self.features_in_step or err.status != A.HK_ERR_UNSUPPORTED:
                    raise
                self.features_in_step = False
        if self.logits_in_step is not False:
            # the agent's masked argmax inside the step's action decode (HK_AXIS_MASKED_LOGITS: shapes with a four-lane
            # step kernel); found out once, on the first expansion (eager, before any graph capture)
            try:
                res = ops.step(obs, action, logits, stages=self.stages, spec=(m, d), out=points[slot], want=want,
                               reward_sign=self.reward_sign)
                self.logits_in_step = True
            except HironakaHipError as err:
                if self.logits_in_step or err.status != A.HK_ERR_UNSUPPORTED:
                    raise
                self.logits_in_step = False
        if self.logits_in_step is False:
            check(L.hk_search_masked_argmax(logits.data_ptr(), action.data_ptr(), axis.data_ptr(), b, d, stream),
                  "hk_search_masked_argmax")
            res = ops.step(obs, action, axis, stages=self.stages, spec=(m, d), out=points[slot], want=want,
                           reward_sign=self.reward_sign)
        host_feat = ops.get_features(points[slot], self.scale_observation, spec=(m, d), out=features[slot])
        prior, value = self.host_model(host_feat, host_params)
        return RecurrentFnOutput(reward=res["reward"], discount=state["discount"], prior_logits=prior,
                                 value=value.reshape(b))


class AgentExpander:
    """The same for an AGENT-role tree (recurrent_fn.py:105-121): embeddings are agent observations (points ++ the
    host's subset); the agent's axis finishes the move, the host network answers on the new points, the agent
    network (behind its action mask) evaluates the new observation:

        hk_search_expand_gather_agent -> hk_step (float mask, int32 axis) -> hk_get_features -> host network
        -> hk_search_expand_scatter_agent (host argmax, subset, new node, agent input) -> agent network
        -> hk_search_mask_logits"""

    def __init__(self, host_model: Callable, agent_model: Callable, spec: Tuple[int, int], discount: float,
                 scale_observation: bool, reposition: bool, rescale_points: bool = False, reward_sign: float = -1.0):
        from . import ops
        self.host_model, self.agent_model = host_model, agent_model
        self.m, self.d = spec
        self.discount = discount
        self.scale_observation = scale_observation
        self.reward_sign = reward_sign
        self.stages = ops.make_stages(shift=True, reposition=reposition, newton=True, rescale=rescale_points)
        self.features_in_step = None  # unknown until the first expansion

    def accepts(self, root_embedding: torch.Tensor) -> bool:
        return (root_embedding.is_cuda and root_embedding.dtype == torch.float32 and root_embedding.dim() == 2
                and root_embedding.shape[1] == (self.m + 1) * self.d)

    def begin(self, tree, root_embedding: torch.Tensor):
        from . import ops
        b, n, width = tree.embeddings.shape
        e, d = self.m * self.d, self.d
        dev = tree.embeddings.device
        # (no per-node features table here: an agent-role expansion never reads a node's features again -- the gather
        # takes the embeddings only and the agent network's input is built from `feat` -- so the scatter gets NULL)
        return {"points": torch.empty((b, e), dtype=torch.float32, device=dev),
                "coords": torch.empty((b, d), dtype=torch.float32, device=dev),
                "agent_feat": torch.empty((b, e + d), dtype=torch.float32, device=dev),
                "cls": torch.empty(b, dtype=torch.int32, device=dev),
                "feat": torch.empty((b, e), dtype=torch.float32, device=dev),
                "discount": torch.full((b,), self.discount, dtype=torch.float32, device=dev)}

    def expand(self, params, key, tree, state, parent: torch.Tensor, action: torch.Tensor, node: torch.Tensor):
        import ctypes as C

        from . import ops
        from ._lib import check, lib
        (agent_params, *_), (host_params, *_) = params
        b, n, _ = tree.embeddings.shape
        m, d = self.m, self.d
        e = m * d
        L = lib()
        stream = C.c_void_p(torch.cuda.current_stream(tree.embeddings.device).cuda_stream)
        points, coords, agent_feat, cls = state["points"], state["coords"], state["agent_feat"], state["cls"]
        check(L.hk_search_expand_gather_agent(tree.embeddings.data_ptr(), parent.data_ptr(), points.data_ptr(),
                                              coords.data_ptr(), b, n, m, d, stream), "hk_search_expand_gather_agent")
        want = ("done", "prev_done", "reward")
        feat = None
        if self.features_in_step is not False:
            # the step and the features of its result in one launch (hk_step_features), where the shape and the stage
            # mask allow; found out once, on the first expansion (eager, before any capture)
            try:
                feat = state["feat"]
                res = ops.step(points, coords, action, stages=self.stages, spec=(m, d), want=want,
                               reward_sign=self.reward_sign, features_out=feat,
                               scale_observation=self.scale_observation)
                self.features_in_step = True
            except HironakaHipError as err:
                if self.features_in_step or err.status != A.HK_ERR_UNSUPPORTED:
                    raise
                self.features_in_step = False
        if self.features_in_step is False:
            res = ops.step(points, coords, action, stages=self.stages, spec=(m, d), want=want,
                           reward_sign=self.reward_sign)
        updated = res["points"].reshape(b, e)
        if self.features_in_step is False:
            feat = ops.get_features(updated, self.scale_observation, spec=(m, d))
        host_logits, _ = self.host_model(feat, host_params)
        host_logits = host_logits.to(torch.float32).contiguous()
        check(L.hk_search_expand_scatter_agent(updated.data_ptr(), feat.data_ptr(), host_logits.data_ptr(),
                                               node.data_ptr(), tree.embeddings.data_ptr(), None,
                                               agent_feat.data_ptr(), cls.data_ptr(), b, n, m, d,
                                               host_logits.shape[1], stream), "hk_search_expand_scatter_agent")
        logits, value = self.agent_model(agent_feat, agent_params)
        logits = logits.to(torch.float32).contiguous()
        prior = torch.empty_like(logits)
        check(L.hk_search_mask_logits(logits.data_ptr(), cls.data_ptr(), prior.data_ptr(), b, d, stream),
              "hk_search_mask_logits")
        return RecurrentFnOutput(reward=res["reward"], discount=state["discount"], prior_logits=prior,
                                 value=value.reshape(b))


def get_recurrent_fn_for_role(role: str, role_fn: Callable, opponent_action_fn: Callable, reward_fn: Callable,
                              spec: Tuple[int, int], discount: float = 0.99, dtype=torch.float32,
                              rescale_points: bool = False, reposition: bool = False,
                              expander: "HostExpander | AgentExpander | None" = None) -> Callable:
    """recurrent_fn.py:17-123.
    role_fn(observations, *args, key=...) -> (policy_prior, value_prior) of the player under evaluation;
    opponent_action_fn(observations, *args, key=...) -> one-hot actions of the fixed opponent;
    reward_fn(dones, prev_dones) -> rewards."""
    m, d = spec
    if role == "host":
        take = get_take_actions(role="host", spec=spec, rescale_points=rescale_points, reposition=reposition)
        batch_decode = get_batch_decode(d)
    elif role == "agent":
        take = get_take_actions(role="agent", spec=spec, rescale_points=rescale_points, reposition=reposition)
        decode_one_hot = get_batch_decode_from_one_hot(d)
        batch_decode_cls = get_batch_decode(d)
    else:
        raise ValueError(f"role must be either 'host' or 'agent'. Got {role}.")

    # the step launch itself reports done-before / done-after (and the reward when reward_fn is one of
    # get_reward_fn's): no separate get_dones / reward kernels inside the search loop
    sign = getattr(reward_fn, "hk_reward_sign", None)
    want = ("done", "prev_done") + (("reward",) if sign is not None else ())
    discounts = {}

    def discount_like(observations: torch.Tensor) -> torch.Tensor:
        key_ = (observations.shape[0], observations.device)
        if key_ not in discounts:
            discounts[key_] = torch.full((observations.shape[0],), discount, dtype=dtype, device=observations.device)
        return discounts[key_]

    # an opponent built by `action_wrapper` also offers its choice as an index: no one-hot round trip
    opponent_index_fn = getattr(opponent_action_fn, "index_fn", None)

    def recurrent_fn(params, key, actions: torch.Tensor, observations: torch.Tensor):
        role_fn_args, opponent_fn_args = params
        if role == "host":
            # host acts (class ids -> masks), the agent answers, then the step happens
            coords = batch_decode(actions, dtype)
            agent_obs = make_agent_obs(observations, coords).to(dtype)
            if opponent_index_fn is not None:
                axis = opponent_index_fn(agent_obs, *opponent_fn_args, key=key)
            else:
                axis = torch.argmax(opponent_action_fn(agent_obs, *opponent_fn_args, key=key), dim=1)
            res = take(observations, coords, axis, want=want, reward_sign=sign or 1.0)
            next_observations = res["points"].to(dtype)
        else:
            # the agent's axis finishes the move first; the host then answers on the new points
            res = take(observations, None, actions, want=want, reward_sign=sign or 1.0)
            updated = res["points"]
            if opponent_index_fn is not None:
                next_coords = batch_decode_cls(opponent_index_fn(updated.to(dtype), *opponent_fn_args, key=key), dtype)
            else:
                next_coords = decode_one_hot(opponent_action_fn(updated.to(dtype), *opponent_fn_args, key=key), dtype)
            next_observations = make_agent_obs(updated, next_coords).to(dtype)
        rewards = res["reward"] if sign is not None else reward_fn(res["done"], res["prev_done"])
        policy_prior, value_prior = role_fn(next_observations, *role_fn_args, key=key)
        out = RecurrentFnOutput(reward=rewards, discount=discount_like(observations), prior_logits=policy_prior,
                                value=value_prior)
        return out, next_observations

    # (not in the reference) a search loop that knows about it runs the expansion through the fused operators
    if expander is not None and not isinstance(expander, HostExpander if role == "host" else AgentExpander):
        raise TypeError(f"a {role}-role tree cannot expand through {type(expander).__name__}")
    recurrent_fn.expander = expander
    return recurrent_fn


def get_unified_recurrent_fn(host_fn: Callable, agent_fn: Callable, reward_fn: Callable, spec: Tuple[int, int],
                             discount: float = 0.99, dtype=torch.float32, rescale_points: bool = False,
                             reposition: bool = False) -> Callable:
    """recurrent_fn.py:126-199 -- one tree for both players: states are [B, m*d + d]; a host state has
    a zero tail, an agent state carries the subset mask; the discount is negated; agent logits are
    padded with -inf to the host's action count."""
    m, d = spec
    obs_preprocess, coords_preprocess = get_preprocess_fns("agent", spec)
    batch_decode = get_batch_decode(d)
    take = get_take_actions(role="host", spec=spec, rescale_points=rescale_points, reposition=reposition)
    discount = -discount
    extra = num_classes(d) - d

    def recurrent_fn(params, key, actions: torch.Tensor, observations: torch.Tensor):
        host_param, agent_param = params
        batch_size = observations.shape[0]
        obs, coord = obs_preprocess(observations), coords_preprocess(observations, None)
        # all states of a batch are of the same kind (recurrent_fn.py:176-178)
        is_host = bool(torch.isclose(coord, torch.zeros((), device=coord.device, dtype=coord.dtype)).all(dim=-1).any())
        if is_host:
            next_coord = batch_decode(actions, dtype)
            next_obs = flatten(obs)
        else:
            next_coord = torch.zeros((batch_size, d), dtype=dtype, device=observations.device)
            next_obs = take(obs.contiguous(), coord.contiguous(), actions)
        next_state = torch.cat([next_obs.to(dtype), next_coord], dim=-1)
        if is_host:  # the next node is an agent node
            policy_prior, value_prior = agent_fn(next_state, *agent_param, key=key)
            policy_prior = torch.nn.functional.pad(policy_prior, (0, extra), value=float("-inf"))
        else:
            policy_prior, value_prior = host_fn(next_state, *host_param, key=key)
        prev_dones = get_dones(obs)
        dones = get_dones(next_obs.reshape(-1, m, d))
        rewards = reward_fn(dones, prev_dones)
        out = RecurrentFnOutput(reward=rewards,
                                discount=torch.full((batch_size,), discount, dtype=dtype, device=observations.device),
                                prior_logits=policy_prior, value=value_prior)
        return out, next_state

    return recurrent_fn
