"""ctypes mirror of ``include/hironaka_hip.h`` (constants, descriptors, prototypes).

Kept free of torch so the CPU-side tests (and the oracle binding, which takes the same
descriptors with host pointers) can import it without a GPU.
"""
import ctypes as C

HK_ABI_VERSION = 4

# status codes
HK_OK = 0
HK_ERR_NULL = -1
HK_ERR_SHAPE = -2
HK_ERR_UNSUPPORTED = -3
HK_ERR_ALIGN = -4
HK_ERR_LAUNCH = -5
HK_ERR_NO_DEVICE = -6

# scalar dtypes
HK_F32, HK_F64, HK_I32, HK_I64, HK_U8 = 0, 1, 2, 3, 4

# coordinate-subset kinds (a mask's dtype code, or one of these)
HK_COORDS_CLASS_I32 = 16
HK_COORDS_CLASS_I64 = 17
HK_COORDS_IN_RECORD = 18
HK_AXIS_MASKED_LOGITS = 32
HK_COORDS_NONE = 19

# stages
HK_STAGE_SHIFT = 1
HK_STAGE_REPOSITION = 2
HK_STAGE_NEWTON = 4
HK_STAGE_RESCALE = 8

# semantics + behaviour flags
HK_SEM_JAX, HK_SEM_TORCH, HK_SEM_LIST, HK_SEM_MASK = 0, 1, 2, 3
HK_FLAG_AXIS_NOOP_IF_INVALID = 4
HK_FLAG_IGNORE_ENDED = 8
HK_FLAG_COMPACT_SORTED = 16
HK_FLAG_FORCE_GENERIC = 32
HK_FLAG_FORCE_TEAM = 64
HK_FLAG_DEFER_COUNTS = 128
HK_FLAG_FORCE_ONE_LANE = 256
HK_FLAG_FORCE_TWO_LANES = 512
HK_FLAG_FORCE_FOUR_LANES = 1024

# fused policies
HK_HOST_RANDOM, HK_HOST_ALL_COORD, HK_HOST_ZEILLINGER = 0, 1, 2
HK_AGENT_RANDOM, HK_AGENT_RANDOM_LEGAL, HK_AGENT_CHOOSE_FIRST, HK_AGENT_CHOOSE_LAST = 0, 1, 2, 3

SEMANTICS = {"jax": HK_SEM_JAX, "torch": HK_SEM_TORCH, "list": HK_SEM_LIST}


class hk_step_desc(C.Structure):
    _fields_ = [
        ("points_in", C.c_void_p),
        ("points_out", C.c_void_p),
        ("in_stride", C.c_int64),
        ("out_stride", C.c_int64),
        ("coords", C.c_void_p),
        ("coords_stride", C.c_int64),
        ("axis", C.c_void_p),
        ("done_out", C.c_void_p),
        ("prev_done_out", C.c_void_p),
        ("reward_out", C.c_void_p),
        ("num_points_out", C.c_void_p),
        ("padding_value", C.c_double),
        ("reward_sign", C.c_float),
        ("batch", C.c_int32),
        ("max_points", C.c_int32),
        ("dim", C.c_int32),
        ("dtype", C.c_int32),
        ("coords_kind", C.c_int32),
        ("axis_dtype", C.c_int32),
        ("stages", C.c_uint32),
        ("flags", C.c_uint32),
    ]


class hk_rollout_desc(C.Structure):
    _fields_ = [
        ("points", C.c_void_p),
        ("points_in", C.c_void_p),
        ("done_count", C.c_void_p),
        ("workspace", C.c_void_p),
        ("workspace_bytes", C.c_uint64),
        ("obs_out", C.c_void_p),
        ("host_class_out", C.c_void_p),
        ("axis_out", C.c_void_p),
        ("done_out", C.c_void_p),
        ("reward_out", C.c_void_p),
        ("game_length_out", C.c_void_p),
        ("seed", C.c_uint64),
        ("game_offset", C.c_uint64),
        ("step_offset", C.c_uint32),
        ("padding_value", C.c_double),
        ("reward_sign", C.c_float),
        ("batch", C.c_int32),
        ("max_points", C.c_int32),
        ("dim", C.c_int32),
        ("dtype", C.c_int32),
        ("steps", C.c_int32),
        ("host_policy", C.c_int32),
        ("agent_policy", C.c_int32),
        ("stages", C.c_uint32),
        ("flags", C.c_uint32),
        ("game_ids", C.c_void_p),
        ("gen_max_value", C.c_int32),
        ("gen_stages", C.c_uint32),
        ("gen_seed", C.c_uint64),
        ("episodes", C.c_int32),
        ("reserved_", C.c_int32),
    ]


class hk_search_tree(C.Structure):
    _fields_ = [
        ("node_visits", C.c_void_p),
        ("raw_values", C.c_void_p),
        ("node_values", C.c_void_p),
        ("parents", C.c_void_p),
        ("action_from_parent", C.c_void_p),
        ("children_index", C.c_void_p),
        ("children_prior_logits", C.c_void_p),
        ("children_visits", C.c_void_p),
        ("children_rewards", C.c_void_p),
        ("children_discounts", C.c_void_p),
        ("children_values", C.c_void_p),
        ("batch", C.c_int32),
        ("num_nodes", C.c_int32),
        ("num_actions", C.c_int32),
    ]


_vp, _i, _i64, _u32, _u64, _d = C.c_void_p, C.c_int, C.c_int64, C.c_uint32, C.c_uint64, C.c_double

# name -> (restype, argtypes) of every symbol the header declares.  `stream` is the trailing
# void* of the device entry points; the oracle exports the same list with prefix hko_ and
# without the stream argument.
PROTOTYPES = {
    "hk_abi_version": (C.c_int, []),
    "hk_strerror": (C.c_char_p, [_i]),
    "hk_has_fast_path": (C.c_int, [_i, _i, _i]),
    "hk_step": (C.c_int, [C.POINTER(hk_step_desc), _vp]),
    "hk_shift": (C.c_int, [_vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _d, _u32, _vp]),
    "hk_reposition": (C.c_int, [_vp, _vp, _i, _i, _i, _i, _d, _u32, _vp]),
    "hk_get_newton_polytope": (C.c_int, [_vp, _vp, _i, _i, _i, _i, _d, _u32, _vp]),
    "hk_rescale": (C.c_int, [_vp, _vp, _i, _i, _i, _i, _d, _u32, _vp]),
    "hk_get_dones": (C.c_int, [_vp, _i64, _vp, _i, _i, _i, _i, _vp]),
    "hk_get_num_points": (C.c_int, [_vp, _i64, _vp, _i, _i, _i, _i, _vp]),
    "hk_generate_points": (C.c_int, [_vp, _i, _i, _i, _i, _i, _u64, _u64, _u32, _d, _u32, _vp]),
    "hk_bin_group_games": (C.c_int, [_i, _i, _i]),
    "hk_bin_unit_games": (C.c_int, [_i, _i, _i]),
    "hk_generate_points_binned": (C.c_int, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _u64, _u64, _u32, _d, _u32, _vp]),
    "hk_bin_by_live_rows": (C.c_int, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "hk_rollout": (C.c_int, [C.POINTER(hk_rollout_desc), _vp]),
    "hk_rollout_workspace_bytes": (C.c_uint64, [C.POINTER(hk_rollout_desc)]),
    "hk_rollout_reduce_counts": (C.c_int, [C.POINTER(hk_rollout_desc), C.c_void_p]),
    "hk_search_select": (C.c_int, [C.POINTER(hk_search_tree), _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "hk_search_backup": (C.c_int, [C.POINTER(hk_search_tree), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "hk_search_policy": (C.c_int, [C.POINTER(hk_search_tree), _vp, _vp, _vp, _vp, _vp]),
    "hk_zeillinger": (C.c_int, [_vp, _i64, _vp, _i, _i, _i, _i, _u32, _vp]),
    "hk_get_features": (C.c_int, [_vp, _i64, _vp, _i64, _i, _i, _i, _i, _i, _d, _vp]),
    "hk_get_features_torch": (C.c_int, [_vp, _i64, _vp, _i64, _i, _i, _i, _i, _d, _vp]),
    "hk_decode_host_class": (C.c_int, [_vp, _vp, _i, _i, _i, _vp]),
    "hk_step_features": (C.c_int, [C.POINTER(hk_step_desc), _vp, _i, _vp]),
    "hk_rollout_values": (C.c_int, [_vp, _vp, _i, _i, _i, _i, _i, C.c_float, C.c_float, C.c_float, _vp]),
    "hk_search_expand_gather": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "hk_search_masked_argmax": (C.c_int, [_vp, _vp, _vp, _i, _i, _vp]),
    "hk_search_expand_scatter": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "hk_search_expand_gather_agent": (C.c_int, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "hk_search_expand_scatter_agent": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "hk_search_mask_logits": (C.c_int, [_vp, _vp, _vp, _i, _i, _vp]),
}

STATUS_TEXT = {
    HK_OK: "ok",
    HK_ERR_NULL: "a required pointer is NULL",
    HK_ERR_SHAPE: "batch / max_points / dim / stride / class id out of range",
    HK_ERR_UNSUPPORTED: "dtype, kind or flag combination not supported",
    HK_ERR_ALIGN: "pointer not aligned to its element size",
    HK_ERR_LAUNCH: "HIP kernel launch failed",
    HK_ERR_NO_DEVICE: "no HIP device",
}


def bind(lib: C.CDLL, prototypes=PROTOTYPES) -> None:
    """Attach restype/argtypes; raises AttributeError for a symbol the library lacks."""
    for name, (res, args) in prototypes.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
