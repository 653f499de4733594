"""ctypes binding of the C-ABI declared in ``include/rl4co_amd.h``.

The library is the product: there is NO CPU fallback. Importing this module is cheap;
the first call to :func:`lib` loads ``librl4co_amd.so`` and raises if it is missing.
"""
from __future__ import annotations

import ctypes as C
import os
from functools import lru_cache

from . import build as _build

RL4CO_OK = 0
ABI_VERSION = 12  # RL4CO_ABI_VERSION of include/rl4co_amd.h this binding's argument lists were written for
ENV_TSP, ENV_CVRP, ENV_OP, ENV_PCTSP, ENV_PDP, ENV_CVRPTW = 0, 1, 2, 3, 4, 5
DECODE_GREEDY, DECODE_SAMPLE, DECODE_EVALUATE = 0, 1, 2
DT_F32, DT_BF16, DT_F16 = 0, 1, 2
VARIANT_AUTO, VARIANT_STREAM, VARIANT_LDS, VARIANT_WIDE, VARIANT_MS = 0, 1, 2, 3, 4

EBIT_NAN_LOGIT = 1
EBIT_INFEASIBLE = 2
EBIT_INVALID_TOUR = 4
EBIT_CAPACITY = 8
EBIT_MAX_STEPS = 16
EBIT_NEG_INF_LOGP = 32
EBIT_DUPLICATES = 64
EBIT_MAX_LENGTH = 128
EBIT_PRIZE = 256
EBIT_NOT_ALL_NODES = 512
EBIT_DEPOT_MIDDLE = 1024
EBIT_NO_PICKUP = 2048
EBIT_TW_NEGATIVE, EBIT_TW_RETURN, EBIT_TW_DURATION, EBIT_TW_EMPTY, EBIT_TW_DEADLINE = 4096, 8192, 16384, 32768, 65536

# Reference assertion messages (file:line in the reference checkout) per sticky bit.
ERROR_MESSAGES = {
    EBIT_NAN_LOGIT: "Logits contain NaNs",  # nn/attention.py:296
    EBIT_INFEASIBLE: "infeasible action selected",  # utils/decoding.py:393,409
    EBIT_INVALID_TOUR: "Invalid tour",  # tsp/env.py:164, cvrp/env.py:163
    EBIT_CAPACITY: "Used more than capacity",  # cvrp/env.py:176
    EBIT_MAX_STEPS: "Exceeded maximum number of steps during decoding",  # constructive/base.py:237
    EBIT_NEG_INF_LOGP: "Logprobs should not be -inf, check sampling procedure!",  # decoding.py:56
    EBIT_DUPLICATES: "Duplicates",  # op/env.py:181
    EBIT_MAX_LENGTH: "Max length exceeded",  # op/env.py:192-194
    EBIT_PRIZE: "Total prize does not satisfy min total prize",  # pctsp/env.py:192-201
    EBIT_NOT_ALL_NODES: "Not visiting all nodes",  # pdp/env.py:208-213
    EBIT_DEPOT_MIDDLE: "Going back to depot in the middle of the tour (not allowed)",  # pdp/env.py:216-218
    EBIT_NO_PICKUP: "Deliverying without pick-up",  # pdp/env.py:220-223
    EBIT_TW_NEGATIVE: "Time windows must be non-negative.",  # cvrptw/env.py:153
    EBIT_TW_RETURN: "vehicle cannot perform service and get back to depot in time.",  # cvrptw/env.py:154-157
    EBIT_TW_DURATION: "Service durations must be non-negative.",  # cvrptw/env.py:158
    EBIT_TW_EMPTY: "there are unfeasible time windows",  # cvrptw/env.py:159-161
    EBIT_TW_DEADLINE: "vehicle cannot start service before deadline",  # cvrptw/env.py:176-179
}

_vp, _i32, _i64, _f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float


class AmDecodeArgs(C.Structure):
    """Mirror of ``struct rl4co_am_decode_args`` (field order and types must match the header)."""

    _fields_ = [
        ("env", _i32), ("B", _i32), ("B_inst", _i32), ("N", _i32),
        ("mode", _i32), ("max_steps", _i32), ("mask_inner", _i32), ("mask_logits", _i32),
        ("tanh_clipping", _f32), ("temperature", _f32),
        ("cache_dtype", _i32), ("variant", _i32),
        ("glimpse_key", _vp), ("glimpse_val", _vp), ("logit_key", _vp),
        ("kvl_row_stride", _i64), ("kvl_batch_stride", _i64),
        ("ctx_first", _vp), ("ctx_cur", _vp), ("q_bias", _vp), ("q_step0", _vp), ("w_cap", _vp),
        ("unfold", _i32), ("ctx_width", _i32), ("node_embed", _vp), ("w_ctx_t", _vp), ("w_out_t", _vp),
        ("w_placeholder", _vp),
        ("action_mask", _vp), ("first_node", _vp), ("current_node", _vp), ("step_i", _vp),
        ("done", _vp),
        ("demand", _vp), ("used_capacity", _vp), ("vehicle_capacity", _vp), ("visited", _vp),
        ("locs", _vp), ("max_length", _vp), ("to_deliver", _vp),
        ("time_windows", _vp), ("durations", _vp), ("current_time", _vp), ("w_time", _vp),
        ("exp_noise", _vp), ("philox_seed", C.c_uint64), ("philox_offset", C.c_uint64), ("philox_seed_dev", _vp),
        ("forced_actions", _vp),
        ("t0", _i32), ("out_stride", _i32),
        ("actions", _vp), ("logps", _vp), ("all_logps", _vp), ("entropy", _vp),
        ("n_steps", _vp), ("steps_summary", _vp), ("err", _vp),
        ("ctx_dtype", _i32), ("reserved0", _i32), ("ctx_row_stride", _i64), ("ctx_batch_stride", _i64),
    ]


class EnvReplayArgs(C.Structure):
    """Mirror of ``struct rl4co_env_replay_args``."""

    _fields_ = [
        ("env", _i32), ("B", _i32), ("B_inst", _i32), ("N", _i32), ("T", _i32), ("reserved0", _i32),
        ("actions", _vp),
        ("action_mask", _vp), ("current_node", _vp), ("done", _vp), ("first_node", _vp), ("step_i", _vp), ("visited", _vp),
        ("to_deliver", _vp), ("scalar", _vp), ("current_time", _vp),
        ("vehicle_capacity", _vp), ("demand", _vp), ("locs", _vp), ("max_length", _vp), ("time_windows", _vp),
        ("durations", _vp), ("rem_base", _vp),
        ("masks", _vp), ("prev", _vp), ("first", _vp), ("use_placeholder", _vp), ("rem", _vp), ("now", _vp), ("err", _vp),
        ("mask_bits", _vp), ("mask_words", _i32), ("reserved1", _i32),
    ]


class CrossAttnArgs(C.Structure):
    """Mirror of ``struct rl4co_cross_attn_args``."""

    _fields_ = [
        ("B", _i32), ("B_inst", _i32), ("T", _i32), ("N", _i32),
        ("q", _vp), ("q_stride", _i64), ("kv", _vp), ("kv_stride", _i64),
        ("mask", _vp), ("mask_words", _i32), ("reserved0", _i32),
        ("out", _vp), ("lse", _vp), ("dout", _vp), ("dq", _vp), ("dkv", _vp), ("dq_partial", _vp),
    ]


# name -> (restype, argtypes); every symbol include/rl4co_amd.h declares.
SYMBOLS = {
    "rl4co_version": (C.c_char_p, []),
    "rl4co_last_error": (C.c_char_p, []),
    "rl4co_abi_version": (C.c_int, []),
    "rl4co_gather_by_index_f32": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp]),
    "rl4co_tour_length_f32": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "rl4co_tour_length_dyn_f32": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "rl4co_tsp_check_solution": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "rl4co_cvrp_check_solution": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "rl4co_tsp_step": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, _vp, _vp]),
    "rl4co_cvrp_step": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "rl4co_op_max_length": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vp, _vp]),
    "rl4co_op_step": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "rl4co_gather_sum_f32": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "rl4co_op_check_solution": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "rl4co_pctsp_step": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "rl4co_pctsp_check_solution": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "rl4co_cvrptw_step": (C.c_int, [_vp] * 12 + [C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "rl4co_cvrptw_check_solution": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "rl4co_pdp_step": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, _vp, _vp]),
    "rl4co_pdp_check_solution": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "rl4co_env_replay": (C.c_int, [C.POINTER(EnvReplayArgs), _vp]),
    "rl4co_am_decode": (C.c_int, [C.POINTER(AmDecodeArgs), _vp]),
    "rl4co_am_teacher_backward": (C.c_int, [_vp, _vp]),
    "rl4co_am_teacher_max_nodes": (C.c_int, []),
    "rl4co_am_teacher_variant": (C.c_int, [_vp]),
    "rl4co_skip_inorm_fwd": (C.c_int, [C.c_int, _vp, _vp, _vp, _vp, C.c_float, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp]),  # (dtype id first: RL4CO_DT_BF16 / RL4CO_DT_F16)
    "rl4co_skip_inorm_bwd": (C.c_int, [C.c_int, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp]),  # (dtype id first: RL4CO_DT_BF16 / RL4CO_DT_F16)
    "rl4co_skip_lnorm_fwd": (C.c_int, [C.c_int, _vp, _vp, C.c_float, C.c_int, C.c_int, _vp, _vp, _vp, _vp]),  # (dtype id first: RL4CO_DT_BF16 / RL4CO_DT_F16)
    "rl4co_skip_lnorm_bwd": (C.c_int, [C.c_int, _vp, _vp, _vp, C.c_int, C.c_int, _vp, _vp]),  # (dtype id first: RL4CO_DT_BF16 / RL4CO_DT_F16)
    "rl4co_skip_inorm_max_nodes": (C.c_int, []),
    "rl4co_mlp_input_grad": (C.c_int, [_vp, _vp, C.c_int64, _vp, _vp, C.c_int, _vp, _vp, _vp]),
    "rl4co_skip_bnorm_stats": (C.c_int, [C.c_int, _vp, _vp, C.c_int64, _vp, _vp, _vp]),  # (dtype id first: RL4CO_DT_BF16 / RL4CO_DT_F16)
    "rl4co_bnorm_apply": (C.c_int, [C.c_int, _vp, _vp, _vp, _vp, _vp, C.c_int64, _vp, _vp]),  # (dtype id first: RL4CO_DT_BF16 / RL4CO_DT_F16)
    "rl4co_skip_bnorm_eval": (C.c_int, [C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int64, _vp, _vp]),  # (dtype id first: RL4CO_DT_BF16 / RL4CO_DT_F16)
    "rl4co_bnorm_bwd": (C.c_int, [C.c_int, _vp, _vp, _vp, _vp, _vp, C.c_int64, _vp, _vp, _vp]),  # (dtype id first: RL4CO_DT_BF16 / RL4CO_DT_F16)
    "rl4co_init_embed": (C.c_int, [C.c_int, _vp, _vp, _vp, C.c_int64, C.c_int, _vp, _vp]),  # (dtype id first: RL4CO_DT_BF16 / RL4CO_DT_F16)
    "rl4co_init_embed_wgrad": (C.c_int, [C.c_int, _vp, _vp, C.c_int64, C.c_int, _vp, _vp, _vp]),  # (dtype id first: RL4CO_DT_BF16 / RL4CO_DT_F16)
    "rl4co_linear": (C.c_int, [C.c_int, _vp, _vp, _vp, _vp, _vp, C.c_int64, C.c_int, C.c_int, C.c_int, _vp, _vp]),  # (dtype id first: RL4CO_DT_BF16 / RL4CO_DT_F16)
    "rl4co_attn_fwd": (C.c_int, [C.c_int, _vp, C.c_int, C.c_int, _vp, _vp, _vp]),  # (dtype id first: RL4CO_DT_BF16 / RL4CO_DT_F16)
    "rl4co_attn_bwd": (C.c_int, [C.c_int, _vp, _vp, _vp, _vp, C.c_int, C.c_int, _vp, _vp]),  # (dtype id first: RL4CO_DT_BF16 / RL4CO_DT_F16)
    "rl4co_attn_max_nodes": (C.c_int, []),
    "rl4co_attn_bwd_wide": (C.c_int, [C.c_int, _vp, _vp, _vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp]),  # (dtype id first)
    "rl4co_attn_wide_max_nodes": (C.c_int, []),
    "rl4co_cross_attn_fwd": (C.c_int, [C.c_int, C.POINTER(CrossAttnArgs), _vp]),
    "rl4co_cross_attn_bwd": (C.c_int, [C.c_int, C.POINTER(CrossAttnArgs), _vp]),
    "rl4co_cross_attn_chunks": (C.c_int, [C.c_int]),
    "rl4co_logit_logp_fwd": (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_int64, C.c_int, C.c_float, C.c_float, _vp, _vp, _vp, _vp]),
    "rl4co_logit_logp_bwd": (C.c_int, [_vp, _vp, C.c_int, _vp, _vp, _vp, C.c_int64, C.c_int, C.c_float, C.c_float, _vp, _vp]),
    "rl4co_skip_inorm_wide_max_nodes": (C.c_int, []),
    "rl4co_attn_flash": (C.c_int, [C.c_int, _vp, C.c_int, C.c_int, _vp, _vp]),  # (dtype id first: RL4CO_DT_BF16 / RL4CO_DT_F16)
    "rl4co_attn_flash_pre": (C.c_int, [C.c_int, _vp, _vp, C.c_int, C.c_int, _vp, _vp]),  # (dtype id first: RL4CO_DT_BF16 / RL4CO_DT_F16)
    "rl4co_wgrad": (C.c_int, [C.c_int, _vp, _vp, C.c_int64, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int64, _vp]),  # (dtype id first: RL4CO_DT_BF16 / RL4CO_DT_F16)
    "rl4co_am_encoder": (C.c_int, [_vp, _vp]),
    "rl4co_am_encoder_max_nodes": (C.c_int, []),
    "rl4co_am_encoder_train_fwd": (C.c_int, [_vp, _vp, _vp]),
    "rl4co_am_encoder_tokens16": (C.c_int, [_vp, _vp, C.c_int64, _vp]),
    "rl4co_am_encoder_tokens16_workspace": (C.c_int64, [C.c_int, C.c_int]),
    "rl4co_am_encoder_f32": (C.c_int, [_vp, _vp]),
    "rl4co_am_encoder_tokens_f32": (C.c_int, [_vp, _vp, C.c_int64, _vp]),
    "rl4co_am_encoder_tokens_f32_workspace": (C.c_int64, [C.c_int, C.c_int]),
    "rl4co_am_encoder_init_embeds16": (C.c_int, [_vp, _vp, _vp]),
    "rl4co_am_encoder_init_embeds_f32": (C.c_int, [_vp, _vp, _vp]),
    "rl4co_am_fold_tables_f32": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, C.c_int, _vp, _vp, _vp, _vp]),
    "rl4co_am_decode_lds_bytes": (C.c_int, [C.c_int, C.c_int]),
    "rl4co_am_decode_row_groups": (C.c_int, [C.POINTER(AmDecodeArgs)]),
    "rl4co_am_decode_variant": (C.c_int, [C.POINTER(AmDecodeArgs)]),
    "rl4co_select_start_nodes": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "rl4co_augment_dihedral8_f32": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp]),
    "rl4co_augment_symmetric_f32": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_float, _vp, _vp]),
    "rl4co_pomo_best": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "rl4co_hbm_read_probe": (C.c_int, [_vp, _i64, _vp, _vp]),
    "rl4co_math_probe_f32": (C.c_int, [C.c_int, _vp, _i64, _vp, _vp]),
    "rl4co_uniform_f32": (C.c_int, [_vp, _i64, C.c_float, C.c_float, C.c_uint64, C.c_uint32, C.c_int, C.c_float, _vp]),
    # IEEE-half twins of the training-encoder / attention kernels (csrc/elem16.h)
}


class AmTrainSave(C.Structure):
    """Mirror of ``struct rl4co_am_train_save`` (field order and types must match the header)."""

    _fields_ = [("x0", _vp), ("out", _vp), ("qkv", _vp), ("att", _vp), ("y1", _vp), ("x1", _vp), ("h", _vp), ("y2", _vp),
                ("lse", _vp), ("stats", _vp)]


def dtype_id(dtype) -> int:
    """torch dtype of the cache planes -> RL4CO_DT_*"""
    import torch

    try:
        return {torch.float32: DT_F32, torch.bfloat16: DT_BF16, torch.float16: DT_F16}[dtype]
    except KeyError:
        raise TypeError(f"cache dtype must be float32, bfloat16 or float16, got {dtype}") from None


class Rl4coLibraryError(RuntimeError):
    pass


@lru_cache(maxsize=None)
def lib() -> C.CDLL:
    """Load ``librl4co_amd.so`` (building it in-tree first if it is stale or absent).

    torch is imported FIRST on purpose: the PyTorch-ROCm wheel bundles its own HIP runtime
    (``torch/lib/libamdhip64.so``, soname ``libamdhip64.so.7``). Our library needs the same soname,
    so once torch's copy is resident the dynamic loader binds us to it and kernels, streams and
    device pointers all live in ONE runtime. Loaded the other way round the process ends up with
    two HIP runtimes (ours from /opt/rocm, torch's bundled one) and torch's stream handles are
    meaningless to ours ("no ROCm-capable device is detected")."""
    import torch  # noqa: F401  (see above)

    # RL4CO_AMD_LIB: load a specific build instead (tools/enc_probe.sh times instrumented variants of one kernel)
    path = os.environ.get("RL4CO_AMD_LIB") or _build.build_library()
    try:
        handle = C.CDLL(str(path))
    except OSError as exc:  # pragma: no cover - depends on the environment
        raise Rl4coLibraryError(
            f"cannot load {path}: {exc}. The HIP extension is required; there is no CPU fallback."
        ) from exc
    for name, (restype, argtypes) in SYMBOLS.items():
        fn = getattr(handle, name)  # AttributeError if the symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    got = handle.rl4co_abi_version()
    if got != ABI_VERSION:  # a stale build under unchanged symbol names would run with shifted arguments: refuse
        raise Rl4coLibraryError(f"{path} was built for ABI version {got}, this binding is written for {ABI_VERSION}: rebuild the library")
    return handle


def decode_row_groups(num_nodes: int, cache_dtype_id: int, max_steps: int, variant: int = VARIANT_AUTO,
                      num_trajectories: int = 1 << 20, num_instances: int | None = None) -> int:
    """Row groups G (the glimpse summation tree) of the kernel variant that serves this shape —
    a pure host query, usable without a GPU; the specified-order oracle mirrors it."""
    a = AmDecodeArgs()
    a.N, a.cache_dtype, a.max_steps, a.variant = int(num_nodes), int(cache_dtype_id), int(max_steps), int(variant)
    a.B = int(num_trajectories)
    a.B_inst = int(num_trajectories if num_instances is None else num_instances)
    g = lib().rl4co_am_decode_row_groups(C.byref(a))
    if g < 0:
        raise Rl4coLibraryError(f"no decode kernel variant {variant} for N={num_nodes}, dtype id {cache_dtype_id}")
    return g


def check(status: int, what: str) -> None:
    if status != RL4CO_OK:
        msg = lib().rl4co_last_error().decode("utf-8", "replace")
        raise Rl4coLibraryError(f"{what} failed with status {status}: {msg}")


def raise_for_error_bits(bits: int) -> None:
    """Re-raise the reference's assertion for the lowest sticky error bit set."""
    if bits == 0:
        return
    for bit, msg in ERROR_MESSAGES.items():
        if bits & bit:
            raise AssertionError(msg)
    raise AssertionError(f"rollout kernel reported error bits {bits:#x}")


_warned: set[str] = set()


def warn_fallback(key: str, message: str) -> None:
    """One ``RuntimeWarning`` per distinct reason when a part of the path leaves the hand-written kernels for the torch
    implementation (a shape or precision regime no kernel serves). The results stay correct — torch runs the same
    algebra — but the speed is torch's; silent fallbacks hid that."""
    if key in _warned:
        return
    _warned.add(key)
    import warnings

    warnings.warn(f"rl4co_amd: {message}", RuntimeWarning, stacklevel=3)
