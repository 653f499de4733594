"""Thin torch-tensor front end over the C-ABI (``include/rl4co_amd.h``).

torch is plumbing here: it owns device memory and the HIP stream; every function below hands
raw device pointers + sizes to ``librl4co_amd.so`` on ``torch.cuda.current_stream()``.
There is no CPU path: a non-CUDA tensor is an error.
"""
from __future__ import annotations

import ctypes as C

import torch
from torch import Tensor

from . import _lib
from .cache import FoldedCache

MODE_IDS = {"greedy": _lib.DECODE_GREEDY, "sampling": _lib.DECODE_SAMPLE, "evaluate": _lib.DECODE_EVALUATE}
ENV_IDS = {"tsp": _lib.ENV_TSP, "cvrp": _lib.ENV_CVRP, "op": _lib.ENV_OP, "pctsp": _lib.ENV_PCTSP, "pdp": _lib.ENV_PDP, "cvrptw": _lib.ENV_CVRPTW}
VARIANT_IDS = {"auto": _lib.VARIANT_AUTO, "stream": _lib.VARIANT_STREAM, "lds": _lib.VARIANT_LDS, "wide": _lib.VARIANT_WIDE, "ms": _lib.VARIANT_MS}


def decode_row_groups(num_nodes: int, cache_dtype: torch.dtype, max_steps: int, variant: str = "auto",
                      num_trajectories: int = 1 << 20, num_instances: int | None = None) -> int:
    """Row groups of the specified-order contract of the variant that would run (0 = the multistart
    MFMA variant, which has none)."""
    dt = _lib.dtype_id(cache_dtype)
    return _lib.decode_row_groups(num_nodes, dt, max_steps, VARIANT_IDS[variant], num_trajectories, num_instances)


def decode_variant(num_nodes: int, cache_dtype: torch.dtype, max_steps: int, num_trajectories: int,
                   num_instances: int | None = None, env_name: str = "tsp") -> int:
    """The RL4CO_VARIANT_* rl4co_am_decode would run for this shape (host query)."""
    a = _lib.AmDecodeArgs()
    a.env = ENV_IDS[env_name]
    a.N, a.max_steps, a.B = int(num_nodes), int(max_steps), int(num_trajectories)
    a.B_inst = int(num_trajectories if num_instances is None else num_instances)
    a.cache_dtype = _lib.dtype_id(cache_dtype)
    return _lib.lib().rl4co_am_decode_variant(C.byref(a))


def _ptr(t: Tensor | None) -> int | None:
    return None if t is None else t.data_ptr()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _dev(t: Tensor, dtype: torch.dtype | None = None, name: str = "tensor") -> Tensor:
    if not t.is_cuda:
        raise _lib.Rl4coLibraryError(
            f"{name} lives on {t.device}; the rl4co_amd kernels only run on the MI355X (no CPU fallback)"
        )
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    return t


def _ctx_table(a, t: Tensor, plane_dtype: torch.dtype, name: str) -> Tensor:
    """A folded context table for ``rl4co_am_decode_args``: dense fp32 [B,N,128] (every variant), or — multistart variant —
    rows in the planes' 16-bit type with any (instance, node) strides, e.g. a column block of the fused fold GEMM's output
    matrix (``teacher.build_cache_autograd(fused_planes=True)``): sets ctx_dtype / ctx_row_stride / ctx_batch_stride."""
    if not t.is_cuda:
        raise _lib.Rl4coLibraryError(f"{name} lives on {t.device}; the rl4co_amd kernels only run on the MI355X (no CPU fallback)")
    if t.dtype == torch.float32:
        return _dev(t, torch.float32, name)
    if t.dtype != plane_dtype or t.dim() != 3 or t.stride(2) != 1:
        raise TypeError(f"{name} must be fp32, or [B,N,128] in the planes' dtype {plane_dtype} with unit channel stride")
    dt, rs, bs = _lib.dtype_id(t.dtype), t.stride(1), t.stride(0)
    if a.ctx_dtype not in (0, dt) or (a.ctx_row_stride not in (0, rs)) or (a.ctx_batch_stride not in (0, bs)):
        raise ValueError("ctx_first and ctx_cur must share dtype and strides")
    a.ctx_dtype, a.ctx_row_stride, a.ctx_batch_stride = dt, rs, bs
    return t


def _u8(t: Tensor, name: str) -> Tensor:
    """bool tensors share the uint8 storage the kernels read/write."""
    _dev(t, None, name)
    if t.dtype == torch.bool:
        return t.view(torch.uint8)
    if t.dtype != torch.uint8:
        raise TypeError(f"{name} must be bool or uint8, got {t.dtype}")
    return t


def _check_rows(b: int, **tensors) -> None:
    """Every per-trajectory tensor of a step call must hold exactly ``b`` rows (the mask's): a shorter ``action``
    (e.g. an instance-level action against batchified multistart state) would be read out of bounds on the device."""
    for name, t in tensors.items():
        if t is not None and t.shape[0] != b:
            raise ValueError(f"{name} has {t.shape[0]} rows, action_mask has {b}")


def new_error_word(device) -> Tensor:
    return torch.zeros(1, dtype=torch.int32, device=device)


def raise_if_error(err: Tensor) -> None:
    """ONE host sync per rollout instead of the reference's 4-5 per decode step."""
    _lib.raise_for_error_bits(int(err.item()))


# ------------------------------------------------------------------------------------------------


def gather_by_index(src: Tensor, idx: Tensor, err: Tensor | None = None) -> Tensor:
    """ops.py:54-66 for src [B,N,D] fp32, idx [B,K] int64 -> [B,K,D]."""
    _dev(src, torch.float32, "src"), _dev(idx, torch.int64, "idx")
    b, n, d = src.shape
    k = idx.shape[1]
    out = torch.empty((b, k, d), dtype=torch.float32, device=src.device)
    st = _lib.lib().rl4co_gather_by_index_f32(_ptr(src), _ptr(idx), b, n, d, k, _ptr(out), _ptr(err), _stream())
    _lib.check(st, "rl4co_gather_by_index_f32")
    return out


def tour_length(locs: Tensor, actions: Tensor, prepend_depot: bool = False, negate: bool = False,
                horizon: tuple[Tensor, int] | None = None) -> Tensor:
    """ops.py:82-90 over gather(locs, actions) (+ depot for CVRP, cvrp/env.py:138-147).

    locs [B_locs,N,2] fp32 with B % B_locs == 0 (s-major multistart), actions [B,T] int64.
    ``horizon = (steps_dev, t_add)``: the tour is the first ``t_add + steps_dev[0]`` columns of the (padded) buffer,
    the count read on the device (rl4co_tour_length_dyn_f32) — no host value needed."""
    _dev(locs, torch.float32, "locs"), _dev(actions, torch.int64, "actions")
    b, t = actions.shape
    b_locs, n, two = locs.shape
    assert two == 2
    out = torch.empty((b,), dtype=torch.float32, device=locs.device)
    if horizon is not None:
        steps_dev, t_add = horizon
        st = _lib.lib().rl4co_tour_length_dyn_f32(_ptr(locs), _ptr(actions), b, b_locs, n, t, _ptr(_dev(steps_dev, torch.int32, "steps")),
                                                  int(t_add), int(prepend_depot), int(negate), _ptr(out), _stream())
        _lib.check(st, "rl4co_tour_length_dyn_f32")
        return out
    st = _lib.lib().rl4co_tour_length_f32(
        _ptr(locs), _ptr(actions), b, b_locs, n, t, int(prepend_depot), int(negate), _ptr(out), _stream()
    )
    _lib.check(st, "rl4co_tour_length_f32")
    return out


def tsp_check_solution(actions: Tensor, num_nodes: int, err: Tensor) -> None:
    _dev(actions, torch.int64, "actions")
    b, t = actions.shape
    st = _lib.lib().rl4co_tsp_check_solution(_ptr(actions), b, num_nodes, t, _ptr(err), _stream())
    _lib.check(st, "rl4co_tsp_check_solution")


def cvrp_check_solution(actions: Tensor, demand: Tensor, vehicle_capacity: Tensor, err: Tensor) -> None:
    _dev(actions, torch.int64, "actions"), _dev(demand, torch.float32, "demand")
    _dev(vehicle_capacity, torch.float32, "vehicle_capacity")
    b, t = actions.shape
    b_inst, n1 = demand.shape
    st = _lib.lib().rl4co_cvrp_check_solution(
        _ptr(actions), _ptr(demand), _ptr(vehicle_capacity), b, b_inst, n1 + 1, t, _ptr(err), _stream()
    )
    _lib.check(st, "rl4co_cvrp_check_solution")


def tsp_step(action: Tensor, action_mask: Tensor, first_node: Tensor, current_node: Tensor,
             step_i: Tensor, done: Tensor, err: Tensor | None = None) -> None:
    """In-place TSPEnv._step (tsp/env.py:60-86)."""
    _dev(action, torch.int64, "action")
    mask = _u8(action_mask, "action_mask")
    b, n = mask.shape
    _check_rows(b, action=action, first_node=first_node, current_node=current_node, i=step_i, done=done)
    st = _lib.lib().rl4co_tsp_step(
        _ptr(action), _ptr(mask), _ptr(_dev(first_node, torch.int64, "first_node")),
        _ptr(_dev(current_node, torch.int64, "current_node")), _ptr(_dev(step_i, torch.int64, "i")),
        _ptr(_u8(done, "done")), b, n, _ptr(err), _stream(),
    )
    _lib.check(st, "rl4co_tsp_step")


def cvrp_step(action: Tensor | None, demand: Tensor, used_capacity: Tensor, vehicle_capacity: Tensor,
              visited: Tensor, current_node: Tensor, action_mask: Tensor, done: Tensor | None,
              err: Tensor | None = None) -> None:
    """In-place CVRPEnv._step + get_action_mask (cvrp/env.py:66-96,126-136); action=None -> mask only."""
    mask = _u8(action_mask, "action_mask")
    b, n = mask.shape
    b_inst = demand.shape[0]
    _check_rows(b, action=action, used_capacity=used_capacity, vehicle_capacity=vehicle_capacity, visited=visited,
                current_node=current_node, done=done)
    st = _lib.lib().rl4co_cvrp_step(
        _ptr(None if action is None else _dev(action, torch.int64, "action")),
        _ptr(_dev(demand, torch.float32, "demand")), _ptr(_dev(used_capacity, torch.float32, "used_capacity")),
        _ptr(_dev(vehicle_capacity, torch.float32, "vehicle_capacity")), _ptr(_u8(visited, "visited")),
        _ptr(_dev(current_node, torch.int64, "current_node")), _ptr(mask),
        _ptr(None if done is None else _u8(done, "done")), b, b_inst, n, _ptr(err), _stream(),
    )
    _lib.check(st, "rl4co_cvrp_step")


def select_start_nodes(batch: int, num_starts: int, num_loc: int, has_depot: bool, device) -> Tensor:
    """ops.py:128-161: s-major ``arange(S).repeat_interleave(B) % num_loc (+1)``."""
    out = torch.empty((batch * num_starts,), dtype=torch.int64, device=device)
    if not out.is_cuda:
        raise _lib.Rl4coLibraryError("select_start_nodes needs a CUDA device")
    st = _lib.lib().rl4co_select_start_nodes(_ptr(out), batch, num_starts, num_loc, int(has_depot), _stream())
    _lib.check(st, "rl4co_select_start_nodes")
    return out


def am_decode(
    cache: FoldedCache,
    state: dict,
    *,
    mode: str,
    max_steps: int,
    actions: Tensor,
    logps: Tensor,
    err: Tensor,
    t0: int = 0,
    tanh_clipping: float = 10.0,
    temperature: float = 1.0,
    mask_inner: bool = True,
    mask_logits: bool = True,
    exp_noise: Tensor | None = None,
    philox_seed: int = 0,
    philox_offset: int = 0,
    philox_seed_dev: Tensor | None = None,
    forced_actions: Tensor | None = None,
    all_logps: Tensor | None = None,
    entropy: Tensor | None = None,
    n_steps: Tensor | None = None,
    steps_summary: Tensor | None = None,
    variant: str = "auto",
) -> None:
    """Run ``max_steps`` fused decode steps (1 = a single step, >= horizon = whole rollout).

    ``state`` holds the environment tensors (updated in place): action_mask [B,N] bool,
    current_node, done; TSP: first_node, i; CVRP: demand, used_capacity, vehicle_capacity, visited.
    """
    env_name = cache.env_name
    a = _lib.AmDecodeArgs()
    mask = _u8(state["action_mask"], "action_mask")
    b, n = mask.shape
    assert n == cache.num_nodes, (n, cache.num_nodes)
    a.env = ENV_IDS[env_name]
    a.B, a.B_inst, a.N = b, cache.num_instances, n
    a.mode = MODE_IDS[mode]
    a.max_steps = int(max_steps)
    a.variant = VARIANT_IDS[variant]
    a.mask_inner, a.mask_logits = int(mask_inner), int(mask_logits)
    a.tanh_clipping, a.temperature = float(tanh_clipping), float(temperature)
    kvl = cache.kvl  # [3, B, N, 128]: plane pointers + (instance, node) strides — any view with unit channel stride
    if not kvl.is_cuda:
        raise _lib.Rl4coLibraryError(f"cache.kvl lives on {kvl.device}; the rl4co_amd kernels only run on the MI355X (no CPU fallback)")
    if not (kvl.dim() == 4 and kvl.stride(3) == 1):
        raise ValueError("cache.kvl must be [3, B, N, 128] with unit stride along the channels")
    a.cache_dtype = _lib.dtype_id(kvl.dtype)
    a.glimpse_key, a.glimpse_val, a.logit_key = (cache.plane(i).data_ptr() for i in range(3))
    a.kvl_row_stride, a.kvl_batch_stride = cache.row_stride, cache.batch_stride
    if cache.unfold:  # reference-association parity mode: per-step GEMVs against the raw weights (cache.py)
        a.unfold, a.ctx_width = 1, cache.w_ctx_t.shape[0]
        a.node_embed = _ptr(_dev(cache.node_embed, torch.float32, "node_embed"))
        a.w_ctx_t = _ptr(_dev(cache.w_ctx_t, torch.float32, "w_ctx_t"))
        a.w_out_t = _ptr(_dev(cache.w_out_t, torch.float32, "w_out_t"))
        a.w_placeholder = _ptr(None if cache.w_placeholder is None else _dev(cache.w_placeholder, torch.float32, "w_placeholder"))
    else:
        a.ctx_cur = _ptr(_ctx_table(a, cache.ctx_cur, kvl.dtype, "ctx_cur"))
    a.q_bias = _ptr(None if cache.q_bias is None else _dev(cache.q_bias, torch.float32, "q_bias"))
    a.action_mask = _ptr(mask)
    a.current_node = _ptr(_dev(state["current_node"], torch.int64, "current_node"))
    a.done = _ptr(_u8(state["done"], "done"))
    if env_name == "tsp":
        if not cache.unfold:
            a.ctx_first = _ptr(_ctx_table(a, cache.ctx_first, kvl.dtype, "ctx_first"))
            a.q_step0 = _ptr(_dev(cache.q_step0, torch.float32, "q_step0"))
        a.first_node = _ptr(_dev(state["first_node"], torch.int64, "first_node"))
        a.step_i = _ptr(_dev(state["i"], torch.int64, "i"))
    elif env_name == "op":
        # orienteering: the tour length rides in the used_capacity slot, the per-node entry limits
        # (max_length table) and the coordinates are instance data like CVRP's demand
        a.w_cap = _ptr(_dev(cache.w_cap, torch.float32, "w_cap"))
        assert state["locs"].shape[0] == cache.num_instances and state["max_length"].shape[0] == cache.num_instances
        a.locs = _ptr(_dev(state["locs"], torch.float32, "locs"))
        a.max_length = _ptr(_dev(state["max_length"], torch.float32, "max_length"))
        a.used_capacity = _ptr(_dev(state["tour_length"], torch.float32, "tour_length"))
        a.step_i = _ptr(_dev(state["i"], torch.int64, "i"))
        a.visited = _ptr(_u8(state["visited"], "visited"))
    elif env_name == "pdp":
        # pickup and delivery: `available` rides in the visited slot; no context scalar
        a.visited = _ptr(_u8(state["available"], "available"))
        a.to_deliver = _ptr(_u8(state["to_deliver"], "to_deliver"))
        a.step_i = _ptr(_dev(state["i"], torch.int64, "i"))
    elif env_name == "pctsp":
        # prize-collecting TSP: the real prize per node (depot column 0) rides in the demand slot, the
        # prize collected so far in used_capacity, prize_required in vehicle_capacity
        a.w_cap = _ptr(_dev(cache.w_cap, torch.float32, "w_cap"))
        assert state["real_prize"].shape == (cache.num_instances, n)
        a.demand = _ptr(_dev(state["real_prize"], torch.float32, "real_prize"))
        a.used_capacity = _ptr(_dev(state["cur_total_prize"], torch.float32, "cur_total_prize"))
        a.vehicle_capacity = _ptr(_dev(state["prize_required"], torch.float32, "prize_required"))
        a.step_i = _ptr(_dev(state["i"], torch.int64, "i"))
        a.visited = _ptr(_u8(state["visited"], "visited"))
    else:
        if env_name == "cvrptw":  # CVRP + clock: coordinates, (start, end) windows, service times as fp32 instance data
            a.w_time = _ptr(_dev(cache.w_time, torch.float32, "w_time"))
            a.locs = _ptr(_dev(state["locs"], torch.float32, "locs"))
            a.time_windows = _ptr(_dev(state["time_windows"], torch.float32, "time_windows"))
            a.durations = _ptr(_dev(state["durations"], torch.float32, "durations"))
            a.current_time = _ptr(_dev(state["current_time"], torch.float32, "current_time"))
            assert state["time_windows"].shape == (cache.num_instances, n, 2)
        if not cache.unfold:
            a.w_cap = _ptr(_dev(cache.w_cap, torch.float32, "w_cap"))
        a.demand = _ptr(_dev(state["demand"], torch.float32, "demand"))
        assert state["demand"].shape[0] in (cache.num_instances,), "demand rows must match cache instances"
        a.used_capacity = _ptr(_dev(state["used_capacity"], torch.float32, "used_capacity"))
        a.vehicle_capacity = _ptr(_dev(state["vehicle_capacity"], torch.float32, "vehicle_capacity"))
        a.visited = _ptr(_u8(state["visited"], "visited"))
    if exp_noise is not None:
        _dev(exp_noise, torch.float32, "exp_noise")
        assert exp_noise.numel() >= max_steps * b * n, "exp_noise must hold [max_steps,B,N] draws"
        a.exp_noise = _ptr(exp_noise)
    a.philox_seed, a.philox_offset = int(philox_seed), int(philox_offset)
    if philox_seed_dev is not None:
        a.philox_seed_dev = _ptr(_dev(philox_seed_dev, torch.int64, "philox_seed_dev"))
    if forced_actions is not None:
        _dev(forced_actions, torch.int64, "forced_actions")
        assert forced_actions.shape == actions.shape
        a.forced_actions = _ptr(forced_actions)
    _dev(actions, torch.int64, "actions"), _dev(logps, torch.float32, "logps")
    assert actions.shape == logps.shape and actions.shape[0] == b
    a.t0, a.out_stride = int(t0), actions.shape[1]
    a.actions, a.logps = _ptr(actions), _ptr(logps)
    a.all_logps = _ptr(None if all_logps is None else _dev(all_logps, torch.float32, "all_logps"))
    a.entropy = _ptr(None if entropy is None else _dev(entropy, torch.float32, "entropy"))
    a.n_steps = _ptr(None if n_steps is None else _dev(n_steps, torch.int32, "n_steps"))
    if steps_summary is not None:  # [max steps, sum steps, rows low, rows high]: the row counter is ONE 64-bit word
        _dev(steps_summary, torch.int32, "steps_summary")
        if steps_summary.numel() < 4 or steps_summary.data_ptr() % 8:
            raise ValueError("steps_summary must be 4 int32 words on an 8-byte boundary (rl4co_am_decode_args.steps_summary)")
    a.steps_summary = _ptr(steps_summary)
    a.err = _ptr(_dev(err, torch.int32, "err"))
    st = _lib.lib().rl4co_am_decode(C.byref(a), _stream())
    _lib.check(st, "rl4co_am_decode")


def op_max_length(locs: Tensor, max_length: Tensor) -> Tensor:
    """op/env.py:118-122: table[b,j] = (max_length[b] - |loc_0 - loc_j|) - 1e-6 (locs include the depot)."""
    b, n, _ = locs.shape
    locs = _dev(locs, torch.float32, "locs")
    ml = _dev(max_length.reshape(-1).contiguous(), torch.float32, "max_length")
    out = torch.empty((b, n), dtype=torch.float32, device=locs.device)
    st = _lib.lib().rl4co_op_max_length(_ptr(locs), _ptr(ml), b, n, _ptr(out), _stream())
    _lib.check(st, "rl4co_op_max_length")
    return out


def op_step(action: Tensor | None, locs: Tensor, max_length: Tensor, tour_length: Tensor, visited: Tensor,
            current_node: Tensor, step_i: Tensor, action_mask: Tensor, done: Tensor, err: Tensor | None = None) -> None:
    """In-place OPEnv._step + get_action_mask (op/env.py:67-98,137-154); action=None -> mask only."""
    b, n = action_mask.shape
    _check_rows(b, action=action, tour_length=tour_length, visited=visited, current_node=current_node, i=step_i, done=done)
    st = _lib.lib().rl4co_op_step(
        _ptr(None if action is None else _dev(action, torch.int64, "action")), _ptr(_dev(locs, torch.float32, "locs")),
        _ptr(_dev(max_length, torch.float32, "max_length")), _ptr(_dev(tour_length, torch.float32, "tour_length")),
        _ptr(_u8(visited, "visited")), _ptr(_dev(current_node, torch.int64, "current_node")),
        _ptr(_dev(step_i, torch.int64, "i")), _ptr(_u8(action_mask, "action_mask")), _ptr(_u8(done, "done")),
        b, locs.shape[0], n, _ptr(err), _stream())
    _lib.check(st, "rl4co_op_step")


def cvrptw_step(action: Tensor | None, demand: Tensor, locs: Tensor, time_windows: Tensor, durations: Tensor,
                used_capacity: Tensor, vehicle_capacity: Tensor, current_time: Tensor, visited: Tensor, current_node: Tensor,
                action_mask: Tensor, done: Tensor | None, err: Tensor | None = None) -> None:
    """In-place CVRPTWEnv._step + get_action_mask (cvrptw/env.py:83-113); action=None -> mask only.
    ``time_windows`` [B_inst,N,2] and ``durations`` [B_inst,N] are fp32 (the reference's integer windows cast)."""
    b, n = action_mask.shape
    _check_rows(b, action=action, used_capacity=used_capacity, vehicle_capacity=vehicle_capacity, current_time=current_time,
                visited=visited, current_node=current_node, done=done)
    st = _lib.lib().rl4co_cvrptw_step(
        _ptr(None if action is None else _dev(action, torch.int64, "action")), _ptr(_dev(demand, torch.float32, "demand")),
        _ptr(_dev(locs, torch.float32, "locs")), _ptr(_dev(time_windows, torch.float32, "time_windows")),
        _ptr(_dev(durations, torch.float32, "durations")), _ptr(_dev(used_capacity, torch.float32, "used_capacity")),
        _ptr(_dev(vehicle_capacity, torch.float32, "vehicle_capacity")), _ptr(_dev(current_time, torch.float32, "current_time")),
        _ptr(_u8(visited, "visited")), _ptr(_dev(current_node, torch.int64, "current_node")),
        _ptr(_u8(action_mask, "action_mask")), _ptr(None if done is None else _u8(done, "done")),
        b, demand.shape[0], n, _ptr(err), _stream())
    _lib.check(st, "rl4co_cvrptw_step")


def cvrptw_check_solution(actions: Tensor, locs: Tensor, time_windows: Tensor, durations: Tensor, err: Tensor) -> None:
    """cvrptw/env.py:151-190 (the part on top of the CVRP check) into the sticky error word."""
    b, t = actions.shape
    st = _lib.lib().rl4co_cvrptw_check_solution(
        _ptr(_dev(actions, torch.int64, "actions")), _ptr(_dev(locs, torch.float32, "locs")),
        _ptr(_dev(time_windows, torch.float32, "time_windows")), _ptr(_dev(durations, torch.float32, "durations")),
        b, locs.shape[0], locs.shape[1], t, _ptr(_dev(err, torch.int32, "err")), _stream())
    _lib.check(st, "rl4co_cvrptw_check_solution")


def pdp_step(action: Tensor | None, available: Tensor, to_deliver: Tensor, current_node: Tensor, step_i: Tensor,
             action_mask: Tensor, done: Tensor, err: Tensor | None = None) -> None:
    """In-place PDPEnv._step (pdp/env.py:64-99); action=None -> mask = available & to_deliver only."""
    b, n = action_mask.shape
    _check_rows(b, action=action, available=available, to_deliver=to_deliver, current_node=current_node, i=step_i, done=done)
    st = _lib.lib().rl4co_pdp_step(
        _ptr(None if action is None else _dev(action, torch.int64, "action")), _ptr(_u8(available, "available")),
        _ptr(_u8(to_deliver, "to_deliver")), _ptr(_dev(current_node, torch.int64, "current_node")),
        _ptr(_dev(step_i, torch.int64, "i")), _ptr(_u8(action_mask, "action_mask")), _ptr(_u8(done, "done")),
        b, n, _ptr(err), _stream())
    _lib.check(st, "rl4co_pdp_step")


def env_replay(env_name: str, state: dict, actions: Tensor, rem_base: Tensor | None, err: Tensor | None = None,
               mask_bits: bool = False) -> dict:
    """``T`` environment transitions of the given trajectories in ONE launch (``rl4co_env_replay``): the state tensors of
    ``policy._initial_state`` are stepped in place with ``actions[:, t]`` exactly as ``T`` calls of the env's step entry
    would, and what the decoder saw BEFORE each step is tabulated — ``masks`` [B,T,N] bool, ``prev`` [B,T] and, by
    environment, ``first`` / ``use_placeholder`` (TSP), ``rem`` (the context scalar: ``rem_base`` minus the running
    capacity / length / prize), ``now`` (CVRPTW), ``mask_bits`` [B,T,W] int32 on request. The `evaluate` decoding's state
    sequence (decoding.py:448-461)."""
    mask = _u8(state["action_mask"], "action_mask")
    b, n = mask.shape
    acts = _dev(actions, torch.int64, "actions")
    if acts.dim() != 2 or acts.shape[0] != b:
        raise ValueError(f"actions must be [B = {b}, T], got {tuple(acts.shape)}")
    t_len = acts.shape[1]
    dev = acts.device
    a = _lib.EnvReplayArgs()
    a.env, a.B, a.N, a.T = ENV_IDS[env_name], b, n, t_len
    out = {"masks": torch.empty((b, t_len, n), dtype=torch.bool, device=dev),
           "prev": torch.empty((b, t_len), dtype=torch.int64, device=dev)}
    a.actions, a.action_mask = acts.data_ptr(), mask.data_ptr()
    a.current_node = _dev(state["current_node"], torch.int64, "current_node").data_ptr()
    a.done = _u8(state["done"], "done").data_ptr()
    a.masks, a.prev = out["masks"].data_ptr(), out["prev"].data_ptr()
    a.err = _ptr(err)
    if mask_bits:  # the same masks as bits, rows padded to whole 128-key chunks (train_ops.glimpse_attention's mask)
        words = 4 * ((n + 127) // 128)
        out["mask_bits"] = torch.empty((b, t_len, words), dtype=torch.int32, device=dev)
        a.mask_bits, a.mask_words = out["mask_bits"].data_ptr(), words
    _check_rows(b, current_node=state["current_node"], done=state["done"])
    b_inst = b

    def f32(key, rows=None):
        t = _dev(state[key], torch.float32, key)
        if rows is not None and t.shape[0] != rows:
            raise ValueError(f"{key} has {t.shape[0]} rows, expected {rows}")
        return t

    if env_name == "tsp":
        out["first"] = torch.empty((b, t_len), dtype=torch.int64, device=dev)
        out["use_placeholder"] = torch.empty((b, t_len), dtype=torch.bool, device=dev)
        _check_rows(b, first_node=state["first_node"], i=state["i"])
        a.first_node = _dev(state["first_node"], torch.int64, "first_node").data_ptr()
        a.step_i = _dev(state["i"], torch.int64, "i").data_ptr()
        a.first, a.use_placeholder = out["first"].data_ptr(), out["use_placeholder"].data_ptr()
    elif env_name == "pdp":
        _check_rows(b, available=state["available"], to_deliver=state["to_deliver"], i=state["i"])
        a.visited = _u8(state["available"], "available").data_ptr()
        a.to_deliver = _u8(state["to_deliver"], "to_deliver").data_ptr()
        a.step_i = _dev(state["i"], torch.int64, "i").data_ptr()
    else:
        scalar_key = {"cvrp": "used_capacity", "cvrptw": "used_capacity", "op": "tour_length", "pctsp": "cur_total_prize"}[env_name]
        if rem_base is None:
            raise ValueError(f"{env_name}: rem_base (the context scalar's minuend, one per trajectory) is required")
        base = _dev(rem_base, torch.float32, "rem_base")
        _check_rows(b, visited=state["visited"], rem_base=base, **{scalar_key: state[scalar_key]})
        out["rem"] = torch.empty((b, t_len), dtype=torch.float32, device=dev)
        a.visited = _u8(state["visited"], "visited").data_ptr()
        a.scalar, a.rem_base, a.rem = f32(scalar_key).data_ptr(), base.data_ptr(), out["rem"].data_ptr()
        if env_name in ("cvrp", "cvrptw"):
            dem = f32("demand")
            b_inst = dem.shape[0]
            if dem.shape[1] != n - 1:
                raise ValueError(f"demand must be [B_inst, {n - 1}], got {tuple(dem.shape)}")
            a.demand, a.vehicle_capacity = dem.data_ptr(), f32("vehicle_capacity", b).data_ptr()
            if env_name == "cvrptw":
                out["now"] = torch.empty((b, t_len), dtype=torch.float32, device=dev)
                a.locs = f32("locs", b_inst).data_ptr()
                a.time_windows, a.durations = f32("time_windows", b_inst).data_ptr(), f32("durations", b_inst).data_ptr()
                a.current_time, a.now = f32("current_time", b).data_ptr(), out["now"].data_ptr()
        elif env_name == "pctsp":
            rp = f32("real_prize")
            b_inst = rp.shape[0]
            if rp.shape[1] != n:
                raise ValueError(f"real_prize must be [B_inst, {n}], got {tuple(rp.shape)}")
            a.demand, a.step_i = rp.data_ptr(), _dev(state["i"], torch.int64, "i").data_ptr()
        else:  # op
            lc = f32("locs")
            b_inst = lc.shape[0]
            a.locs, a.max_length = lc.data_ptr(), f32("max_length", b_inst).data_ptr()
            a.step_i = _dev(state["i"], torch.int64, "i").data_ptr()
    if b % b_inst:
        raise ValueError(f"{b} trajectories over {b_inst} instances")
    a.B_inst = b_inst
    import ctypes

    st = _lib.lib().rl4co_env_replay(ctypes.byref(a), _stream())
    _lib.check(st, "rl4co_env_replay")
    return out


def pdp_check_solution(actions: Tensor, num_nodes: int, force_start_at_depot: bool, err: Tensor) -> None:
    """pdp/env.py:204-223 into the sticky error word (NOT_ALL_NODES / DEPOT_MIDDLE / NO_PICKUP)."""
    b, t = actions.shape
    st = _lib.lib().rl4co_pdp_check_solution(_ptr(_dev(actions, torch.int64, "actions")), b, num_nodes, t,
                                             int(force_start_at_depot), _ptr(_dev(err, torch.int32, "err")), _stream())
    _lib.check(st, "rl4co_pdp_check_solution")


def pctsp_step(action: Tensor | None, real_prize: Tensor, cur_total_prize: Tensor, visited: Tensor, current_node: Tensor,
               step_i: Tensor, action_mask: Tensor, done: Tensor, err: Tensor | None = None) -> None:
    """In-place PCTSPEnv._step + get_action_mask (pctsp/env.py:62-91,141-148); action=None -> mask only.
    ``real_prize`` [B_inst, N] carries 0 in the depot column."""
    b, n = action_mask.shape
    _check_rows(b, action=action, cur_total_prize=cur_total_prize, visited=visited, current_node=current_node, i=step_i,
                done=done)
    st = _lib.lib().rl4co_pctsp_step(
        _ptr(None if action is None else _dev(action, torch.int64, "action")),
        _ptr(_dev(real_prize, torch.float32, "real_prize")), _ptr(_dev(cur_total_prize, torch.float32, "cur_total_prize")),
        _ptr(_u8(visited, "visited")), _ptr(_dev(current_node, torch.int64, "current_node")),
        _ptr(_dev(step_i, torch.int64, "i")), _ptr(_u8(action_mask, "action_mask")), _ptr(_u8(done, "done")),
        b, real_prize.shape[0], n, _ptr(err), _stream())
    _lib.check(st, "rl4co_pctsp_step")


def pctsp_check_solution(actions: Tensor, real_prize: Tensor, err: Tensor) -> None:
    """pctsp/env.py:175-201 into the sticky error word (RL4CO_EBIT_DUPLICATES / RL4CO_EBIT_PRIZE)."""
    b, t = actions.shape
    prize_sum = gather_sum(real_prize, actions)
    st = _lib.lib().rl4co_pctsp_check_solution(_ptr(_dev(actions, torch.int64, "actions")), _ptr(prize_sum), b,
                                               real_prize.shape[1], t, _ptr(_dev(err, torch.int32, "err")), _stream())
    _lib.check(st, "rl4co_pctsp_check_solution")


def gather_sum(values: Tensor, actions: Tensor) -> Tensor:
    """out[b] = sum_t values[b % B_values, actions[b, t]] in ATen's inner-dim sum order (op/env.py:156-166)."""
    b, t = actions.shape
    values = _dev(values, torch.float32, "values")
    out = torch.empty((b,), dtype=torch.float32, device=actions.device)
    st = _lib.lib().rl4co_gather_sum_f32(_ptr(values), _ptr(_dev(actions, torch.int64, "actions")), b, values.shape[0],
                                         values.shape[1], t, _ptr(out), _stream())
    _lib.check(st, "rl4co_gather_sum_f32")
    return out


def op_check_solution(actions: Tensor, locs: Tensor, max_length: Tensor, err: Tensor) -> None:
    """op/env.py:168-194 into the sticky error word (RL4CO_EBIT_DUPLICATES / RL4CO_EBIT_MAX_LENGTH)."""
    b, t = actions.shape
    st = _lib.lib().rl4co_op_check_solution(_ptr(_dev(actions, torch.int64, "actions")), _ptr(_dev(locs, torch.float32, "locs")),
                                            _ptr(_dev(max_length, torch.float32, "max_length")), b, locs.shape[0],
                                            locs.shape[1], t, _ptr(_dev(err, torch.int32, "err")), _stream())
    _lib.check(st, "rl4co_op_check_solution")


def hbm_read_probe(buf: Tensor, sink: Tensor) -> None:
    st = _lib.lib().rl4co_hbm_read_probe(_ptr(buf), buf.numel() * buf.element_size(), _ptr(sink), _stream())
    _lib.check(st, "rl4co_hbm_read_probe")


def math_probe(fn: str, x: Tensor) -> Tensor:
    """exp / log / tanh of csrc/rl4co_math.h evaluated on the device (tests/test_math.py)."""
    x = _dev(x.contiguous(), torch.float32, "x")
    y = torch.empty_like(x)
    st = _lib.lib().rl4co_math_probe_f32({"exp": 0, "log": 1, "tanh": 2}[fn], _ptr(x), x.numel(), _ptr(y), _stream())
    _lib.check(st, "rl4co_math_probe_f32")
    return y


def uniform(shape, low: float, high: float, seed: int, stream_id: int, device, demand_capacity: float | None = None) -> Tensor:
    """U(low, high) drawn on the device in one launch (rl4co_uniform_f32, Philox4x32-10 keyed by ``seed`` /
    ``stream_id``); ``demand_capacity``: CVRP's integer-demand map ``(trunc(v) + 1) / capacity`` on top."""
    out = torch.empty(tuple(shape), dtype=torch.float32, device=device)
    _dev(out, torch.float32, "out")
    st = _lib.lib().rl4co_uniform_f32(_ptr(out), out.numel(), float(low), float(high), int(seed) & ((1 << 64) - 1), int(stream_id),
                                      0 if demand_capacity is None else 1, float(demand_capacity or 1.0), _stream())
    _lib.check(st, "rl4co_uniform_f32")
    return out


# ---- N3: state augmentation and the POMO evaluation epilogue (csrc/augment.hip) -------------------------------------

def augment_dihedral8(xy: Tensor) -> Tensor:
    """[B, N, 2] -> [8 * B, N, 2], aug-major: the 8 symmetries of the unit square (data/transforms.py:16-46)."""
    b, n, two = xy.shape
    assert two == 2
    xy = _dev(xy, torch.float32, "xy")
    out = torch.empty((8 * b, n, 2), dtype=torch.float32, device=xy.device)
    st = _lib.lib().rl4co_augment_dihedral8_f32(_ptr(xy), b, n, _ptr(out), _stream())
    _lib.check(st, "rl4co_augment_dihedral8_f32")
    return out


def augment_symmetric(xy: Tensor, cos_phi: Tensor, sin_phi: Tensor, swap_axes: Tensor, offset: float = 0.5) -> Tensor:
    """[B, N, 2] and one (cos, sin, swap) per OUTPUT row [A * B] -> [A * B, N, 2] (data/transforms.py:49-69)."""
    b, n, two = xy.shape
    rows = cos_phi.numel()
    assert two == 2 and rows % b == 0 and sin_phi.numel() == rows and swap_axes.numel() == rows
    xy = _dev(xy, torch.float32, "xy")
    out = torch.empty((rows, n, 2), dtype=torch.float32, device=xy.device)
    st = _lib.lib().rl4co_augment_symmetric_f32(_ptr(xy), _ptr(_dev(cos_phi, torch.float32, "cos_phi")),
                                                _ptr(_dev(sin_phi, torch.float32, "sin_phi")), _ptr(_u8(swap_axes, "swap_axes")),
                                                b, rows // b, n, float(offset), _ptr(out), _stream())
    _lib.check(st, "rl4co_augment_symmetric_f32")
    return out


def pomo_best(reward: Tensor, actions: Tensor | None, num_augment: int, num_starts: int) -> dict:
    """Best start per augmentation and best augmentation per instance of a multistart rollout over an augmented batch
    (rows (s * A + a) * B + b), with the selected action rows, in one launch (zoo/pomo/model.py:112-140)."""
    a, s = int(num_augment), int(num_starts)
    rows = reward.numel()
    assert rows % (a * s) == 0
    b = rows // (a * s)
    reward = _dev(reward.reshape(-1), torch.float32, "reward")
    dev = reward.device
    out = {"max_reward": torch.empty((b, a), dtype=torch.float32, device=dev),
           "best_start": torch.empty((b, a), dtype=torch.int64, device=dev),
           "max_aug_reward": torch.empty(b, dtype=torch.float32, device=dev),
           "best_aug": torch.empty(b, dtype=torch.int64, device=dev)}
    t = 0
    if actions is not None:
        actions = _dev(actions, torch.int64, "actions")
        assert actions.shape[0] == rows
        t = actions.shape[1]
        out["best_multistart_actions"] = torch.empty((b, a, t), dtype=torch.int64, device=dev)
        out["best_aug_actions"] = torch.empty((b, t), dtype=torch.int64, device=dev)
    st = _lib.lib().rl4co_pomo_best(_ptr(reward), _ptr(actions), a, s, b, t, _ptr(out["max_reward"]), _ptr(out["best_start"]),
                                    _ptr(out["max_aug_reward"]), _ptr(out["best_aug"]), _ptr(out.get("best_multistart_actions")),
                                    _ptr(out.get("best_aug_actions")), _stream())
    _lib.check(st, "rl4co_pomo_best")
    return out
