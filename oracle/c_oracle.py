"""ORACLE (test infrastructure): ctypes front end of ``oracle/rollout_ref.c``.

Builds ``oracle/_build/librollout_ref.so`` with gcc on first use. Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may import this module.
All tensors are CPU torch tensors; the argument block is the product's own
``rl4co_am_decode_args`` (same struct, host pointers).
"""
from __future__ import annotations

import ctypes as C
import hashlib
import subprocess
from functools import lru_cache
from pathlib import Path

import torch
from torch import Tensor

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent
SRC = HERE / "rollout_ref.c"
DEPS = [SRC, ROOT / "include" / "rl4co_amd.h", ROOT / "rl4co_amd" / "csrc" / "rl4co_math.h"]
OUT = HERE / "_build" / "librollout_ref.so"
STAMP = HERE / "_build" / "librollout_ref.so.hash"
CFLAGS = ["-O2", "-std=c11", "-ffp-contract=off", "-mfma", "-fPIC", "-shared"]


def _hash() -> str:
    h = hashlib.sha256()
    for p in DEPS:
        h.update(p.read_bytes())
    h.update(" ".join(CFLAGS).encode())
    return h.hexdigest()


def build(force: bool = False) -> Path:
    if not force and OUT.exists() and STAMP.exists() and STAMP.read_text().strip() == _hash():
        return OUT
    OUT.parent.mkdir(parents=True, exist_ok=True)
    cmd = ["gcc", *CFLAGS, "-o", str(OUT), str(SRC), "-lm"]
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError(f"gcc failed:\n{proc.stderr}")
    STAMP.write_text(_hash() + "\n")
    return OUT


@lru_cache(maxsize=None)
def lib() -> C.CDLL:
    from rl4co_amd._lib import AmDecodeArgs  # the struct definition is shared with the product header

    h = C.CDLL(str(build()))
    vp, i = C.c_void_p, C.c_int
    h.oracle_tour_length_f32.argtypes = [vp, vp, i, i, i, i, i, i, vp]
    h.oracle_tsp_step.argtypes = [vp] * 6 + [i, i]
    h.oracle_cvrp_step.argtypes = [vp] * 8 + [i, i, i]
    h.oracle_op_step.argtypes = [vp] * 9 + [i, i, i]
    h.oracle_pctsp_step.argtypes = [vp] * 8 + [i, i, i]
    h.oracle_pdp_step.argtypes = [vp] * 7 + [i, i]
    h.oracle_cvrptw_step.argtypes = [vp] * 12 + [i, i, i]
    h.oracle_op_max_length.argtypes = [vp, vp, i, i, vp]
    h.oracle_gather_sum_f32.argtypes = [vp, vp, i, i, i, i, vp]
    h.oracle_am_decode.argtypes = [C.POINTER(AmDecodeArgs), i]
    h.oracle_am_decode_ms.argtypes = [C.POINTER(AmDecodeArgs)]
    for name in ("oracle_expf", "oracle_logf", "oracle_tanhf"):
        fn = getattr(h, name)
        fn.argtypes = [C.c_float]
        fn.restype = C.c_float
    h.oracle_math_array.argtypes = [i, vp, C.c_int64, vp]
    h.oracle_uniform_f32.argtypes = [vp, C.c_int64, C.c_float, C.c_float, C.c_uint64, C.c_uint32, i, C.c_float]
    h.oracle_exp1_noise.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32]
    h.oracle_exp1_noise.restype = C.c_float
    return h


def _cpu(t: Tensor, dtype=None) -> Tensor:
    assert t.device.type == "cpu" and t.is_contiguous(), "oracle tensors must be contiguous CPU tensors"
    if dtype is not None:
        assert t.dtype == dtype, (t.dtype, dtype)
    return t


def _p(t):
    return None if t is None else t.data_ptr()


def tour_length(locs: Tensor, actions: Tensor, prepend_depot=False, negate=False) -> Tensor:
    _cpu(locs, torch.float32), _cpu(actions, torch.int64)
    b, t = actions.shape
    b_locs, n, _ = locs.shape
    out = torch.empty(b, dtype=torch.float32)
    st = lib().oracle_tour_length_f32(_p(locs), _p(actions), b, b_locs, n, t, int(prepend_depot), int(negate), _p(out))
    assert st == 0
    return out


def tsp_step(action, mask, first, cur, step_i, done) -> None:
    b, n = mask.shape
    st = lib().oracle_tsp_step(_p(_cpu(action, torch.int64)), _p(_u8(mask)), _p(first), _p(cur), _p(step_i),
                               _p(_u8(done)), b, n)
    assert st == 0


def cvrp_step(action, demand, used, cap, visited, cur, mask, done) -> None:
    b, n = mask.shape
    st = lib().oracle_cvrp_step(_p(action), _p(_cpu(demand, torch.float32)), _p(used), _p(cap), _p(_u8(visited)),
                                _p(cur), _p(_u8(mask)), _p(None if done is None else _u8(done)), b, demand.shape[0], n)
    assert st == 0


def op_max_length(locs: Tensor, max_length: Tensor) -> Tensor:
    b, n, _ = locs.shape
    out = torch.empty((b, n), dtype=torch.float32)
    st = lib().oracle_op_max_length(_p(_cpu(locs, torch.float32)), _p(_cpu(max_length.reshape(-1).contiguous(), torch.float32)),
                                    b, n, _p(out))
    assert st == 0
    return out


def op_step(action, locs, max_length, tour_length, visited, cur, step_i, mask, done) -> None:
    b, n = mask.shape
    st = lib().oracle_op_step(_p(None if action is None else _cpu(action, torch.int64)), _p(_cpu(locs, torch.float32)),
                              _p(_cpu(max_length, torch.float32)), _p(_cpu(tour_length, torch.float32)), _p(_u8(visited)),
                              _p(_cpu(cur, torch.int64)), _p(_cpu(step_i, torch.int64)), _p(_u8(mask)), _p(_u8(done)),
                              b, locs.shape[0], n)
    assert st == 0, "oracle_op_step: action out of range"


def cvrptw_step(action, demand, locs, time_windows, durations, used, cap, current_time, visited, cur, mask, done) -> None:
    b, n = mask.shape
    st = lib().oracle_cvrptw_step(_p(None if action is None else _cpu(action, torch.int64)), _p(_cpu(demand, torch.float32)),
                                  _p(_cpu(locs, torch.float32)), _p(_cpu(time_windows, torch.float32)),
                                  _p(_cpu(durations, torch.float32)), _p(_cpu(used, torch.float32)), _p(_cpu(cap, torch.float32)),
                                  _p(_cpu(current_time, torch.float32)), _p(_u8(visited)), _p(_cpu(cur, torch.int64)),
                                  _p(_u8(mask)), _p(None if done is None else _u8(done)), b, demand.shape[0], n)
    assert st == 0, "oracle_cvrptw_step: action out of range"


def pdp_step(action, available, to_deliver, cur, step_i, mask, done) -> None:
    b, n = mask.shape
    st = lib().oracle_pdp_step(_p(None if action is None else _cpu(action, torch.int64)), _p(_u8(available)),
                               _p(_u8(to_deliver)), _p(_cpu(cur, torch.int64)), _p(_cpu(step_i, torch.int64)),
                               _p(_u8(mask)), _p(_u8(done)), b, n)
    assert st == 0, "oracle_pdp_step: action out of range"


def pctsp_step(action, real_prize, cur_total_prize, visited, cur, step_i, mask, done) -> None:
    b, n = mask.shape
    st = lib().oracle_pctsp_step(_p(None if action is None else _cpu(action, torch.int64)), _p(_cpu(real_prize, torch.float32)),
                                 _p(_cpu(cur_total_prize, torch.float32)), _p(_u8(visited)), _p(_cpu(cur, torch.int64)),
                                 _p(_cpu(step_i, torch.int64)), _p(_u8(mask)), _p(_u8(done)), b, real_prize.shape[0], n)
    assert st == 0, "oracle_pctsp_step: action out of range"


def gather_sum(values: Tensor, actions: Tensor) -> Tensor:
    b, t = actions.shape
    out = torch.empty((b,), dtype=torch.float32)
    st = lib().oracle_gather_sum_f32(_p(_cpu(values, torch.float32)), _p(_cpu(actions, torch.int64)), b, values.shape[0],
                                     values.shape[1], t, _p(out))
    assert st == 0
    return out


def _u8(t: Tensor) -> Tensor:
    _cpu(t)
    return t.view(torch.uint8) if t.dtype == torch.bool else t


def am_decode(cache, state: dict, *, mode: str, max_steps: int, actions: Tensor, logps: Tensor, err: Tensor,
              row_groups, t0: int = 0, tanh_clipping: float = 10.0, temperature: float = 1.0,
              mask_inner: bool = True, mask_logits: bool = True, exp_noise: Tensor | None = None,
              philox_seed: int = 0, philox_offset: int = 0, forced_actions: Tensor | None = None,
              all_logps: Tensor | None = None, entropy: Tensor | None = None, n_steps: Tensor | None = None,
              steps_summary: Tensor | None = None) -> None:
    """Mirror of ``rl4co_amd.kernels.am_decode`` for CPU tensors, run by the C oracle.

    ``cache`` is a ``rl4co_amd.cache.FoldedCache`` whose tensors live on the CPU (bf16 planes are
    read as raw uint16)."""
    from rl4co_amd import _lib

    a = _lib.AmDecodeArgs()
    mask = _u8(state["action_mask"])
    b, n = mask.shape
    a.env = {"tsp": _lib.ENV_TSP, "cvrp": _lib.ENV_CVRP, "op": _lib.ENV_OP, "pctsp": _lib.ENV_PCTSP, "pdp": _lib.ENV_PDP, "cvrptw": _lib.ENV_CVRPTW}[cache.env_name]
    a.B, a.B_inst, a.N = b, cache.num_instances, n
    a.mode = {"greedy": 0, "sampling": 1, "evaluate": 2}[mode]
    a.max_steps = int(max_steps)
    a.mask_inner, a.mask_logits = int(mask_inner), int(mask_logits)
    a.tanh_clipping, a.temperature = float(tanh_clipping), float(temperature)
    kvl = _cpu(cache.kvl)
    a.cache_dtype = _lib.dtype_id(kvl.dtype)
    a.glimpse_key, a.glimpse_val, a.logit_key = (cache.plane(i).data_ptr() for i in range(3))
    a.kvl_row_stride, a.kvl_batch_stride = cache.row_stride, cache.batch_stride
    if getattr(cache, "unfold", False):
        a.unfold, a.ctx_width = 1, cache.w_ctx_t.shape[0]
        a.node_embed, a.w_ctx_t, a.w_out_t = (_p(_cpu(x, torch.float32)) for x in (cache.node_embed, cache.w_ctx_t, cache.w_out_t))
        a.w_placeholder = _p(None if cache.w_placeholder is None else _cpu(cache.w_placeholder, torch.float32))
    else:
        # (16-bit context tables of the 16-bit regime: widened exactly, as the kernels do on load; the locals keep them alive)
        ctx_cur32 = cache.ctx_cur.float().contiguous()
        a.ctx_cur = _p(_cpu(ctx_cur32, torch.float32))
    a.q_bias = _p(None if cache.q_bias is None else _cpu(cache.q_bias, torch.float32))
    a.action_mask = _p(mask)
    a.current_node = _p(_cpu(state["current_node"], torch.int64))
    a.done = _p(_u8(state["done"]))
    if cache.env_name == "tsp":
        if not a.unfold:
            ctx_first32 = cache.ctx_first.float().contiguous()
            a.ctx_first = _p(_cpu(ctx_first32, torch.float32))
            a.q_step0 = _p(_cpu(cache.q_step0, torch.float32))
        a.first_node = _p(_cpu(state["first_node"], torch.int64))
        a.step_i = _p(_cpu(state["i"], torch.int64))
    elif cache.env_name == "pdp":
        a.visited = _p(_u8(state["available"]))
        a.to_deliver = _p(_u8(state["to_deliver"]))
        a.step_i = _p(_cpu(state["i"], torch.int64))
    elif cache.env_name == "pctsp":
        a.w_cap = _p(_cpu(cache.w_cap, torch.float32))
        a.demand = _p(_cpu(state["real_prize"], torch.float32))
        a.used_capacity = _p(_cpu(state["cur_total_prize"], torch.float32))
        a.vehicle_capacity = _p(_cpu(state["prize_required"], torch.float32))
        a.step_i = _p(_cpu(state["i"], torch.int64))
        a.visited = _p(_u8(state["visited"]))
    elif cache.env_name == "op":
        a.w_cap = _p(_cpu(cache.w_cap, torch.float32))
        a.locs = _p(_cpu(state["locs"], torch.float32))
        a.max_length = _p(_cpu(state["max_length"], torch.float32))
        a.used_capacity = _p(_cpu(state["tour_length"], torch.float32))
        a.step_i = _p(_cpu(state["i"], torch.int64))
        a.visited = _p(_u8(state["visited"]))
    else:
        if cache.env_name == "cvrptw":
            a.w_time = _p(_cpu(cache.w_time, torch.float32))
            a.locs = _p(_cpu(state["locs"], torch.float32))
            a.time_windows = _p(_cpu(state["time_windows"], torch.float32))
            a.durations = _p(_cpu(state["durations"], torch.float32))
            a.current_time = _p(_cpu(state["current_time"], torch.float32))
        if not a.unfold:
            a.w_cap = _p(_cpu(cache.w_cap, torch.float32))
        a.demand = _p(_cpu(state["demand"], torch.float32))
        a.used_capacity = _p(_cpu(state["used_capacity"], torch.float32))
        a.vehicle_capacity = _p(_cpu(state["vehicle_capacity"], torch.float32))
        a.visited = _p(_u8(state["visited"]))
    a.exp_noise = _p(None if exp_noise is None else _cpu(exp_noise, torch.float32))
    a.philox_seed, a.philox_offset = int(philox_seed), int(philox_offset)
    a.forced_actions = _p(None if forced_actions is None else _cpu(forced_actions, torch.int64))
    a.t0, a.out_stride = int(t0), actions.shape[1]
    a.actions, a.logps = _p(_cpu(actions, torch.int64)), _p(_cpu(logps, torch.float32))
    a.all_logps = _p(None if all_logps is None else _cpu(all_logps, torch.float32))
    a.entropy = _p(None if entropy is None else _cpu(entropy, torch.float32))
    a.n_steps = _p(None if n_steps is None else _cpu(n_steps, torch.int32))
    if steps_summary is not None:
        assert steps_summary.numel() >= 4 and steps_summary.data_ptr() % 8 == 0, "steps_summary: 4 int32 words, 8-byte aligned"
    a.steps_summary = _p(None if steps_summary is None else _cpu(steps_summary, torch.int32))
    a.err = _p(_cpu(err, torch.int32))
    if row_groups == "ms":  # rounding-model oracle of the multistart MFMA variant (bf16 query / numerators / glimpse)
        st = lib().oracle_am_decode_ms(C.byref(a))
    else:
        st = lib().oracle_am_decode(C.byref(a), int(row_groups))
    assert st == 0, "oracle_am_decode rejected its arguments"


def math_array(fn: str, x: Tensor) -> Tensor:
    """exp / log / tanh of the shared deterministic header over a whole tensor (tests/test_math.py)."""
    _cpu(x, torch.float32)
    y = torch.empty_like(x)
    assert lib().oracle_math_array({"exp": 0, "log": 1, "tanh": 2}[fn], _p(x), x.numel(), _p(y)) == 0
    return y


def uniform(shape, low: float, high: float, seed: int, stream_id: int, demand_capacity: float | None = None) -> Tensor:
    """Host restatement of kernels.uniform (rl4co_uniform_f32): the same Philox words."""
    out = torch.empty(tuple(shape), dtype=torch.float32)
    assert lib().oracle_uniform_f32(_p(out), out.numel(), float(low), float(high), int(seed) & ((1 << 64) - 1), int(stream_id),
                                    0 if demand_capacity is None else 1, float(demand_capacity or 1.0)) == 0
    return out


def augment_dihedral8(xy: Tensor) -> Tensor:
    """Host restatement of kernels.augment_dihedral8."""
    b, n, _ = xy.shape
    out = torch.empty((8 * b, n, 2), dtype=torch.float32)
    h = lib()
    h.oracle_augment_dihedral8_f32.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    assert h.oracle_augment_dihedral8_f32(_p(_cpu(xy, torch.float32)), b, n, _p(out)) == 0
    return out


def augment_symmetric(xy: Tensor, cos_phi: Tensor, sin_phi: Tensor, swap_axes: Tensor, offset: float = 0.5) -> Tensor:
    """Host restatement of kernels.augment_symmetric."""
    b, n, _ = xy.shape
    rows = cos_phi.numel()
    out = torch.empty((rows, n, 2), dtype=torch.float32)
    h = lib()
    h.oracle_augment_symmetric_f32.argtypes = [C.c_void_p] * 4 + [C.c_int] * 3 + [C.c_float, C.c_void_p]
    assert h.oracle_augment_symmetric_f32(_p(_cpu(xy, torch.float32)), _p(_cpu(cos_phi, torch.float32)), _p(_cpu(sin_phi, torch.float32)),
                                          _p(_u8(swap_axes)), b, rows // b, n, float(offset), _p(out)) == 0
    return out


def pomo_best(reward: Tensor, actions: Tensor | None, num_augment: int, num_starts: int) -> dict:
    """Host restatement of kernels.pomo_best."""
    a, s = int(num_augment), int(num_starts)
    reward = _cpu(reward.reshape(-1).contiguous(), torch.float32)
    b = reward.numel() // (a * s)
    out = {"max_reward": torch.empty((b, a)), "best_start": torch.empty((b, a), dtype=torch.int64),
           "max_aug_reward": torch.empty(b), "best_aug": torch.empty(b, dtype=torch.int64)}
    t = 0
    if actions is not None:
        _cpu(actions, torch.int64)
        t = actions.shape[1]
        out["best_multistart_actions"] = torch.empty((b, a, t), dtype=torch.int64)
        out["best_aug_actions"] = torch.empty((b, t), dtype=torch.int64)
    h = lib()
    h.oracle_pomo_best.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p] * 6
    assert h.oracle_pomo_best(_p(reward), _p(actions), a, s, b, t, _p(out["max_reward"]), _p(out["best_start"]), _p(out["max_aug_reward"]),
                              _p(out["best_aug"]), _p(out.get("best_multistart_actions")), _p(out.get("best_aug_actions"))) == 0
    return out
