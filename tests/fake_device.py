"""TEST INFRASTRUCTURE: run the host-side mirror (rl4co_amd.policy / rl4co_amd.envs) without a GPU.

The product has no CPU path (rl4co_amd.kernels refuses non-CUDA tensors). To cover the HOST logic
(decode-type parsing, multistart layout, horizon handling, select_best, teacher-forced
log-likelihood, output dict) in the CPU suite, this fixture swaps the kernel front end for the
C oracle — the test plays the device. Never used outside tests/.
"""
from __future__ import annotations

import pytest
import torch

from oracle import c_oracle


def _tour_length(locs, actions, prepend_depot=False, negate=False, horizon=None):
    if horizon is not None:  # device-side horizon: on this stand-in "device" the count is simply readable
        steps, t_add = horizon
        actions = actions[:, : min(actions.shape[1], int(t_add) + int(steps[0]))]
    return c_oracle.tour_length(locs.contiguous(), actions.contiguous(), prepend_depot, negate)


def _tsp_step(action, action_mask, first_node, current_node, step_i, done, err=None):
    c_oracle.tsp_step(action.contiguous(), action_mask, first_node, current_node, step_i, done)


def _cvrp_step(action, demand, used_capacity, vehicle_capacity, visited, current_node, action_mask, done, err=None):
    c_oracle.cvrp_step(None if action is None else action.contiguous(), demand, used_capacity, vehicle_capacity,
                       visited, current_node, action_mask, done)


def _op_step(action, locs, max_length, tour_length, visited, current_node, step_i, action_mask, done, err=None):
    c_oracle.op_step(None if action is None else action.contiguous(), locs, max_length, tour_length, visited, current_node,
                     step_i, action_mask, done)


def _op_check(actions, locs, max_length, err):
    from oracle import reference_torch as R

    s = actions.shape[0] // locs.shape[0]
    td = {"locs": R.batchify(locs, s) if s > 1 else locs, "max_length": R.batchify(max_length, s) if s > 1 else max_length}
    try:
        R.OPEnv.check_solution_validity(td, actions)
    except AssertionError as e:
        err |= 64 if "Duplicates" in str(e) else 128


def _pctsp_step(action, real_prize, cur_total_prize, visited, current_node, step_i, action_mask, done, err=None):
    c_oracle.pctsp_step(None if action is None else action.contiguous(), real_prize, cur_total_prize, visited, current_node,
                        step_i, action_mask, done)


def _pctsp_check(actions, real_prize, err):
    from oracle import reference_torch as R

    s = actions.shape[0] // real_prize.shape[0]
    rp = R.batchify(real_prize, s) if s > 1 else real_prize
    try:
        R.PCTSPEnv.check_solution_validity({"real_prize": rp, "locs": rp[..., None]}, actions)
    except AssertionError as e:
        err |= 64 if "Duplicates" in str(e) else 256


def _cvrptw_step(action, demand, locs, time_windows, durations, used_capacity, vehicle_capacity, current_time, visited,
                 current_node, action_mask, done, err=None):
    c_oracle.cvrptw_step(None if action is None else action.contiguous(), demand, locs, time_windows, durations,
                         used_capacity, vehicle_capacity, current_time, visited, current_node, action_mask, done)


def _cvrptw_check(actions, locs, time_windows, durations, err):
    from oracle import reference_torch as R

    s = actions.shape[0] // locs.shape[0]
    rep = (lambda x: R.batchify(x, s)) if s > 1 else (lambda x: x)
    td = {"locs": rep(locs), "time_windows": rep(time_windows), "durations": rep(durations)}
    orig = R.CVRPEnv.__dict__["check_solution_validity"]  # the staticmethod object itself
    R.CVRPEnv.check_solution_validity = staticmethod(lambda td, actions: None)  # the CVRP part has its own stand-in
    try:
        R.CVRPTWEnv.check_solution_validity(td, actions)
    except AssertionError as e:
        msg = str(e)
        err |= (4096 if "Time windows" in msg else 8192 if "get back" in msg else 16384 if "durations" in msg
                else 32768 if "unfeasible" in msg else 65536)
    finally:
        R.CVRPEnv.check_solution_validity = orig


def _pdp_step(action, available, to_deliver, current_node, step_i, action_mask, done, err=None):
    c_oracle.pdp_step(None if action is None else action.contiguous(), available, to_deliver, current_node, step_i,
                      action_mask, done)


def _pdp_check(actions, num_nodes, force_start_at_depot, err):
    from oracle import reference_torch as R

    env = R.PDPEnv(num_loc=num_nodes - 1, force_start_at_depot=force_start_at_depot)
    try:
        env.check_solution_validity({}, actions)
    except AssertionError as e:
        err |= 512 if "Not visiting" in str(e) else (1024 if "depot" in str(e) else 2048)


def _select_start_nodes(batch, num_starts, num_loc, has_depot, device):
    return torch.arange(num_starts).repeat_interleave(batch) % num_loc + (1 if has_depot else 0)


def _tsp_check(actions, num_nodes, err):
    n = actions.shape[1]
    ok = n == num_nodes and bool((actions.sort(1).values == torch.arange(n)).all())
    if not ok:
        err |= 4


def _cvrp_check(actions, demand, vehicle_capacity, err):
    from oracle import reference_torch as R

    s = actions.shape[0] // demand.shape[0]
    td = {"demand": R.batchify(demand, s) if s > 1 else demand,
          "vehicle_capacity": (R.batchify(vehicle_capacity, s) if s > 1 else vehicle_capacity).reshape(-1, 1)}
    try:
        R.CVRPEnv.check_solution_validity(td, actions)
    except AssertionError as e:  # map the message back to the sticky bit
        err |= 8 if "capacity" in str(e) else 4


def _am_decode(cache, state, **kw):
    from rl4co_amd import _lib

    variant = {"auto": 0, "stream": 1, "lds": 2, "wide": 3, "ms": 4}[kw.pop("variant", "auto")]
    kw.pop("philox_seed_dev", None)
    dt = _lib.dtype_id(cache.kvl.dtype)
    b = state["action_mask"].shape[0]
    groups = _lib.decode_row_groups(cache.num_nodes, dt, kw["max_steps"], variant, b, cache.num_instances)  # host-only
    if groups == 0:  # the MFMA multistart variant has no specified-order oracle: mirror the streaming one
        groups = _lib.decode_row_groups(cache.num_nodes, dt, kw["max_steps"], 1, b, cache.num_instances)
    c_oracle.am_decode(cache, state, row_groups=groups, **kw)


def _env_replay(env_name, state, actions, rem_base, err=None, mask_bits=False):
    """kernels.env_replay on CPU tensors: the tabulation between T calls of the (patched) step functions."""
    from rl4co_amd import kernels as K

    b, t_len = actions.shape
    n = state["action_mask"].shape[1]
    out = {"masks": torch.empty((b, t_len, n), dtype=torch.bool), "prev": torch.empty((b, t_len), dtype=torch.int64)}
    if env_name == "tsp":
        out["first"] = torch.empty((b, t_len), dtype=torch.int64)
        out["use_placeholder"] = torch.empty((b, t_len), dtype=torch.bool)
    elif env_name != "pdp":
        out["rem"] = torch.empty((b, t_len), dtype=torch.float32)
        if env_name == "cvrptw":
            out["now"] = torch.empty((b, t_len), dtype=torch.float32)
    scalar = {"cvrp": "used_capacity", "cvrptw": "used_capacity", "op": "tour_length", "pctsp": "cur_total_prize"}.get(env_name)
    e = err if err is not None else torch.zeros(1, dtype=torch.int32)
    for t in range(t_len):
        out["masks"][:, t] = state["action_mask"]
        out["prev"][:, t] = state["current_node"]
        if env_name == "tsp":
            out["first"][:, t] = state["first_node"]
            out["use_placeholder"][:, t] = state["i"] < 1
        elif env_name != "pdp":
            r = rem_base - state[scalar]
            out["rem"][:, t] = torch.clamp(r, min=0) if env_name == "pctsp" else r
            if env_name == "cvrptw":
                out["now"][:, t] = state["current_time"]
        a = actions[:, t].contiguous()
        if env_name == "tsp":
            K.tsp_step(a, state["action_mask"], state["first_node"], state["current_node"], state["i"], state["done"], e)
        elif env_name == "op":
            K.op_step(a, state["locs"], state["max_length"], state["tour_length"], state["visited"], state["current_node"],
                      state["i"], state["action_mask"], state["done"], e)
        elif env_name == "cvrptw":
            K.cvrptw_step(a, state["demand"], state["locs"], state["time_windows"], state["durations"], state["used_capacity"],
                          state["vehicle_capacity"], state["current_time"], state["visited"], state["current_node"],
                          state["action_mask"], state["done"], e)
        elif env_name == "pdp":
            K.pdp_step(a, state["available"], state["to_deliver"], state["current_node"], state["i"], state["action_mask"],
                       state["done"], e)
        elif env_name == "pctsp":
            K.pctsp_step(a, state["real_prize"], state["cur_total_prize"], state["visited"], state["current_node"], state["i"],
                         state["action_mask"], state["done"], e)
        else:
            K.cvrp_step(a, state["demand"], state["used_capacity"], state["vehicle_capacity"], state["visited"],
                        state["current_node"], state["action_mask"], state["done"], e)
    return out


@pytest.fixture
def cpu_device(monkeypatch):
    """Patch rl4co_amd.kernels so that policy/env host code runs on CPU tensors via the C oracle."""
    from rl4co_amd import kernels as K

    monkeypatch.setattr(K, "tour_length", _tour_length)
    monkeypatch.setattr(K, "tsp_step", _tsp_step)
    monkeypatch.setattr(K, "cvrp_step", _cvrp_step)
    monkeypatch.setattr(K, "select_start_nodes", _select_start_nodes)
    monkeypatch.setattr(K, "tsp_check_solution", _tsp_check)
    monkeypatch.setattr(K, "cvrp_check_solution", _cvrp_check)
    monkeypatch.setattr(K, "op_step", _op_step)
    monkeypatch.setattr(K, "op_max_length", c_oracle.op_max_length)
    monkeypatch.setattr(K, "gather_sum", lambda values, actions: c_oracle.gather_sum(values.contiguous(), actions.contiguous()))
    monkeypatch.setattr(K, "op_check_solution", _op_check)
    monkeypatch.setattr(K, "cvrptw_step", _cvrptw_step)
    monkeypatch.setattr(K, "cvrptw_check_solution", _cvrptw_check)
    monkeypatch.setattr(K, "pdp_step", _pdp_step)
    monkeypatch.setattr(K, "pdp_check_solution", _pdp_check)
    monkeypatch.setattr(K, "pctsp_step", _pctsp_step)
    monkeypatch.setattr(K, "pctsp_check_solution", _pctsp_check)
    monkeypatch.setattr(K, "am_decode", _am_decode)
    monkeypatch.setattr(K, "env_replay", _env_replay)
    monkeypatch.setattr(K, "augment_dihedral8", c_oracle.augment_dihedral8)
    monkeypatch.setattr(K, "augment_symmetric", c_oracle.augment_symmetric)
    monkeypatch.setattr(K, "pomo_best", c_oracle.pomo_best)
    return "cpu"
