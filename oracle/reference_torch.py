"""ORACLE (test infrastructure, not product): CPU restatement of the ai4co/rl4co rollout hot path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module. Nothing under ``rl4co_amd/`` does.

The reference's hot path is 100 % stock ATen ops; the third-party packages it needs and that are
absent here (tensordict, torchrl, lightning, hydra) are containers and loop glue, not arithmetic
(SURVEY.md §8c). This file restates the path op-for-op — same ATen calls, same order — with the
state held in a plain ``dict`` instead of a TensorDict. Each function cites the reference
file:line it follows (paths relative to the reference checkout).

Pinning: ``oracle/gen_golden.py`` imports the reference's own source files verbatim (through the
test-only stand-ins in ``oracle/shims``) and checks this restatement against them bit-for-bit;
the vectors it emits are committed under ``tests/golden``. See DESIGN.md §3.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Callable

import torch
import torch.nn as nn
import torch.nn.functional as F
from einops import rearrange
from torch import Tensor

# ----------------------------------------------------------------------------------------------
# rl4co/utils/ops.py
# ----------------------------------------------------------------------------------------------


def _batchify_single(x: Tensor, repeats: int) -> Tensor:
    """ops.py:10-13"""
    s = x.shape
    return x.expand(repeats, *s).contiguous().view(s[0] * repeats, *s[1:])


def batchify(x, shape):
    """ops.py:16-30 (dict = TensorDict stand-in: applied per entry)"""
    if isinstance(x, dict):
        return {k: batchify(v, shape) for k, v in x.items()}
    shape = [shape] if isinstance(shape, int) else shape
    for s in reversed(shape):
        x = _batchify_single(x, s) if s > 0 else x
    return x


def _unbatchify_single(x: Tensor, repeats: int) -> Tensor:
    """ops.py:33-36"""
    s = x.shape
    return x.view(repeats, s[0] // repeats, *s[1:]).permute(1, 0, *range(2, len(s) + 1))


def unbatchify(x, shape):
    """ops.py:39-51"""
    if isinstance(x, dict):
        return {k: unbatchify(v, shape) for k, v in x.items()}
    shape = [shape] if isinstance(shape, int) else shape
    for s in reversed(shape):
        x = _unbatchify_single(x, s) if s > 0 else x
    return x


def gather_by_index(src: Tensor, idx: Tensor, dim: int = 1, squeeze: bool = True) -> Tensor:
    """ops.py:54-66"""
    expanded_shape = list(src.shape)
    expanded_shape[dim] = -1
    idx = idx.view(idx.shape + (1,) * (src.dim() - idx.dim())).expand(expanded_shape)
    squeeze = idx.size(dim) == 1 and squeeze
    return src.gather(dim, idx).squeeze(dim) if squeeze else src.gather(dim, idx)


def unbatchify_and_gather(x, idx: Tensor, n: int):
    """ops.py:69-74"""
    if isinstance(x, dict):
        return {k: unbatchify_and_gather(v, idx, n) for k, v in x.items()}
    x = unbatchify(x, n)
    return gather_by_index(x, idx, dim=idx.dim())


def get_distance(x: Tensor, y: Tensor) -> Tensor:
    """ops.py:77-79"""
    return (x - y).norm(p=2, dim=-1)


def get_tour_length(ordered_locs: Tensor) -> Tensor:
    """ops.py:82-90"""
    ordered_locs_next = torch.roll(ordered_locs, -1, dims=-2)
    return get_distance(ordered_locs_next, ordered_locs).sum(-1)


def calculate_entropy(logprobs: Tensor) -> Tensor:
    """ops.py:103-111"""
    logprobs = torch.nan_to_num(logprobs, nan=0.0)
    entropy = -(logprobs.exp() * logprobs).sum(dim=-1)
    entropy = entropy.sum(dim=1)
    assert entropy.isfinite().all(), "Entropy is not finite"
    return entropy


def get_num_starts(td: dict, env_name: str | None = None) -> int:
    """ops.py:115-125"""
    num_starts = td["action_mask"].shape[-1]
    if env_name == "pdp":
        num_starts = (num_starts - 1) // 2
    elif env_name in ["cvrp", "cvrptw", "sdvrp", "mtsp", "op", "pctsp", "spctsp"]:
        num_starts = num_starts - 1
    return num_starts


def select_start_nodes(td: dict, env, num_starts: int) -> Tensor:
    """ops.py:128-161 (tsp / depot branches)"""
    num_loc = env.num_loc
    batch = td["action_mask"].shape[0]
    device = td["action_mask"].device
    if env.name in ["tsp", "atsp", "flp", "mcp"]:
        return torch.arange(num_starts, device=device).repeat_interleave(batch) % num_loc
    if env.name == "pdp":  # pdp/env.py:216-225: only the pickups can start a tour
        return torch.arange(num_starts, device=device).repeat_interleave(batch) % (num_loc // 2) + 1
    selected = torch.arange(num_starts, device=device).repeat_interleave(batch) % num_loc + 1
    if env.name == "op" and bool((td["action_mask"][..., 1:].float().sum(-1) < num_starts).any()):
        # ops.py:150-160: some customers cannot be entered at all (too far for max_length): the start nodes are
        # resampled from the feasible ones, with replacement, "b n -> (n b)"
        selected = torch.multinomial(td["action_mask"][..., 1:].float(), num_starts, replacement=True) + 1
        selected = selected.t().reshape(-1)
    return selected


# ----------------------------------------------------------------------------------------------
# environments: rl4co/envs/routing/{tsp,cvrp}/{env,generator}.py, rl4co/envs/common/base.py
# ----------------------------------------------------------------------------------------------

CAPACITIES = {  # cvrp/generator.py:15-30
    10: 20.0, 15: 25.0, 20: 30.0, 30: 33.0, 40: 37.0, 50: 40.0, 60: 43.0, 75: 45.0,
    100: 50.0, 125: 55.0, 150: 60.0, 200: 70.0, 500: 100.0, 1000: 150.0,
}


class TSPEnv:
    """envs/routing/tsp/env.py:22-164 with envs/common/base.py:121-190."""

    name = "tsp"

    def __init__(self, num_loc: int = 20, check_solution: bool = True):
        self.num_loc = num_loc
        self.check_solution = check_solution

    def generate(self, batch_size: int) -> dict:
        """tsp/generator.py:49-58: Uniform(0,1).sample((B,N,2))"""
        u = torch.distributions.Uniform(low=0.0, high=1.0)
        return {"locs": u.sample((batch_size, self.num_loc, 2))}

    def reset(self, td: dict | None = None, batch_size: int | None = None) -> dict:
        """base.py:135-143 -> tsp/env.py:88-113; torchrl's EnvBase.reset adds done/terminated."""
        if td is None:
            td = self.generate(batch_size)
        init_locs = td["locs"]
        b = init_locs.shape[0]
        device = init_locs.device
        num_loc = init_locs.shape[-2]
        current_node = torch.zeros((b,), dtype=torch.int64, device=device)
        return {
            "locs": init_locs,
            "first_node": current_node,
            "current_node": current_node,
            "i": torch.zeros((b, 1), dtype=torch.int64, device=device),
            "action_mask": torch.ones((b, num_loc), dtype=torch.bool, device=device),
            "reward": torch.zeros((b, 1), dtype=torch.float32),
            "done": torch.zeros((b, 1), dtype=torch.bool, device=device),
        }

    def step(self, td: dict) -> dict:
        """tsp/env.py:60-86"""
        current_node = td["action"]
        first_node = current_node if td["i"].all() == 0 else td["first_node"]
        available = td["action_mask"].scatter(
            -1, current_node.unsqueeze(-1).expand_as(td["action_mask"]), 0
        )
        done = torch.sum(available, dim=-1) == 0
        reward = torch.zeros_like(done)
        td.update(
            {
                "first_node": first_node,
                "current_node": current_node,
                "i": td["i"] + 1,
                "action_mask": available,
                "reward": reward,
                "done": done,
            }
        )
        return td

    def get_reward(self, td: dict, actions: Tensor, check_solution: bool | None = None) -> Tensor:
        """base.py:180-190 -> tsp/env.py:150-156"""
        check_solution = self.check_solution if check_solution is None else check_solution
        if check_solution:
            self.check_solution_validity(td, actions)
        locs_ordered = gather_by_index(td["locs"], actions)
        return -get_tour_length(locs_ordered)

    @staticmethod
    def check_solution_validity(td: dict, actions: Tensor) -> None:
        """tsp/env.py:158-164"""
        assert (
            torch.arange(actions.size(1), out=actions.data.new()).view(1, -1).expand_as(actions)
            == actions.data.sort(1)[0]
        ).all(), "Invalid tour"

    def get_num_starts(self, td):
        return get_num_starts(td, self.name)

    def select_start_nodes(self, td, num_starts):
        return select_start_nodes(td, self, num_starts)


class CVRPEnv:
    """envs/routing/cvrp/env.py:22-177"""

    name = "cvrp"

    def __init__(self, num_loc: int = 20, check_solution: bool = True, capacity: float | None = None):
        self.num_loc = num_loc
        self.check_solution = check_solution
        self.vehicle_capacity = 1.0
        if capacity is None:  # cvrp/generator.py:99-110
            capacity = CAPACITIES.get(num_loc, None)
        if capacity is None:
            closest = min(CAPACITIES.keys(), key=lambda x: abs(x - num_loc))
            capacity = CAPACITIES[closest]
        self.capacity = capacity

    def generate(self, batch_size: int) -> dict:
        """cvrp/generator.py:114-140 (depot sampled with the locations; demand U{1..9}/capacity)"""
        loc_sampler = torch.distributions.Uniform(low=0.0, high=1.0)
        demand_sampler = torch.distributions.Uniform(low=0, high=9)  # min_demand-1, max_demand-1
        locs = loc_sampler.sample((batch_size, self.num_loc + 1, 2))
        depot = locs[..., 0, :]
        locs = locs[..., 1:, :]
        demand = demand_sampler.sample((batch_size, self.num_loc))
        demand = (demand.int() + 1).float()
        capacity = torch.full((batch_size, 1), self.capacity)
        return {"locs": locs, "depot": depot, "demand": demand / self.capacity, "capacity": capacity}

    def reset(self, td: dict | None = None, batch_size: int | None = None) -> dict:
        """cvrp/env.py:98-124"""
        if td is None:
            td = self.generate(batch_size)
        b = td["locs"].shape[0]
        device = td["locs"].device
        td_reset = {
            "locs": torch.cat((td["depot"][:, None, :], td["locs"]), -2),
            "demand": td["demand"],
            "current_node": torch.zeros(b, 1, dtype=torch.long, device=device),
            "used_capacity": torch.zeros((b, 1), device=device),
            "vehicle_capacity": torch.full((b, 1), self.vehicle_capacity, device=device),
            "visited": torch.zeros((b, td["locs"].shape[-2] + 1), dtype=torch.uint8, device=device),
            "done": torch.zeros((b, 1), dtype=torch.bool, device=device),
        }
        td_reset["action_mask"] = self.get_action_mask(td_reset)
        return td_reset

    def step(self, td: dict) -> dict:
        """cvrp/env.py:66-96"""
        current_node = td["action"][:, None]
        n_loc = td["demand"].size(-1)
        selected_demand = gather_by_index(
            td["demand"], torch.clamp(current_node - 1, 0, n_loc - 1), squeeze=False
        )
        used_capacity = (td["used_capacity"] + selected_demand) * (current_node != 0).float()
        visited = td["visited"].scatter(-1, current_node, 1)
        done = visited.sum(-1) == visited.size(-1)
        reward = torch.zeros_like(done)
        td.update(
            {
                "current_node": current_node,
                "used_capacity": used_capacity,
                "visited": visited,
                "reward": reward,
                "done": done,
            }
        )
        td["action_mask"] = self.get_action_mask(td)
        return td

    @staticmethod
    def get_action_mask(td: dict) -> Tensor:
        """cvrp/env.py:126-136"""
        exceeds_cap = td["demand"] + td["used_capacity"] > td["vehicle_capacity"] + 1e-5
        mask_loc = td["visited"][..., 1:].to(exceeds_cap.dtype) | exceeds_cap
        mask_depot = (td["current_node"] == 0) & ((mask_loc == 0).int().sum(-1) > 0)[:, None]
        return ~torch.cat((mask_depot, mask_loc), -1)

    def get_reward(self, td: dict, actions: Tensor, check_solution: bool | None = None) -> Tensor:
        """base.py:180-190 -> cvrp/env.py:138-147"""
        check_solution = self.check_solution if check_solution is None else check_solution
        if check_solution:
            self.check_solution_validity(td, actions)
        locs_ordered = torch.cat(
            [td["locs"][..., 0:1, :], gather_by_index(td["locs"], actions)], dim=1
        )
        return -get_tour_length(locs_ordered)

    @staticmethod
    def check_solution_validity(td: dict, actions: Tensor) -> None:
        """cvrp/env.py:149-177"""
        batch_size, graph_size = td["demand"].size()
        sorted_pi = actions.data.sort(1)[0]
        assert (
            torch.arange(1, graph_size + 1, out=sorted_pi.data.new())
            .view(1, -1)
            .expand(batch_size, graph_size)
            == sorted_pi[:, -graph_size:]
        ).all() and (sorted_pi[:, :-graph_size] == 0).all(), "Invalid tour"
        demand_with_depot = torch.cat((-td["vehicle_capacity"], td["demand"]), 1)
        d = demand_with_depot.gather(1, actions)
        used_cap = torch.zeros_like(td["demand"][:, 0])
        for i in range(actions.size(1)):
            used_cap += d[:, i]
            used_cap[used_cap < 0] = 0
            assert (used_cap <= td["vehicle_capacity"][:, 0] + 1e-5).all(), "Used more than capacity"

    def get_num_starts(self, td):
        return get_num_starts(td, self.name)

    def select_start_nodes(self, td, num_starts):
        return select_start_nodes(td, self, num_starts)


class CVRPTWEnv(CVRPEnv):
    """envs/routing/cvrptw/env.py:16-199 with cvrptw/generator.py:13-158 (CVRP + time windows; unscaled by default:
    coordinates in [0, 150], integer-valued time windows in [0, 480], service durations 0)"""

    name = "cvrptw"

    def __init__(self, num_loc: int = 20, check_solution: bool = True, capacity: float | None = None,
                 max_loc: float = 150.0, max_time: float = 480, scale: bool = False):
        super().__init__(num_loc, check_solution, capacity)
        self.max_loc, self.max_time, self.scale = max_loc, max_time, scale

    def generate(self, batch_size: int) -> dict:
        """cvrp/generator.py:114-140 with a depot sampler (cvrptw/generator.py:49), then cvrptw/generator.py:77-158"""
        loc_sampler = torch.distributions.Uniform(low=0.0, high=self.max_loc)
        depot = loc_sampler.sample((batch_size, 2))
        locs = loc_sampler.sample((batch_size, self.num_loc, 2))
        demand = torch.distributions.Uniform(low=0, high=9).sample((batch_size, self.num_loc))
        demand = (demand.int() + 1).float()
        td = {"locs": locs, "depot": depot, "demand": demand / self.capacity,
              "capacity": torch.full((batch_size, 1), self.capacity)}
        durations = torch.zeros(batch_size, self.num_loc + 1, dtype=torch.float32)
        dist = get_distance(td["depot"], td["locs"].transpose(0, 1)).transpose(0, 1)
        dist = torch.cat((torch.zeros(batch_size, 1), dist), dim=1)
        upper_bound = self.max_time - dist - durations
        ts_1 = torch.rand(batch_size, self.num_loc + 1)
        ts_2 = torch.rand(batch_size, self.num_loc + 1)
        min_ts = (dist + (upper_bound - dist) * ts_1).int()
        max_ts = (dist + (upper_bound - dist) * ts_2).int()
        min_times = torch.min(min_ts, max_ts)
        max_times = torch.max(min_ts, max_ts)
        min_times[..., :, 0] = 0.0
        max_times[..., :, 0] = self.max_time
        mask = min_times == max_times
        if torch.any(mask):
            min_tmp = min_times.clone()
            min_tmp[mask] = torch.max(dist[mask].int(), min_tmp[mask] - 1)
            min_times = min_tmp
            mask = min_times == max_times
            if torch.any(mask):
                max_tmp = max_times.clone()
                max_tmp[mask] = torch.min(
                    torch.floor(upper_bound[mask]).int(),
                    torch.max(torch.ceil(min_tmp[mask] + durations[mask]).int(), max_tmp[mask] + 1),
                )
                max_times = max_tmp
        if self.scale:
            durations = durations / self.max_time
            min_times = min_times / self.max_time
            max_times = max_times / self.max_time
            td["depot"] = td["depot"] / self.max_time
            td["locs"] = td["locs"] / self.max_time
        time_windows = torch.stack((min_times, max_times), dim=-1)
        assert torch.all(min_times < max_times)
        durations[:, 0] = 0.0
        td.update({"durations": durations, "time_windows": time_windows})
        return td

    def reset(self, td: dict | None = None, batch_size: int | None = None) -> dict:
        """cvrptw/env.py:115-139"""
        if td is None:
            td = self.generate(batch_size)
        b = td["locs"].shape[0]
        device = td["locs"].device
        td_reset = {
            "locs": torch.cat((td["depot"][..., None, :], td["locs"]), -2),
            "demand": td["demand"],
            "current_node": torch.zeros(b, 1, dtype=torch.long, device=device),
            "current_time": torch.zeros(b, 1, dtype=torch.float32, device=device),
            "used_capacity": torch.zeros((b, 1), device=device),
            "vehicle_capacity": torch.full((b, 1), self.vehicle_capacity, device=device),
            "visited": torch.zeros((b, td["locs"].shape[-2] + 1), dtype=torch.uint8, device=device),
            "durations": td["durations"],
            "time_windows": td["time_windows"],
            "done": torch.zeros((b, 1), dtype=torch.bool, device=device),
        }
        td_reset["action_mask"] = self.get_action_mask(td_reset)
        return td_reset

    @staticmethod
    def get_action_mask(td: dict) -> Tensor:
        """cvrptw/env.py:83-95"""
        not_masked = CVRPEnv.get_action_mask(td)
        current_loc = gather_by_index(td["locs"], td["current_node"])
        dist = get_distance(current_loc[..., None, :], td["locs"])
        td.update({"current_loc": current_loc, "distances": dist})
        can_reach_in_time = td["current_time"] + dist <= td["time_windows"][..., 1]
        return not_masked & can_reach_in_time

    def step(self, td: dict) -> dict:
        """cvrptw/env.py:97-113"""
        batch_size = td["locs"].shape[0]
        distance = gather_by_index(td["distances"], td["action"]).reshape([batch_size, 1])
        duration = gather_by_index(td["durations"], td["action"]).reshape([batch_size, 1])
        start_times = gather_by_index(td["time_windows"], td["action"])[..., 0].reshape([batch_size, 1])
        td["current_time"] = (td["action"][:, None] != 0) * (
            torch.max(td["current_time"] + distance, start_times) + duration
        )
        return super().step(td)

    @staticmethod
    def check_solution_validity(td: dict, actions: Tensor) -> None:
        """cvrptw/env.py:146-190"""
        CVRPEnv.check_solution_validity(td, actions)
        batch_size = td["locs"].shape[0]
        distances = get_distance(td["locs"][..., 0, :], td["locs"].transpose(0, 1)).transpose(0, 1)
        assert torch.all(distances >= 0.0), "Distances must be non-negative."
        assert torch.all(td["time_windows"] >= 0.0), "Time windows must be non-negative."
        assert torch.all(
            td["time_windows"][..., :, 0] + distances + td["durations"] <= td["time_windows"][..., 0, 1][0]
        ), "vehicle cannot perform service and get back to depot in time."
        assert torch.all(td["durations"] >= 0.0), "Service durations must be non-negative."
        assert torch.all(td["time_windows"][..., 0] < td["time_windows"][..., 1]), "there are unfeasible time windows"
        curr_time = torch.zeros(batch_size, 1, dtype=torch.float32, device=actions.device)
        curr_node = torch.zeros_like(curr_time, dtype=torch.int64)
        for ii in range(actions.size(1)):
            next_node = actions[:, ii]
            dist = get_distance(
                gather_by_index(td["locs"], curr_node).reshape([batch_size, 2]),
                gather_by_index(td["locs"], next_node).reshape([batch_size, 2]),
            ).reshape([batch_size, 1])
            curr_time = torch.max(
                (curr_time + dist).int(),
                gather_by_index(td["time_windows"], next_node)[..., 0].reshape([batch_size, 1]),
            )
            assert torch.all(
                curr_time <= gather_by_index(td["time_windows"], next_node)[..., 1].reshape([batch_size, 1])
            ), "vehicle cannot start service before deadline"
            curr_time = curr_time + gather_by_index(td["durations"], next_node).reshape([batch_size, 1])
            curr_node = next_node
            curr_time[curr_node == 0] = 0.0


OP_MAX_LENGTHS = {20: 2.0, 50: 3.0, 100: 4.0}  # op/generator.py:13


class OPEnv:
    """envs/routing/op/env.py:16-182 (orienteering: collect prizes, be back at the depot within max_length)"""

    name = "op"

    def __init__(self, num_loc: int = 20, check_solution: bool = True, max_length: float | None = None,
                 prize_type: str = "dist"):
        self.num_loc = num_loc
        self.check_solution = check_solution
        self.prize_type = prize_type
        if max_length is None:  # op/generator.py:90-101
            max_length = OP_MAX_LENGTHS.get(num_loc, None)
        if max_length is None:
            closest = min(OP_MAX_LENGTHS.keys(), key=lambda x: abs(x - num_loc))
            max_length = OP_MAX_LENGTHS[closest]
        self.max_length = max_length

    def generate(self, batch_size: int) -> dict:
        """op/generator.py:103-142 (depot sampled with the locations)"""
        loc_sampler = torch.distributions.Uniform(low=0.0, high=1.0)
        locs = loc_sampler.sample((batch_size, self.num_loc + 1, 2))
        locs_with_depot = locs
        if self.prize_type == "const":
            prize = torch.ones(batch_size, self.num_loc)
        elif self.prize_type == "unif":
            prize = (1 + torch.randint(0, 100, (batch_size, self.num_loc)).float()) / 100
        else:  # "dist": the distance to the depot, quantised to 1..100 hundredths
            prize = (locs_with_depot[..., 0:1, :] - locs_with_depot[..., 1:, :]).norm(p=2, dim=-1)
            prize = (1 + (prize / prize.max(dim=-1, keepdim=True)[0] * 99).int()).float() / 100
        return {"locs": locs_with_depot[..., 1:, :], "depot": locs_with_depot[..., 0, :], "prize": prize,
                "max_length": torch.full((batch_size,), self.max_length)}

    def reset(self, td: dict | None = None, batch_size: int | None = None) -> dict:
        """op/env.py:100-135"""
        if td is None:
            td = self.generate(batch_size)
        b = td["locs"].shape[0]
        device = td["locs"].device
        locs_with_depot = torch.cat((td["depot"][:, None, :], td["locs"]), -2)
        td_reset = {
            "locs": locs_with_depot,
            "prize": torch.nn.functional.pad(td["prize"], (1, 0), mode="constant", value=0),
            "tour_length": torch.zeros(b, device=device),
            # the longest tour with which a node may still be ENTERED: the way back to the depot and a
            # 1e-6 margin are taken off once, here
            "max_length": td["max_length"][..., None]
            - (td["depot"][..., None, :] - locs_with_depot).norm(p=2, dim=-1)
            - 1e-6,
            "current_node": torch.zeros(b, 1, dtype=torch.long, device=device),
            "visited": torch.zeros((b, locs_with_depot.shape[-2]), dtype=torch.bool, device=device),
            "current_total_prize": torch.zeros(b, dtype=torch.float, device=device),
            "i": torch.zeros((b,), dtype=torch.int64, device=device),
            "done": torch.zeros((b,), dtype=torch.bool, device=device),
        }
        td_reset["action_mask"] = self.get_action_mask(td_reset)
        return td_reset

    def step(self, td: dict) -> dict:
        """op/env.py:67-98"""
        current_node = td["action"][:, None]
        previus_loc = gather_by_index(td["locs"], td["current_node"])
        current_loc = gather_by_index(td["locs"], current_node)
        tour_length = td["tour_length"] + (current_loc - previus_loc).norm(p=2, dim=-1)
        current_total_prize = td["current_total_prize"] + gather_by_index(td["prize"], current_node, dim=-1)
        visited = td["visited"].scatter(-1, current_node, 1)
        done = (current_node.squeeze(-1) == 0) & (td["i"] > 0)
        td.update(
            {
                "tour_length": tour_length,
                "current_node": current_node,
                "visited": visited,
                "current_total_prize": current_total_prize,
                "i": td["i"] + 1,
                "reward": torch.zeros_like(done),
                "done": done,
            }
        )
        td["action_mask"] = self.get_action_mask(td)
        return td

    @staticmethod
    def get_action_mask(td: dict) -> Tensor:
        """op/env.py:137-154"""
        current_loc = gather_by_index(td["locs"], td["current_node"])[..., None, :]
        exceeds_length = td["tour_length"][..., None] + (td["locs"] - current_loc).norm(p=2, dim=-1) > td["max_length"]
        mask = td["visited"] | td["visited"][..., 0:1] | exceeds_length
        action_mask = ~mask
        action_mask[..., 0] = 1  # the depot can always be visited
        return action_mask

    def get_reward(self, td: dict, actions: Tensor, check_solution: bool | None = None) -> Tensor:
        """base.py:180-190 -> op/env.py:156-166"""
        check_solution = self.check_solution if check_solution is None else check_solution
        if check_solution:
            self.check_solution_validity(td, actions)
        if actions.size(-1) == 1:
            assert (actions == 0).all(), "If all length 1 tours, they should be zero"
            return torch.zeros(actions.size(0), dtype=torch.float, device=actions.device)
        return td["prize"].gather(1, actions).sum(-1)

    @staticmethod
    def check_solution_validity(td: dict, actions: Tensor, add_distance_to_depot: bool = True) -> None:
        """op/env.py:168-194"""
        sorted_actions = actions.data.sort(1)[0]
        assert ((sorted_actions[:, 1:] == 0) | (sorted_actions[:, 1:] > sorted_actions[:, :-1])).all(), "Duplicates"
        locs_ordered = gather_by_index(td["locs"], actions)
        length = get_tour_length(locs_ordered)
        max_length = td["max_length"]
        if add_distance_to_depot:
            max_length = max_length + (td["locs"][..., 0:1, :] - td["locs"]).norm(p=2, dim=-1) + 1e-6
        assert (length[..., None] <= max_length + 1e-5).all(), "Max length exceeded"

    def get_num_starts(self, td):
        return get_num_starts(td, self.name)

    def select_start_nodes(self, td, num_starts):
        return select_start_nodes(td, self, num_starts)


class PCTSPEnv:
    """envs/routing/pctsp/env.py:17-219 (prize-collecting TSP, deterministic prizes)"""

    name = "pctsp"

    def __init__(self, num_loc: int = 20, check_solution: bool = True, penalty_factor: float = 3.0,
                 prize_required: float = 1.0, stochastic: bool = False):
        self.stochastic = stochastic  # spctsp/env.py:8-21: the real prize differs from the expected one
        if stochastic:
            self.name = "spctsp"
        self.num_loc = num_loc
        self.check_solution = check_solution
        self.prize_required = prize_required
        max_penalty = OP_MAX_LENGTHS.get(num_loc, None)  # pctsp/generator.py:13,74-87 (same table)
        if max_penalty is None:
            closest = min(OP_MAX_LENGTHS.keys(), key=lambda x: abs(x - num_loc))
            max_penalty = OP_MAX_LENGTHS[closest]
        self.max_penalty = max_penalty * penalty_factor / num_loc

    def generate(self, batch_size: int) -> dict:
        """pctsp/generator.py:91-122: locations (+ depot), penalties, deterministic prizes, stochastic prizes"""
        locs = torch.distributions.Uniform(low=0.0, high=1.0).sample((batch_size, self.num_loc + 1, 2))
        depot, locs = locs[..., 0, :], locs[..., 1:, :]
        penalty = torch.distributions.Uniform(low=0.0, high=self.max_penalty).sample((batch_size, self.num_loc))
        det = torch.distributions.Uniform(low=0.0, high=4.0 / self.num_loc).sample((batch_size, self.num_loc))
        sto = torch.distributions.Uniform(low=0.0, high=2.0).sample((batch_size, self.num_loc)) * det
        return {"locs": locs, "depot": depot, "penalty": penalty, "deterministic_prize": det, "stochastic_prize": sto}

    def reset(self, td: dict | None = None, batch_size: int | None = None) -> dict:
        """pctsp/env.py:93-139"""
        if td is None:
            td = self.generate(batch_size)
        b = td["locs"].shape[0]
        device = td["locs"].device
        real_prize = td["stochastic_prize"] if self.stochastic else td["deterministic_prize"]  # pctsp/env.py:98
        penalty = td["penalty"]
        td_reset = {
            "locs": torch.cat([td["depot"][..., None, :], td["locs"]], dim=-2),
            "current_node": torch.zeros((b,), dtype=torch.int64, device=device),
            "expected_prize": td["deterministic_prize"],
            "real_prize": torch.cat([torch.zeros_like(real_prize[..., :1]), real_prize], dim=-1),
            "penalty": torch.nn.functional.pad(penalty, (1, 0), mode="constant", value=0),
            "cur_total_prize": torch.zeros(b, device=device),
            "cur_total_penalty": penalty.sum(-1),
            "visited": torch.zeros((b, self.num_loc + 1), dtype=torch.bool, device=device),
            "prize_required": torch.full((b,), self.prize_required, device=device),
            "i": torch.zeros((b,), dtype=torch.int64, device=device),
            "done": torch.zeros((b,), dtype=torch.bool, device=device),
        }
        td_reset["action_mask"] = self.get_action_mask(td_reset)
        return td_reset

    def step(self, td: dict) -> dict:
        """pctsp/env.py:62-91"""
        current_node = td["action"]
        cur_total_prize = td["cur_total_prize"] + gather_by_index(td["real_prize"], current_node)
        cur_total_penalty = td["cur_total_penalty"] + gather_by_index(td["penalty"], current_node)
        visited = td["visited"].scatter(-1, current_node[..., None], 1)
        done = (td["i"] > 0) & (current_node == 0)
        td.update(
            {
                "current_node": current_node,
                "cur_total_prize": cur_total_prize,
                "cur_total_penalty": cur_total_penalty,
                "visited": visited,
                "i": td["i"] + 1,
                "reward": torch.zeros_like(done),
                "done": done,
            }
        )
        td["action_mask"] = self.get_action_mask(td)
        return td

    @staticmethod
    def get_action_mask(td: dict) -> Tensor:
        """pctsp/env.py:141-148: the depot opens once a total prize of 1 is collected (or nothing is left)"""
        mask = td["visited"] | td["visited"][..., 0:1]
        mask[..., 0] = (td["cur_total_prize"] < 1.0) & (
            td["visited"][..., 1:].int().sum(-1) < td["visited"][..., 1:].size(-1)
        )
        return ~(mask > 0)

    def get_reward(self, td: dict, actions: Tensor, check_solution: bool | None = None) -> Tensor:
        """base.py:180-190 -> pctsp/env.py:150-173: saved penalties - (tour length + all penalties)"""
        check_solution = self.check_solution if check_solution is None else check_solution
        if check_solution:
            self.check_solution_validity(td, actions)
        if actions.size(-1) == 1:
            assert (actions == 0).all(), "If all length 1 tours, they should be zero"
            return torch.zeros(actions.size(0), dtype=torch.float, device=actions.device)
        locs_ordered = torch.cat([td["locs"][..., 0:1, :], gather_by_index(td["locs"], actions)], dim=1)
        length = get_tour_length(locs_ordered)
        saved_penalty = td["penalty"].gather(1, actions)
        return saved_penalty.sum(-1) - (length + td["penalty"][..., 1:].sum(-1))

    @staticmethod
    def check_solution_validity(td: dict, actions: Tensor) -> None:
        """pctsp/env.py:175-201"""
        sorted_actions = actions.data.sort(1)[0]
        assert ((sorted_actions[..., 1:] == 0) | (sorted_actions[..., 1:] > sorted_actions[..., :-1])).all(), "Duplicates"
        prize = td["real_prize"][..., 1:]
        prize_with_depot = torch.cat((torch.zeros_like(prize[:, :1]), prize), 1)
        p = prize_with_depot.gather(1, actions)
        assert (
            (p.sum(-1) >= 1 - 1e-5)
            | (sorted_actions.size(-1) - (sorted_actions == 0).int().sum(-1) == (td["locs"].size(-2) - 1))
        ).all(), "Total prize does not satisfy min total prize"

    def get_num_starts(self, td):
        return get_num_starts(td, self.name)

    def select_start_nodes(self, td, num_starts):
        return select_start_nodes(td, self, num_starts)


class PDPEnv:
    """envs/routing/pdp/env.py:17-225 (pickup and delivery: node j in 1..n/2 is picked up before j + n/2 is delivered)"""

    name = "pdp"

    def __init__(self, num_loc: int = 20, check_solution: bool = True, force_start_at_depot: bool = False):
        self.num_loc = num_loc + (num_loc % 2)  # pdp/generator.py:48-51: the number of locations must be even
        self.check_solution = check_solution
        self.force_start_at_depot = force_start_at_depot

    def generate(self, batch_size: int) -> dict:
        """pdp/generator.py:71-88 (depot sampled with the locations)"""
        locs = torch.distributions.Uniform(low=0.0, high=1.0).sample((batch_size, self.num_loc + 1, 2))
        return {"locs": locs[..., 1:, :], "depot": locs[..., 0, :]}

    def reset(self, td: dict | None = None, batch_size: int | None = None) -> dict:
        """pdp/env.py:101-150"""
        if td is None:
            td = self.generate(batch_size)
        b = td["locs"].shape[0]
        device = td["locs"].device
        n = self.num_loc
        locs = torch.cat((td["depot"][:, None, :], td["locs"]), -2)
        to_deliver = torch.cat([torch.ones(b, n // 2 + 1, dtype=torch.bool, device=device),
                                torch.zeros(b, n // 2, dtype=torch.bool, device=device)], dim=-1)
        available = torch.ones((b, n + 1), dtype=torch.bool, device=device)
        action_mask = torch.ones_like(available)
        if self.force_start_at_depot:
            action_mask[..., 1:] = False
        else:
            action_mask = action_mask & to_deliver
            available[..., 0] = False
            action_mask[..., 0] = False
        return {
            "locs": locs,
            "current_node": torch.zeros((b, 1), dtype=torch.int64, device=device),
            "to_deliver": to_deliver,
            "available": available,
            "i": torch.zeros((b, 1), dtype=torch.int64, device=device),
            "action_mask": action_mask,
            "done": torch.zeros((b,), dtype=torch.bool, device=device),
        }

    @staticmethod
    def step(td: dict) -> dict:
        """pdp/env.py:64-99"""
        current_node = td["action"].unsqueeze(-1)
        num_loc = td["locs"].shape[-2] - 1
        new_to_deliver = (current_node + num_loc // 2) % (num_loc + 1)
        available = td["available"].scatter(-1, current_node.expand_as(td["action_mask"]), 0)
        to_deliver = td["to_deliver"].scatter(-1, new_to_deliver.expand_as(td["to_deliver"]), 1)
        action_mask = available & to_deliver
        done = torch.count_nonzero(available, dim=-1) == 0
        td.update(
            {
                "current_node": current_node,
                "available": available,
                "to_deliver": to_deliver,
                "i": td["i"] + 1,
                "action_mask": action_mask,
                "reward": torch.zeros_like(done),
                "done": done,
            }
        )
        return td

    def get_reward(self, td: dict, actions: Tensor, check_solution: bool | None = None) -> Tensor:
        """base.py:180-190 -> pdp/env.py:191-202"""
        check_solution = self.check_solution if check_solution is None else check_solution
        if check_solution:
            self.check_solution_validity(td, actions)
        locs_ordered = torch.cat([td["locs"][..., 0:1, :], gather_by_index(td["locs"], actions)], dim=1)
        return -get_tour_length(locs_ordered)

    def check_solution_validity(self, td: dict, actions: Tensor) -> None:
        """pdp/env.py:204-223"""
        if not self.force_start_at_depot:
            actions = torch.cat((torch.zeros_like(actions[:, 0:1]), actions), dim=-1)
        assert (
            torch.arange(actions.size(1), device=actions.device).view(1, -1).expand_as(actions) == actions.sort(1)[0]
        ).all(), "Not visiting all nodes"
        assert (actions[:, 1:-1] != 0).all(), "Going back to depot in the middle of the tour (not allowed)"
        visited_time = torch.argsort(actions, 1)
        assert (
            visited_time[:, 1 : actions.size(1) // 2 + 1] < visited_time[:, actions.size(1) // 2 + 1 :]
        ).all(), "Deliverying without pick-up"

    def get_num_starts(self, td):
        return (td["locs"].shape[-2] - 1) // 2

    def select_start_nodes(self, td, num_starts):
        return select_start_nodes(td, self, num_starts)


def get_env(name: str, num_loc: int, **kw):
    if name == "spctsp":
        return PCTSPEnv(num_loc=num_loc, stochastic=True, **kw)
    return {"tsp": TSPEnv, "cvrp": CVRPEnv, "op": OPEnv, "pctsp": PCTSPEnv, "pdp": PDPEnv,
            "cvrptw": CVRPTWEnv}[name](num_loc=num_loc, **kw)


# ----------------------------------------------------------------------------------------------
# rl4co/models/nn/attention.py
# ----------------------------------------------------------------------------------------------


def scaled_dot_product_attention_simple(q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False):
    """attention.py:19-50"""
    scores = torch.matmul(q, k.transpose(-2, -1)) / (k.size(-1) ** 0.5)
    if attn_mask is not None:
        if attn_mask.dtype == torch.bool:
            scores.masked_fill_(~attn_mask, float("-inf"))
        else:
            scores += attn_mask
    attn_weights = F.softmax(scores, dim=-1)
    return torch.matmul(attn_weights, v)


def _resolve_sdpa(sdpa_fn) -> Callable:
    """attention.py:259-272"""
    if sdpa_fn is None or sdpa_fn == "default":
        return F.scaled_dot_product_attention
    if sdpa_fn == "simple":
        return scaled_dot_product_attention_simple
    if callable(sdpa_fn):
        return sdpa_fn
    raise ValueError(f"Unknown sdpa_fn: {sdpa_fn}")


class MultiHeadAttention(nn.Module):
    """attention.py:64-134"""

    def __init__(self, embed_dim: int, num_heads: int, bias: bool = True, sdpa_fn=None):
        super().__init__()
        self.embed_dim = embed_dim
        self.num_heads = num_heads
        self.sdpa_fn = _resolve_sdpa(sdpa_fn)
        self.Wqkv = nn.Linear(embed_dim, 3 * embed_dim, bias=bias)
        self.out_proj = nn.Linear(embed_dim, embed_dim, bias=bias)

    def forward(self, x, attn_mask=None):
        q, k, v = rearrange(
            self.Wqkv(x), "b s (three h d) -> three b h s d", three=3, h=self.num_heads
        ).unbind(dim=0)
        out = self.sdpa_fn(q, k, v, attn_mask=attn_mask, dropout_p=0.0)
        return self.out_proj(rearrange(out, "b h s d -> b s (h d)"))


class PointerAttention(nn.Module):
    """attention.py:218-320"""

    def __init__(self, embed_dim, num_heads, mask_inner=True, out_bias=False, check_nan=True,
                 sdpa_fn="default"):
        super().__init__()
        self.num_heads = num_heads
        self.mask_inner = mask_inner
        self.project_out = nn.Linear(embed_dim, embed_dim, bias=out_bias)
        self.check_nan = check_nan
        self.sdpa_fn = _resolve_sdpa(sdpa_fn)

    def forward(self, query, key, value, logit_key, attn_mask=None):
        heads = self._inner_mha(query, key, value, attn_mask)
        glimpse = self.project_out(heads)
        logits = (torch.bmm(glimpse, logit_key.squeeze(-2).transpose(-2, -1))).squeeze(
            -2
        ) / math.sqrt(glimpse.size(-1))
        if self.check_nan:
            assert not torch.isnan(logits).any(), "Logits contain NaNs"
        return logits

    def _inner_mha(self, query, key, value, attn_mask):
        q = self._make_heads(query)
        k = self._make_heads(key)
        v = self._make_heads(value)
        if self.mask_inner:
            attn_mask = (
                attn_mask.unsqueeze(1) if attn_mask.ndim == 3 else attn_mask.unsqueeze(1).unsqueeze(2)
            )
        else:
            attn_mask = None
        heads = self.sdpa_fn(q, k, v, attn_mask=attn_mask)
        return rearrange(heads, "... h n g -> ... n (h g)", h=self.num_heads)

    def _make_heads(self, v):
        return rearrange(v, "... g (h s) -> ... h g s", h=self.num_heads)


# ----------------------------------------------------------------------------------------------
# rl4co/models/nn/{ops,mlp}.py, nn/graph/attnnet.py, nn/env_embeddings/*, zoo/am/*
# ----------------------------------------------------------------------------------------------


class SkipConnection(nn.Module):
    """nn/ops.py:9-15"""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, x):
        return x + self.module(x)


class Normalization(nn.Module):
    """nn/ops.py:30-54"""

    def __init__(self, embed_dim, normalization="batch"):
        super().__init__()
        if normalization != "layer":
            cls = {"batch": nn.BatchNorm1d, "instance": nn.InstanceNorm1d}.get(normalization, None)
            self.normalizer = cls(embed_dim, affine=True)
        else:
            self.normalizer = "layer"

    def forward(self, x):
        if isinstance(self.normalizer, nn.BatchNorm1d):
            return self.normalizer(x.view(-1, x.size(-1))).view(*x.size())
        elif isinstance(self.normalizer, nn.InstanceNorm1d):
            return self.normalizer(x.permute(0, 2, 1)).permute(0, 2, 1)
        elif self.normalizer == "layer":
            return (x - x.mean((1, 2)).view(-1, 1, 1)) / torch.sqrt(
                x.var((1, 2)).view(-1, 1, 1) + 1e-05
            )
        return x


class MLP(nn.Module):
    """nn/mlp.py:8-61 (hidden ReLU, identity output, no norms/dropout in the AM encoder)"""

    def __init__(self, input_dim, output_dim, num_neurons):
        super().__init__()
        input_dims = [input_dim] + num_neurons
        output_dims = num_neurons + [output_dim]
        self.lins = nn.ModuleList()
        for in_dim, out_dim in zip(input_dims, output_dims):
            self.lins.append(nn.Linear(in_dim, out_dim))

    def forward(self, xs):
        for lin in self.lins[:-1]:
            xs = F.relu(lin(xs))
        return self.lins[-1](xs)


class MultiHeadAttentionLayer(nn.Sequential):
    """nn/graph/attnnet.py:16-54 — NB the FFN is constructed before the MHA (RNG order)."""

    def __init__(self, embed_dim, num_heads=8, feedforward_hidden=512, normalization="batch",
                 sdpa_fn=None):
        num_neurons = [feedforward_hidden] if feedforward_hidden > 0 else []
        ffn = MLP(embed_dim, embed_dim, num_neurons)
        super().__init__(
            SkipConnection(MultiHeadAttention(embed_dim, num_heads, bias=True, sdpa_fn=sdpa_fn)),
            Normalization(embed_dim, normalization),
            SkipConnection(ffn),
            Normalization(embed_dim, normalization),
        )


class GraphAttentionNetwork(nn.Module):
    """nn/graph/attnnet.py:57-106"""

    def __init__(self, num_heads, embed_dim, num_layers, normalization="batch",
                 feedforward_hidden=512, sdpa_fn=None):
        super().__init__()
        self.layers = nn.Sequential(
            *(
                MultiHeadAttentionLayer(embed_dim, num_heads, feedforward_hidden, normalization, sdpa_fn)
                for _ in range(num_layers)
            )
        )

    def forward(self, x):
        return self.layers(x)


class TSPInitEmbedding(nn.Module):
    """env_embeddings/init.py:55-68"""

    def __init__(self, embed_dim):
        super().__init__()
        self.init_embed = nn.Linear(2, embed_dim, True)

    def forward(self, td):
        return self.init_embed(td["locs"])


class VRPInitEmbedding(nn.Module):
    """env_embeddings/init.py:115-136"""

    def __init__(self, embed_dim):
        super().__init__()
        self.init_embed = nn.Linear(3, embed_dim, True)
        self.init_embed_depot = nn.Linear(2, embed_dim, True)

    def forward(self, td):
        depot, cities = td["locs"][:, :1, :], td["locs"][:, 1:, :]
        depot_embedding = self.init_embed_depot(depot)
        node_embeddings = self.init_embed(torch.cat((cities, td["demand"][..., None]), -1))
        return torch.cat((depot_embedding, node_embeddings), -2)


class VRPTWInitEmbedding(nn.Module):
    """env_embeddings/init.py:139-153: customers (x, y, demand, tw start, tw end, service time)"""

    def __init__(self, embed_dim):
        super().__init__()
        self.init_embed = nn.Linear(6, embed_dim, True)
        self.init_embed_depot = nn.Linear(2, embed_dim, True)

    def forward(self, td):
        depot, cities = td["locs"][:, :1, :], td["locs"][:, 1:, :]
        durations = td["durations"][..., 1:]
        time_windows = td["time_windows"][..., 1:, :]
        depot_embedding = self.init_embed_depot(depot)
        node_embeddings = self.init_embed(
            torch.cat((cities, td["demand"][..., None], time_windows, durations[..., None]), -1)
        )
        return torch.cat((depot_embedding, node_embeddings), -2)


class OPInitEmbedding(nn.Module):
    """env_embeddings/init.py:254-280"""

    def __init__(self, embed_dim):
        super().__init__()
        self.init_embed = nn.Linear(3, embed_dim, True)
        self.init_embed_depot = nn.Linear(2, embed_dim, True)

    def forward(self, td):
        depot, cities = td["locs"][:, :1, :], td["locs"][:, 1:, :]
        depot_embedding = self.init_embed_depot(depot)
        node_embeddings = self.init_embed(torch.cat((cities, td["prize"][..., 1:, None]), -1))
        return torch.cat((depot_embedding, node_embeddings), -2)


class PCTSPInitEmbedding(nn.Module):
    """env_embeddings/init.py:221-251: x, y, expected prize, penalty"""

    def __init__(self, embed_dim):
        super().__init__()
        self.init_embed = nn.Linear(4, embed_dim, True)
        self.init_embed_depot = nn.Linear(2, embed_dim, True)

    def forward(self, td):
        depot, cities = td["locs"][:, :1, :], td["locs"][:, 1:, :]
        depot_embedding = self.init_embed_depot(depot)
        node_embeddings = self.init_embed(
            torch.cat((cities, td["expected_prize"][..., None], td["penalty"][..., 1:, None]), -1)
        )
        return torch.cat((depot_embedding, node_embeddings), -2)


class PDPInitEmbedding(nn.Module):
    """env_embeddings/init.py:335-360: depot (x, y); pickup (x, y, x', y' of its delivery); delivery (x, y)"""

    def __init__(self, embed_dim):
        super().__init__()
        self.init_embed_depot = nn.Linear(2, embed_dim, True)
        self.init_embed_pick = nn.Linear(4, embed_dim, True)
        self.init_embed_delivery = nn.Linear(2, embed_dim, True)

    def forward(self, td):
        depot, locs = td["locs"][..., 0:1, :], td["locs"][..., 1:, :]
        num_locs = locs.size(-2)
        pick_feats = torch.cat([locs[:, : num_locs // 2, :], locs[:, num_locs // 2 :, :]], -1)
        delivery_feats = locs[:, num_locs // 2 :, :]
        return torch.cat([self.init_embed_depot(depot), self.init_embed_pick(pick_feats),
                          self.init_embed_delivery(delivery_feats)], -2)


class PDPContext(nn.Module):
    """env_embeddings/context.py:50-65,232-243: the current node embedding alone"""

    def __init__(self, embed_dim):
        super().__init__()
        self.embed_dim = embed_dim
        self.project_context = nn.Linear(embed_dim, embed_dim, bias=False)

    def forward(self, embeddings, td):
        cur_node_embedding = gather_by_index(embeddings, td["current_node"]).squeeze()
        return self.project_context(cur_node_embedding)


class TSPContext(nn.Module):
    """env_embeddings/context.py:50-60,105-134"""

    def __init__(self, embed_dim):
        super().__init__()
        self.embed_dim = embed_dim
        self.project_context = nn.Linear(2 * embed_dim, embed_dim, bias=False)
        self.W_placeholder = nn.Parameter(torch.Tensor(2 * self.embed_dim).uniform_(-1, 1))

    def forward(self, embeddings, td):
        batch_size = embeddings.size(0)
        node_dim = (-1,) if td["first_node"].dim() == 1 else (td["first_node"].size(-1), -1)
        if td["i"][(0,) * td["i"].dim()].item() < 1:
            if td["first_node"].dim() < 2:  # len(td.batch_size) < 2
                context_embedding = self.W_placeholder[None, :].expand(
                    batch_size, self.W_placeholder.size(-1)
                )
            else:
                context_embedding = self.W_placeholder[None, None, :].expand(
                    batch_size, td["first_node"].size(1), self.W_placeholder.size(-1)
                )
        else:
            context_embedding = gather_by_index(
                embeddings,
                torch.stack([td["first_node"], td["current_node"]], -1).view(batch_size, -1),
            ).view(batch_size, *node_dim)
        return self.project_context(context_embedding)


class VRPContext(nn.Module):
    """env_embeddings/context.py:50-74,137-149"""

    def __init__(self, embed_dim):
        super().__init__()
        self.embed_dim = embed_dim
        self.project_context = nn.Linear(embed_dim + 1, embed_dim, bias=False)

    def forward(self, embeddings, td):
        cur_node_embedding = gather_by_index(embeddings, td["current_node"])
        state_embedding = td["vehicle_capacity"] - td["used_capacity"]
        context_embedding = torch.cat([cur_node_embedding, state_embedding], -1)
        return self.project_context(context_embedding)


class VRPTWContext(nn.Module):
    """env_embeddings/context.py:50-74,137-166: current node embedding, remaining capacity, current time"""

    def __init__(self, embed_dim):
        super().__init__()
        self.embed_dim = embed_dim
        self.project_context = nn.Linear(embed_dim + 2, embed_dim, bias=False)

    def forward(self, embeddings, td):
        cur_node_embedding = gather_by_index(embeddings, td["current_node"])
        capacity = td["vehicle_capacity"] - td["used_capacity"]
        state_embedding = torch.cat([capacity, td["current_time"]], -1)
        context_embedding = torch.cat([cur_node_embedding, state_embedding], -1)
        return self.project_context(context_embedding)


class OPContext(nn.Module):
    """env_embeddings/context.py:50-74,201-213: current node embedding + remaining length"""

    def __init__(self, embed_dim):
        super().__init__()
        self.embed_dim = embed_dim
        self.project_context = nn.Linear(embed_dim + 1, embed_dim, bias=False)

    def forward(self, embeddings, td):
        cur_node_embedding = gather_by_index(embeddings, td["current_node"])
        state_embedding = (td["max_length"][..., 0] - td["tour_length"])[..., None]
        context_embedding = torch.cat([cur_node_embedding, state_embedding], -1)
        return self.project_context(context_embedding)


class PCTSPContext(nn.Module):
    """env_embeddings/context.py:50-74,184-198: current node embedding + prize still to collect (clamped at 0)"""

    def __init__(self, embed_dim):
        super().__init__()
        self.embed_dim = embed_dim
        self.project_context = nn.Linear(embed_dim + 1, embed_dim, bias=False)

    def forward(self, embeddings, td):
        cur_node_embedding = gather_by_index(embeddings, td["current_node"])
        state_embedding = torch.clamp(td["prize_required"] - td["cur_total_prize"], min=0)[..., None]
        context_embedding = torch.cat([cur_node_embedding, state_embedding], -1)
        return self.project_context(context_embedding)


class StaticEmbedding(nn.Module):
    """env_embeddings/dynamic.py:47-57"""

    def forward(self, td):
        return 0, 0, 0


@dataclass
class PrecomputedCache:
    """zoo/am/decoder.py:21-40"""

    node_embeddings: Tensor
    graph_context: Tensor | float
    glimpse_key: Tensor
    glimpse_val: Tensor
    logit_key: Tensor


class AttentionModelEncoder(nn.Module):
    """zoo/am/encoder.py:12-87"""

    def __init__(self, embed_dim=128, env_name="tsp", num_heads=8, num_layers=3,
                 normalization="batch", feedforward_hidden=512, sdpa_fn=None):
        super().__init__()
        self.env_name = env_name
        self.init_embedding = {"tsp": TSPInitEmbedding, "cvrp": VRPInitEmbedding, "op": OPInitEmbedding, "pctsp": PCTSPInitEmbedding,
                               "pdp": PDPInitEmbedding, "cvrptw": VRPTWInitEmbedding,
                               "spctsp": PCTSPInitEmbedding}[env_name](embed_dim)
        self.net = GraphAttentionNetwork(
            num_heads, embed_dim, num_layers, normalization, feedforward_hidden, sdpa_fn=sdpa_fn
        )

    def forward(self, td):
        init_h = self.init_embedding(td)
        h = self.net(init_h)
        return h, init_h


class AttentionModelDecoder(nn.Module):
    """zoo/am/decoder.py:43-228"""

    def __init__(self, embed_dim=128, num_heads=8, env_name="tsp", mask_inner=True,
                 out_bias_pointer_attn=False, linear_bias=False, use_graph_context=True,
                 check_nan=True, sdpa_fn=None):
        super().__init__()
        self.env_name = env_name
        self.embed_dim = embed_dim
        self.num_heads = num_heads
        self.context_embedding = {"tsp": TSPContext, "cvrp": VRPContext, "op": OPContext, "pctsp": PCTSPContext,
                                  "pdp": PDPContext, "cvrptw": VRPTWContext,
                                  "spctsp": PCTSPContext}[env_name](embed_dim)
        self.dynamic_embedding = StaticEmbedding()
        self.is_dynamic_embedding = False
        self.pointer = PointerAttention(
            embed_dim, num_heads, mask_inner=mask_inner, out_bias=out_bias_pointer_attn,
            check_nan=check_nan, sdpa_fn=sdpa_fn,
        )
        self.project_node_embeddings = nn.Linear(embed_dim, 3 * embed_dim, bias=linear_bias)
        self.project_fixed_context = nn.Linear(embed_dim, embed_dim, bias=linear_bias)
        self.use_graph_context = use_graph_context

    def _compute_q(self, cached: PrecomputedCache, td: dict):
        """decoder.py:128-140 (td.dim() == 2  <=>  state tensors carry a starts dimension)"""
        graph_context_cache = cached.graph_context
        if td["action_mask"].dim() == 3 and isinstance(graph_context_cache, Tensor):
            graph_context_cache = graph_context_cache.unsqueeze(1)
        step_context = self.context_embedding(cached.node_embeddings, td)
        glimpse_q = step_context + graph_context_cache
        glimpse_q = glimpse_q.unsqueeze(1) if glimpse_q.ndim == 2 else glimpse_q
        return glimpse_q

    def _compute_kvl(self, cached: PrecomputedCache, td: dict):
        """decoder.py:142-154"""
        glimpse_k_dyn, glimpse_v_dyn, logit_k_dyn = self.dynamic_embedding(td)
        glimpse_k = cached.glimpse_key + glimpse_k_dyn
        glimpse_v = cached.glimpse_val + glimpse_v_dyn
        logit_k = cached.logit_key + logit_k_dyn
        return glimpse_k, glimpse_v, logit_k

    def forward(self, td: dict, cached: PrecomputedCache, num_starts: int = 0):
        """decoder.py:156-193"""
        if num_starts > 1:
            td = unbatchify(td, num_starts)
        glimpse_q = self._compute_q(cached, td)
        glimpse_k, glimpse_v, logit_k = self._compute_kvl(cached, td)
        mask = td["action_mask"]
        logits = self.pointer(glimpse_q, glimpse_k, glimpse_v, logit_k, mask)
        if num_starts > 1:
            logits = rearrange(logits, "b s l -> (s b) l", s=num_starts)
            mask = rearrange(mask, "b s l -> (s b) l", s=num_starts)
        return logits, mask

    def pre_decoder_hook(self, td, env, embeddings, num_starts: int = 0):
        return td, env, self._precompute_cache(embeddings, num_starts=num_starts)

    def _precompute_cache(self, embeddings: Tensor, num_starts: int = 0) -> PrecomputedCache:
        """decoder.py:201-228"""
        glimpse_key_fixed, glimpse_val_fixed, logit_key_fixed = self.project_node_embeddings(
            embeddings
        ).chunk(3, dim=-1)
        if self.use_graph_context:
            graph_context = self.project_fixed_context(embeddings.mean(1))
        else:
            graph_context = 0
        return PrecomputedCache(
            node_embeddings=embeddings,
            graph_context=graph_context,
            glimpse_key=glimpse_key_fixed,
            glimpse_val=glimpse_val_fixed,
            logit_key=logit_key_fixed,
        )


# ----------------------------------------------------------------------------------------------
# rl4co/utils/decoding.py
# ----------------------------------------------------------------------------------------------


def get_log_likelihood(logprobs, actions=None, mask=None, return_sum: bool = True):
    """decoding.py:38-62"""
    if actions is not None and logprobs.dim() == 3:
        logprobs = logprobs.gather(-1, actions.unsqueeze(-1)).squeeze(-1)
    if mask is not None:
        logprobs[~mask] = 0
    assert (logprobs > -1000).data.all(), "Logprobs should not be -inf, check sampling procedure!"
    return logprobs.sum(1) if return_sum else logprobs


def process_logits(logits, mask=None, temperature=1.0, tanh_clipping=0, mask_logits=True):
    """decoding.py:138-188 (top-k / top-p are out of scope)"""
    if tanh_clipping > 0:
        logits = torch.tanh(logits) * tanh_clipping
    if mask_logits:
        assert mask is not None, "mask must be provided if mask_logits is True"
        logits[~mask] = float("-inf")
    logits = logits / temperature
    return F.log_softmax(logits, dim=-1)


class DecodingStrategy:
    """decoding.py:191-423 (greedy / sampling / evaluate, optional multistart)."""

    name = "base"

    def __init__(self, temperature=1.0, mask_logits=True, tanh_clipping=0, num_starts=None,
                 multistart=False, select_best=False, store_all_logp=False, num_samples=None,
                 multisample=False, noise_recorder: list | None = None, **kwargs):
        self.temperature = temperature
        self.mask_logits = mask_logits
        self.tanh_clipping = tanh_clipping
        assert not (multistart and multisample)
        if num_samples is not None:
            multisample = True if num_samples > 1 else False
        if num_starts is not None:
            multistart = True if num_starts > 1 else False
        self.multistart = multistart
        self.multisample = multisample
        self.num_starts = num_starts if multistart else num_samples
        self.select_best = select_best
        self.store_all_logp = store_all_logp
        self.actions = []
        self.logprobs = []
        self.noise_recorder = noise_recorder

    def pre_decoder_hook(self, td: dict, env, action=None):
        """decoding.py:282-330"""
        if self.multistart or self.multisample:
            if self.num_starts is None:
                self.num_starts = env.get_num_starts(td)
        else:
            self.num_starts = 0
        if self.num_starts >= 1:
            if self.multistart:
                if action is None:
                    action = env.select_start_nodes(td, num_starts=self.num_starts)
                td = batchify(td, self.num_starts)
                td["action"] = action
                td = env.step(td)
                if self.store_all_logp:
                    logprobs = torch.zeros_like(td["action_mask"])
                else:
                    logprobs = torch.zeros_like(action, device=action.device)
                self.logprobs.append(logprobs)
                self.actions.append(action)
            else:
                td = batchify(td, self.num_starts)
        return td, env, self.num_starts

    def post_decoder_hook(self, td: dict, env):
        """decoding.py:332-342"""
        assert len(self.logprobs) > 0
        logprobs = torch.stack(self.logprobs, 1)
        actions = torch.stack(self.actions, 1)
        if self.num_starts > 0 and self.select_best:
            logprobs, actions, td, env = self._select_best(logprobs, actions, td, env)
        return logprobs, actions, td, env

    def step(self, logits, mask, td: dict, action=None):
        """decoding.py:344-385"""
        if not self.mask_logits:
            mask = None
        logprobs = process_logits(
            logits, mask, temperature=self.temperature, tanh_clipping=self.tanh_clipping,
            mask_logits=self.mask_logits,
        )
        logprobs, selected_action, td = self._step(logprobs, mask, td, action=action)
        if not self.store_all_logp:
            logprobs = gather_by_index(logprobs, selected_action, dim=1)
        td["action"] = selected_action
        self.actions.append(selected_action)
        self.logprobs.append(logprobs)
        return td

    @staticmethod
    def greedy(logprobs, mask=None):
        """decoding.py:387-397"""
        selected = logprobs.argmax(dim=-1)
        if mask is not None:
            assert not (~mask).gather(1, selected.unsqueeze(-1)).data.any(), "infeasible action selected"
        return selected

    def sampling(self, logprobs, mask=None):
        """decoding.py:399-413. torch.multinomial(p, 1) == argmax(p / q), q = empty_like(p).
        exponential_(1) on the same generator [SURVEY.md §8c probe]; when a recorder is attached
        the draw is made explicitly so the HIP kernel can consume the very same noise."""
        probs = logprobs.exp()
        if self.noise_recorder is not None:
            q = torch.empty_like(probs).exponential_(1)
            self.noise_recorder.append(q.clone())
            selected = torch.div(probs, q).argmax(dim=-1)
        else:
            selected = torch.multinomial(probs, 1).squeeze(1)
        if mask is not None:
            assert not (~mask).gather(1, selected.unsqueeze(-1)).data.any(), "infeasible action selected"
        return selected

    def _select_best(self, logprobs, actions, td: dict, env):
        """decoding.py:415-423"""
        rewards = env.get_reward(td, actions)
        _, max_idxs = unbatchify(rewards, self.num_starts).max(dim=-1)
        actions = unbatchify_and_gather(actions, max_idxs, self.num_starts)
        logprobs = unbatchify_and_gather(logprobs, max_idxs, self.num_starts)
        td = unbatchify_and_gather(td, max_idxs, self.num_starts)
        return logprobs, actions, td, env


class Greedy(DecodingStrategy):
    name = "greedy"

    def _step(self, logprobs, mask, td, **kw):
        return logprobs, self.greedy(logprobs, mask), td


class Sampling(DecodingStrategy):
    name = "sampling"

    def _step(self, logprobs, mask, td, **kw):
        return logprobs, self.sampling(logprobs, mask), td


class Evaluate(DecodingStrategy):
    name = "evaluate"

    def _step(self, logprobs, mask, td, action=None, **kw):
        return logprobs, action, td


def get_decoding_strategy(decoding_strategy: str, **config) -> DecodingStrategy:
    """decoding.py:17-35"""
    registry = {
        "greedy": Greedy, "sampling": Sampling, "multistart_greedy": Greedy,
        "multistart_sampling": Sampling, "evaluate": Evaluate,
    }
    if "multistart" in decoding_strategy:
        config["multistart"] = True
    return registry.get(decoding_strategy, Sampling)(**config)


# ----------------------------------------------------------------------------------------------
# rl4co/models/zoo/am/policy.py + rl4co/models/common/constructive/base.py
# ----------------------------------------------------------------------------------------------


class AttentionModelPolicy(nn.Module):
    """zoo/am/policy.py:10-122 (constructor defaults) + constructive/base.py:154-263 (forward).

    Modules are created in the reference's order, so ``torch.manual_seed(s)`` followed by
    construction yields the reference's weights and ``state_dict()`` keys."""

    def __init__(self, env_name="tsp", embed_dim=128, num_encoder_layers=3, num_heads=8,
                 normalization="batch", feedforward_hidden=512, use_graph_context=True,
                 linear_bias_decoder=False, sdpa_fn=None, sdpa_fn_encoder=None, sdpa_fn_decoder=None,
                 mask_inner=True, out_bias_pointer_attn=False, check_nan=True, temperature=1.0,
                 tanh_clipping=10.0, mask_logits=True, train_decode_type="sampling",
                 val_decode_type="greedy", test_decode_type="greedy"):
        super().__init__()
        self.env_name = env_name
        self.encoder = AttentionModelEncoder(
            embed_dim=embed_dim, env_name=env_name, num_heads=num_heads,
            num_layers=num_encoder_layers, normalization=normalization,
            feedforward_hidden=feedforward_hidden,
            sdpa_fn=sdpa_fn if sdpa_fn_encoder is None else sdpa_fn_encoder,
        )
        self.decoder = AttentionModelDecoder(
            embed_dim=embed_dim, num_heads=num_heads, env_name=env_name, mask_inner=mask_inner,
            out_bias_pointer_attn=out_bias_pointer_attn, linear_bias=linear_bias_decoder,
            use_graph_context=use_graph_context, check_nan=check_nan,
            sdpa_fn=sdpa_fn if sdpa_fn_decoder is None else sdpa_fn_decoder,
        )
        self.temperature = temperature
        self.tanh_clipping = tanh_clipping
        self.mask_logits = mask_logits
        self.train_decode_type = train_decode_type
        self.val_decode_type = val_decode_type
        self.test_decode_type = test_decode_type

    def forward(self, td: dict, env, phase: str = "train", calc_reward: bool = True,
                return_actions: bool = True, return_entropy: bool = False,
                return_hidden: bool = False, return_init_embeds: bool = False,
                return_sum_log_likelihood: bool = True, actions=None, max_steps=1_000_000,
                **decoding_kwargs) -> dict:
        """constructive/base.py:154-263"""
        hidden, init_embeds = self.encoder(td)
        decode_type = decoding_kwargs.pop("decode_type", None)
        if actions is not None:
            decode_type = "evaluate"
        elif decode_type is None:
            decode_type = getattr(self, f"{phase}_decode_type")
        decode_strategy = get_decoding_strategy(
            decode_type,
            temperature=decoding_kwargs.pop("temperature", self.temperature),
            tanh_clipping=decoding_kwargs.pop("tanh_clipping", self.tanh_clipping),
            mask_logits=decoding_kwargs.pop("mask_logits", self.mask_logits),
            store_all_logp=decoding_kwargs.pop("store_all_logp", return_entropy),
            **decoding_kwargs,
        )
        td, env, num_starts = decode_strategy.pre_decoder_hook(td, env)
        td, env, hidden = self.decoder.pre_decoder_hook(td, env, hidden, num_starts)
        step = 0
        while not td["done"].all():
            logits, mask = self.decoder(td, hidden, num_starts)
            td = decode_strategy.step(
                logits, mask, td, action=actions[..., step] if actions is not None else None
            )
            td = env.step(td)
            step += 1
            if step > max_steps:
                break
        logprobs, actions, td, env = decode_strategy.post_decoder_hook(td, env)
        if calc_reward:
            td["reward"] = env.get_reward(td, actions)
        outdict = {
            "reward": td["reward"],
            "log_likelihood": get_log_likelihood(
                logprobs, actions, td.get("mask", None), return_sum_log_likelihood
            ),
        }
        if return_actions:
            outdict["actions"] = actions
        if return_entropy:
            outdict["entropy"] = calculate_entropy(logprobs)
        if return_hidden:
            outdict["hidden"] = hidden
        if return_init_embeds:
            outdict["init_embeds"] = init_embeds
        return outdict


def pomo_policy(env_name="tsp", **kw) -> AttentionModelPolicy:
    """zoo/pomo/model.py:52-67: 6 layers, instance norm, no graph context, multistart decode types."""
    cfg = dict(
        env_name=env_name, num_encoder_layers=6, normalization="instance", use_graph_context=False,
        train_decode_type="multistart_sampling", val_decode_type="multistart_greedy",
        test_decode_type="multistart_greedy",
    )
    cfg.update(kw)
    return AttentionModelPolicy(**cfg)
