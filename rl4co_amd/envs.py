"""Host-side mirror of the reference environment surface for TSP, CVRP and the routing environments that share
their decode kernel (OP, PCTSP / SPCTSP, PDP, CVRPTW).

Same names, arguments and error behaviour as ``RL4COEnvBase`` (envs/common/base.py:19-333),
``TSPEnv`` (envs/routing/tsp/env.py:22-192) and ``CVRPEnv`` (envs/routing/cvrp/env.py:22-256)
for the methods on the rollout path — ``reset``, ``step``, ``get_reward``,
``check_solution_validity``, ``get_action_mask``, ``get_num_starts``, ``select_start_nodes`` —
with the arithmetic done by the HIP kernels behind ``include/rl4co_amd.h``.

Differences a caller can observe, both deliberate:
  * state tensors are updated IN PLACE by the kernels (the reference re-allocates them every
    step and swaps them into the TensorDict);
  * validity asserts are evaluated on the device and raised once, by ``get_reward``, with the
    reference's messages, instead of synchronising the host every step.
``dataset`` / ``load_data`` (base.py:234-286) serve npz instance files straight to the device (rl4co_amd/data.py).
Out of scope (not on the rollout path): rendering, local search, torchrl specs.
"""
from __future__ import annotations

import os

import torch
import torch.nn.functional as F
from torch import Tensor

from . import kernels as K
from .tensordict import TensorDict

CAPACITIES = {  # cvrp/generator.py:15-30 (Kool et al. 2019 and follow-ups)
    10: 20.0, 15: 25.0, 20: 30.0, 30: 33.0, 40: 37.0, 50: 40.0, 60: 43.0, 75: 45.0,
    100: 50.0, 125: 55.0, 150: 60.0, 200: 70.0, 500: 100.0, 1000: 150.0,
}


class Generator:
    """envs/common/utils.py:19-31"""

    def __call__(self, batch_size) -> TensorDict:
        batch_size = [batch_size] if isinstance(batch_size, int) else list(batch_size)
        return self._generate(batch_size)

    def _generate(self, batch_size) -> TensorDict:
        raise NotImplementedError


class TSPGenerator(Generator):
    """tsp/generator.py:14-58 (uniform locations; other samplers are out of scope).

    ``device`` chooses where the instances are drawn: "cpu" reproduces the reference stream of
    the global torch generator exactly; a CUDA device draws directly into HBM (row N2 of §8f)."""

    def __init__(self, num_loc: int = 20, min_loc: float = 0.0, max_loc: float = 1.0, device="cpu"):
        self.num_loc = num_loc
        self.min_loc = min_loc
        self.max_loc = max_loc
        self.device = device

    def _uniform(self, shape, low, high, demand_capacity=None):
        """U(low, high). "cpu": the reference's own sampler on the global torch generator (its exact stream). A CUDA
        device: ONE launch of rl4co_uniform_f32 straight into HBM, keyed by a seed drawn from the (CPU) torch generator —
        `torch.manual_seed` still makes the instances reproducible, nothing is synchronised or uploaded.
        ``demand_capacity``: CVRP's integer demands over the capacity in the same launch."""
        if str(self.device) == "cpu":
            v = torch.distributions.Uniform(low=low, high=high).sample(shape)
            return v if demand_capacity is None else (v.int() + 1).float() / demand_capacity
        seed = int(torch.randint(0, 2**62, (1,)).item())
        return K.uniform(shape, low, high, seed, 0, self.device, demand_capacity=demand_capacity)

    def _generate(self, batch_size) -> TensorDict:
        locs = self._uniform((*batch_size, self.num_loc, 2), self.min_loc, self.max_loc)
        return TensorDict({"locs": locs}, batch_size=batch_size)


class CVRPGenerator(TSPGenerator):
    """cvrp/generator.py:33-140"""

    def __init__(self, num_loc: int = 20, min_loc: float = 0.0, max_loc: float = 1.0,
                 min_demand: int = 1, max_demand: int = 10, vehicle_capacity: float = 1.0,
                 capacity: float | None = None, device="cpu"):
        super().__init__(num_loc, min_loc, max_loc, device)
        self.min_demand = min_demand
        self.max_demand = max_demand
        self.vehicle_capacity = vehicle_capacity
        if capacity is None:
            capacity = CAPACITIES.get(num_loc, None)
        if capacity is None:
            closest = min(CAPACITIES.keys(), key=lambda x: abs(x - num_loc))
            capacity = CAPACITIES[closest]
        self.capacity = capacity

    def _generate(self, batch_size) -> TensorDict:
        locs = self._uniform((*batch_size, self.num_loc + 1, 2), self.min_loc, self.max_loc)
        depot = locs[..., 0, :]
        locs = locs[..., 1:, :]
        demand = self._uniform((*batch_size, self.num_loc), self.min_demand - 1, self.max_demand - 1,
                               demand_capacity=self.capacity)  # (U.int() + 1).float() / capacity, cvrp/generator.py:127-136
        capacity = torch.full((*batch_size, 1), self.capacity, device=demand.device)
        return TensorDict(
            {"locs": locs, "depot": depot, "demand": demand, "capacity": capacity},
            batch_size=batch_size,
        )


class CVRPTWGenerator(CVRPGenerator):
    """cvrptw/generator.py:13-158: CVRP data (depot sampled on its own) plus integer-valued time windows inside
    [distance from the depot, max_time - distance back] and zero service times; unscaled unless ``scale``"""

    def __init__(self, num_loc: int = 20, min_loc: float = 0.0, max_loc: float = 150.0, min_demand: int = 1,
                 max_demand: int = 10, vehicle_capacity: float = 1.0, capacity: float | None = None,
                 max_time: float = 480, scale: bool = False, device="cpu", **unused):
        super().__init__(num_loc, min_loc, max_loc, min_demand, max_demand, vehicle_capacity, capacity, device)
        self.min_time, self.max_time, self.scale = 0.0, max_time, scale

    def _generate(self, batch_size) -> TensorDict:
        depot = self._uniform((*batch_size, 2), self.min_loc, self.max_loc)
        locs = self._uniform((*batch_size, self.num_loc, 2), self.min_loc, self.max_loc)
        demand = self._uniform((*batch_size, self.num_loc), self.min_demand - 1, self.max_demand - 1)
        demand = (demand.int() + 1).float()
        dev = locs.device
        capacity = torch.full((*batch_size, 1), self.capacity, device=dev)
        durations = torch.zeros(*batch_size, self.num_loc + 1, dtype=torch.float32, device=dev)
        dist = (depot[..., None, :] - locs).norm(p=2, dim=-1)
        dist = torch.cat((torch.zeros(*batch_size, 1, device=dev), dist), dim=-1)
        upper_bound = self.max_time - dist - durations
        if str(self.device) == "cpu":
            ts_1, ts_2 = torch.rand(*batch_size, self.num_loc + 1), torch.rand(*batch_size, self.num_loc + 1)
        else:
            ts_1 = torch.rand(*batch_size, self.num_loc + 1, device=dev)
            ts_2 = torch.rand(*batch_size, self.num_loc + 1, device=dev)
        min_ts = (dist + (upper_bound - dist) * ts_1).int()
        max_ts = (dist + (upper_bound - dist) * ts_2).int()
        min_times, max_times = torch.min(min_ts, max_ts), torch.max(min_ts, max_ts)
        min_times[..., 0] = 0
        max_times[..., 0] = int(self.max_time)
        mask = min_times == max_times  # cvrptw/generator.py:113-133: windows must not be empty
        if bool(mask.any()):
            min_times = torch.where(mask, torch.max(dist.int(), min_times - 1), min_times)
            mask = min_times == max_times
            if bool(mask.any()):
                widened = torch.min(torch.floor(upper_bound).int(),
                                    torch.max(torch.ceil(min_times + durations).int(), max_times + 1))
                max_times = torch.where(mask, widened, max_times)
        if self.scale:
            durations, min_times, max_times = durations / self.max_time, min_times / self.max_time, max_times / self.max_time
            depot, locs = depot / self.max_time, locs / self.max_time
        time_windows = torch.stack((min_times, max_times), dim=-1)
        assert bool((min_times < max_times).all()), \
            "Please make sure the relation between max_loc and max_time allows for feasible solutions."
        return TensorDict({"locs": locs, "depot": depot, "demand": demand / self.capacity, "capacity": capacity,
                           "durations": durations, "time_windows": time_windows}, batch_size=batch_size)


OP_MAX_LENGTHS = {20: 2.0, 50: 3.0, 100: 4.0}  # op/generator.py:13


class OPGenerator(TSPGenerator):
    """op/generator.py:16-142 (uniform locations, depot sampled with them; prize_type const | unif | dist)"""

    def __init__(self, num_loc: int = 20, min_loc: float = 0.0, max_loc: float = 1.0, prize_type: str = "dist",
                 max_length: float | None = None, device="cpu", **unused):
        super().__init__(num_loc, min_loc, max_loc, device)
        assert prize_type in ("dist", "unif", "const"), f"Invalid prize_type: {prize_type}"
        self.prize_type = prize_type
        if max_length is None:
            max_length = OP_MAX_LENGTHS.get(num_loc, None)
        if max_length is None:
            closest = min(OP_MAX_LENGTHS.keys(), key=lambda x: abs(x - num_loc))
            max_length = OP_MAX_LENGTHS[closest]
        self.max_length = max_length

    def _generate(self, batch_size) -> TensorDict:
        locs_with_depot = self._uniform((*batch_size, self.num_loc + 1, 2), self.min_loc, self.max_loc)
        dev = locs_with_depot.device
        if self.prize_type == "const":
            prize = torch.ones(*batch_size, self.num_loc, device=dev)
        elif self.prize_type == "unif":
            prize = (1 + torch.randint(0, 100, (*batch_size, self.num_loc), device=dev).float()) / 100
        else:  # the distance to the depot, quantised to 1..100 hundredths (op/generator.py:119-121)
            prize = (locs_with_depot[..., 0:1, :] - locs_with_depot[..., 1:, :]).norm(p=2, dim=-1)
            prize = (1 + (prize / prize.max(dim=-1, keepdim=True)[0] * 99).int()).float() / 100
        max_length = torch.full((*batch_size,), self.max_length, device=dev)
        return TensorDict({"locs": locs_with_depot[..., 1:, :], "depot": locs_with_depot[..., 0, :], "prize": prize,
                           "max_length": max_length}, batch_size=batch_size)


class PCTSPGenerator(TSPGenerator):
    """pctsp/generator.py:37-128: uniform locations (depot sampled with them), penalties U(0, max_penalty),
    deterministic prizes U(0, 4/n) and stochastic prizes U(0, 2) x deterministic"""

    def __init__(self, num_loc: int = 20, min_loc: float = 0.0, max_loc: float = 1.0, penalty_factor: float = 3.0,
                 prize_required: float = 1.0, max_penalty: float | None = None, device="cpu", **unused):
        super().__init__(num_loc, min_loc, max_loc, device)
        self.prize_required = prize_required
        if max_penalty is None:
            max_penalty = OP_MAX_LENGTHS.get(num_loc, None)  # the same table as OP (pctsp/generator.py:13)
        if max_penalty is None:
            closest = min(OP_MAX_LENGTHS.keys(), key=lambda x: abs(x - num_loc))
            max_penalty = OP_MAX_LENGTHS[closest]
        self.max_penalty = max_penalty * penalty_factor / num_loc  # Kool et al. (2019) scaling

    def _generate(self, batch_size) -> TensorDict:
        locs_with_depot = self._uniform((*batch_size, self.num_loc + 1, 2), self.min_loc, self.max_loc)
        penalty = self._uniform((*batch_size, self.num_loc), 0.0, self.max_penalty)
        det = self._uniform((*batch_size, self.num_loc), 0.0, 4.0 / self.num_loc)
        sto = self._uniform((*batch_size, self.num_loc), 0.0, 2.0) * det
        return TensorDict({"locs": locs_with_depot[..., 1:, :], "depot": locs_with_depot[..., 0, :], "penalty": penalty,
                           "deterministic_prize": det, "stochastic_prize": sto}, batch_size=batch_size)


class PDPGenerator(TSPGenerator):
    """pdp/generator.py:30-88: uniform locations, depot sampled with them; an even number of locations
    (the first half are pickups, node j + n/2 is the delivery of pickup j)"""

    def __init__(self, num_loc: int = 20, min_loc: float = 0.0, max_loc: float = 1.0, device="cpu", **unused):
        super().__init__(num_loc + (num_loc % 2), min_loc, max_loc, device)  # pdp/generator.py:48-51

    def _generate(self, batch_size) -> TensorDict:
        locs = self._uniform((*batch_size, self.num_loc + 1, 2), self.min_loc, self.max_loc)
        return TensorDict({"locs": locs[..., 1:, :], "depot": locs[..., 0, :]}, batch_size=batch_size)


def _zero_state(device, **fields) -> dict:
    """Zero-initialised state tensors as views of ONE allocation (one fill launch instead of one per field: reset
    sits between two rollouts with the GPU idle). fields: name=(shape, dtype); every view starts 16-byte aligned."""
    offs, total = {}, 0
    for name, (shape, dtype) in fields.items():
        nbytes = int(torch.tensor([], dtype=dtype).element_size())
        for d in shape:
            nbytes *= int(d)
        offs[name] = (total, nbytes)
        total += (nbytes + 15) // 16 * 16
    buf = torch.zeros(max(total, 16), dtype=torch.uint8, device=device)
    return {name: buf[o : o + nb].view(fields[name][1]).view(fields[name][0]) for name, (o, nb) in offs.items()}


class RL4COEnvBase:
    """envs/common/base.py:19-333, rollout-path methods only."""

    name = "base"
    has_depot = False

    def __init__(self, *, generator: Generator | None = None, generator_params: dict | None = None,
                 check_solution: bool = True, device="cuda", seed: int | None = None, **unused):
        self.check_solution = check_solution
        self.device = torch.device(device)
        self.data_dir = unused.get("data_dir", "data/")  # base.py:57-75: per-phase instance files (npz), optional
        for phase in ("train", "val", "test"):
            f = unused.get(f"{phase}_file", None)
            setattr(self, f"{phase}_file", None if f is None else os.path.join(self.data_dir, f))
        self.generator = generator if generator is not None else self._default_generator(**(generator_params or {}))
        if seed is not None:
            torch.manual_seed(seed)

    # -- RL4COEnvBase.reset (base.py:135-143) ---------------------------------------------------
    def reset(self, td: TensorDict | None = None, batch_size=None) -> TensorDict:
        if batch_size is None:
            batch_size = [] if td is None else td.batch_size
        # base.py:135-143 tests `td.is_empty()` (no KEYS), not len(): with the real tensordict package len() is the
        # leading batch size, which is 0 / undefined for a populated TensorDict with batch_size=[]
        if td is None or (td.is_empty() if hasattr(td, "is_empty") else len(td) == 0):
            td = self.generator(batch_size=batch_size)
        batch_size = [batch_size] if isinstance(batch_size, int) else list(batch_size)
        td = td.to(self.device)
        out = self._reset(td, batch_size=batch_size)
        if "terminated" not in out.keys():  # torchrl's EnvBase.reset adds `done` and `terminated`
            out.set("terminated", torch.zeros_like(out["done"]))
        return out

    # -- RL4COEnvBase.step (base.py:121-133) ----------------------------------------------------
    def step(self, td: TensorDict) -> dict:
        return {"next": self._step(td)}

    # -- RL4COEnvBase.get_reward (base.py:180-190) ------------------------------------------------
    def get_reward(self, td: TensorDict, actions: Tensor, check_solution: bool | None = None, horizon=None) -> Tensor:
        """``horizon = (steps_dev, t_add)`` (tour-length environments): ``actions`` is a padded buffer whose real
        length the device knows (kernels.tour_length) — the policy's way to issue the reward before its read-back."""
        check_solution = self.check_solution if check_solution is None else check_solution
        if check_solution:
            self.check_solution_validity(td, actions)
        return self._get_reward(td, actions) if horizon is None else self._get_reward(td, actions, horizon=horizon)

    def accepts_reward_horizon(self) -> bool:
        """True when ``_get_reward`` takes the device-side ``horizon`` (this package's tour-length environments). A user
        subclass that overrides ``_get_reward(self, td, actions)`` with the reference's two-argument signature does not:
        the policy then computes its reward after the read-back, from the unpadded actions, as the reference does."""
        import inspect

        try:
            return "horizon" in inspect.signature(type(self)._get_reward).parameters
        except (TypeError, ValueError):
            return False

    # -- RL4COEnvBase.dataset / load_data (base.py:234-286) ----------------------------------------
    def dataset(self, batch_size=(), phase: str = "train", filename: str | None = None):
        """Instances of one phase as a dataset: loaded from ``<phase>_file`` / ``filename`` (npz, straight to this
        env's device) or generated on the device. A missing file falls back to generation, as in the reference."""
        from .data import TensorDictDataset

        f = getattr(self, f"{phase}_file", None) if filename is None else filename
        batch_size = [batch_size] if isinstance(batch_size, int) else list(batch_size)
        td = None
        if f is not None:
            try:
                td = self.load_data(f, batch_size, device=self.device)
            except FileNotFoundError:
                td = None
        if td is None:
            td = self.generator(batch_size=batch_size)
        return TensorDictDataset(td)

    @staticmethod
    def load_data(fpath, batch_size=(), device=None):
        from .data import load_npz_to_tensordict

        return load_npz_to_tensordict(fpath, device=device)

    def get_num_starts(self, td) -> int:
        """ops.py:115-125"""
        num_starts = td["action_mask"].shape[-1]
        return num_starts - 1 if self.has_depot else num_starts

    def select_start_nodes(self, td, num_starts: int) -> Tensor:
        """ops.py:128-161"""
        num_loc = getattr(self.generator, "num_loc", 0xFFFFFFFF)
        batch = td["action_mask"].shape[0]
        return K.select_start_nodes(batch, num_starts, num_loc, self.has_depot, td["action_mask"].device)

    def to(self, device):
        if device is not None:
            self.device = torch.device(device)
        return self

    # subclasses
    def _default_generator(self, **kw) -> Generator:
        raise NotImplementedError

    def _reset(self, td, batch_size):
        raise NotImplementedError

    def _step(self, td):
        raise NotImplementedError

    def _get_reward(self, td, actions):
        raise NotImplementedError

    def check_solution_validity(self, td, actions) -> None:
        raise NotImplementedError


class TSPEnv(RL4COEnvBase):
    name = "tsp"
    has_depot = False

    def _default_generator(self, **kw):
        return TSPGenerator(**kw)

    def _reset(self, td: TensorDict, batch_size) -> TensorDict:
        """tsp/env.py:88-113 (+ torchrl's done flag)"""
        init_locs = td["locs"].contiguous()
        device = init_locs.device
        b = init_locs.shape[0]
        num_loc = init_locs.shape[-2]
        z = _zero_state(device, first_node=((b,), torch.int64), current_node=((b,), torch.int64), i=((b, 1), torch.int64),
                        reward=((b, 1), torch.float32), done=((b,), torch.bool))
        return TensorDict(
            {
                "locs": init_locs,
                "first_node": z["first_node"],
                "current_node": z["current_node"],
                "i": z["i"],
                "action_mask": torch.ones((b, num_loc), dtype=torch.bool, device=device),
                "reward": z["reward"],
                "done": z["done"],
            },
            batch_size=[b],
        )

    def _step(self, td: TensorDict) -> TensorDict:
        """tsp/env.py:60-86 via rl4co_tsp_step (in place)."""
        K.tsp_step(td["action"].contiguous(), td["action_mask"], td["first_node"], td["current_node"],
                   td["i"], td["done"])
        return td

    def _get_reward(self, td: TensorDict, actions: Tensor, horizon=None) -> Tensor:
        """tsp/env.py:150-156"""
        return K.tour_length(td["locs"].contiguous(), actions.contiguous(), prepend_depot=False, negate=True, horizon=horizon)

    def check_solution_validity(self, td: TensorDict, actions: Tensor, err: Tensor | None = None) -> None:
        """tsp/env.py:158-164. With ``err`` the violation bits are OR-ed into the caller's error
        word and the host check is left to the caller (one sync per rollout)."""
        own = err is None
        err = K.new_error_word(actions.device) if own else err
        K.tsp_check_solution(actions.contiguous(), td["locs"].shape[-2], err)
        if own:
            K.raise_if_error(err)


class CVRPEnv(RL4COEnvBase):
    name = "cvrp"
    has_depot = True

    def _default_generator(self, **kw):
        return CVRPGenerator(**kw)

    def _reset(self, td: TensorDict, batch_size) -> TensorDict:
        """cvrp/env.py:98-124"""
        device = td["locs"].device
        b = td["locs"].shape[0]
        n = td["locs"].shape[-2] + 1
        td_reset = TensorDict(
            {
                "locs": torch.cat((td["depot"][:, None, :], td["locs"]), -2).contiguous(),
                "demand": td["demand"].contiguous(),
                "current_node": torch.zeros(b, 1, dtype=torch.long, device=device),
                "used_capacity": torch.zeros((b, 1), device=device),
                "vehicle_capacity": torch.full((b, 1), self.generator.vehicle_capacity, device=device),
                "visited": torch.zeros((b, n), dtype=torch.uint8, device=device),
                "action_mask": torch.zeros((b, n), dtype=torch.bool, device=device),
                "done": torch.zeros((b,), dtype=torch.bool, device=device),
            },
            batch_size=[b],
        )
        self.get_action_mask(td_reset)
        return td_reset

    def _step(self, td: TensorDict) -> TensorDict:
        """cvrp/env.py:66-96 via rl4co_cvrp_step (in place, mask included)."""
        K.cvrp_step(td["action"].contiguous(), td["demand"], td["used_capacity"], td["vehicle_capacity"],
                    td["visited"], td["current_node"], td["action_mask"], td["done"])
        return td

    def get_action_mask(self, td: TensorDict) -> Tensor:
        """cvrp/env.py:126-136 (recomputed in place into td['action_mask'])."""
        K.cvrp_step(None, td["demand"], td["used_capacity"], td["vehicle_capacity"], td["visited"],
                    td["current_node"], td["action_mask"], None)
        return td["action_mask"]

    @staticmethod
    def load_data(fpath, batch_size=(), device=None):
        """cvrp/env.py:179-186: the instance files hold integer demands 1..9 and the capacity; normalise to [0, 1]"""
        from .data import load_npz_to_tensordict

        td = load_npz_to_tensordict(fpath, device=device)
        td.set("demand", td["demand"] / td["capacity"][:, None])
        return td

    def _get_reward(self, td: TensorDict, actions: Tensor, horizon=None) -> Tensor:
        """cvrp/env.py:138-147"""
        return K.tour_length(td["locs"].contiguous(), actions.contiguous(), prepend_depot=True, negate=True, horizon=horizon)

    def check_solution_validity(self, td: TensorDict, actions: Tensor, err: Tensor | None = None) -> None:
        """cvrp/env.py:149-177 (trailing depot padding is neutral). ``err``: see TSPEnv."""
        own = err is None
        err = K.new_error_word(actions.device) if own else err
        K.cvrp_check_solution(actions.contiguous(), td["demand"], td["vehicle_capacity"].reshape(-1).contiguous(), err)
        if own:
            K.raise_if_error(err)


class CVRPTWEnv(CVRPEnv):
    """CVRP with time windows (envs/routing/cvrptw/env.py:16-199): a customer can only be entered while its window
    is open on arrival; the vehicle waits for the window to open, serves, and the clock restarts at the depot.
    ``time_windows`` keeps the generator's dtype (integers unless scaled); the kernels read fp32 copies
    (exact: the values are below 2**24)."""

    name = "cvrptw"

    def _default_generator(self, **kw):
        return CVRPTWGenerator(**kw)

    @staticmethod
    def _tw(td):
        return td["time_windows"].float().contiguous(), td["durations"].float().contiguous()

    def _reset(self, td: TensorDict, batch_size) -> TensorDict:
        """cvrptw/env.py:115-139"""
        td_reset = super()._reset(td, batch_size)
        b = td["locs"].shape[0]
        td_reset.set("current_time", torch.zeros(b, 1, dtype=torch.float32, device=td["locs"].device))
        td_reset.set("durations", td["durations"])
        td_reset.set("time_windows", td["time_windows"])
        self.get_action_mask(td_reset)
        return td_reset

    def _step(self, td: TensorDict) -> TensorDict:
        """cvrptw/env.py:97-113 then cvrp/env.py:66-96, one kernel (rl4co_cvrptw_step), in place"""
        tw, dur = self._tw(td)
        K.cvrptw_step(td["action"].contiguous(), td["demand"], td["locs"], tw, dur, td["used_capacity"],
                      td["vehicle_capacity"], td["current_time"], td["visited"], td["current_node"], td["action_mask"],
                      td["done"])
        self._attach_distances(td)
        return td

    def get_action_mask(self, td: TensorDict) -> Tensor:
        """cvrptw/env.py:83-95 (recomputed in place into td['action_mask'])"""
        if "time_windows" not in td.keys():  # CVRPEnv._reset asks for the mask before the windows are attached
            return super().get_action_mask(td)
        tw, dur = self._tw(td)
        K.cvrptw_step(None, td["demand"], td["locs"], tw, dur, td["used_capacity"], td["vehicle_capacity"],
                      td["current_time"], td["visited"], td["current_node"], td["action_mask"], None)
        self._attach_distances(td)
        return td["action_mask"]

    @staticmethod
    def _attach_distances(td: TensorDict) -> None:
        """cvrptw/env.py:88-90: the reference leaves `current_loc` and `distances` (from the current node to every node)
        in the state as side products of its mask; the kernels recompute them, these copies are for readers of the td"""
        cur = td["current_node"].reshape(-1, 1, 1).expand(-1, 1, 2)
        current_loc = td["locs"].gather(1, cur).squeeze(1)
        td.set("current_loc", current_loc)
        td.set("distances", (current_loc[:, None, :] - td["locs"]).norm(p=2, dim=-1))

    def check_solution_validity(self, td: TensorDict, actions: Tensor, err: Tensor | None = None) -> None:
        """cvrptw/env.py:146-190: the CVRP check, then the window data assertions and the deadline replay"""
        own = err is None
        err = K.new_error_word(actions.device) if own else err
        super().check_solution_validity(td, actions, err=err)
        tw, dur = self._tw(td)
        K.cvrptw_check_solution(actions.contiguous(), td["locs"].contiguous(), tw, dur, err)
        if own:
            K.raise_if_error(err)


class OPEnv(RL4COEnvBase):
    """Orienteering problem (envs/routing/op/env.py:16-194): collect prizes and be back at the depot
    within ``max_length``. State and arithmetic live in ``rl4co_op_*`` (csrc/env_step.hip)."""

    name = "op"
    has_depot = True

    def _default_generator(self, **kw):
        return OPGenerator(**kw)

    def _reset(self, td: TensorDict, batch_size) -> TensorDict:
        """op/env.py:100-135; ``max_length`` becomes the per-node entry limit table"""
        device = td["locs"].device
        b = td["locs"].shape[0]
        n = td["locs"].shape[-2] + 1
        locs = torch.cat((td["depot"][:, None, :], td["locs"]), -2).contiguous()
        td_reset = TensorDict(
            {
                "locs": locs,
                "prize": F.pad(td["prize"], (1, 0), mode="constant", value=0).contiguous(),  # 0 for the depot
                "tour_length": torch.zeros(b, device=device),
                "max_length": K.op_max_length(locs, td["max_length"]),
                "current_node": torch.zeros(b, 1, dtype=torch.long, device=device),
                "visited": torch.zeros((b, n), dtype=torch.bool, device=device),  # bool as in the reference; the kernels see its uint8 storage
                "current_total_prize": torch.zeros(b, dtype=torch.float, device=device),
                "i": torch.zeros((b,), dtype=torch.int64, device=device),
                "action_mask": torch.zeros((b, n), dtype=torch.bool, device=device),
                "done": torch.zeros((b,), dtype=torch.bool, device=device),
            },
            batch_size=[b],
        )
        self.get_action_mask(td_reset)
        return td_reset

    def _step(self, td: TensorDict) -> TensorDict:
        """op/env.py:67-98 via rl4co_op_step (in place, mask included)"""
        action = td["action"].contiguous()
        td["current_total_prize"] += td["prize"].gather(1, action[:, None]).squeeze(1)
        K.op_step(action, td["locs"], td["max_length"], td["tour_length"], td["visited"], td["current_node"], td["i"],
                  td["action_mask"], td["done"])
        return td

    def get_action_mask(self, td: TensorDict) -> Tensor:
        """op/env.py:137-154 (recomputed in place into td['action_mask'])"""
        K.op_step(None, td["locs"], td["max_length"], td["tour_length"], td["visited"], td["current_node"], td["i"],
                  td["action_mask"], td["done"])
        return td["action_mask"]

    def select_start_nodes(self, td, num_starts: int) -> Tensor:
        """ops.py:128-161, orienteering branch: when an instance has fewer feasible customers than starts (nodes too far
        to be entered within max_length), the start nodes are resampled from the feasible ones with replacement
        (torch.multinomial on the mask, s-major "b n -> (n b)") instead of the s % num_loc + 1 rule."""
        feasible = td["action_mask"][..., 1:].float()
        if bool((feasible.sum(-1) < num_starts).any()):
            return (torch.multinomial(feasible, num_starts, replacement=True) + 1).t().reshape(-1)
        return super().select_start_nodes(td, num_starts)

    def _get_reward(self, td: TensorDict, actions: Tensor) -> Tensor:
        """op/env.py:156-166"""
        if actions.size(-1) == 1:
            assert bool((actions == 0).all()), "If all length 1 tours, they should be zero"
            return torch.zeros(actions.size(0), dtype=torch.float, device=actions.device)
        return K.gather_sum(td["prize"], actions.contiguous())

    def check_solution_validity(self, td: TensorDict, actions: Tensor, err: Tensor | None = None) -> None:
        """op/env.py:168-194. ``err``: see TSPEnv."""
        own = err is None
        err = K.new_error_word(actions.device) if own else err
        K.op_check_solution(actions.contiguous(), td["locs"], td["max_length"], err)
        if own:
            K.raise_if_error(err)


class PCTSPEnv(RL4COEnvBase):
    """Prize-collecting TSP (envs/routing/pctsp/env.py:17-219, deterministic prizes): the tour may
    return to the depot once a total prize of 1 is collected; the cost is the tour length plus the
    penalties of the customers left out. State and arithmetic live in ``rl4co_pctsp_*``."""

    name = "pctsp"
    has_depot = True
    _stochastic = False

    def _default_generator(self, **kw):
        return PCTSPGenerator(**kw)

    def _reset(self, td: TensorDict, batch_size) -> TensorDict:
        """pctsp/env.py:93-139"""
        device = td["locs"].device
        b = td["locs"].shape[0]
        n = td["locs"].shape[-2] + 1
        expected_prize = td["deterministic_prize"]
        real_prize = td["stochastic_prize"] if self._stochastic else td["deterministic_prize"]
        penalty = td["penalty"]
        td_reset = TensorDict(
            {
                "locs": torch.cat((td["depot"][:, None, :], td["locs"]), -2).contiguous(),
                "current_node": torch.zeros(b, dtype=torch.long, device=device),
                "expected_prize": expected_prize.contiguous(),
                "real_prize": F.pad(real_prize, (1, 0), mode="constant", value=0).contiguous(),  # 0 for the depot
                "penalty": F.pad(penalty, (1, 0), mode="constant", value=0).contiguous(),
                "cur_total_prize": torch.zeros(b, device=device),
                "cur_total_penalty": penalty.sum(-1),  # sum all penalties (minus the visited ones)
                "visited": torch.zeros((b, n), dtype=torch.bool, device=device),  # bool as in the reference; the kernels see its uint8 storage
                "prize_required": torch.full((b,), float(self.generator.prize_required), device=device),
                "i": torch.zeros((b,), dtype=torch.int64, device=device),
                "action_mask": torch.zeros((b, n), dtype=torch.bool, device=device),
                "done": torch.zeros((b,), dtype=torch.bool, device=device),
            },
            batch_size=[b],
        )
        self.get_action_mask(td_reset)
        return td_reset

    def _step(self, td: TensorDict) -> TensorDict:
        """pctsp/env.py:62-91 via rl4co_pctsp_step (in place, mask included)"""
        action = td["action"].contiguous()
        td["cur_total_penalty"] += td["penalty"].gather(1, action[:, None]).squeeze(1)  # pctsp/env.py:67-69
        K.pctsp_step(action, td["real_prize"], td["cur_total_prize"], td["visited"], td["current_node"], td["i"],
                     td["action_mask"], td["done"])
        return td

    def get_action_mask(self, td: TensorDict) -> Tensor:
        """pctsp/env.py:141-148 (recomputed in place into td['action_mask'])"""
        K.pctsp_step(None, td["real_prize"], td["cur_total_prize"], td["visited"], td["current_node"], td["i"],
                     td["action_mask"], td["done"])
        return td["action_mask"]

    def _get_reward(self, td: TensorDict, actions: Tensor) -> Tensor:
        """pctsp/env.py:150-173: saved penalties - (tour length from the depot + all penalties)"""
        if actions.size(-1) == 1:
            assert bool((actions == 0).all()), "If all length 1 tours, they should be zero"
            return torch.zeros(actions.size(0), dtype=torch.float, device=actions.device)
        actions = actions.contiguous()
        length = K.tour_length(td["locs"].contiguous(), actions, prepend_depot=True, negate=False)
        saved = K.gather_sum(td["penalty"], actions)
        n = td["penalty"].shape[-1]
        every = torch.arange(1, n, device=actions.device).expand(actions.shape[0], n - 1).contiguous()
        return saved - (length + K.gather_sum(td["penalty"], every))  # the three sums in the reference's order

    def check_solution_validity(self, td: TensorDict, actions: Tensor, err: Tensor | None = None) -> None:
        """pctsp/env.py:175-201. ``err``: see TSPEnv."""
        own = err is None
        err = K.new_error_word(actions.device) if own else err
        K.pctsp_check_solution(actions.contiguous(), td["real_prize"], err)
        if own:
            K.raise_if_error(err)


class SPCTSPEnv(PCTSPEnv):
    """Stochastic PCTSP (envs/routing/spctsp/env.py:8-21): the prize collected at a node is the generator's
    ``stochastic_prize``; the policy still embeds the expected one. Same kernels as PCTSP."""

    name = "spctsp"
    _stochastic = True


class PDPEnv(RL4COEnvBase):
    """Pickup and delivery problem (envs/routing/pdp/env.py:17-225): every pickup before its delivery, one
    vehicle of unlimited capacity, tour closed through the depot. State in ``rl4co_pdp_*`` (csrc/env_step.hip)."""

    name = "pdp"
    has_depot = True

    def __init__(self, *, force_start_at_depot: bool = False, **kw):
        super().__init__(**kw)
        self.force_start_at_depot = force_start_at_depot

    def _default_generator(self, **kw):
        return PDPGenerator(**kw)

    def _reset(self, td: TensorDict, batch_size) -> TensorDict:
        """pdp/env.py:101-150"""
        device = td["locs"].device
        b = td["locs"].shape[0]
        n = td["locs"].shape[-2]
        assert n % 2 == 0, "PDP needs an even number of locations (pickup / delivery pairs)"
        to_deliver = torch.zeros((b, n + 1), dtype=torch.bool, device=device)
        to_deliver[:, : n // 2 + 1] = True  # the depot and the pickups; deliveries open with their pickup
        available = torch.ones((b, n + 1), dtype=torch.bool, device=device)
        action_mask = torch.ones((b, n + 1), dtype=torch.bool, device=device)
        if self.force_start_at_depot:
            action_mask[:, 1:] = False
        else:
            action_mask = action_mask & to_deliver
            available[:, 0] = False  # the depot is added by get_reward
            action_mask[:, 0] = False
        return TensorDict(
            {
                "locs": torch.cat((td["depot"][:, None, :], td["locs"]), -2).contiguous(),
                "current_node": torch.zeros(b, 1, dtype=torch.long, device=device),
                "to_deliver": to_deliver,
                "available": available,
                "i": torch.zeros((b, 1), dtype=torch.int64, device=device),
                "action_mask": action_mask,
                "done": torch.zeros((b,), dtype=torch.bool, device=device),
            },
            batch_size=[b],
        )

    def _step(self, td: TensorDict) -> TensorDict:
        """pdp/env.py:64-99 via rl4co_pdp_step (in place)"""
        K.pdp_step(td["action"].contiguous(), td["available"], td["to_deliver"], td["current_node"], td["i"],
                   td["action_mask"], td["done"])
        return td

    def get_action_mask(self, td: TensorDict) -> Tensor:
        """pdp/env.py:79 (recomputed in place into td['action_mask'])"""
        K.pdp_step(None, td["available"], td["to_deliver"], td["current_node"], td["i"], td["action_mask"], td["done"])
        return td["action_mask"]

    def _get_reward(self, td: TensorDict, actions: Tensor, horizon=None) -> Tensor:
        """pdp/env.py:191-202"""
        return K.tour_length(td["locs"].contiguous(), actions.contiguous(), prepend_depot=True, negate=True, horizon=horizon)

    def check_solution_validity(self, td: TensorDict, actions: Tensor, err: Tensor | None = None) -> None:
        """pdp/env.py:204-223. ``err``: see TSPEnv."""
        own = err is None
        err = K.new_error_word(actions.device) if own else err
        K.pdp_check_solution(actions.contiguous(), td["locs"].shape[-2], self.force_start_at_depot, err)
        if own:
            K.raise_if_error(err)

    def get_num_starts(self, td) -> int:
        """pdp/env.py:225-227: only the pickups can start a tour"""
        return (td["locs"].shape[-2] - 1) // 2

    def select_start_nodes(self, td, num_starts: int) -> Tensor:
        """pdp/env.py:229-238"""
        half = (td["locs"].shape[-2] - 1) // 2
        return K.select_start_nodes(td["action_mask"].shape[0], num_starts, half, True, td["action_mask"].device)


def get_env(name: str, **kw) -> RL4COEnvBase:
    return {"tsp": TSPEnv, "cvrp": CVRPEnv, "op": OPEnv, "pctsp": PCTSPEnv, "pdp": PDPEnv, "cvrptw": CVRPTWEnv,
            "spctsp": SPCTSPEnv}[name](**kw)
