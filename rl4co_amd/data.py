"""The callers / data formats either side of the rollout.

* npz instance files with the reference's ``{locs, depot, demand, capacity}`` schema
  (``rl4co/data/utils.py:11-30``), loaded straight to the GPU; the file generators of
  ``rl4co/data/generate_data.py`` (same numpy draws under the same seed -> the same val / test files) and a
  batch-indexed dataset (``rl4co/data/dataset.py``) that keeps the instances on the device.
* dihedral-8 state augmentation (``rl4co/data/transforms.py:16-46,105-151``) and the POMO
  evaluation epilogue — best over starts, then best over augmentations
  (``rl4co/models/zoo/pomo/model.py:88-143``): pure gathers/maxima around the same fused rollout.
"""
from __future__ import annotations

import math
import os

import numpy as np
import torch
from torch import Tensor

from .tensordict import TensorDict


# ---- npz I/O (data/utils.py:11-39) ---------------------------------------------------------------

def check_extension(filename: str, extension: str = ".npz") -> str:
    return filename if os.path.splitext(filename)[1] == extension else filename + extension


def load_npz_to_tensordict(filename: str, device=None) -> TensorDict:
    """A npz of numpy arrays -> TensorDict with batch size = leading dim of the first array."""
    x = np.load(filename)
    x_dict = {k: torch.from_numpy(np.asarray(v)) for k, v in dict(x).items()}
    batch_size = next(iter(x_dict.values())).shape[0]
    td = TensorDict(x_dict, batch_size=[batch_size])
    return td.to(device) if device is not None else td


def save_tensordict_to_npz(td, filename: str, compress: bool = False) -> None:
    x_dict = {k: v.detach().cpu().numpy() for k, v in td.items()}
    (np.savez_compressed if compress else np.savez)(filename, **x_dict)


# ---- instance files (data/generate_data.py:24-311) -------------------------------------------------
# The reference's dataset files are plain numpy draws under np.random.seed(seed): the same calls in the
# same order reproduce its val / test files byte for byte (tests/test_data_cpu.py checks that against
# the reference source). Problems on the path: tsp, vrp (= CVRP), pdp, op, pctsp.

CAPACITIES = {10: 20.0, 15: 25.0, 20: 30.0, 30: 33.0, 40: 37.0, 50: 40.0, 60: 43.0, 75: 45.0, 100: 50.0, 125: 55.0,
              150: 60.0, 200: 70.0, 500: 100.0, 1000: 150.0}  # generate_data.py:44-58
MAX_LENGTHS = {20: 2.0, 50: 3.0, 100: 4.0}  # generate_data.py:103,128
DISTRIBUTIONS_PER_PROBLEM = {"tsp": [None], "vrp": [None], "pctsp": [None], "op": ["const", "unif", "dist"], "pdp": [None]}


def generate_tsp_data(dataset_size: int, tsp_size: int) -> dict:
    return {"locs": np.random.uniform(size=(dataset_size, tsp_size, 2)).astype(np.float32)}


def generate_vrp_data(dataset_size: int, vrp_size: int, capacities: dict | None = None) -> dict:
    caps = dict(CAPACITIES)
    if capacities is not None:
        caps.update({k: v for k, v in capacities.items() if k in caps})
    return {
        "depot": np.random.uniform(size=(dataset_size, 2)).astype(np.float32),
        "locs": np.random.uniform(size=(dataset_size, vrp_size, 2)).astype(np.float32),
        "demand": np.random.randint(1, 10, size=(dataset_size, vrp_size)).astype(np.float32),  # 1 ... 9, NOT normalised
        "capacity": np.full(dataset_size, caps[vrp_size]).astype(np.float32),
    }


def generate_pdp_data(dataset_size: int, pdp_size: int) -> dict:
    depot = np.random.uniform(size=(dataset_size, 2))
    loc = np.random.uniform(size=(dataset_size, pdp_size, 2))
    return {"locs": loc.astype(np.float32), "depot": depot.astype(np.float32)}


def generate_op_data(dataset_size: int, op_size: int, prize_type: str = "const", max_lengths: dict | None = None) -> dict:
    depot = np.random.uniform(size=(dataset_size, 2))
    loc = np.random.uniform(size=(dataset_size, op_size, 2))
    if prize_type == "const":
        prize = np.ones((dataset_size, op_size))
    elif prize_type == "unif":
        prize = (1 + np.random.randint(0, 100, size=(dataset_size, op_size))) / 100.0
    else:
        assert prize_type == "dist"
        prize_ = np.linalg.norm(depot[:, None, :] - loc, axis=-1)
        prize = (1 + (prize_ / prize_.max(axis=-1, keepdims=True) * 99).astype(int)) / 100.0
    max_lengths = MAX_LENGTHS if max_lengths is None else max_lengths
    return {"depot": depot.astype(np.float32), "locs": loc.astype(np.float32), "prize": prize.astype(np.float32),
            "max_length": np.full(dataset_size, max_lengths[op_size]).astype(np.float32)}


def generate_pctsp_data(dataset_size: int, pctsp_size: int, penalty_factor: float = 3, max_lengths: dict | None = None) -> dict:
    depot = np.random.uniform(size=(dataset_size, 2))
    loc = np.random.uniform(size=(dataset_size, pctsp_size, 2))
    max_lengths = MAX_LENGTHS if max_lengths is None else max_lengths
    penalty_max = max_lengths[pctsp_size] * (penalty_factor) / float(pctsp_size)
    penalty = np.random.uniform(size=(dataset_size, pctsp_size)) * penalty_max
    deterministic_prize = np.random.uniform(size=(dataset_size, pctsp_size)) * 4 / float(pctsp_size)
    stochastic_prize = np.random.uniform(size=(dataset_size, pctsp_size)) * deterministic_prize * 2
    return {"locs": loc.astype(np.float32), "depot": depot.astype(np.float32), "penalty": penalty.astype(np.float32),
            "deterministic_prize": deterministic_prize.astype(np.float32),
            "stochastic_prize": stochastic_prize.astype(np.float32)}


_GENERATORS = {"tsp": generate_tsp_data, "vrp": generate_vrp_data, "pdp": generate_pdp_data, "op": generate_op_data,
               "pctsp": generate_pctsp_data}


def generate_env_data(env_type: str, *args, **kwargs) -> dict:
    """generate_data.py:24-34 (``None`` arguments are dropped, as there)"""
    if env_type not in _GENERATORS:
        raise NotImplementedError(f"Environment type {env_type} not implemented")
    return _GENERATORS[env_type](*[a for a in args if a is not None], **kwargs)


def generate_dataset(filename=None, data_dir: str = "data", name: str | None = None, problem="all",
                     data_distribution: str = "all", dataset_size: int = 10000, graph_sizes=(20, 50, 100),
                     overwrite: bool = False, seed: int = 1234, distributions_per_problem: dict | None = None) -> list[str]:
    """generate_data.py:214-311: one npz per (problem, distribution, graph size), named
    ``<data_dir>/<problem>/<problem>[_<dist>]<size>_<name>_seed<seed>.npz``. Returns the files written."""
    if isinstance(problem, list) and len(problem) == 1:
        problem = problem[0]
    graph_sizes = [graph_sizes] if isinstance(graph_sizes, int) else list(graph_sizes)
    dpp = DISTRIBUTIONS_PER_PROBLEM if distributions_per_problem is None else distributions_per_problem
    problems = dpp if problem == "all" else {problem: (dpp[problem] if data_distribution == "all" else [data_distribution])}
    filenames = [filename] if isinstance(filename, str) else filename
    written, it = [], 0
    for prob, distributions in problems.items():
        for distribution in distributions or [None]:
            for graph_size in graph_sizes:
                if filename is None:
                    datadir = os.path.join(data_dir, prob)
                    os.makedirs(datadir, exist_ok=True)
                    fname = os.path.join(datadir, "{}{}{}_{}_seed{}.npz".format(
                        prob, (f"_{distribution}" if distribution is not None else ""), graph_size, name, seed))
                else:
                    if it >= len(filenames):
                        raise ValueError("Number of filenames does not match number of problems")
                    fname = check_extension(filenames[it], extension=".npz")
                    if os.path.dirname(fname):
                        os.makedirs(os.path.dirname(fname), exist_ok=True)
                    it += 1
                if not overwrite and os.path.isfile(fname):
                    continue
                np.random.seed(seed)
                dataset = generate_env_data(prob, dataset_size, graph_size, distribution)
                np.savez(fname, **dataset)
                written.append(fname)
    return written


def generate_default_datasets(data_dir: str) -> list[str]:
    """generate_data.py:314-317: the val (seed 4321) and test (seed 1234) files of every problem on the path"""
    return (generate_dataset(data_dir=data_dir, name="val", problem="all", seed=4321)
            + generate_dataset(data_dir=data_dir, name="test", problem="all", seed=1234))


# ---- datasets (data/dataset.py:14-130) -------------------------------------------------------------

class TensorDictDataset:
    """A TensorDict of instances served batch by batch (``__getitems__``): the tensors stay where they are (on the
    GPU after ``load_npz_to_tensordict(..., device=)``) and a batch is ONE gather per key — what the reference's
    ``FastTdDataset`` / ``TensorDictDatasetFastGeneration`` do, without the per-item dict disassembly of its default
    ``TensorDictDataset`` (data/dataset.py:41-71) that dominates an epoch once the rollout is fast."""

    def __init__(self, td):
        self.data = td
        self.data_len = td.batch_size[0]

    def __len__(self) -> int:
        return self.data_len

    def __getitems__(self, index):
        idx = torch.as_tensor(index, device=next(iter(self.data.values())).device)
        return TensorDict({k: v[idx] for k, v in self.data.items()}, batch_size=[idx.numel()])

    def __getitem__(self, i: int):
        return {k: v[i] for k, v in self.data.items()}

    def add_key(self, key: str, value: Tensor):
        """dataset.py:63-64,118-120: e.g. the rollout baseline's rewards ("extra") next to every instance"""
        assert len(value) == self.data_len, "Data and extra must be same length"
        self.data.set(key, value)
        return self

    @staticmethod
    def collate_fn(batch):
        """Batches come out of ``__getitems__`` assembled; a list of per-item dicts is stacked (dataset.py:66-73)."""
        if isinstance(batch, (list, tuple)):
            return TensorDict({k: torch.stack([b[k] for b in batch]) for k in batch[0].keys()}, batch_size=[len(batch)])
        return batch

    def batches(self, batch_size: int, shuffle: bool = False, generator: torch.Generator | None = None):
        """One epoch of device-resident batches (the DataLoader of rl/common/base.py:264-273 without worker processes)."""
        dev = next(iter(self.data.values())).device
        order = torch.randperm(self.data_len, generator=generator).to(dev) if shuffle else torch.arange(self.data_len, device=dev)
        for lo in range(0, self.data_len, batch_size):
            yield self.__getitems__(order[lo : lo + batch_size])


def greedy_rollout_rewards(policy, env, dataset: TensorDictDataset, batch_size: int = 64) -> Tensor:
    """``RolloutBaseline.rollout`` (rl/reinforce/baselines.py:218-237): the policy's greedy reward of every instance
    of the dataset, batch by batch through the fused rollout; instances and rewards stay on the device."""
    was_training = policy.training
    policy.eval()
    try:
        with torch.inference_mode():
            rewards = [policy(env.reset(batch), env, phase="test", decode_type="greedy")["reward"]
                       for batch in dataset.batches(batch_size)]
    finally:
        policy.train(was_training)
    return torch.cat(rewards, 0)


def wrap_dataset_with_baseline(policy, env, dataset: TensorDictDataset, batch_size: int = 64) -> TensorDictDataset:
    """``RolloutBaseline.wrap_dataset`` (baselines.py:239-248): the baseline policy's greedy reward rides with
    every instance as ``extra`` — evaluated once per epoch over the whole dataset (1.28 M instances by default,
    SURVEY.md §8f N2), which is why it runs through the same rollout kernels with the data kept on the GPU."""
    return dataset.add_key("extra", greedy_rollout_rewards(policy, env, dataset, batch_size).detach())


# ---- augmentation (data/transforms.py) -------------------------------------------------------------

def dihedral_8_augmentation(xy: Tensor) -> Tensor:
    """transforms.py:16-38: the 8 rotations/reflections of the unit square, aug-major [8*B, N, 2]."""
    x, y = xy.split(1, dim=2)
    zs = ((x, y), (1 - x, y), (x, 1 - y), (1 - x, 1 - y), (y, x), (1 - y, x), (y, 1 - x), (1 - y, 1 - x))
    return torch.cat([torch.cat(z, dim=2) for z in zs], dim=0)


def dihedral_8_augmentation_wrapper(xy: Tensor, reduce: bool = True, *args, **kw) -> Tensor:
    """transforms.py:41-46: with ``reduce`` only the first 1/8 of the (already batchified) rows is augmented."""
    xy = xy[: xy.shape[0] // 8, ...] if reduce else xy
    return dihedral_8_augmentation(xy)


def symmetric_transform(x: Tensor, y: Tensor, phi: Tensor, offset: float = 0.5) -> Tensor:
    """transforms.py:49-69 (SymNCO's rotation / reflection group, vectorised): rotate by ``phi`` about the centre of
    the unit square and swap the axes where ``phi > 2 pi`` (half of the draws)."""
    x, y = x - offset, y - offset
    x_prime = torch.cos(phi) * x - torch.sin(phi) * y
    y_prime = torch.sin(phi) * x + torch.cos(phi) * y
    mask = phi > 2 * math.pi
    xy = torch.cat((x_prime, y_prime), dim=-1)
    xy = torch.where(mask, xy.flip(-1), xy)
    return xy + offset


def symmetric_augmentation(xy: Tensor, num_augment: int = 8, first_augment: bool = False, phi: Tensor | None = None) -> Tensor:
    """transforms.py:72-87: one random angle in [0, 4 pi) per (batchified) row, drawn on ``xy``'s device from the
    global generator — a CPU tensor consumes the reference's stream exactly; the first ``B / num_augment`` rows keep
    ``phi = 0`` (the identity) unless ``first_augment``. ``phi`` injects the angles (parity tests on the GPU, whose
    generator is not the CPU's)."""
    if phi is None:
        phi = torch.rand(xy.shape[0], device=xy.device) * 4 * math.pi
    else:
        phi = phi.to(device=xy.device, dtype=xy.dtype).clone()
    if not first_augment:
        phi[: xy.shape[0] // num_augment] = 0.0
    x, y = xy[..., [0]], xy[..., [1]]
    return symmetric_transform(x, y, phi[:, None, None])


def min_max_normalize(x: Tensor) -> Tensor:
    return (x - x.min()) / (x.max() - x.min())


def get_augment_function(augment_fn):
    """transforms.py:94-103"""
    if callable(augment_fn):
        return augment_fn
    if augment_fn == "dihedral8":
        return dihedral_8_augmentation_wrapper
    if augment_fn == "symmetric":
        return symmetric_augmentation
    raise ValueError(f"Unknown augment_fn: {augment_fn}. Available options: 'symmetric', 'dihedral8' or a custom callable")


def _batchify(td, n: int):
    """utils/ops.py:10-30 for the TensorDict stand-in / the real TensorDict."""
    bs = td.batch_size[0]
    return td.expand(n, bs).contiguous().view(bs * n)


class StateAugmentation:
    """transforms.py:105-151, argument for argument: the batch is repeated ``num_augment`` times (aug-major) and every
    feature in ``feats`` is passed through ``augment_fn`` — "symmetric" (the default: random rotations / reflections,
    first block the identity), "dihedral8" (POMO: the 8 symmetries of the square) or a callable
    ``fn(batchified_feature, num_augment)``. Pure elementwise work on the device the instances live on."""

    def __init__(self, num_augment: int = 8, augment_fn="symmetric", first_aug_identity: bool = True,
                 normalize: bool = False, feats: list | None = None):
        self.augmentation = get_augment_function(augment_fn)
        assert not (self.augmentation == dihedral_8_augmentation_wrapper and num_augment != 8), (
            "When using the `dihedral8` augmentation function, then num_augment must be 8")
        self.feats = ["locs"] if feats is None else feats
        self.num_augment = num_augment
        self.normalize = normalize
        self.first_aug_identity = first_aug_identity

    def __call__(self, td):
        td_aug = _batchify(td, self.num_augment)
        for feat in self.feats:
            if not self.first_aug_identity:  # (the reference's own indexing: row `batch size`, node 0)
                init_aug_feat = td_aug[feat][list(td.size()), 0].clone()
            aug_feat = self.augmentation(td_aug[feat], self.num_augment)
            if self.normalize:
                aug_feat = min_max_normalize(aug_feat)
            if not self.first_aug_identity:
                aug_feat[list(td.size()), 0] = init_aug_feat
            td_aug[feat] = aug_feat
        return td_aug


# ---- POMO evaluation epilogue (zoo/pomo/model.py:88-143, phase != "train") ---------------------------

def _unbatchify(x: Tensor, shape) -> Tensor:
    """utils/ops.py:33-51 for tensors: [prod(shape)*B, ...] -> [B, *shape, ...]."""
    for s in reversed(shape):
        if s > 0:
            sh = x.shape
            x = x.view(s, sh[0] // s, *sh[1:]).permute(1, 0, *range(2, len(sh) + 1))
    return x


def _gather_by_index(src: Tensor, idx: Tensor, dim: int) -> Tensor:
    """utils/ops.py:54-66"""
    shape = list(src.shape)
    shape[dim] = -1
    idx = idx.view(idx.shape + (1,) * (src.dim() - idx.dim())).expand(shape)
    out = src.gather(dim, idx)
    return out.squeeze(dim) if idx.size(dim) == 1 else out


def pomo_evaluate(policy, env, td, num_augment: int = 8, num_starts: int | None = None, phase: str = "test",
                  augment_fn="dihedral8", first_aug_identity: bool = True, feats: list | None = None) -> dict:
    """val/test branch of ``POMO.shared_step``: augment (pomo/model.py:71-80 builds ``StateAugmentation(num_augment,
    augment_fn, first_aug_identity, feats)`` with dihedral-8 as POMO's default), multistart-greedy rollout, best start
    per augmentation, best augmentation per instance. ``td`` is a reset state (``env.reset(batch)``)."""
    n_aug = num_augment
    n_start = env.get_num_starts(td) if num_starts is None else num_starts
    if n_aug > 1:
        td = StateAugmentation(num_augment=n_aug, augment_fn=augment_fn, first_aug_identity=first_aug_identity, feats=feats)(td)
    out = policy(td, env, phase=phase, num_starts=n_start)
    reward = _unbatchify(out["reward"], (n_aug, n_start))
    out["reward_per_aug_start"] = reward
    if n_start > 1:
        max_reward, max_idxs = reward.max(dim=-1)
        out["max_reward"] = max_reward
        if out.get("actions", None) is not None:
            actions = _unbatchify(out["actions"], (n_aug, n_start))
            out["best_multistart_actions"] = _gather_by_index(actions, max_idxs, dim=max_idxs.dim())
            out["actions"] = actions
    if n_aug > 1:
        reward_ = out["max_reward"] if n_start > 1 else reward
        max_aug_reward, max_idxs = reward_.max(dim=1)
        out["max_aug_reward"] = max_aug_reward
        if out.get("actions", None) is not None:
            actions_ = out["best_multistart_actions"] if n_start > 1 else out["actions"]
            out["best_aug_actions"] = _gather_by_index(actions_, max_idxs, dim=1)
    return out
