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


def _dataset_jobs(problem, data_distribution, graph_sizes, distributions_per_problem):
    """Every (problem, distribution, graph size) a call of ``generate_dataset`` covers, in the reference's order
    (generate_data.py:252-268): problems in table order, each with its distributions, each at every size."""
    table = DISTRIBUTIONS_PER_PROBLEM if distributions_per_problem is None else distributions_per_problem
    if isinstance(problem, (list, tuple)) and len(problem) == 1:
        problem = problem[0]
    if problem == "all":
        plan = dict(table)
    else:
        plan = {problem: table[problem] if data_distribution == "all" else [data_distribution]}
    sizes = [graph_sizes] if isinstance(graph_sizes, int) else list(graph_sizes)
    return [(prob, dist, size) for prob, dists in plan.items() for dist in (dists or [None]) for size in sizes]


def generate_dataset(filename=None, data_dir: str = "data", name: str | None = None, problem="all",
                     data_distribution: str = "all", dataset_size: int = 10000, graph_sizes=(20, 50, 100),
                     overwrite: bool = False, seed: int = 1234, distributions_per_problem: dict | None = None) -> list[str]:
    """The instance files of ``rl4co/data/generate_data.py:214-311``: one npz per job, by default at
    ``<data_dir>/<problem>/<problem>[_<dist>]<size>_<name>_seed<seed>.npz``; explicit ``filename``(s) are consumed one
    per job. Every file is drawn under a fresh ``np.random.seed(seed)`` so that a file does not depend on which other
    files the call wrote; existing files are kept unless ``overwrite``. Returns the files written."""
    jobs = _dataset_jobs(problem, data_distribution, graph_sizes, distributions_per_problem)
    explicit = None if filename is None else ([filename] if isinstance(filename, str) else list(filename))
    if explicit is not None and len(explicit) < len(jobs):
        raise ValueError("Number of filenames does not match number of problems")
    written = []
    for k, (prob, dist, size) in enumerate(jobs):
        if explicit is None:
            tag = "" if dist is None else f"_{dist}"
            path = os.path.join(data_dir, prob, f"{prob}{tag}{size}_{name}_seed{seed}.npz")
        else:
            path = check_extension(explicit[k], extension=".npz")
        if os.path.isfile(path) and not overwrite:
            continue
        if os.path.dirname(path):
            os.makedirs(os.path.dirname(path), exist_ok=True)
        np.random.seed(seed)
        np.savez(path, **generate_env_data(prob, dataset_size, size, dist))
        written.append(path)
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


# ---- augmentation (data/transforms.py) -> csrc/augment.hip -------------------------------------------
# The instances are read ONCE by a kernel that writes the aug-major [A*B, N, 2] layout the multistart rollout consumes
# (kernels.augment_dihedral8 / augment_symmetric); only the angle draw of the symmetric group stays on the host side of
# the boundary, because it IS host state: the reference takes it from torch's global generator (transforms.py:81) and a
# seeded run must consume that stream identically.

# Device contract (differs from the reference, which runs these on any device): the features must live on the GPU — a CPU
# tensor raises Rl4coLibraryError, like every other entry of the package (no CPU fallback) — in any floating dtype and
# layout (converted to contiguous fp32, the type the kernels and the reference's generators use).

_AUGMENT_KINDS = ("symmetric", "dihedral8")


def _f32c(x: Tensor) -> Tensor:
    return x if (x.dtype == torch.float32 and x.is_contiguous()) else x.float().contiguous()


def dihedral_8_augmentation(xy: Tensor) -> Tensor:
    """[B, N, 2] -> [8B, N, 2]: the 8 symmetries of the unit square, aug-major (transforms.py:16-38), one launch."""
    from . import kernels as K

    return K.augment_dihedral8(_f32c(xy))


def dihedral_8_augmentation_wrapper(xy: Tensor, reduce: bool = True, *args, **kw) -> Tensor:
    """transforms.py:41-46: a batchified feature carries 8 copies of the instances; with ``reduce`` the first copy is
    what gets augmented (so the output has as many rows as the input)."""
    return dihedral_8_augmentation(xy[: xy.shape[0] // 8] if reduce else xy)


def _rotation_terms(phi: Tensor):
    """(cos phi, sin phi, phi > 2 pi) of one angle per output row: what ``rl4co_augment_symmetric_f32`` takes."""
    phi = phi.float()
    return torch.cos(phi).contiguous(), torch.sin(phi).contiguous(), (phi > 2 * math.pi).contiguous()


def _draw_angles(rows: int, identity_rows: int, device, phi: Tensor | None = None) -> Tensor:
    """U[0, 4 pi) per output row from the global generator of ``device`` (transforms.py:81; on the GPU that is the device
    generator — a seeded run reproduces itself, not the reference's CPU stream: the parity tests inject the reference's
    angles through ``phi``), or the injected ``phi``; the first ``identity_rows`` rows get angle 0 — the identity."""
    phi = torch.rand(rows, device=device) * 4 * math.pi if phi is None else phi.to(device=device, dtype=torch.float32).clone()
    phi[:identity_rows] = 0.0
    return phi


def symmetric_augmentation(xy: Tensor, num_augment: int = 8, first_augment: bool = False, phi: Tensor | None = None) -> Tensor:
    """transforms.py:72-87 on an (already batchified) feature [R, N, 2]: every row is rotated about the centre of the
    unit square by its own angle and mirrored (axis swap) where the angle exceeds 2 pi; unless ``first_augment`` the
    first ``R / num_augment`` rows keep angle 0. ``phi`` injects the angles (parity tests)."""
    from . import kernels as K

    rows = xy.shape[0]
    phi = _draw_angles(rows, 0 if first_augment else rows // num_augment, xy.device, phi)
    return K.augment_symmetric(_f32c(xy), *_rotation_terms(phi))


def min_max_normalize(x: Tensor) -> Tensor:
    lo, hi = torch.aminmax(x)
    return (x - lo) / (hi - lo)


def get_augment_function(augment_fn):
    """transforms.py:94-103: a callable passes through, the two names map to this module's functions."""
    if callable(augment_fn):
        return augment_fn
    if augment_fn not in _AUGMENT_KINDS:
        raise ValueError(f"Unknown augment_fn: {augment_fn}. Available options: 'symmetric', 'dihedral8' or a custom callable")
    return symmetric_augmentation if augment_fn == "symmetric" else dihedral_8_augmentation_wrapper


def _tile_rows(v: Tensor, times: int) -> Tensor:
    """[B, ...] -> [times * B, ...], copy-major (the row order of utils/ops.py:10-30's batchify)."""
    return v.unsqueeze(0).expand(times, *v.shape).reshape(times * v.shape[0], *v.shape[1:])


class StateAugmentation:
    """The reference's ``StateAugmentation`` (transforms.py:105-151), argument for argument, on the device kernels.

    The output holds ``num_augment`` copies of the batch (copy-major rows); every feature in ``feats`` is replaced by its
    augmented version: "symmetric" (default; a fresh angle draw per feature, first copy the identity), "dihedral8" (the 8
    symmetries of the square; ``num_augment`` must be 8) or a callable ``fn(tiled_feature, num_augment)``. The named
    groups run as one kernel over the UNTILED feature; the other keys are tiled by one strided copy each."""

    def __init__(self, num_augment: int = 8, augment_fn="symmetric", first_aug_identity: bool = True,
                 normalize: bool = False, feats: list | None = None):
        self.augmentation = get_augment_function(augment_fn)
        self.kind = augment_fn if (not callable(augment_fn)) else None
        assert not (self.kind == "dihedral8" and num_augment != 8), (
            "When using the `dihedral8` augmentation function, then num_augment must be 8")
        self.feats = ["locs"] if feats is None else feats
        self.num_augment = num_augment
        self.normalize = normalize
        self.first_aug_identity = first_aug_identity

    def _augmented(self, base: Tensor) -> Tensor:
        from . import kernels as K

        a, b = self.num_augment, base.shape[0]
        if self.kind == "dihedral8":
            return K.augment_dihedral8(_f32c(base))
        if self.kind == "symmetric":
            return K.augment_symmetric(_f32c(base), *_rotation_terms(_draw_angles(a * b, b, base.device)))
        return self.augmentation(_tile_rows(base, a), a)

    def __call__(self, td):
        a, b = self.num_augment, td.batch_size[0]
        out = TensorDict({k: _tile_rows(v, a) for k, v in td.items() if k not in self.feats}, batch_size=[a * b])
        for feat in self.feats:
            new = self._augmented(td[feat])
            if self.normalize:
                new = min_max_normalize(new)
            if not self.first_aug_identity:
                # the reference saves and restores ONE coordinate pair around the augmentation — row `batch size` (the
                # first row of the second copy), node 0 — through its `[list(td.size()), 0]` index (transforms.py:139-147)
                new[b, 0] = td[feat][0, 0]
            out[feat] = new
        return out


# ---- POMO evaluation epilogue (zoo/pomo/model.py:88-143, phase != "train") -> csrc/augment.hip ----------------------

def pomo_evaluate(policy, env, td, num_augment: int = 8, num_starts: int | None = None, phase: str = "test",
                  augment_fn="dihedral8", first_aug_identity: bool = True, feats: list | None = None) -> dict:
    """The val / test branch of ``POMO.shared_step``: augment the reset state ``td``, roll every (augmentation, start)
    out greedily, keep the best start per augmentation and the best augmentation per instance. The two maxima and the
    two action gathers are ONE launch (``rl4co_pomo_best``) over the rollout's flat outputs, whose rows are ordered
    start-major over augmentation-major over instances. Keys as the reference's: ``max_reward`` [B, A] and
    ``best_multistart_actions`` [B, A, T] (several starts), ``max_aug_reward`` [B] and ``best_aug_actions`` [B, T] (several
    augmentations), ``actions`` regrouped to [B, A, S, T]; plus ``reward_per_aug_start`` [B, A, S]."""
    from . import kernels as K

    n_aug = int(num_augment)
    n_start = env.get_num_starts(td) if num_starts is None else int(num_starts)
    if n_aug > 1:
        td = StateAugmentation(num_augment=n_aug, augment_fn=augment_fn, first_aug_identity=first_aug_identity, feats=feats)(td)
    out = policy(td, env, phase=phase, num_starts=n_start)
    a, s = max(n_aug, 1), max(n_start, 1)
    flat_actions = out.get("actions", None)
    best = K.pomo_best(out["reward"].contiguous(), None if flat_actions is None else flat_actions.contiguous(), a, s)
    b = best["max_aug_reward"].shape[0]
    out["reward_per_aug_start"] = out["reward"].view(s, a, b).permute(2, 1, 0)
    if n_start > 1:
        out["max_reward"] = best["max_reward"]
        if flat_actions is not None:
            out["best_multistart_actions"] = best["best_multistart_actions"]
            out["actions"] = flat_actions.view(s, a, b, -1).permute(2, 1, 0, 3)
    if n_aug > 1:
        # (one explicit start leaves a unit axis on the reference's result: its unbatchify keeps the start axis of size 1)
        out["max_aug_reward"] = best["max_aug_reward"][:, None] if n_start == 1 else best["max_aug_reward"]
        if flat_actions is not None:
            out["best_aug_actions"] = best["best_aug_actions"]
    return out
