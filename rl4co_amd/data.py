"""The callers / data formats either side of the rollout.

* npz instance files with the reference's ``{locs, depot, demand, capacity}`` schema
  (``rl4co/data/utils.py:11-30``), loaded straight to the GPU.
* dihedral-8 state augmentation (``rl4co/data/transforms.py:16-46,105-151``) and the POMO
  evaluation epilogue — best over starts, then best over augmentations
  (``rl4co/models/zoo/pomo/model.py:88-143``): pure gathers/maxima around the same fused rollout.
"""
from __future__ import annotations

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


# ---- augmentation (data/transforms.py) -------------------------------------------------------------

def dihedral_8_augmentation(xy: Tensor) -> Tensor:
    """transforms.py:16-38: the 8 rotations/reflections of the unit square, aug-major [8*B, N, 2]."""
    x, y = xy.split(1, dim=2)
    zs = ((x, y), (1 - x, y), (x, 1 - y), (1 - x, 1 - y), (y, x), (1 - y, x), (y, 1 - x), (1 - y, 1 - x))
    return torch.cat([torch.cat(z, dim=2) for z in zs], dim=0)


def _batchify(td, n: int):
    """utils/ops.py:10-30 for the TensorDict stand-in / the real TensorDict."""
    bs = td.batch_size[0]
    return td.expand(n, bs).contiguous().view(bs * n)


class StateAugmentation:
    """transforms.py:105-151 with ``augment_fn="dihedral8"`` (POMO's default): the batch is
    repeated 8 times (aug-major) and ``locs`` of block k gets the k-th symmetry."""

    def __init__(self, num_augment: int = 8, augment_fn: str = "dihedral8", feats: list | None = None):
        if augment_fn != "dihedral8":
            raise NotImplementedError("only the dihedral-8 augmentation of POMO is on the accelerated path")
        assert num_augment == 8, "When using the `dihedral8` augmentation function, then num_augment must be 8"
        self.num_augment = num_augment
        self.feats = ["locs"] if feats is None else feats

    def __call__(self, td):
        td_aug = _batchify(td, self.num_augment)
        for feat in self.feats:
            x = td_aug[feat]
            td_aug[feat] = dihedral_8_augmentation(x[: x.shape[0] // 8])  # the wrapper's reduce=True
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


def pomo_evaluate(policy, env, td, num_augment: int = 8, num_starts: int | None = None, phase: str = "test") -> dict:
    """val/test branch of ``POMO.shared_step``: augment x8, multistart-greedy rollout, best start per
    augmentation, best augmentation per instance. ``td`` is a reset state (``env.reset(batch)``)."""
    n_aug = num_augment
    n_start = env.get_num_starts(td) if num_starts is None else num_starts
    if n_aug > 1:
        td = StateAugmentation(num_augment=n_aug)(td)
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
