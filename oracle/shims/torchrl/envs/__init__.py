"""Stand-in for ``torchrl.envs.EnvBase``: the loop glue the reference env inherits.

Only what ``RL4COEnvBase`` (envs/common/base.py:19-333) relies on for the rollout path:
constructor arguments, ``reset`` (calls ``_reset`` and injects the ``done``/``terminated`` flags
from the done spec, shape [*batch, 1] bool — the key ``ConstructivePolicy.forward`` reads on the
first loop test), ``set_seed`` -> ``_set_seed``, ``to`` and ``batch_size``.
"""
from __future__ import annotations

import torch


class EnvBase:
    batch_locked = True

    def __init__(self, *, device=None, batch_size=None, run_type_checks=False, allow_done_after_reset=False, **_unused):
        self.device = torch.device(device) if device is not None else None
        self.batch_size = torch.Size(batch_size) if batch_size is not None else torch.Size([])
        self.run_type_checks = run_type_checks
        self.allow_done_after_reset = allow_done_after_reset

    def set_seed(self, seed=None, static_seed=False):
        if seed is not None:
            torch.manual_seed(seed)
        self._set_seed(seed)
        return seed

    def _set_seed(self, seed):
        rng = torch.manual_seed(seed)
        self.rng = rng

    def to(self, device):
        self.device = torch.device(device) if device is not None else None
        return self

    def reset(self, tensordict=None, **kwargs):
        td = self._reset(tensordict, **kwargs)
        bs = tuple(td.batch_size)
        dev = td.device
        for key in ("done", "terminated"):
            if key not in td.keys():
                td.set(key, torch.zeros((*bs, 1), dtype=torch.bool, device=dev))
        return td
