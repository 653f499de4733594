"""Test-only stand-in for the `lightning` package: just enough of ``LightningModule`` for the reference's RL modules
(rl4co/models/rl/common/base.py, reinforce/reinforce.py, zoo/pomo/model.py) to be CONSTRUCTED and to have their
``shared_step`` called directly from a test — the training loop, loggers, checkpoints and strategies of the real
package are out of scope (control plane). Holds no arithmetic."""
import inspect

import torch.nn as nn


class _HParams(dict):
    __getattr__ = dict.get

    def __setattr__(self, k, v):
        self[k] = v


class LightningModule(nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()
        self.__dict__.setdefault("_hparams", _HParams())  # POMO saves its hyper-parameters before super().__init__()
        self.trainer = None
        self.current_epoch = 0
        self.logged = []  # (dict, kwargs) of every log_dict call, for the tests

    @property
    def hparams(self):
        return self.__dict__.setdefault("_hparams", _HParams())

    @property
    def loggers(self):
        return None

    @property
    def device(self):
        try:
            return next(self.parameters()).device
        except StopIteration:
            import torch

            return torch.device("cpu")

    def save_hyperparameters(self, *args, ignore=None, logger=True, **kwargs):
        """The caller's __init__ arguments, like the real method (frame inspection), minus `ignore`."""
        frame = inspect.currentframe().f_back
        values = {k: v for k, v in frame.f_locals.items() if k not in ("self", "__class__", "kwargs")}
        values.update(frame.f_locals.get("kwargs", {}) or {})
        for k in ignore or []:
            values.pop(k, None)
        self.hparams.update(values)

    def log_dict(self, dictionary, **kwargs):
        self.logged.append((dict(dictionary), kwargs))

    def log(self, name, value, **kwargs):
        self.logged.append(({name: value}, kwargs))
