"""State container for the env / policy boundary.

The reference keeps rollout state in a ``tensordict.TensorDict``. That package is only a
container on this path; when it is installed we use it unchanged, otherwise this
minimal stand-in provides the handful of methods the boundary touches: mapping access,
``set/get/update``, ``batch_size``/``shape``/``device``, ``clone``, ``to`` and the
``expand/contiguous/view/permute`` quartet that ``batchify``/``unbatchify`` rely on
(utils/ops.py:10-51), and ``gather/squeeze`` over the batch dimensions (``unbatchify_and_gather``).
"""
from __future__ import annotations

import torch

try:  # pragma: no cover - not installed in the build container
    from tensordict import TensorDict  # type: ignore

    HAVE_TENSORDICT = True
except Exception:  # noqa: BLE001
    HAVE_TENSORDICT = False

    class TensorDict(dict):  # type: ignore[no-redef]
        def __init__(self, source=None, batch_size=None, device=None):
            super().__init__(source or {})
            if batch_size is None:
                batch_size = []
            if isinstance(batch_size, int):
                batch_size = [batch_size]
            self.batch_size = torch.Size(batch_size)
            self._device = device

        # -- container protocol ---------------------------------------------------------------
        @property
        def device(self):
            if self._device is not None:
                return torch.device(self._device)
            for v in self.values():
                if torch.is_tensor(v):
                    return v.device
            return None

        @property
        def shape(self):
            return self.batch_size

        def dim(self):
            return len(self.batch_size)

        def size(self, i=None):
            return self.batch_size if i is None else self.batch_size[i]

        def is_empty(self):
            return len(self) == 0

        def set(self, key, value):
            self[key] = value
            return self

        def exclude(self, *keys):
            return TensorDict({k: v for k, v in self.items() if k not in keys}, self.batch_size, self._device)

        def select(self, *keys):
            return TensorDict({k: self[k] for k in keys}, self.batch_size, self._device)

        def clone(self):
            return TensorDict(
                {k: (v.clone() if torch.is_tensor(v) else v) for k, v in self.items()},
                self.batch_size, self._device,
            )

        def to(self, device):
            return TensorDict(
                {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in self.items()},
                self.batch_size, device,
            )

        def _map(self, fn, batch_size):
            return TensorDict({k: fn(v) for k, v in self.items()}, batch_size, self._device)

        # -- the tensor-like methods batchify/unbatchify use ------------------------------------
        def expand(self, *shape):
            nb = len(self.batch_size)
            lead = tuple(shape[: len(shape) - nb])
            return self._map(lambda v: v.expand(*lead, *v.shape), torch.Size(shape))

        def contiguous(self):
            return self._map(lambda v: v.contiguous(), self.batch_size)

        def view(self, *shape):
            nb = len(self.batch_size)
            return self._map(lambda v: v.view(*shape, *v.shape[nb:]), torch.Size(shape))

        def permute(self, *dims):
            nb = len(self.batch_size)
            new_bs = torch.Size([self.batch_size[d] for d in dims])
            return self._map(
                lambda v: v.permute(*dims, *range(nb, v.dim())), new_bs
            )

        def gather(self, dim, index):
            """Every entry gathered along batch dimension ``dim`` (``unbatchify_and_gather``, utils/ops.py:69-74:
            the best-of-starts selection of the reference's decode loop applies it to the state it got back)."""
            nb = len(self.batch_size)

            def g(v):
                idx = index.view(*index.shape, *([1] * (v.dim() - nb))).expand(*index.shape, *v.shape[nb:])
                return v.gather(dim, idx)

            return self._map(g, index.shape)

        def squeeze(self, dim):
            bs = list(self.batch_size)
            if bs[dim] != 1:
                return self
            bs.pop(dim)
            return self._map(lambda v: v.squeeze(dim), torch.Size(bs))

        def __getitem__(self, key):
            if isinstance(key, str):
                return super().__getitem__(key)
            # batch indexing
            sub = {k: v[key] for k, v in self.items()}
            probe = torch.empty(self.batch_size, device="meta")[key]
            return TensorDict(sub, probe.shape, self._device)
