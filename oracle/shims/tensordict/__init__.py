"""Test-only stand-in for the `tensordict` package (a keyed tensor container with a batch shape).

Implements exactly the surface the reference's rollout path touches:
construction with ``batch_size``, mapping access, ``get/set/update/keys/items/values``, ``clone``,
``to``, ``device``, ``shape/batch_size/dim/size``, ``is_empty``, ``exclude/select`` and the
``expand/contiguous/view/permute`` methods ``batchify``/``unbatchify`` call (utils/ops.py:10-51), plus
``gather/squeeze`` over the batch dimensions for ``unbatchify_and_gather`` (utils/ops.py:69-74).
"""
from __future__ import annotations

import torch


__version__ = "0.6.0"


class TensorDictBase:
    pass


class TensorDict(TensorDictBase):
    def __init__(self, source=None, batch_size=None, device=None, **_unused):
        if isinstance(source, TensorDict):
            source = dict(source._data)
        self._data = dict(source or {})
        if batch_size is None:
            batch_size = []
        if isinstance(batch_size, int):
            batch_size = [batch_size]
        self._batch_size = torch.Size(batch_size)
        self._device = torch.device(device) if device is not None else None
        for k, v in self._data.items():
            self._check(k, v)

    # ---- bookkeeping ---------------------------------------------------------------------
    def _check(self, key, value):
        if torch.is_tensor(value):
            nb = len(self._batch_size)
            if tuple(value.shape[:nb]) != tuple(self._batch_size):
                raise RuntimeError(
                    f"batch dimension mismatch for {key!r}: {tuple(value.shape)} vs batch_size {tuple(self._batch_size)}"
                )

    @property
    def batch_size(self):
        return self._batch_size

    @batch_size.setter
    def batch_size(self, value):
        self._batch_size = torch.Size(value)

    @property
    def shape(self):
        return self._batch_size

    @property
    def device(self):
        if self._device is not None:
            return self._device
        devs = {v.device for v in self._data.values() if torch.is_tensor(v)}
        return devs.pop() if len(devs) == 1 else None

    def dim(self):
        return len(self._batch_size)

    def size(self, i=None):
        return self._batch_size if i is None else self._batch_size[i]

    def numel(self):
        n = 1
        for s in self._batch_size:
            n *= s
        return n

    def is_empty(self):
        return len(self._data) == 0

    # ---- mapping -------------------------------------------------------------------------
    def keys(self, *a, **k):
        return self._data.keys()

    def items(self):
        return self._data.items()

    def values(self):
        return self._data.values()

    def __contains__(self, key):
        return key in self._data

    def __len__(self):
        return self._batch_size[0] if len(self._batch_size) else 0

    def __iter__(self):
        raise TypeError("iteration over a TensorDict stand-in is not supported")

    def get(self, key, default=None):
        return self._data.get(key, default)

    def set(self, key, value, inplace=False):
        self._check(key, value)
        self._data[key] = value
        return self

    def __setitem__(self, key, value):
        if not isinstance(key, str):
            raise NotImplementedError("index assignment is outside the rollout path")
        self.set(key, value)

    def __getitem__(self, key):
        if isinstance(key, str):
            return self._data[key]
        sub = {k: v[key] for k, v in self._data.items()}
        probe = torch.empty(self._batch_size, device="meta")[key]
        return TensorDict(sub, probe.shape, self._device)

    def pop(self, key, default=None):
        return self._data.pop(key, default)

    def update(self, other, **_unused):
        items = other.items() if hasattr(other, "items") else other
        for k, v in items:
            self.set(k, v)
        return self

    def exclude(self, *keys):
        return TensorDict({k: v for k, v in self._data.items() if k not in keys}, self._batch_size, self._device)

    def select(self, *keys):
        return TensorDict({k: self._data[k] for k in keys}, self._batch_size, self._device)

    # ---- tensor-like ---------------------------------------------------------------------
    def _map(self, fn, batch_size):
        return TensorDict(
            {k: (fn(v) if torch.is_tensor(v) else v) for k, v in self._data.items()}, batch_size, self._device
        )

    def clone(self, recurse=True):
        return self._map((lambda v: v.clone()) if recurse else (lambda v: v), self._batch_size)

    def to(self, device, **_unused):
        out = self._map(lambda v: v.to(device), self._batch_size)
        out._device = torch.device(device) if device is not None else None
        return out

    def detach(self):
        return self._map(lambda v: v.detach(), self._batch_size)

    def expand(self, *shape):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)):
            shape = tuple(shape[0])
        nb = len(self._batch_size)
        return self._map(lambda v: v.expand(*shape, *v.shape[nb:]), torch.Size(shape))

    def contiguous(self):
        return self._map(lambda v: v.contiguous(), self._batch_size)

    def view(self, *shape):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)):
            shape = tuple(shape[0])
        nb = len(self._batch_size)
        return self._map(lambda v: v.view(*shape, *v.shape[nb:]), torch.Size(shape))

    def reshape(self, *shape):
        nb = len(self._batch_size)
        return self._map(lambda v: v.reshape(*shape, *v.shape[nb:]), torch.Size(shape))

    def permute(self, *dims):
        if len(dims) == 1 and isinstance(dims[0], (tuple, list)):
            dims = tuple(dims[0])
        nb = len(self._batch_size)
        new_bs = torch.Size([self._batch_size[d] for d in dims])
        return self._map(lambda v: v.permute(*dims, *range(nb, v.dim())), new_bs)

    def gather(self, dim, index):
        """Every entry gathered along batch dimension ``dim`` (the index is broadcast over the entry's feature
        dimensions) — what ``unbatchify_and_gather`` (utils/ops.py:69-74) asks of the best-of-starts selection."""
        nb = len(self._batch_size)

        def g(v):
            idx = index.view(*index.shape, *([1] * (v.dim() - nb))).expand(*index.shape, *v.shape[nb:])
            return v.gather(dim, idx)

        return self._map(g, index.shape)

    def squeeze(self, dim):
        bs = list(self._batch_size)
        if bs[dim] != 1:
            return self
        bs.pop(dim)
        return self._map(lambda v: v.squeeze(dim), torch.Size(bs))

    def __repr__(self):
        fields = ", ".join(f"{k}: {tuple(v.shape) if torch.is_tensor(v) else type(v).__name__}" for k, v in self._data.items())
        return f"TensorDict(batch_size={tuple(self._batch_size)}, {fields})"
