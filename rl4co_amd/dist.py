"""One process per GPU: instance sharding and the single exchange step of the path.

The rollout shards embarrassingly by instance — no collective on the data path.
The reference's only communication is Lightning DDP's gradient bucket all-reduce
(``rl4co/utils/trainer.py:83-86``: ``DDPStrategy(find_unused_parameters=True,
gradient_as_bucket_view=True)``; AM-3L 710 144 params = 2.84 MB fp32, POMO-6L ≈ 5.2 MB — one
bucket) plus scalar metric reductions (``rl/common/base.py:220-227`` ``sync_dist=True``).

MI355X mapping: ONE all-reduce(sum) of one flat fp32 buffer per optimizer step over RCCL/xGMI
(backend "nccl" on ROCm), then divide by the world size — DDP's mean-of-per-rank-mean-loss
semantics. At a few MB the ring is latency-bound (7 xGMI links x ~153 GB/s per GPU), so a single
message beats per-parameter launches; the flat buffer is a persistent view the parameters' grads
alias (``gradient_as_bucket_view``), so there is no pack/unpack copy.
Backend-agnostic (gloo on CPU for the tests, nccl/RCCL on the GPUs).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist
from torch import Tensor, nn


def _free_port() -> int:
    import socket

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def init_process_group(backend: str | None = None, device: torch.device | None = None,
                       single_process_ok: bool = False) -> tuple[int, int]:
    """Initialise from the torchrun environment (RANK / WORLD_SIZE / MASTER_*). Returns (rank, world).

    ``single_process_ok``: also create the group for a lone process (world size 1, rendezvous on a free local port)
    — the gradient all-reduce then runs through the real backend (RCCL on the GPU) instead of being skipped, which
    is how the one-GPU bench and the ``-m gpu`` tests execute the collective path."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if (world > 1 or single_process_ok) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


def device_identity(device: torch.device | None) -> str:
    """What distinguishes this process's compute device from its peers': the GPU's UUID (falls back to its PCI bus id /
    its index) — for a CPU process the process id (every process is its own "device")."""
    if device is None or torch.device(device).type != "cuda":
        return f"cpu:{os.getpid()}"
    props = torch.cuda.get_device_properties(device)
    for attr in ("uuid", "pci_bus_id"):
        v = getattr(props, attr, None)
        if v is not None:
            return f"{attr}:{v}:{getattr(props, 'pci_device_id', '')}"
    return f"index:{torch.device(device).index}"


def check_placement(device: torch.device | None, expect_world: int, allow_shared: bool = False,
                    identity: str | None = None) -> dict:
    """Start-up self-check of a one-process-per-GPU job, BEFORE anything is timed (the reference leaves this to Lightning's
    DDP strategy, utils/trainer.py:73-86): the group has exactly ``expect_world`` ranks, every rank sits on a device of
    its own (two ranks time-sharing one GPU would report a "scaling" that is nothing of the kind), the process's current
    device is the one it was handed, and the backend's all-reduce really sums over all ranks. Raises ``RuntimeError``
    on any rank that sees a violation — every rank sees the same gathered table, so all of them fail together."""
    rank, world = world_info()
    if world != expect_world:
        raise RuntimeError(f"process group has {world} rank(s), the job was launched for {expect_world}")
    dev = torch.device(device) if device is not None else torch.device("cpu")
    if dev.type == "cuda" and torch.cuda.current_device() != (dev.index or 0):
        raise RuntimeError(f"rank {rank}: current device {torch.cuda.current_device()} is not the assigned {dev}")
    mine = (rank, identity or device_identity(dev), dev.index if dev.type == "cuda" else -1)
    table = [mine]
    if world > 1:
        table = [None] * world
        dist.all_gather_object(table, mine)
    idents = [t[1] for t in table]
    if sorted(t[0] for t in table) != list(range(world)):
        raise RuntimeError(f"ranks are not 0..{world - 1}: {table}")
    if len(set(idents)) != world and not allow_shared:
        raise RuntimeError(f"ranks share a device (one process per GPU expected): {table}")
    backend = dist.get_backend() if (dist.is_available() and dist.is_initialized()) else None
    if backend is not None:
        t = torch.full((4,), float(rank + 1), dtype=torch.float32, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        want = world * (world + 1) / 2
        if not bool((t == want).all()):
            raise RuntimeError(f"rank {rank}: all-reduce over {backend} returned {t.tolist()}, expected {want}")
    return {"world": world, "backend": backend, "distinct_devices": len(set(idents)), "devices": idents}


def world_info() -> tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_bounds(total: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous block [lo, hi) of ``total`` instances owned by ``rank`` (sizes differ by <= 1)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_instances(td, rank: int | None = None, world: int | None = None):
    """Slice a batch (TensorDict / dict of tensors / tensor) to this rank's contiguous shard."""
    r, w = world_info()
    rank = r if rank is None else rank
    world = w if world is None else world
    if isinstance(td, Tensor):
        lo, hi = shard_bounds(td.shape[0], rank, world)
        return td[lo:hi]
    if isinstance(td, dict) and not hasattr(td, "batch_size"):
        return {k: shard_instances(v, rank, world) for k, v in td.items()}
    lo, hi = shard_bounds(td.batch_size[0], rank, world)
    return td[lo:hi]


class FlatGradBucket:
    """All trainable gradients of a module as views into ONE flat fp32 buffer.

    ``allreduce_mean()`` = one collective per optimizer step (the reference's DDP bucket). Parameters
    that received no gradient this step contribute zeros (DDP ``find_unused_parameters=True``)."""

    def __init__(self, module: nn.Module):
        self.params = [p for p in module.parameters() if p.requires_grad]
        if not self.params:
            raise ValueError("module has no trainable parameters")
        device, dtype = self.params[0].device, torch.float32
        self.numel = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(self.numel, dtype=dtype, device=device)
        off = 0
        for p in self.params:
            if p.dtype != dtype:
                raise TypeError("FlatGradBucket expects fp32 master parameters")
            view = self.flat[off : off + p.numel()].view_as(p)
            if p.grad is not None:
                view.copy_(p.grad)
            p.grad = view  # autograd accumulates in place into the bucket from now on
            off += p.numel()

    @property
    def nbytes(self) -> int:
        return self.numel * 4

    def zero_(self) -> None:
        self.flat.zero_()

    def _rebind(self) -> None:
        """``optimizer.zero_grad(set_to_none=True)`` drops the views; re-attach them."""
        off = 0
        dst, src, empty = [], [], []
        for p in self.params:
            view = self.flat[off : off + p.numel()].view_as(p)
            if p.grad is None:
                empty.append(view)
                p.grad = view
            elif p.grad.data_ptr() != view.data_ptr():
                dst.append(view)
                src.append(p.grad)
                p.grad = view
            off += p.numel()
        if empty:
            torch._foreach_zero_(empty)
        if dst:
            torch._foreach_copy_(dst, src)  # one multi-tensor launch for all detached gradients

    def release(self) -> None:
        """Detach the gradient views before a backward pass (``p.grad = None``): autograd then hands every parameter its
        freshly computed gradient instead of ADDING it into the (zeroed) view — 80 five-microsecond add kernels per POMO
        step — and ``allreduce_mean`` gathers them with one multi-tensor copy. Equivalent to ``zero_()`` + accumulate."""
        for p in self.params:
            p.grad = None

    def allreduce_mean(self, async_op: bool = False):
        """sum over ranks / world size, in place. Returns the work handle when ``async_op``."""
        self._rebind()
        _, world = world_info()
        if not (dist.is_available() and dist.is_initialized()):
            return None  # no process group: a lone process, nothing to exchange
        # an initialised group of ONE rank still issues the collective (see init_process_group(single_process_ok))
        work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, async_op=True)
        if async_op:
            return _ScaledWork(work, self.flat, world)
        work.wait()
        if world > 1:
            self.flat.div_(world)
        return None


class _ScaledWork:
    def __init__(self, work, flat: Tensor, world: int):
        self.work, self.flat, self.world = work, flat, world

    def wait(self) -> None:
        self.work.wait()  # nccl: orders the current stream after the collective, does not block the host
        if self.world > 1:
            self.flat.div_(self.world)


def allreduce_scalars(values: dict[str, float | Tensor], device=None, op: str = "mean") -> dict[str, float]:
    """Logged metrics (``sync_dist=True``): ONE small all-reduce for all keys of a step."""
    _, world = world_info()
    keys = sorted(values)
    buf = torch.tensor([float(values[k]) for k in keys], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(buf, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM)
        if op == "mean":
            buf /= world
    return {k: float(v) for k, v in zip(keys, buf.tolist())}


def barrier() -> None:
    """Process-group barrier (no-op for a single process)."""
    if world_info()[1] > 1:
        dist.barrier()


def reduce_scalar(value: float, op: str = "max", device=None) -> float:
    """max / sum of one python scalar over the ranks — the timing aggregation of bench.py
    (wall time = max over ranks, work = sum over ranks)."""
    _, world = world_info()
    if world == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM)
    return float(t.item())


def broadcast_decision(flag: bool, src: int = 0, device=None) -> bool:
    """RolloutBaseline's per-rank t-test decision (reinforce/baselines.py:200-218) must agree on
    every rank: rank ``src`` decides, everyone follows."""
    _, world = world_info()
    if world == 1:
        return bool(flag)
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=device)
    dist.broadcast(t, src=src)
    return bool(int(t.item()))
