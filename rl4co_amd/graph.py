"""The whole rollout as ONE captured HIP graph (`GraphedRollout`) — removes the host from between the kernels.

An inference rollout is a fixed sequence of launches whose arguments do not depend on the data: reset-state fills, the
fused encoder, state clones, the persistent decode launch, the validity check, the reward (its horizon is read on the
device, ``kernels.tour_length(horizon=)``) — and then ONE read-back. Issued from Python that sequence costs ~150 us of
GPU idle time per TSP-100 x 4096 rollout (17 dispatches; measured with tools/gap_summary.py: 5 % of the step). Captured
once with ``torch.cuda.graph`` (= hipStreamBeginCapture on the stream every ``rl4co_*`` entry point launches on) and
replayed with one ``hipGraphLaunch``, the dispatches run back to back; the host only copies the new instances into the
graph's static input buffer, replays, and reads the 16-byte status word.

    rollout = GraphedRollout(policy, env, example_batch, decode_type="greedy")
    out = rollout(batch)          # same dict as policy(env.reset(batch), env, phase="test", decode_type=...)

Semantics (those of every CUDA/HIP-graph wrapper): the returned tensors are views of buffers the graph owns and are
overwritten by the next call — clone what must outlive it; shapes, decode arguments and the policy's weights VALUES may
change between calls only in ways a fixed launch sequence tolerates (new weight values: the packed encoder buffers are
refreshed in place before the replay; a different batch shape: a new capture). Sampling draws fresh noise on every
replay through a device-resident seed word (``rl4co_am_decode_args.philox_seed_dev``). Inference only (no autograd).
"""
from __future__ import annotations

import torch

from .tensordict import TensorDict


class GraphedRollout:
    def __init__(self, policy, env, example, decode_type: str = "greedy", warmup: int = 2, **forward_kwargs):
        if not example["locs"].is_cuda:
            raise RuntimeError("GraphedRollout needs CUDA tensors (HIP graph capture)")
        self.policy, self.env = policy, env
        self.decode_type = decode_type
        self.kw = dict(forward_kwargs)
        self.batch = example.batch_size[0]
        dev = example["locs"].device
        # static input buffers: the graph reads the instances from here
        self.static_in = TensorDict({k: v.clone() for k, v in example.items() if torch.is_tensor(v)}, batch_size=[self.batch])
        self.seed_dev = torch.zeros(1, dtype=torch.int64, device=dev)
        self._calls = 0
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.inference_mode():
            for _ in range(max(1, warmup)):  # library init, hipFuncSetAttribute, allocator warm-up outside the capture
                self._enqueue()()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize(dev)
        # Did the warm-up rollouts take the fused-encoder path? Only that path reads PACKED copies of the weights (every
        # other encoder reads the live parameters inside the graph, so new weight values need nothing from us); the packed
        # encoder is never built, refreshed or version-checked for a policy that does not use it
        pe = getattr(policy, "_packed", None)
        self._fused = pe is not None and bool(pe.t) and pe.version == pe._current_version()
        self._captured: dict = {}
        self._packed_version = None
        if self._fused:
            # the captured launches point at THESE buffers for as long as the graph lives: hold them (a later eager
            # rollout after a weight update rebinds pe.t to fresh tensors — without this reference the captured ones
            # would be freed under the graph)
            self._captured = dict(pe.t)
            self._packed_version = pe.version
        self.graph = torch.cuda.CUDAGraph()
        with torch.inference_mode(), torch.cuda.graph(self.graph):
            self._finish = self._enqueue()
        if self._fused:
            assert all(pe.t[k] is v for k, v in self._captured.items()), "the capture re-packed the encoder weights"

    def _enqueue(self):
        td = self.env.reset(self.static_in)
        return self.policy(td, self.env, phase="test", decode_type=self.decode_type, philox_seed_dev=self.seed_dev,
                           _defer_finish=True, **self.kw)

    def _sync_packed_weights(self) -> None:
        """New weight values (or a packed encoder rebound by an eager rollout in between): re-pack, copy INTO the buffers
        the captured launches point at, and hand those buffers back to the packed encoder. The captured buffers hold the
        values of version `_packed_version` — nobody else writes them — so an unchanged version needs no copy whatever
        `pe.t` currently points at."""
        pe = self.policy._packed
        ver = pe._current_version()
        if ver == self._packed_version:
            return
        fresh = pe.refresh()
        for k, held in self._captured.items():
            new = fresh.get(k)
            if torch.is_tensor(held):
                if not torch.is_tensor(new) or new.shape != held.shape or new.dtype != held.dtype:
                    raise RuntimeError(f"packed encoder entry {k!r} changed layout: capture a new GraphedRollout")
                if new.data_ptr() != held.data_ptr():
                    held.copy_(new)
            elif new is not None:
                raise RuntimeError(f"packed encoder entry {k!r} appeared after the capture: capture a new GraphedRollout")
        pe.t = dict(self._captured)
        self._packed_version = ver

    def __call__(self, batch) -> dict:
        if batch.batch_size[0] != self.batch:
            raise ValueError(f"captured for {self.batch} instances, got {batch.batch_size[0]}")
        with torch.inference_mode():
            for k, v in self.static_in.items():
                src = batch[k]
                if src.data_ptr() != v.data_ptr():
                    v.copy_(src)
            if self._fused:
                self._sync_packed_weights()
            self._calls += 1
            self.seed_dev.fill_(self._calls * 0x9E3779B97F4A7C15 % (1 << 62))
            self.graph.replay()
            return self._finish()
