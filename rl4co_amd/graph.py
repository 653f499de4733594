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

``PipelinedRollout`` keeps two (or more) such graphs in flight on separate streams over a stream of batches: the next
batch's launches fill the CUs the previous batch's finishing decode waves release (measured +3 % on TSP-100 x 4096,
+25 % on CVRP-100 x 4096 whose trajectories end at different steps).
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
            self._act_dtype = pe.act_dtype  # the 16-bit regime the captured launches compute in
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
        if pe.act_dtype != self._act_dtype:
            # an eager rollout under the OTHER 16-bit regime ran in between and re-packed the weights in its element type:
            # bring the packed encoder back to the regime of the capture (no weight changed: the version below then tells)
            pe.refresh(act_dtype=self._act_dtype)
        ver = pe._current_version()
        if ver == self._packed_version:
            if any(pe.t.get(k) is not v for k, v in self._captured.items()):
                pe.t = dict(self._captured)  # rebound by that eager rollout: the captured buffers hold this version's values
            return
        fresh = pe.refresh(act_dtype=self._act_dtype)
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

    def enqueue(self, batch, call_index: int | None = None) -> None:
        """Copy the instances into the static input buffers and replay the graph on the CURRENT stream — no host sync.
        ``finish()`` (same stream) performs the rollout's one read-back. ``call_index``: position of this rollout in the
        caller's stream of submissions — ``PipelinedRollout`` passes ONE counter over all its slots, so two slots captured
        with the same ``seed=`` never replay the same noise (default: this rollout's own call count)."""
        if batch.batch_size[0] != self.batch:
            raise ValueError(f"captured for {self.batch} instances, got {batch.batch_size[0]}")
        with torch.inference_mode():
            for k, v in self.static_in.items():
                src = batch[k]
                if src.data_ptr() != v.data_ptr():
                    v.copy_(src, non_blocking=True)
            if self._fused:
                self._sync_packed_weights()
            self._calls += 1
            index = self._calls if call_index is None else int(call_index)
            self.seed_dev.fill_(index * 0x9E3779B97F4A7C15 % (1 << 62))
            self.graph.replay()

    def finish(self) -> dict:
        with torch.inference_mode():
            return self._finish()

    def __call__(self, batch) -> dict:
        self.enqueue(batch)
        return self.finish()


def _tensors_of(obj):
    """Every CUDA tensor reachable from a rollout's output dict (the TensorDict of the final state included)."""
    if torch.is_tensor(obj):
        if obj.is_cuda:
            yield obj
    elif isinstance(obj, dict) or hasattr(obj, "items"):
        for _, v in obj.items():
            yield from _tensors_of(v)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            yield from _tensors_of(v)


class PipelinedRollout:
    """Cross-batch overlap: ``depth`` captured rollouts in flight on ``depth`` HIP streams over a stream of batches.

    One rollout is a matrix-core-bound encoder launch followed by an HBM-bound persistent decode launch whose waves retire
    at different times (CVRP: trajectories end between ~110 and ~180 steps and the launch lasts as long as the longest).
    With a second rollout queued on another stream, the next batch's encoder and decode workgroups take the CUs the
    finishing one releases instead of waiting for its last straggler. Measured on MI355X (r03, tools/overlap_bench.py,
    profiles/r03_overlap_two_streams.json): TSP-100 x 4096 3.47 -> 3.35 ms per batch, CVRP-100 x 4096 4.13 -> 3.31 ms
    (+25 %: the ragged tail), identical results. (While a decode launch is fully resident it owns every register of the
    chip — 16 waves x 124 VGPRs per CU — so the overlap happens at the tails, not under the whole launch.)

        pipe = PipelinedRollout(policy, env, example_batch, decode_type="greedy")
        for out in pipe.map(batches):       # outputs in submission order; each valid until its slot is reused
            ...

    ``submit`` / ``collect`` give the same with explicit tickets. Outputs are views of the slot's graph buffers: clone
    what must outlive ``depth`` further submissions."""

    def __init__(self, policy, env, example, decode_type: str = "greedy", depth: int = 2, **forward_kwargs):
        dev = example["locs"].device
        self.depth = int(depth)
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(self.depth)]
        self.slots = []
        for s in self.streams:
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                self.slots.append(GraphedRollout(policy, env, example, decode_type=decode_type, **forward_kwargs))
        torch.cuda.synchronize(dev)
        self.policy = policy
        self._pending = [False] * self.depth
        self._next = 0
        self._submitted = 0  # ONE counter over all slots: the device-resident Philox seed word is derived from it
        self._weights = self._weights_version()

    def _weights_version(self):
        pe = getattr(self.policy, "_packed", None)
        return pe._current_version() if (pe is not None and self.slots[0]._fused) else None

    def submit(self, batch) -> int:
        """Enqueue one batch on the next slot; returns the ticket to ``collect``. The slot's previous result must have
        been collected (``map`` does that)."""
        k = self._next
        if self._pending[k]:
            raise RuntimeError("slot still holds an uncollected result: collect() it before submitting further batches")
        ver = self._weights_version()
        if ver != self._weights:
            # new weight values: the slots share the packed-weight buffers their graphs point at; drain everything, let
            # the first slot copy the new values in, and only then replay anywhere
            torch.cuda.synchronize()
            for slot, s in zip(self.slots, self.streams):
                with torch.cuda.stream(s), torch.inference_mode():
                    slot._sync_packed_weights()
            torch.cuda.synchronize()
            self._weights = ver
        s = self.streams[k]
        s.wait_stream(torch.cuda.current_stream())  # the batch may have been produced on the caller's stream
        self._submitted += 1
        with torch.cuda.stream(s):
            self.slots[k].enqueue(batch, call_index=self._submitted)
        self._pending[k] = True
        self._next = (k + 1) % self.depth
        return k

    def collect(self, ticket: int) -> dict:
        if not self._pending[ticket]:
            raise RuntimeError("nothing pending on this ticket")
        side = self.streams[ticket]
        with torch.cuda.stream(side):  # the read-back is ordered after the slot's replay: same stream
            out = self.slots[ticket].finish()
        # finish() launches further kernels on the slot's stream AFTER its read-back (the trimmed action copy, the summed
        # log-likelihood, rewards of the environments without a device-side horizon) and allocates their outputs from that
        # stream's pool: hand both over to the caller's stream
        cur = torch.cuda.current_stream()
        cur.wait_stream(side)
        for v in _tensors_of(out):
            v.record_stream(cur)  # (buffers owned by the graph's private pool are never freed while it lives: a no-op there)
        self._pending[ticket] = False
        return out

    def map(self, batches):
        """Outputs of ``batches`` in order, with up to ``depth`` rollouts in flight."""
        from collections import deque

        inflight = deque()
        for batch in batches:
            if len(inflight) == self.depth:
                yield self.collect(inflight.popleft())
            inflight.append(self.submit(batch))
        while inflight:
            yield self.collect(inflight.popleft())
