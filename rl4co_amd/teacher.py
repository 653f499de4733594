"""Differentiable teacher-forced log-likelihood on the HIP backward kernel (``csrc/am_teacher.hip``).

Training (REINFORCE / POMO, ``rl/reinforce/reinforce.py:99-102``) differentiates the
log-likelihood of the sampled trajectories. The rollout kernel already produced the forward values
(per-step log-probs); this module supplies their gradient w.r.t. the folded decoder cache through
one launch of ``rl4co_am_teacher_backward`` and lets torch autograd carry it on through the fold
GEMMs and the encoder. Falls back to ``policy.evaluate_log_probs`` (pure torch) for graphs larger
than the kernel supports.
"""
from __future__ import annotations

import ctypes as C

import torch
from torch import Tensor

from . import _lib
from .cache import EMBED_DIM, FoldedCache, fold_weights

_vp, _i32, _i64, _f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float


class AmTeacherArgs(C.Structure):
    """Mirror of ``struct rl4co_am_teacher_args`` (field order and types must match the header)."""

    _fields_ = [
        ("env", _i32), ("B", _i32), ("B_inst", _i32), ("N", _i32), ("T", _i32), ("t0", _i32),
        ("mask_inner", _i32), ("mask_logits", _i32), ("tanh_clipping", _f32), ("temperature", _f32),
        ("cache_dtype", _i32), ("variant", _i32),
        ("glimpse_key", _vp), ("glimpse_val", _vp), ("logit_key", _vp),
        ("kvl_row_stride", _i64), ("kvl_batch_stride", _i64),
        ("ctx_first", _vp), ("ctx_cur", _vp), ("q_bias", _vp), ("q_step0", _vp), ("w_cap", _vp),
        ("actions", _vp), ("demand", _vp), ("vehicle_capacity", _vp), ("locs", _vp), ("max_length", _vp),
        ("time_windows", _vp), ("durations", _vp), ("w_time", _vp), ("grad_logp", _vp),
        ("d_kvl", _vp), ("d_ctx_first", _vp), ("d_ctx_cur", _vp), ("d_q_bias", _vp), ("d_q_step0", _vp),
        ("d_w_cap", _vp), ("d_w_time", _vp), ("logp_out", _vp), ("err", _vp),
        ("d_planes_bf16", _vp), ("d_planes_row_stride", C.c_int64), ("d_planes_batch_stride", C.c_int64),
        ("d_planes_plane_stride", C.c_int64),
        ("ctx_dtype", _i32), ("d_ctx_in_planes", _i32), ("ctx_row_stride", _i64), ("ctx_batch_stride", _i64),
    ]


VARIANT_IDS = {"auto": 0, "replay": 1, "mma": 2}


def max_nodes() -> int:
    return _lib.lib().rl4co_am_teacher_max_nodes()


def supports(env_name: str, cache_dtype: torch.dtype, num_nodes: int) -> bool:
    """TSP, CVRP, orienteering, prize-collecting TSP, pickup-delivery, CVRP with time windows — on both variants (the
    MMA one needs bf16 planes; fp32 planes take the replay kernel)."""
    if num_nodes > max_nodes() or cache_dtype not in (torch.float32, torch.bfloat16, torch.float16):
        return False
    return env_name in ("tsp", "cvrp", "op", "pctsp", "pdp", "cvrptw")


backward_events: list | None = None  # set to [] by bench.py to time the teacher-forced backward launches


def run_backward(cache: FoldedCache, actions: Tensor, grad_logp: Tensor, meta: dict, variant: str = "auto",
                 want_logp: bool = False, d_planes: Tensor | None = None) -> dict:
    """One launch of ``rl4co_am_teacher_backward``: dL/d(folded cache) for L = sum grad_logp * log p.

    ``variant``: "replay" (``csrc/am_teacher.hip``, fp32 step-by-step), "mma"
    (``csrc/am_teacher_mma.hip``, 16-step blocks on the matrix cores, bf16 planes) or "auto".
    Returns the gradient tensors, the variant that ran and (``want_logp``) the recomputed log-probs.
    ``d_planes``: a bf16 [3, B_inst, N, 128] view (any plane / instance / node strides, unit channel stride) that
    receives the three plane gradients instead of a fresh fp32 ``d_kvl`` (MMA variant only).
    """
    b, t = actions.shape
    b_inst, n = cache.num_instances, cache.num_nodes
    dev = actions.device
    tsp = cache.env_name == "tsp"
    f32 = dict(dtype=torch.float32, device=dev)
    d_kvl = torch.empty((3, b_inst, n, EMBED_DIM), **f32) if d_planes is None else None
    d_ctx_cur = torch.empty((b_inst, n, EMBED_DIM), **f32)
    d_ctx_first = torch.zeros((b_inst, n, EMBED_DIM), **f32) if tsp else None
    d_q_bias = torch.empty((b_inst, EMBED_DIM), **f32) if cache.q_bias is not None else None
    d_extra = torch.zeros((EMBED_DIM,), **f32)
    d_time = torch.zeros((EMBED_DIM,), **f32) if cache.w_time is not None else None
    logp = torch.zeros((b, t), **f32) if want_logp else None
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    a = AmTeacherArgs()
    a.env = {"tsp": _lib.ENV_TSP, "cvrp": _lib.ENV_CVRP, "op": _lib.ENV_OP, "pctsp": _lib.ENV_PCTSP, "pdp": _lib.ENV_PDP, "cvrptw": _lib.ENV_CVRPTW}[cache.env_name]
    a.B, a.B_inst, a.N, a.T, a.t0 = b, b_inst, n, t, int(meta["t0"])
    a.mask_inner, a.mask_logits = int(meta["mask_inner"]), int(meta["mask_logits"])
    a.tanh_clipping, a.temperature = float(meta["tanh_clipping"]), float(meta["temperature"])

    a.cache_dtype = _lib.dtype_id(cache.kvl.dtype)
    a.variant = VARIANT_IDS[variant]
    a.glimpse_key, a.glimpse_val, a.logit_key = (cache.plane(i).data_ptr() for i in range(3))
    a.kvl_row_stride, a.kvl_batch_stride = cache.row_stride, cache.batch_stride
    ptr = lambda x: None if x is None else x.data_ptr()  # noqa: E731
    a.ctx_first, a.ctx_cur, a.q_bias = ptr(cache.ctx_first), ptr(cache.ctx_cur), ptr(cache.q_bias)
    if cache.ctx_cur.dtype != torch.float32:  # context tables in the planes' 16-bit type (strided views of the fold GEMM's output)
        cc = cache.ctx_cur
        assert cc.dtype == cache.kvl.dtype and cc.dim() == 3 and cc.stride(2) == 1
        assert cache.ctx_first is None or (cache.ctx_first.dtype == cc.dtype and cache.ctx_first.stride() == cc.stride())
        a.ctx_dtype, a.ctx_row_stride, a.ctx_batch_stride = _lib.dtype_id(cc.dtype), cc.stride(1), cc.stride(0)
    a.q_step0, a.w_cap = ptr(cache.q_step0), ptr(cache.w_cap)
    acts = actions.contiguous()
    g = grad_logp.contiguous().float()
    a.actions, a.grad_logp = acts.data_ptr(), g.data_ptr()
    if cache.env_name in ("cvrp", "cvrptw"):
        demand = meta["demand"].contiguous()
        vcap = meta["vehicle_capacity"].reshape(-1).contiguous()
        a.demand, a.vehicle_capacity = demand.data_ptr(), vcap.data_ptr()
        if cache.env_name == "cvrptw":
            locs, tw = meta["locs"].float().contiguous(), meta["time_windows"].float().contiguous()
            dur = meta["durations"].float().contiguous()
            a.locs, a.time_windows, a.durations = locs.data_ptr(), tw.data_ptr(), dur.data_ptr()
            a.w_time, a.d_w_time = ptr(cache.w_time), d_time.data_ptr()
    elif cache.env_name == "pctsp":
        demand = meta["real_prize"].float().contiguous()  # [B_inst, N], depot column 0
        vcap = meta["prize_required"].float().reshape(-1).contiguous()
        a.demand, a.vehicle_capacity = demand.data_ptr(), vcap.data_ptr()
    elif cache.env_name == "op":
        locs, maxlen = meta["locs"].float().contiguous(), meta["max_length"].float().contiguous()
        a.locs, a.max_length = locs.data_ptr(), maxlen.data_ptr()
    a.d_kvl, a.d_ctx_cur, a.d_ctx_first, a.d_q_bias = ptr(d_kvl), ptr(d_ctx_cur), ptr(d_ctx_first), ptr(d_q_bias)
    if d_planes is not None:
        # [3, ...]: the three plane gradients; [5, ...] (TSP) / [4, ...] (depot environments): the context-table gradients as
        # well, converted by the kernel into planes 3 / 4 (d_ctx_in_planes) — the fp32 tensors below are then its scratch
        nblk = 5 if tsp else 4
        assert d_planes.dtype == cache.kvl.dtype and d_planes.shape[1:] == (b_inst, n, EMBED_DIM) and d_planes.stride(3) == 1
        assert d_planes.shape[0] in (3, nblk)
        a.d_planes_bf16 = d_planes.data_ptr()
        a.d_planes_plane_stride, a.d_planes_batch_stride, a.d_planes_row_stride = d_planes.stride()[:3]
        a.d_ctx_in_planes = int(d_planes.shape[0] == nblk)
    if tsp:
        a.d_q_step0 = d_extra.data_ptr()
    else:
        a.d_w_cap = d_extra.data_ptr()
    a.logp_out = ptr(logp)
    a.err = err.data_ptr()
    ran = _lib.lib().rl4co_am_teacher_variant(C.byref(a))
    if backward_events is not None:  # bench.py: HIP events around the launch, on the stream it is issued on
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    st = _lib.lib().rl4co_am_teacher_backward(C.byref(a), torch.cuda.current_stream().cuda_stream)
    _lib.check(st, "rl4co_am_teacher_backward")
    if backward_events is not None:
        ev1.record()
        backward_events.append((ev0, ev1))
    return {"d_kvl": d_kvl if d_planes is None else d_planes, "d_ctx_first": d_ctx_first, "d_ctx_cur": d_ctx_cur, "d_q_bias": d_q_bias,
            "d_extra": d_extra, "d_time": d_time, "logp": logp, "err": err, "variant": {1: "replay", 2: "mma"}.get(ran, "invalid")}


def build_cache_autograd(env_name: str, h: Tensor, decoder, fused_planes: bool = False) -> dict[str, Tensor]:
    """The folded cache as differentiable tensors (same algebra as cache.build_folded_cache).

    ``fused_planes`` (bf16 encoder output on the GPU, bf16 planes, MMA backward): the five (TSP) / four per-node
    projections are ONE [B*N,128] x [128, 5*128] GEMM on the tall-skinny MFMA kernel whose output rows hold the planes
    side by side — the rollout streams them through strided views — and the backward of all of them is ONE input-
    gradient GEMM and ONE weight-gradient launch on the bf16 gradient matrix the teacher kernel writes in place
    (``TeacherForcedFoldLogLik``): no per-plane autograd nodes, no stack, no fp32 plane gradients, no gradient adds."""
    d = EMBED_DIM
    w_ctx = decoder.context_embedding.project_context.weight.float()
    blocks = fold_weights(env_name, decoder.project_node_embeddings.weight.float(),
                          decoder.pointer.project_out.weight.float(), w_ctx)
    out: dict = {}
    if fused_planes and h.is_cuda and h.dtype in (torch.bfloat16, torch.float16):
        from . import train_ops

        b, n, _ = h.shape
        w_all = torch.cat(blocks, 0)  # [nblk * 128, 128] fp32, differentiable through the fold
        h2 = h.detach().reshape(b * n, d).contiguous()
        w16 = w_all.detach().to(h.dtype).contiguous()
        planes = train_ops._gemm(h2, w16).view(b, n, len(blocks), d)  # row (b, n): [Kg | V | Kl' | ctx ...]
        out.update(fused=True, h=h, h2=h2, w_all=w_all, w16=w16, planes=planes)
        out["kvl"] = planes.permute(2, 0, 1, 3)[:3]                       # [3, B, N, 128] strided view, bf16
        # (r06) the context tables stay columns of the same matrix: the multistart rollout and the MMA teacher backward read
        # 16-bit rows at its row stride and widen them on load (two 68 us fp32 copies per step less)
        out["ctx_cur"] = planes[:, :, 4 if env_name == "tsp" else 3]
        if env_name == "tsp":
            out["ctx_first"] = planes[:, :, 3]
    else:
        if h.is_cuda and h.dtype in (torch.bfloat16, torch.float16):
            # 16-bit encoder output (autocast training): the fold GEMMs and their backward run on the
            # tall-skinny MFMA kernels (csrc/am_train_ops.hip) instead of five fp32 library GEMMs
            from . import train_ops

            planes = [train_ops.linear(h, w, None) for w in blocks]  # bf16: the rollout streams bf16 planes anyway
            planes = planes[:3] + [p.float() for p in planes[3:]]     # the context tables are read as fp32
        else:
            planes = [torch.matmul(h.float(), w.t()) for w in blocks]
        out.update(kvl=torch.stack(planes[:3], 0), ctx_cur=planes[4] if env_name == "tsp" else planes[-1])
        if env_name == "tsp":
            out["ctx_first"] = planes[3]
    if env_name == "tsp":
        out["q_step0"] = torch.mv(w_ctx, decoder.context_embedding.W_placeholder.float())
    elif w_ctx.shape[1] > d:  # PDP has no context scalar
        out["w_cap"] = w_ctx[:, d]
        if w_ctx.shape[1] > d + 1:  # CVRPTW: the current-time column
            out["w_time"] = w_ctx[:, d + 1]
    out["q_bias"] = (torch.matmul(h.mean(1, dtype=torch.float32), decoder.project_fixed_context.weight.float().t())
                     if decoder.use_graph_context else None)
    return out


def detached_cache(env_name: str, g: dict[str, Tensor], cache_dtype: torch.dtype) -> FoldedCache:
    """Rollout view of the autograd cache: detached, planes in the streaming dtype."""
    det = lambda x: None if x is None else x.detach().contiguous()  # noqa: E731
    view = det
    if g.get("fused"):
        assert cache_dtype == g["kvl"].dtype and cache_dtype in (torch.bfloat16, torch.float16)
        kvl = g["kvl"]  # strided view of the fused GEMM's output rows: the kernels take plane pointers and strides
        view = lambda x: None if x is None else x.detach()  # noqa: E731  (the context tables: column blocks of the same rows)
    else:
        kvl = g["kvl"].detach().to(cache_dtype).contiguous()
    return FoldedCache(env_name, kvl, view(g.get("ctx_first")), view(g["ctx_cur"]), det(g.get("q_bias")), det(g.get("q_step0")),
                       det(g.get("w_cap")), det(g.get("w_time")))


class TeacherForcedLogLik(torch.autograd.Function):
    """log p(a_t | s_t) [B,T]: forward = the rollout kernel's values, backward = HIP kernel."""

    @staticmethod
    def forward(ctx, kvl, ctx_first, ctx_cur, q_bias, q_extra, q_time, logps, cache: FoldedCache, actions: Tensor,
                meta: dict):
        ctx.cache, ctx.actions, ctx.meta = cache, actions, meta
        ctx.has = (ctx_first is not None, q_bias is not None, q_extra is not None, q_time is not None)
        return logps.detach().clone()

    @staticmethod
    def backward(ctx, grad_logp):
        out = run_backward(ctx.cache, ctx.actions, grad_logp, ctx.meta, variant=ctx.meta.get("teacher_variant", "auto"))
        # the kernel's sticky error word (infeasible / out-of-range replayed action, NaN): OR-ed into the caller's
        # deferred word, which the policy reads together with the NEXT rollout's status (no sync inside backward);
        # without a sink the word is checked here
        sink = ctx.meta.get("err_sink")
        if sink is not None:
            sink.bitwise_or_(out["err"])
        else:
            _lib.raise_for_error_bits(int(out["err"].item()))
        has_first, has_bias, has_extra, has_time = ctx.has
        return (out["d_kvl"], out["d_ctx_first"] if has_first else None, out["d_ctx_cur"],
                out["d_q_bias"] if has_bias else None, out["d_extra"] if has_extra else None,
                out["d_time"] if has_time else None, None, None, None, None)


class TeacherForcedFoldLogLik(torch.autograd.Function):
    """The same node with the cache fold inside it (``build_cache_autograd(fused_planes=True)``): backward = teacher
    kernel (plane gradients written as bf16 columns of one [B*N, nblk*128] matrix, context-table gradients converted
    into its remaining columns) -> ONE input-gradient GEMM (d h) and ONE weight-gradient launch (d W_all)."""

    @staticmethod
    def forward(ctx, h, w_all, q_bias, q_extra, q_time, logps, cache: FoldedCache, actions: Tensor, meta: dict, g: dict):
        ctx.cache, ctx.actions, ctx.meta = cache, actions, meta
        ctx.h2, ctx.w16, ctx.h_shape, ctx.nblk = g["h2"], g["w16"], h.shape, w_all.shape[0] // EMBED_DIM
        ctx.has = (q_bias is not None, q_extra is not None, q_time is not None)
        ctx.dtypes = (h.dtype, w_all.dtype)
        return logps.detach().clone()

    @staticmethod
    def backward(ctx, grad_logp):
        from . import train_ops

        b, n, d = ctx.h_shape
        dp = torch.empty((b, n, ctx.nblk, d), dtype=ctx.h2.dtype, device=grad_logp.device)  # 16-bit, the planes' type
        # all nblk column blocks come out of the kernel: the three plane gradients and (r06) the context-table gradients,
        # converted on the way out (two 53 us conversion copies per step less)
        out = run_backward(ctx.cache, ctx.actions, grad_logp, ctx.meta, variant="mma", d_planes=dp.permute(2, 0, 1, 3))
        sink = ctx.meta.get("err_sink")
        if sink is not None:
            sink.bitwise_or_(out["err"])
        else:
            _lib.raise_for_error_bits(int(out["err"].item()))
        dp2 = dp.view(b * n, ctx.nblk * d)
        dh = train_ops._gemm(dp2, ctx.w16.t().contiguous()).view(b, n, d)
        dw = train_ops._wgrad(dp2, ctx.h2)
        has_bias, has_extra, has_time = ctx.has
        return (dh.to(ctx.dtypes[0]), dw.to(ctx.dtypes[1]), out["d_q_bias"] if has_bias else None,
                out["d_extra"] if has_extra else None, out["d_time"] if has_time else None, None, None, None, None, None)


def teacher_forced_logps(env_name: str, g: dict[str, Tensor], cache: FoldedCache, actions: Tensor, logps: Tensor,
                         meta: dict) -> Tensor:
    """Differentiable per-step log-probs of ``actions`` (values = ``logps`` from the rollout)."""
    extra = g["q_step0"] if env_name == "tsp" else g.get("w_cap")  # None for PDP (no context scalar)
    if g.get("fused"):
        return TeacherForcedFoldLogLik.apply(g["h"], g["w_all"], g.get("q_bias"), extra, g.get("w_time"), logps, cache,
                                             actions, meta, g)
    return TeacherForcedLogLik.apply(g["kvl"], g.get("ctx_first"), g["ctx_cur"], g.get("q_bias"), extra, g.get("w_time"),
                                     logps, cache, actions, meta)
