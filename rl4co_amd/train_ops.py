"""Training-time encoder pieces on HIP kernels (``csrc/am_train_ops.hip``).

``skip_instance_norm(x, s, weight, bias, eps)`` is ``Normalization("instance")(x + s)`` of the
reference encoder layer (``nn/ops.py:9-15,30-54``, ``nn/graph/attnnet.py:16-54``) as ONE kernel
forward and ONE backward over bf16 activations, instead of the dozen elementwise / reduction
launches autograd builds for the written-out norm. Used by ``policy._EncoderLayer`` in training
when the encoder runs under bf16 / fp16 autocast on the GPU; anything else keeps the torch path.

Since r04 the FORWARD of a whole instance-norm stack (POMO) is one launch (``encoder_stack`` /
``_FusedEncoderStack``: ``rl4co_am_encoder_train_fwd`` keeps what the backward kernels read) and its backward sums the
weight-gradient / norm partials of all layers with one reduction per kind (``_GradArena``).
"""
from __future__ import annotations

import torch
from torch import Tensor

from . import _lib

EMBED_DIM = 128
HALF = (torch.bfloat16, torch.float16)  # element types the kernels are built for (csrc/elem16.h)


def _k(stem: str, dtype: torch.dtype):
    """The entry point ``stem`` bound to the element type of ``dtype`` (its first argument since r06: RL4CO_DT_BF16, or
    RL4CO_DT_F16 — the reference's default "16-mixed" precision is fp16 autocast, utils/trainer.py:57)."""
    if dtype not in HALF:
        raise TypeError(f"the training kernels take bfloat16 or float16 activations, got {dtype}")
    fn, did = getattr(_lib.lib(), stem), _lib.dtype_id(dtype)
    return lambda *args: fn(did, *args)


def max_nodes() -> int:
    return _lib.lib().rl4co_skip_inorm_max_nodes()


def _inorm_forward(xc: Tensor, sc: Tensor, w32: Tensor, b32: Tensor, eps: float):
    """(out, y = x + s, mean, rstd) of Normalization("instance")(x + s); bf16 [B,N,128] in / out, fp32 statistics."""
    b, n, d = xc.shape
    y = torch.empty_like(xc)
    out = torch.empty_like(xc)
    mean = torch.empty((b, d), dtype=torch.float32, device=xc.device)
    rstd = torch.empty((b, d), dtype=torch.float32, device=xc.device)
    st = _k("rl4co_skip_inorm_fwd", xc.dtype)(xc.data_ptr(), sc.data_ptr(), w32.data_ptr(), b32.data_ptr(), float(eps),
                                              b, n, y.data_ptr(), out.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                              torch.cuda.current_stream().cuda_stream)
    _lib.check(st, "rl4co_skip_inorm_fwd")
    return out, y, mean, rstd


def _inorm_backward(dout: Tensor, y: Tensor, w32: Tensor, mean: Tensor, rstd: Tensor, arena=None, key=None):
    """(d (x + s) bf16, d gamma fp32, d beta fp32); with an ``arena`` the two are a deferred handle (see _GradArena)."""
    b, n, d = y.shape
    dc = dout.contiguous()
    if dc.dtype != y.dtype:
        dc = dc.to(y.dtype)
    dy = torch.empty_like(y)
    part = (torch.empty((2, b, d), dtype=torch.float32, device=y.device) if arena is None
            else arena.norm_slot(key, b, d, y.device))  # per-instance d gamma | d beta
    st = _k("rl4co_skip_inorm_bwd", y.dtype)(dc.data_ptr(), y.data_ptr(), w32.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                              b, n, dy.data_ptr(), part[0].data_ptr(), part[1].data_ptr(),
                                              torch.cuda.current_stream().cuda_stream)
    _lib.check(st, "rl4co_skip_inorm_bwd")
    if arena is not None:
        return dy, key, None
    g = part.sum(1)  # one reduction over the instances for both, fixed order
    return dy, g[0], g[1]


class _SkipInstanceNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor, s: Tensor, weight: Tensor, bias: Tensor, eps: float):
        w32, b32 = weight.detach().float().contiguous(), bias.detach().float().contiguous()
        out, y, mean, rstd = _inorm_forward(x.contiguous(), s.contiguous(), w32, b32, eps)
        ctx.save_for_backward(y, w32, mean, rstd)
        ctx.param_dtype = weight.dtype
        return out

    @staticmethod
    def backward(ctx, dout: Tensor):
        dy, dgamma, dbeta = _inorm_backward(dout, *ctx.saved_tensors)
        return dy, dy, dgamma.to(ctx.param_dtype), dbeta.to(ctx.param_dtype), None


def _bnorm_forward(xc: Tensor, sc: Tensor, w32: Tensor, b32: Tensor, eps: float):
    """(out, y = x + s, mean, rstd, var) of BatchNorm1d(x + s) over all B x N rows with batch statistics."""
    m = xc.numel() // EMBED_DIM
    y, out = torch.empty_like(xc), torch.empty_like(xc)
    sums = torch.zeros((2, EMBED_DIM), dtype=torch.float32, device=xc.device)
    stream = torch.cuda.current_stream().cuda_stream
    _lib.check(_k("rl4co_skip_bnorm_stats", xc.dtype)(xc.data_ptr(), sc.data_ptr(), m, y.data_ptr(), sums.data_ptr(), stream),
               "rl4co_skip_bnorm_stats")
    mean = sums[0] / m
    var = (sums[1] / m - mean * mean).clamp_min_(0.0)  # biased, as F.batch_norm normalises with
    rstd = torch.rsqrt(var + eps)
    _lib.check(_k("rl4co_bnorm_apply", y.dtype)(y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), w32.data_ptr(), b32.data_ptr(),
                                                 m, out.data_ptr(), stream), "rl4co_bnorm_apply")
    return out, y, mean, rstd, var


def _bnorm_backward(dout: Tensor, y: Tensor, w32: Tensor, mean: Tensor, rstd: Tensor):
    """(d (x + s) bf16, d gamma fp32, d beta fp32)."""
    m = y.numel() // EMBED_DIM
    d = dout.to(y.dtype).contiguous()
    dy = torch.empty_like(y)
    sums = torch.zeros((2, EMBED_DIM), dtype=torch.float32, device=y.device)
    _lib.check(_k("rl4co_bnorm_bwd", y.dtype)(d.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), w32.data_ptr(), m,
                                               sums.data_ptr(), dy.data_ptr(), torch.cuda.current_stream().cuda_stream),
               "rl4co_bnorm_bwd")
    return dy, sums[1], sums[0]


class _SkipBatchNorm(torch.autograd.Function):
    """Normalization("batch")(x + s) in training: BatchNorm1d over all B x N rows with batch statistics."""

    @staticmethod
    def forward(ctx, x: Tensor, s: Tensor, weight: Tensor, bias: Tensor, eps: float):
        w32, b32 = weight.detach().float().contiguous(), bias.detach().float().contiguous()
        out, y, mean, rstd, var = _bnorm_forward(x.contiguous(), s.contiguous(), w32, b32, eps)
        ctx.save_for_backward(y, w32, mean, rstd)
        ctx.pdt = weight.dtype
        ctx.mark_non_differentiable(mean, var)
        return out, mean, var

    @staticmethod
    def backward(ctx, dout: Tensor, _dmean, _dvar):
        dy, dgamma, dbeta = _bnorm_backward(dout, *ctx.saved_tensors)
        return dy, dy, dgamma.to(ctx.pdt), dbeta.to(ctx.pdt), None


def _update_running_stats(bn: torch.nn.BatchNorm1d, mean: Tensor, var: Tensor, m: int) -> None:
    """nn.BatchNorm1d's bookkeeping in training mode (momentum, unbiased running variance, num_batches_tracked)."""
    if bn.track_running_stats and bn.running_mean is not None:
        with torch.no_grad():
            bn.num_batches_tracked += 1
            mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
            bn.running_mean.mul_(1 - mom).add_(mean.to(bn.running_mean.dtype), alpha=mom)
            bn.running_var.mul_(1 - mom).add_((var * (m / max(m - 1, 1))).to(bn.running_var.dtype), alpha=mom)


def skip_batch_norm(x: Tensor, s: Tensor, bn: torch.nn.BatchNorm1d) -> Tensor:
    """``bn((x + s).view(-1, 128)).view_as(x)`` in training mode, running statistics updated like
    nn.BatchNorm1d (momentum, unbiased running variance, num_batches_tracked)."""
    out, mean, var = _SkipBatchNorm.apply(x, s, bn.weight, bn.bias, bn.eps)
    _update_running_stats(bn, mean, var, x.numel() // EMBED_DIM)
    return out


def batch_norm_eval(y: Tensor, bn: torch.nn.BatchNorm1d) -> Tensor:
    """Eval-mode BatchNorm1d on bf16 rows [M,128]: one affine pass with the running statistics."""
    yc = y.contiguous()
    m = yc.numel() // EMBED_DIM
    out = torch.empty_like(yc)
    mean = bn.running_mean.float().contiguous()
    rstd = torch.rsqrt(bn.running_var.float() + bn.eps).contiguous()
    st = _k("rl4co_bnorm_apply", yc.dtype)(yc.data_ptr(), mean.data_ptr(), rstd.data_ptr(), bn.weight.detach().float().contiguous().data_ptr(),
                                           bn.bias.detach().float().contiguous().data_ptr(), m, out.data_ptr(),
                                           torch.cuda.current_stream().cuda_stream)
    _lib.check(st, "rl4co_bnorm_apply")
    return out


def skip_batch_norm_eval(x: Tensor, s: Tensor, bn: torch.nn.BatchNorm1d) -> Tensor:
    """Eval-mode ``bn(x + s)`` on bf16 rows [M,128] in one pass (the sum stays in fp32)."""
    xc, sc = x.contiguous(), s.contiguous()
    m = xc.numel() // EMBED_DIM
    out = torch.empty_like(xc)
    mean = bn.running_mean.float().contiguous()
    rstd = torch.rsqrt(bn.running_var.float() + bn.eps).contiguous()
    st = _k("rl4co_skip_bnorm_eval", xc.dtype)(xc.data_ptr(), sc.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                               bn.weight.detach().float().contiguous().data_ptr(),
                                               bn.bias.detach().float().contiguous().data_ptr(), m, out.data_ptr(),
                                               torch.cuda.current_stream().cuda_stream)
    _lib.check(st, "rl4co_skip_bnorm_eval")
    return out


def inorm_max_nodes() -> int:
    """Nodes the skip + instance-norm kernels serve (beyond ``max_nodes()`` the rows are re-read per pass, r06)."""
    return _lib.lib().rl4co_skip_inorm_wide_max_nodes()


def usable(x: Tensor, s: Tensor, kind: str = "layer") -> bool:
    """16-bit [B,N,128] activations on the GPU with N inside the norm kernels' limit (``kind``: "instance" or "layer")."""
    limit = inorm_max_nodes()  # (r06: both formulas re-read their rows per pass beyond ``max_nodes()``)
    return (x.is_cuda and x.dtype in HALF and s.dtype == x.dtype and x.dim() == 3
            and x.shape == s.shape and x.shape[-1] == EMBED_DIM and x.shape[1] <= limit)


def batch_usable(x: Tensor, s: Tensor) -> bool:
    return (x.is_cuda and x.dtype in HALF and s.dtype == x.dtype and x.shape == s.shape
            and x.shape[-1] == EMBED_DIM)


def skip_instance_norm(x: Tensor, s: Tensor, weight: Tensor, bias: Tensor, eps: float = 1e-5) -> Tensor:
    return _SkipInstanceNorm.apply(x, s, weight, bias, eps)


LAYER_NORM_EPS = 1e-5  # nn/ops.py:50: a literal in the reference's formula


def _lnorm_forward(xc: Tensor, sc: Tensor, eps: float = LAYER_NORM_EPS):
    """(out, y = x + s, stats [B,2] = (mean, 1 / sqrt(var + eps))) of the reference's "layer" normalisation: ONE mean and
    ONE unbiased variance over all N x 128 values of an instance, no affine (nn/ops.py:48-51)."""
    b, n, _ = xc.shape
    y, out = torch.empty_like(xc), torch.empty_like(xc)
    stats = torch.empty((b, 2), dtype=torch.float32, device=xc.device)
    st = _k("rl4co_skip_lnorm_fwd", xc.dtype)(xc.data_ptr(), sc.data_ptr(), float(eps), b, n, y.data_ptr(), out.data_ptr(),
                                              stats.data_ptr(), torch.cuda.current_stream().cuda_stream)
    _lib.check(st, "rl4co_skip_lnorm_fwd")
    return out, y, stats


def _lnorm_backward(dout: Tensor, y: Tensor, stats: Tensor) -> Tensor:
    b, n, _ = y.shape
    d = dout.to(y.dtype).contiguous()
    dy = torch.empty_like(y)
    st = _k("rl4co_skip_lnorm_bwd", y.dtype)(d.data_ptr(), y.data_ptr(), stats.data_ptr(), b, n, dy.data_ptr(),
                                             torch.cuda.current_stream().cuda_stream)
    _lib.check(st, "rl4co_skip_lnorm_bwd")
    return dy


class _SkipLayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor, s: Tensor):
        out, y, stats = _lnorm_forward(x.contiguous(), s.contiguous())
        ctx.save_for_backward(y, stats)
        return out

    @staticmethod
    def backward(ctx, dout: Tensor):
        dy = _lnorm_backward(dout, *ctx.saved_tensors)
        return dy, dy


def skip_layer_norm(x: Tensor, s: Tensor) -> Tensor:
    """``Normalization("layer")(x + s)`` on 16-bit [B, N <= 128, 128] activations, forward and backward on csrc/am_train_ops.hip."""
    return _SkipLayerNorm.apply(x, s)


# ---------------------------------------------------------------------------------------------------
# nn.Linear over the token rows on the tall-skinny MFMA kernel (csrc/am_train_ops.hip)
# ---------------------------------------------------------------------------------------------------
def _gemm(a2d: Tensor, w: Tensor, bias: Tensor | None = None, mask: Tensor | None = None, relu: bool = False,
          out: Tensor | None = None, residual: Tensor | None = None) -> Tensor:
    """out[M,N] = epilogue(a2d[M,K] @ w[N,K]^T + bias) (+ residual[M,N]); a2d / w / out / residual (/ mask) in ONE 16-bit
    type (bfloat16 or float16), fp32 bias."""
    m, k = a2d.shape
    n = w.shape[0]
    dt = a2d.dtype
    assert w.dtype == dt and (mask is None or mask.dtype == dt), (dt, w.dtype)
    if out is None:
        out = torch.empty((m, n), dtype=dt, device=a2d.device)
    else:
        assert out.shape == (m, n) and out.dtype == dt and out.is_contiguous()
    if residual is not None:
        assert residual.shape == (m, n) and residual.dtype == dt and residual.is_contiguous()
    st = _k("rl4co_linear", a2d.dtype)(a2d.data_ptr(), w.data_ptr(), None if bias is None else bias.data_ptr(),
                                      None if mask is None else mask.data_ptr(),
                                      None if residual is None else residual.data_ptr(), m, n, k, int(relu), out.data_ptr(),
                                      torch.cuda.current_stream().cuda_stream)
    _lib.check(st, "rl4co_linear")
    return out


# row chunks x output tiles per weight-gradient launch: every chunk writes an fp32 partial that one reduction sums.
# 512 workgroups (two per CU): measured on the four training shapes at 409 600 rows, kernel + reduction
# (tools/wgrad_chunks_bench.py; r02 with the 32-deep MFMA, tools/linear_bench.py: 256 -> 479 us, 384 -> 453, 512 -> 417,
# 768 -> 493, 1024 -> 499; before: 2048 -> 684 us, 1024 -> 592, 512 -> 484, 256 -> 631) — beyond two per CU the partials
# (67 MB written and re-read at 1024) cost more than the extra parallelism buys
_WGRAD_MAX_WORKGROUPS = int(__import__("os").environ.get("RL4CO_WGRAD_WORKGROUPS", "512"))


def _wgrad_chunks(m: int, n: int, k: int) -> int:
    tiles = (n // 128) * (k // 128)
    # (a single-tile layer, 128 x 128, is best at one workgroup per CU: 60 us against 67 with two — its partials are
    # as large as the operands)
    budget = _WGRAD_MAX_WORKGROUPS // 2 if tiles == 1 else _WGRAD_MAX_WORKGROUPS
    chunks = max(1, min(budget // tiles, (m + 255) // 256))
    if chunks >= 8:
        chunks -= chunks % 8  # a multiple of 8: the kernel then keeps the tiles of a chunk on one XCD (shared rows meet in its L2)
    return chunks


def _wgrad(d2: Tensor, x2: Tensor, with_bias: bool = False, arena: "_GradArena | None" = None, key=None):
    """dW[N,K] = d2[M,N]^T @ x2[M,K] and (``with_bias``) db[N] = column sums of d2, both fp32: split over
    the rows on the kernel, partials summed by torch. With an ``arena`` the partials go into its slot ``key`` and the
    result is a deferred handle: the arena sums the partials of ALL its slots of one shape in one launch (``finish``)."""
    m, n = d2.shape
    k = x2.shape[1]
    chunks = _wgrad_chunks(m, n, k)
    # one buffer per chunk: [N*K weight partials | N bias partials] -> ONE reduction over the chunk axis for both
    width = n * k + (n if with_bias else 0)
    partial = (torch.empty((chunks, width), dtype=torch.float32, device=d2.device) if arena is None
               else arena.slot(key, chunks, width, d2.device))
    st = _k("rl4co_wgrad", d2.dtype)(d2.data_ptr(), x2.data_ptr(), m, n, k, chunks, partial.data_ptr(),
                                     partial.data_ptr() + 4 * n * k if with_bias else None, width,
                                     torch.cuda.current_stream().cuda_stream)
    _lib.check(st, "rl4co_wgrad")
    if arena is not None:
        return (key, n, k, with_bias)
    g = partial.sum(0)
    dw = g[: n * k].view(n, k)
    return (dw, g[n * k :]) if with_bias else dw


class _GradArena:
    """Partial sums of one backward pass over a STACK of identical layers: every (kind, layer) gets a slot of the kind's
    buffer [L, chunks, width]; ``finish`` reduces each kind with ONE launch over its chunk axis instead of one per layer
    (38 reduction launches per POMO step before, 0.7 ms). Deterministic: fixed chunking, fixed summation order."""

    def __init__(self, n_layers: int):
        self.n_layers = n_layers
        self.buf: dict = {}
        self.sums: dict = {}

    def slot(self, key, chunks: int, width: int, device) -> Tensor:
        kind, layer = key
        if kind not in self.buf:
            self.buf[kind] = torch.empty((self.n_layers, chunks, width), dtype=torch.float32, device=device)
        buf = self.buf[kind]
        assert buf.shape[1:] == (chunks, width), (kind, buf.shape, chunks, width)
        return buf[layer]

    def norm_slot(self, key, b: int, d: int, device) -> Tensor:
        kind, layer = key
        if kind not in self.buf:
            self.buf[kind] = torch.empty((self.n_layers, 2, b, d), dtype=torch.float32, device=device)
        return self.buf[kind][layer]

    def finish(self) -> None:
        # weight-gradient kinds [L, chunks, width] -> [L, width]; norm kinds [L, 2, B, 128] -> [L, 2, 128]
        self.sums = {kind: buf.sum(1 if buf.dim() == 3 else 2) for kind, buf in self.buf.items()}

    def wgrad(self, handle):
        (kind, layer), n, k, with_bias = handle
        g = self.sums[kind][layer]
        dw = g[: n * k].view(n, k)
        return (dw, g[n * k:]) if with_bias else dw

    def norm(self, key):
        kind, layer = key
        g = self.sums[kind][layer]
        return g[0], g[1]


def linear_usable(x: Tensor, *weights: Tensor) -> bool:
    return (x.is_cuda and x.dtype in HALF and x.shape[-1] % 128 == 0
            and all(w.shape[0] % 128 == 0 and w.shape[1] % 128 == 0 for w in weights))


class _Linear(torch.autograd.Function):
    """y = x W^T + b (Wqkv, out_proj). Backward: dX on the same kernel with W^T; dW / db by torch."""

    @staticmethod
    def forward(ctx, x: Tensor, weight: Tensor, bias: Tensor | None):
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        w16 = weight.detach().to(x2.dtype).contiguous()
        out = _gemm(x2, w16, None if bias is None else bias.detach().float().contiguous())
        ctx.save_for_backward(x2, w16)
        ctx.pdt, ctx.has_bias = weight.dtype, bias is not None
        return out.view(*x.shape[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, dout: Tensor):
        x2, w16 = ctx.saved_tensors
        d = dout.reshape(-1, dout.shape[-1]).to(x2.dtype).contiguous()
        dx = _gemm(d, w16.t().contiguous())
        if ctx.has_bias:
            dw, db = _wgrad(d, x2, with_bias=True)
            return dx.view(*dout.shape[:-1], x2.shape[-1]), dw.to(ctx.pdt), db.to(ctx.pdt)
        return dx.view(*dout.shape[:-1], x2.shape[-1]), _wgrad(d, x2).to(ctx.pdt), None


class _MLP(torch.autograd.Function):
    """y = relu(x W1^T + b1) W2^T + b2 (nn/mlp.py:52-61); the ReLU and its backward mask ride in the
    GEMM epilogues, the hidden activation is written once and read twice."""

    @staticmethod
    def forward(ctx, x: Tensor, w1: Tensor, b1: Tensor, w2: Tensor, b2: Tensor):
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        w1_16, w2_16 = w1.detach().to(x2.dtype).contiguous(), w2.detach().to(x2.dtype).contiguous()
        h = _gemm(x2, w1_16, b1.detach().float().contiguous(), relu=True)
        y = _gemm(h, w2_16, b2.detach().float().contiguous())
        ctx.save_for_backward(x2, h, w1_16, w2_16)
        ctx.pdt = w1.dtype
        return y.view(*x.shape[:-1], w2.shape[0])

    @staticmethod
    def backward(ctx, dy: Tensor):
        x2, h, w1_16, w2_16 = ctx.saved_tensors
        d = dy.reshape(-1, dy.shape[-1]).to(x2.dtype).contiguous()
        dh = _gemm(d, w2_16.t().contiguous(), mask=h)  # (d W2) * [h > 0]
        dw2, db2 = _wgrad(d, h, with_bias=True)
        dx = _gemm(dh, w1_16.t().contiguous())
        dw1, db1 = _wgrad(dh, x2, with_bias=True)
        return (dx.view(*dy.shape[:-1], x2.shape[-1]), dw1.to(ctx.pdt), db1.to(ctx.pdt), dw2.to(ctx.pdt), db2.to(ctx.pdt))


def linear(x: Tensor, weight: Tensor, bias: Tensor | None) -> Tensor:
    return _Linear.apply(x, weight, bias)


def mlp(x: Tensor, w1: Tensor, b1: Tensor, w2: Tensor, b2: Tensor) -> Tensor:
    return _MLP.apply(x, w1, b1, w2, b2)


# ---------------------------------------------------------------------------------------------------
# encoder self-attention on the packed qkv rows (csrc/am_train_attn.hip)
# ---------------------------------------------------------------------------------------------------
def attn_max_nodes() -> int:
    """Nodes the training attention serves: forward and backward; beyond ``rl4co_attn_max_nodes()`` (one workgroup holds an
    instance's keys) on the key-streaming forward and the key-chunk backward (r06)."""
    return _lib.lib().rl4co_attn_wide_max_nodes()


def _attn_backward(qkv: Tensor, att: Tensor, datt: Tensor, lse: Tensor) -> Tensor:
    """d qkv [B,N,384] of the packed attention from d att (csrc/am_train_attn.hip); ``att`` is the forward's own output."""
    b, n, _ = qkv.shape
    dqkv = torch.empty_like(qkv)
    stream = torch.cuda.current_stream().cuda_stream
    if n <= _lib.lib().rl4co_attn_max_nodes():
        _lib.check(_k("rl4co_attn_bwd", qkv.dtype)(qkv.data_ptr(), att.data_ptr(), datt.data_ptr(), lse.data_ptr(), b, n,
                                                   dqkv.data_ptr(), stream), "rl4co_attn_bwd")
    else:  # key chunks of 128: their shares of d q meet in an fp32 workspace
        part = torch.empty(((n + 127) // 128, b, n, EMBED_DIM), dtype=torch.float32, device=qkv.device)
        _lib.check(_k("rl4co_attn_bwd_wide", qkv.dtype)(qkv.data_ptr(), att.data_ptr(), datt.data_ptr(), lse.data_ptr(), b, n,
                                                        dqkv.data_ptr(), part.data_ptr(), stream), "rl4co_attn_bwd_wide")
    return dqkv


class _Attention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv: Tensor):
        b, n, _ = qkv.shape
        q = qkv.contiguous()
        out = torch.empty((b, n, EMBED_DIM), dtype=q.dtype, device=qkv.device)
        lse = torch.empty((b, 8, n), dtype=torch.float32, device=qkv.device)
        st = _k("rl4co_attn_fwd", q.dtype)(q.data_ptr(), b, n, out.data_ptr(), lse.data_ptr(),
                                            torch.cuda.current_stream().cuda_stream)
        _lib.check(st, "rl4co_attn_fwd")
        ctx.save_for_backward(q, lse, out)
        return out

    @staticmethod
    def backward(ctx, dout: Tensor):
        q, lse, out = ctx.saved_tensors
        return _attn_backward(q, out, dout.to(q.dtype).contiguous(), lse)


class _GlimpseAttention(torch.autograd.Function):
    """The decoder's masked glimpse attention over all steps of given trajectories (csrc/am_cross_attn.hip)."""

    @staticmethod
    def _args(q, kv, bits, out, lse):
        a = _lib.CrossAttnArgs()
        a.B, a.T, a.B_inst, a.N = q.shape[0], q.shape[1], kv.shape[0], kv.shape[1]
        a.q, a.q_stride, a.kv, a.kv_stride = q.data_ptr(), q.stride(1), kv.data_ptr(), kv.stride(1)
        if bits is not None:
            a.mask, a.mask_words = bits.data_ptr(), bits.shape[-1]
        a.out, a.lse = out.data_ptr(), lse.data_ptr()
        return a

    @staticmethod
    def forward(ctx, q: Tensor, kv: Tensor, bits: Tensor | None):
        import ctypes

        q, kv = q.contiguous(), kv.contiguous()
        b, t, _ = q.shape
        out = torch.empty((b, t, EMBED_DIM), dtype=q.dtype, device=q.device)
        lse = torch.empty((b, 8, t), dtype=torch.float32, device=q.device)
        a = _GlimpseAttention._args(q, kv, bits, out, lse)
        st = _lib.lib().rl4co_cross_attn_fwd(_lib.dtype_id(q.dtype), ctypes.byref(a), torch.cuda.current_stream().cuda_stream)
        _lib.check(st, "rl4co_cross_attn_fwd")
        ctx.save_for_backward(q, kv, out, lse, *(() if bits is None else (bits,)))
        return out

    @staticmethod
    def backward(ctx, dout: Tensor):
        import ctypes

        q, kv, out, lse, *rest = ctx.saved_tensors
        bits = rest[0] if rest else None
        d = dout.to(q.dtype).contiguous()
        dq, dkv = torch.empty_like(q), torch.empty_like(kv)
        part = torch.empty((_lib.lib().rl4co_cross_attn_chunks(kv.shape[1]), q.shape[0], q.shape[1], EMBED_DIM),
                           dtype=torch.float32, device=q.device)
        a = _GlimpseAttention._args(q, kv, bits, out, lse)
        a.dout, a.dq, a.dkv, a.dq_partial = d.data_ptr(), dq.data_ptr(), dkv.data_ptr(), part.data_ptr()
        st = _lib.lib().rl4co_cross_attn_bwd(_lib.dtype_id(q.dtype), ctypes.byref(a), torch.cuda.current_stream().cuda_stream)
        _lib.check(st, "rl4co_cross_attn_bwd")
        return dq, dkv, None


def glimpse_attention_usable(q: Tensor, kv: Tensor, bits: Tensor | None) -> bool:
    return (q.is_cuda and q.dtype in HALF and kv.dtype == q.dtype and q.dim() == 3 and kv.dim() == 3
            and q.shape[-1] == EMBED_DIM and kv.shape[-1] == 2 * EMBED_DIM and q.shape[0] % kv.shape[0] == 0
            and (bits is None or (bits.dtype == torch.int32 and bits.is_contiguous() and bits.shape[:2] == q.shape[:2]
                                  and bits.shape[-1] % 4 == 0 and bits.shape[-1] * 32 >= kv.shape[1])))


def glimpse_attention(q: Tensor, kv: Tensor, bits: Tensor | None) -> Tensor:
    """``softmax_keys(q k^T / 4, masked) v`` per head (8 x 16) for q [B,T,128] step queries against kv [B_inst,N, k 128 | v 128]
    (trajectory b reads instance b % B_inst), ``bits`` [B,T,W] int32 feasibility bits (``kernels.env_replay(mask_bits=True)``)
    -> heads [B,T,128]; forward and backward on csrc/am_cross_attn.hip (nn/attention.py:255-296's inner attention)."""
    return _GlimpseAttention.apply(q, kv, bits)


class _LogitLogp(torch.autograd.Function):
    """log p of given actions from the pointer's raw logits, all steps at once (csrc/am_logit_logp.hip)."""

    @staticmethod
    def forward(ctx, raw: Tensor, bits: Tensor | None, actions: Tensor, tanh_clipping: float, temperature: float, err: Tensor | None):
        raw, acts = raw.contiguous(), actions.contiguous()
        b, t, n = raw.shape
        logp = torch.empty((b, t), dtype=torch.float32, device=raw.device)
        lse = torch.empty((b, t), dtype=torch.float32, device=raw.device)
        st = _lib.lib().rl4co_logit_logp_fwd(raw.data_ptr(), None if bits is None else bits.data_ptr(), 0 if bits is None else bits.shape[-1],
                                             acts.data_ptr(), b * t, n, float(tanh_clipping), float(temperature), logp.data_ptr(),
                                             lse.data_ptr(), None if err is None else err.data_ptr(),
                                             torch.cuda.current_stream().cuda_stream)
        _lib.check(st, "rl4co_logit_logp_fwd")
        ctx.save_for_backward(raw, acts, lse, *(() if bits is None else (bits,)))
        ctx.clip, ctx.temp = float(tanh_clipping), float(temperature)
        return logp

    @staticmethod
    def backward(ctx, g: Tensor):
        raw, acts, lse, *rest = ctx.saved_tensors
        bits = rest[0] if rest else None
        b, t, n = raw.shape
        gg = g.float().contiguous()
        d_raw = torch.empty_like(raw)
        st = _lib.lib().rl4co_logit_logp_bwd(raw.data_ptr(), None if bits is None else bits.data_ptr(), 0 if bits is None else bits.shape[-1],
                                             acts.data_ptr(), lse.data_ptr(), gg.data_ptr(), b * t, n, ctx.clip, ctx.temp,
                                             d_raw.data_ptr(), torch.cuda.current_stream().cuda_stream)
        _lib.check(st, "rl4co_logit_logp_bwd")
        return d_raw, None, None, None, None, None


def logit_logp(raw: Tensor, bits: Tensor | None, actions: Tensor, tanh_clipping: float, temperature: float,
               err: Tensor | None = None) -> Tensor:
    """``log_softmax(mask(tanh_clipping * tanh(raw / sqrt(128))) / temperature)[actions]`` for raw logits [B,T,N] fp32 (the
    pointer's glimpse . logit_key), ``bits`` [B,T,W] int32 feasibility bits (None: no masking of the logits), ``actions``
    [B,T] -> [B,T]; one pass each way (nn/attention.py:291-293, utils/decoding.py:169-188)."""
    assert raw.is_cuda and raw.dtype == torch.float32 and raw.dim() == 3 and actions.shape == raw.shape[:2] and actions.dtype == torch.int64
    assert bits is None or (bits.dtype == torch.int32 and bits.is_contiguous() and bits.shape[:2] == raw.shape[:2]
                            and bits.shape[-1] * 32 >= raw.shape[-1])
    return _LogitLogp.apply(raw, bits, actions, tanh_clipping, temperature, err)


def attention_usable(qkv: Tensor) -> bool:
    return (qkv.is_cuda and qkv.dtype in HALF and qkv.dim() == 3 and qkv.shape[-1] == 3 * EMBED_DIM
            and qkv.shape[1] <= attn_max_nodes())


def attention(qkv: Tensor) -> Tensor:
    """softmax(q k^T / 4) v per head on qkv [B,N,384] (q | k | v, 8 heads x 16) -> [B,N,128]."""
    return _Attention.apply(qkv)


def attention_flash(qkv: Tensor) -> Tensor:
    """Inference attention for any N (csrc/am_attn_flash.hip): softmax(q k^T / 4) v per head on the packed
    qkv [B,N,384] bf16 -> [B,N,128] bf16; keys / values stream through LDS, no N x N matrix, no autograd."""
    assert qkv.is_cuda and qkv.dtype in HALF and qkv.dim() == 3 and qkv.shape[-1] == 3 * EMBED_DIM
    qkv = qkv.contiguous()
    b, n, _ = qkv.shape
    out = torch.empty((b, n, EMBED_DIM), dtype=qkv.dtype, device=qkv.device)
    st = _k("rl4co_attn_flash", qkv.dtype)(qkv.data_ptr(), b, n, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    _lib.check(st, "rl4co_attn_flash")
    return out


# ---------------------------------------------------------------------------------------------------
# a whole residual sub-block of the encoder layer as ONE autograd node (nn/graph/attnnet.py:16-54):
#   Normalization(x + MHA(x))   and   Normalization(x + MLP(x))
# Same kernels as the pieces above; what the single node buys is the backward of the skip connection: the gradient
# of x is (gradient through the branch) + (gradient of the sum), and the second term rides in the epilogue of the
# branch's last input-gradient GEMM (``residual``) instead of being added by autograd in one more pass over both
# (12 passes of 315 MB per POMO step at 4096 x 100 nodes).
# ---------------------------------------------------------------------------------------------------
def _norm_forward(kind: str, xc: Tensor, sc: Tensor, w32: Tensor, b32: Tensor, eps: float):
    if kind == "layer":  # no affine; the statistics travel in the `mean` slot as [B, 2] = (mean, rstd)
        out, y, stats = _lnorm_forward(xc, sc, eps)
        return out, y, stats, None, None
    if kind == "instance":
        out, y, mean, rstd = _inorm_forward(xc, sc, w32, b32, eps)
        return out, y, mean, rstd, None
    return _bnorm_forward(xc, sc, w32, b32, eps)


def _norm_backward(kind: str, dout: Tensor, y: Tensor, w32: Tensor, mean: Tensor, rstd: Tensor):
    if kind == "layer":
        return _lnorm_backward(dout, y, mean), None, None
    return (_inorm_backward if kind == "instance" else _bnorm_backward)(dout, y, w32, mean, rstd)


def _h16(w: Tensor, dtype: torch.dtype) -> Tensor:
    return w.detach().to(dtype).contiguous()


def _f32(w: Tensor | None) -> Tensor | None:
    return None if w is None else w.detach().float().contiguous()


def _attention_block_bwd(kind, dout, x2, wqkv16, wo16, qkv, lse, att, y, g32, mean, rstd, arena=None, layer=0, wt=None):
    """Backward of Normalization(x + MHA(x)) from the forward's saved tensors: (dx [B,N,128], dWqkv + dbqkv, dWo + dbo,
    dgamma + dbeta) — gradients in fp32, dx in the activations' type. ``arena``: the reductions are deferred (the three
    results are then handles for _GradArena); ``wt``: (Wqkv^T, Wo^T) already transposed (a stack transposes all layers once)."""
    b, n, d = y.shape
    if arena is not None:  # (instance norm: the stack's only kind)
        dy, hnorm, _ = _inorm_backward(dout, y, g32, mean, rstd, arena, ("norm1", layer))
    else:
        dy, dgamma, dbeta = _norm_backward(kind, dout, y, g32, mean, rstd)  # d (x + s): the branch AND the skip
        hnorm = (dgamma, dbeta)
    d2 = dy.view(-1, d)
    datt = _gemm(d2, wo16.t().contiguous() if wt is None else wt[1])
    hwo = _wgrad(d2, att.reshape(-1, d), with_bias=True, arena=arena, key=("wo", layer))
    dqkv = _attn_backward(qkv, att, datt.view(b, n, d), lse)
    dq2 = dqkv.view(-1, 3 * d)
    dx = _gemm(dq2, wqkv16.t().contiguous() if wt is None else wt[0], residual=d2)
    hwqkv = _wgrad(dq2, x2.reshape(-1, d), with_bias=True, arena=arena, key=("wqkv", layer))
    if arena is not None:
        return dx.view(b, n, d), hwqkv, hwo, hnorm
    return dx.view(b, n, d), hwqkv[0], hwqkv[1], hwo[0], hwo[1], hnorm[0], hnorm[1]


def mlp_input_grad(d2: Tensor, h2: Tensor, w1t_packed: Tensor, w2t_packed: Tensor) -> tuple[Tensor, Tensor]:
    """(dh [M,512], dx [M,128]) = ((dy W2) * [h > 0], dh W1 + dy) in ONE launch (``rl4co_mlp_input_grad``): the 512-wide
    gradient goes through LDS instead of being written by one GEMM launch and read back by the next. ``w1t_packed`` /
    ``w2t_packed``: W1^T [128,512] / W2^T [512,128] in the encoder's fragment order (``_pack_stack``)."""
    m = d2.shape[0]
    dh = torch.empty((m, 4 * EMBED_DIM), dtype=d2.dtype, device=d2.device)
    dx = torch.empty((m, EMBED_DIM), dtype=d2.dtype, device=d2.device)
    st = _lib.lib().rl4co_mlp_input_grad(d2.data_ptr(), h2.data_ptr(), m, w2t_packed.data_ptr(), w1t_packed.data_ptr(),
                                         _lib.dtype_id(d2.dtype), dh.data_ptr(), dx.data_ptr(), torch.cuda.current_stream().cuda_stream)
    _lib.check(st, "rl4co_mlp_input_grad")
    return dh, dx


def _packed_transposes(w1_16: Tensor, w2_16: Tensor) -> tuple[Tensor, Tensor]:
    """(W1^T, W2^T) in the fused kernels' fragment order, packed per call (two 128 x 512 transposes and two packing launches).
    A cache keyed on the 16-bit copies' identity (r04 / r05) never hit: under autocast those copies are fresh tensors every
    forward, and the cache kept them alive — so their addresses were never reused — pinning ~33 MB of dead weights."""
    return _pack_stack(w1_16.t().contiguous()[None])[0], _pack_stack(w2_16.t().contiguous()[None])[0]


def _mlp_block_bwd(kind, dout, x2, h, w1_16, w2_16, y, g32, mean, rstd, arena=None, layer=0, wt=None, wp=None):
    """Backward of Normalization(x + MLP(x)): (dx, dW1 + db1, dW2 + db2, dgamma + dbeta); ``arena`` / ``wt`` (W1^T, W2^T) as above;
    ``wp``: (W1^T, W2^T) in fragment order — both input-gradient GEMMs then run as one launch (``mlp_input_grad``)."""
    b, n, d = y.shape
    if arena is not None:
        dy, hnorm, _ = _inorm_backward(dout, y, g32, mean, rstd, arena, ("norm2", layer))
    else:
        dy, dgamma, dbeta = _norm_backward(kind, dout, y, g32, mean, rstd)
        hnorm = (dgamma, dbeta)
    d2 = dy.view(-1, d)
    h2 = h.reshape(-1, h.shape[-1])
    if (wp is None and FUSED_MLP_INPUT_GRAD and tuple(w1_16.shape) == (4 * EMBED_DIM, EMBED_DIM)
            and tuple(w2_16.shape) == (EMBED_DIM, 4 * EMBED_DIM)):
        # a single sub-block node (batch / layer norm training, the piecewise path): this layer's transposes in fragment order
        wp = _packed_transposes(w1_16, w2_16)
    if wp is not None and h2.shape[-1] == 4 * EMBED_DIM and d2.is_contiguous() and h2.is_contiguous():
        dh, dx = mlp_input_grad(d2, h2, wp[0], wp[1])
        hw2 = _wgrad(d2, h2, with_bias=True, arena=arena, key=("w2", layer))
    else:
        dh = _gemm(d2, w2_16.t().contiguous() if wt is None else wt[1], mask=h2)  # (d W2) * [h > 0]
        hw2 = _wgrad(d2, h2, with_bias=True, arena=arena, key=("w2", layer))
        dx = _gemm(dh, w1_16.t().contiguous() if wt is None else wt[0], residual=d2)
    hw1 = _wgrad(dh, x2.reshape(-1, d), with_bias=True, arena=arena, key=("w1", layer))
    if arena is not None:
        return dx.view(b, n, d), hw1, hw2, hnorm
    return dx.view(b, n, d), hw1[0], hw1[1], hw2[0], hw2[1], hnorm[0], hnorm[1]


class _AttentionBlock(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, wqkv, bqkv, wo, bo, gamma, beta, eps, kind):
        b, n, d = x.shape
        xc = x.contiguous()
        x2 = xc.view(-1, d)
        wqkv16, wo16, g32 = _h16(wqkv, x.dtype), _h16(wo, x.dtype), _f32(gamma)
        qkv = _gemm(x2, wqkv16, _f32(bqkv)).view(b, n, 3 * d)
        att = torch.empty((b, n, d), dtype=x.dtype, device=x.device)
        lse = torch.empty((b, 8, n), dtype=torch.float32, device=x.device)
        stream = torch.cuda.current_stream().cuda_stream
        _lib.check(_k("rl4co_attn_fwd", qkv.dtype)(qkv.data_ptr(), b, n, att.data_ptr(), lse.data_ptr(), stream), "rl4co_attn_fwd")
        s = _gemm(att.view(-1, d), wo16, _f32(bo)).view(b, n, d)
        out, y, mean, rstd, var = _norm_forward(kind, xc, s, g32, _f32(beta), eps)
        ctx.save_for_backward(x2, wqkv16, wo16, qkv, lse, att, y, g32, mean, rstd)
        ctx.kind, ctx.pdt = kind, wqkv.dtype
        if kind == "batch":
            ctx.mark_non_differentiable(mean, var)
            return out, mean, var
        return out

    @staticmethod
    def backward(ctx, dout, *_unused):
        dx, *grads = _attention_block_bwd(ctx.kind, dout, *ctx.saved_tensors)
        return (dx, *(None if g is None else g.to(ctx.pdt) for g in grads), None, None)


class _MLPBlock(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, gamma, beta, eps, kind):
        b, n, d = x.shape
        xc = x.contiguous()
        x2 = xc.view(-1, d)
        w1_16, w2_16, g32 = _h16(w1, x.dtype), _h16(w2, x.dtype), _f32(gamma)
        h = _gemm(x2, w1_16, _f32(b1), relu=True)
        s = _gemm(h, w2_16, _f32(b2)).view(b, n, d)
        out, y, mean, rstd, var = _norm_forward(kind, xc, s, g32, _f32(beta), eps)
        ctx.save_for_backward(x2, h, w1_16, w2_16, y, g32, mean, rstd)
        ctx.kind, ctx.pdt = kind, w1.dtype
        if kind == "batch":
            ctx.mark_non_differentiable(mean, var)
            return out, mean, var
        return out

    @staticmethod
    def backward(ctx, dout, *_unused):
        dx, *grads = _mlp_block_bwd(ctx.kind, dout, *ctx.saved_tensors)
        return (dx, *(None if g is None else g.to(ctx.pdt) for g in grads), None, None)


# ---------------------------------------------------------------------------------------------------
# the whole encoder stack's FORWARD as one launch (instance norm: POMO, zoo/pomo/model.py:59-63), the per-op backward
# kernels fed from what it saved (csrc/am_encoder.hip: am_encoder_kernel<.., TRAIN>, rl4co_am_encoder_train_fwd)
# ---------------------------------------------------------------------------------------------------
FUSED_MLP_INPUT_GRAD = True  # (False: the two GEMM launches — tools / tests compare the two)
_STACK_PARAMS = 12  # per layer: Wqkv, bqkv, Wo, bo, gamma1, beta1, W1, b1, W2, b2, gamma2, beta2


def _pack_stack(w: Tensor) -> Tensor:
    """[L, out, in] 16-bit -> the fused kernel's fragment order [L, out/32, in/16, 64, 8] (encoder.pack_weight, all layers at once)."""
    l, out_f, in_f = w.shape
    return w.view(l, out_f // 32, 32, in_f // 16, 2, 8).permute(0, 1, 3, 4, 2, 5).contiguous()


class _FusedEncoderStack(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x0: Tensor, n_layers: int, eps: float, *params: Tensor):
        import ctypes as C

        from .encoder import AmEncoderArgs

        assert len(params) == _STACK_PARAMS * n_layers
        b, n, d = x0.shape
        dt, dev = x0.dtype, x0.device
        x0c = x0.contiguous()
        per = [params[i::_STACK_PARAMS] for i in range(_STACK_PARAMS)]  # per kind: one tensor per layer

        def stack16(ts):  # [L, out, in] in the activations' type: ONE multi-tensor copy casts and stacks the layers' weights
            buf = torch.empty((n_layers, *ts[0].shape), dtype=dt, device=dev)
            torch._foreach_copy_(list(buf.unbind(0)), [t.detach() for t in ts])
            return buf

        wqkv16, wo16, w1_16, w2_16 = stack16(per[0]), stack16(per[2]), stack16(per[6]), stack16(per[8])
        # the six small fp32 kinds (biases, gammas, betas) side by side in one buffer, gathered by one multi-tensor copy
        small = [per[1], per[7], per[4], per[5], per[10], per[11]]
        widths = [ts[0].numel() for ts in small]
        flat = torch.empty(n_layers * sum(widths), dtype=torch.float32, device=dev)
        views, off = [], 0
        for wdt in widths:  # kind-major: every kind's [L, width] block is contiguous
            views.append(flat[off:off + n_layers * wdt].view(n_layers, wdt))
            off += n_layers * wdt
        torch._foreach_copy_([v[l] for v, ts in zip(views, small) for l in range(n_layers)],
                             [t.detach() for ts in small for t in ts])
        bqkv, b1, g1, be1, g2, be2 = views
        packed = [_pack_stack(w) for w in (wqkv16, wo16, w1_16, w2_16)]
        new = lambda *shape, dtype=dt: torch.empty(shape, dtype=dtype, device=dev)  # noqa: E731
        out, qkv, att = new(n_layers, b, n, d), new(n_layers, b, n, 3 * d), new(n_layers, b, n, d)
        y1, x1, h, y2 = new(n_layers, b, n, d), new(n_layers, b, n, d), new(n_layers, b, n, 4 * d), new(n_layers, b, n, d)
        lse = new(n_layers, b, 8, n, dtype=torch.float32)
        stats = new(n_layers, 4, b, d, dtype=torch.float32)
        a = AmEncoderArgs()
        a.B, a.N, a.num_layers, a.norm = b, n, n_layers, 1
        a.act_dtype = a.cache_dtype = _lib.dtype_id(dt)
        a.wqkv_packed, a.wo_packed, a.w1_packed, a.w2_packed = (p.data_ptr() for p in packed)
        a.bqkv, a.b1 = bqkv.data_ptr(), b1.data_ptr()
        a.n1_scale, a.n1_shift, a.n2_scale, a.n2_shift = g1.data_ptr(), be1.data_ptr(), g2.data_ptr(), be2.data_ptr()
        sv = _lib.AmTrainSave()
        sv.x0, sv.out, sv.qkv, sv.att, sv.y1, sv.x1, sv.h, sv.y2 = (t.data_ptr() for t in (x0c, out, qkv, att, y1, x1, h, y2))
        sv.lse, sv.stats = lse.data_ptr(), stats.data_ptr()
        assert abs(float(eps) - 1e-5) < 1e-12, "the fused training forward has InstanceNorm1d's default epsilon compiled in"
        st = _lib.lib().rl4co_am_encoder_train_fwd(C.byref(a), C.byref(sv), torch.cuda.current_stream().cuda_stream)
        _lib.check(st, "rl4co_am_encoder_train_fwd")
        ctx.save_for_backward(x0c, out, qkv, att, y1, x1, h, y2, lse, stats, wqkv16, wo16, w1_16, w2_16, g1, g2)
        ctx.n_layers, ctx.pdt = n_layers, params[0].dtype
        return out[n_layers - 1]

    @staticmethod
    def backward(ctx, dout: Tensor):
        x0c, out, qkv, att, y1, x1, h, y2, lse, stats, wqkv16, wo16, w1_16, w2_16, g1, g2 = ctx.saved_tensors
        t = ctx.pdt
        nl = ctx.n_layers
        arena = _GradArena(nl)
        # the four weight stacks transposed once for all layers (the input-gradient GEMMs read W^T rows)
        wqkv_t, wo_t, w1_t, w2_t = (w.transpose(1, 2).contiguous() for w in (wqkv16, wo16, w1_16, w2_16))
        # ... and the MLP's two in fragment order: its input gradient is one launch per layer (mlp_input_grad)
        w1t_p, w2t_p = (_pack_stack(w1_t), _pack_stack(w2_t)) if FUSED_MLP_INPUT_GRAD else (None, None)
        handles = []
        d = dout
        for l in reversed(range(nl)):
            d, hw1, hw2, hn2 = _mlp_block_bwd("instance", d, x1[l], h[l], w1_16[l], w2_16[l], y2[l], g2[l], stats[l, 2], stats[l, 3],
                                              arena=arena, layer=l, wt=(w1_t[l], w2_t[l]),
                                              wp=None if w1t_p is None else (w1t_p[l], w2t_p[l]))
            x_in = x0c if l == 0 else out[l - 1]
            d, hwqkv, hwo, hn1 = _attention_block_bwd("instance", d, x_in, wqkv16[l], wo16[l], qkv[l], lse[l], att[l], y1[l], g1[l],
                                                      stats[l, 0], stats[l, 1], arena=arena, layer=l, wt=(wqkv_t[l], wo_t[l]))
            handles.append((l, hwqkv, hwo, hn1, hw1, hw2, hn2))
        arena.finish()  # ONE reduction launch per kind for all layers
        grads: list = [None] * (_STACK_PARAMS * nl)
        for l, hwqkv, hwo, hn1, hw1, hw2, hn2 in handles:
            dwqkv, dbqkv = arena.wgrad(hwqkv)
            dwo, dbo = arena.wgrad(hwo)
            dw1, db1 = arena.wgrad(hw1)
            dw2, db2 = arena.wgrad(hw2)
            dg1, dbe1 = arena.norm(hn1)
            dg2, dbe2 = arena.norm(hn2)
            grads[_STACK_PARAMS * l:_STACK_PARAMS * (l + 1)] = [g.to(t) for g in (dwqkv, dbqkv, dwo, dbo, dg1, dbe1, dw1, db1,
                                                                                dw2, db2, dg2, dbe2)]
        return (d, None, None, *grads)


def stack_usable(x: Tensor, layers) -> bool:
    """Every layer an (8-head attention, instance norm, 128 -> 512 -> 128 MLP, instance norm) block with biases, 16-bit
    [B, N <= 128, 128] activations on the GPU: the fused training forward serves the whole stack."""
    if not (x.is_cuda and x.dtype in HALF and x.dim() == 3 and x.shape[-1] == EMBED_DIM and 2 <= x.shape[1] <= 128):
        return False
    if x.shape[1] > min(max_nodes(), _lib.lib().rl4co_attn_max_nodes()):
        return False
    for layer in layers:
        attn, n1, ffn, n2 = layer[0].module, layer[1], layer[2].module, layer[3]
        if n1.kind != "instance" or n2.kind != "instance" or attn.num_heads != 8 or len(ffn.lins) != 2:
            return False
        if tuple(ffn.lins[0].weight.shape) != (4 * EMBED_DIM, EMBED_DIM) or tuple(ffn.lins[1].weight.shape) != (EMBED_DIM, 4 * EMBED_DIM):
            return False
        if any(lin.bias is None for lin in (attn.Wqkv, attn.out_proj, *ffn.lins)):
            return False
        if any(abs(nm.normalizer.eps - 1e-5) > 1e-12 or nm.normalizer.weight is None for nm in (n1, n2)):
            return False
    return True


def encoder_stack(x: Tensor, layers) -> Tensor:
    """``layers(x)`` for a stack of instance-norm encoder layers: ONE forward launch, the per-op backward kernels."""
    params = []
    for layer in layers:
        attn, n1, ffn, n2 = layer[0].module, layer[1].normalizer, layer[2].module, layer[3].normalizer
        params += [attn.Wqkv.weight, attn.Wqkv.bias, attn.out_proj.weight, attn.out_proj.bias, n1.weight, n1.bias,
                   ffn.lins[0].weight, ffn.lins[0].bias, ffn.lins[1].weight, ffn.lins[1].bias, n2.weight, n2.bias]
    return _FusedEncoderStack.apply(x, len(layers), float(layers[0][1].normalizer.eps), *params)


def _block_norm_args(norm_module):
    if norm_module.kind == "layer":  # no parameters (nn/ops.py:48-51)
        return None, None, LAYER_NORM_EPS
    nz = norm_module.normalizer
    return nz.weight, nz.bias, nz.eps


def block_usable(x: Tensor, kind: str, *weights: Tensor) -> bool:
    """bf16 [B,N,128] rows whose every kernel (GEMMs, attention, skip + norm) is served: one autograd node per sub-block."""
    if kind not in ("instance", "batch", "layer") or not linear_usable(x, *weights) or x.dim() != 3 or x.shape[-1] != EMBED_DIM:
        return False
    if x.shape[1] > attn_max_nodes():
        return False
    return kind == "batch" or x.shape[1] <= inorm_max_nodes()


def attention_block(x: Tensor, attn, norm) -> Tensor:
    """``norm(x + attn(x))`` (SkipConnection(MultiHeadAttention) + Normalization) as one autograd node."""
    gamma, beta, eps = _block_norm_args(norm)
    res = _AttentionBlock.apply(x, attn.Wqkv.weight, attn.Wqkv.bias, attn.out_proj.weight, attn.out_proj.bias, gamma, beta,
                                eps, norm.kind)
    if norm.kind == "batch":
        out, mean, var = res
        _update_running_stats(norm.normalizer, mean, var, x.numel() // EMBED_DIM)
        return out
    return res


def mlp_block(x: Tensor, ffn, norm) -> Tensor:
    """``norm(x + ffn(x))`` (SkipConnection(MLP 128 -> 512 -> 128) + Normalization) as one autograd node."""
    gamma, beta, eps = _block_norm_args(norm)
    l1, l2 = ffn.lins
    res = _MLPBlock.apply(x, l1.weight, l1.bias, l2.weight, l2.bias, gamma, beta, eps, norm.kind)
    if norm.kind == "batch":
        out, mean, var = res
        _update_running_stats(norm.normalizer, mean, var, x.numel() // EMBED_DIM)
        return out
    return res


# ---------------------------------------------------------------------------------------------------
# init embedding (K = 2 / 3 "GEMM") in training
# ---------------------------------------------------------------------------------------------------
class _InitEmbed(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats: Tensor, weight: Tensor, bias: Tensor, dtype: torch.dtype = torch.bfloat16):
        f2 = feats.reshape(-1, feats.shape[-1]).float().contiguous()
        out = torch.empty((f2.shape[0], EMBED_DIM), dtype=dtype, device=feats.device)
        st = _k("rl4co_init_embed", dtype)(f2.data_ptr(), weight.detach().float().contiguous().data_ptr(),
                                              bias.detach().float().contiguous().data_ptr(), f2.shape[0], f2.shape[1],
                                              out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        _lib.check(st, "rl4co_init_embed")
        ctx.save_for_backward(f2)
        ctx.pdt, ctx.adt = weight.dtype, dtype
        return out.view(*feats.shape[:-1], EMBED_DIM)

    @staticmethod
    def backward(ctx, dout: Tensor):
        (f2,) = ctx.saved_tensors
        d = dout.reshape(-1, EMBED_DIM)
        if d.dtype != ctx.adt:  # not under autocast: the library path
            return None, torch.matmul(d.t().float(), f2).to(ctx.pdt), d.sum(0, dtype=torch.float32).to(ctx.pdt), None
        # a [128, M] x [M, F] product with F <= 6 is a reduction, not a GEMM (the library needs 1 ms and an fp32 copy of
        # dout for it at M = 409 600): per-block partial sums of dW and db in one pass over dout, summed in a fixed order
        d = d.contiguous()
        m, f = f2.shape
        import ctypes as C

        nblk = C.c_int(0)
        wgrad = _k("rl4co_init_embed_wgrad", ctx.adt)
        _lib.check(wgrad(None, None, m, f, None, C.byref(nblk), None), "rl4co_init_embed_wgrad")
        partial = torch.empty((nblk.value, EMBED_DIM, f + 1), dtype=torch.float32, device=d.device)
        st = wgrad(d.data_ptr(), f2.data_ptr(), m, f, partial.data_ptr(), None,
                                             torch.cuda.current_stream().cuda_stream)
        _lib.check(st, "rl4co_init_embed_wgrad")
        g = partial.sum(0)
        return None, g[:, :f].to(ctx.pdt), g[:, f].to(ctx.pdt), None


def init_embed(feats: Tensor, lin: torch.nn.Linear, dtype: torch.dtype | None = None) -> Tensor:
    """``lin(feats)`` for the 2- ... 6-feature init embeddings with a 16-bit output: ``dtype``, else the ambient CUDA
    autocast type when that is bfloat16 / float16, else bfloat16."""
    if dtype is None:
        amb = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled() else None
        dtype = amb if amb in HALF else torch.bfloat16
    return _InitEmbed.apply(feats, lin.weight, lin.bias, dtype)
