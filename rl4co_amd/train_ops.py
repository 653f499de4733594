"""Training-time encoder pieces on HIP kernels (``csrc/am_train_ops.hip``).

``skip_instance_norm(x, s, weight, bias, eps)`` is ``Normalization("instance")(x + s)`` of the
reference encoder layer (``nn/ops.py:9-15,30-54``, ``nn/graph/attnnet.py:16-54``) as ONE kernel
forward and ONE backward over bf16 activations, instead of the dozen elementwise / reduction
launches autograd builds for the written-out norm. Used by ``policy._EncoderLayer`` in training
when the encoder runs under bf16 autocast on the GPU; anything else keeps the torch path.
"""
from __future__ import annotations

import torch
from torch import Tensor

from . import _lib

EMBED_DIM = 128


def max_nodes() -> int:
    return _lib.lib().rl4co_skip_inorm_max_nodes()


class _SkipInstanceNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor, s: Tensor, weight: Tensor, bias: Tensor, eps: float):
        b, n, d = x.shape
        xc, sc = x.contiguous(), s.contiguous()
        w32, b32 = weight.detach().float().contiguous(), bias.detach().float().contiguous()
        y = torch.empty_like(xc)
        out = torch.empty_like(xc)
        mean = torch.empty((b, d), dtype=torch.float32, device=x.device)
        rstd = torch.empty((b, d), dtype=torch.float32, device=x.device)
        st = _lib.lib().rl4co_skip_inorm_fwd_bf16(xc.data_ptr(), sc.data_ptr(), w32.data_ptr(), b32.data_ptr(), float(eps),
                                                  b, n, y.data_ptr(), out.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                                  torch.cuda.current_stream().cuda_stream)
        _lib.check(st, "rl4co_skip_inorm_fwd_bf16")
        ctx.save_for_backward(y, w32, mean, rstd)
        ctx.param_dtype = weight.dtype
        return out

    @staticmethod
    def backward(ctx, dout: Tensor):
        y, w32, mean, rstd = ctx.saved_tensors
        b, n, d = y.shape
        dc = dout.contiguous()
        if dc.dtype != torch.bfloat16:
            dc = dc.to(torch.bfloat16)
        dy = torch.empty_like(y)
        dgamma = torch.zeros(d, dtype=torch.float32, device=y.device)
        dbeta = torch.zeros(d, dtype=torch.float32, device=y.device)
        st = _lib.lib().rl4co_skip_inorm_bwd_bf16(dc.data_ptr(), y.data_ptr(), w32.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                                  b, n, dy.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
                                                  torch.cuda.current_stream().cuda_stream)
        _lib.check(st, "rl4co_skip_inorm_bwd_bf16")
        return dy, dy, dgamma.to(ctx.param_dtype), dbeta.to(ctx.param_dtype), None


def usable(x: Tensor, s: Tensor) -> bool:
    """bf16 [B,N,128] activations on the GPU with N inside the kernel's register budget."""
    return (x.is_cuda and x.dtype == torch.bfloat16 and s.dtype == torch.bfloat16 and x.dim() == 3
            and x.shape == s.shape and x.shape[-1] == EMBED_DIM and x.shape[1] <= max_nodes())


def skip_instance_norm(x: Tensor, s: Tensor, weight: Tensor, bias: Tensor, eps: float = 1e-5) -> Tensor:
    return _SkipInstanceNorm.apply(x, s, weight, bias, eps)
