"""Host-side mirror of ``AttentionModelPolicy`` whose rollout runs in the fused HIP kernel.

Drop-in at the reference's ``policy`` seam: same constructor arguments as
``zoo/am/policy.py:50-122``, same ``forward`` signature and output dict as
``models/common/constructive/base.py:154-263``, and the SAME module tree — hence identical
``state_dict()`` keys, so reference checkpoints load unchanged.

What runs where
  * encoder + cache fold (``zoo/am/encoder.py``, ``nn/graph/attnnet.py``, ``zoo/am/decoder.py:201-228``): the MFMA-shaped
    part of the path, on the library's own kernels in every precision regime for inference rollouts — the fused per-instance
    kernels up to 128 nodes (``csrc/am_encoder.hip`` 16-bit, ``csrc/am_encoder_f32.hip`` fp32 = the bit-identical
    configuration), token-tile launches beyond; training: one fused forward launch for an instance-norm stack and per-op
    backward kernels behind autograd (``train_ops.py``). The torch modules compute only where no kernel serves the call
    (training beyond 128 nodes, train-mode batch norm beyond the kernels' shapes, ``fused_encoder=False``), announced by one RuntimeWarning.
  * the whole ``while not done`` loop (base.py:226-238) — context, pointer attention, logits
    processing, selection, env transition: ONE launch of ``rl4co_am_decode`` (no grad).
  * reward: ``env.get_reward`` -> ``rl4co_tour_length_f32``.
  * training: REINFORCE differentiates ``log_likelihood`` (reinforce.py:101). The rollout is
    sampled without grad by the kernel; ``log_likelihood`` is then re-evaluated teacher-forced
    over all T steps at once with autograd (the reference's own ``decode_type="evaluate"``
    semantics, decoding.py:448-461, as PPO already does, rl/ppo/ppo.py:128-170).
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import Tensor

from . import kernels as K
from .cache import FoldedCache, build_folded_cache, canonical_env
from .envs import RL4COEnvBase, get_env
from .tensordict import TensorDict

# ------------------------------------------------------------------------------------------------
# module tree (names = reference attribute names, so state_dict keys coincide)
# ------------------------------------------------------------------------------------------------


class _Skip(nn.Module):
    """nn/ops.py:9-15 SkipConnection: parameters live under ``.module``."""

    def __init__(self, module: nn.Module):
        super().__init__()
        self.module = module

    def forward(self, x):
        return x + self.module(x)


class _Norm(nn.Module):
    """nn/ops.py:30-54 Normalization: parameters live under ``.normalizer``."""

    def __init__(self, embed_dim: int, normalization: str):
        super().__init__()
        self.kind = normalization
        if normalization == "batch":
            self.normalizer = nn.BatchNorm1d(embed_dim, affine=True)
        elif normalization == "instance":
            self.normalizer = nn.InstanceNorm1d(embed_dim, affine=True)
        elif normalization == "layer":
            self.normalizer = "layer"
        else:
            raise ValueError(f"unknown normalization {normalization!r}")

    def forward(self, x: Tensor) -> Tensor:
        if self.kind == "batch":
            b, n, d = x.shape
            return self.normalizer(x.reshape(b * n, d)).view(b, n, d)
        if self.kind == "instance":
            if self.training and x.is_cuda and torch.is_grad_enabled():
                # same arithmetic as InstanceNorm1d (biased variance over the nodes, eps inside the
                # sqrt) written out: MIOpen's batch-norm backward that F.instance_norm dispatches to
                # costs 24 ms per POMO training step at 4096 x 100 nodes, these fused-by-autograd
                # elementwise ops about 2 ms
                n = self.normalizer
                mean = x.mean(dim=1, keepdim=True)
                var = x.var(dim=1, unbiased=False, keepdim=True)
                return (x - mean) * torch.rsqrt(var + n.eps) * n.weight + n.bias
            return self.normalizer(x.transpose(1, 2)).transpose(1, 2)
        mean = x.mean((1, 2), keepdim=True)
        var = x.var((1, 2), keepdim=True)
        return (x - mean) / torch.sqrt(var + 1e-5)


class _SelfAttention(nn.Module):
    """nn/attention.py:64-134 MultiHeadAttention (``Wqkv`` + ``out_proj``)."""

    def __init__(self, embed_dim: int, num_heads: int):
        super().__init__()
        self.num_heads = num_heads
        self.Wqkv = nn.Linear(embed_dim, 3 * embed_dim, bias=True)
        self.out_proj = nn.Linear(embed_dim, embed_dim, bias=True)

    def forward(self, x: Tensor, fused: bool = False) -> Tensor:
        b, n, d = x.shape
        if fused:  # training under bf16 autocast: the two projections on csrc/am_train_ops.hip
            from . import train_ops

            qkv = train_ops.linear(x, self.Wqkv.weight, self.Wqkv.bias)
            if self.num_heads == 8 and train_ops.attention_usable(qkv):
                return train_ops.linear(train_ops.attention(qkv), self.out_proj.weight, self.out_proj.bias)
            qkv = qkv.view(b, n, 3, self.num_heads, d // self.num_heads)
        else:
            qkv = self.Wqkv(x).view(b, n, 3, self.num_heads, d // self.num_heads)
        q, k, v = qkv.permute(2, 0, 3, 1, 4).unbind(0)
        out = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(b, n, d)
        if fused:
            return train_ops.linear(out, self.out_proj.weight, self.out_proj.bias)
        return self.out_proj(out)


class _FeedForward(nn.Module):
    """nn/mlp.py MLP(128 -> 512 -> 128, ReLU): parameters live under ``.lins``."""

    def __init__(self, embed_dim: int, hidden: int):
        super().__init__()
        dims = [embed_dim] + ([hidden] if hidden > 0 else []) + [embed_dim]
        self.lins = nn.ModuleList(nn.Linear(i, o) for i, o in zip(dims[:-1], dims[1:]))

    def forward(self, x: Tensor, fused: bool = False) -> Tensor:
        if fused and len(self.lins) == 2:
            from . import train_ops

            return train_ops.mlp(x, self.lins[0].weight, self.lins[0].bias, self.lins[1].weight, self.lins[1].bias)
        for lin in self.lins[:-1]:
            x = F.relu(lin(x))
        return self.lins[-1](x)


class _EncoderLayer(nn.Sequential):
    """nn/graph/attnnet.py:16-54. The FFN is created before the attention block, as in the
    reference, so a seeded construction consumes the RNG in the same order."""

    def __init__(self, embed_dim, num_heads, feedforward_hidden, normalization):
        ffn = _FeedForward(embed_dim, feedforward_hidden)
        attn = _SelfAttention(embed_dim, num_heads)
        super().__init__(_Skip(attn), _Norm(embed_dim, normalization), _Skip(ffn), _Norm(embed_dim, normalization))
        self.fused_linear = True  # training: projections / MLP / attention on the HIP kernels (False: library)
        self.fused_train = True   # training under bf16 autocast: skip + norm (and the above) on HIP kernels

    def forward(self, x):
        # training under bf16 autocast: projections, MLP, attention and skip + norm (instance: one kernel
        # each way; batch: statistics + apply) on csrc/am_train_ops.hip / am_train_attn.hip instead of
        # library GEMMs and autograd's elementwise chains
        if (self.fused_train and self.training and torch.is_grad_enabled() and x.is_cuda and self[1].kind in ("instance", "batch", "layer")
                and torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") in (torch.bfloat16, torch.float16)):
            from . import train_ops

            s_dtype = torch.get_autocast_dtype("cuda")
            x = x.to(s_dtype)
            attn, ffn = self[0].module, self[2].module
            gemm_ok = (self.fused_linear and train_ops.linear_usable(x, attn.Wqkv.weight, attn.out_proj.weight,
                                                                      *(lin.weight for lin in ffn.lins)))
            # the usual case: each residual sub-block is one autograd node (the skip's gradient joins in a GEMM epilogue)
            if (gemm_ok and attn.num_heads == 8 and len(ffn.lins) == 2 and self[1].kind == self[3].kind
                    and all(lin.bias is not None for lin in (attn.Wqkv, attn.out_proj, *ffn.lins))
                    and train_ops.block_usable(x, self[1].kind, attn.Wqkv.weight, attn.out_proj.weight,
                                               *(lin.weight for lin in ffn.lins))):
                x = train_ops.attention_block(x, attn, self[1])
                return train_ops.mlp_block(x, ffn, self[3])
            from . import _lib as _l

            _l.warn_fallback(f"train-block/{tuple(x.shape[1:])}/{self[1].kind}",
                             f"training encoder layer on {tuple(x.shape)} {self[1].kind}-norm activations is not served by the fused "
                             "sub-block kernels (graph beyond the attention / norm kernels' node limit, or an unusual layer "
                             "shape): falling back piecewise to the per-op kernels and torch")
            for skip, norm in ((self[0], self[1]), (self[2], self[3])):
                if x.dtype != s_dtype:  # torch's norm under autocast hands back fp32: the kernels take the 16-bit rows
                    x = x.to(s_dtype)
                s = skip.module(x, fused=gemm_ok)
                if norm.kind == "batch" and train_ops.batch_usable(x, s):
                    x = train_ops.skip_batch_norm(x, s, norm.normalizer)
                elif norm.kind == "instance" and train_ops.usable(x, s, "instance"):
                    x = train_ops.skip_instance_norm(x, s, norm.normalizer.weight, norm.normalizer.bias, norm.normalizer.eps)
                elif norm.kind == "layer" and train_ops.usable(x, s):
                    x = train_ops.skip_layer_norm(x, s)
                else:
                    x = norm(x + s)
            return x
        if x.is_cuda and self.training and torch.is_grad_enabled() and torch.is_autocast_enabled():
            from . import _lib as _l

            _l.warn_fallback(f"train-autocast/{torch.get_autocast_dtype('cuda')}",
                             f"training under torch.autocast({torch.get_autocast_dtype('cuda')}): the training-encoder kernels "
                             "serve bfloat16 and float16 autocast; this regime (or a layer-norm / unusual layer shape) trains "
                             "on the torch encoder")
        return super().forward(x)


class _GraphAttentionNetwork(nn.Module):
    """nn/graph/attnnet.py:57-106"""

    def __init__(self, num_heads, embed_dim, num_layers, normalization, feedforward_hidden):
        super().__init__()
        self.layers = nn.Sequential(
            *(_EncoderLayer(embed_dim, num_heads, feedforward_hidden, normalization) for _ in range(num_layers))
        )
        self.fused_stack = True  # training, instance norm: one forward launch for the whole stack (False: per sub-block)

    def forward(self, x):
        # training under 16-bit autocast, instance norm (POMO): the whole stack's forward as ONE launch of the fused
        # per-instance kernel, which keeps what the per-op backward kernels read (train_ops.encoder_stack)
        if (self.fused_stack and self.training and torch.is_grad_enabled() and x.is_cuda and torch.is_autocast_enabled()
                and torch.get_autocast_dtype("cuda") in (torch.bfloat16, torch.float16)
                and all(getattr(l, "fused_train", False) and getattr(l, "fused_linear", False) for l in self.layers)):
            from . import train_ops

            x16 = x.to(torch.get_autocast_dtype("cuda"))
            if train_ops.stack_usable(x16, self.layers):
                return train_ops.encoder_stack(x16, self.layers)
        return self.layers(x)


def _train_kernels_active(x: Tensor) -> bool:
    """16-bit autocast training on the GPU (bfloat16, or float16 = the reference's default "16-mixed"): the regimes
    csrc/am_train_ops.hip is built for (csrc/elem16.h)."""
    return (x.is_cuda and torch.is_grad_enabled() and torch.is_autocast_enabled()
            and torch.get_autocast_dtype("cuda") in (torch.bfloat16, torch.float16))


class _TSPInit(nn.Module):
    """env_embeddings/init.py:55-68"""

    def __init__(self, embed_dim):
        super().__init__()
        self.init_embed = nn.Linear(2, embed_dim, True)

    def forward(self, td):
        locs = td["locs"]
        if _train_kernels_active(locs):
            from . import train_ops

            return train_ops.init_embed(locs, self.init_embed)
        return self.init_embed(locs)


class _VRPInit(nn.Module):
    """env_embeddings/init.py:115-136"""

    def __init__(self, embed_dim):
        super().__init__()
        self.init_embed = nn.Linear(3, embed_dim, True)
        self.init_embed_depot = nn.Linear(2, embed_dim, True)

    def forward(self, td):
        locs = td["locs"]
        feats = torch.cat((locs[:, 1:, :], td["demand"][..., None]), -1)
        if _train_kernels_active(locs):
            from . import train_ops

            return torch.cat((train_ops.init_embed(locs[:, :1, :], self.init_embed_depot),
                              train_ops.init_embed(feats, self.init_embed)), -2)
        return torch.cat((self.init_embed_depot(locs[:, :1, :]), self.init_embed(feats)), -2)


class _OPInit(_VRPInit):
    """env_embeddings/init.py:254-280: as the VRP embedding with the customers' PRIZE as third feature"""

    def forward(self, td):
        locs = td["locs"]
        feats = torch.cat((locs[:, 1:, :], td["prize"][..., 1:, None]), -1)
        if _train_kernels_active(locs):
            from . import train_ops

            return torch.cat((train_ops.init_embed(locs[:, :1, :], self.init_embed_depot),
                              train_ops.init_embed(feats, self.init_embed)), -2)
        return torch.cat((self.init_embed_depot(locs[:, :1, :]), self.init_embed(feats)), -2)


class _PCTSPInit(nn.Module):
    """env_embeddings/init.py:283-312: customers (x, y, expected prize, penalty), depot (x, y)"""

    def __init__(self, embed_dim):
        super().__init__()
        self.init_embed = nn.Linear(4, embed_dim, True)
        self.init_embed_depot = nn.Linear(2, embed_dim, True)

    def forward(self, td):
        locs = td["locs"]
        feats = torch.cat((locs[:, 1:, :], td["expected_prize"][..., None], td["penalty"][..., 1:, None]), -1)
        if _train_kernels_active(locs):
            from . import train_ops

            return torch.cat((train_ops.init_embed(locs[:, :1, :], self.init_embed_depot),
                              train_ops.init_embed(feats, self.init_embed)), -2)
        return torch.cat((self.init_embed_depot(locs[:, :1, :]), self.init_embed(feats)), -2)


class _VRPTWInit(nn.Module):
    """env_embeddings/init.py:139-153: customers (x, y, demand, tw start, tw end, service time), depot (x, y)"""

    def __init__(self, embed_dim):
        super().__init__()
        self.init_embed = nn.Linear(6, embed_dim, True)
        self.init_embed_depot = nn.Linear(2, embed_dim, True)

    def forward(self, td):
        locs = td["locs"]
        feats = torch.cat((locs[:, 1:, :], td["demand"][..., None], td["time_windows"][..., 1:, :].to(locs.dtype),
                           td["durations"][..., 1:, None].to(locs.dtype)), -1)
        if _train_kernels_active(locs):
            from . import train_ops

            return torch.cat((train_ops.init_embed(locs[:, :1, :], self.init_embed_depot),
                              train_ops.init_embed(feats, self.init_embed)), -2)
        return torch.cat((self.init_embed_depot(locs[:, :1, :]), self.init_embed(feats)), -2)


class _VRPTWContext(nn.Module):
    """env_embeddings/context.py:152-166: current node embedding, remaining capacity, current time"""

    def __init__(self, embed_dim):
        super().__init__()
        self.project_context = nn.Linear(embed_dim + 2, embed_dim, bias=False)


class _PDPInit(nn.Module):
    """env_embeddings/init.py:335-360: depot (x, y) | pickups (x, y, x', y' of the delivery) | deliveries (x, y)"""

    def __init__(self, embed_dim):
        super().__init__()
        self.init_embed_depot = nn.Linear(2, embed_dim, True)
        self.init_embed_pick = nn.Linear(4, embed_dim, True)
        self.init_embed_delivery = nn.Linear(2, embed_dim, True)

    def features(self, td):
        locs = td["locs"]
        half = (locs.shape[-2] - 1) // 2
        pick = torch.cat((locs[:, 1 : half + 1, :], locs[:, half + 1 :, :]), -1)
        return ((locs[:, :1, :], self.init_embed_depot), (pick, self.init_embed_pick),
                (locs[:, half + 1 :, :], self.init_embed_delivery))

    def forward(self, td):
        if _train_kernels_active(td["locs"]):
            from . import train_ops

            return torch.cat([train_ops.init_embed(f.contiguous(), lin) for f, lin in self.features(td)], -2)
        return torch.cat([lin(f) for f, lin in self.features(td)], -2)


class _NodeContext(nn.Module):
    """env_embeddings/context.py:232-243 (PDP): the current node embedding alone"""

    def __init__(self, embed_dim):
        super().__init__()
        self.project_context = nn.Linear(embed_dim, embed_dim, bias=False)


class AttentionModelEncoder(nn.Module):
    """zoo/am/encoder.py:12-87"""

    def __init__(self, embed_dim=128, env_name="tsp", num_heads=8, num_layers=3, normalization="batch",
                 feedforward_hidden=512):
        super().__init__()
        self.env_name = env_name = canonical_env(env_name)
        self.init_embedding = {"tsp": _TSPInit, "cvrp": _VRPInit, "op": _OPInit, "pctsp": _PCTSPInit,
                               "pdp": _PDPInit, "cvrptw": _VRPTWInit}[env_name](embed_dim)
        self.net = _GraphAttentionNetwork(num_heads, embed_dim, num_layers, normalization, feedforward_hidden)

    def forward(self, td):
        init_h = self.init_embedding(td)
        return self.net(init_h), init_h


class _TSPContext(nn.Module):
    """env_embeddings/context.py:105-134 — parameters only; the arithmetic is in the kernel."""

    def __init__(self, embed_dim):
        super().__init__()
        self.project_context = nn.Linear(2 * embed_dim, embed_dim, bias=False)
        self.W_placeholder = nn.Parameter(torch.Tensor(2 * embed_dim).uniform_(-1, 1))


class _VRPContext(nn.Module):
    """env_embeddings/context.py:137-149"""

    def __init__(self, embed_dim):
        super().__init__()
        self.project_context = nn.Linear(embed_dim + 1, embed_dim, bias=False)


class _Pointer(nn.Module):
    """nn/attention.py:218-320 PointerAttention: holds ``project_out``."""

    def __init__(self, embed_dim, out_bias=False):
        super().__init__()
        self.project_out = nn.Linear(embed_dim, embed_dim, bias=out_bias)


class AttentionModelDecoder(nn.Module):
    """zoo/am/decoder.py:43-228 parameter holder + cache builder for the fused kernel."""

    def __init__(self, embed_dim=128, num_heads=8, env_name="tsp", mask_inner=True,
                 use_graph_context=True, check_nan=True):
        super().__init__()
        self.env_name = env_name = canonical_env(env_name)
        self.embed_dim = embed_dim
        self.num_heads = num_heads
        self.mask_inner = mask_inner
        self.check_nan = check_nan
        self.context_embedding = {"tsp": _TSPContext, "cvrp": _VRPContext, "op": _VRPContext, "pctsp": _VRPContext,
                                  "pdp": _NodeContext, "cvrptw": _VRPTWContext}[env_name](embed_dim)
        self.dynamic_embedding = nn.Module()  # StaticEmbedding (dynamic.py:47-57): no parameters
        self.pointer = _Pointer(embed_dim)
        self.project_node_embeddings = nn.Linear(embed_dim, 3 * embed_dim, bias=False)
        self.project_fixed_context = nn.Linear(embed_dim, embed_dim, bias=False)
        self.use_graph_context = use_graph_context

    def precompute_cache(self, h: Tensor, cache_dtype: torch.dtype,
                         gemm_dtype: torch.dtype = torch.float32, fold: bool = True) -> FoldedCache:
        """zoo/am/decoder.py:201-228, folded (rl4co_amd/cache.py)."""
        return build_folded_cache(
            self.env_name, h,
            w_node=self.project_node_embeddings.weight,
            w_out=self.pointer.project_out.weight,
            w_ctx=self.context_embedding.project_context.weight,
            w_fixed=self.project_fixed_context.weight if self.use_graph_context else None,
            w_placeholder=getattr(self.context_embedding, "W_placeholder", None),
            cache_dtype=cache_dtype,
            gemm_dtype=gemm_dtype,
            fold=fold,
        )


# ------------------------------------------------------------------------------------------------
# decode-type parsing (utils/decoding.py:17-35, 191-330)
# ------------------------------------------------------------------------------------------------

_KNOWN_DECODE_TYPES = ("greedy", "sampling", "multistart_greedy", "multistart_sampling", "evaluate")


def _cuda_tensors(objs):
    """The CUDA tensors of an encoder result (a FoldedCache or None, the node embeddings, the initial embeddings)."""
    for o in objs:
        if torch.is_tensor(o):
            if o.is_cuda:
                yield o
        elif isinstance(o, FoldedCache):
            for k in FoldedCache.TENSOR_FIELDS:
                t = getattr(o, k)
                if t is not None and t.is_cuda:
                    yield t


def _parse_decode_type(decode_type: str):
    if decode_type not in _KNOWN_DECODE_TYPES:
        if decode_type == "beam_search":
            raise NotImplementedError("beam_search is outside the accelerated hot path (SURVEY.md §2 row 2)")
        decode_type = "sampling"  # decoding.py:27-35 falls back to Sampling
    multistart = "multistart" in decode_type
    mode = "greedy" if "greedy" in decode_type else ("evaluate" if decode_type == "evaluate" else "sampling")
    return mode, multistart


class AttentionModelPolicy(nn.Module):
    """Attention Model policy (Kool et al. 2019) with the autoregressive loop on the MI355X.

    Args follow ``zoo/am/policy.py:50-85``. Extra, engine-specific arguments:
        cache_dtype: dtype of the three streamed cache planes (``torch.bfloat16`` / ``torch.float16`` halve the bytes
            per decode step; ``torch.float32`` is the parity configuration). ``None`` (the default SINCE r03; it was
            float32 before: callers that relied on fp32 planes under an ambient autocast must now ask for them) follows the
            precision regime the way the reference's own cache does — its K / V / logit key are the output of a Linear
            and therefore 16-bit under autocast (zoo/am/decoder.py:201-228): float32 without autocast, bfloat16 under
            bf16 autocast, under fp16 autocast float16 for inference rollouts and bfloat16 for training steps (wider
            exponent for the backward's intermediates; ``cache_dtype=torch.float16`` selects the fp16 builds of the
            multistart rollout and the MMA backward instead). So ``RL4COTrainer()`` with its default
            ``precision="16-mixed"`` over ``AttentionModelPolicy(env_name)`` reaches the fast kernels with no extra argument.
        encoder_autocast: optional autocast dtype for the encoder GEMMs (otherwise the ambient autocast decides).
    """

    def __init__(self, env_name: str = "tsp", embed_dim: int = 128, num_encoder_layers: int = 3,
                 num_heads: int = 8, normalization: str = "batch", feedforward_hidden: int = 512,
                 use_graph_context: bool = True, mask_inner: bool = True, check_nan: bool = True,
                 temperature: float = 1.0, tanh_clipping: float = 10.0, mask_logits: bool = True,
                 train_decode_type: str = "sampling", val_decode_type: str = "greedy",
                 test_decode_type: str = "greedy", cache_dtype: torch.dtype | None = None,
                 encoder_autocast: torch.dtype | None = None, fused_encoder: bool = True,
                 fused_backward: bool = True, teacher_variant: str = "auto", fold: bool = True,
                 train_half_as_bf16: bool = False, **unused_kwargs):
        super().__init__()
        if isinstance(env_name, RL4COEnvBase):
            env_name = env_name.name
        if embed_dim != 128 or num_heads != 8:
            raise ValueError("the fused decode kernel is specialised for embed_dim=128, num_heads=8")
        self.env_name = env_name = canonical_env(env_name)
        self.encoder = AttentionModelEncoder(embed_dim, env_name, num_heads, num_encoder_layers,
                                             normalization, feedforward_hidden)
        self.decoder = AttentionModelDecoder(embed_dim, num_heads, env_name, mask_inner,
                                             use_graph_context, check_nan)
        self.temperature = temperature
        self.tanh_clipping = tanh_clipping
        self.mask_logits = mask_logits
        self.train_decode_type = train_decode_type
        self.val_decode_type = val_decode_type
        self.test_decode_type = test_decode_type
        self.cache_dtype = cache_dtype
        self.encoder_autocast = encoder_autocast
        # inference rollouts run encoder + cache fold on the hand-written MFMA kernels of their regime (csrc/am_encoder.hip
        # 16-bit, csrc/am_encoder_f32.hip / am_tokens_f32.hip fp32); False: the torch encoder (a debugging switch)
        self.fused_encoder = fused_encoder
        # training: gradient of the log-likelihood w.r.t. the folded cache from the HIP backward
        # kernel (csrc/am_teacher.hip) instead of a dense [B,T,N] torch re-evaluation
        self.fused_backward = fused_backward
        self.teacher_variant = teacher_variant  # "auto" | "replay" | "mma" (teacher.run_backward)
        # fold=False: the reference's own association of the decoder (per-step project_context / project_out GEMVs,
        # raw logit key; cache.py) — the strictest greedy-parity configuration (fp32 encoder kernel, TSP / CVRP,
        # inference only): measured 4 instead of 10 near-tie flips in 4096 TSP-100 tours against the reference
        self.fold = fold
        # TRAINING steps under fp16 autocast (Lightning's default "16-mixed"). Off (default): the step runs on the fp16
        # builds of the training-encoder kernels (csrc/*_f16.hip: IEEE-half operands, conversions overflow to infinity,
        # which is what GradScaler's inf check expects) — the reference's own regime. On: the same step on the bf16 builds
        # (same 16-bit storage and MFMA rate, 8 instead of 11 significant bits, no overflow under the loss scale) — a
        # precision CHANGE the caller opts into. Inference under fp16 autocast is served by fp16 kernels either way
        self.train_half_as_bf16 = train_half_as_bf16
        self._packed = None
        self._ambient_autocast = None  # set for the duration of a forward() entered under torch.autocast
        self._bwd_err = None  # device int32 word the teacher backward ORs its sticky bits into (read with the next status)
        # (r06) TRAINING steps on a fixed-horizon environment (TSP: every trajectory takes exactly n steps) need nothing from
        # the rollout's status words to go on — the horizon is known — so their read-back is ASYNCHRONOUS: copied into pinned
        # memory behind the launches, checked (the reference's assertions raised) when the NEXT step finishes or at
        # check_backward_errors(). The synchronous read-back held the host at the end of every rollout and the chip then
        # idled while Python enqueued the loss, the backward and the optimizer (~150 launches): ~0.5 ms of a 19 ms POMO step.
        # Off: every rollout raises its own assertions before it returns, as the reference does.
        self.async_train_status = not __import__("os").environ.get("RL4CO_SYNC_TRAIN_STATUS")  # (the variable: same-box A/B timing)
        self._pending_status = None  # (pinned int32[6], event) of the last asynchronously read status
        self._status_host: list = []
        self._philox_calls = 0
        self.last_instance_steps = 0
        self.last_rows_read = 0
        self.encode_events: list | None = None
        self.decode_events: list | None = None  # set to [] by bench.py to time the decode launches

    # -- helpers --------------------------------------------------------------------------------
    def _packed_encoder(self):
        if self._packed is None:
            from .encoder import PackedEncoder

            self._packed = PackedEncoder(self)
        return self._packed

    def _encode_tokens_bf16(self, td):
        """Inference encoder for graphs beyond the fused kernel's 128 nodes (C5: CVRP-500): the same
        layer algebra on the token-parallel kernels of the training path — init embedding, the four
        projections per layer with bias / ReLU epilogues (csrc/am_train_ops.hip), eval-mode batch norm
        as one affine pass — and the flash-style attention kernel (csrc/am_attn_flash.hip: keys / values
        streamed through LDS with an online softmax) for the N x N attention. bf16 activations; no ATen
        kernel on the path."""
        from . import train_ops as T

        enc = self.encoder
        init = enc.init_embedding
        bf = self._encoder_regime()  # the 16-bit autocast type this rollout computes in (bfloat16 or float16)
        assert bf in (torch.bfloat16, torch.float16)
        embed = lambda f, lin: T.init_embed(f, lin, dtype=bf)  # noqa: E731
        if self.env_name == "pdp":
            x = torch.cat([embed(f.contiguous(), lin) for f, lin in init.features(td)], -2)
        elif self.env_name == "cvrptw":
            locs = td["locs"]
            feats = torch.cat((locs[:, 1:, :], td["demand"][..., None], td["time_windows"][..., 1:, :].float(),
                               td["durations"][..., 1:, None].float()), -1)
            x = torch.cat((embed(locs[:, :1, :], init.init_embed_depot), embed(feats, init.init_embed)), -2)
        elif self.env_name in ("cvrp", "op", "pctsp"):
            locs = td["locs"]
            third = {"cvrp": "demand", "op": "prize", "pctsp": "expected_prize"}[self.env_name]
            third = td[third][..., 1:] if self.env_name == "op" else td[third]
            feats = torch.cat((locs[:, 1:, :], third[..., None]), -1)
            if self.env_name == "pctsp":
                feats = torch.cat((feats, td["penalty"][..., 1:, None]), -1)
            x = torch.cat((embed(locs[:, :1, :], init.init_embed_depot), embed(feats, init.init_embed)), -2)
        else:
            x = embed(td["locs"], init.init_embed)
        init_h = x
        b, n, d = x.shape
        for layer in enc.net.layers:
            attn, norm1, ffn, norm2 = layer[0].module, layer[1].normalizer, layer[2].module, layer[3].normalizer
            x2 = x.reshape(b * n, d)
            qkv = T._gemm(x2, attn.Wqkv.weight.to(bf).contiguous(), attn.Wqkv.bias.float().contiguous())
            o = T.attention_flash(qkv.view(b, n, 3 * d)).view(b * n, d)  # keys / values streamed through LDS, any N
            s_ = T._gemm(o, attn.out_proj.weight.to(bf).contiguous(), attn.out_proj.bias.float().contiguous())
            x2 = T.skip_batch_norm_eval(x2, s_, norm1)
            h = T._gemm(x2, ffn.lins[0].weight.to(bf).contiguous(), ffn.lins[0].bias.float().contiguous(), relu=True)
            s_ = T._gemm(h, ffn.lins[1].weight.to(bf).contiguous(), ffn.lins[1].bias.float().contiguous())
            x = T.skip_batch_norm_eval(x2, s_, norm2).view(b, n, d)
        return x, init_h

    def _bf16_regime(self) -> bool:
        """The encoder's GEMM inputs are bf16: asked for by the constructor (``encoder_autocast=torch.bfloat16``) or by
        an ambient ``torch.autocast("cuda", dtype=torch.bfloat16)`` — what Lightning's ``precision="bf16-mixed"`` wraps
        around training steps AND the validation / ``RolloutBaseline`` rollouts (utils/trainer.py:57 picks the
        precision; the reference's default "16-mixed" is fp16 autocast, served by the fp16 builds of the same kernels)."""
        return self._encoder_regime() == torch.bfloat16

    def _plane_dtype(self, training: bool) -> torch.dtype:
        """dtype of the streamed planes of this call: the constructor's ``cache_dtype``, else by regime (see the class
        docstring)."""
        if self.cache_dtype is not None:
            return self.cache_dtype
        regime = self._encoder_regime()
        if regime == torch.bfloat16:
            return torch.bfloat16
        if regime == torch.float16:
            # training: bf16 planes — the matrix-core rollout and backward then carry their intermediates (softmax
            # numerators, d logits, d scores) with an 8-bit exponent: no underflow of small REINFORCE advantages and no
            # overflow under GradScaler's loss scale, at the same speed (fp16 builds of both kernels exist and are taken
            # when the caller asks for cache_dtype=torch.float16)
            return torch.bfloat16 if training else torch.float16
        return torch.float32

    def _encoder_regime(self):
        """Autocast dtype of the encoder: the constructor's choice, else the ambient one (None = fp32)."""
        return self.encoder_autocast if self.encoder_autocast is not None else self._ambient_autocast

    def _token_encoder_usable(self, td) -> bool:
        layer0 = self.encoder.net.layers[0]
        return (td["locs"].is_cuda and self._encoder_regime() in (torch.bfloat16, torch.float16) and layer0[1].kind == "batch"
                and not self.training and len(layer0[2].module.lins) == 2 and layer0[2].module.lins[0].out_features % 128 == 0)

    def _encode(self, td):
        regime = self._encoder_regime()
        if regime == torch.float16 and self.train_half_as_bf16 and torch.is_grad_enabled():
            regime = torch.bfloat16  # opt-in: fp16-autocast TRAINING steps on the bf16 training kernels
        if regime is not None and td["locs"].is_cuda:
            with torch.autocast("cuda", dtype=regime):
                return self.encoder(td)
        return self.encoder(td)

    @staticmethod
    def _max_horizon(env_name: str, n: int) -> int:
        # TSP: exactly N steps. CVRP: every customer + at most one depot visit per customer + 1.
        # OP: every customer once, the closing depot visit, and a depot pick at step 0 costs one more.
        # PCTSP: every customer once and the closing depot visit (the depot is masked at step 0).
        # PDP: every node once (the depot too under force_start_at_depot).
        return n if env_name in ("tsp", "pctsp", "pdp") else (n + 2 if env_name == "op" else 2 * n)

    def _initial_state(self, td, num_starts: int):
        """State tensors the kernel updates in place; with multistart the rows are expanded
        s-major (batchify, ops.py:10-30) while instance-level data (cache, demand) stays [B,...]."""
        s = max(num_starts, 1)

        def rep(x: Tensor) -> Tensor:
            x = x.reshape(x.shape[0], -1) if x.dim() > 1 else x
            out = x.unsqueeze(0).expand(s, *x.shape).reshape(s * x.shape[0], *x.shape[1:]).contiguous()
            return out

        st = {
            "action_mask": rep(td["action_mask"]),
            "current_node": rep(td["current_node"].reshape(-1)),
            "done": rep(td["done"].reshape(-1)),
        }
        if self.env_name == "tsp":
            st["first_node"] = rep(td["first_node"].reshape(-1))
            st["i"] = rep(td["i"].reshape(-1))
        elif self.env_name == "pdp":
            st["available"] = rep(td["available"])
            st["to_deliver"] = rep(td["to_deliver"])
            st["i"] = rep(td["i"].reshape(-1))
        elif self.env_name == "pctsp":
            st["real_prize"] = td["real_prize"].contiguous()  # instance data [B_inst, N], depot column 0
            st["cur_total_prize"] = rep(td["cur_total_prize"].reshape(-1))
            st["prize_required"] = rep(td["prize_required"].reshape(-1))
            st["i"] = rep(td["i"].reshape(-1))
            st["visited"] = rep(td["visited"])
        elif self.env_name == "op":
            st["locs"] = td["locs"].contiguous()              # instance data, like CVRP's demand
            st["max_length"] = td["max_length"].contiguous()  # [B_inst, N] entry limits
            st["tour_length"] = rep(td["tour_length"].reshape(-1))
            st["i"] = rep(td["i"].reshape(-1))
            st["visited"] = rep(td["visited"])
        else:
            st["demand"] = td["demand"].contiguous()
            st["used_capacity"] = rep(td["used_capacity"].reshape(-1))
            st["vehicle_capacity"] = rep(td["vehicle_capacity"].reshape(-1))
            st["visited"] = rep(td["visited"])
            if self.env_name == "cvrptw":  # instance data as fp32 (the reference keeps integer-valued windows)
                st["locs"] = td["locs"].contiguous()
                st["time_windows"] = td["time_windows"].float().contiguous()
                st["durations"] = td["durations"].float().contiguous()
                st["current_time"] = rep(td["current_time"].reshape(-1))
        if s == 1:
            st = {k: (v.clone() if k not in ("demand", "locs", "max_length", "real_prize", "time_windows", "durations") else v) for k, v in st.items()}
        return st

    # -- forward (constructive/base.py:154-263) ---------------------------------------------------
    def forward(self, td: TensorDict, *args, **kwargs) -> dict:
        """Ambient autocast (Lightning's mixed-precision plugin wraps every step in it) selects the ENCODER's regime
        only: cache fold, context tables, reward and log-likelihood arithmetic are fp32 by contract, so everything
        but the encoder runs with autocast switched off and the encoder re-enters it explicitly (``_encode``)."""
        on_cuda = td["locs"].is_cuda if "locs" in td.keys() else torch.cuda.is_available()
        if on_cuda and torch.is_autocast_enabled():
            self._ambient_autocast = torch.get_autocast_dtype("cuda")
            try:
                with torch.autocast("cuda", enabled=False):
                    return self._forward(td, *args, **kwargs)
            finally:
                self._ambient_autocast = None
        return self._forward(td, *args, **kwargs)

    def _encode_for_rollout(self, td: TensorDict, grad_path: bool, cache_dtype, return_hidden: bool, return_init_embeds: bool):
        """Step 1 of ``_forward`` (constructive/base.py:196-204): (cache | None, hidden | None, init_embeds | None). Inference:
        the fused kernels (16-bit regimes and exact fp32, N <= 128) return the folded cache itself; the token-tile kernels
        (N > 128) and the torch modules (training: autograd) return the node embeddings for the fold that follows."""
        init_embeds = None
        regime16 = self._encoder_regime() if self._encoder_regime() in (torch.bfloat16, torch.float16) else None
        # fused MFMA encoder: a 16-bit autocast regime (bf16, or fp16 = the reference's default "16-mixed") whose planes
        # are fp32 or that same 16-bit type
        # (return_init_embeds: one more launch with the instance in grid.y — up to 65535 instances)
        ie_ok = not return_init_embeds or td["locs"].shape[0] <= 65535
        use_fused = (self.fused_encoder and self.fold and regime16 is not None and not grad_path
                     and cache_dtype in (torch.float32, regime16) and ie_ok
                     and self._packed_encoder().supported(td))
        # fp32 regime (no autocast: the bit-identical configuration): the exact-fp32 MFMA encoder (csrc/am_encoder_f32.hip),
        # planes in any type, also with fold=False (tsp / cvrp: the reference's own association of the decoder)
        use_fused_f32 = (self.fused_encoder and self._encoder_regime() is None and not grad_path and td["locs"].is_cuda
                         and (self.fold or self.env_name in ("tsp", "cvrp")) and ie_ok
                         and self._packed_encoder().supported(td, torch.float32))
        if use_fused_f32:
            use_fused, regime16 = True, torch.float32
        if use_fused:
            if self.encode_events is not None:  # bench.py: HIP events around the encoder launch
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record()
            # return_init_embeds (zoo/am/encoder.py:84-103): one more launch of the kernels' own init-embedding routine
            init_embeds = (torch.empty((*td["locs"].shape[:2], 128), dtype=regime16, device=td["locs"].device)
                           if return_init_embeds else None)
            cache, hidden = self._packed_encoder().encode(td, cache_dtype, want_hidden=return_hidden, act_dtype=regime16,
                                                          fold=self.fold, init_embeds_out=init_embeds)
            if self.encode_events is not None:
                ev1.record()
                self.encode_events.append((ev0, ev1))
        else:
            cache = None
            if not grad_path and self.fused_encoder and self._token_encoder_usable(td):
                if self.encode_events is not None:
                    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    ev0.record()
                hidden, init_embeds = self._encode_tokens_bf16(td)  # N > 128: token-parallel kernels + flash attention
                if self.encode_events is not None:
                    ev1.record()
                    self.encode_events.append((ev0, ev1))
            else:
                if (td["locs"].is_cuda and not grad_path and self._encoder_regime() is not None
                        and self.fused_encoder and self.fold):  # (switched off by the caller: not a fallback)
                    from . import _lib as _l

                    n_nodes = td["action_mask"].shape[-1]
                    _l.warn_fallback(f"infer-encoder/{self._encoder_regime()}/{n_nodes > 128}/{self.fused_encoder}/{self.fold}",
                                     f"inference encoder for {n_nodes} nodes under autocast({self._encoder_regime()}) runs on torch: "
                                     "the fused MFMA encoder (up to 128 nodes) and the token-tile kernels (beyond) serve bf16 / fp16 with the "
                                     "folded cache and planes in fp32 or the activations' type")
                hidden, init_embeds = self._encode(td)
        return cache, hidden, init_embeds

    def _parse_decoding(self, td: TensorDict, env, phase: str, actions, decoding_kwargs: dict):
        """Step 2 of ``_forward``: the decoding arguments (utils/decoding.py:238-255, get_decoding_strategy / DecodingStrategy
        .__init__) as a namespace; pops what it consumes from ``decoding_kwargs`` (``philox_seed_dev`` / ``_defer_finish`` stay)."""
        from types import SimpleNamespace

        decode_type = decoding_kwargs.pop("decode_type", None)
        if actions is not None:
            decode_type = "evaluate"
        elif decode_type is None:
            decode_type = getattr(self, f"{phase}_decode_type")
        mode, multistart = _parse_decode_type(decode_type)
        temperature = decoding_kwargs.pop("temperature", self.temperature)
        tanh_clipping = decoding_kwargs.pop("tanh_clipping", self.tanh_clipping)
        mask_logits = decoding_kwargs.pop("mask_logits", self.mask_logits)
        # (not a reference argument) hand the [B, T, N] log-softmax of every step back as out["all_logp"]: the parity
        # tests and bench.py's parity block measure argmax regret / per-step agreement along a forced trajectory with it
        return_all_logp = bool(decoding_kwargs.pop("return_all_logp", False))
        store_all_logp = bool(decoding_kwargs.pop("store_all_logp", False)) or return_all_logp
        select_best = decoding_kwargs.pop("select_best", False)
        num_starts = decoding_kwargs.pop("num_starts", None)
        num_samples = decoding_kwargs.pop("num_samples", None)
        exp_noise = decoding_kwargs.pop("exp_noise", None)  # parity hook: injected Exp(1) draws
        seed = decoding_kwargs.pop("seed", None)
        multisample = bool(decoding_kwargs.pop("multisample", False))
        select_start_nodes_fn = decoding_kwargs.pop("select_start_nodes_fn", None)
        # sampling modifiers outside the path (top-k / top-p / a second temperature): their neutral values pass
        if decoding_kwargs.pop("top_k", 0) or decoding_kwargs.pop("top_p", 0.0):
            raise NotImplementedError("top-k / top-p sampling is not part of the fused decode kernel")
        decoding_kwargs.pop("top_p", None)
        if decoding_kwargs.pop("softmax_temp", None) is not None:
            raise NotImplementedError("softmax_temp is not supported; use temperature")
        # decoding.py:238-255, in the reference's order: the flags are checked as PASSED, then overridden by the counts
        # (SamplingEval passes decode_type="sampling", multisample=True, num_starts=n: it ends up multistart AND
        # multisample, i.e. n sampled rollouts per instance whose first node comes from `select_start_nodes_fn`)
        assert not (multistart and multisample), "Using both multistart and multisample is not supported"
        if num_samples and num_starts:
            assert not (num_samples > 1 and num_starts > 1), f"num_samples={num_samples} and num_starts={num_starts} are both > 1"
        if num_samples is not None:
            multisample = num_samples > 1
        if num_starts is not None:
            multistart = num_starts > 1
        if multistart or multisample:
            n_rep = num_starts if multistart else num_samples
            if n_rep is None:
                n_rep = env.get_num_starts(td)
        else:
            n_rep = 0

        return SimpleNamespace(mode=mode, multistart=multistart, n_rep=n_rep, temperature=temperature, tanh_clipping=tanh_clipping,
                               mask_logits=mask_logits, return_all_logp=return_all_logp, store_all_logp=store_all_logp,
                               select_best=select_best, exp_noise=exp_noise, seed=seed, select_start_nodes_fn=select_start_nodes_fn)

    def _forward(self, td: TensorDict, env: str | RL4COEnvBase | None = None, phase: str = "train",
                 calc_reward: bool = True, return_actions: bool = True, return_entropy: bool = False,
                 return_hidden: bool = False, return_init_embeds: bool = False,
                 return_sum_log_likelihood: bool = True, actions: Tensor | None = None,
                 max_steps: int = 1_000_000, **decoding_kwargs) -> dict:
        grad_path = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        # fold=False (the reference's own association of the decoder, tsp / cvrp) also trains since r06: the rollout runs on the
        # unfolded decode kernel, the gradient comes from the dense torch re-evaluation of the same trajectories
        # (evaluate_log_probs: the reference's modules applied in the reference's order) — no teacher-forced kernel, said once
        cache_dtype = self._plane_dtype(grad_path)
        # 1. encoder (+ the cache fold where the fused kernels produce it)
        es = getattr(self, "encoder_stream", None)
        if es is not None and not grad_path and td["locs"].is_cuda:
            # (graph.PriorityPipeline) the encoder launch goes out on a stream of its own — a high-priority one shared by
            # every rollout in flight — and this rollout's stream picks the cache up behind it
            cur = torch.cuda.current_stream()
            es.wait_stream(cur)
            with torch.cuda.stream(es):
                cache, hidden, init_embeds = self._encode_for_rollout(td, grad_path, cache_dtype, return_hidden, return_init_embeds)
            cur.wait_stream(es)
            for t in _cuda_tensors((cache, hidden, init_embeds)):
                t.record_stream(cur)
        else:
            cache, hidden, init_embeds = self._encode_for_rollout(td, grad_path, cache_dtype, return_hidden, return_init_embeds)
        if isinstance(env, str) or env is None:
            env = get_env(self.env_name if env is None else env)
        # 2. decoding arguments, in the reference's order of precedence
        opt = self._parse_decoding(td, env, phase, actions, decoding_kwargs)
        mode, multistart, n_rep = opt.mode, opt.multistart, opt.n_rep
        temperature, tanh_clipping, mask_logits = opt.temperature, opt.tanh_clipping, opt.mask_logits
        return_all_logp, store_all_logp, select_best = opt.return_all_logp, opt.store_all_logp or return_entropy, opt.select_best
        exp_noise, seed, select_start_nodes_fn = opt.exp_noise, opt.seed, opt.select_start_nodes_fn

        device = td["action_mask"].device
        b_inst, n = td["action_mask"].shape[0], td["action_mask"].shape[-1]
        b = b_inst * max(n_rep, 1)
        cache_g = None
        if cache is None and grad_path and self.fused_backward and hidden.is_cuda and self.fold:
            from . import teacher

            if teacher.supports(self.env_name, cache_dtype, n) and not return_entropy:
                # bf16 encoder output + bf16 planes + the MMA backward: ONE fold GEMM each way, the planes side by side
                half = (torch.bfloat16, torch.float16)
                fused_planes = hidden.dtype in half and cache_dtype in half and self.teacher_variant != "replay"
                # (activations in one 16-bit type, planes asked for in the other: the fold runs in the planes' type)
                h_fold = hidden.to(cache_dtype) if (fused_planes and hidden.dtype != cache_dtype) else hidden
                cache_g = teacher.build_cache_autograd(self.env_name, h_fold, self.decoder, fused_planes=fused_planes)
                cache = teacher.detached_cache(self.env_name, cache_g, cache_dtype)
        if cache is None:
            with torch.no_grad():
                regime = self._encoder_regime()  # fp16 (the reference's default "16-mixed"): the fold stays fp32
                cache = self.decoder.precompute_cache(hidden.detach(), cache_dtype,
                                                      torch.bfloat16 if (regime == torch.bfloat16 and cache_dtype != torch.float16)
                                                      else torch.float32, fold=self.fold)
        # (bench.py's byte model: the context rows a decode step gathers are fp32 or, r06, the planes' 16-bit type)
        self.last_ctx_elem_bytes = cache.ctx_cur.element_size() if cache.ctx_cur is not None else 4
        state = self._initial_state(td, n_rep)
        horizon = min(self._max_horizon(self.env_name, n), max_steps)
        if self.env_name == "pdp" and not getattr(env, "force_start_at_depot", False):
            horizon = min(horizon, n - 1)  # the depot is never visited: exactly one step per location (no padding
            #                                column: a trailing 0 would read as a depot visit in check_solution_validity)
        # status words read back ONCE per rollout: [sticky error bits, -, longest trajectory, streamed instance-steps,
        # cache rows streamed (one 64-bit counter: low word, high word)]
        status = torch.zeros(6, dtype=torch.int32, device=device)
        err = status[:1]

        # pre_decoder_hook (decoding.py:306-326): with multistart the first action is imposed per
        # start (log-prob 0) and consumes NO column of a caller-provided `actions` tensor — the
        # evaluate strategy replays `actions[..., step]` from the first DECODED step
        # (constructive/base.py:219-232).
        t0 = 1 if (multistart and n_rep >= 1) else 0
        if mode == "evaluate":
            given = actions.contiguous()
            tmax = t0 + given.shape[1]
            forced = torch.zeros((b, tmax), dtype=torch.int64, device=device)
            forced[:, t0:] = given
        else:
            forced = None
            tmax = horizon
        out_actions = torch.zeros((b, tmax), dtype=torch.int64, device=device)
        logps = torch.zeros((b, tmax), dtype=torch.float32, device=device)
        all_logps = torch.zeros((b, tmax, n), dtype=torch.float32, device=device) if store_all_logp else None

        if t0 == 1:
            if select_start_nodes_fn is not None:  # decoding.py:308-311: (td, env, num_starts) -> [num_starts * B] nodes, s-major
                first = select_start_nodes_fn(td, env, n_rep).to(device=device, dtype=torch.int64).contiguous()
            else:
                first = env.select_start_nodes(td, num_starts=n_rep)
            out_actions[:, 0] = first
            self._env_step_state(state, first, err)

        if mode == "sampling" and exp_noise is None:
            self._philox_calls += 1
            philox_seed = int(seed) if seed is not None else int(torch.randint(0, 2**62, (1,)).item())
        else:
            philox_seed = 0
        seed_dev = decoding_kwargs.pop("philox_seed_dev", None)  # graph.GraphedRollout: fresh noise per replay
        if self.decode_events is not None:  # bench.py: HIP events around the decode kernel launch
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        K.am_decode(
            cache, state, mode=mode, max_steps=tmax - t0, t0=t0, actions=out_actions, logps=logps, err=err,
            tanh_clipping=tanh_clipping, temperature=temperature, mask_inner=self.decoder.mask_inner,
            mask_logits=mask_logits, exp_noise=exp_noise, philox_seed=philox_seed, philox_seed_dev=seed_dev,
            forced_actions=forced, all_logps=all_logps, steps_summary=status[2:6],
            variant=getattr(self, "decode_variant", "auto"),  # ("auto": the library chooses; bench / probes may pin one)
        )
        if self.decode_events is not None:
            ev1.record()
            self.decode_events.append((ev0, ev1))
        # validity check on the padded action buffer (trailing depot zeros are neutral), into the
        # same error word — then ONE host sync for the whole rollout: horizon + every sticky bit
        # (an environment that is not one of this package's — e.g. the reference's own torch environment passed as
        # `env` — has no device-side check: its get_reward validates the unpadded actions itself, after the read-back)
        native_env = isinstance(env, RL4COEnvBase)
        checked = bool(native_env and calc_reward and env.check_solution and not (n_rep > 0 and select_best))
        if checked:
            env.check_solution_validity(td, out_actions, err=err)
        # tour-length environments: the reward goes out BEFORE the read-back. The buffer is padded to the longest
        # possible rollout and trailing zeros would change the association of the reference-ordered sums, so the kernel
        # takes the real horizon from the device (the decode launch's own step count, kernels.tour_length(horizon=))
        td_early = reward_early = None
        if (native_env and self.env_name in ("tsp", "pdp", "cvrp", "cvrptw") and calc_reward and mode != "evaluate"
                and (checked or not env.check_solution) and not (n_rep > 0 and select_best)
                and env.accepts_reward_horizon()):
            td_early = self._final_td(td, state, n_rep)
            reward_early = env.get_reward(td_early, out_actions, check_solution=False, horizon=(status[2:3], t0))
        if self._bwd_err is not None and self._bwd_err.device == device:
            # sticky bits of earlier teacher-forced backward launches ride on this read-back. The sink is ONE persistent
            # word, consumed IN PLACE in stream order (OR into the status, then zero): a backward that runs only after
            # further forwards (baseline / validation rollouts between a grad forward and its loss.backward(), PPO
            # re-evaluations) still writes into the live word and is reported by the next read-back
            status[:1].bitwise_or_(self._bwd_err)
            self._bwd_err.zero_()
        # everything above only ENQUEUES device work; `finish` performs the rollout's one host read-back and assembles the
        # reference's output dict. graph.GraphedRollout captures the part above in a HIP graph and calls `finish` after
        # every replay (it only reads the buffers the launches wrote).
        defer = bool(decoding_kwargs.pop("_defer_finish", False))
        launched = (out_actions, logps, all_logps, td_early, reward_early)

        from types import SimpleNamespace

        r = SimpleNamespace(
            launched=launched, status=status, t0=t0, td=td, env=env, state=state, n_rep=n_rep, b_inst=b_inst, n=n,
            device=device, grad_path=grad_path, cache_g=cache_g, cache=cache, cache_dtype=cache_dtype, hidden=hidden,
            init_embeds=init_embeds, mask_logits=mask_logits, tanh_clipping=tanh_clipping, temperature=temperature,
            return_entropy=return_entropy, select_best=select_best, calc_reward=calc_reward, checked=checked,
            return_sum_log_likelihood=return_sum_log_likelihood, return_actions=return_actions,
            return_all_logp=return_all_logp, return_hidden=return_hidden, return_init_embeds=return_init_embeds,
        )
        if defer:
            return lambda: self._finish_rollout(r)
        return self._finish_rollout(r)

    def _finish_rollout(self, r) -> dict:
        """Step 4 of ``_forward``: the rollout's ONE host read-back (status words), the reference's assertions, the
        differentiable re-evaluation of a training step, best-of selection, reward and the output dict (constructive/base.py:
        240-263). Re-runnable: ``graph.GraphedRollout`` calls it after every replay of the captured launches — it only reads
        the buffers they wrote (``r``: what ``_forward`` enqueued)."""
        out_actions, logps, all_logps, td_early, reward_early = r.launched  # re-runnable: a graph replay refills the same buffers
        status = r.status
        t0 = r.t0
        td = r.td
        env = r.env
        state = r.state
        n_rep = r.n_rep
        b_inst = r.b_inst
        n = r.n
        device = r.device
        grad_path = r.grad_path
        cache_g = r.cache_g
        cache = r.cache
        cache_dtype = r.cache_dtype
        hidden = r.hidden
        init_embeds = r.init_embeds
        mask_logits = r.mask_logits
        tanh_clipping = r.tanh_clipping
        temperature = r.temperature
        return_entropy = r.return_entropy
        select_best = r.select_best
        calc_reward = r.calc_reward
        checked = r.checked
        return_sum_log_likelihood = r.return_sum_log_likelihood
        return_actions = r.return_actions
        return_all_logp = r.return_all_logp
        return_hidden = r.return_hidden
        return_init_embeds = r.return_init_embeds
        from . import _lib as _l

        self._drain_status()  # an earlier training step's words (copied long ago: no wait) — its assertions come first
        if (self.async_train_status and grad_path and cache_g is not None and self.env_name == "tsp" and status.is_cuda
                and not select_best):
            # fixed horizon: nothing below depends on the status words; they travel to pinned memory behind the launches
            if len(self._status_host) < 2:
                self._status_host.append(torch.empty(6, dtype=torch.int32, pin_memory=True))
            host = self._status_host.pop(0)
            self._status_host.append(host)
            host.copy_(status, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._pending_status = (host, ev)
            t_used = out_actions.shape[1]
            self.last_instance_steps = out_actions.shape[0] * (t_used - t0)
            self.last_rows_read = 0
        else:
            err_bits, _, horizon_used, streamed, rows_lo, rows_hi = status.tolist()  # one 24-byte read-back, no reduction launches
            rows_read = (rows_hi << 32) | (rows_lo & 0xFFFFFFFF)
            t_used = t0 + int(horizon_used)
            self.last_instance_steps = int(streamed)  # instance-steps the decode launch really streamed
            self.last_rows_read = int(rows_read)      # cache rows (per plane) it read from HBM doing so
            _l.raise_for_error_bits(int(err_bits))
        out_actions = out_actions[:, :t_used].contiguous()
        logps = logps[:, :t_used]
        if all_logps is not None:
            all_logps = all_logps[:, :t_used]

        # td mirrors the reference's final state (batchified rows when multistart)
        td_out = td_early if td_early is not None else self._final_td(td, state, n_rep)
        td_out.set("action", out_actions[:, -1])

        # differentiable re-evaluation of the ROLLED-OUT rows (all s * b_inst of them: the replay needs the imposed
        # start nodes and the batchified state) — before any best-of selection narrows the rows
        full_logp = None  # [B, T, N] differentiable log-softmax, only when a differentiable entropy is asked for
        if grad_path and cache_g is not None:
            from . import teacher

            if self._bwd_err is None or self._bwd_err.device != device:
                self._bwd_err = torch.zeros(1, dtype=torch.int32, device=device)
            meta = dict(t0=t0, mask_inner=self.decoder.mask_inner, mask_logits=mask_logits, err_sink=self._bwd_err,
                        tanh_clipping=tanh_clipping, temperature=temperature, teacher_variant=self.teacher_variant)
            if self.env_name in ("cvrp", "cvrptw"):
                meta.update(demand=td["demand"], vehicle_capacity=td["vehicle_capacity"])
                if self.env_name == "cvrptw":
                    meta.update(locs=td["locs"], time_windows=td["time_windows"], durations=td["durations"])
            elif self.env_name == "op":
                meta.update(locs=td["locs"], max_length=td["max_length"])
            elif self.env_name == "pctsp":
                meta.update(real_prize=td["real_prize"], prize_required=td["prize_required"])
            step_logps = teacher.teacher_forced_logps(self.env_name, cache_g, cache, out_actions, logps, meta)
        elif grad_path:
            if hidden.is_cuda and self.fused_backward and not return_entropy:
                from . import _lib as _l

                t_max = __import__("rl4co_amd.teacher", fromlist=["max_nodes"]).max_nodes()
                why = ("fold=False keeps the reference's per-step association, which the backward kernels do not implement" if not self.fold else
                       f"{n} nodes are beyond the kernels' limit ({t_max})" if n > t_max else
                       f"{cache_dtype} planes are not served by the backward kernels (float32, bfloat16 or float16 planes) for this call")
                _l.warn_fallback(f"teacher/{self.env_name}/{n}/{cache_dtype}",
                                 f"teacher-forced backward for {self.env_name}: {why} — dense re-evaluation of all steps with autograd "
                                 "(16-bit regimes: glimpse attention and log-prob kernels between library GEMMs; fp32: torch)")
            step_logps = self.evaluate_log_probs(td, hidden, out_actions, n_rep, tanh_clipping, temperature,
                                                 mask_logits, skip_first=(t0 == 1), return_full=return_entropy)
            if return_entropy:
                step_logps, full_logp = step_logps
        else:
            step_logps = logps

        if n_rep > 0 and select_best:
            rewards = env.get_reward(td_out, out_actions)
            best = rewards.view(n_rep, b_inst).transpose(0, 1).max(dim=-1)[1]  # unbatchify + max
            rows = best * b_inst + torch.arange(b_inst, device=device)
            out_actions, logps, step_logps = out_actions[rows], logps[rows], step_logps[rows]
            if all_logps is not None:
                all_logps = all_logps[rows]
            if full_logp is not None:
                full_logp = full_logp[rows]
            td_out = td_out[rows] if hasattr(td_out, "__getitem__") else td_out
            reward = rewards[rows] if calc_reward else None
        elif reward_early is not None:
            reward = reward_early
        else:
            reward = (env.get_reward(td_out, out_actions, check_solution=False if checked else None)
                      if calc_reward else td_out.get("reward", None))
        if calc_reward:
            td_out.set("reward", reward)
        # decoding.py:56: on the kernel path this is the RL4CO_EBIT_NEG_INF_LOGP sticky bit (already
        # raised above); only the autograd re-evaluation needs its own check
        if grad_path and not bool((step_logps.detach() > -1000).all()):
            raise AssertionError("Logprobs should not be -inf, check sampling procedure!")
        outdict = {
            "reward": reward,
            "log_likelihood": step_logps.sum(1) if return_sum_log_likelihood else step_logps,
        }
        if return_actions:
            outdict["actions"] = out_actions
        if return_entropy:
            # ops.py:103-111 on the [B, T, N] log-probs. Under autograd the reference's entropy carries history (PPO's
            # entropy bonus differentiates it): then it is built from the differentiable log-softmax of the
            # re-evaluation, not from the kernel's (history-free) all_logps; the imposed multistart step has p = 1
            lp_src = all_logps
            if full_logp is not None:
                lp_src = full_logp if t0 == 0 else torch.cat([all_logps[:, :1], full_logp[:, 1:]], 1)
            lp = torch.nan_to_num(lp_src, nan=0.0)
            entropy = -(lp.exp() * lp).sum(dim=-1).sum(dim=1)
            assert entropy.isfinite().all(), "Entropy is not finite"
            outdict["entropy"] = entropy
        if return_all_logp:
            outdict["all_logp"] = all_logps
        if return_hidden:
            outdict["hidden"] = hidden
        if return_init_embeds:
            outdict["init_embeds"] = init_embeds
        return outdict

    def _drain_status(self) -> None:
        """The asynchronously read status words of the last training step (async_train_status): wait for their copy — over
        long ago unless called right behind the step — and raise the reference's assertion for any sticky bit in them."""
        if self._pending_status is not None:
            host, ev = self._pending_status
            self._pending_status = None
            ev.synchronize()
            from . import _lib as _l

            _l.raise_for_error_bits(int(host[0]))

    def check_backward_errors(self) -> None:
        """Raise the reference's assertion for any sticky bit the LAST training step's rollout (async_train_status) or
        teacher-forced backward kernel set (a sync). Rollouts do this on their own: the words ride on the next read-back."""
        self._drain_status()
        if self._bwd_err is not None:
            bits = int(self._bwd_err.item())
            self._bwd_err.zero_()
            from . import _lib as _l

            _l.raise_for_error_bits(bits)

    # -- pieces -------------------------------------------------------------------------------------
    def _env_step_state(self, state: dict, action: Tensor, err: Tensor) -> None:
        if self.env_name == "tsp":
            K.tsp_step(action, state["action_mask"], state["first_node"], state["current_node"],
                       state["i"], state["done"], err)
        elif self.env_name == "op":
            K.op_step(action, state["locs"], state["max_length"], state["tour_length"], state["visited"],
                      state["current_node"], state["i"], state["action_mask"], state["done"], err)
        elif self.env_name == "cvrptw":
            K.cvrptw_step(action, state["demand"], state["locs"], state["time_windows"], state["durations"],
                          state["used_capacity"], state["vehicle_capacity"], state["current_time"], state["visited"],
                          state["current_node"], state["action_mask"], state["done"], err)
        elif self.env_name == "pdp":
            K.pdp_step(action, state["available"], state["to_deliver"], state["current_node"], state["i"],
                       state["action_mask"], state["done"], err)
        elif self.env_name == "pctsp":
            K.pctsp_step(action, state["real_prize"], state["cur_total_prize"], state["visited"], state["current_node"],
                         state["i"], state["action_mask"], state["done"], err)
        else:
            K.cvrp_step(action, state["demand"], state["used_capacity"], state["vehicle_capacity"],
                        state["visited"], state["current_node"], state["action_mask"], state["done"], err)

    def _final_td(self, td, state: dict, n_rep: int) -> TensorDict:
        s = max(n_rep, 1)
        b_inst = td["action_mask"].shape[0]

        def rep(x: Tensor) -> Tensor:
            if s == 1:
                return x
            return x.unsqueeze(0).expand(s, *x.shape).reshape(s * x.shape[0], *x.shape[1:])

        out = {"locs": rep(td["locs"]), "action_mask": state["action_mask"], "done": state["done"]}
        if self.env_name == "tsp":
            out.update(first_node=state["first_node"], current_node=state["current_node"],
                       i=state["i"].view(-1, 1))
        elif self.env_name == "op":
            out.update(prize=rep(td["prize"]), max_length=rep(td["max_length"]), current_node=state["current_node"].view(-1, 1),
                       tour_length=state["tour_length"], visited=state["visited"], i=state["i"])
        elif self.env_name == "pdp":
            out.update(available=state["available"], to_deliver=state["to_deliver"],
                       current_node=state["current_node"].view(-1, 1), i=state["i"].view(-1, 1))
        elif self.env_name == "pctsp":
            out.update(real_prize=rep(td["real_prize"]), expected_prize=rep(td["expected_prize"]), penalty=rep(td["penalty"]),
                       prize_required=state["prize_required"], cur_total_prize=state["cur_total_prize"],
                       current_node=state["current_node"], visited=state["visited"], i=state["i"])
        else:
            out.update(demand=rep(td["demand"]), current_node=state["current_node"].view(-1, 1),
                       used_capacity=state["used_capacity"].view(-1, 1),
                       vehicle_capacity=state["vehicle_capacity"].view(-1, 1), visited=state["visited"])
            if self.env_name == "cvrptw":
                out.update(time_windows=rep(td["time_windows"]), durations=rep(td["durations"]),
                           current_time=state["current_time"].view(-1, 1))
        return TensorDict(out, batch_size=[s * b_inst])

    # -- teacher-forced, differentiable re-evaluation (row N1 of SURVEY.md §8f) -------------------
    def evaluate_log_probs(self, td, hidden: Tensor, actions: Tensor, n_rep: int, tanh_clipping: float,
                           temperature: float, mask_logits: bool, skip_first: bool = False, return_full: bool = False):
        """log p(a_t | s_t) for all t at once, with autograd through encoder and decoder weights.

        With the actions known every step's query is known up front, so the T sequential
        single-query attentions of the reference loop become ONE masked [T x N] attention per
        instance (dense, MFMA-friendly). Masks/contexts are replayed with the env-step kernels."""
        dec = self.decoder
        s = max(n_rep, 1)
        hidden = hidden.float()  # encoder_autocast may hand over bf16 activations
        b_inst, n, d = hidden.shape
        b, t_len = actions.shape
        # 16-bit regime on the GPU: the masked glimpse attention of all steps on csrc/am_cross_attn.hip (keys shared by the
        # starts of an instance, the mask as bits from the same replay launch); otherwise torch's SDPA in fp32
        regime = self._encoder_regime()
        glimpse_kernel = (hidden.is_cuda and regime in (torch.bfloat16, torch.float16) and dec.mask_inner and dec.num_heads == 8
                          and d == 128 and self.fused_backward)
        masks, ctx_nodes, extras, mask_bits = self._replay(td, actions, n_rep, mask_bits=glimpse_kernel)
        h = hidden if s == 1 else hidden.unsqueeze(0).expand(s, b_inst, n, d).reshape(b, n, d)
        w_ctx = dec.context_embedding.project_context.weight
        if self.env_name == "tsp":
            first, prev = ctx_nodes  # [B,T] each (t = 0 is the placeholder)
            idx = torch.stack([first, prev], -1).view(b, t_len * 2)
            ctx = h.gather(1, idx[..., None].expand(b, t_len * 2, d)).view(b, t_len, 2 * d)
            placeholder = dec.context_embedding.W_placeholder.expand(b, 1, 2 * d)
            use_ph = extras  # [B,T] bool: step used the placeholder context
            ctx = torch.where(use_ph[..., None], placeholder, ctx)
        else:
            (prev,) = ctx_nodes
            cur = h.gather(1, prev[..., None].expand(b, t_len, d))
            if self.env_name == "pdp":
                ctx = cur
            elif self.env_name == "cvrptw":
                ctx = torch.cat([cur, extras], -1)  # + remaining capacity, current time
            else:
                ctx = torch.cat([cur, extras[..., None]], -1)  # + remaining capacity
        q = F.linear(ctx, w_ctx)
        if dec.use_graph_context:
            g = dec.project_fixed_context(hidden.mean(1))
            g = g if s == 1 else g.unsqueeze(0).expand(s, b_inst, d).reshape(b, d)
            q = q + g[:, None, :]
        kvl_inst = dec.project_node_embeddings(hidden)  # [B_inst, N, 3 d]
        kvl = kvl_inst if s == 1 else kvl_inst.unsqueeze(0).expand(s, b_inst, n, 3 * d).reshape(b, n, 3 * d)
        k_g, v_g, k_l = kvl.chunk(3, dim=-1)
        nh = dec.num_heads
        heads = None
        if glimpse_kernel:
            from . import train_ops

            q16, kv16 = q.to(regime), kvl_inst[..., : 2 * d].to(regime)
            if train_ops.glimpse_attention_usable(q16, kv16, mask_bits):
                heads = train_ops.glimpse_attention(q16, kv16, mask_bits).float()
        if heads is None:
            qh = q.view(b, t_len, nh, d // nh).transpose(1, 2)
            kh = k_g.reshape(b, n, nh, d // nh).transpose(1, 2)
            vh = v_g.reshape(b, n, nh, d // nh).transpose(1, 2)
            attn_mask = masks[:, None, :, :] if dec.mask_inner else None
            heads = F.scaled_dot_product_attention(qh, kh, vh, attn_mask=attn_mask).transpose(1, 2).reshape(b, t_len, d)
        glimpse = dec.pointer.project_out(heads)
        if glimpse_kernel and mask_bits is not None and not return_full:
            # clip, mask, log-softmax and the gather of the given action in one pass each way (csrc/am_logit_logp.hip); the
            # logit keys stay per instance: starts s-major -> [B_inst, s T, 128] against [B_inst, 128, N]
            from . import train_ops

            kl_inst = kvl_inst[..., 2 * d:]
            gl = glimpse if s == 1 else glimpse.view(s, b_inst, t_len, d).transpose(0, 1).reshape(b_inst, s * t_len, d)
            raw = torch.bmm(gl, kl_inst.transpose(1, 2))
            if s > 1:
                raw = raw.view(b_inst, s, t_len, n).transpose(0, 1).reshape(b, t_len, n)
            step_logps = train_ops.logit_logp(raw.float(), mask_bits if mask_logits else None, actions, tanh_clipping, temperature)
            if skip_first:  # multistart: the first action is imposed, its log-prob is 0 (decoding.py:318-323)
                step_logps = torch.cat([torch.zeros_like(step_logps[:, :1]), step_logps[:, 1:]], 1)
            return step_logps
        logits = torch.bmm(glimpse, k_l.transpose(1, 2)) / math.sqrt(d)
        if tanh_clipping > 0:
            logits = torch.tanh(logits) * tanh_clipping
        if mask_logits:
            logits = logits.masked_fill(~masks, float("-inf"))
        logp = F.log_softmax(logits / temperature, dim=-1)
        step_logps = logp.gather(-1, actions[..., None]).squeeze(-1)
        if skip_first:  # multistart: the first action is imposed, its log-prob is 0 (decoding.py:318-323)
            step_logps = torch.cat([torch.zeros_like(step_logps[:, :1]), step_logps[:, 1:]], 1)
        return (step_logps, logp) if return_full else step_logps

    @torch.no_grad()
    def _replay(self, td, actions: Tensor, n_rep: int, mask_bits: bool = False):
        """Per step of the given trajectories: the mask the decoder saw, the context node(s) and the context scalar(s) —
        ONE launch (``rl4co_env_replay``: the env-step device code looped over T on the device; r06 — the T x ~2 launches
        of ``_replay_stepwise`` were 20 of the 46 ms of a CVRP-500 x 64 REINFORCE step)."""
        state = self._initial_state(td, n_rep)
        b = actions.shape[0]
        rem_base = None
        if self.env_name == "op":
            ml0 = state["max_length"][:, 0]
            rem_base = (ml0 if ml0.shape[0] == b else ml0.repeat(b // ml0.shape[0])).contiguous()
        elif self.env_name == "pctsp":
            rem_base = state["prize_required"]
        elif self.env_name in ("cvrp", "cvrptw"):
            rem_base = state["vehicle_capacity"]
        err = K.new_error_word(actions.device)
        r = K.env_replay(self.env_name, state, actions.contiguous(), rem_base, err, mask_bits=mask_bits)
        bits = r.get("mask_bits")
        if self.env_name == "tsp":
            return r["masks"], (r["first"], r["prev"]), r["use_placeholder"], bits
        if self.env_name == "pdp":
            return r["masks"], (r["prev"],), None, bits
        return (r["masks"], (r["prev"],), (r["rem"] if self.env_name != "cvrptw" else torch.stack((r["rem"], r["now"]), -1)),
                bits)

    @torch.no_grad()
    def _replay_stepwise(self, td, actions: Tensor, n_rep: int):
        """``_replay`` as T calls of the env-step kernels (what ``rl4co_env_replay`` loops on the device): the tests'
        cross-check of the one-launch form."""
        state = self._initial_state(td, n_rep)
        b, t_len = actions.shape
        n = state["action_mask"].shape[1]
        device = actions.device
        masks = torch.empty((b, t_len, n), dtype=torch.bool, device=device)
        prev = torch.empty((b, t_len), dtype=torch.int64, device=device)
        err = K.new_error_word(device)
        if self.env_name == "tsp":
            first = torch.empty((b, t_len), dtype=torch.int64, device=device)
            use_ph = torch.empty((b, t_len), dtype=torch.bool, device=device)
        else:
            rem = torch.empty((b, t_len), dtype=torch.float32, device=device)
            now = torch.empty((b, t_len), dtype=torch.float32, device=device) if self.env_name == "cvrptw" else None
        for t in range(t_len):
            masks[:, t] = state["action_mask"]
            prev[:, t] = state["current_node"]
            if self.env_name == "tsp":
                first[:, t] = state["first_node"]
                use_ph[:, t] = state["i"] < 1
            elif self.env_name == "op":
                ml0 = state["max_length"][:, 0]
                rem[:, t] = (ml0 if ml0.shape[0] == b else ml0.repeat(b // ml0.shape[0])) - state["tour_length"]
            elif self.env_name == "pctsp":
                rem[:, t] = torch.clamp(state["prize_required"] - state["cur_total_prize"], min=0)
            elif self.env_name == "pdp":
                pass  # no context scalar
            else:
                rem[:, t] = state["vehicle_capacity"] - state["used_capacity"]
                if now is not None:
                    now[:, t] = state["current_time"]
            self._env_step_state(state, actions[:, t].contiguous(), err)
        if self.env_name == "tsp":
            return masks, (first, prev), use_ph
        return masks, (prev,), (rem if self.env_name != "cvrptw" else torch.stack((rem, now), -1))
