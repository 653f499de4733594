"""Folded decoder cache — the HBM-resident operand of the fused decode kernel.

The reference's ``PrecomputedCache`` (zoo/am/decoder.py:21-40,201-228) stores
``(K_g, V_g, K_l) = project_node_embeddings(h)`` and re-applies three batch-shared weight
matrices at every decode step: ``project_context`` on the gathered ``[h_first ; h_cur]``
(context.py:130-134), ``project_out`` on the glimpse (attention.py:287) — 48 k FMAs per
instance-step that stream ~160 KB of weights through every CU.

MI355X-first restructuring: fold those matrices into per-node rows ONCE per rollout
(dense GEMMs, the MFMA-friendly side of the path), so one decode step is a pure HBM stream
over three ``[N,128]`` planes plus two gathered rows:

    glimpse_key = h Wk^T            glimpse_val = h Wv^T
    logit_key   = h (W_out^T Wl)^T              logits_j = heads . logit_key_j / sqrt(128)
    ctx_first   = h W_ctx[:, :128]^T ; ctx_cur = h W_ctx[:, 128:256]^T        (TSP)
    ctx_cur     = h W_ctx[:, :128]^T ; w_cap = W_ctx[:, 128]                   (CVRP)
    q_bias      = project_fixed_context(mean_j h_j)   (None for POMO)
    q_step0     = W_ctx W_placeholder                  (TSP, context.py:120-128)

Algebraically identical to the reference; rounding differs at the 1e-7 level.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch
from torch import Tensor

EMBED_DIM = 128

# environments that share another one's kernels (same state, masks, embeddings): stochastic PCTSP only
# differs in which generated prize its reset() calls "real" (spctsp/env.py:8-21)
KERNEL_ENV = {"spctsp": "pctsp"}


def canonical_env(env_name: str) -> str:
    return KERNEL_ENV.get(env_name, env_name)


@dataclass
class FoldedCache:
    env_name: str
    kvl: Tensor  # [3, B, N, 128] fp32 or bf16, plane-major: (glimpse_key, glimpse_val, logit_key)
    ctx_first: Tensor | None  # [B, N, 128] fp32 (TSP)
    ctx_cur: Tensor  # [B, N, 128] fp32
    q_bias: Tensor | None  # [B, 128] fp32
    q_step0: Tensor | None  # [128] fp32 (TSP)
    w_cap: Tensor | None  # [128] fp32 (CVRP)
    w_time: Tensor | None = None  # [128] fp32 (CVRPTW: W_ctx[:, 129], the current-time column)
    # "unfolded" parity mode (build_folded_cache(fold=False), TSP / CVRP): plane 2 of `kvl` is the RAW logit key and
    # the three batch-shared matrices are applied per decode step in the reference's association
    unfold: bool = False
    node_embed: Tensor | None = None     # [B, N, 128] fp32 encoder output
    w_ctx_t: Tensor | None = None        # [256 | 129, 128] fp32 project_context.weight^T
    w_out_t: Tensor | None = None        # [128, 128] fp32 project_out.weight^T
    w_placeholder: Tensor | None = None  # [256] fp32 (TSP)

    TENSOR_FIELDS = ("kvl", "ctx_first", "ctx_cur", "q_bias", "q_step0", "w_cap", "w_time", "node_embed", "w_ctx_t",
                     "w_out_t", "w_placeholder")

    def to(self, device) -> "FoldedCache":
        """Same cache with every tensor moved (contiguous) to `device` — tests hand the kernel's bytes to the oracle."""
        kw = {k: (None if getattr(self, k) is None else getattr(self, k).to(device).contiguous()) for k in self.TENSOR_FIELDS}
        return FoldedCache(env_name=self.env_name, unfold=self.unfold, **kw)

    @property
    def num_instances(self) -> int:
        return self.kvl.shape[1]

    @property
    def num_nodes(self) -> int:
        return self.kvl.shape[2]

    def plane(self, i: int) -> Tensor:
        return self.kvl[i]

    @property
    def row_stride(self) -> int:
        return self.kvl.stride(2)

    @property
    def batch_stride(self) -> int:
        return self.kvl.stride(1)


def fold_weights(env_name: str, w_node: Tensor, w_out: Tensor, w_ctx: Tensor) -> list[Tensor]:
    """Per-node projection matrices ``[Wk, Wv, W_out^T Wl, W_ctx blocks...]``, each [128,128]."""
    d = EMBED_DIM
    wk, wv, wl = w_node[:d], w_node[d : 2 * d], w_node[2 * d :]
    wl_folded = w_out.t() @ wl  # logits = heads^T W_out^T (Wl h_j)
    if env_name == "tsp":
        return [wk, wv, wl_folded, w_ctx[:, :d], w_ctx[:, d : 2 * d]]
    if env_name in ("cvrp", "op", "pctsp", "pdp", "cvrptw"):  # current-node embedding (+ one scalar: capacity / remaining length / prize)
        return [wk, wv, wl_folded, w_ctx[:, :d]]
    raise ValueError(f"fused decode supports tsp/cvrp/op, got {env_name!r}")


def _fold_tables_f32(h: Tensor, blocks: list[Tensor], w_fixed: Tensor | None):
    """[h W_i^T for the [128,128] blocks] (fp32 [B,N,128] each) and project_fixed_context(mean_j h_j) on
    ``rl4co_am_fold_tables_f32`` — no library GEMM, no reduction launch."""
    import ctypes as C

    from . import _lib
    from .encoder import pack_weight_f32

    hc = h.contiguous()
    b, n, d = hc.shape
    outs = [torch.empty((b, n, d), dtype=torch.float32, device=h.device) for _ in blocks]
    packed = torch.stack([pack_weight_f32(w) for w in blocks]).contiguous() if blocks else None
    ptrs = (C.c_void_p * max(1, len(outs)))(*[o.data_ptr() for o in outs])
    q_bias = torch.empty((b, d), dtype=torch.float32, device=h.device) if w_fixed is not None else None
    wf = w_fixed.detach().float().contiguous() if w_fixed is not None else None
    st = _lib.lib().rl4co_am_fold_tables_f32(hc.data_ptr(), _lib.dtype_id(hc.dtype), b, n, None if packed is None else packed.data_ptr(),
                                             len(outs), ptrs, None if wf is None else wf.data_ptr(),
                                             None if q_bias is None else q_bias.data_ptr(), torch.cuda.current_stream().cuda_stream)
    _lib.check(st, "rl4co_am_fold_tables_f32")
    return outs, q_bias


def build_folded_cache(
    env_name: str,
    h: Tensor,
    w_node: Tensor,
    w_out: Tensor,
    w_ctx: Tensor,
    w_fixed: Tensor | None,
    w_placeholder: Tensor | None,
    cache_dtype: torch.dtype = torch.float32,
    gemm_dtype: torch.dtype = torch.float32,
    fold: bool = True,
) -> FoldedCache:
    """One ``[B*N,128] x [128,128]`` GEMM per plane, written straight into its plane of the
    plane-major cache (no permute/cast copies), plus a GEMV for the graph context.

    ``gemm_dtype`` is the input type of the three streamed-plane GEMMs (fp32 = the parity
    configuration; bf16 = MFMA rate, the reference's own mixed-precision regime,
    utils/trainer.py:57). The two gathered context tables are always folded in fp32."""
    assert h.dim() == 3 and h.shape[-1] == EMBED_DIM
    d = EMBED_DIM
    b, n, _ = h.shape
    if not fold:
        # reference association (zoo/am/decoder.py:201-228): the cache is project_node_embeddings(h) chunked in three,
        # the graph context project_fixed_context(h.mean(1)); context / output projections stay per-step GEMVs
        if env_name not in ("tsp", "cvrp"):
            raise ValueError("the unfolded parity mode serves tsp / cvrp")
        kvl = torch.empty((3, b, n, d), dtype=cache_dtype, device=h.device)
        h32 = h.reshape(b * n, d).float()
        for i in range(3):
            kvl[i].view(b * n, d).copy_(torch.matmul(h32, w_node.float()[i * d : (i + 1) * d].t()))
        q_bias = torch.matmul(h.float().mean(1), w_fixed.float().t()).contiguous() if w_fixed is not None else None
        return FoldedCache(env_name, kvl, None, None, q_bias, None, None, None, unfold=True,
                           node_embed=h.float().contiguous(), w_ctx_t=w_ctx.float().t().contiguous(),
                           w_out_t=w_out.float().t().contiguous(),
                           w_placeholder=None if w_placeholder is None else w_placeholder.detach().float().contiguous())
    w_blocks = fold_weights(env_name, w_node.float(), w_out.float(), w_ctx.float())
    kvl = torch.empty((3, b, n, d), dtype=cache_dtype, device=h.device)
    h_g = h.reshape(b * n, d).to(gemm_dtype)
    own_gemm = h.is_cuda and gemm_dtype == cache_dtype == torch.bfloat16 and not torch.is_grad_enabled()
    for i in range(3):
        if own_gemm:  # bf16 planes of an inference rollout: the tall-skinny GEMM kernel (csrc/am_train_ops.hip), straight into the plane
            from . import train_ops

            train_ops._gemm(h_g.contiguous(), w_blocks[i].to(torch.bfloat16).contiguous(), out=kvl[i].view(b * n, d))
            continue
        w_t = w_blocks[i].to(gemm_dtype).t()
        if gemm_dtype == cache_dtype:
            torch.matmul(h_g, w_t, out=kvl[i].view(b * n, d))
        else:
            kvl[i].view(b * n, d).copy_(torch.matmul(h_g, w_t))
    if h.is_cuda and not torch.is_grad_enabled() and h.dtype in (torch.float32, torch.bfloat16, torch.float16) and b <= 65535:  # (the entry carries the instance in grid.y)
        # inference: the fp32 side of the fold (context tables, graph context) on the library's own fp32-MFMA kernel
        # (csrc/am_tokens_f32.hip: rl4co_am_fold_tables_f32; the 16-bit embeddings of the token path are widened on load)
        ctx, q_bias = _fold_tables_f32(h, w_blocks[3:], w_fixed)
    else:
        h32 = h.reshape(b * n, d).float()
        ctx = [torch.matmul(h32, w.t()).view(b, n, d) for w in w_blocks[3:]]
        q_bias = None
        if w_fixed is not None:
            q_bias = torch.matmul(h.mean(1, dtype=torch.float32), w_fixed.float().t()).contiguous()
    if env_name == "tsp":
        ctx_first, ctx_cur = ctx
        q_step0 = torch.mv(w_ctx.float(), w_placeholder.float()).contiguous()
        w_cap = None
    else:
        ctx_first, ctx_cur = None, ctx[0]
        q_step0 = None
        w_cap = w_ctx.float()[:, d].contiguous() if w_ctx.shape[1] > d else None  # PDP: no context scalar
    w_time = w_ctx.float()[:, d + 1].contiguous() if w_ctx.shape[1] > d + 1 else None  # CVRPTW: current time
    return FoldedCache(env_name, kvl, ctx_first, ctx_cur, q_bias, q_step0, w_cap, w_time)
