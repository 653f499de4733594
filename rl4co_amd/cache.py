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


@dataclass
class FoldedCache:
    env_name: str
    kvl: Tensor  # [B, 3, N, 128] fp32 or bf16: planes (glimpse_key, glimpse_val, logit_key)
    ctx_first: Tensor | None  # [B, N, 128] fp32 (TSP)
    ctx_cur: Tensor  # [B, N, 128] fp32
    q_bias: Tensor | None  # [B, 128] fp32
    q_step0: Tensor | None  # [128] fp32 (TSP)
    w_cap: Tensor | None  # [128] fp32 (CVRP)

    @property
    def num_instances(self) -> int:
        return self.kvl.shape[0]

    @property
    def num_nodes(self) -> int:
        return self.kvl.shape[2]

    def plane(self, i: int) -> Tensor:
        return self.kvl[:, i]

    @property
    def row_stride(self) -> int:
        return self.kvl.stride(2)

    @property
    def batch_stride(self) -> int:
        return self.kvl.stride(0)


def fold_weights(env_name: str, w_node: Tensor, w_out: Tensor, w_ctx: Tensor) -> Tensor:
    """Stack the per-node projection matrices: ``[Wk; Wv; W_out^T Wl; W_ctx blocks]`` -> [R,128]."""
    d = EMBED_DIM
    wk, wv, wl = w_node[:d], w_node[d : 2 * d], w_node[2 * d :]
    wl_folded = w_out.t() @ wl  # logits = heads^T W_out^T (Wl h_j)
    if env_name == "tsp":
        blocks = [wk, wv, wl_folded, w_ctx[:, :d], w_ctx[:, d : 2 * d]]
    elif env_name == "cvrp":
        blocks = [wk, wv, wl_folded, w_ctx[:, :d]]
    else:
        raise ValueError(f"fused decode supports tsp/cvrp, got {env_name!r}")
    return torch.cat(blocks, 0)


def build_folded_cache(
    env_name: str,
    h: Tensor,
    w_node: Tensor,
    w_out: Tensor,
    w_ctx: Tensor,
    w_fixed: Tensor | None,
    w_placeholder: Tensor | None,
    cache_dtype: torch.dtype = torch.float32,
) -> FoldedCache:
    """One GEMM ``[B*N,128] x [128,R]`` + a GEMV for the graph context; runs in fp32."""
    assert h.dim() == 3 and h.shape[-1] == EMBED_DIM
    d = EMBED_DIM
    h = h.float()
    w_all = fold_weights(env_name, w_node.float(), w_out.float(), w_ctx.float())
    proj = torch.matmul(h, w_all.t())  # [B, N, R]
    b, n, _ = proj.shape
    # planar [B,3,N,128]: each decode pass streams one contiguous N*128 plane per instance
    kvl = proj[..., : 3 * d].reshape(b, n, 3, d).permute(0, 2, 1, 3).to(cache_dtype).contiguous()
    q_bias = None
    if w_fixed is not None:
        q_bias = torch.matmul(h.mean(1), w_fixed.float().t()).contiguous()
    if env_name == "tsp":
        ctx_first = proj[..., 3 * d : 4 * d].contiguous()
        ctx_cur = proj[..., 4 * d : 5 * d].contiguous()
        q_step0 = torch.mv(w_ctx.float(), w_placeholder.float()).contiguous()
        w_cap = None
    else:
        ctx_first = None
        ctx_cur = proj[..., 3 * d : 4 * d].contiguous()
        q_step0 = None
        w_cap = w_ctx.float()[:, d].contiguous()
    return FoldedCache(env_name, kvl, ctx_first, ctx_cur, q_bias, q_step0, w_cap)
