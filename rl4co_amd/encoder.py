"""Host side of the inference encoder + cache-fold kernels.

Packs the policy's own parameters (same module tree / state_dict as the reference) into the fragment order the kernels
stream, folds eval-mode batch norm into a per-channel affine, and launches, by regime and graph size:

    16-bit (bf16 / fp16 autocast)   rl4co_am_encoder (N <= 128: one workgroup per instance, activations never leave the CU)
                                    rl4co_am_encoder_tokens16 (any N: tiles of 128 nodes)            csrc/am_encoder.hip
    fp32 (no autocast)              rl4co_am_encoder_f32 (N <= 128) / rl4co_am_encoder_tokens_f32    csrc/am_encoder_f32.hip,
                                    on v_mfma_f32_16x16x4_f32 — the bit-identical configuration      csrc/am_tokens_f32.hip

Inference only; training goes through ``train_ops`` (autograd around the HIP kernels) or the torch modules.
"""
from __future__ import annotations

import ctypes as C
import os

import torch
from torch import Tensor

from . import _lib
from .cache import EMBED_DIM, FoldedCache, fold_weights

_vp, _i32, _i64 = C.c_void_p, C.c_int32, C.c_int64


class AmEncoderArgs(C.Structure):
    """Mirror of ``struct rl4co_am_encoder_args`` (field order and types must match the header)."""

    _fields_ = [
        ("env", _i32), ("B", _i32), ("N", _i32), ("num_layers", _i32), ("norm", _i32), ("cache_dtype", _i32),
        ("act_dtype", _i32), ("ctx_dtype", _i32),
        ("locs", _vp), ("demand", _vp), ("feature4", _vp), ("feature5", _vp), ("feature6", _vp), ("w_init", _vp), ("b_init", _vp), ("w_depot", _vp), ("b_depot", _vp), ("w_extra", _vp), ("b_extra", _vp),
        ("wqkv_packed", _vp), ("bqkv", _vp), ("wo_packed", _vp), ("bo", _vp), ("n1_scale", _vp), ("n1_shift", _vp),
        ("w1_packed", _vp), ("b1", _vp), ("w2_packed", _vp), ("b2", _vp), ("n2_scale", _vp), ("n2_shift", _vp),
        ("wfold_packed", _vp), ("w_fixed", _vp),
        ("kvl", _vp), ("kvl_plane_stride", _i64), ("kvl_batch_stride", _i64),
        ("ctx_first", _vp), ("ctx_cur", _vp), ("q_bias", _vp), ("hidden", _vp),
    ]


def pack_weight(w: Tensor, dtype: torch.dtype = torch.bfloat16) -> Tensor:
    """nn.Linear weight [out,in] -> 16-bit [out/32, in/16, 64, 8] in MFMA fragment order:
    lane = 32*hi + row, element s = W[32*tile + row][16*kstep + 8*hi + s]. ``dtype``: the element type of the
    kernel's MFMA operands (bfloat16, or float16 for the reference's default "16-mixed" regime)."""
    out_f, in_f = w.shape
    assert out_f % 32 == 0 and in_f % 16 == 0, (out_f, in_f)
    t = w.detach().to(dtype).view(out_f // 32, 32, in_f // 16, 2, 8)  # [tile,row,ks,hi,s]
    return t.permute(0, 2, 3, 1, 4).contiguous().view(out_f // 32, in_f // 16, 64, 8)


def pack_weight_f32(w: Tensor) -> Tensor:
    """nn.Linear weight [out,in] -> fp32 [out/16, in/16, 64, 4] in the fragment order of ``v_mfma_f32_16x16x4_f32``
    (csrc/am_encoder_f32.hip): lane = 16*g + c, element s = W[16*tile + c][16*chunk + 4*g + s]."""
    out_f, in_f = w.shape
    assert out_f % 16 == 0 and in_f % 16 == 0, (out_f, in_f)
    t = w.detach().float().view(out_f // 16, 16, in_f // 16, 4, 4)  # [tile, c, chunk, g, s]
    return t.permute(0, 2, 3, 1, 4).contiguous().view(out_f // 16, in_f // 16, 64, 4)


def _norm_affine_f32(norm_module, device=None) -> tuple[Tensor, Tensor, int]:
    """(alpha, beta, kind) the way ATen's CPU batch-norm kernel forms them (native/cpu/batch_norm_kernel.cpp):
    invstd = 1 / sqrt(var + eps), alpha = invstd * weight, beta = bias - mean * alpha — eval-mode batch norm; instance norm
    hands over (gamma, beta) and the kernel builds the same pair from the instance's own statistics."""
    n = norm_module.normalizer
    if norm_module.kind == "layer":  # nn/ops.py:48-51: whole-instance statistics, no affine — the kernel needs no table
        return torch.ones(EMBED_DIM, device=device), torch.zeros(EMBED_DIM, device=device), 2
    if norm_module.kind == "batch":
        invstd = 1.0 / torch.sqrt(n.running_var.float() + n.eps)
        alpha = invstd * n.weight.detach().float()
        return alpha, n.bias.detach().float() - n.running_mean.float() * alpha, 0
    if norm_module.kind == "instance":
        return n.weight.detach().float(), n.bias.detach().float(), 1
    raise NotImplementedError("fused encoder supports batch (eval), instance and layer normalisation")


def _norm_affine(norm_module, device=None) -> tuple[Tensor, Tensor, int]:
    """(scale, shift, kind): eval-mode batch norm folded to an affine; instance norm -> gamma/beta; layer norm (no affine)
    -> (1, 0): its shift slot carries the bias of the GEMM in front of it (see ``PackedEncoder.refresh``)."""
    n = norm_module.normalizer
    if norm_module.kind == "layer":
        return torch.ones(EMBED_DIM, device=device), torch.zeros(EMBED_DIM, device=device), 2
    if norm_module.kind == "batch":
        scale = n.weight.detach().float() * torch.rsqrt(n.running_var.float() + n.eps)
        shift = n.bias.detach().float() - n.running_mean.float() * scale
        return scale, shift, 0
    if norm_module.kind == "instance":
        return n.weight.detach().float(), n.bias.detach().float(), 1
    raise NotImplementedError("fused encoder supports batch (eval), instance and layer normalisation")


class PackedEncoder:
    """Device-resident packed parameters of one policy, rebuilt when any parameter changes."""

    def __init__(self, policy):
        self.policy = policy
        self.version = None
        self.t: dict[str, Tensor] = {}
        self._tensors: list[Tensor] | None = None
        self.act_dtype = torch.bfloat16  # element type the weights are packed in (a regime switch re-packs)

    def _current_version(self):
        """Cheap change detector on the path of every rollout (it sits between the previous rollout's
        sync and this one's first launch): the module tree is walked once, afterwards only the version
        counters and storage addresses of the same Parameter / buffer objects are read. Modules keep
        their Parameter objects across ``.to()`` / ``load_state_dict`` / optimizer steps (the storage
        address or the version counter moves instead); ``refresh(force=True)`` re-walks the tree after
        structural surgery (a parameter object replaced by assignment)."""
        pol = self.policy
        tensors = self._tensors
        if tensors is None:
            tensors = self._tensors = list(pol.parameters()) + list(pol.buffers())
        # inference tensors (a model built or loaded under torch.inference_mode) carry no version counter:
        # for them only the storage address is available, so in-place updates of such weights need refresh(force=True)
        return (tuple(0 if p.is_inference() else p._version for p in tensors), tuple(p.data_ptr() for p in tensors),
                pol.training, self.act_dtype)

    def refresh(self, force: bool = False, act_dtype: torch.dtype | None = None) -> dict[str, Tensor]:
        if force:
            self._tensors, self.version = None, None
        if act_dtype is not None:
            if act_dtype not in (torch.bfloat16, torch.float16, torch.float32):
                raise TypeError(f"the fused encoder computes in bfloat16, float16 or float32, not {act_dtype}")
            self.act_dtype = act_dtype
        ver = self._current_version()
        if ver == self.version:
            return self.t
        pol = self.policy
        enc, dec = pol.encoder, pol.decoder
        layers = list(enc.net.layers)
        f32 = lambda x: x.detach().float().contiguous()  # noqa: E731
        exact = self.act_dtype == torch.float32  # csrc/am_encoder_f32.hip: fp32 MFMA, the reference's own arithmetic order
        pack_weight = (pack_weight_f32 if exact else lambda w: globals()["pack_weight"](w, self.act_dtype))  # noqa: E731
        t: dict[str, Tensor] = {}
        ie = enc.init_embedding
        if pol.env_name == "pdp":
            t["w_init"], t["b_init"] = f32(ie.init_embed_pick.weight), f32(ie.init_embed_pick.bias)
            t["w_extra"], t["b_extra"] = f32(ie.init_embed_delivery.weight), f32(ie.init_embed_delivery.bias)
        else:
            t["w_init"], t["b_init"] = f32(ie.init_embed.weight), f32(ie.init_embed.bias)
        if pol.env_name != "tsp":
            t["w_depot"], t["b_depot"] = f32(ie.init_embed_depot.weight), f32(ie.init_embed_depot.bias)
        # the query rows of Wqkv (and their bias) carry head_dim^-1/2 * log2(e): the kernel's softmax is exp2(q . k)
        # (fp32 kernel: only the power of two — exact — and exp(s - max) as exp2((s - max) * log2 e) in the kernel)
        qscale = torch.ones(3 * EMBED_DIM, 1, device=layers[0][0].module.Wqkv.weight.device)
        qscale[:EMBED_DIM] = 0.25 if exact else 0.25 * 1.4426950408889634
        t["wqkv"] = torch.stack([pack_weight(l[0].module.Wqkv.weight.detach().float() * qscale) for l in layers]).contiguous()
        t["bqkv"] = torch.stack([f32(l[0].module.Wqkv.bias.detach().float() * qscale[:, 0]) for l in layers]).contiguous()
        t["wo"] = torch.stack([pack_weight(l[0].module.out_proj.weight) for l in layers]).contiguous()
        t["bo"] = torch.stack([f32(l[0].module.out_proj.bias) for l in layers]).contiguous()
        t["w1"] = torch.stack([pack_weight(l[2].module.lins[0].weight) for l in layers]).contiguous()
        t["b1"] = torch.stack([f32(l[2].module.lins[0].bias) for l in layers]).contiguous()
        t["w2"] = torch.stack([pack_weight(l[2].module.lins[1].weight) for l in layers]).contiguous()
        t["b2"] = torch.stack([f32(l[2].module.lins[1].bias) for l in layers]).contiguous()
        kinds = set()
        # the bias of the GEMM in front of a norm (out_proj before norm1, the MLP's second linear before norm2) is a
        # per-channel constant added to every token: under eval-mode batch norm it moves the affine's shift by
        # bias * scale; under instance norm it cancels in the per-channel mean over the nodes. The kernel adds neither
        # (rl4co_am_encoder_args.bo / b2 are still passed for reference and ignored). Under layer norm (one mean over the
        # whole instance) it does NOT cancel: the 16-bit kernel takes it from the shift slot and adds it before the statistics.
        dev = layers[0][0].module.Wqkv.weight.device
        for name, idx, bias_of in (("n1", 1, lambda l: l[0].module.out_proj.bias), ("n2", 3, lambda l: l[2].module.lins[1].bias)):
            sc, sh = [], []
            for l in layers:
                a, b, k = (_norm_affine_f32 if exact else _norm_affine)(l[idx], dev)
                if k == 0 and not exact:  # (the fp32 kernel adds bo / b2 itself, in the reference's order)
                    b = b + bias_of(l).detach().float() * a
                elif k == 2 and not exact:
                    b = bias_of(l).detach().float()
                sc.append(a), sh.append(b), kinds.add(k)
            t[f"{name}_scale"], t[f"{name}_shift"] = torch.stack(sc).contiguous(), torch.stack(sh).contiguous()
        assert len(kinds) == 1
        self.norm_kind = kinds.pop()
        w_ctx = dec.context_embedding.project_context.weight.detach().float()
        blocks = fold_weights(pol.env_name, dec.project_node_embeddings.weight.detach().float(),
                              dec.pointer.project_out.weight.detach().float(), w_ctx)
        t["wfold"] = torch.stack([pack_weight(b) for b in blocks]).contiguous()
        if exact and pol.env_name in ("tsp", "cvrp"):
            # fold=False (the reference's own association of the decoder, cache.py): the three raw planes K_g, V_g, K_l
            w_node = dec.project_node_embeddings.weight.detach().float()
            t["wnode"] = torch.stack([pack_weight(w_node[i * EMBED_DIM:(i + 1) * EMBED_DIM]) for i in range(3)]).contiguous()
        t["w_fixed"] = f32(dec.project_fixed_context.weight) if dec.use_graph_context else None
        if pol.env_name == "tsp":
            t["q_step0"] = torch.mv(w_ctx, dec.context_embedding.W_placeholder.detach().float()).contiguous()
            t["w_cap"] = None
        else:
            t["q_step0"] = None
            t["w_cap"] = w_ctx[:, EMBED_DIM].contiguous() if w_ctx.shape[1] > EMBED_DIM else None  # PDP: no scalar
        t["w_time"] = w_ctx[:, EMBED_DIM + 1].contiguous() if w_ctx.shape[1] > EMBED_DIM + 1 else None  # CVRPTW
        self.num_layers = len(layers)
        self.t, self.version = t, ver
        return t

    def _stack_shape_ok(self) -> bool:
        """Every kernel (fused, token-tile, fp32) has the AttentionModel's layer shape compiled in: 8 heads over 128
        channels, a 128 -> 512 -> 128 MLP with biases (csrc/am_encoder.hip: kD, kFF). The constructor accepts other
        ``feedforward_hidden`` / ``num_heads`` values (zoo/am/encoder.py:40-57); those stacks are NOT packed — packed with
        the wrong shape the kernels would index ``layer * 512 * 128`` into a smaller weight block — and the caller's torch
        or per-op path serves them."""
        for layer in self.policy.encoder.net.layers:
            attn, ffn = layer[0].module, layer[2].module
            lins = getattr(ffn, "lins", None)
            if lins is None or len(lins) != 2 or getattr(attn, "num_heads", None) != 8:
                return False
            if tuple(attn.Wqkv.weight.shape) != (3 * EMBED_DIM, EMBED_DIM) or tuple(attn.out_proj.weight.shape) != (EMBED_DIM, EMBED_DIM):
                return False
            if tuple(lins[0].weight.shape) != (4 * EMBED_DIM, EMBED_DIM) or tuple(lins[1].weight.shape) != (EMBED_DIM, 4 * EMBED_DIM):
                return False
            if any(lin.bias is None for lin in (attn.Wqkv, attn.out_proj, lins[0], lins[1])):
                return False
        return True

    def supported(self, td, act_dtype: torch.dtype | None = None) -> bool:
        """``act_dtype=torch.float32``: the exact-fp32 kernels — the fused one up to 128 nodes, the token-tile launches
        (csrc/am_tokens_f32.hip) for any graph size; every normalisation kind of nn/ops.py:30-54 (batch in eval mode)."""
        pol = self.policy
        n = td["action_mask"].shape[-1]
        kind = pol.encoder.net.layers[0][1].kind
        if kind == "batch" and pol.training:  # batch statistics couple instances: torch path
            return False
        if not self._stack_shape_ok():
            return False
        if n > _lib.lib().rl4co_am_encoder_max_nodes():
            if td["locs"].shape[0] > 65535:  # the token-tile launches carry the instance in grid.y
                return False
            # token-tile launches (fp32: csrc/am_tokens_f32.hip; 16-bit: the token kernels of csrc/am_encoder.hip). Instance /
            # layer statistics couple all nodes of an instance: the layer's halves then stop before their norm and an apply
            # kernel normalises with the tiles' combined statistics. The staged features must fit the LDS
            return kind in ("batch", "instance", "layer") and td["locs"].is_cuda and 6 * n * 4 + 36 * 1024 <= 80 * 1024
        return kind in ("batch", "instance", "layer") and td["locs"].is_cuda

    def encode(self, td, cache_dtype: torch.dtype, want_hidden: bool = False,
               act_dtype: torch.dtype | None = None, fold: bool = True, tokens: bool | None = None,
               init_embeds_out: Tensor | None = None) -> tuple[FoldedCache, Tensor | None]:
        """``act_dtype``: bfloat16 / float16 = the autocast regime the encoder is asked to compute in; the planes are
        written as ``cache_dtype`` = float32 or that same 16-bit type. float32 = the exact-fp32 kernel
        (``rl4co_am_encoder_f32``; planes in any of the three types). ``fold=False`` (fp32 kernel, tsp / cvrp): the
        reference's own association of the decoder — raw K_g / V_g / K_l planes and the node embeddings, no context tables.
        ``tokens``: force (True) / forbid (False) the token-tile launches of the fp32 encoder (default: beyond 128 nodes).
        ``init_embeds_out`` ([B, N, 128] of ``act_dtype``): also filled with the init embeddings (``return_init_embeds``) by one
        more launch of the routine the encoder kernels run internally (``rl4co_am_encoder_init_embeds16`` / ``_f32``)."""
        t = self.refresh(act_dtype=act_dtype)
        exact = self.act_dtype == torch.float32
        if exact:
            if cache_dtype not in (torch.float32, torch.bfloat16, torch.float16):
                raise TypeError(f"cache planes must be float32, bfloat16 or float16, got {cache_dtype}")
        elif cache_dtype not in (torch.float32, self.act_dtype):
            raise TypeError(f"cache planes must be float32 or the encoder's {self.act_dtype}, got {cache_dtype}")
        if not fold and not (exact and "wnode" in t):
            raise NotImplementedError("fold=False is served by the fp32 encoder kernel for tsp / cvrp")
        pol = self.policy
        locs = td["locs"]
        if locs.dtype != torch.float32 or not locs.is_contiguous():
            locs = locs.float().contiguous()
        b, n, _ = locs.shape
        dev = locs.device
        d = EMBED_DIM
        kvl = torch.empty((3, b, n, d), dtype=cache_dtype, device=dev)
        if tokens is None:
            tokens = n > _lib.lib().rl4co_am_encoder_max_nodes()
        # (r06) 16-bit regime, fused kernel, 16-bit planes: the context tables leave in the activations' type too — the decode
        # kernels widen the rows on load (FoldedCache.ctx_* may then be 16-bit; kernels._ctx_table). The fold is bound by its
        # HBM writes: 179 -> 128 KB per instance at TSP-100. RL4CO_CTX_FP32=1 keeps the fp32 tables of r05 (A/B timing).
        ctx_dt = torch.float32
        if not exact and not tokens and cache_dtype == self.act_dtype and not os.environ.get("RL4CO_CTX_FP32"):
            ctx_dt = self.act_dtype
        ctx_cur = torch.empty((b, n, d), dtype=ctx_dt, device=dev) if fold else None
        ctx_first = torch.empty((b, n, d), dtype=ctx_dt, device=dev) if (fold and pol.env_name == "tsp") else None
        q_bias = torch.empty((b, d), dtype=torch.float32, device=dev) if t["w_fixed"] is not None else None
        hidden = torch.empty((b, n, d), dtype=torch.float32, device=dev) if (want_hidden or not fold) else None
        a = AmEncoderArgs()
        a.env = {"tsp": _lib.ENV_TSP, "pdp": _lib.ENV_PDP}.get(pol.env_name, _lib.ENV_CVRP)
        a.B, a.N, a.num_layers, a.norm = b, n, self.num_layers, self.norm_kind
        a.cache_dtype, a.act_dtype = _lib.dtype_id(cache_dtype), _lib.dtype_id(self.act_dtype)
        a.ctx_dtype = _lib.dtype_id(ctx_dt) if ctx_dt != torch.float32 else 0
        a.locs = locs.data_ptr()
        ptr = lambda x: None if x is None else x.data_ptr()  # noqa: E731
        if pol.env_name in ("cvrp", "op", "pctsp", "cvrptw"):
            # OP embeds the customers' prize where CVRP embeds their demand (init.py:115-136, 254-280);
            # PCTSP the expected prize and, as a fourth feature, the penalty (init.py:283-312)
            third = {"cvrp": "demand", "op": "prize", "pctsp": "expected_prize", "cvrptw": "demand"}[pol.env_name]
            third = td[third][..., 1:] if pol.env_name == "op" else td[third]
            demand = third.float().contiguous()
            a.demand, a.w_depot, a.b_depot = demand.data_ptr(), ptr(t["w_depot"]), ptr(t["b_depot"])
            if pol.env_name == "pctsp":
                penalty = td["penalty"][..., 1:].float().contiguous()
                a.feature4 = penalty.data_ptr()
            if pol.env_name == "cvrptw":  # init.py:139-153: + tw start, tw end, service time
                tw0 = td["time_windows"][..., 1:, 0].float().contiguous()
                tw1 = td["time_windows"][..., 1:, 1].float().contiguous()
                dur = td["durations"][..., 1:].float().contiguous()
                a.feature4, a.feature5, a.feature6 = tw0.data_ptr(), tw1.data_ptr(), dur.data_ptr()
        if pol.env_name == "pdp":
            a.w_depot, a.b_depot = ptr(t["w_depot"]), ptr(t["b_depot"])
            a.w_extra, a.b_extra = ptr(t["w_extra"]), ptr(t["b_extra"])
        a.w_init, a.b_init = ptr(t["w_init"]), ptr(t["b_init"])
        a.wqkv_packed, a.bqkv, a.wo_packed, a.bo = ptr(t["wqkv"]), ptr(t["bqkv"]), ptr(t["wo"]), ptr(t["bo"])
        a.n1_scale, a.n1_shift, a.n2_scale, a.n2_shift = (ptr(t[k]) for k in ("n1_scale", "n1_shift", "n2_scale", "n2_shift"))
        a.w1_packed, a.b1, a.w2_packed, a.b2 = ptr(t["w1"]), ptr(t["b1"]), ptr(t["w2"]), ptr(t["b2"])
        a.wfold_packed, a.w_fixed = ptr(t["wfold"] if fold else t["wnode"]), ptr(t["w_fixed"])
        a.kvl, a.kvl_plane_stride, a.kvl_batch_stride = kvl.data_ptr(), kvl.stride(0), kvl.stride(1)
        a.ctx_first, a.ctx_cur, a.q_bias, a.hidden = ptr(ctx_first), ptr(ctx_cur), ptr(q_bias), ptr(hidden)
        if tokens:
            entry = "rl4co_am_encoder_tokens_f32" if exact else "rl4co_am_encoder_tokens16"
            need = getattr(_lib.lib(), entry + "_workspace")(b, n)
            ws = torch.empty(need, dtype=torch.uint8, device=dev)  # (the caching allocator hands the same block back every rollout)
            st = getattr(_lib.lib(), entry)(C.byref(a), ws.data_ptr(), need, torch.cuda.current_stream().cuda_stream)
        else:
            entry = "rl4co_am_encoder_f32" if exact else "rl4co_am_encoder"
            st = getattr(_lib.lib(), entry)(C.byref(a), torch.cuda.current_stream().cuda_stream)
        _lib.check(st, entry)
        if init_embeds_out is not None:
            if (init_embeds_out.shape != (b, n, d) or init_embeds_out.dtype != self.act_dtype or not init_embeds_out.is_contiguous()
                    or init_embeds_out.device != dev):
                raise ValueError("init_embeds_out must be a contiguous [B, N, 128] tensor of the encoder's activation type")
            entry = "rl4co_am_encoder_init_embeds_f32" if exact else "rl4co_am_encoder_init_embeds16"
            _lib.check(getattr(_lib.lib(), entry)(C.byref(a), init_embeds_out.data_ptr(), torch.cuda.current_stream().cuda_stream), entry)
        if not fold:
            dec = pol.decoder
            ph = getattr(dec.context_embedding, "W_placeholder", None)
            cache = FoldedCache(pol.env_name, kvl, None, None, q_bias, None, None, None, unfold=True, node_embed=hidden,
                                w_ctx_t=dec.context_embedding.project_context.weight.detach().float().t().contiguous(),
                                w_out_t=dec.pointer.project_out.weight.detach().float().t().contiguous(),
                                w_placeholder=None if ph is None else ph.detach().float().contiguous())
            return cache, hidden
        cache = FoldedCache(pol.env_name, kvl, ctx_first, ctx_cur, q_bias, t["q_step0"], t["w_cap"], t["w_time"])
        return cache, hidden
