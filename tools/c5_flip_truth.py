"""GPU box: the C5 greedy fp32 flips (product tour != the reference's CPU fp32 tour, trained weights, CVRP-500 x 1024) judged by an
fp64 evaluation of the same policy at the first divergent step:

* margin64 = logp64[reference's action] - logp64[product's action] at the shared state: > 0 means the fp64 policy sides with the
  reference (the product's arithmetic flipped a near-tie), < 0 that it sides with the product (the reference's CPU run flipped it);
* the encoders' distance to the fp64 embeddings: product fp32 token-tile kernels, torch's fp32 GPU encoder, torch's fp32 CPU encoder.

    python tools/c5_flip_truth.py [torch]        ('torch': the product with fused_encoder=False, i.e. torch's fp32 GPU encoder)
"""
import math
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from trained_parity import CONFIGS, TrainedCase, _pad  # noqa: E402

dev = "cuda"
case = TrainedCase(sys.argv[2] if len(sys.argv) > 2 else "t5_cvrp500_b1024_greedy")
use_torch_encoder = len(sys.argv) > 1 and sys.argv[1] == "torch"
kw = dict(CONFIGS["fp32"])
if use_torch_encoder:
    kw["fused_encoder"] = False
pol, env, data = case.policy(dev, **kw), case.env(dev), case.instances(dev)
ref = case.actions.to(dev)
with torch.inference_mode():
    out = pol(env.reset(data.clone()), env, phase="test", decode_type="greedy", return_hidden=True)
acts = out["actions"]
t = max(acts.shape[1], ref.shape[1])
a, r = _pad(acts, t), _pad(ref, t)
same = (a == r).all(1)
rows = (~same).nonzero().flatten()
first = (a[rows] == r[rows]).long().cumprod(1).sum(1)
print(f"{case.name}: {int(same.sum())} / {a.shape[0]} tours identical ({'torch fp32 GPU encoder' if use_torch_encoder else 'product fp32 kernels'}); "
      f"flipped rows {rows.tolist()} at steps {first.tolist()}")

# ---- fp64 policy: encoder + dense teacher-forced evaluation in double, along the reference's tours of the flipped rows ----
pol64 = case.policy(dev, cache_dtype=torch.float32, fused_encoder=False).double()
td = env.reset(data.clone())
td64 = td.clone()
for k in ("locs", "demand", "vehicle_capacity"):
    if k in td64.keys():
        td64[k] = td64[k].double()
with torch.inference_mode():
    h64, _ = pol64.encoder(td64)
    h32_torch, _ = case.policy(dev, cache_dtype=torch.float32, fused_encoder=False).encoder(td)
    pol_cpu = case.policy("cpu", cache_dtype=torch.float32, fused_encoder=False)
    h32_cpu, _ = pol_cpu.encoder(td.to("cpu")[slice(0, 64)])  # (the state reset on the device: the product has no CPU env kernels)
rel = lambda x, y: float((x.double() - y).norm() / y.norm())  # noqa: E731
print(f"encoder vs fp64: product run {rel(out['hidden'], h64):.3e}   torch fp32 GPU {rel(h32_torch, h64):.3e}"
      + (f"   torch fp32 CPU (64 instances) {rel(h32_cpu.to(dev), h64[:64]):.3e}" if h32_cpu is not None else ""))

if rows.numel():
    dec = pol64.decoder
    sub = rows
    hs = h64[sub]
    b, n, d = hs.shape
    with torch.no_grad():
        masks, ctx_nodes, extras = pol._replay(env.reset(data.clone()[sub.tolist()]), ref[sub], 0)  # discrete state replay on the env kernels (fp32 state)
    t_len = ref.shape[1]
    (prev,) = ctx_nodes
    cur = hs.gather(1, prev[..., None].expand(b, t_len, d))
    ctx = torch.cat([cur, extras[..., None].double()], -1)
    q = F.linear(ctx, dec.context_embedding.project_context.weight)
    if dec.use_graph_context:
        q = q + dec.project_fixed_context(hs.mean(1))[:, None, :]
    k_g, v_g, k_l = dec.project_node_embeddings(hs).chunk(3, dim=-1)
    nh = dec.num_heads
    qh = q.view(b, t_len, nh, d // nh).transpose(1, 2)
    kh = k_g.reshape(b, n, nh, d // nh).transpose(1, 2)
    vh = v_g.reshape(b, n, nh, d // nh).transpose(1, 2)
    sc = qh @ kh.transpose(-1, -2) / math.sqrt(d // nh)
    if dec.mask_inner:
        sc = sc.masked_fill(~masks[:, None, :, :], float("-inf"))
    heads = torch.softmax(sc, -1) @ vh
    glimpse = dec.pointer.project_out(heads.transpose(1, 2).reshape(b, t_len, d))
    logits = torch.bmm(glimpse, k_l.transpose(1, 2)) / math.sqrt(d)
    logits = torch.tanh(logits) * pol.tanh_clipping
    logits = logits.masked_fill(~masks, float("-inf"))
    lp = F.log_softmax(logits / pol.temperature, -1)
    i = torch.arange(b, device=dev)
    f = first.clamp(max=t_len - 1)
    at = lp[i, f]
    m_ref = at.gather(-1, r[rows, f][:, None]).squeeze(-1)
    m_ours = at.gather(-1, a[rows, f][:, None]).squeeze(-1)
    margin = (m_ref - m_ours)
    for j in range(b):
        print(f"  row {int(rows[j]):4d} step {int(f[j]):3d}: reference -> node {int(r[rows[j], f[j]]):3d}, product -> node {int(a[rows[j], f[j]]):3d}, "
              f"fp64 margin (reference - product) {float(margin[j]):+.3e}  => fp64 sides with the {'reference' if margin[j] > 0 else 'product'}")
    print(f"fp64 sides with the reference in {int((margin > 0).sum())} of {b} flips")
