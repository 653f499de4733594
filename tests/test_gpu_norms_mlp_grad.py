"""GPU (r04): every normalisation of nn/ops.py:30-54 on the kernels, and the one-launch MLP input gradient.

Normalization("layer") of the reference (nn/ops.py:48-51 — ONE mean and ONE unbiased variance over all N x 128 values of an
instance, eps 1e-5, no affine):
* inference: the fused fp32 encoder (csrc/am_encoder_f32.hip, LAYER instantiation) against the float64 evaluation of the
  same modules, tolerance 3e-6 relative Frobenius error like the other fp32-encoder tests; the fused 16-bit encoder
  (csrc/am_encoder.hip) at its 16-bit tolerances and no worse than 2.5 x torch's own autocast path;
* training: skip + layer norm forward / backward (csrc/am_train_ops.hip: rl4co_skip_lnorm_*) against torch autograd in
  fp32 on the same 16-bit inputs, and a whole encoder layer (sub-block autograd nodes) against the torch modules.
Instance / layer norm beyond 128 nodes on the token tiles (split sub-blocks + statistics apply kernel), fp32 and 16-bit.
rl4co_mlp_input_grad (csrc/am_encoder.hip: tok16_mlp_bwd_kernel) against the two GEMM launches it replaces.
Floating point => tolerance tests; every tolerance is written at its assertion.
"""
import pytest
import torch

from tests.test_gpu_encoder import _rel
from tests.test_gpu_encoder_f32 import _double_reference

pytestmark = pytest.mark.gpu


def _policy_and_td(env_name, num_loc, batch, **kw):
    from rl4co_amd.envs import get_env
    from rl4co_amd.policy import AttentionModelPolicy

    torch.manual_seed(0)
    pol = AttentionModelPolicy(env_name, normalization="layer", **kw).cuda().eval()
    with torch.no_grad():  # biases that matter: under layer norm the bias in front of the norm does not cancel
        for layer in pol.encoder.net.layers:
            layer[0].module.out_proj.bias.normal_(0.0, 0.3)
            layer[2].module.lins[1].bias.normal_(0.0, 0.3)
    env = get_env(env_name, generator_params=dict(num_loc=num_loc, device="cuda"), device="cuda", check_solution=True)
    torch.manual_seed(3)
    return pol, env, env.reset(env.generator(batch_size=[batch]))


@pytest.mark.parametrize("env_name,num_loc,batch", [("tsp", 100, 64), ("cvrp", 100, 48), ("tsp", 20, 32), ("cvrp", 127, 8), ("pdp", 20, 16),
                                                    ("tsp", 7, 5)])
def test_fp32_fused_encoder_layer_norm_matches_float64(env_name, num_loc, batch):
    pol, env, td = _policy_and_td(env_name, num_loc, batch)
    pe = pol._packed_encoder()
    assert pe.supported(td, torch.float32)
    with torch.inference_mode():
        cache, hidden = pe.encode(td, torch.float32, want_hidden=True, act_dtype=torch.float32)
        h32, _ = pol.encoder(td)
    torch.cuda.synchronize()
    h64, c64 = _double_reference(pol, td)
    for nm, got, exact in (("hidden", hidden, h64), ("plane0", cache.kvl[0], c64.kvl[0]), ("plane2", cache.kvl[2], c64.kvl[2]),
                           ("ctx_cur", cache.ctx_cur, c64.ctx_cur), ("q_bias", cache.q_bias, c64.q_bias)):
        assert torch.isfinite(got).all(), nm
        e_k = float((got.double() - exact).norm() / exact.norm())
        assert e_k <= 3e-6, f"{nm}: kernel rel err {e_k:.3e}"  # fp32 operands and accumulation: only the order of sums differs
    e_t = float((h32.double() - h64).norm() / h64.norm())
    e_k = float((hidden.double() - h64).norm() / h64.norm())
    assert e_k <= 3 * e_t + 2e-7, (e_k, e_t)  # no worse than torch's own fp32 GPU path
    # and the whole fp32 rollout runs without the torch encoder
    pol.encoder.forward = lambda *a, **k: (_ for _ in ()).throw(AssertionError("torch encoder reached"))
    with torch.inference_mode():
        out = pol(td, env, phase="test", decode_type="greedy")
    assert torch.isfinite(out["reward"]).all()


@pytest.mark.parametrize("act", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("env_name,num_loc,batch", [("tsp", 100, 64), ("cvrp", 100, 48), ("tsp", 50, 32), ("cvrp", 20, 16), ("tsp", 128, 8)])
def test_16bit_fused_encoder_layer_norm_matches_float64(env_name, num_loc, batch, act):
    pol, env, td = _policy_and_td(env_name, num_loc, batch, encoder_autocast=act, cache_dtype=act)
    pe = pol._packed_encoder()
    assert pe.supported(td)
    with torch.inference_mode():
        cache, hidden = pe.encode(td, act, want_hidden=True, act_dtype=act)
        with torch.autocast("cuda", dtype=act):
            h16, _ = pol.encoder(td)
    torch.cuda.synchronize()
    h64, c64 = _double_reference(pol, td)
    tol = 3e-2 if act == torch.bfloat16 else 1e-2  # 16-bit operands and residual stream (the autocast regimes' own level)
    for nm, got, exact in (("hidden", hidden, h64), ("plane0", cache.kvl[0], c64.kvl[0]), ("plane2", cache.kvl[2], c64.kvl[2]),
                           ("ctx_cur", cache.ctx_cur, c64.ctx_cur), ("q_bias", cache.q_bias, c64.q_bias)):
        assert torch.isfinite(got.float()).all(), nm
        e = float((got.double() - exact).norm() / exact.norm())
        assert e <= tol, f"{nm}: {e:.4f}"
    e_auto = float((h16.double() - h64).norm() / h64.norm())
    e_k = float((hidden.double() - h64).norm() / h64.norm())
    assert e_k <= 2.5 * e_auto + 2e-3, (e_k, e_auto)
    pol.encoder.forward = lambda *a, **k: (_ for _ in ()).throw(AssertionError("torch encoder reached"))
    with torch.inference_mode():
        out = pol(td, env, phase="test", decode_type="greedy")
    assert torch.isfinite(out["reward"]).all()


def _layer_norm_ref(v):
    mean = v.mean((1, 2), keepdim=True)
    return (v - mean) / torch.sqrt(v.var((1, 2), keepdim=True) + 1e-5)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("b,n", [(64, 100), (2, 2), (3, 128), (17, 37), (5, 129), (9, 501), (2, 1024)])  # (beyond 128: rows re-read per pass, r06)
def test_skip_layer_norm_forward_backward_match_torch_autograd(b, n, dt):
    from rl4co_amd import train_ops

    torch.manual_seed(b * 1000 + n)
    x = (torch.randn(b, n, 128, device="cuda") * 1.5 + 0.3).to(dt).requires_grad_()
    s = (torch.randn(b, n, 128, device="cuda") * 0.7 - 0.2).to(dt).requires_grad_()
    g = torch.randn(b, n, 128, device="cuda").to(dt)
    out = train_ops.skip_layer_norm(x, s)
    out.backward(g)
    dx, ds = x.grad.clone(), s.grad.clone()
    # reference: the same 16-bit inputs, the skip sum rounded to the element type (as autocast's x + module(x)), fp32 after
    x32, s32 = x.detach().float().requires_grad_(), s.detach().float().requires_grad_()
    a = x32 + s32
    y = a.detach().to(dt).float() + (a - a.detach())  # value rounded to the element type, gradient straight through
    ref = _layer_norm_ref(y)
    ref.backward(g.float())
    eps16 = 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11
    assert out.dtype == dt and _rel(out, ref.detach()) <= 1.5 * eps16      # one rounding of the output
    assert torch.equal(dx, ds)                                             # the gradient of both skip inputs
    assert _rel(dx, x32.grad) <= 1.5 * eps16 and _rel(dx, s32.grad) <= 1.5 * eps16
    # statistics really are whole-instance and unbiased: mean 0, unbiased variance 1 (up to 16-bit rounding and eps)
    o32 = out.float()
    assert float(o32.mean((1, 2)).abs().max()) <= 4 * eps16
    assert float((o32.var((1, 2)) - 1).abs().max()) <= 8 * eps16


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("fused_blocks", [True, False], ids=["sub-block-nodes", "piecewise"])
def test_training_encoder_layer_norm_on_kernels_matches_torch_modules(dt, fused_blocks):
    """A layer-norm encoder under 16-bit autocast in training: the sub-block autograd nodes (and the piecewise path) on the
    kernels against the same modules on torch under the same autocast — outputs and every parameter gradient."""
    from rl4co_amd.policy import AttentionModelPolicy

    torch.manual_seed(0)
    pol = AttentionModelPolicy("tsp", normalization="layer", num_encoder_layers=2).cuda().train()
    ref = AttentionModelPolicy("tsp", normalization="layer", num_encoder_layers=2).cuda().train()
    ref.load_state_dict(pol.state_dict())
    for layer in ref.encoder.net.layers:
        layer.fused_train = False
    if not fused_blocks:  # force the piecewise branch (per-op linear / attention / skip + norm nodes)
        from rl4co_amd import train_ops

        real = train_ops.block_usable
        train_ops.block_usable = lambda *a, **k: False
    torch.manual_seed(1)
    x = torch.randn(48, 50, 128, device="cuda") * 0.8
    g = torch.randn(48, 50, 128, device="cuda")
    try:
        import warnings

        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            with torch.autocast("cuda", dtype=dt):
                o_k = pol.encoder.net(x.clone().requires_grad_())
                o_t = ref.encoder.net(x.clone().requires_grad_())
        (o_k.float() * g).sum().backward()
        (o_t.float() * g).sum().backward()
    finally:
        if not fused_blocks:
            train_ops.block_usable = real
    assert o_k.dtype == dt
    assert _rel(o_k, o_t) <= (3e-2 if dt == torch.bfloat16 else 6e-3)  # two 16-bit evaluations of two layers
    checked = 0
    for (nm, p), (_, q) in zip(pol.encoder.net.named_parameters(), ref.encoder.net.named_parameters()):
        assert p.grad is not None and q.grad is not None, nm
        cos = float(torch.nn.functional.cosine_similarity(p.grad.flatten().double(), q.grad.flatten().double(), dim=0))
        assert cos >= (0.99 if dt == torch.bfloat16 else 0.999), (nm, cos)  # gradient direction of each parameter tensor
        checked += 1
    assert checked == 16  # 2 layers x (Wqkv, bqkv, Wo, bo, W1, b1, W2, b2): layer norm has no parameters


# ---------------------------------------------------------------------------------------------------------------------
# beyond 128 nodes: instance / layer norm on the token-tile launches (the layer's halves stop before their norm, an apply
# kernel normalises with the tiles' combined statistics) — POMO checkpoints evaluated on larger graphs
# ---------------------------------------------------------------------------------------------------------------------
def _big_policy(env_name, normalization, num_loc, batch, layers=3, **kw):
    from rl4co_amd.envs import get_env
    from rl4co_amd.policy import AttentionModelPolicy

    torch.manual_seed(0)
    pol = AttentionModelPolicy(env_name, normalization=normalization, num_encoder_layers=layers, **kw).cuda().eval()
    with torch.no_grad():
        for layer in pol.encoder.net.layers:
            layer[0].module.out_proj.bias.normal_(0.0, 0.3)
            layer[2].module.lins[1].bias.normal_(0.0, 0.3)
            for nm in (layer[1], layer[3]):
                if nm.kind == "instance":
                    nm.normalizer.weight.uniform_(0.5, 1.5)
                    nm.normalizer.bias.normal_(0.0, 0.2)
    env = get_env(env_name, generator_params=dict(num_loc=num_loc, device="cuda"), device="cuda", check_solution=True)
    torch.manual_seed(3)
    return pol, env, env.reset(env.generator(batch_size=[batch]))


@pytest.mark.parametrize("normalization", ["instance", "layer"])
@pytest.mark.parametrize("env_name,num_loc,batch", [("tsp", 200, 24), ("cvrp", 200, 16), ("tsp", 129, 3), ("cvrp", 500, 8), ("tsp", 100, 32),
                                                    ("pdp", 140, 8)])
def test_fp32_token_tiles_instance_and_layer_norm_match_float64(env_name, num_loc, batch, normalization):
    pol, env, td = _big_policy(env_name, normalization, num_loc, batch)
    pe = pol._packed_encoder()
    assert pe.supported(td, torch.float32)
    with torch.inference_mode():
        cache, hidden = pe.encode(td, torch.float32, want_hidden=True, act_dtype=torch.float32, tokens=True)
        h32, _ = pol.encoder(td)
    torch.cuda.synchronize()
    h64, c64 = _double_reference(pol, td)
    for nm, got, exact in (("hidden", hidden, h64), ("plane0", cache.kvl[0], c64.kvl[0]), ("plane2", cache.kvl[2], c64.kvl[2]),
                           ("ctx_cur", cache.ctx_cur, c64.ctx_cur), ("q_bias", cache.q_bias, c64.q_bias)):
        assert torch.isfinite(got).all(), nm
        e_k = float((got.double() - exact).norm() / exact.norm())
        assert e_k <= 3e-6, f"{nm}: kernel rel err {e_k:.3e}"  # fp32 throughout: only the order of sums differs
    e_t = float((h32.double() - h64).norm() / h64.norm())
    e_k = float((hidden.double() - h64).norm() / h64.norm())
    assert e_k <= 3 * e_t + 2e-7, (e_k, e_t)
    if td["action_mask"].shape[-1] <= 128:  # the fused kernel on the same graph: same GEMM routine, other statistics order
        with torch.inference_mode():
            fused, hf = pe.encode(td, torch.float32, want_hidden=True, act_dtype=torch.float32, tokens=False)
        assert _rel(hidden, hf) <= 2e-6 and _rel(cache.kvl, fused.kvl) <= 2e-6
    else:  # the whole fp32 rollout stays off the torch encoder
        pol.encoder.forward = lambda *a, **k: (_ for _ in ()).throw(AssertionError("torch encoder reached"))
        with torch.inference_mode():
            out = pol(td, env, phase="test", decode_type="greedy")
        assert torch.isfinite(out["reward"]).all()


@pytest.mark.parametrize("act", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("normalization", ["instance", "layer"])
@pytest.mark.parametrize("env_name,num_loc,batch", [("tsp", 200, 24), ("cvrp", 300, 8), ("tsp", 129, 3), ("tsp", 100, 32)])
def test_16bit_token_tiles_instance_and_layer_norm_match_float64(env_name, num_loc, batch, normalization, act):
    pol, env, td = _big_policy(env_name, normalization, num_loc, batch, encoder_autocast=act, cache_dtype=act)
    pe = pol._packed_encoder()
    assert pe.supported(td)
    with torch.inference_mode():
        cache, hidden = pe.encode(td, act, want_hidden=True, act_dtype=act, tokens=True)
        with torch.autocast("cuda", dtype=act):
            h16, _ = pol.encoder(td)
    torch.cuda.synchronize()
    h64, c64 = _double_reference(pol, td)
    tol = 3e-2 if act == torch.bfloat16 else 1e-2  # 16-bit operands and residual stream
    for nm, got, exact in (("hidden", hidden, h64), ("plane0", cache.kvl[0], c64.kvl[0]), ("plane2", cache.kvl[2], c64.kvl[2]),
                           ("ctx_cur", cache.ctx_cur, c64.ctx_cur), ("q_bias", cache.q_bias, c64.q_bias)):
        assert torch.isfinite(got.float()).all(), nm
        e = float((got.double() - exact).norm() / exact.norm())
        assert e <= tol, f"{nm}: {e:.4f}"
    e_auto = float((h16.double() - h64).norm() / h64.norm())
    e_k = float((hidden.double() - h64).norm() / h64.norm())
    assert e_k <= 2.5 * e_auto + 2e-3, (e_k, e_auto)
    if td["action_mask"].shape[-1] <= 128:
        with torch.inference_mode():
            fused, hf = pe.encode(td, act, want_hidden=True, act_dtype=act, tokens=False)
        assert _rel(hidden, hf) <= tol
    else:
        pol.encoder.forward = lambda *a, **k: (_ for _ in ()).throw(AssertionError("torch encoder reached"))
        with torch.inference_mode():
            out = pol(td, env, phase="test", decode_type="greedy")
        assert torch.isfinite(out["reward"]).all()


def test_pomo_policy_evaluated_beyond_its_training_size_runs_on_the_kernels():
    """The generalisation protocol: a POMO policy (6 layers, instance norm, no graph context) rolled out with multistart
    greedy decoding on TSP-200 — encoder on the token tiles, decode on the fused kernels, the torch encoder unreachable —
    in the fp32 regime and under bf16 autocast; the two regimes' best-of-starts tour lengths agree to 3 % (a random-init
    policy is near-uniform: every step a near-tie, so the two regimes walk different — equally random — tours; measured 1.2 %)."""
    from rl4co_amd.envs import get_env
    from rl4co_amd.policy import AttentionModelPolicy

    torch.manual_seed(0)
    kw = dict(normalization="instance", num_encoder_layers=6, use_graph_context=False)
    pol = AttentionModelPolicy("tsp", **kw).cuda().eval()
    p16 = AttentionModelPolicy("tsp", encoder_autocast=torch.bfloat16, cache_dtype=torch.bfloat16, **kw).cuda().eval()
    p16.load_state_dict(pol.state_dict())
    env = get_env("tsp", generator_params=dict(num_loc=200, device="cuda"), device="cuda", check_solution=True)
    torch.manual_seed(5)
    td = env.reset(env.generator(batch_size=[8]))
    best = []
    for p in (pol, p16):
        p.encoder.forward = lambda *a, **k: (_ for _ in ()).throw(AssertionError("torch encoder reached"))
        with torch.inference_mode():
            out = p(td.clone(), env, phase="test", decode_type="multistart_greedy", num_starts=16, select_best=True)
        assert torch.isfinite(out["reward"]).all()
        best.append(out["reward"].double().mean())
    assert abs(float(best[0] - best[1])) <= 0.03 * abs(float(best[0]))


# ---------------------------------------------------------------------------------------------------------------------
# training: the MLP's input gradient as one launch (csrc/am_encoder.hip: tok16_mlp_bwd_kernel, rl4co_mlp_input_grad)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("m", [4096 * 100, 128, 129, 1000, 7])
def test_mlp_input_grad_one_launch_equals_the_two_gemm_launches(m, dt):
    """dh = (dy W2) * [h > 0], dx = dh W1 + dy: against the two tall-skinny GEMM launches it replaces (same 16-bit operands,
    fp32 accumulation, one rounding per output: dh must agree to the last bit except where the two kernels' fp32 summation
    orders round differently — bounded at one 16-bit ulp, < 0.1 % of the entries — and dx to 16-bit rounding) and against
    fp32 torch on the same operands."""
    from rl4co_amd import train_ops as T

    torch.manual_seed(m)
    dy = (torch.randn(m, 128, device="cuda") * 0.5).to(dt)
    h = torch.relu(torch.randn(m, 512, device="cuda")).to(dt)
    w1 = (torch.randn(512, 128, device="cuda") * 0.08).to(dt)   # lin1.weight [512,128]
    w2 = (torch.randn(128, 512, device="cuda") * 0.05).to(dt)   # lin2.weight [128,512]
    w1_t, w2_t = w1.t().contiguous(), w2.t().contiguous()
    dh, dx = T.mlp_input_grad(dy, h, T._pack_stack(w1_t[None])[0], T._pack_stack(w2_t[None])[0])
    dh_ref = T._gemm(dy, w2_t, mask=h)
    dx_ref = T._gemm(dh_ref, w1_t, residual=dy)
    torch.cuda.synchronize()
    eps16 = 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11
    assert dh.shape == (m, 512) and dx.shape == (m, 128) and dh.dtype == dt
    assert torch.equal(dh == 0, dh_ref == 0)                                   # the ReLU mask, exactly
    differ = (dh != dh_ref).float().mean()
    assert float(differ) <= 1e-3 and _rel(dh, dh_ref) <= 0.25 * eps16, (float(differ), _rel(dh, dh_ref))
    assert _rel(dx, dx_ref) <= 1.0 * eps16, _rel(dx, dx_ref)
    dh32 = (dy.float() @ w2.float()) * (h.float() > 0)
    dx32 = dh32.to(dt).float() @ w1.float() + dy.float()
    assert _rel(dh, dh32) <= 1.0 * eps16 and _rel(dx, dx32) <= 1.5 * eps16


def test_stack_backward_with_and_without_the_fused_mlp_input_grad_agree():
    """The instance-norm stack's backward (POMO training) with the one-launch MLP input gradient against the same backward on
    the two GEMM launches. The two differ by 16-bit roundings of dh / dx that propagate down the stack, so: every gradient
    that carries signal (norm >= 10: the input, the weight matrices, the norms' weights) has cosine >= 0.999 with its twin
    (>= 0.998 below norm 100);
    the rest — biases in front of an instance norm, whose exact gradient is ZERO (the bias cancels in the per-channel mean),
    the key bias of the attention — are round-off in both runs: below 1 % of the largest gradient norm."""
    from rl4co_amd import train_ops as T
    from rl4co_amd.policy import _GraphAttentionNetwork

    grads = {}
    for fused in (True, False):
        torch.manual_seed(0)
        net = _GraphAttentionNetwork(8, 128, 3, "instance", 512).cuda().train()
        torch.manual_seed(1)
        x = torch.randn(64, 100, 128, device="cuda") * 0.7
        g = torch.randn(64, 100, 128, device="cuda")
        T.FUSED_MLP_INPUT_GRAD = fused
        try:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                xin = x.clone().requires_grad_()
                out = net(xin)
            (out.float() * g).sum().backward()
        finally:
            T.FUSED_MLP_INPUT_GRAD = True
        grads[fused] = [xin.grad.clone()] + [p.grad.clone() for p in net.parameters()]
    assert len(grads[True]) == len(grads[False]) > 30
    top = max(float(b.norm()) for b in grads[False])
    checked, seen = 0, []
    for a, b in zip(grads[True], grads[False]):
        if min(float(a.norm()), float(b.norm())) >= 10.0:
            cos = float(torch.nn.functional.cosine_similarity(a.flatten().double(), b.flatten().double(), dim=0))
            seen.append((round(cos, 5), tuple(a.shape), round(float(b.norm()), 1)))
            checked += 1
        else:
            assert max(float(a.norm()), float(b.norm())) <= 0.01 * top
    print("lowest cosines (cos, shape, norm):", sorted(seen)[:6])
    # (r05: two 128-vectors of norm ~ 31 — a thirtieth of the largest — sit at 0.9990 / 0.9995, everything else >= 0.9999)
    assert min(c for c, _, nrm in seen if nrm >= 100.0) >= 0.999 and min(c for c, _, _ in seen) >= 0.998, sorted(seen)[:3]
    assert checked >= 20
