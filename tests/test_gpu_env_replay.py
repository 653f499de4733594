"""rl4co_env_replay: T environment transitions of given trajectories in one launch, against T calls of the step entry points
(the `evaluate` decoding's state sequence: /root/reference/rl4co/utils/decoding.py:448-461, constructive/base.py:226-263),
and the REINFORCE step beyond the backward kernels' node limit that is built on it."""
import warnings

import pytest
import torch

pytestmark = pytest.mark.gpu

ENVS = [("tsp", 20), ("cvrp", 20), ("op", 20), ("pctsp", 20), ("pdp", 20), ("cvrptw", 20), ("cvrp", 150), ("tsp", 200)]


def _rollout(env_name, num_loc, batch, starts):
    from rl4co_amd.envs import get_env
    from rl4co_amd.policy import AttentionModelPolicy

    torch.manual_seed(0)
    policy = AttentionModelPolicy(env_name).cuda().eval()
    env = get_env(env_name, generator_params=dict(num_loc=num_loc, device="cuda"), device="cuda", check_solution=False)
    torch.manual_seed(5)
    data = env.generator(batch_size=[batch])
    kw = dict(decode_type="multistart_sampling", num_starts=starts) if starts else dict(decode_type="sampling")
    with torch.no_grad():
        out = policy(env.reset(data), env, phase="test", seed=11, **kw)
    return policy, env, data, out["actions"]


@pytest.mark.parametrize("env_name,num_loc", ENVS)
@pytest.mark.parametrize("starts", [0, 3])
def test_one_launch_replay_equals_the_step_by_step_replay(env_name, num_loc, starts):
    """Bit-exact: masks, context nodes, context scalars of every step; the state is left where T step calls leave it."""
    if starts and env_name in ("pdp",):
        pytest.skip("no multistart for pickup-delivery in this package's environments")
    policy, env, data, actions = _rollout(env_name, num_loc, 12, starts)
    a = policy._replay(env.reset(data), actions, starts)
    b = policy._replay_stepwise(env.reset(data), actions, starts)
    assert a[0].dtype == torch.bool and torch.equal(a[0], b[0])
    assert len(a) == 4 and a[3] is None  # (mask bits only on request)
    assert len(a[1]) == len(b[1]) and all(torch.equal(x, y) for x, y in zip(a[1], b[1]))
    if env_name == "pdp":
        assert a[2] is None
    else:
        assert a[2].dtype == b[2].dtype and torch.equal(a[2], b[2])
    # (ragged environments: some trajectories are padded past their end — the padding steps are replayed like any other)


def test_replay_rejects_rows_that_do_not_match_and_reports_actions_out_of_range():
    from rl4co_amd import _lib
    from rl4co_amd import kernels as K

    policy, env, data, actions = _rollout("cvrp", 20, 8, 0)
    state = policy._initial_state(env.reset(data), 0)
    with pytest.raises(ValueError):
        K.env_replay("cvrp", state, actions[:4].contiguous(), state["vehicle_capacity"])
    with pytest.raises(ValueError):
        K.env_replay("cvrp", state, actions, None)
    err = K.new_error_word("cuda")
    bad = actions.clone()
    bad[0, 3] = 99
    K.env_replay("cvrp", state, bad, state["vehicle_capacity"], err)
    assert int(err.item()) & _lib.EBIT_INFEASIBLE


@pytest.mark.parametrize("env_name,num_loc,starts,norm", [("cvrp", 150, 0, "instance"), ("tsp", 160, 4, "instance"),
                                                          ("tsp", 140, 0, "batch"), ("cvrp", 129, 0, "layer")])
def test_reinforce_step_beyond_the_backward_kernels_node_limit(env_name, num_loc, starts, norm):
    """Graphs beyond 128 nodes train through the per-op kernels where they serve, torch where they do not, and the dense
    re-evaluation (r06: the step used to raise a TypeError under 16-bit autocast — torch's norm hands fp32 rows to the
    16-bit MLP kernel). The log-likelihood the step differentiates equals the rollout's, and the gradients are those of the
    same step on the torch encoder (cosine over all parameters)."""
    from rl4co_amd.envs import get_env
    from rl4co_amd.policy import AttentionModelPolicy, _EncoderLayer

    env = get_env(env_name, generator_params=dict(num_loc=num_loc, device="cuda"), device="cuda", check_solution=False)
    torch.manual_seed(2)
    data = env.generator(batch_size=[6])
    grads, lls = {}, {}
    for fused in (True, False):
        torch.manual_seed(0)
        pol = AttentionModelPolicy(env_name, num_encoder_layers=3, normalization=norm, use_graph_context=norm == "batch",
                                   cache_dtype=torch.bfloat16, encoder_autocast=torch.bfloat16,
                                   train_decode_type="multistart_sampling" if starts else "sampling").cuda().train()
        for m in pol.modules():
            if isinstance(m, _EncoderLayer):
                m.fused_train = fused
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            kw = dict(num_starts=starts) if starts else {}
            out = pol(env.reset(data), env, phase="train", seed=7, **kw)
        ll = out["log_likelihood"]
        assert torch.isfinite(ll).all() and ll.requires_grad
        adv = torch.linspace(-1.0, 1.0, ll.shape[0], device="cuda")
        (adv * ll).mean().backward()
        grads[fused] = {k: p.grad.detach().float().flatten() for k, p in pol.named_parameters() if p.grad is not None}
        lls[fused] = (ll.detach(), out["actions"])
        assert all(torch.isfinite(g).all() for g in grads[fused].values())
    if torch.equal(lls[True][1], lls[False][1]):  # same trajectories sampled (the encoders differ by 16-bit rounding only)
        a, b = grads[True], grads[False]
        dots = sum(float(a[k] @ b[k]) for k in a)
        cos = dots / (sum(float(v @ v) for v in a.values()) * sum(float(v @ v) for v in b.values())) ** 0.5
        assert cos >= 0.98, cos


def _bits_of(mask):
    """[B,T,N] bool -> [B,T,W] int32 feasibility bits, rows padded to whole 128-key chunks (what env_replay emits)."""
    b, t, n = mask.shape
    w = 4 * ((n + 127) // 128)
    m = torch.zeros((b, t, w * 32), dtype=torch.bool, device=mask.device)
    m[..., :n] = mask
    weights = (1 << torch.arange(32, device=mask.device, dtype=torch.int64))
    words = (m.view(b, t, w, 32).to(torch.int64) * weights).sum(-1)
    return (words & 0xFFFFFFFF).to(torch.int64).where(words < 2**31, words - 2**32).to(torch.int32).contiguous()


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("b_inst,s,t,n,masked", [(6, 1, 20, 20, True), (4, 3, 37, 50, True), (3, 2, 130, 129, True),
                                                 (5, 1, 200, 200, True), (2, 2, 300, 501, True), (3, 1, 64, 100, False),
                                                 (2, 1, 1, 2, True), (9, 2, 5, 17, True), (1, 1, 700, 1025, True)])
def test_glimpse_attention_matches_torch_sdpa_with_the_same_mask(b_inst, s, t, n, masked, dt):
    """csrc/am_cross_attn.hip forward / backward vs torch SDPA in fp32 on the same 16-bit operands and the same mask:
    heads within 1.5e-2 absolute, dq / dk / dv within 3e-2 relative Frobenius error (the training attention's bars);
    keys shared by the s starts of an instance, their gradients summed over the starts."""
    import torch.nn.functional as F

    from rl4co_amd import train_ops

    torch.manual_seed(100 * n + t)
    b = b_inst * s
    q = torch.randn(b, t, 128, device="cuda").to(dt)
    kv = torch.randn(b_inst, n, 256, device="cuda").to(dt)
    go = torch.randn(b, t, 128, device="cuda").to(dt)
    mask = torch.rand(b, t, n, device="cuda") < 0.6
    mask[..., 0] |= ~mask.any(-1)  # every query keeps a feasible key
    if n > 2:
        mask[:, :, -1] = False
    bits = _bits_of(mask) if masked else None
    assert train_ops.glimpse_attention_usable(q, kv, bits)
    qk, kk = q.clone().requires_grad_(True), kv.clone().requires_grad_(True)
    out = train_ops.glimpse_attention(qk, kk, bits)
    gq, gkv = torch.autograd.grad(out, [qk, kk], go)
    qr, kr = q.float().requires_grad_(True), kv.float().requires_grad_(True)
    kx = kr.unsqueeze(0).expand(s, b_inst, n, 256).reshape(b, n, 256)
    qh = qr.view(b, t, 8, 16).transpose(1, 2)
    kh = kx[..., :128].reshape(b, n, 8, 16).transpose(1, 2)
    vh = kx[..., 128:].reshape(b, n, 8, 16).transpose(1, 2)
    ref = F.scaled_dot_product_attention(qh, kh, vh, attn_mask=mask[:, None] if masked else None).transpose(1, 2).reshape(b, t, 128)
    rq, rkv = torch.autograd.grad(ref, [qr, kr], go.float())
    torch.testing.assert_close(out.detach().float(), ref.detach(), rtol=1.6e-2, atol=1.5e-2)
    for name, got, want in (("dq", gq, rq), ("dk", gkv[..., :128], rkv[..., :128]), ("dv", gkv[..., 128:], rkv[..., 128:])):
        rel = float((got.float() - want).norm() / want.norm())
        assert rel <= 3e-2, (name, rel)


def test_replay_mask_bits_are_the_mask_rows():
    policy, env, data, actions = _rollout("cvrp", 150, 5, 0)
    from rl4co_amd import kernels as K

    state = policy._initial_state(env.reset(data), 0)
    r = K.env_replay("cvrp", state, actions.contiguous(), state["vehicle_capacity"], None, mask_bits=True)
    assert r["mask_bits"].shape[-1] == 8 and torch.equal(r["mask_bits"], _bits_of(r["masks"]))


@pytest.mark.parametrize("clip,temp,masked", [(10.0, 1.0, True), (0.0, 1.0, True), (10.0, 0.7, False), (5.0, 2.0, True)])
@pytest.mark.parametrize("b,t,n", [(7, 20, 20), (3, 37, 101), (2, 130, 501)])
def test_logit_logp_matches_the_torch_chain_it_replaces(b, t, n, clip, temp, masked):
    """csrc/am_logit_logp.hip vs tanh-clip -> mask -> / temperature -> log_softmax -> gather in torch (fp32 both):
    log-probs within 2e-5 absolute, d raw within 2e-5 relative Frobenius (+ exact zeros on the infeasible nodes)."""
    import math

    from rl4co_amd import train_ops

    torch.manual_seed(n + t)
    raw = (30.0 * torch.randn(b, t, n, device="cuda")).requires_grad_(True)
    mask = torch.rand(b, t, n, device="cuda") < 0.5
    acts = torch.randint(0, n, (b, t), device="cuda")
    mask.scatter_(-1, acts[..., None], True)  # the given action is feasible
    g = torch.randn(b, t, device="cuda")
    out = train_ops.logit_logp(raw, _bits_of(mask) if masked else None, acts, clip, temp)
    (dr,) = torch.autograd.grad(out, [raw], g)
    ref_in = raw.detach().clone().requires_grad_(True)
    z = ref_in / math.sqrt(128)
    if clip > 0:
        z = torch.tanh(z) * clip
    if masked:
        z = z.masked_fill(~mask, float("-inf"))
    ref = torch.log_softmax(z / temp, -1).gather(-1, acts[..., None]).squeeze(-1)
    (rr,) = torch.autograd.grad(ref, [ref_in], g)
    torch.testing.assert_close(out.detach(), ref.detach(), rtol=0, atol=2e-5)
    assert float((dr - rr).norm() / rr.norm()) <= 2e-5
    if masked:
        assert bool((dr[~mask] == 0).all())


@pytest.mark.parametrize("env_name", ["tsp", "cvrp", "op", "pctsp", "pdp", "cvrptw"])
def test_replay_leaves_the_state_where_the_step_calls_leave_it(env_name):
    from rl4co_amd import kernels as K

    policy, env, data, actions = _rollout(env_name, 20, 9, 0)
    s1 = policy._initial_state(env.reset(data), 0)
    s2 = policy._initial_state(env.reset(data), 0)
    base = {"op": lambda s: s["max_length"][:, 0].contiguous(), "pctsp": lambda s: s["prize_required"],
            "cvrp": lambda s: s["vehicle_capacity"], "cvrptw": lambda s: s["vehicle_capacity"]}.get(env_name, lambda s: None)(s1)
    K.env_replay(env_name, s1, actions.contiguous(), base)
    err = K.new_error_word("cuda")
    for t in range(actions.shape[1]):
        policy._env_step_state(s2, actions[:, t].contiguous(), err)
    assert s1.keys() == s2.keys()
    for k in s1:
        assert torch.equal(s1[k], s2[k]), k


def test_training_beyond_128_nodes_learns():
    """POMO REINFORCE at TSP-150 (encoder sub-blocks, multistart rollout, dense re-evaluation: replay, glimpse attention and
    log-prob kernels): the sampled tour length falls from ~51 to ~19 in 60 steps (tools/probes/wide_learns.py); 40 here."""
    from rl4co_amd.envs import get_env
    from rl4co_amd.policy import AttentionModelPolicy

    torch.manual_seed(0)
    n, batch, starts = 150, 128, 8
    pol = AttentionModelPolicy("tsp", num_encoder_layers=3, normalization="instance", use_graph_context=False,
                               cache_dtype=torch.bfloat16, encoder_autocast=torch.bfloat16,
                               train_decode_type="multistart_sampling").cuda().train()
    env = get_env("tsp", generator_params=dict(num_loc=n, device="cuda"), device="cuda", check_solution=False)
    opt = torch.optim.Adam(pol.parameters(), lr=3e-4)
    hist = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for i in range(40):
            data = env.generator(batch_size=[batch])
            out = pol(env.reset(data), env, phase="train", seed=i, num_starts=starts)
            r = out["reward"].view(starts, batch).t()
            ll = out["log_likelihood"].view(starts, batch).t()
            loss = -((r - r.mean(1, keepdim=True)).detach() * ll).mean()
            assert torch.isfinite(loss)
            opt.zero_grad(set_to_none=True)
            loss.backward()
            torch.nn.utils.clip_grad_norm_(pol.parameters(), 1.0)
            opt.step()
            hist.append(float(-r.mean()))
    assert sum(hist[-5:]) / 5 < 0.6 * sum(hist[:5]) / 5, (hist[:5], hist[-5:])


def test_replay_of_a_2500_node_tour():
    """(beyond 64 KB of dynamic LDS for the state and instance data)"""
    from rl4co_amd.envs import get_env
    from rl4co_amd.policy import AttentionModelPolicy

    torch.manual_seed(0)
    policy = AttentionModelPolicy("tsp").cuda().eval()
    env = get_env("tsp", generator_params=dict(num_loc=2500, device="cuda"), device="cuda", check_solution=False)
    data = env.generator(batch_size=[2])
    actions = torch.stack([torch.randperm(2500, device="cuda") for _ in range(2)])
    a = policy._replay(env.reset(data), actions, 0, mask_bits=True)
    b = policy._replay_stepwise(env.reset(data), actions, 0)
    assert torch.equal(a[0], b[0]) and all(torch.equal(x, y) for x, y in zip(a[1], b[1])) and torch.equal(a[2], b[2])
    assert torch.equal(a[3], _bits_of(a[0]))
