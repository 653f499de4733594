"""GPU: teacher-forced backward on the matrix cores (csrc/am_teacher_mma.hip) vs the fp32 replay
kernel (csrc/am_teacher.hip, itself tested against torch autograd in test_gpu_teacher.py).

Both kernels get the SAME bf16 planes, trajectories and upstream gradients. The MMA variant rounds
its MFMA operands (queries, softmax numerators, glimpses, d logits, d scores) to bf16 and
accumulates in fp32, so this is a tolerance test: every gradient tensor within 3e-2 relative
Frobenius error of the replay kernel's (measured 3e-3 .. 6e-3); recomputed per-step log-probs
within 0.15 absolute at the worst step of the worst trajectory (measured 8e-3 TSP, 4e-2 … 1.1e-1
CVRP, where the capacity column widens the bf16 query's range) and 1e-2 on average.
"""
import pytest
import torch

from tests.helpers import GoldenCase
from tests.test_gpu_teacher import _policy, _td

pytestmark = pytest.mark.gpu

GRAD_RTOL = 3e-2
LOGP_ATOL = 1.5e-1
LOGP_MEAN = 1e-2


def _capture(name, starts, cache_dtype=torch.bfloat16, seed=5):
    """Sampled trajectories + the folded cache the policy would hand to the backward kernel."""
    from rl4co_amd import teacher

    g = GoldenCase(name)
    env, td = _td(g)
    pol = _policy(g, cache_dtype=cache_dtype)
    kw = dict(num_starts=starts) if starts else {}
    got = {}
    orig = teacher.teacher_forced_logps

    def spy(env_name, cache_g, cache, actions, logps, meta):
        got.update(cache=cache, actions=actions.clone(), logps=logps.detach().clone(), meta=dict(meta))
        return orig(env_name, cache_g, cache, actions, logps, meta)

    teacher.teacher_forced_logps = spy
    try:
        pol(env.reset(td), env, phase="train", decode_type="multistart_sampling" if starts else "sampling", seed=seed, **kw)
    finally:
        teacher.teacher_forced_logps = orig
    assert got, "the policy did not take the kernel backward path"
    return got


def _compare(got, grad_scale=1.0):
    from rl4co_amd import teacher

    torch.manual_seed(3)
    grad = torch.randn(got["actions"].shape, device="cuda") * grad_scale
    ref = teacher.run_backward(got["cache"], got["actions"], grad, got["meta"], variant="replay", want_logp=True)
    out = teacher.run_backward(got["cache"], got["actions"], grad, got["meta"], variant="mma", want_logp=True)
    assert ref["variant"] == "replay" and out["variant"] == "mma"
    assert int(ref["err"]) == 0 and int(out["err"]) == 0
    worst = {}
    for k in ("d_kvl", "d_ctx_first", "d_ctx_cur", "d_q_bias", "d_extra"):
        if ref[k] is None:
            assert out[k] is None
            continue
        if k == "d_kvl":
            for i, nm in enumerate(("glimpse_key", "glimpse_val", "logit_key")):
                e = float((out[k][i] - ref[k][i]).norm()) / max(float(ref[k][i].norm()), 1e-30)
                worst[f"d_{nm}"] = e
        else:
            worst[k] = float((out[k] - ref[k]).norm()) / max(float(ref[k].norm()), 1e-30)
    # decoded columns only: both kernels leave imposed / finished columns untouched (zero here)
    dlogp = float((out["logp"] - ref["logp"]).abs().max())
    print({k: f"{v:.2e}" for k, v in worst.items()}, f"max |dlogp| {dlogp:.2e}")
    for k, e in worst.items():
        assert e <= GRAD_RTOL, f"{k}: relative error {e:.3e}"
    assert dlogp <= LOGP_ATOL
    assert float((out["logp"] - ref["logp"]).abs().mean()) <= LOGP_MEAN
    # the recomputed log-probs also agree with the rollout kernel's own values
    dec = ref["logp"] != 0
    assert float((out["logp"] - got["logps"])[dec].abs().max()) <= LOGP_ATOL
    return worst


@pytest.mark.parametrize("name,starts", [("tsp20_b64_greedy_simple", 0), ("tsp50_b64_greedy", 0),
                                         ("cvrp20_b128_greedy", 0), ("tsp100_b64_greedy", 0),
                                         ("cvrp100_b64_greedy", 0), ("pomo_tsp20_b16_msgreedy", 5),
                                         ("pomo_cvrp20_b16_msgreedy", 4), ("c4_pomo_tsp100_b32_s8_sampling", 8),
                                         ("op50_b64_sampling", 0), ("pctsp50_b64_sampling", 0), ("pdp50_b64_sampling", 0),
                                         ("cvrptw50_b64_sampling", 0), ("pomo_pdp20_b16_msgreedy", 5),
                                         ("pomo_cvrptw20_b16_mssampling", 6)])
def test_mma_matches_replay(name, starts):
    _compare(_capture(name, starts))


@pytest.mark.parametrize("name,starts", [("tsp50_b64_greedy", 0), ("cvrp100_b64_greedy", 0), ("pomo_tsp20_b16_msgreedy", 5),
                                         ("c4_pomo_tsp100_b32_s8_sampling", 8), ("pdp50_b64_sampling", 0), ("cvrptw50_b64_sampling", 0)])
def test_mma_fp16_build_matches_replay_on_fp16_planes(name, starts):
    """csrc/am_teacher_mma_f16.hip (fp16 planes and intermediates on v_mfma_f32_16x16x16_f16) against the fp32 replay
    kernel reading the SAME fp16 planes: same bounds as the bf16 build (11 significant bits instead of 8: the measured
    errors are smaller)."""
    got = _capture(name, starts, cache_dtype=torch.float16)
    assert got["cache"].kvl.dtype == torch.float16
    _compare(got)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("name,starts", [("tsp50_b64_greedy", 0), ("c4_pomo_tsp100_b32_s8_sampling", 8), ("pomo_tsp20_b16_msgreedy", 5),
                                         ("cvrp100_b64_greedy", 0), ("pomo_cvrp20_b16_msgreedy", 4), ("pdp50_b64_sampling", 0)])
def test_mma_16bit_context_tables_and_gradients_in_the_planes_matrix(name, starts, dtype):
    """(r06) ctx_dtype + d_ctx_in_planes: the context tables read as 16-bit rows at the stride of the fused fold's
    [B, N, nblk * 128] matrix, their gradients written (converted) as planes 3 / 4 of the gradient matrix — against the
    same launch on fp32 tables holding the same rounded values: plane gradients bit-identical (register sums, fixed order),
    context-table gradients equal up to the 16-bit rounding of the output (their fp32 sums are built by L2 atomics)."""
    import dataclasses

    from rl4co_amd import teacher

    got = _capture(name, starts, cache_dtype=dtype)
    cache = got["cache"]
    tsp = cache.env_name == "tsp"
    nblk = 5 if tsp else 4
    b, n = cache.num_instances, cache.num_nodes
    big = torch.zeros(b, n, nblk, 128, dtype=dtype, device="cuda")
    big[:, :, :3] = cache.kvl.permute(1, 2, 0, 3)
    big[:, :, nblk - 1] = cache.ctx_cur.to(dtype)
    if tsp:
        big[:, :, 3] = cache.ctx_first.to(dtype)
    wide = dataclasses.replace(cache, ctx_cur=cache.ctx_cur.to(dtype).float(),
                               ctx_first=cache.ctx_first.to(dtype).float() if tsp else None)
    cols = dataclasses.replace(cache, kvl=big.permute(2, 0, 1, 3)[:3], ctx_cur=big[:, :, nblk - 1], ctx_first=big[:, :, 3] if tsp else None)
    torch.manual_seed(3)
    grad = torch.randn(got["actions"].shape, device="cuda")
    dp_a = torch.zeros(b, n, 3, 128, dtype=dtype, device="cuda")
    dp_b = torch.full((b, n, nblk, 128), float("nan"), dtype=dtype, device="cuda")
    out_a = teacher.run_backward(wide, got["actions"], grad, got["meta"], variant="mma", want_logp=True, d_planes=dp_a.permute(2, 0, 1, 3))
    out_b = teacher.run_backward(cols, got["actions"], grad, got["meta"], variant="mma", want_logp=True, d_planes=dp_b.permute(2, 0, 1, 3))
    assert int(out_a["err"]) == 0 and int(out_b["err"]) == 0
    assert torch.equal(out_a["logp"], out_b["logp"])
    assert torch.equal(dp_a.view(torch.int16), dp_b[:, :, :3].contiguous().view(torch.int16))
    ulp = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
    want_cur = out_a["d_ctx_cur"]
    torch.testing.assert_close(dp_b[:, :, nblk - 1].float(), want_cur, rtol=2 * ulp, atol=1e-6 * float(want_cur.abs().max()) + 1e-30)
    if tsp:
        want_first = out_a["d_ctx_first"]
        torch.testing.assert_close(dp_b[:, :, 3].float(), want_first, rtol=2 * ulp, atol=1e-6 * float(want_first.abs().max()) + 1e-30)
    for k in ("d_q_bias", "d_extra"):
        if out_a[k] is not None:
            torch.testing.assert_close(out_b[k], out_a[k], rtol=1e-5, atol=1e-6 * float(out_a[k].abs().max()))


def test_auto_picks_mma_for_bf16_and_replay_for_f32():
    from rl4co_amd import teacher

    got = _capture("tsp20_b64_greedy_simple", 0)
    grad = torch.ones(got["actions"].shape, device="cuda")
    assert teacher.run_backward(got["cache"], got["actions"], grad, got["meta"])["variant"] == "mma"
    got32 = _capture("tsp20_b64_greedy_simple", 0, cache_dtype=torch.float32)
    assert teacher.run_backward(got32["cache"], got32["actions"], grad, got32["meta"])["variant"] == "replay"
    with pytest.raises(RuntimeError):
        teacher.run_backward(got32["cache"], got32["actions"], grad, got32["meta"], variant="mma")


def test_mma_flags_infeasible_actions():
    from rl4co_amd import _lib, teacher

    got = _capture("tsp20_b64_greedy_simple", 0)
    acts = got["actions"].clone()
    acts[3, 5] = acts[3, 2]  # node visited twice
    grad = torch.ones(acts.shape, device="cuda")
    out = teacher.run_backward(got["cache"], acts, grad, got["meta"], variant="mma")
    assert int(out["err"]) & _lib.EBIT_INFEASIBLE


def test_mma_gradient_is_linear_in_upstream_gradient():
    """d(cache) is linear in grad_logp: scaling the upstream gradient scales every output (bf16 operands
    carry an 8-bit exponent, so tiny REINFORCE advantages do not underflow)."""
    from rl4co_amd import teacher

    got = _capture("tsp50_b64_greedy", 0)
    torch.manual_seed(1)
    grad = torch.randn(got["actions"].shape, device="cuda")
    a = teacher.run_backward(got["cache"], got["actions"], grad, got["meta"], variant="mma")
    b = teacher.run_backward(got["cache"], got["actions"], grad * 2.0 ** -20, got["meta"], variant="mma")
    for k in ("d_kvl", "d_ctx_first", "d_ctx_cur", "d_q_bias", "d_extra"):
        if a[k] is not None:
            torch.testing.assert_close(b[k] * 2.0 ** 20, a[k], rtol=1e-5, atol=1e-6 * float(a[k].abs().max()))


@pytest.mark.parametrize("env_name,num_loc,starts", [("tsp", 112, 3), ("cvrp", 111, 0), ("cvrp", 100, 5), ("tsp", 113, 2),
                                                     ("tsp", 128, 3), ("cvrp", 127, 0)])
def test_mma_at_the_node_limit_and_long_horizons(env_name, num_loc, starts):
    """N = 112 (seven node tiles full), N = 113 .. 128 (the eight-tile instantiation), CVRP horizons beyond 128 columns (more than eight 16-step blocks),
    odd multistart counts — on freshly drawn instances instead of the golden ones."""
    from rl4co_amd import teacher
    from rl4co_amd.envs import get_env
    from rl4co_amd.policy import AttentionModelPolicy

    torch.manual_seed(11)
    pol = AttentionModelPolicy(env_name, cache_dtype=torch.bfloat16).cuda().train()
    env = get_env(env_name, generator_params=dict(num_loc=num_loc, device="cuda"), device="cuda", check_solution=False)
    td = env.reset(batch_size=[48])
    got = {}
    orig = teacher.teacher_forced_logps

    def spy(env_name_, cache_g, cache, actions, logps, meta):
        got.update(cache=cache, actions=actions.clone(), logps=logps.detach().clone(), meta=dict(meta))
        return orig(env_name_, cache_g, cache, actions, logps, meta)

    teacher.teacher_forced_logps = spy
    try:
        kw = dict(num_starts=starts, decode_type="multistart_sampling") if starts else dict(decode_type="sampling")
        pol(td, env, phase="train", seed=2, **kw)
    finally:
        teacher.teacher_forced_logps = orig
    assert got and got["actions"].shape[1] <= 256
    if env_name == "cvrp" and not starts:
        assert got["actions"].shape[1] > 128, got["actions"].shape  # really exercises > 8 step blocks
    _compare(got)


@pytest.mark.parametrize("env_name,num_loc", [("op", 20), ("op", 100), ("pctsp", 20), ("pctsp", 100), ("pdp", 20),
                                              ("pdp", 100), ("cvrptw", 20), ("cvrptw", 50), ("op", 127), ("pdp", 126),
                                              ("tsp", 128), ("cvrp", 120)])
def test_mma_orienteering_matches_torch_autograd(env_name, num_loc):
    """Orienteering and prize-collecting TSP have no replay kernel: the MMA backward (closed-form replay of tour
    length / collected prize and of their masks) is checked against torch autograd through the dense re-evaluation on the same trajectories.
    bf16 planes on the kernel side => parameter gradients within 3e-2 relative Frobenius error (plus the
    noise floor of tensors that carry no signal), log-likelihood within 0.15 per trajectory."""
    from rl4co_amd.envs import get_env
    from rl4co_amd.policy import AttentionModelPolicy

    env = get_env(env_name, generator_params=dict(num_loc=num_loc, device="cuda"), device="cuda", check_solution=False)
    torch.manual_seed(4)
    data = env.generator(batch_size=[96])

    def make(fused):
        torch.manual_seed(0)
        return AttentionModelPolicy(env_name, cache_dtype=torch.bfloat16, fused_backward=fused).cuda().train()

    with torch.no_grad():
        out0 = make(False).eval()(env.reset(data), env, phase="test", decode_type="sampling", seed=9)
    actions = out0["actions"]
    adv = torch.linspace(-1.0, 1.0, actions.shape[0], device="cuda")
    res = {}
    for fused in (True, False):
        pol = make(fused)
        used = []
        if fused:
            from rl4co_amd import teacher

            orig = teacher.run_backward
            teacher.run_backward = lambda *a, **k: (used.append(1), orig(*a, **k))[1]
        out = pol(env.reset(data), env, phase="train", actions=actions)
        (adv * out["log_likelihood"]).mean().backward()
        if fused:
            teacher.run_backward = orig
            assert used, "the MMA backward kernel was not used for the orienteering policy"
        res[fused] = (out["log_likelihood"].detach(), {k: p.grad for k, p in pol.named_parameters() if p.grad is not None})
    # sampled untrained trajectories of PCTSP-100 run ~100 steps at log p ~ -4 each: the bf16-plane noise of the
    # sum grows with T, hence the relative term (measured 0.18 on a log-likelihood of -428)
    # CVRPTW feeds unnormalised coordinates / times (to 150 / 480) through the same planes: the clipped logits saturate
    # and the bf16 noise of single trajectories is larger (measured 2.0e-3 / 5.4e-3 relative on single instances of the 20- / 50-node draws of the
    # device generator, r02; 8e-3 seen through the GPU encoder in r01; tests/helpers.ll_rtol documents the same effect for the fp32 planes)
    torch.testing.assert_close(res[True][0], res[False][0], rtol=1e-2 if env_name == "cvrptw" else 1e-3, atol=0.15)
    scale = max(float(g.norm()) for g in res[False][1].values())
    for k, gt in res[False][1].items():
        gk = res[True][1][k]
        err = float((gk - gt).norm())
        assert err <= 3e-2 * float(gt.norm()) + 2e-3 * scale, (k, err, float(gt.norm()), scale)
