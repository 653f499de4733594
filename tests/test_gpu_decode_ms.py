"""GPU: multistart decode on the matrix cores (csrc/am_decode_ms.hip) vs the streaming kernel and vs its rounding-model
oracle (every environment of the decode kernel: TSP, CVRP, orienteering, prize-collecting TSP, pickup-delivery, CVRP-TW).

Floating-point variant => tolerance test (tolerances stated here). The MS kernel rounds the query
and the glimpse to bf16 for the MFMAs, so it cannot be bit-identical to the fp32 specified order;
it must (a) produce valid tours, (b) assign each chosen action a log-prob that the streaming kernel,
EVALUATING the same trajectory on the same bf16 planes, reproduces within 0.05 per step and 2 %
over the whole trajectory, (c) be greedy under that reference scoring up to the same noise
(chosen log-prob >= max log-prob - 0.1), (d) reach the same tour quality (mean within 1 %).
"""
import pytest
import torch

from oracle import reference_torch as R
from tests.helpers import GoldenCase, fold_cache, max_horizon, rollout_state

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def K():
    from rl4co_amd import kernels

    return kernels


def _rollout(K, g, td0, cache, starts, variant, mode="greedy", forced=None, want_all=False, **kw):
    st = rollout_state(g.env_name, td0, device="cuda", num_starts=starts)
    b, n = st["action_mask"].shape
    tmax = max_horizon(g.env_name, n)
    actions = torch.zeros(b, tmax, dtype=torch.int64, device="cuda")
    logps = torch.zeros(b, tmax, device="cuda")
    n_steps = torch.zeros(b, dtype=torch.int32, device="cuda")
    err = K.new_error_word("cuda")
    t0 = 0
    if starts > 0:
        from tests.helpers import apply_step

        torch.manual_seed(4242)  # (the orienteering problem draws its start nodes: the same draw on every call)
        first = g.env.select_start_nodes(td0, starts).cuda()
        actions[:, 0] = first
        apply_step(K, g.env_name, first, st)
        t0 = 1
    all_lp = torch.zeros(b, tmax, n, device="cuda") if want_all else None
    if forced is not None:
        f = torch.zeros(b, tmax, dtype=torch.int64, device="cuda")
        f[:, : forced.shape[1]] = forced
        forced = f
    K.am_decode(cache, st, mode=mode, max_steps=tmax - t0, t0=t0, actions=actions, logps=logps, err=err,
                n_steps=n_steps, variant=variant, forced_actions=forced, all_logps=all_lp, **kw)
    torch.cuda.synchronize()
    t = t0 + int(n_steps.max())
    return actions[:, :t], logps[:, :t], st, int(err.item()), all_lp


CASES = [("pomo_tsp20_b16_msgreedy", 20), ("pomo_tsp50_b8_mssampling", 8), ("pomo_tsp50_b8_mssampling", 40),
         ("pomo_cvrp20_b16_msgreedy", 20), ("c4_pomo_tsp100_b32_s8_sampling", 8), ("tsp100_b64_greedy", 5),
         ("cvrp100_b64_greedy", 6)]


@pytest.mark.parametrize("name,starts", CASES)
def test_ms_greedy_consistent_with_streaming_kernel(K, name, starts):
    g = GoldenCase(name)
    td0 = g.reset()
    with torch.inference_mode():
        h, _ = g.policy.encoder(td0)
    cache = fold_cache(g.policy, g.env_name, h, torch.bfloat16, device="cuda")
    a_ms, l_ms, st_ms, err, _ = _rollout(K, g, td0, cache, starts, "ms")
    assert err == 0 and bool(st_ms["done"].all())
    # validity
    rows = R.batchify({k: v for k, v in td0.items() if torch.is_tensor(v)}, starts)
    g.env.check_solution_validity(rows, a_ms.cpu())
    # the streaming kernel evaluates the same trajectories on the same planes
    a_ev, l_ev, st_ev, err2, all_lp = _rollout(K, g, td0, cache, starts, "stream", mode="evaluate",
                                               forced=a_ms, want_all=True)  # same column layout as `actions`
    assert err2 == 0 and torch.equal(a_ev, a_ms)
    t = a_ms.shape[1]
    assert float((l_ms - l_ev).abs().max()) <= 0.05, float((l_ms - l_ev).abs().max())
    ll_ms, ll_ev = l_ms.sum(1), l_ev.sum(1)
    assert float(((ll_ms - ll_ev).abs() / ll_ev.abs().clamp_min(1.0)).max()) <= 2e-2
    # greedy under the reference scoring, up to the bf16 noise
    best = all_lp[:, :t].max(-1).values
    decided = slice(1, t) if starts else slice(0, t)
    assert float((best[:, decided] - l_ev[:, decided]).max()) <= 0.1
    # same tour quality as the streaming kernel's own greedy rollout
    a_st, _, _, err3, _ = _rollout(K, g, td0, cache, starts, "stream")
    locs = td0["locs"].cuda()
    r_ms = K.tour_length(locs, a_ms.contiguous(), prepend_depot=(g.env_name == "cvrp"), negate=True)
    r_st = K.tour_length(locs, a_st.contiguous(), prepend_depot=(g.env_name == "cvrp"), negate=True)
    assert abs(float(r_ms.mean() - r_st.mean())) <= 1e-2 * abs(float(r_st.mean()))
    same = (a_ms.shape == a_st.shape) and float((a_ms == a_st).all(1).float().mean())
    print(f"{name} S={starts}: {same:.1%} trajectories identical to the streaming kernel, "
          f"max |dlogp| {float((l_ms - l_ev).abs().max()):.4f}")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("name,starts,mode", [("pomo_tsp50_b8_mssampling", 8, "sampling"), ("c4_pomo_tsp100_b32_s8_sampling", 8, "sampling"),
                                              ("pomo_cvrp20_b16_msgreedy", 20, "greedy"), ("pomo_tsp20_b16_msgreedy", 20, "greedy")])
def test_ms_16bit_context_tables_equal_their_widened_fp32_form(K, name, starts, mode, dtype):
    """(r06) ctx_dtype in the multistart kernel: context rows in the planes' 16-bit type at the row stride of the fused
    fold's [B, N, 5 * 128] matrix — widened on load, so actions and log-probs are bit for bit those of fp32 tables holding
    the same rounded values."""
    from tests.test_gpu_decode import _ctx_as_columns, _ctx_rounded

    g = GoldenCase(name)
    td0 = g.reset()
    with torch.inference_mode():
        h, _ = g.policy.encoder(td0)
    cache = fold_cache(g.policy, g.env_name, h, dtype, device="cuda")
    kw = dict(philox_seed=77) if mode == "sampling" else {}
    a1, l1, st1, e1, _ = _rollout(K, g, td0, _ctx_rounded(cache, dtype), starts, "ms", mode=mode, **kw)
    a2, l2, st2, e2, _ = _rollout(K, g, td0, _ctx_as_columns(cache, dtype), starts, "ms", mode=mode, **kw)
    assert e1 == 0 and e2 == 0 and torch.equal(a1, a2)
    assert torch.equal(l1.view(torch.int32), l2.view(torch.int32))
    for k in st1:
        assert torch.equal(st1[k], st2[k]), k


@pytest.mark.parametrize("name,starts", [("pomo_tsp50_b8_mssampling", 8), ("pomo_cvrp20_b16_msgreedy", 6)])
def test_ms_sampling_valid_and_consistent(K, name, starts):
    g = GoldenCase(name)
    td0 = g.reset()
    with torch.inference_mode():
        h, _ = g.policy.encoder(td0)
    cache = fold_cache(g.policy, g.env_name, h, torch.bfloat16, device="cuda")
    a1, l1, st1, err, _ = _rollout(K, g, td0, cache, starts, "ms", mode="sampling", philox_seed=11)
    a2, _, _, _, _ = _rollout(K, g, td0, cache, starts, "ms", mode="sampling", philox_seed=11)
    a3, _, _, _, _ = _rollout(K, g, td0, cache, starts, "ms", mode="sampling", philox_seed=12)
    assert err == 0 and torch.equal(a1, a2) and not torch.equal(a1, a3)
    rows = R.batchify({k: v for k, v in td0.items() if torch.is_tensor(v)}, starts)
    g.env.check_solution_validity(rows, a1.cpu())
    _, l_ev, _, err2, _ = _rollout(K, g, td0, cache, starts, "stream", mode="evaluate", forced=a1)
    assert err2 == 0 and float((l1 - l_ev).abs().max()) <= 0.05
    # sampled trajectories are not the greedy ones and their likelihood is lower on average
    ag, lg, _, _, _ = _rollout(K, g, td0, cache, starts, "ms")
    assert float(l1.sum(1).mean()) < float(lg.sum(1).mean())


def test_ms_is_auto_selected_for_multistart_and_policy_runs(K):
    from rl4co_amd.envs import get_env
    from rl4co_amd.policy import AttentionModelPolicy

    assert K.decode_row_groups(100, torch.bfloat16, 99, "auto", 4096 * 32, 4096) == 0     # MS from 3 starts up (r03: two
    assert K.decode_row_groups(100, torch.bfloat16, 99, "auto", 4096 * 8, 4096) == 0      # instances per column tile)
    assert K.decode_row_groups(100, torch.bfloat16, 99, "auto", 4096 * 3, 4096) == 0
    assert K.decode_row_groups(100, torch.bfloat16, 99, "auto", 4096 * 2, 4096) == 4      # 2 starts: streaming
    assert K.decode_row_groups(100, torch.bfloat16, 100, "auto", 4096, 4096) == 4          # single start: stream
    assert K.decode_row_groups(100, torch.float32, 99, "auto", 4096 * 8, 4096) == 2        # fp32 planes: stream
    # per environment, where MS was measured faster than one wave per trajectory (csrc/am_decode.hip resolve_variant)
    MS = 4
    for env_name, starts, want_ms in [("tsp", 8, True), ("tsp", 3, True), ("pdp", 3, True), ("pdp", 2, False), ("pctsp", 3, True),
                                      ("cvrp", 4, False), ("cvrp", 8, True), ("cvrp", 16, True), ("op", 32, False),
                                      ("cvrptw", 32, False)]:
        got = K.decode_variant(101, torch.bfloat16, 150, 4096 * starts, num_instances=4096, env_name=env_name)
        assert (got == MS) == want_ms, (env_name, starts, got)
    torch.manual_seed(0)
    kw = dict(num_encoder_layers=6, normalization="instance", use_graph_context=False, cache_dtype=torch.bfloat16,
              encoder_autocast=torch.bfloat16)
    pol = AttentionModelPolicy("tsp", **kw).cuda().eval()
    env = get_env("tsp", generator_params=dict(num_loc=50, device="cuda"), device="cuda")
    td = env.reset(batch_size=[64])
    with torch.inference_mode():
        out = pol(td, env, phase="test", decode_type="multistart_greedy", num_starts=50)  # check_solution on
    assert out["actions"].shape == (64 * 50, 50)
    best = out["reward"].view(50, 64).max(0).values
    with torch.inference_mode():
        single = pol(td, env, phase="test", decode_type="greedy")
    assert float(best.mean()) >= float(single["reward"].mean()) - 1e-3


@pytest.mark.parametrize("env_name,num_loc,starts", [("tsp", 128, 40), ("tsp", 120, 17), ("cvrp", 127, 9), ("cvrp", 111, 33)])
def test_ms_large_graphs_and_odd_start_counts_match_streaming_quality(env_name, num_loc, starts):
    """Edges of the multistart MFMA variant: N up to 128 (8 node tiles, context rows read from HBM when the
    fp32 table no longer fits LDS), start counts that are not multiples of a 16-wide column tile (half
    tiles, an odd tile after the pairs). Valid tours (check_solution on) and the same tour quality as the
    streaming kernel on the same instances: mean best-of-starts reward within 0.5 %."""
    from rl4co_amd.envs import get_env
    from rl4co_amd.policy import AttentionModelPolicy
    from rl4co_amd import kernels as Kmod

    torch.manual_seed(3)
    pol = AttentionModelPolicy(env_name, num_encoder_layers=3, normalization="instance", use_graph_context=False,
                               cache_dtype=torch.bfloat16, encoder_autocast=torch.bfloat16).cuda().eval()
    env = get_env(env_name, generator_params=dict(num_loc=num_loc, device="cuda"), device="cuda")
    td = env.reset(batch_size=[24])
    rewards = {}
    orig = Kmod.am_decode
    for variant in ("ms", "stream"):
        Kmod.am_decode = lambda *a, _v=variant, **k: orig(*a, **{**k, "variant": _v})
        try:
            with torch.inference_mode():
                out = pol(td, env, phase="test", decode_type="multistart_greedy", num_starts=starts)
        finally:
            Kmod.am_decode = orig
        rewards[variant] = out["reward"].view(starts, 24).max(0).values.mean().item()
    assert abs(rewards["ms"] - rewards["stream"]) <= 5e-3 * abs(rewards["stream"]), rewards


# ---------------------------------------------------------------------------------------------
# against the rounding-model oracle (oracle/rollout_ref.c: oracle_am_decode_ms)
# ---------------------------------------------------------------------------------------------

def _c_ms_rollout(g, td0, cache_cpu, starts, mode, exp_noise=None, forced=None):
    from oracle import c_oracle
    from tests.helpers import apply_step

    st = rollout_state(g.env_name, td0, device="cpu", num_starts=starts)
    b, n = st["action_mask"].shape
    tmax = max_horizon(g.env_name, n)
    actions = torch.zeros(b, tmax, dtype=torch.int64)
    logps = torch.zeros(b, tmax)
    n_steps = torch.zeros(b, dtype=torch.int32)
    err = torch.zeros(1, dtype=torch.int32)
    torch.manual_seed(4242)
    first = g.env.select_start_nodes(td0, starts)
    actions[:, 0] = first
    apply_step(c_oracle, g.env_name, first, st)
    if forced is not None:
        f = torch.zeros(b, tmax, dtype=torch.int64)
        f[:, : forced.shape[1]] = forced
        forced = f
    c_oracle.am_decode(cache_cpu, st, mode=mode, max_steps=tmax - 1, t0=1, actions=actions, logps=logps, err=err,
                       n_steps=n_steps, row_groups="ms", exp_noise=exp_noise, forced_actions=forced)
    assert int(err.item()) == 0
    t = 1 + int(n_steps.max())
    return actions[:, :t], logps[:, :t]


# MS kernel vs its rounding model, per decode step on the common prefix / under teacher forcing. The residual is the
# hardware's exp2 / log / rcp approximations and the MFMA summation order acting through the bf16 rounding points
# (a last-bit difference before a bf16 rounding moves one operand by 2^-9 relative).
MS_LOGP_TOL = 5e-3  # measured (r02): max 2.1e-3 on one step of 25 k, mean < 1e-6; identical trajectories 100 %
MS_IDENTICAL_FLOOR = 0.97


@pytest.mark.parametrize("name,starts,mode", [("c4_pomo_tsp100_b32_s8_sampling", 8, "sampling"),
                                              ("c4_pomo_tsp100_b32_s8_sampling", 8, "greedy"),
                                              ("pomo_tsp50_b8_mssampling", 40, "sampling"),
                                              ("pomo_tsp20_b16_msgreedy", 20, "greedy"),
                                              ("pomo_cvrp20_b16_msgreedy", 20, "greedy"),
                                              ("cvrp100_b64_greedy", 9, "sampling"),
                                              # r02: every environment of the decode kernel on the matrix cores
                                              ("pomo_pdp20_b16_msgreedy", 10, "greedy"), ("pdp100_b64_greedy", 16, "sampling"),
                                              ("pomo_pctsp20_b16_msgreedy", 20, "greedy"), ("pctsp100_b64_greedy", 9, "sampling"),
                                              ("pomo_op20_b16_msgreedy", 20, "greedy"), ("op100_b64_greedy", 8, "sampling"),
                                              ("pomo_cvrptw20_b16_mssampling", 20, "sampling"), ("cvrptw100_b64_greedy", 8, "greedy")])
def test_ms_kernel_follows_its_rounding_model_oracle(K, name, starts, mode):
    _check_ms_vs_model(K, name, starts, mode, torch.bfloat16)


@pytest.mark.parametrize("name,starts,mode", [("c4_pomo_tsp100_b32_s8_sampling", 8, "sampling"), ("pomo_tsp20_b16_msgreedy", 20, "greedy"),
                                              ("cvrp100_b64_greedy", 9, "sampling"), ("pdp100_b64_greedy", 16, "sampling"),
                                              ("pomo_cvrptw20_b16_mssampling", 20, "sampling")])
def test_ms_kernel_fp16_build_follows_its_rounding_model_oracle(K, name, starts, mode):
    """The IEEE-half build (csrc/am_decode_ms_f16.hip: fp16 planes, query, softmax numerators and glimpse on
    v_mfma_f32_16x16x16_f16) against the same model oracle with half rounding at the same three points."""
    _check_ms_vs_model(K, name, starts, mode, torch.float16)


def _check_ms_vs_model(K, name, starts, mode, dt):
    """BASELINE configs[3]'s rollout kernel (auto-selected from 8 starts on bf16 planes) against the C restatement
    with the SAME bf16 rounding points (query, softmax numerators, glimpse) and fp32 arithmetic elsewhere:
    (a) teacher-forced on the oracle's own trajectories every per-step log-probability agrees within MS_LOGP_TOL;
    (b) free-running, trajectories are identical except at near-ties (>= MS_IDENTICAL_FLOOR identical), and agree
    within the same tolerance on their common prefix."""
    from tests.test_gpu_decode import _record

    g = GoldenCase(name)
    td0 = g.reset()
    with torch.inference_mode():
        h, _ = g.policy.encoder(td0)
    cache = fold_cache(g.policy, g.env_name, h, dt, device="cuda")
    cache_cpu = cache.to("cpu")
    b = g.batch * starts
    n = g.num_loc + (g.env_name != "tsp")
    noise = None
    if mode == "sampling":
        torch.manual_seed(77)
        noise = torch.empty(max_horizon(g.env_name, n), b, n).exponential_(1).contiguous()
    a_c, l_c = _c_ms_rollout(g, td0, cache_cpu, starts, mode, exp_noise=noise)
    kw = {} if noise is None else dict(exp_noise=noise.cuda())
    a_ms, l_ms, st, err, _ = _rollout(K, g, td0, cache, starts, "ms", mode=mode, **kw)
    assert err == 0 and bool(st["done"].all())
    a_ms, l_ms = a_ms.cpu(), l_ms.cpu()
    rows = R.batchify({k: v for k, v in td0.items() if torch.is_tensor(v)}, starts)
    g.env.check_solution_validity(rows, a_ms)  # the reference's own validity rules on the kernel's tours
    t = min(a_ms.shape[1], a_c.shape[1])
    agree = (a_ms[:, :t] == a_c[:, :t])
    prefix = agree.long().cumprod(1).bool()            # columns before the first disagreement
    identical = float(agree.all(1).float().mean()) if a_ms.shape == a_c.shape else 0.0
    gap_prefix = float(((l_ms[:, :t] - l_c[:, :t]).abs() * prefix).max())
    # (a) teacher forcing: the kernel evaluates the ORACLE's trajectories
    a_ev, l_ev, _, err2, _ = _rollout(K, g, td0, cache, starts, "ms", mode="evaluate", forced=a_c.cuda())  # same column layout as `actions`
    assert err2 == 0 and torch.equal(a_ev.cpu()[:, : a_c.shape[1]], a_c)
    gap_forced = (l_ev.cpu()[:, : a_c.shape[1]] - l_c).abs()
    print(f"{name} S={starts} {mode}: {identical:.1%} of {b} trajectories identical to the rounding-model oracle; per-step "
          f"log-prob gap: teacher-forced max {float(gap_forced.max()):.2e} mean {float(gap_forced.mean()):.2e}, "
          f"common prefix max {gap_prefix:.2e}")
    _record(f"ms_vs_model/{name}/S{starts}/{mode}" + ("" if dt == torch.bfloat16 else "/f16"), {"identical_frac": identical, "forced_gap_max": float(gap_forced.max()),
                                                     "forced_gap_mean": float(gap_forced.mean()), "prefix_gap_max": gap_prefix})
    assert float(gap_forced.max()) <= MS_LOGP_TOL
    assert gap_prefix <= MS_LOGP_TOL
    assert identical >= MS_IDENTICAL_FLOOR


@pytest.mark.parametrize("env_name,num_loc,starts", [("tsp", 100, 8), ("cvrp", 50, 5), ("tsp", 20, 8), ("pdp", 20, 8)])
def test_ms_pair_mode_two_instances_per_column_tile(K, env_name, num_loc, starts):
    """At most 8 starts per instance: the MS kernel packs TWO instances into one 16-column tile (am_decode_ms.hip,
    make_layout). Instances are independent, so the rollout of the first 7 instances must not depend on whether an 8th
    shares the last workgroup (odd instance count: the second half of that tile is dead), nor on the order of the
    instances (pairs are formed from neighbours): bit-identical actions and log-probs either way."""
    from rl4co_amd.envs import get_env
    from rl4co_amd.policy import AttentionModelPolicy
    from rl4co_amd.tensordict import TensorDict

    torch.manual_seed(0)
    pol = AttentionModelPolicy(env_name, num_encoder_layers=3, normalization="instance", use_graph_context=False,
                               cache_dtype=torch.bfloat16, encoder_autocast=torch.bfloat16).cuda().eval()
    env = get_env(env_name, generator_params=dict(num_loc=num_loc, device="cuda"), device="cuda")
    torch.manual_seed(1)
    data = env.generator(batch_size=[8])
    n_nodes = num_loc + (env_name != "tsp")

    def rollout(idx):
        td = TensorDict({k: v[idx].contiguous() for k, v in data.items()}, batch_size=[len(idx)])
        td0 = env.reset(td)
        with torch.inference_mode():
            cache, _ = pol._packed_encoder().encode(td0, torch.bfloat16)
        st = pol._initial_state(td0, starts)
        b = len(idx) * starts
        tmax = pol._max_horizon(env_name, n_nodes)
        actions = torch.zeros(b, tmax, dtype=torch.int64, device="cuda")
        logps = torch.zeros(b, tmax, device="cuda")
        err = K.new_error_word("cuda")
        first = env.select_start_nodes(td0, num_starts=starts)
        actions[:, 0] = first
        pol._env_step_state(st, first, err)
        K.am_decode(cache, st, mode="greedy", max_steps=tmax - 1, t0=1, actions=actions, logps=logps, err=err, variant="ms")
        torch.cuda.synchronize()
        assert int(err.item()) == 0
        return actions.view(starts, len(idx), tmax), logps.view(starts, len(idx), tmax)

    a8, l8 = rollout(list(range(8)))
    a7, l7 = rollout(list(range(7)))                 # odd count: the last workgroup holds one instance
    assert torch.equal(a7, a8[:, :7]) and torch.equal(l7.view(torch.int32), l8[:, :7].view(torch.int32))
    perm = [3, 0, 6, 1, 7, 2, 5, 4]
    ap, lp = rollout(perm)                           # other neighbours
    inv = torch.tensor(perm).argsort().tolist()
    assert torch.equal(ap[:, inv], a8) and torch.equal(lp[:, inv].view(torch.int32), l8.view(torch.int32))
    a1, l1 = rollout([5])                            # a single instance: the one-instance kernel
    assert torch.equal(a1[:, 0], a8[:, 5]) and torch.equal(l1[:, 0].view(torch.int32), l8[:, 5].view(torch.int32))
