"""CPU tests of the oracle chain (no GPU):

  reference source  ==(bit-exact, oracle/gen_golden.py)==  torch restatement  -> tests/golden
  torch restatement / goldens  ~=  C specified-order oracle   (this file: flip rate, 1e-5 logp)
  C oracle tour length / env steps  ==  torch (bit-exact, this file)
"""
import numpy as np
import pytest
import torch

from oracle import c_oracle
from oracle import reference_torch as R
from tests.helpers import ll_rtol  # noqa: E402
from tests.helpers import (GoldenCase, apply_step, decode_level, oracle_reward, clone_td, fold_cache, make_instances, manifest, max_horizon,
                           rollout_state)

ALL_CASES = sorted(manifest())
SMALL_CASES = [c for c in ALL_CASES if manifest()[c]["batch"] <= 256]


# ---------------------------------------------------------------------------------------------
# goldens vs the torch restatement
# ---------------------------------------------------------------------------------------------

@pytest.mark.parametrize("name", SMALL_CASES)
def test_restatement_reproduces_reference_golden(name):
    """The restatement, re-run from the seeds, reproduces the REAL reference's stored outputs."""
    g = GoldenCase(name)
    torch.manual_seed(g.meta["sample_seed"])
    with torch.inference_mode():
        out = g.policy(g.reset(), g.env, phase="test", decode_type=g.meta["decode_type"], **g.meta["forward_kwargs"])
    assert torch.equal(out["actions"], g.actions)
    assert torch.equal(out["reward"], g.reward)
    assert torch.equal(out["log_likelihood"], g.log_likelihood)


# ---------------------------------------------------------------------------------------------
# reward arithmetic: C oracle == ATen, bit for bit
# ---------------------------------------------------------------------------------------------

@pytest.mark.parametrize("n", [2, 3, 7, 8, 9, 20, 21, 31, 32, 33, 50, 100, 101, 127, 128, 201, 500, 501, 531,
                               511, 512, 513, 640, 1001, 2049, 4100])
def test_tour_length_bit_exact_vs_aten(n):
    g = torch.Generator().manual_seed(n)
    b = 64
    locs = torch.rand(b, n, 2, generator=g)
    actions = torch.stack([torch.randperm(n, generator=g) for _ in range(b)])
    want = R.get_tour_length(R.gather_by_index(locs, actions))
    got = c_oracle.tour_length(locs, actions)
    assert torch.equal(got, want)


@pytest.mark.parametrize("n,t", [(21, 30), (101, 118), (101, 160), (501, 531), (501, 640)])
def test_cvrp_reward_bit_exact_vs_aten(n, t):
    """CVRP: depot prepended, customers once, padding depot visits (zero-length segments)."""
    g = torch.Generator().manual_seed(n * 1000 + t)
    b = 32
    locs = torch.rand(b, n, 2, generator=g)
    actions = torch.zeros(b, t, dtype=torch.int64)
    for i in range(b):
        pos = torch.randperm(t, generator=g)[: n - 1].sort().values
        actions[i, pos] = torch.randperm(n - 1, generator=g) + 1
    env = R.CVRPEnv(num_loc=n - 1, check_solution=False)
    want = env.get_reward({"locs": locs}, actions)
    got = c_oracle.tour_length(locs, actions, prepend_depot=True, negate=True)
    assert torch.equal(got, want)


def test_tour_length_multistart_row_mapping():
    """Trajectory b reads instance b % B_locs (s-major batchify, ops.py:10-28)."""
    g = torch.Generator().manual_seed(5)
    locs = torch.rand(4, 20, 2, generator=g)
    s = 3
    actions = torch.stack([torch.randperm(20, generator=g) for _ in range(4 * s)])
    want = R.get_tour_length(R.gather_by_index(R.batchify(locs, s), actions))
    assert torch.equal(c_oracle.tour_length(locs, actions), want)


# ---------------------------------------------------------------------------------------------
# env transitions: C oracle == torch restatement under a random feasible policy
# ---------------------------------------------------------------------------------------------

@pytest.mark.parametrize("env_name,num_loc", [("op", 20), ("op", 100), ("pctsp", 20), ("pctsp", 100), ("pdp", 20),
                                              ("pdp", 100), ("cvrptw", 20), ("cvrptw", 100)])
def test_depot_env_steps_match_restatement(env_name, num_loc):
    """Orienteering / prize-collecting TSP transitions and masks: C oracle == restatement, bit for bit,
    under a random feasible policy; then the reward composition of tests/helpers.oracle_reward."""
    env, data = make_instances(env_name, num_loc, 64)
    td0 = env.reset(clone_td(data))
    td = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in td0.items()}
    st = rollout_state(env_name, td)
    g = torch.Generator().manual_seed(3)
    acts = []
    while not bool(td["done"].all()):
        action = torch.multinomial(td["action_mask"].float(), 1, generator=g).squeeze(1)
        td["action"] = action
        td = env.step(td)
        acts.append(action)
        if env_name == "op":
            c_oracle.op_step(action, st["locs"], st["max_length"], st["tour_length"], st["visited"], st["current_node"],
                             st["i"], st["action_mask"], st["done"])
            assert torch.equal(st["tour_length"], td["tour_length"])
        elif env_name == "cvrptw":
            apply_step(c_oracle, "cvrptw", action, st)
            assert torch.equal(st["current_time"], td["current_time"].reshape(-1))
            assert torch.equal(st["used_capacity"], td["used_capacity"].reshape(-1))
        elif env_name == "pdp":
            apply_step(c_oracle, "pdp", action, st)
            assert torch.equal(st["available"].bool(), td["available"])
            assert torch.equal(st["to_deliver"].bool(), td["to_deliver"])
        else:
            c_oracle.pctsp_step(action, st["real_prize"], st["cur_total_prize"], st["visited"], st["current_node"],
                                st["i"], st["action_mask"], st["done"])
            assert torch.equal(st["cur_total_prize"], td["cur_total_prize"])
        if env_name != "pdp":
            assert torch.equal(st["visited"].bool(), td["visited"].bool())
        assert torch.equal(st["action_mask"], td["action_mask"])
        assert torch.equal(st["done"], td["done"].reshape(-1))
        if env_name != "cvrptw":
            assert torch.equal(st["i"], td["i"].reshape(-1))
    actions = torch.stack(acts, 1)
    assert torch.equal(oracle_reward(env_name, td0, actions), env.get_reward(td0, actions))


@pytest.mark.parametrize("env_name,num_loc", [("tsp", 20), ("cvrp", 20), ("cvrp", 50)])
def test_env_steps_match_restatement(env_name, num_loc):
    env, data = make_instances(env_name, num_loc, 64)
    td = env.reset(clone_td(data))
    st = rollout_state(env_name, td)
    g = torch.Generator().manual_seed(3)
    for _ in range(3 * num_loc):
        if bool(td["done"].all()):
            break
        probs = td["action_mask"].float()
        action = torch.multinomial(probs, 1, generator=g).squeeze(1)
        td["action"] = action
        td = env.step(td)
        if env_name == "tsp":
            c_oracle.tsp_step(action, st["action_mask"], st["first_node"], st["current_node"], st["i"], st["done"])
            assert torch.equal(st["first_node"], td["first_node"])
            assert torch.equal(st["i"], td["i"].reshape(-1))
        else:
            c_oracle.cvrp_step(action, st["demand"], st["used_capacity"], st["vehicle_capacity"], st["visited"],
                               st["current_node"], st["action_mask"], st["done"])
            assert torch.equal(st["used_capacity"], td["used_capacity"].reshape(-1))
            assert torch.equal(st["visited"], td["visited"])
        assert torch.equal(st["action_mask"], td["action_mask"])
        assert torch.equal(st["current_node"], td["current_node"].reshape(-1))
        assert torch.equal(st["done"], td["done"].reshape(-1))
    assert bool(td["done"].all())


# ---------------------------------------------------------------------------------------------
# decode loop: C specified-order oracle vs the reference goldens
# ---------------------------------------------------------------------------------------------

def c_rollout(g: GoldenCase, mode: str, cache_dtype=torch.float32, row_groups=None, exp_noise=None, fold=True):
    """Encoder through the restatement (stock torch), decode loop through the C oracle."""
    td0 = g.reset()
    with torch.inference_mode():
        h, _ = g.policy.encoder(td0)
    cache = fold_cache(g.policy, g.env_name, h, cache_dtype, fold=fold)
    s = g.num_starts
    st = rollout_state(g.env_name, td0, num_starts=s)
    b, n = st["action_mask"].shape
    tmax = max_horizon(g.env_name, n)
    actions = torch.zeros(b, tmax, dtype=torch.int64)
    logps = torch.zeros(b, tmax)
    n_steps = torch.zeros(b, dtype=torch.int32)
    err = torch.zeros(1, dtype=torch.int32)
    t0 = 0
    if s > 0:
        first = g.start_nodes(td0, s)
        actions[:, 0] = first
        apply_step(c_oracle, g.env_name, first, st)
        t0 = 1
    if row_groups is None:
        row_groups = 2 if cache_dtype == torch.float32 else 4
    c_oracle.am_decode(cache, st, mode=mode, max_steps=tmax - t0, t0=t0, actions=actions, logps=logps, err=err,
                       row_groups=row_groups, n_steps=n_steps, exp_noise=exp_noise,
                       mask_inner=True, tanh_clipping=10.0)
    assert int(err.item()) == 0
    t = t0 + int(n_steps.max())
    return actions[:, :t].contiguous(), logps[:, :t], td0


GREEDY_SMALL = [c for c in SMALL_CASES if "greedy" in manifest()[c]["decode_type"] and decode_level(c)]


@pytest.mark.parametrize("name", GREEDY_SMALL)
def test_c_oracle_greedy_matches_reference(name):
    """fp32 folded cache: same trajectories as the reference except fp32 near-tie flips (bounded),
    identical tour lengths on identical trajectories, log-likelihood within 1e-5."""
    g = GoldenCase(name)
    actions, logps, td0 = c_rollout(g, "greedy")
    assert actions.shape == g.actions.shape
    same = (actions == g.actions).all(1)
    flips = int((~same).sum())
    assert flips <= max(1, g.actions.shape[0] // 100), f"{flips} of {len(same)} trajectories differ"
    reward = oracle_reward(g.env_name, td0, actions)
    assert torch.equal(reward[same], g.reward[same])
    torch.testing.assert_close(logps.sum(1)[same], g.log_likelihood[same], rtol=ll_rtol(g.env_name), atol=2e-5)
    # a flipped trajectory is still a valid tour of near-identical quality
    td_rows = R.batchify({k: v for k, v in td0.items() if torch.is_tensor(v)}, g.num_starts) if g.num_starts else td0
    g.env.check_solution_validity(td_rows, actions)
    assert abs(float(reward.mean() - g.reward.mean())) < 5e-3 * abs(float(g.reward.mean()))


@pytest.mark.parametrize("name", [c for c in SMALL_CASES if "sampling" in manifest()[c]["decode_type"] and decode_level(c)])
def test_c_oracle_sampling_matches_reference(name):
    """Sampling parity with a fixed seed: the reference's multinomial stream is one [B,N]
    exponential_ draw per step (checked in gen_golden); fed the same draws the oracle reproduces
    the reference's sampled trajectories (up to near-tie flips) and rewards within 1e-5."""
    g = GoldenCase(name)
    s = g.num_starts
    b = g.batch * max(s, 1)
    n = g.num_loc + (0 if g.env_name == "tsp" else 1)
    steps = g.actions.shape[1] - (1 if s > 0 else 0)
    torch.manual_seed(g.meta["sample_seed"])
    noise = torch.stack([torch.empty(b, n).exponential_(1) for _ in range(max_horizon(g.env_name, n))], 0)
    actions, logps, td0 = c_rollout(g, "sampling", exp_noise=noise.contiguous())
    assert actions.shape[1] >= steps
    t = g.actions.shape[1]
    same = (actions[:, :t] == g.actions).all(1) if actions.shape[1] == t else torch.zeros(b, dtype=torch.bool)
    assert int((~same).sum()) <= max(1, b // 50)
    reward = oracle_reward(g.env_name, td0, actions)
    assert torch.equal(reward[same], g.reward[same])
    torch.testing.assert_close(reward.mean(), g.reward.mean(), rtol=1e-5, atol=0) if bool(same.all()) else None
    torch.testing.assert_close(logps[:, :t].sum(1)[same], g.log_likelihood[same], rtol=ll_rtol(g.env_name), atol=5e-5)


@pytest.mark.parametrize("name", ["c1_tsp20_b256_greedy", "tsp100_b64_greedy", "cvrp20_b128_greedy", "cvrp100_b64_greedy",
                                  "pomo_tsp20_b16_msgreedy", "pomo_cvrp20_b16_msgreedy"])
def test_c_oracle_unfolded_greedy_matches_reference(name):
    """fold = off: the reference's own association (per-step project_context / project_out GEMVs, raw logit key,
    zoo/am/decoder.py:128-228, nn/attention.py:287-293) in a specified order. Same bar as the folded cache — and the
    per-step log-probabilities of the two associations agree to 1e-5: the fold is algebra, not approximation."""
    g = GoldenCase(name)
    actions, logps, td0 = c_rollout(g, "greedy", fold=False)
    assert actions.shape == g.actions.shape
    same = (actions == g.actions).all(1)
    assert int((~same).sum()) <= max(1, g.actions.shape[0] // 100)
    reward = oracle_reward(g.env_name, td0, actions)
    assert torch.equal(reward[same], g.reward[same])
    torch.testing.assert_close(logps.sum(1)[same], g.log_likelihood[same], rtol=ll_rtol(g.env_name), atol=2e-5)
    a_f, l_f, _ = c_rollout(g, "greedy", fold=True)
    both = (a_f == actions).all(1)
    assert both.float().mean() >= 0.98
    torch.testing.assert_close(l_f[both], logps[both], rtol=0, atol=2e-5)


def test_c_oracle_unfolded_sampling_matches_reference():
    g = GoldenCase("cvrp100_b64_sampling")
    n = g.num_loc + 1
    torch.manual_seed(g.meta["sample_seed"])
    noise = torch.stack([torch.empty(g.batch, n).exponential_(1) for _ in range(max_horizon(g.env_name, n))], 0)
    actions, logps, td0 = c_rollout(g, "sampling", exp_noise=noise.contiguous(), fold=False)
    t = g.actions.shape[1]
    same = (actions[:, :t] == g.actions).all(1)
    assert int((~same).sum()) <= 1
    assert torch.equal(oracle_reward(g.env_name, td0, actions)[same], g.reward[same])


def test_c_oracle_bf16_cache_quality():
    """bf16 cache planes (the throughput configuration): trajectories legitimately diverge from
    the fp32 reference; what must hold is validity and tour quality (mean within 1 %)."""
    g = GoldenCase("tsp50_b64_greedy")
    actions, logps, td0 = c_rollout(g, "greedy", cache_dtype=torch.bfloat16)
    g.env.check_solution_validity(td0, actions)
    reward = c_oracle.tour_length(td0["locs"], actions, negate=True)
    assert abs(float(reward.mean() - g.reward.mean())) < 1e-2 * abs(float(g.reward.mean()))


def test_c_oracle_row_group_invariance_of_actions():
    """The summation tree (row groups G) only moves results at the ulp level."""
    g = GoldenCase("tsp20_b64_greedy_simple")
    a2, l2, _ = c_rollout(g, "greedy", row_groups=2)
    a8, l8, _ = c_rollout(g, "greedy", row_groups=8)
    assert (a2 == a8).all(1).float().mean() > 0.98
    torch.testing.assert_close(l2.sum(1), l8.sum(1), rtol=1e-5, atol=1e-5)


def test_philox_noise_is_strictly_positive_and_exponential():
    """In-kernel Exp(1) noise: never 0 / inf / NaN (a zero draw on a masked node made the
    sampling key 0/0), mean and variance of Exp(1)."""
    import ctypes

    h = c_oracle.lib()
    vals = np.array([h.oracle_exp1_noise(ctypes.c_uint64(7), ctypes.c_uint64(s), t, n)
                     for s in range(8) for t in range(64) for n in range(100)], dtype=np.float64)
    assert np.isfinite(vals).all() and (vals > 0).all()
    assert abs(vals.mean() - 1.0) < 0.02 and abs(vals.var() - 1.0) < 0.06
    # the extreme uniform words map strictly inside (0, 1) in fp32
    for k in (0, 2**23 - 1):
        u = np.float32(np.float32(k) + np.float32(0.5)) * np.float32(2.0 ** -23)
        assert 0.0 < float(u) < 1.0


def test_c_oracle_mask_inner_off_matches_restatement():
    """PointerAttention(mask_inner=False): the glimpse attends to every node, so the step list
    holds all N nodes (no compaction) — still the restatement's trajectories / log-likelihoods."""
    g = GoldenCase("tsp20_b64_greedy_simple")
    torch.manual_seed(g.meta["weight_seed"])
    pol = R.AttentionModelPolicy(env_name="tsp", mask_inner=False).eval()
    td0 = g.reset()
    with torch.inference_mode():
        want = pol(g.reset(), g.env, phase="test", decode_type="greedy")
        h, _ = pol.encoder(td0)
    cache = fold_cache(pol, "tsp", h)
    st = rollout_state("tsp", td0)
    actions = torch.zeros(g.batch, 20, dtype=torch.int64)
    logps = torch.zeros(g.batch, 20)
    err = torch.zeros(1, dtype=torch.int32)
    c_oracle.am_decode(cache, st, mode="greedy", max_steps=20, actions=actions, logps=logps, err=err,
                       row_groups=2, mask_inner=False)
    assert int(err.item()) == 0
    same = (actions == want["actions"]).all(1)
    assert int((~same).sum()) <= 1
    torch.testing.assert_close(logps.sum(1)[same], want["log_likelihood"][same], rtol=1e-5, atol=2e-5)


def _random_shapes():
    import random

    rnd = random.Random(20240924)
    out = []
    for env_name in ("tsp", "cvrp", "op", "pctsp", "pdp", "cvrptw"):
        for _ in range(5):
            n = rnd.randint(2, 40) if env_name == "tsp" else rnd.randint(1, 40)
            if env_name == "pdp":
                n = max(2, n + (n % 2))
            out.append((env_name, n, rnd.randint(1, 9), rnd.randint(0, 10**6)))
    return out


@pytest.mark.parametrize("env_name,num_loc,batch,seed", _random_shapes())
def test_random_shapes_env_walks_match_restatement(env_name, num_loc, batch, seed):
    """Thirty random (environment, size, batch) shapes — odd sizes, a single customer, a batch of one: a random
    feasible walk to the end through the C oracle's transition == the restatement's, state and mask bit for bit,
    and the reward composition of the walk == the restatement's get_reward (validity check included)."""
    env, data = make_instances(env_name, num_loc, batch, seed=seed)
    td0 = env.reset(clone_td(data))
    td = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in td0.items()}
    st = rollout_state(env_name, td)
    g = torch.Generator().manual_seed(seed)
    acts = []
    for _ in range(4 * (num_loc + 2)):
        if bool(td["done"].all()):
            break
        probs = td["action_mask"].float()
        probs[td["done"].reshape(-1)] = 0.0
        probs[td["done"].reshape(-1), 0] = 1.0  # finished rows wait at node 0 (the kernels' padding convention)
        action = torch.multinomial(probs, 1, generator=g).squeeze(1)
        if env_name in ("tsp", "pdp", "op", "pctsp"):  # these finish in lock-step or ignore further steps: stop rows exactly
            action = torch.where(td["done"].reshape(-1), td["current_node"].reshape(-1), action)
        td["action"] = action
        live = ~td["done"].reshape(-1).clone()
        if not bool(live.all()) and env_name in ("op", "pctsp"):
            break  # ragged finish: the remaining rows are covered by the lock-step shapes
        td = env.step(td)
        apply_step(c_oracle, env_name, action, st)
        acts.append(action)
        assert torch.equal(st["action_mask"], td["action_mask"])
        assert torch.equal(st["done"], td["done"].reshape(-1))
        assert torch.equal(st["current_node"], td["current_node"].reshape(-1))
    if bool(td["done"].all()) and acts:
        actions = torch.stack(acts, 1)
        assert torch.equal(oracle_reward(env_name, td0, actions), env.get_reward(td0, actions))


@pytest.mark.parametrize("name", ["pomo_tsp20_b16_msgreedy", "pomo_cvrp20_b16_msgreedy"])
def test_ms_rounding_model_oracle_is_the_same_policy_up_to_bf16(name):
    """oracle_am_decode_ms (bf16 query / softmax numerators / glimpse, the multistart MFMA kernel's rounding points)
    against the specified-order oracle on the same bf16 planes: same decode step up to the model's bf16 error —
    teacher-forced per-step log-probs within 0.05, identical greedy decisions almost everywhere."""
    g = GoldenCase(name)
    td0 = g.reset()
    with torch.inference_mode():
        h, _ = g.policy.encoder(td0)
    cache = fold_cache(g.policy, g.env_name, h, torch.bfloat16)
    s = g.num_starts
    outs = {}
    for groups in (4, "ms"):
        st = rollout_state(g.env_name, td0, num_starts=s)
        b, n = st["action_mask"].shape
        tmax = max_horizon(g.env_name, n)
        actions, logps = torch.zeros(b, tmax, dtype=torch.int64), torch.zeros(b, tmax)
        n_steps, err = torch.zeros(b, dtype=torch.int32), torch.zeros(1, dtype=torch.int32)
        first = g.start_nodes(td0, s)
        actions[:, 0] = first
        apply_step(c_oracle, g.env_name, first, st)
        forced = None
        if groups == "ms":  # evaluate the specified-order oracle's trajectories under the rounding model
            forced = outs[4][0].clone()
        c_oracle.am_decode(cache, st, mode="greedy" if forced is None else "evaluate", max_steps=tmax - 1, t0=1,
                           actions=actions, logps=logps, err=err, n_steps=n_steps, row_groups=groups, forced_actions=forced)
        assert int(err.item()) == 0 and bool(st["done"].all())
        outs[groups] = (actions, logps, int(n_steps.max()))
    t = 1 + outs[4][2]
    assert torch.equal(outs["ms"][0][:, :t], outs[4][0][:, :t])
    gap = (outs["ms"][1][:, :t] - outs[4][1][:, :t]).abs()
    assert float(gap.max()) <= 0.05 and float(gap.mean()) <= 5e-3, (float(gap.max()), float(gap.mean()))


# ---------------------------------------------------------------------------------------------
# fp16 planes (the reference's default "16-mixed" regime, utils/trainer.py:57)
# ---------------------------------------------------------------------------------------------

def test_half_to_float_is_exact_for_every_bit_pattern():
    """oracle/rollout_ref.c half_bits_to_float (what the C oracle reads fp16 planes through, = v_cvt_f32_f16 on the
    device) against torch's conversion, all 65 536 patterns: normals, subnormals, zeros, infinities, NaN payloads."""
    import ctypes as C

    h = c_oracle.lib()
    h.oracle_half_to_float_array.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
    bits = torch.arange(65536, dtype=torch.int32).to(torch.int16)
    out = torch.empty(65536, dtype=torch.float32)
    assert h.oracle_half_to_float_array(bits.data_ptr(), 65536, out.data_ptr()) == 0
    want = bits.view(torch.float16).float()
    nan = want.isnan()
    assert torch.equal(out[~nan].view(torch.int32), want[~nan].view(torch.int32))
    assert bool(out[nan].isnan().all())


@pytest.mark.parametrize("name", ["tsp50_b64_greedy", "cvrp20_b128_greedy"])
def test_c_oracle_fp16_planes_track_fp32_planes(name):
    """fp16 planes carry 11 significant bits (bf16: 8): the rollout on them stays closer to the fp32-plane rollout than
    the bf16 one — same tours on most rows, per-step log-probs within 2e-3 where the tours coincide."""
    g = GoldenCase(name)
    a32, l32, _ = c_rollout(g, "greedy")
    a16, l16, _ = c_rollout(g, "greedy", cache_dtype=torch.float16)
    ab, lb, _ = c_rollout(g, "greedy", cache_dtype=torch.bfloat16)
    t = min(a32.shape[1], a16.shape[1], ab.shape[1])
    same16 = (a32[:, :t] == a16[:, :t]).all(1)
    sameb = (a32[:, :t] == ab[:, :t]).all(1)
    assert int(same16.sum()) >= int(sameb.sum()) and float(same16.float().mean()) >= 0.5
    gap = (l16[:, :t] - l32[:, :t])[same16].abs()
    assert float(gap.max()) <= 2e-3


def test_half_round_is_torchs_float_to_half_conversion():
    """oracle half_round (the rounding points of the fp16 MS kernel's model oracle) == torch's fp32 -> fp16 -> fp32:
    normals, ties to even, subnormals, overflow to infinity, signed zeros."""
    import ctypes as C

    h = c_oracle.lib()
    h.oracle_half_round_array.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
    torch.manual_seed(0)
    x = torch.cat([torch.randn(200000) * 10.0 ** torch.randint(-9, 6, (200000,)).float(),
                   torch.tensor([0.0, -0.0, 65504.0, 65519.9, 65520.0, 70000.0, -65520.0, 6.1e-5, 6.0e-5, 5.96e-8, 2.98e-8, 2.99e-8,
                                 1.0 + 2.0 ** -11, 1.0 + 3 * 2.0 ** -11, 1.0 + 2.0 ** -11 + 2.0 ** -20, float("inf"), -float("inf")])])
    x = x.contiguous()
    out = torch.empty_like(x)
    assert h.oracle_half_round_array(x.data_ptr(), x.numel(), out.data_ptr()) == 0
    want = x.half().float()
    assert torch.equal(out.view(torch.int32), want.view(torch.int32)), (x[out != want][:5], out[out != want][:5], want[out != want][:5])
