"""Edge cases of the path: tiny and large graphs, batch of one, multistart wrap-around, ragged
CVRP horizons — CPU (C oracle vs the torch restatement) and GPU (HIP vs C oracle, bit-exact)."""
import pytest
import torch

from oracle import c_oracle
from oracle import reference_torch as R
from tests.helpers import (clone_td, fold_cache, kernel_reward, make_instances, make_policy, max_horizon, oracle_reward,
                           rollout_state)


def _setup(env_name, num_loc, batch, seed=7):
    pol = make_policy(env_name, seed=3)
    env, data = make_instances(env_name, num_loc, batch, seed=seed)
    td0 = env.reset(clone_td(data))
    with torch.inference_mode():
        h, _ = pol.encoder(td0)
    return pol, env, td0, h


def _c_rollout(pol, env_name, td0, h, dtype=torch.float32, variant_groups=None, cache=None):
    cache = cache or fold_cache(pol, env_name, h, dtype)
    st = rollout_state(env_name, td0)
    b, n = st["action_mask"].shape
    tmax = max_horizon(env_name, n)
    actions = torch.zeros(b, tmax, dtype=torch.int64)
    logps = torch.zeros(b, tmax)
    n_steps = torch.zeros(b, dtype=torch.int32)
    err = torch.zeros(1, dtype=torch.int32)
    groups = variant_groups or (4 if dtype == torch.bfloat16 else 2)
    c_oracle.am_decode(cache, st, mode="greedy", max_steps=tmax, actions=actions, logps=logps, err=err,
                       n_steps=n_steps, row_groups=groups)
    t = int(n_steps.max())
    return actions[:, :t].contiguous(), logps[:, :t], st, n_steps, int(err.item())


@pytest.mark.parametrize("env_name,num_loc,batch", [("tsp", 2, 5), ("tsp", 3, 4), ("tsp", 7, 9), ("cvrp", 1, 6),
                                                    ("cvrp", 2, 6), ("cvrp", 9, 1), ("tsp", 20, 1), ("op", 1, 5),
                                                    ("op", 3, 7), ("pctsp", 1, 5), ("pctsp", 4, 3), ("pdp", 2, 6),
                                                    ("pdp", 4, 1), ("cvrptw", 1, 5), ("cvrptw", 3, 4)])
def test_tiny_graphs_cpu(env_name, num_loc, batch):
    """N = 2, 3 nodes, a single customer, a batch of one: the C oracle reproduces the
    restatement's greedy rollout (valid tours, same rewards on identical trajectories)."""
    pol, env, td0, h = _setup(env_name, num_loc, batch)
    with torch.inference_mode():
        want = pol(clone_td(td0), env, phase="test", decode_type="greedy")
    actions, logps, st, n_steps, err = _c_rollout(pol, env_name, td0, h)
    assert err == 0 and bool(st["done"].all())
    assert actions.shape == want["actions"].shape
    same = (actions == want["actions"]).all(1)
    assert bool(same.all()) or int((~same).sum()) <= 1
    reward = oracle_reward(env_name, td0, actions)
    assert torch.equal(reward[same], want["reward"][same])
    env.check_solution_validity(td0, actions)


def test_ragged_cvrp_horizons_cpu():
    """Rows finish at different steps; finished rows emit depot / log-prob 0 up to the global
    horizon exactly like the reference's `while not done.all()` loop."""
    pol, env, td0, h = _setup("cvrp", 20, 32, seed=11)
    with torch.inference_mode():
        want = pol(clone_td(td0), env, phase="test", decode_type="greedy")
    actions, logps, st, n_steps, err = _c_rollout(pol, "cvrp", td0, h)
    assert len(set(n_steps.tolist())) > 1, "test needs ragged horizons"
    assert actions.shape == want["actions"].shape
    for r in range(actions.shape[0]):
        assert (actions[r, int(n_steps[r]):] == 0).all() and (logps[r, int(n_steps[r]):] == 0).all()
    same = (actions == want["actions"]).all(1)
    assert same.float().mean() > 0.9
    torch.testing.assert_close(logps.sum(1)[same], want["log_likelihood"][same], rtol=1e-5, atol=2e-5)


def test_multistart_more_starts_than_nodes_wraps():
    """select_start_nodes: start s uses node s % num_loc (+1 with a depot), ops.py:128-161."""
    class E:
        name = "tsp"
        num_loc = 5

    td = {"action_mask": torch.ones(3, 5, dtype=torch.bool)}
    got = R.select_start_nodes(td, E, 12)
    assert got.tolist() == [s % 5 for s in range(12) for _ in range(3)]


# ---------------------------------------------------------------------------------------------
# GPU: HIP == C oracle on shapes the goldens do not cover
# ---------------------------------------------------------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize("env_name,num_loc,batch,dtype,variant", [
    ("tsp", 2, 5, torch.float32, "stream"), ("tsp", 3, 70, torch.bfloat16, "stream"),
    ("tsp", 3, 4, torch.bfloat16, "lds"), ("cvrp", 1, 6, torch.bfloat16, "wide"),
    ("cvrp", 2, 6, torch.float32, "stream"), ("tsp", 63, 9, torch.bfloat16, "lds"),
    ("tsp", 64, 9, torch.bfloat16, "stream"), ("tsp", 65, 9, torch.bfloat16, "wide"),
    ("cvrp", 200, 12, torch.bfloat16, "wide"), ("tsp", 1000, 3, torch.bfloat16, "stream"),
    ("tsp", 1000, 3, torch.float32, "stream"), ("cvrp", 127, 5, torch.bfloat16, "stream"),
    ("op", 1, 5, torch.float32, "stream"), ("op", 63, 9, torch.bfloat16, "stream"), ("op", 300, 4, torch.bfloat16, "stream"),
    ("pctsp", 1, 5, torch.float32, "stream"), ("pctsp", 64, 9, torch.bfloat16, "stream"),
    ("pctsp", 300, 4, torch.float32, "stream"), ("pdp", 2, 6, torch.float32, "stream"), ("pdp", 64, 9, torch.bfloat16, "stream"),
    ("pdp", 300, 3, torch.bfloat16, "stream"), ("cvrptw", 1, 5, torch.float32, "stream"),
    ("cvrptw", 65, 9, torch.bfloat16, "stream"), ("cvrptw", 200, 3, torch.float32, "stream"),
])
def test_shapes_bit_exact_gpu(env_name, num_loc, batch, dtype, variant):
    from rl4co_amd import kernels as K

    pol, env, td0, h = _setup(env_name, num_loc, batch)
    cache = fold_cache(pol, env_name, h, dtype, device="cuda")
    cache_cpu = type(cache)(cache.env_name, *(None if x is None else x.cpu().contiguous() for x in (
        cache.kvl, cache.ctx_first, cache.ctx_cur, cache.q_bias, cache.q_step0, cache.w_cap, cache.w_time)))
    n = td0["action_mask"].shape[1]
    tmax = max_horizon(env_name, n)
    groups = K.decode_row_groups(n, dtype, tmax, variant, batch)
    a_c, l_c, st_c, n_c, err_c = _c_rollout(pol, env_name, td0, h, dtype, groups, cache_cpu)
    st = rollout_state(env_name, td0, device="cuda")
    actions = torch.zeros(batch, tmax, dtype=torch.int64, device="cuda")
    logps = torch.zeros(batch, tmax, device="cuda")
    n_steps = torch.zeros(batch, dtype=torch.int32, device="cuda")
    err = K.new_error_word("cuda")
    K.am_decode(cache, st, mode="greedy", max_steps=tmax, actions=actions, logps=logps, err=err, n_steps=n_steps,
                variant=variant)
    torch.cuda.synchronize()
    t = int(n_steps.max())
    assert int(err.item()) == err_c == 0
    assert torch.equal(n_steps.cpu(), n_c)
    assert torch.equal(actions[:, :t].cpu(), a_c)
    assert torch.equal(logps[:, :t].cpu().view(torch.int32), l_c.contiguous().view(torch.int32))
    for k in st_c:
        assert torch.equal(st[k].cpu(), st_c[k]), k
    reward = kernel_reward(K, env_name, td0, actions[:, :t].contiguous()).cpu()
    assert torch.equal(reward, env.get_reward(td0, a_c))


@pytest.mark.gpu
def test_policy_batch_of_one_and_odd_batches_gpu():
    """Batch sizes that are not multiples of 8 (XCD map falls back to the identity) incl. B = 1."""
    from rl4co_amd.envs import get_env
    from rl4co_amd.policy import AttentionModelPolicy

    torch.manual_seed(0)
    pol = AttentionModelPolicy("tsp", cache_dtype=torch.bfloat16, encoder_autocast=torch.bfloat16).cuda().eval()
    env = get_env("tsp", generator_params=dict(num_loc=30, device="cuda"), device="cuda")
    for b in (1, 3, 13):
        td = env.reset(batch_size=[b])
        with torch.inference_mode():
            out = pol(td, env, phase="test")
            ms = pol(td, env, phase="test", decode_type="multistart_greedy", num_starts=7)
        assert out["actions"].shape == (b, 30) and out["reward"].shape == (b,)
        assert ms["actions"].shape == (7 * b, 30)
        assert bool((ms["reward"].view(7, b).max(0).values >= out["reward"] - 2.0).all())
