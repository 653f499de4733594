"""GPU (`-m gpu`): the rollout captured as one HIP graph (rl4co_amd/graph.py) == the eagerly launched rollout.

The graph is the same launches with the same arguments: greedy outputs must be bit-identical to the eager path
(actions, rewards, log-likelihoods, also for CVRP whose horizon the reward kernel reads on the device); sampling must
draw fresh noise on every replay; new instances and new weight values must be picked up without a new capture."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(env_name, num_loc, batch, **pol_kw):
    from rl4co_amd.envs import get_env
    from rl4co_amd.policy import AttentionModelPolicy

    torch.manual_seed(0)
    pol = AttentionModelPolicy(env_name, cache_dtype=torch.bfloat16, encoder_autocast=torch.bfloat16, **pol_kw).cuda().eval()
    env = get_env(env_name, generator_params=dict(num_loc=num_loc, device="cuda"), device="cuda")
    torch.manual_seed(1)
    return pol, env, env.generator(batch_size=[batch]), env.generator(batch_size=[batch])


@pytest.mark.parametrize("env_name,num_loc", [("tsp", 50), ("cvrp", 50), ("pdp", 20), ("cvrptw", 20)])
def test_graphed_greedy_rollout_equals_eager(env_name, num_loc):
    from rl4co_amd.graph import GraphedRollout

    pol, env, d1, d2 = _setup(env_name, num_loc, 256)
    g = GraphedRollout(pol, env, d1, decode_type="greedy")
    for data in (d1, d2, d1):
        out = g(data)
        got = {k: out[k].clone() for k in ("actions", "reward", "log_likelihood")}
        with torch.inference_mode():
            want = pol(env.reset(data), env, phase="test", decode_type="greedy")
        assert got["actions"].shape == want["actions"].shape
        assert torch.equal(got["actions"], want["actions"])
        assert torch.equal(got["reward"], want["reward"])
        assert torch.equal(got["log_likelihood"], want["log_likelihood"])
    with pytest.raises(ValueError):
        g(env.generator(batch_size=[128]))


def test_graphed_sampling_draws_fresh_noise_and_follows_weight_updates():
    from rl4co_amd.graph import GraphedRollout

    pol, env, d1, _ = _setup("tsp", 50, 256)
    g = GraphedRollout(pol, env, d1, decode_type="sampling")
    a1 = g(d1)["actions"].clone()
    a2 = g(d1)["actions"].clone()
    assert not torch.equal(a1, a2), "two replays sampled the same tours: the seed word is not reaching the kernel"
    assert torch.equal(a1.sort(1).values, torch.arange(50, device="cuda").expand_as(a1))
    gg = GraphedRollout(pol, env, d1, decode_type="greedy")
    before = gg(d1)["reward"].clone()
    with torch.no_grad():
        for p in pol.encoder.parameters():
            p.mul_(1.05)
    after = gg(d1)["reward"].clone()
    with torch.inference_mode():
        want = pol(env.reset(d1), env, phase="test", decode_type="greedy")["reward"]
    assert not torch.equal(before, after) and torch.equal(after, want)


def test_device_horizon_reward_equals_host_horizon_reward():
    """kernels.tour_length(horizon=): the padded CVRP action buffer with the step count read on the device gives the
    bits of the reward computed from the sliced buffer (the association of ATen's row sum depends on the length)."""
    from rl4co_amd import kernels as K

    torch.manual_seed(0)
    locs = torch.rand(64, 101, 2, device="cuda")
    acts = torch.randint(0, 101, (64, 202), device="cuda")
    for t in (7, 8, 100, 117, 150, 202):
        steps = torch.tensor([t - 1], dtype=torch.int32, device="cuda")
        padded = acts.clone()
        padded[:, t:] = 0
        want = K.tour_length(locs, padded[:, :t].contiguous(), prepend_depot=True, negate=True)
        got = K.tour_length(locs, padded, prepend_depot=True, negate=True, horizon=(steps, 1))
        assert torch.equal(got, want), t
