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


def test_graphed_rollout_survives_an_eager_rollout_after_a_weight_update():
    """graphed call, optimizer step, EAGER rollout (which rebinds the packed encoder to fresh
    tensors), graphed call again — the replay must read the new weights from the buffers it captured, not freed or
    stale memory. Also through a train()/eval() toggle, which is part of the packed encoder's version."""
    from rl4co_amd.graph import GraphedRollout

    pol, env, d1, d2 = _setup("tsp", 50, 256)
    g = GraphedRollout(pol, env, d1, decode_type="greedy")
    assert g._fused
    first = g(d1)["reward"].clone()
    for round_ in range(3):
        with torch.no_grad():
            for p in pol.parameters():
                p.mul_(1.03)
        if round_ == 1:
            pol.train()
            pol.eval()
        with torch.inference_mode():
            eager = {k: v.clone() for k, v in pol(env.reset(d1), env, phase="test", decode_type="greedy").items()
                     if k in ("actions", "reward")}
        filler = [torch.empty(1 << 22, device="cuda").normal_() for _ in range(8)]  # recycle whatever was freed
        out = g(d1)
        assert torch.equal(out["actions"], eager["actions"]) and torch.equal(out["reward"], eager["reward"]), round_
        del filler
    assert not torch.equal(first, out["reward"])
    # a second eager rollout must keep working on the (captured) buffers the packed encoder now points at
    with torch.inference_mode():
        again = pol(env.reset(d2), env, phase="test", decode_type="greedy")["reward"]
    assert torch.equal(g(d2)["reward"], again)


def test_graphed_rollout_leaves_the_packed_encoder_alone_when_the_fused_path_is_not_taken():
    """Torch encoder inside the graph (fused_encoder=False): no packed encoder is built; weight updates are read live."""
    from rl4co_amd.envs import get_env
    from rl4co_amd.graph import GraphedRollout
    from rl4co_amd.policy import AttentionModelPolicy

    torch.manual_seed(0)
    pol = AttentionModelPolicy("tsp", normalization="layer", fused_encoder=False).cuda().eval()
    env = get_env("tsp", generator_params=dict(num_loc=20, device="cuda"), device="cuda")
    torch.manual_seed(1)
    data = env.generator(batch_size=[64])
    g = GraphedRollout(pol, env, data, decode_type="greedy")
    assert not g._fused and (pol._packed is None or not pol._packed.t)
    before = g(data)["reward"].clone()
    with torch.no_grad():
        for p in pol.encoder.parameters():
            p.mul_(1.1)
    with torch.inference_mode():
        want = pol(env.reset(data), env, phase="test", decode_type="greedy")["reward"]
    after = g(data)["reward"]
    assert torch.equal(after, want) and not torch.equal(before, after)


@pytest.mark.parametrize("env_name,num_loc", [("tsp", 50), ("cvrp", 50)])
def test_pipelined_rollouts_equal_eager_over_a_stream_of_batches(env_name, num_loc):
    """PipelinedRollout: two captured rollouts in flight on two streams; outputs in submission order, bit-identical to
    the eager rollout of each batch; explicit tickets; a weight update in the middle of the stream is picked up."""
    from rl4co_amd.graph import PipelinedRollout

    pol, env, d1, d2 = _setup(env_name, num_loc, 256)
    torch.manual_seed(3)
    batches = [d1, d2] + [env.generator(batch_size=[256]) for _ in range(5)]
    with torch.inference_mode():
        want = [{k: v.clone() for k, v in pol(env.reset(b), env, phase="test", decode_type="greedy").items() if k in ("actions", "reward")}
                for b in batches]
    pipe = PipelinedRollout(pol, env, d1, decode_type="greedy", depth=2)
    for i, out in enumerate(pipe.map(batches)):
        assert torch.equal(out["actions"], want[i]["actions"]) and torch.equal(out["reward"], want[i]["reward"]), i
    t0 = pipe.submit(batches[0])
    t1 = pipe.submit(batches[1])
    with pytest.raises(RuntimeError):
        pipe.submit(batches[2])  # both slots hold uncollected results
    assert torch.equal(pipe.collect(t0)["reward"], want[0]["reward"])
    assert torch.equal(pipe.collect(t1)["reward"], want[1]["reward"])
    with torch.no_grad():
        for p in pol.parameters():
            p.mul_(1.02)
    with torch.inference_mode():
        new = [pol(env.reset(b), env, phase="test", decode_type="greedy")["reward"].clone() for b in batches[:3]]
    got = [out["reward"].clone() for out in pipe.map(batches[:3])]
    assert all(torch.equal(a, b) for a, b in zip(got, new)) and not torch.equal(new[0], want[0]["reward"])


def test_pipelined_fixed_seed_sampling_never_repeats_noise_across_slots():
    """with ``seed=`` both slots capture the same Philox key; the per-replay seed word must come from
    ONE submission counter, or batches 2k and 2k + 1 replay bit-identical noise (best-of-N sampling of one batch would
    return its samples in duplicated pairs)."""
    from rl4co_amd.graph import PipelinedRollout

    pol, env, d1, _ = _setup("tsp", 50, 256)
    pipe = PipelinedRollout(pol, env, d1, decode_type="sampling", depth=2, seed=1234)
    tours = [out["actions"].clone() for out in pipe.map([d1] * 6)]
    for i in range(len(tours)):
        for j in range(i + 1, len(tours)):
            assert not torch.equal(tours[i], tours[j]), (i, j)


@pytest.mark.parametrize("env_name,num_loc", [("cvrp", 50), ("op", 20)])
def test_pipelined_outputs_are_handed_over_to_the_callers_stream(env_name, num_loc):
    """finish() launches on the slot's side stream after its read-back (trimmed actions, summed
    log-likelihood, OP's reward): the caller consumes them on ITS stream right away, with memory churn in between."""
    from rl4co_amd.graph import PipelinedRollout

    pol, env, d1, d2 = _setup(env_name, num_loc, 512)
    torch.manual_seed(5)
    batches = [d1, d2] + [env.generator(batch_size=[512]) for _ in range(6)]
    with torch.inference_mode():
        want = [{k: v.clone() for k, v in pol(env.reset(b), env, phase="test", decode_type="greedy").items()
                 if k in ("actions", "reward", "log_likelihood")} for b in batches]
    pipe = PipelinedRollout(pol, env, d1, decode_type="greedy", depth=2)
    sums = []
    for out in pipe.map(batches):
        # consumed on the current (default) stream immediately, no synchronize; allocations recycle freed blocks
        sums.append((out["log_likelihood"].double().sum() + out["reward"].double().sum() + out["actions"].double().sum()).clone())
        junk = [torch.full((1 << 20,), float("nan"), device="cuda") for _ in range(4)]
        del junk
    for i, s in enumerate(sums):
        w = want[i]["log_likelihood"].double().sum() + want[i]["reward"].double().sum() + want[i]["actions"].double().sum()
        assert torch.equal(s, w), i


def test_graphed_rollout_survives_an_eager_rollout_in_the_other_16bit_regime():
    """an eager fp16 rollout between replays of a bf16 capture re-packs the weights in fp16; the next
    replay must bring the packed encoder back to bf16 without claiming a layout change."""
    from rl4co_amd.graph import GraphedRollout

    pol, env, d1, d2 = _setup("tsp", 50, 256)
    g = GraphedRollout(pol, env, d1, decode_type="greedy")
    want = g(d1)["reward"].clone()
    pol.encoder_autocast, pol.cache_dtype = torch.float16, torch.float16
    with torch.inference_mode():
        pol(env.reset(d2), env, phase="test", decode_type="greedy")
    pol.encoder_autocast, pol.cache_dtype = torch.bfloat16, torch.bfloat16
    assert torch.equal(g(d1)["reward"], want)
