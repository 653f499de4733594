"""GPU parity: fused AttentionModel decode / rollout kernel vs the torch oracle.

The kernel is fed the oracle's own encoder output ``h`` (so this file isolates the decode
path); end-to-end policy parity lives in test_gpu_policy.py.

Tolerances (north_star): greedy tour lengths bit-identical wherever the action sequences are
identical; action sequences may differ from the stock-ATen oracle only through fp32 near-ties
(SURVEY.md §7 "hard parts": no fixed reduction order is bitwise equal to ATen's SDPA/GEMM), so
the trajectory flip rate is bounded (<= 0.5 % of instances) instead of required to be zero;
log-likelihoods within 1e-5 relative; sampled mean reward within 1e-5 relative.
"""
import pytest
import torch

from oracle import reference_torch as R
from tests.helpers import clone_td, decoder_weights, device_state, make_instances, make_policy

pytestmark = pytest.mark.gpu

MAX_FLIP_FRACTION = 0.005
LL_RTOL = 1e-5


@pytest.fixture(scope="module")
def K():
    from rl4co_amd import kernels

    return kernels


def _fold(pol, env_name, h, dtype=torch.float32):
    from rl4co_amd.cache import build_folded_cache

    w = {k: (v.detach().cuda() if v is not None else None) for k, v in decoder_weights(pol).items()}
    return build_folded_cache(env_name, h.cuda(), cache_dtype=dtype, **w)


def _hip_rollout(K, cache, env_name, td_reset, mode, tmax, **kw):
    st = device_state(env_name, td_reset, "cuda")
    b = st["action_mask"].shape[0]
    actions = torch.zeros((b, tmax), dtype=torch.int64, device="cuda")
    logps = torch.zeros((b, tmax), dtype=torch.float32, device="cuda")
    n_steps = torch.zeros((b,), dtype=torch.int32, device="cuda")
    err = K.new_error_word("cuda")
    K.am_decode(cache, st, mode=mode, max_steps=tmax, actions=actions, logps=logps, err=err,
                n_steps=n_steps, **kw)
    torch.cuda.synchronize()
    K.raise_if_error(err)
    t = int(n_steps.max().item())
    return actions[:, :t].cpu(), logps[:, :t].cpu(), st, n_steps.cpu()


def _compare(env, data_td, out_ref, actions, logps, K, env_name, max_flip=MAX_FLIP_FRACTION):
    ref_actions = out_ref["actions"]
    assert actions.shape == ref_actions.shape, (actions.shape, ref_actions.shape)
    same = (actions == ref_actions).all(dim=1)
    flip = 1.0 - same.float().mean().item()
    assert flip <= max_flip, f"trajectory flip rate {flip:.4%}"
    # reward of the kernel's own actions: bit-exact against the oracle's get_reward on them
    locs = data_td["locs"]
    got_reward = K.tour_length(locs.cuda(), actions.cuda(), prepend_depot=(env_name == "cvrp"), negate=True).cpu()
    want_reward = env.get_reward(data_td, actions)
    assert torch.equal(got_reward, want_reward)
    # identical trajectories => bit-identical tour lengths vs the reference rollout
    assert torch.equal(got_reward[same], out_ref["reward"][same])
    ll = logps.sum(1)
    torch.testing.assert_close(ll[same], out_ref["log_likelihood"][same], rtol=LL_RTOL, atol=1e-5)
    return flip


@pytest.mark.parametrize("sdpa", ["default", "simple"])
@pytest.mark.parametrize("num_loc,batch", [(20, 256), (50, 128), (100, 96)])
def test_tsp_greedy_fp32(K, num_loc, batch, sdpa):
    pol = make_policy("tsp", sdpa_fn=sdpa)
    env, data = make_instances("tsp", num_loc, batch)
    with torch.inference_mode():
        td0 = env.reset(clone_td(data))
        out = pol(clone_td(td0), env, phase="test")
        h, _ = pol.encoder(td0)
    cache = _fold(pol, "tsp", h)
    actions, logps, st, n_steps = _hip_rollout(K, cache, "tsp", td0, "greedy", num_loc)
    assert (n_steps == num_loc).all()
    assert st["done"].all() and not st["action_mask"].any()
    _compare(env, td0, out, actions, logps, K, "tsp")


@pytest.mark.parametrize("num_loc,batch", [(20, 128), (100, 64)])
def test_cvrp_greedy_fp32(K, num_loc, batch):
    pol = make_policy("cvrp")
    env, data = make_instances("cvrp", num_loc, batch)
    with torch.inference_mode():
        td0 = env.reset(clone_td(data))
        out = pol(clone_td(td0), env, phase="test")
        h, _ = pol.encoder(td0)
    cache = _fold(pol, "cvrp", h)
    actions, logps, st, n_steps = _hip_rollout(K, cache, "cvrp", td0, "greedy", 2 * num_loc + 2)
    assert st["done"].all()
    _compare(env, td0, out, actions, logps, K, "cvrp")
    # finished rows keep emitting the depot with log-prob 0 (cvrp/env.py:135)
    t = actions.shape[1]
    for b in range(actions.shape[0]):
        nb = int(n_steps[b])
        assert (actions[b, nb:t] == 0).all() and (logps[b, nb:t] == 0).all()


@pytest.mark.parametrize("env_name,num_loc,batch", [("tsp", 20, 256), ("tsp", 100, 64), ("cvrp", 50, 64)])
def test_sampling_with_injected_noise(K, env_name, num_loc, batch):
    """multinomial(p,1) == argmax(p / Exp(1)): feed the oracle's draws to the kernel."""
    pol = make_policy(env_name)
    env, data = make_instances(env_name, num_loc, batch)
    rec = []
    with torch.inference_mode():
        td0 = env.reset(clone_td(data))
        torch.manual_seed(77)
        out = pol(clone_td(td0), env, phase="train", noise_recorder=rec)
        h, _ = pol.encoder(td0)
    noise = torch.stack(rec, 0).cuda().contiguous()  # [T,B,N]
    cache = _fold(pol, env_name, h)
    t = noise.shape[0]
    actions, logps, st, _ = _hip_rollout(K, cache, env_name, td0, "sampling", t, exp_noise=noise)
    flip = _compare(env, td0, out, actions, logps, K, env_name, max_flip=0.01)
    got = K.tour_length(td0["locs"].cuda(), actions.cuda(), prepend_depot=(env_name == "cvrp"), negate=True).cpu()
    rel = abs(got.mean().item() - out["reward"].mean().item()) / abs(out["reward"].mean().item())
    assert rel <= 1e-5 or flip > 0, rel


def test_sampling_noise_matches_multinomial():
    """The oracle's explicit exponential race equals torch.multinomial bit-for-bit (same seed)."""
    pol = make_policy("tsp")
    env, data = make_instances("tsp", 20, 64)
    with torch.inference_mode():
        td0 = env.reset(clone_td(data))
        torch.manual_seed(5)
        a = pol(clone_td(td0), env, phase="train")
        torch.manual_seed(5)
        b = pol(clone_td(td0), env, phase="train", noise_recorder=[])
    assert torch.equal(a["actions"], b["actions"])


@pytest.mark.parametrize("env_name,num_loc", [("tsp", 50), ("cvrp", 50)])
def test_evaluate_mode_logps(K, env_name, num_loc):
    """decode_type='evaluate' (decoding.py:448-461): forced actions, log-probs within tolerance."""
    pol = make_policy(env_name)
    env, data = make_instances(env_name, num_loc, 64)
    with torch.inference_mode():
        td0 = env.reset(clone_td(data))
        torch.manual_seed(3)
        out = pol(clone_td(td0), env, phase="train", return_sum_log_likelihood=False)
        h, _ = pol.encoder(td0)
    cache = _fold(pol, env_name, h)
    forced = out["actions"].cuda().contiguous()
    t = forced.shape[1]
    actions, logps, st, _ = _hip_rollout(K, cache, env_name, td0, "evaluate", t, forced_actions=forced)
    assert torch.equal(actions, out["actions"])
    torch.testing.assert_close(logps, out["log_likelihood"], rtol=1e-4, atol=2e-6)


def test_tsp_bf16_cache_matches_oracle_on_quantised_cache(K):
    """bf16 cache variant: parity is against the oracle evaluated on the SAME bf16-rounded cache
    (cached tensors replaced by their bf16 round trip), fp32 arithmetic on both sides."""
    num_loc, batch = 50, 128
    pol = make_policy("tsp", sdpa_fn="simple")
    env, data = make_instances("tsp", num_loc, batch)
    from rl4co_amd.cache import build_folded_cache

    with torch.inference_mode():
        td0 = env.reset(clone_td(data))
        h, _ = pol.encoder(td0)
    w = {k: (v.detach().cuda() if v is not None else None) for k, v in decoder_weights(pol).items()}
    cache16 = build_folded_cache("tsp", h.cuda(), cache_dtype=torch.bfloat16, **w)
    cache32 = build_folded_cache("tsp", h.cuda(), cache_dtype=torch.float32, **w)
    cache32.kvl = cache16.kvl.float().contiguous()
    a16, l16, _, _ = _hip_rollout(K, cache16, "tsp", td0, "greedy", num_loc)
    a32, l32, _, _ = _hip_rollout(K, cache32, "tsp", td0, "greedy", num_loc)
    same = (a16 == a32).all(1)
    assert same.float().mean() >= 0.99
    torch.testing.assert_close(l16[same], l32[same], rtol=1e-4, atol=1e-5)


def test_single_step_api_matches_rollout(K):
    """max_steps=1 called T times (the RL4COEnvBase.step-style loop) == one persistent launch."""
    num_loc, batch = 20, 64
    pol = make_policy("tsp")
    env, data = make_instances("tsp", num_loc, batch)
    with torch.inference_mode():
        td0 = env.reset(clone_td(data))
        h, _ = pol.encoder(td0)
    cache = _fold(pol, "tsp", h)
    a_ref, l_ref, _, _ = _hip_rollout(K, cache, "tsp", td0, "greedy", num_loc)
    st = device_state("tsp", td0, "cuda")
    actions = torch.zeros((batch, num_loc), dtype=torch.int64, device="cuda")
    logps = torch.zeros((batch, num_loc), dtype=torch.float32, device="cuda")
    err = K.new_error_word("cuda")
    for t in range(num_loc):
        K.am_decode(cache, st, mode="greedy", max_steps=1, t0=t, actions=actions, logps=logps, err=err)
    K.raise_if_error(err)
    assert torch.equal(actions.cpu(), a_ref) and torch.equal(logps.cpu(), l_ref)


def test_multistart_shares_cache(K):
    """POMO layout: S*B trajectories (s-major) read B cache rows; first action forced per start."""
    num_loc, batch, starts = 20, 16, 20
    pol = make_policy("tsp", pomo=True)
    env, data = make_instances("tsp", num_loc, batch)
    with torch.inference_mode():
        td0 = env.reset(clone_td(data))
        out = pol(clone_td(td0), env, phase="test", num_starts=starts)
        h, _ = pol.encoder(td0)
    cache = _fold(pol, "tsp", h)
    # pre_decoder_hook (decoding.py:282-330): batchify, forced first action through env.step
    tdb = R.batchify(clone_td(td0), starts)
    st = device_state("tsp", tdb, "cuda")
    first = K.select_start_nodes(batch, starts, num_loc, False, "cuda")
    err = K.new_error_word("cuda")
    K.tsp_step(first, st["action_mask"], st["first_node"], st["current_node"], st["i"], st["done"], err)
    b = batch * starts
    actions = torch.zeros((b, num_loc), dtype=torch.int64, device="cuda")
    logps = torch.zeros((b, num_loc), dtype=torch.float32, device="cuda")
    actions[:, 0] = first
    K.am_decode(cache, st, mode="greedy", max_steps=num_loc - 1, t0=1, actions=actions, logps=logps, err=err)
    K.raise_if_error(err)
    same = (actions.cpu() == out["actions"]).all(1)
    assert same.float().mean() >= 1 - MAX_FLIP_FRACTION
    got = K.tour_length(td0["locs"].cuda(), actions, negate=True).cpu()
    assert torch.equal(got[same], out["reward"][same])


def test_error_bits_surface_reference_assertions(K):
    """A forced infeasible action must raise the reference's message once, after the rollout."""
    num_loc, batch = 10, 8
    pol = make_policy("tsp")
    env, data = make_instances("tsp", num_loc, batch)
    with torch.inference_mode():
        td0 = env.reset(clone_td(data))
        h, _ = pol.encoder(td0)
    cache = _fold(pol, "tsp", h)
    forced = torch.zeros((batch, num_loc), dtype=torch.int64, device="cuda")  # node 0 again and again
    with pytest.raises(AssertionError, match="infeasible action selected"):
        _hip_rollout(K, cache, "tsp", td0, "evaluate", num_loc, forced_actions=forced)
