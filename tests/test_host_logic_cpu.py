"""CPU coverage of the host-side mirror of the reference interface (no GPU).

The arithmetic kernels are replaced by the C oracle through tests/fake_device.py, so these tests
exercise exactly the Python the product runs on the GPU box: env reset/step/get_reward surfaces,
policy.forward orchestration for every decode type, state_dict compatibility with the reference,
the teacher-forced differentiable log-likelihood.
"""
import pytest
import torch

from oracle import reference_torch as R
from tests.fake_device import cpu_device  # noqa: F401  (fixture)
from tests.helpers import ll_rtol  # noqa: E402
from tests.helpers import GoldenCase, clone_td, manifest

SMALL = sorted(c for c, m in manifest().items() if m["batch"] <= 128)


def _product_policy(g: GoldenCase, **kw):
    from rl4co_amd.policy import AttentionModelPolicy

    pk = dict(g.meta["policy_kwargs"])
    pk.pop("sdpa_fn_decoder", None)
    pol = AttentionModelPolicy(env_name=g.env_label, **pk, **kw).eval()
    missing = pol.load_state_dict(g.policy.state_dict(), strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return pol


def _product_env(g: GoldenCase):
    from rl4co_amd.envs import get_env

    return get_env(g.env_label, generator_params=dict(num_loc=g.num_loc), device="cpu")


def _product_td(g: GoldenCase):
    from rl4co_amd.tensordict import TensorDict

    return TensorDict({k: v.clone() for k, v in g.data.items()}, batch_size=[g.batch])


def test_state_dict_keys_equal_reference():
    """Checkpoints of the reference's AttentionModelPolicy load unchanged (same keys and shapes)."""
    from rl4co_amd.policy import AttentionModelPolicy

    for env_name in ("tsp", "cvrp"):
        ref = R.AttentionModelPolicy(env_name)
        ours = AttentionModelPolicy(env_name)
        assert list(ref.state_dict().keys()) == list(ours.state_dict().keys())
        for k, v in ref.state_dict().items():
            assert ours.state_dict()[k].shape == v.shape, k
    ref = R.pomo_policy("tsp")
    ours = AttentionModelPolicy("tsp", num_encoder_layers=6, normalization="instance", use_graph_context=False)
    assert list(ref.state_dict().keys()) == list(ours.state_dict().keys())


def test_seeded_construction_matches_reference_weights():
    from rl4co_amd.policy import AttentionModelPolicy

    torch.manual_seed(0)
    ref = R.AttentionModelPolicy("cvrp")
    torch.manual_seed(0)
    ours = AttentionModelPolicy("cvrp")
    for k, v in ref.state_dict().items():
        assert torch.equal(ours.state_dict()[k], v), k


@pytest.mark.parametrize("name", SMALL)
def test_policy_forward_matches_reference_golden(cpu_device, name):
    g = GoldenCase(name)
    pol, env = _product_policy(g), _product_env(g)
    td = env.reset(_product_td(g))
    kw = dict(g.meta["forward_kwargs"])
    if "sampling" in g.meta["decode_type"]:
        b = g.rollout_rows
        n = g.num_loc + (g.env_name != "tsp")
        torch.manual_seed(g.meta["sample_seed"])
        kw["exp_noise"] = torch.stack([torch.empty(b, n).exponential_(1) for _ in range(2 * n)], 0).contiguous()
    torch.manual_seed(g.meta["sample_seed"])  # OP multistart may resample its start nodes from the global generator
    with torch.inference_mode():
        out = pol(td, env, phase="test", decode_type=g.meta["decode_type"], **kw)
    assert out["actions"].shape == g.actions.shape
    same = (out["actions"] == g.actions).all(1)
    assert int((~same).sum()) <= max(1, len(same) // 50)
    assert torch.equal(out["reward"][same], g.reward[same])
    torch.testing.assert_close(out["log_likelihood"][same], g.log_likelihood[same], rtol=ll_rtol(g.env_name), atol=5e-5)
    if g.entropy is not None:  # calculate_entropy (utils/ops.py:103-111) over the stored per-step distributions
        torch.testing.assert_close(out["entropy"][same], g.entropy[same], rtol=1e-4, atol=1e-4)


def test_env_surface_step_by_step(cpu_device):
    """reset / step / get_reward / get_action_mask driven like the reference's rollout() helper."""
    for name in ("tsp50_b64_greedy", "cvrp20_b128_greedy"):
        g = GoldenCase(name)
        env = _product_env(g)
        td = env.reset(_product_td(g))
        ref_td = g.reset()
        assert torch.equal(td["action_mask"], ref_td["action_mask"])
        assert torch.equal(td["locs"], ref_td["locs"])
        t = g.actions.shape[1]
        for i in range(t):
            td.set("action", g.actions[:, i].contiguous())
            td = env.step(td)["next"]
        assert bool(td["done"].all())
        assert torch.equal(env.get_reward(td, g.actions), g.reward)
        bad = g.actions.clone()
        bad[0, 0] = bad[0, 1] if g.env_name == "tsp" else bad[0, (bad[0] > 0).nonzero()[1]]
        if g.env_name == "cvrp":
            bad[0, (g.actions[0] > 0).nonzero()[0]] = g.actions[0, (g.actions[0] > 0).nonzero()[1]]
        with pytest.raises(AssertionError, match="Invalid tour"):
            env.get_reward(td, bad)


def test_select_best_and_num_starts(cpu_device):
    g = GoldenCase("pomo_tsp20_b16_msgreedy")
    pol, env = _product_policy(g), _product_env(g)
    with torch.inference_mode():
        out = pol(env.reset(_product_td(g)), env, phase="test", decode_type="multistart_greedy", select_best=True)
        ref = g.policy(g.reset(), g.env, phase="test", decode_type="multistart_greedy", select_best=True)
    assert out["reward"].shape == (g.batch,)
    assert torch.equal(out["reward"], ref["reward"])
    assert torch.equal(out["actions"], ref["actions"])


@pytest.mark.parametrize("name", ["tsp20_b64_greedy_simple", "cvrp20_b128_greedy", "pomo_tsp20_b16_msgreedy",
                                  "op20_b128_greedy", "pctsp20_b128_greedy", "spctsp20_b128_greedy", "pdp20_b128_greedy",
                                  "cvrptw20_b128_greedy", "pomo_pdp20_b16_msgreedy", "pomo_cvrptw20_b16_mssampling"])
def test_training_log_likelihood_has_reference_gradients(cpu_device, name):
    """phase='train': sampled by the kernel, log-likelihood re-evaluated teacher-forced with
    autograd; value and parameter gradients must match the reference's decode_type='evaluate'
    path on the same actions (reinforce.py:99-102 differentiates exactly this)."""
    g = GoldenCase(name)
    pol, env = _product_policy(g), _product_env(g)
    pol.train()
    ref_pol = g.policy
    ref_pol.train()
    s = g.num_starts
    decode = "multistart_sampling" if s else "sampling"
    out = pol(env.reset(_product_td(g)), env, phase="train", decode_type=decode, seed=11,
              **g.meta["forward_kwargs"])
    assert out["log_likelihood"].requires_grad
    # reference quirk: with multistart the forced `actions` start AFTER the imposed start node
    # (pre_decoder_hook consumes no column, constructive/base.py:219-232)
    forced = out["actions"][:, 1:] if s else out["actions"]
    ref = ref_pol(g.reset(), g.env, phase="train", actions=forced, **(dict(num_starts=s) if s else {}))
    assert torch.equal(ref["actions"], out["actions"])
    # and the product's own evaluate path follows the same convention
    with torch.no_grad():
        ev = pol(env.reset(_product_td(g)), env, phase="train", actions=forced, **(dict(num_starts=s) if s else {}))
    assert torch.equal(ev["actions"], out["actions"])
    torch.testing.assert_close(ev["log_likelihood"], ref["log_likelihood"].detach(), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(out["reward"], ref["reward"], rtol=0, atol=0)
    torch.testing.assert_close(out["log_likelihood"], ref["log_likelihood"], rtol=1e-4, atol=1e-4)
    adv = torch.linspace(-1, 1, out["log_likelihood"].shape[0])
    (adv * out["log_likelihood"]).mean().backward()
    (adv * ref["log_likelihood"]).mean().backward()
    ours = dict(pol.named_parameters())
    for k, p in ref_pol.named_parameters():
        if p.grad is None:
            assert ours[k].grad is None or float(ours[k].grad.abs().max()) == 0.0, k
            continue
        if g.env_name == "cvrptw":  # unnormalised inputs (to 480): gradients span orders of magnitude inside one tensor
            err, nrm = float((ours[k].grad - p.grad).norm()), float(p.grad.norm())
            assert err <= 1e-3 * nrm + 1e-6, (k, err, nrm)
            continue
        torch.testing.assert_close(ours[k].grad, p.grad, rtol=2e-3, atol=2e-5, msg=lambda m: f"{k}: {m}")


@pytest.mark.parametrize("decode,kw", [("multistart_greedy", {}), ("sampling", dict(num_samples=4))])
def test_select_best_with_grad_enabled(cpu_device, decode, kw):
    """Best-of selection OUTSIDE no_grad with trainable parameters (policy.py:766): the differentiable
    re-evaluation replays all s x B rows (imposed start nodes, batchified state) and THEN narrows to the selected
    rows — values as the reference's (decoding.py:332-342,415-423), with autograd history into the parameters."""
    g = GoldenCase("pomo_tsp20_b16_msgreedy" if "multistart" in decode else "cvrp20_b128_greedy")
    pol, env = _product_policy(g), _product_env(g)
    n = g.num_loc + (g.env_name != "tsp")
    extra = {}
    if decode == "sampling":
        torch.manual_seed(5)
        extra["exp_noise"] = torch.stack([torch.empty(g.batch * 4, n).exponential_(1) for _ in range(2 * n)], 0).contiguous()
    out = pol(env.reset(_product_td(g)), env, phase="test", decode_type=decode, select_best=True, **kw, **extra)
    assert out["reward"].shape == (g.batch,) and out["actions"].shape[0] == g.batch
    assert out["log_likelihood"].shape == (g.batch,) and out["log_likelihood"].requires_grad
    with torch.inference_mode():
        want = pol(env.reset(_product_td(g)), env, phase="test", decode_type=decode, select_best=True, **kw, **extra)
    assert torch.equal(out["actions"], want["actions"]) and torch.equal(out["reward"], want["reward"])
    torch.testing.assert_close(out["log_likelihood"].detach(), want["log_likelihood"], rtol=1e-4, atol=1e-4)
    out["log_likelihood"].sum().backward()
    assert any(p.grad is not None and float(p.grad.abs().max()) > 0 for p in pol.parameters())


def test_entropy_is_differentiable_under_grad(cpu_device):
    """return_entropy with autograd on (PPO's entropy bonus, rl/ppo/ppo.py): the reference's entropy carries history;
    ours is built from the differentiable re-evaluation and equals the kernel's value."""
    g = GoldenCase("cvrp20_b128_greedy")
    pol, env = _product_policy(g), _product_env(g)
    out = pol(env.reset(_product_td(g)), env, phase="test", decode_type="greedy", return_entropy=True)
    assert out["entropy"].requires_grad
    with torch.inference_mode():
        want = pol(env.reset(_product_td(g)), env, phase="test", decode_type="greedy", return_entropy=True)
    torch.testing.assert_close(out["entropy"].detach(), want["entropy"], rtol=1e-4, atol=1e-4)
    ref = g.policy(g.reset(), g.env, phase="test", decode_type="greedy", return_entropy=True)
    torch.testing.assert_close(out["entropy"].detach(), ref["entropy"].detach(), rtol=1e-4, atol=1e-4)
    out["entropy"].sum().backward()
    ref["entropy"].sum().backward()
    ours = dict(pol.named_parameters())
    for k, p in g.policy.named_parameters():
        if p.grad is not None and float(p.grad.abs().max()) > 0:
            torch.testing.assert_close(ours[k].grad, p.grad, rtol=5e-3, atol=5e-5, msg=lambda m: f"{k}: {m}")


def test_step_wrappers_reject_mismatched_rows():
    """kernels.*_step: an action tensor with fewer rows than the mask would be read out of bounds on the device
    (ADVICE r1): refused on the host, before the non-CUDA check even matters."""
    from rl4co_amd import kernels as K

    with pytest.raises(ValueError, match="rows"):
        K._check_rows(8, action=torch.zeros(4, dtype=torch.int64), done=torch.zeros(8, dtype=torch.bool))
    K._check_rows(8, action=None, done=torch.zeros(8, dtype=torch.bool))


def test_reset_regenerates_only_an_empty_tensordict(cpu_device):
    """base.py:135-143 tests td.is_empty(), not len(td)."""
    from rl4co_amd.envs import get_env
    from rl4co_amd.tensordict import TensorDict

    env = get_env("tsp", generator_params=dict(num_loc=10), device="cpu")
    td = env.reset(TensorDict({}, batch_size=[4]))
    assert td["locs"].shape == (4, 10, 2)
    full = TensorDict({"locs": torch.rand(3, 10, 2)}, batch_size=[3])
    assert torch.equal(env.reset(full)["locs"], full["locs"])


def test_user_env_subclass_with_reference_signature_get_reward(cpu_device):
    """a subclass that overrides ``_get_reward(self, td, actions)`` with the reference's signature (no
    ``horizon``) must keep working — the policy then takes the post-read-back reward path."""
    from rl4co_amd.envs import TSPEnv
    from rl4co_amd.policy import AttentionModelPolicy

    calls = []

    class MyTSP(TSPEnv):
        def _get_reward(self, td, actions):  # the reference's signature (tsp/env.py:150)
            calls.append(tuple(actions.shape))
            return super()._get_reward(td, actions) * 2.0

    env = MyTSP(generator_params=dict(num_loc=10), device="cpu")
    assert not env.accepts_reward_horizon() and TSPEnv(generator_params=dict(num_loc=10), device="cpu").accepts_reward_horizon()
    torch.manual_seed(0)
    pol = AttentionModelPolicy("tsp").eval()
    torch.manual_seed(1)
    td = env.reset(batch_size=[8])
    with torch.inference_mode():
        out = pol(td, env, phase="test", decode_type="greedy")
    base = TSPEnv(generator_params=dict(num_loc=10), device="cpu")
    assert calls == [(8, 10)]
    assert torch.equal(out["reward"], base.get_reward(td, out["actions"]) * 2.0)


def test_backward_error_sink_survives_forwards_between_forward_and_backward(cpu_device):
    """the sticky word of the teacher-forced backward is ONE persistent tensor consumed in place; a
    forward issued between a grad forward and its backward must not orphan it."""
    from rl4co_amd.policy import AttentionModelPolicy

    pol = AttentionModelPolicy("tsp")
    pol._bwd_err = torch.zeros(1, dtype=torch.int32)
    sink = pol._bwd_err
    env = __import__("rl4co_amd.envs", fromlist=["get_env"]).get_env("tsp", generator_params=dict(num_loc=10), device="cpu")
    torch.manual_seed(1)
    td = env.reset(batch_size=[4])
    with torch.inference_mode():
        pol.eval()(td, env, phase="test", decode_type="greedy")  # consumes the (zero) word, keeps the tensor
    assert pol._bwd_err is sink
    sink.fill_(2)  # a late backward reports an infeasible action
    with pytest.raises(AssertionError, match="infeasible action selected"), torch.inference_mode():
        pol(env.reset(batch_size=[4]), env, phase="test", decode_type="greedy")
    assert pol._bwd_err is sink and int(sink.item()) == 0
    sink.fill_(1)
    with pytest.raises(AssertionError, match="Logits contain NaNs"):
        pol.check_backward_errors()
    assert int(sink.item()) == 0


def test_trained_parity_measurement_on_the_fake_device(cpu_device):
    """tests/trained_parity.compare (what the `-m gpu` trained-weight tests and bench.py's parity block run) end to end
    on the stand-in device: the tanh-plateau fixture (logit key x 400: nearly every greedy step an exact tie at +10,
    lowest index wins) — the C oracle's specified-order arithmetic must reproduce the reference's tours, and any flip
    must be a proven near-tie."""
    from trained_parity import TrainedCase, compare

    case = TrainedCase("sharpkl400_tsp100_b512_greedy")
    case.batch = 64  # a prefix of the seeded batch is not the same draw: regenerate and slice instead
    full = TrainedCase("sharpkl400_tsp100_b512_greedy")
    data = full.instances("cpu")
    case.instances = lambda device: data[:64]
    case.actions, case.reward = full.actions[:64], full.reward[:64]
    rec = compare(case, "fp32", "cpu", against="fp32")
    assert rec["of"] == 64 and rec["flips"] <= 1 and rec["flip_regret_max"] <= 2e-5, rec
    assert rec["step_agreement"] >= 0.999 and rec["rewards_bit_identical_on_identical"] is True
