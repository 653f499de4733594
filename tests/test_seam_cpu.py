"""The `env` seam of SURVEY.md §8(b), exercised from the reference's side: the REAL reference policy
(`rl4co/models/zoo/am/policy.py` and its decode loop, imported verbatim by oracle/ref_import.py) drives the PRODUCT
environments of rl4co_amd.envs — `reset`, `step`, `get_action_mask`, `get_reward` with the validity check on,
`get_num_starts`, `select_start_nodes` — and must reproduce the goldens of the all-reference run bit for bit.
The kernels behind the product environments are played by the C oracle (tests/fake_device.py): on the GPU box the
reference tree does not exist, there the same kernels are checked against that oracle bit for bit instead."""
import pytest
import torch

from oracle import ref_import
from tests.fake_device import cpu_device  # noqa: F401
from tests.helpers import GoldenCase, manifest

pytestmark = pytest.mark.skipif(not ref_import.available(), reason="reference checkout not present")

CASES = sorted(c for c, m in manifest().items() if m["batch"] <= 128)


@pytest.mark.parametrize("name", CASES)
def test_reference_policy_drives_product_env(cpu_device, name):
    from rl4co_amd.envs import get_env

    ref = ref_import.load()
    g = GoldenCase(name)
    torch.manual_seed(g.meta["weight_seed"])
    pol = ref.AttentionModelPolicy(env_name=g.env_label, **g.meta["policy_kwargs"]).eval()
    env = get_env(g.env_label, generator_params=dict(num_loc=g.num_loc), device="cpu")
    td = ref.TensorDict({k: v.clone() for k, v in g.data.items()}, batch_size=[g.batch])
    torch.manual_seed(g.meta["sample_seed"])
    with torch.inference_mode():
        out = pol(env.reset(td), env, phase="test", decode_type=g.meta["decode_type"], **g.meta["forward_kwargs"])
    assert torch.equal(out["actions"], g.actions)
    assert torch.equal(out["reward"], g.reward)
    assert torch.equal(out["log_likelihood"], g.log_likelihood)
    if g.entropy is not None:
        assert torch.equal(out["entropy"], g.entropy)


@pytest.mark.parametrize("name", CASES)
def test_product_policy_runs_on_reference_env(cpu_device, name):
    """The `policy` seam the other way round: the product policy handed the REFERENCE's own environment object
    (state tensors from its reset, reward and validity check by its get_reward, start nodes by its
    select_start_nodes) — same outputs as the all-reference run up to fp32 near-tie flips."""
    from rl4co_amd.policy import AttentionModelPolicy
    from tests.helpers import ll_rtol

    ref = ref_import.load()
    g = GoldenCase(name)
    pk = dict(g.meta["policy_kwargs"])
    pk.pop("sdpa_fn_decoder", None)
    pol = AttentionModelPolicy(env_name=g.env_label, **pk).eval()
    pol.load_state_dict(g.policy.state_dict())
    env_cls = {"tsp": ref.TSPEnv, "cvrp": ref.CVRPEnv, "op": ref.OPEnv, "pctsp": ref.PCTSPEnv, "spctsp": ref.SPCTSPEnv,
               "pdp": ref.PDPEnv, "cvrptw": ref.CVRPTWEnv}[g.env_label]
    gen_kw = dict(num_loc=g.num_loc)
    if g.env_label == "op":
        gen_kw["prize_distribution"] = "dist"  # as in oracle/gen_golden.py
    env = env_cls(generator_params=gen_kw)
    td = ref.TensorDict({k: v.clone() for k, v in g.data.items()}, batch_size=[g.batch])
    kw = dict(g.meta["forward_kwargs"])
    if "sampling" in g.meta["decode_type"]:
        n = g.num_loc + (g.env_name != "tsp")
        torch.manual_seed(g.meta["sample_seed"])
        kw["exp_noise"] = torch.stack([torch.empty(g.rollout_rows, n).exponential_(1) for _ in range(2 * n)], 0).contiguous()
    torch.manual_seed(g.meta["sample_seed"])
    with torch.inference_mode():
        out = pol(env.reset(td), env, phase="test", decode_type=g.meta["decode_type"], **kw)
    assert out["actions"].shape == g.actions.shape
    same = (out["actions"] == g.actions).all(1)
    assert int((~same).sum()) <= max(1, len(same) // 50)
    assert torch.equal(out["reward"][same], g.reward[same])
    torch.testing.assert_close(out["log_likelihood"][same], g.log_likelihood[same], rtol=ll_rtol(g.env_name), atol=5e-5)


@pytest.mark.parametrize("evaluator,kw", [("GreedyEval", {}), ("AugmentationEval", dict(num_augment=8, force_dihedral_8=True)),
                                           ("GreedyMultiStartEval", dict(num_starts=20)),
                                           ("GreedyMultiStartAugmentEval", dict(num_starts=20, num_augment=8, force_dihedral_8=True))])
@pytest.mark.parametrize("name", ["pomo_tsp20_b16_msgreedy", "pomo_cvrp20_b16_msgreedy"])
def test_reference_evaluators_run_unchanged(cpu_device, name, evaluator, kw):
    """`EvalBase` and its subclasses (rl4co/tasks/eval.py:18-299, verbatim) — a caller SURVEY.md §8a lists as "must keep
    working unchanged": they call `policy(td.clone(), decode_type=..., num_starts=...)` WITHOUT an environment, reset
    and score through `env`, augment with the reference's StateAugmentation and pick the best of augmentations x
    starts. Product (env, policy, dataset) vs reference (env, policy) through the same evaluator: same rewards."""
    import importlib

    from torch.utils.data import DataLoader

    from rl4co_amd import data as D
    from rl4co_amd.envs import get_env
    from rl4co_amd.policy import AttentionModelPolicy
    from rl4co_amd.tensordict import TensorDict

    ref = ref_import.load()
    ev = importlib.import_module("rl4co.tasks.eval")
    g = GoldenCase(name)
    pk = dict(g.meta["policy_kwargs"])
    # product side
    pol = AttentionModelPolicy(env_name=g.env_label, **pk).eval()
    pol.load_state_dict(g.policy.state_dict())
    env = get_env(g.env_label, generator_params=dict(num_loc=g.num_loc), device="cpu")
    ds = D.TensorDictDataset(TensorDict({k: v.clone() for k, v in g.data.items()}, batch_size=[g.batch]))
    dl = DataLoader(ds, batch_size=8, collate_fn=ds.collate_fn)
    got = getattr(ev, evaluator)(env, progress=False, **kw)(pol, dl)
    # reference side
    torch.manual_seed(g.meta["weight_seed"])
    rpol = ref.AttentionModelPolicy(env_name=g.env_label, **pk).eval()
    renv = {"tsp": ref.TSPEnv, "cvrp": ref.CVRPEnv}[g.env_label](generator_params=dict(num_loc=g.num_loc))
    rtd = ref.TensorDict({k: v.clone() for k, v in g.data.items()}, batch_size=[g.batch])
    rdl = [rtd[i : i + 8] for i in range(0, g.batch, 8)]
    want = getattr(ev, evaluator)(renv, progress=False, **kw)(rpol, rdl)
    assert got["rewards"].shape == want["rewards"].shape == (g.batch,)
    same = (got["rewards"] == want["rewards"])
    assert int((~same).sum()) <= 1, (got["rewards"], want["rewards"])
    torch.testing.assert_close(got["rewards"], want["rewards"], rtol=2e-2, atol=0)  # a flipped row is still near-optimal


def test_reference_reinforce_baselines_run_unchanged(cpu_device):
    """REINFORCE's baselines (rl4co/models/rl/reinforce/baselines.py:55-259, verbatim; §8a: "callers that must keep
    working unchanged"): SharedBaseline on multistart rewards, ExponentialBaseline, and RolloutBaseline — deep copy of
    the policy, `env.dataset(...)`, greedy evaluation over a DataLoader, `wrap_dataset` (rewards attached as "extra"),
    the one-sided paired t-test of `epoch_callback` — over the product policy, environment and dataset; the baseline
    values equal the reference pair's on the same instances, and the REINFORCE loss built from them differentiates
    through the product policy's log-likelihood."""
    import importlib

    from rl4co_amd import data as D
    from rl4co_amd.envs import get_env
    from rl4co_amd.policy import AttentionModelPolicy
    from rl4co_amd.tensordict import TensorDict

    ref = ref_import.load()
    bl = importlib.import_module("rl4co.models.rl.reinforce.baselines")
    g = GoldenCase("tsp20_b64_greedy_simple")
    pk = {k: v for k, v in g.meta["policy_kwargs"].items() if k != "sdpa_fn_decoder"}
    pol = AttentionModelPolicy(env_name="tsp", **pk)
    pol.load_state_dict(g.policy.state_dict())
    env = get_env("tsp", generator_params=dict(num_loc=20), device="cpu")
    ds = D.TensorDictDataset(TensorDict({k: v.clone() for k, v in g.data.items()}, batch_size=[g.batch]))

    rollout = bl.RolloutBaseline(bl_alpha=0.05)
    rollout.setup(pol, env, batch_size=16, device="cpu", dataset_size=32)  # evaluation set from env.dataset
    assert rollout.bl_vals.shape == (32,) and rollout.policy is not pol
    wrapped = rollout.wrap_dataset(ds, env, batch_size=16, device="cpu")
    assert torch.equal(wrapped.data["extra"], g.reward)  # greedy rewards of the golden run, instance by instance
    batch = wrapped.__getitems__(list(range(16)))
    bl_val, bl_loss = rollout.eval(env.reset(batch), None, env)
    # (RolloutBaseline.eval calls the copied policy with its defaults, i.e. phase="train" -> a SAMPLED rollout)
    assert bl_val.shape == (16,) and bool(torch.isfinite(bl_val).all()) and bl_loss == 0
    assert float(bl_val.mean()) < float(g.reward[:16].mean())  # sampled tours of an untrained policy are longer than greedy
    rollout.epoch_callback(pol, env, batch_size=16, device="cpu", epoch=0, dataset_size=32)  # same policy: no update

    # SharedBaseline / ExponentialBaseline on a sampled multistart rollout, REINFORCE loss as in reinforce.py:99-111
    pol.train()
    td = env.reset(ds.__getitems__(list(range(8))))
    out = pol(td, env, phase="train", decode_type="multistart_sampling", num_starts=5, seed=1)
    reward = out["reward"].view(5, 8).t()  # unbatchify: [instances, starts]
    shared, _ = bl.SharedBaseline().eval(td, reward)
    assert torch.equal(shared, reward.mean(1, keepdim=True))
    expo = bl.ExponentialBaseline(beta=0.8)
    v1, _ = expo.eval(td, out["reward"])
    v2, _ = expo.eval(td, out["reward"] * 2)
    torch.testing.assert_close(v2, 0.8 * v1 + 0.2 * (out["reward"] * 2).mean())
    ll = out["log_likelihood"].view(5, 8).t()
    loss = -((reward - shared) * ll).mean()
    loss.backward()
    grads = [p.grad for p in pol.parameters() if p.grad is not None]
    assert len(grads) > 20 and all(torch.isfinite(x).all() for x in grads) and any(float(x.abs().max()) > 0 for x in grads)


def test_reference_pomo_and_reinforce_modules_step_unchanged(cpu_device):
    """`POMO.shared_step` (zoo/pomo/model.py:88-143) and `REINFORCE.shared_step / calculate_loss`
    (rl/reinforce/reinforce.py:59-111), the reference's own training-step code, over the product policy and
    environment: train (multistart sampling, shared baseline, REINFORCE loss with gradients), validation (dihedral-8
    augmentation x multistart, best-of selection), and AM's REINFORCE with the exponential baseline. The Lightning
    base class is a test stand-in (oracle/shims/lightning: constructor + log_dict only); the step code is verbatim.
    Reference policy + reference env through the same module give the same validation rewards."""
    import importlib

    from rl4co_amd.envs import get_env
    from rl4co_amd.policy import AttentionModelPolicy
    from rl4co_amd.tensordict import TensorDict

    ref = ref_import.load()
    pomo = importlib.import_module("rl4co.models.zoo.pomo.model")
    reinforce = importlib.import_module("rl4co.models.rl.reinforce.reinforce")
    g = GoldenCase("pomo_tsp20_b16_msgreedy")
    pk = dict(g.meta["policy_kwargs"])
    pol = AttentionModelPolicy(env_name="tsp", **pk)
    pol.load_state_dict(g.policy.state_dict())
    env = get_env("tsp", generator_params=dict(num_loc=20), device="cpu")
    batch = TensorDict({k: v.clone() for k, v in g.data.items()}, batch_size=[g.batch])

    model = pomo.POMO(env, policy=pol, num_augment=8)
    assert pol.train_decode_type == "multistart_sampling" and pol.val_decode_type == "multistart_greedy"
    torch.manual_seed(0)
    out = model.shared_step(batch, 0, "train")
    assert out["loss"].requires_grad and torch.isfinite(out["loss"])
    out["loss"].backward()
    grads = [p.grad for p in pol.parameters() if p.grad is not None]
    assert len(grads) > 20 and all(torch.isfinite(x).all() for x in grads) and any(float(x.abs().max()) > 0 for x in grads)
    with torch.inference_mode():
        val = model.shared_step(batch, 0, "val")
    logged = model.logged[-1][0]
    assert "val/reward" in logged

    # the same module over the reference's own policy and environment: same validation metrics (greedy, deterministic)
    torch.manual_seed(g.meta["weight_seed"])
    rpol = ref.AttentionModelPolicy(env_name="tsp", **pk)
    rmodel = pomo.POMO(ref.TSPEnv(generator_params=dict(num_loc=20)), policy=rpol, num_augment=8)
    rbatch = ref.TensorDict({k: v.clone() for k, v in g.data.items()}, batch_size=[g.batch])
    with torch.inference_mode():
        rmodel.shared_step(rbatch, 0, "val")
    rlogged = rmodel.logged[-1][0]
    for k in rlogged:
        torch.testing.assert_close(torch.as_tensor(logged[k]).float(), torch.as_tensor(rlogged[k]).float(), rtol=2e-3, atol=0)

    # AttentionModel-style REINFORCE with the exponential baseline
    g2 = GoldenCase("tsp20_b64_greedy_simple")
    pk2 = {k: v for k, v in g2.meta["policy_kwargs"].items() if k != "sdpa_fn_decoder"}
    pol2 = AttentionModelPolicy(env_name="tsp", **pk2)
    pol2.load_state_dict(g2.policy.state_dict())
    am = reinforce.REINFORCE(env, pol2, baseline="exponential")
    b2 = TensorDict({k: v.clone() for k, v in g2.data.items()}, batch_size=[g2.batch])
    out2 = am.shared_step(b2, 0, "train")
    assert out2["loss"].requires_grad
    out2["loss"].backward()
    # test step on untouched weights (the training forward above moved the batch-norm running statistics); the
    # trainer puts the module in eval mode for it
    pol3 = AttentionModelPolicy(env_name="tsp", **pk2)
    pol3.load_state_dict(g2.policy.state_dict())
    am = reinforce.REINFORCE(env, pol3, baseline="exponential").eval()
    with torch.inference_mode():
        am.shared_step(b2, 0, "test")
    torch.testing.assert_close(torch.as_tensor(am.logged[-1][0]["test/reward"]).float(), g2.reward.mean(), rtol=1e-6, atol=0)


def test_reference_training_epoch_glue_runs_unchanged(cpu_device):
    """`RL4COLitModule.setup` / dataloaders / optimizer (rl/common/base.py:117-330, verbatim) over the product
    objects: datasets from `env.dataset(size, phase)`, torch DataLoaders over the device-resident dataset (batched
    `__getitems__` + `collate_fn`), REINFORCE's rollout baseline attached to the training set by `wrap_dataset`, the
    optimizer built by `configure_optimizers`, and a few optimisation steps through `shared_step`."""
    import importlib

    from rl4co_amd.envs import get_env
    from rl4co_amd.policy import AttentionModelPolicy

    ref_import.load()
    reinforce = importlib.import_module("rl4co.models.rl.reinforce.reinforce")
    pomo = importlib.import_module("rl4co.models.zoo.pomo.model")
    torch.manual_seed(0)
    env = get_env("cvrp", generator_params=dict(num_loc=10), device="cpu")

    am = reinforce.REINFORCE(env, AttentionModelPolicy("cvrp"), baseline="rollout", batch_size=8, train_data_size=24,
                             val_data_size=16, test_data_size=8, optimizer_kwargs={"lr": 1e-3})
    am.setup("fit")
    assert len(am.train_dataset) == 24  # epoch 0 of baseline="rollout" is its warm-up: exponential baseline, no wrapping
    opt = am.configure_optimizers()

    def train_epoch():
        seen = 0
        for i, batch in enumerate(am.train_dataloader()):
            assert batch.batch_size[0] == 8
            out = am.shared_step(batch, i, "train")
            opt.zero_grad()
            out["loss"].backward()
            opt.step()
            seen += 1
        return seen

    assert train_epoch() == 3
    am.eval()
    with torch.inference_mode():
        for i, batch in enumerate(am.val_dataloader()):
            am.shared_step(batch, i, "val")
    assert "val/reward" in am.logged[-1][0]
    # end of epoch 0 (reinforce.py:125-135, base.py:263-272): the warm-up ends, the greedy rollout baseline takes over
    # and the next epoch's training set carries its rewards as "extra"
    am.train()
    am.baseline.epoch_callback(am.policy, env=env, batch_size=8, device="cpu", epoch=0, dataset_size=16)
    assert am.baseline.alpha == 1
    am.train_dataset = am.wrap_dataset(env.dataset(24, "train"))
    assert "extra" in am.train_dataset.data.keys() and am.train_dataset.data["extra"].shape == (24,)
    first = next(iter(am.train_dataloader()))
    assert first["extra"].shape == (8,)
    assert train_epoch() == 3

    pm = pomo.POMO(env, policy=AttentionModelPolicy("cvrp", num_encoder_layers=3, normalization="instance",
                                                     use_graph_context=False),
                   batch_size=4, train_data_size=8, val_data_size=4, test_data_size=4, num_augment=8)
    pm.setup("fit")
    opt = pm.configure_optimizers()
    for i, batch in enumerate(pm.train_dataloader()):
        out = pm.shared_step(batch, i, "train")
        opt.zero_grad()
        out["loss"].backward()
        opt.step()
    pm.eval()
    with torch.inference_mode():
        for i, batch in enumerate(pm.test_dataloader()):
            pm.shared_step(batch, i, "test")
    assert "test/reward" in pm.logged[-1][0]


def test_reference_sampling_eval_runs_unchanged(cpu_device):
    """`SamplingEval` (tasks/eval.py:143-192): decode_type="sampling" with multisample=True AND num_starts=n, neutral
    top-k / top-p / softmax_temp, select_best, and a `select_start_nodes_fn` drawing random feasible first nodes
    (utils/ops.py sample_n_random_actions). In the reference's flag logic (decoding.py:238-255) that combination ends up
    multistart and multisample at once: n sampled rollouts per instance from random start nodes, best one kept."""
    import importlib

    from torch.utils.data import DataLoader

    from rl4co_amd import data as D
    from rl4co_amd.envs import get_env
    from rl4co_amd.policy import AttentionModelPolicy
    from rl4co_amd.tensordict import TensorDict

    ref_import.load()
    ev = importlib.import_module("rl4co.tasks.eval")
    g = GoldenCase("cvrp20_b128_greedy")
    pol = AttentionModelPolicy(env_name="cvrp").eval()
    pol.load_state_dict(g.policy.state_dict())
    env = get_env("cvrp", generator_params=dict(num_loc=20), device="cpu")
    ds = D.TensorDictDataset(TensorDict({k: v[:16].clone() for k, v in g.data.items()}, batch_size=[16]))
    torch.manual_seed(0)
    got = ev.SamplingEval(env, samples=12, progress=False)(pol, DataLoader(ds, batch_size=8, collate_fn=ds.collate_fn))
    assert got["rewards"].shape == (16,) and got["actions"].shape[0] == 16
    env.check_solution_validity(env.reset(ds.data), got["actions"])  # the selected tours are valid CVRP solutions
    greedy = ev.GreedyEval(env, progress=False)(pol, DataLoader(ds, batch_size=8, collate_fn=ds.collate_fn))
    assert float(got["rewards"].mean()) > float(greedy["rewards"].mean()) - 0.5  # best of 12 samples: around greedy or better
    with pytest.raises(NotImplementedError):
        pol(env.reset(ds.data), env, phase="test", decode_type="sampling", top_k=5)


@pytest.mark.parametrize("name", ["tsp50_b64_greedy", "cvrp20_b128_greedy", "pdp20_b128_greedy", "cvrptw20_b128_greedy"])
def test_ppo_style_reevaluation_matches_reference_policy(cpu_device, name):
    """The re-evaluation call of the reference's PPO (rl/ppo/ppo.py:164-171), the two-phase pattern row N1 builds on:
    `policy(td, actions=old_actions, env=env, return_entropy=True, return_sum_log_likelihood=False)` — per-step
    log-likelihoods of GIVEN actions and the policy entropy, product policy + product env vs reference policy +
    reference env on the golden trajectories."""
    from rl4co_amd.envs import get_env
    from rl4co_amd.policy import AttentionModelPolicy
    from rl4co_amd.tensordict import TensorDict
    from tests.helpers import ll_rtol

    ref = ref_import.load()
    g = GoldenCase(name)
    pk = {k: v for k, v in g.meta["policy_kwargs"].items() if k != "sdpa_fn_decoder"}
    pol = AttentionModelPolicy(env_name=g.env_label, **pk).eval()
    pol.load_state_dict(g.policy.state_dict())
    env = get_env(g.env_label, generator_params=dict(num_loc=g.num_loc), device="cpu")
    td = env.reset(TensorDict({k: v.clone() for k, v in g.data.items()}, batch_size=[g.batch]))
    with torch.no_grad():
        got = pol(td, actions=g.actions, env=env, return_entropy=True, return_sum_log_likelihood=False)

    torch.manual_seed(g.meta["weight_seed"])
    rpol = ref.AttentionModelPolicy(env_name=g.env_label, **g.meta["policy_kwargs"]).eval()
    renv_cls = {"tsp": ref.TSPEnv, "cvrp": ref.CVRPEnv, "pdp": ref.PDPEnv, "cvrptw": ref.CVRPTWEnv}[g.env_label]
    renv = renv_cls(generator_params=dict(num_loc=g.num_loc))
    rtd = renv.reset(ref.TensorDict({k: v.clone() for k, v in g.data.items()}, batch_size=[g.batch]))
    with torch.no_grad():
        want = rpol(rtd, actions=g.actions, env=renv, return_entropy=True, return_sum_log_likelihood=False)
    assert got["log_likelihood"].shape == want["log_likelihood"].shape == g.actions.shape
    torch.testing.assert_close(got["log_likelihood"].sum(1), want["log_likelihood"].sum(1), rtol=ll_rtol(g.env_name), atol=5e-5)
    torch.testing.assert_close(got["log_likelihood"], want["log_likelihood"], rtol=1e-3, atol=2e-4)
    torch.testing.assert_close(got["entropy"], want["entropy"], rtol=1e-4, atol=1e-4)
    assert torch.equal(got["reward"], want["reward"])


@pytest.mark.parametrize("env_name,num_loc", [("tsp", 20), ("cvrp", 20), ("op", 20), ("pctsp", 20), ("spctsp", 20), ("pdp", 20),
                                              ("cvrptw", 20)])
def test_reference_rollout_helper_with_random_policy(cpu_device, env_name, num_loc):
    """`rollout(env, td, policy)` + `random_policy` (utils/decoding.py:78-112, verbatim): the step-by-step driver the
    reference's own environment tests use — reset, `env.step(td)["next"]` until `td["done"].all()`, `env.get_reward` with
    the validity check — over every product environment."""
    from rl4co_amd.envs import get_env

    ref = ref_import.load()
    env = get_env(env_name, generator_params=dict(num_loc=num_loc), device="cpu")
    torch.manual_seed(5)
    td = env.reset(batch_size=[16])
    reward, td_out, actions = ref.decoding.rollout(env, td, ref.decoding.random_policy)
    assert reward.shape == (16,) and bool(torch.isfinite(reward).all()) and actions.shape[0] == 16
    assert bool(td_out["done"].all())


@pytest.mark.parametrize("env_name,num_loc", [("tsp", 20), ("tsp", 50), ("cvrp", 20), ("cvrp", 100), ("op", 20), ("op", 100),
                                              ("pctsp", 50), ("spctsp", 20), ("pdp", 20), ("pdp", 50), ("cvrptw", 20),
                                              ("cvrptw", 100)])
def test_product_generators_follow_the_reference_stream(env_name, num_loc):
    """Row a10: on the CPU the product generators consume the global torch generator exactly as the reference's do —
    same seed, same instances, key by key, dtype included (CVRPTW's integer-valued windows stay int32)."""
    from rl4co_amd.envs import get_env

    ref = ref_import.load()
    env_cls = {"tsp": ref.TSPEnv, "cvrp": ref.CVRPEnv, "op": ref.OPEnv, "pctsp": ref.PCTSPEnv, "spctsp": ref.SPCTSPEnv,
               "pdp": ref.PDPEnv, "cvrptw": ref.CVRPTWEnv}[env_name]
    gen_kw = dict(num_loc=num_loc)
    if env_name == "op":
        gen_kw["prize_distribution"] = "dist"
    renv = env_cls(generator_params=gen_kw)
    env = get_env(env_name, generator_params=dict(num_loc=num_loc), device="cpu")
    for seed in (0, 7):
        torch.manual_seed(seed)
        want = renv.generator(batch_size=[33])
        torch.manual_seed(seed)
        got = env.generator(batch_size=[33])
        assert sorted(got.keys()) == sorted(want.keys()), (sorted(got.keys()), sorted(want.keys()))
        for k in want.keys():
            assert got[k].dtype == want[k].dtype and torch.equal(got[k], want[k]), (k, got[k].dtype, want[k].dtype)


@pytest.mark.parametrize("env_name", ["tsp", "cvrp", "op", "pctsp", "spctsp", "pdp", "cvrptw"])
def test_reset_state_has_the_reference_keys_and_dtypes(cpu_device, env_name):
    """What a reader of the state TensorDict sees after `reset`: every key the reference's reset leaves (torchrl's
    `terminated` and CVRPTW's `current_loc` / `distances` side products included), same dtypes and values; the only
    deliberate differences are the shape of `done` / `terminated` ([B] instead of torchrl's [B, 1]) and CVRP's
    pre-existing uint8 `visited`, which the reference shares."""
    from rl4co_amd.envs import get_env
    from rl4co_amd.tensordict import TensorDict

    ref = ref_import.load()
    env_cls = {"tsp": ref.TSPEnv, "cvrp": ref.CVRPEnv, "op": ref.OPEnv, "pctsp": ref.PCTSPEnv, "spctsp": ref.SPCTSPEnv,
               "pdp": ref.PDPEnv, "cvrptw": ref.CVRPTWEnv}[env_name]
    kw = dict(num_loc=20)
    if env_name == "op":
        kw["prize_distribution"] = "dist"
    renv = env_cls(generator_params=kw)
    env = get_env(env_name, generator_params=dict(num_loc=20), device="cpu")
    torch.manual_seed(0)
    data = renv.generator(batch_size=[5])
    want = renv.reset(data.clone())
    got = env.reset(TensorDict({k: v.clone() for k, v in data.items()}, batch_size=[5]))
    assert set(want.keys()) <= set(got.keys()), sorted(set(want.keys()) - set(got.keys()))
    for k in want.keys():
        a, b = want[k], got[k]
        if k in ("done", "terminated"):
            assert b.dtype == a.dtype and torch.equal(a.reshape(-1), b.reshape(-1))
            continue
        assert a.dtype == b.dtype and a.shape == b.shape and torch.equal(a, b), (k, a.dtype, b.dtype, a.shape, b.shape)


@pytest.mark.parametrize("env_name", ["tsp", "cvrp", "op", "pctsp", "spctsp", "pdp", "cvrptw"])
def test_stepped_state_matches_the_reference_key_by_key(cpu_device, env_name):
    """The same after every `env.step(td)["next"]` of a random feasible walk to the end: every key of the reference's
    state is in the product's with the same values (flags compared as booleans, `done`-like keys flattened)."""
    from rl4co_amd.envs import get_env
    from rl4co_amd.tensordict import TensorDict

    ref = ref_import.load()
    env_cls = {"tsp": ref.TSPEnv, "cvrp": ref.CVRPEnv, "op": ref.OPEnv, "pctsp": ref.PCTSPEnv, "spctsp": ref.SPCTSPEnv,
               "pdp": ref.PDPEnv, "cvrptw": ref.CVRPTWEnv}[env_name]
    kw = dict(num_loc=20)
    if env_name == "op":
        kw["prize_distribution"] = "dist"
    renv = env_cls(generator_params=kw)
    env = get_env(env_name, generator_params=dict(num_loc=20), device="cpu")
    torch.manual_seed(1)
    data = renv.generator(batch_size=[6])
    want = renv.reset(data.clone())
    got = env.reset(TensorDict({k: v.clone() for k, v in data.items()}, batch_size=[6]))
    gen = torch.Generator().manual_seed(2)
    for _ in range(60):
        if bool(want["done"].all()):
            break
        action = torch.multinomial(want["action_mask"].float(), 1, generator=gen).squeeze(-1)
        want.set("action", action)
        got.set("action", action.clone())
        want = renv.step(want)["next"]
        got = env.step(got)["next"]
        missing = set(want.keys()) - set(got.keys()) - {"reward"}
        assert not missing, sorted(missing)
        for k in want.keys():
            if k in ("reward", "action") or k not in got.keys():
                continue
            a, b = want[k], got[k]
            if a.dtype in (torch.bool, torch.uint8) or b.dtype in (torch.bool, torch.uint8):
                assert torch.equal(a.reshape(a.shape[0], -1).bool(), b.reshape(b.shape[0], -1).bool()), k
            else:
                assert torch.equal(a.reshape(a.shape[0], -1), b.reshape(b.shape[0], -1).to(a.dtype)), k
    assert bool(want["done"].all()) and bool(got["done"].all())
