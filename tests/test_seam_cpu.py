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
