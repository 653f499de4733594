"""Shared builders for the parity tests: seeded instances, oracle policies, device state."""
from __future__ import annotations

import torch

from oracle import reference_torch as R

WEIGHT_SEED = 0
DATA_SEED = 1234


def make_policy(env_name: str, pomo: bool = False, seed: int = WEIGHT_SEED, **kw):
    torch.manual_seed(seed)
    pol = R.pomo_policy(env_name, **kw) if pomo else R.AttentionModelPolicy(env_name, **kw)
    return pol.eval()


def make_instances(env_name: str, num_loc: int, batch: int, seed: int = DATA_SEED, check_solution=True):
    env = R.get_env(env_name, num_loc, check_solution=check_solution)
    torch.manual_seed(seed)
    data = env.generate(batch)
    return env, data


def clone_td(td: dict) -> dict:
    return {k: (v.clone() if torch.is_tensor(v) else v) for k, v in td.items()}


def decoder_weights(pol) -> dict:
    dec = pol.decoder
    return dict(
        w_node=dec.project_node_embeddings.weight.detach(),
        w_out=dec.pointer.project_out.weight.detach(),
        w_ctx=dec.context_embedding.project_context.weight.detach(),
        w_fixed=dec.project_fixed_context.weight.detach() if dec.use_graph_context else None,
        w_placeholder=getattr(dec.context_embedding, "W_placeholder", None),
    )


def device_state(env_name: str, td: dict, device) -> dict:
    """Copy an oracle reset state onto the GPU in the layout the kernels update in place."""
    st = {
        "action_mask": td["action_mask"].to(device).contiguous(),
        "current_node": td["current_node"].reshape(-1).to(device).contiguous(),
        "done": td["done"].reshape(-1).to(device).contiguous(),
    }
    if env_name == "tsp":
        st["first_node"] = td["first_node"].to(device).clone().contiguous()
        st["i"] = td["i"].reshape(-1).to(device).contiguous()
    else:
        st["demand"] = td["demand"].to(device).contiguous()
        st["used_capacity"] = td["used_capacity"].reshape(-1).to(device).contiguous()
        st["vehicle_capacity"] = td["vehicle_capacity"].reshape(-1).to(device).contiguous()
        st["visited"] = td["visited"].to(device).contiguous()
    return st
