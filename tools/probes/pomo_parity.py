import json, sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from trained_parity import TrainedCase, compare, compare_augmented
c = TrainedCase("t4_pomo_tsp100_b256_msgreedy")
res = {}
for cfg, ag in (("fp32", "fp32"), ("bf16", "fp32"), ("bf16", "bf16"), ("fp16", "fp32")):
    res[f"{cfg}_vs_{ag}"] = compare(c, cfg, "cuda", against=ag, decode="multistart_greedy", regret=False)
for cfg in ("fp32", "bf16", "fp16"):
    res[f"aug_{cfg}"] = compare_augmented(c, cfg, "cuda")
print(json.dumps(res, indent=1))
