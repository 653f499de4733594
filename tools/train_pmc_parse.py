"""Summarise the PMC passes of tools/train_pmc.sh: per kernel of the training step, mean counter values per launch and
the derived pipe utilisation. SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves,
SQ_VALU_MFMA_BUSY_CYCLES cycles summed over SIMDs (MI355X_MICROARCH.md); GRBM_GUI_ACTIVE sums the 8 XCDs.

    python tools/train_pmc_parse.py gpurun_out/<tag>      # writes <dir>/c4_train_pmc.json
"""
import collections
import csv
import glob
import json
import sys

src = sys.argv[1]
# every kernel with >= 2 % of the step (r04's file had no entry for the one-launch training forward and the MLP's
# input-gradient kernel: VERDICT r04 "missing" 3)
KERNELS = ["am_teacher_mma_kernel", "am_decode_ms_kernel", "am_encoder_kernel", "tok16_mlp_bwd_kernel", "linear_bf16_kernel",
           "linear_k128_kernel", "wgrad_bf16_kernel", "attn_fwd_kernel", "attn_bwd_kernel", "skip_inorm_fwd_kernel",
           "skip_inorm_bwd_kernel", "reduce_kernel"]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(src + "/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        for k in KERNELS:
            if k in r["Kernel_Name"]:
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, counters in agg.items():
    m = {c: sum(v) / len(v) for c, v in counters.items()}
    rec = {"counters_mean_per_launch": m, "launches": len(next(iter(counters.values())))}
    wc = m.get("SQ_WAVE_CYCLES")
    if wc:
        rec["wave_resident_quadcycles"] = wc
        for name, c in (("issue_any", "SQ_ACTIVE_INST_ANY"), ("issue_valu", "SQ_ACTIVE_INST_VALU"), ("issue_lds", "SQ_ACTIVE_INST_LDS"),
                        ("issue_scalar", "SQ_ACTIVE_INST_SCA"), ("issue_vmem", "SQ_ACTIVE_INST_VMEM"), ("wait_inst_any", "SQ_WAIT_INST_ANY"),
                        ("wait_any", "SQ_WAIT_ANY"), ("wait_inst_lds", "SQ_WAIT_INST_LDS")):
            if c in m:
                rec[f"frac_of_wave_time_{name}"] = m[c] / wc
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m:
        # cycles the launch lasted on one XCD x 1024 SIMDs = the matrix pipes' available cycles
        rec["mfma_pipe_busy_frac"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
    if "SQ_WAVES" in m:
        for c in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"):
            if c in m:
                rec[f"{c.lower()}_per_wave"] = m[c] / m["SQ_WAVES"]
    if "SQ_LDS_BANK_CONFLICT" in m and "SQ_LDS_IDX_ACTIVE" in m and m["SQ_LDS_IDX_ACTIVE"]:
        rec["lds_bank_conflict_frac"] = m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"]
    out[k] = rec
stats = glob.glob(src + "/trace/*/*_kernel_stats.csv")
if stats:
    rows = [r for r in csv.DictReader(open(stats[0])) if "hbm_read_probe" not in r["Name"]]  # (bench.py's untimed warm-up)
    out["_kernel_stats_top"] = [{"name": r["Name"][:90], "calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                                 "pct": float(r["Percentage"])} for r in rows[:14]]
out["_note"] = ("rocprofv3 --pmc (three separate counter-only passes) + one --kernel-trace --stats pass over "
                "`python bench.py --legs c4_train --steps 3 --warmup 2 --no-cpu-baseline --no-parity` (POMO-6L, TSP-100 x 4096 x 8 starts)")
json.dump(out, open(src + "/c4_train_pmc.json", "w"), indent=1)
for k in ("am_teacher_mma_kernel", "am_decode_ms_kernel"):
    if k in out:
        print(k, {a: round(b, 4) for a, b in out[k].items() if isinstance(b, float)})
