"""gpurun_out/<tag>_mall/<B>/... (tools/decode_vs_mall.sh) -> one JSON: per batch the decode launch's duration (rocprofv3
kernel stats and the run's own HIP events), must-move bytes, FETCH_SIZE x correction + WRITE_SIZE, and the rates they give.

    python tools/decode_vs_mall_parse.py gpurun_out/r06_mall profiles/r06_decode_vs_mall.json
"""
import collections
import csv
import glob
import json
import os
import sys

src, dst = sys.argv[1], sys.argv[2]
out = {"what": "am_decode_kernel STREAM, TSP-100 greedy, bf16 planes: the headline batch (planes 315 MB: the feasible rows fall below the "
               "256 MiB Infinity Cache after ~20 of 100 steps) against a batch whose planes are 4 x that (1.26 GB)",
       "peak_GBs": 8000.0, "batches": {}}
for bdir in sorted(glob.glob(os.path.join(src, "*")), key=lambda p: int(os.path.basename(p)) if os.path.basename(p).isdigit() else 0):
    if not os.path.basename(bdir).isdigit():
        continue
    rec = {}
    try:
        d = json.load(open(os.path.join(bdir, "detail.json")))
        r = d["roofline"]
        rec.update(launch_ms_hip_events=r["launch_ms_mean"], must_move_bytes=r["bytes_per_launch"], contract_bytes=r["algorithmic_bytes_contract"],
                   must_move_GBs=r["achieved"], frac_must_move=r["frac"], rows_per_launch=r["cache_rows_per_launch"])
    except (OSError, ValueError, KeyError) as exc:
        rec["error"] = f"no detail: {exc}"
    for f in sorted(glob.glob(os.path.join(bdir, "trace", "*", "*_kernel_stats.csv")), key=os.path.getmtime)[-1:]:
        for row in csv.DictReader(open(f)):
            if "am_decode_kernel" in row["Name"]:
                rec["launch_ms_rocprof"] = float(row["AverageNs"]) / 1e6
                rec["rocprof_calls"] = int(row["Calls"])
                break
    cnt = {}
    probe = {}
    for sub, name in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        for f in sorted(glob.glob(os.path.join(bdir, sub, "*", "*_counter_collection.csv")), key=os.path.getmtime)[-1:]:
            agg = collections.defaultdict(list)
            for row in csv.DictReader(open(f)):
                if row["Counter_Name"] == name:
                    agg[row["Kernel_Name"]].append(float(row["Counter_Value"]))
            for k, v in agg.items():
                if "am_decode_kernel" in k:
                    cnt[name] = sum(v) / len(v) * 1024.0
                if "hbm_read_probe" in k:
                    big = [x for x in v if x >= 0.5 * max(v)]
                    probe[name] = sum(big) / len(big) * 1024.0
    if "FETCH_SIZE" in cnt:
        corr = (2 << 30) / probe["FETCH_SIZE"] if probe.get("FETCH_SIZE") else 2.0  # gfx950: FETCH_SIZE reports half of a wide stream
        traffic = cnt["FETCH_SIZE"] * corr + cnt.get("WRITE_SIZE", 0.0)
        ms = rec.get("launch_ms_rocprof") or rec.get("launch_ms_hip_events")
        rec.update(fetch_bytes_raw=cnt["FETCH_SIZE"], fetch_correction=corr, write_bytes=cnt.get("WRITE_SIZE"), traffic_bytes=traffic)
        if ms:
            rec["traffic_GBs"] = traffic / (ms * 1e-3) / 1e9
            rec["frac_traffic"] = rec["traffic_GBs"] / 8000.0
        if rec.get("must_move_bytes"):
            rec["traffic_over_must_move"] = traffic / rec["must_move_bytes"]
    out["batches"][os.path.basename(bdir)] = rec
b = out["batches"]
if "4096" in b and "16384" in b and b["4096"].get("must_move_GBs") and b["16384"].get("must_move_GBs"):
    small, large = b["4096"], b["16384"]
    out["reading"] = {
        "headline_batch_GBs": small["must_move_GBs"], "large_batch_GBs": large["must_move_GBs"],
        "large_over_headline": large["must_move_GBs"] / small["must_move_GBs"],
        "hbm_fraction_to_quote": min(small["frac_must_move"], large["frac_must_move"]),
        "note": "both rates are must-move bytes / launch duration. At 16 384 instances the feasible rows exceed the 256 MiB Infinity "
                "Cache for most of the rollout, so its rate is an HBM rate; where it is LOWER than the headline batch's, the "
                "difference is the MALL's share and the large-batch figure is the one to quote as the HBM fraction",
    }
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out.get("reading", out), indent=1))
