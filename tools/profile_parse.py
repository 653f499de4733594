"""Turn the rocprofv3 output of tools/profile_round.sh (gpurun_out/<tag>/...) into the small,
committed summaries under profiles/: kernel stats table, PMC HBM traffic, and
profiles/pmc_traffic.json (read by bench.py for the `roofline.traffic` field).

    python tools/profile_parse.py gpurun_out/final r01_final
"""
import collections
import csv
import glob
import json
import sys
from pathlib import Path

src, tag = Path(sys.argv[1]), sys.argv[2]
out = Path(__file__).resolve().parents[1] / "profiles"
out.mkdir(exist_ok=True)

stats = glob.glob(str(src / "trace" / "*" / "*_kernel_stats.csv"))
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    lines = [f"# rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline   ({tag}, MI355X)",
             f"{'kernel':100s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}"]
    for r in rows[:25]:
        lines.append(f"{r['Name'][:100]:100s} {r['Calls']:>6s} {float(r['TotalDurationNs'])/1e6:10.3f} "
                     f"{float(r['AverageNs'])/1e3:10.2f} {float(r['MinNs'])/1e3:10.2f} {float(r['MaxNs'])/1e3:10.2f} "
                     f"{float(r['Percentage']):6.2f}")
    (out / f"{tag}_kernel_stats.txt").write_text("\n".join(lines) + "\n")
    print("\n".join(lines[:8]))

res = {}
for sub, name in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    for f in glob.glob(str(src / sub / "*" / "*_counter_collection.csv")):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            if any(s in k for s in ("am_decode", "am_encoder", "hbm_read_probe", "tour_length")):
                res.setdefault(k[:90], {})[name] = {"dispatches": len(v), "mean_KB": sum(v) / len(v)}
if res:
    (out / f"{tag}_pmc_hbm.json").write_text(json.dumps(res, indent=1) + "\n")
    dec = next((v for k, v in res.items() if "am_decode" in k), None)
    probe = next((v for k, v in res.items() if "hbm_read_probe" in k), None)
    if dec and "FETCH_SIZE" in dec:
        # gfx950: FETCH_SIZE reports 1/2 of a wide coalesced stream (MI355X_MICROARCH.md §HBM); the
        # factor is re-derived here from the probe kernel of the same run (2 GiB read per dispatch)
        corr = 2.0
        if probe and "FETCH_SIZE" in probe:
            corr = (2 << 30) / (probe["FETCH_SIZE"]["mean_KB"] * 1024.0)
        traffic = dec["FETCH_SIZE"]["mean_KB"] * 1024.0 * corr + dec.get("WRITE_SIZE", {}).get("mean_KB", 0.0) * 1024.0
        bench_line = json.loads((src / "bench_trace.json").read_text()) if (src / "bench_trace.json").exists() else {}
        key = bench_line.get("config", {}).get("workload", "unknown")
        pt = out / "pmc_traffic.json"
        table = json.loads(pt.read_text()) if pt.exists() else {}
        table[key] = {"traffic_bytes_per_launch": traffic, "fetch_correction": corr, "source": f"profiles/{tag}_pmc_hbm.json"}
        pt.write_text(json.dumps(table, indent=1) + "\n")
        print("traffic", key, traffic / 1e9, "GB per launch, correction", corr)
