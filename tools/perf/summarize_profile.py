"""Turns gpurun_out/prof_<tag>_<config> (tools/perf/profile_round.sh) into the committed profiles/<tag>/:
kernel_stats[_<config>].csv, kernel_trace.csv (mw_* dispatches of the headline config only), pmc_all_summary.json (mean of
every counter per kernel, headline), pmc_hbm_summary[_<config>].json (FETCH_SIZE / WRITE_SIZE, KiB per launch — read by
bench.py for roofline.traffic), bench_line[_<config>].json.   usage: python tools/perf/summarize_profile.py r02b"""
import collections, csv, glob, json, os, shutil, sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dst = os.path.join(root, "profiles", tag)
os.makedirs(dst, exist_ok=True)
for cfg in ("hallway", "oneroom_rgbd", "maze", "pickup_dr"):
    src = os.path.join(root, "gpurun_out", f"prof_{tag}_{cfg}")
    if not os.path.isdir(src):
        continue
    suffix = "" if cfg == "hallway" else "_" + cfg
    shutil.copy(os.path.join(src, "trace", "bench_kernel_stats.csv"), os.path.join(dst, f"kernel_stats{suffix}.csv"))
    if cfg == "hallway":
        with open(os.path.join(src, "trace", "bench_kernel_trace.csv")) as f, open(os.path.join(dst, "kernel_trace.csv"), "w") as g:
            rows = list(csv.reader(f))
            w = csv.writer(g)
            w.writerow(rows[0])
            name = rows[0].index("Kernel_Name")
            for r in [r for r in rows[1:] if r[name].startswith("mw_")][-200:]:
                w.writerow(r)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(src, "pmc_*", "bench_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith("mw_"):
                agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    allc, hbm = {}, {}
    for k, cs in agg.items():
        allc[k], hbm[k] = {}, {}
        for c, v in cs.items():
            key = c + ("_KiB" if c in ("FETCH_SIZE", "WRITE_SIZE") else "")
            allc[k][key + "_mean"] = sum(v) / len(v)
            allc[k][c + "_n"] = len(v)
            if c in ("FETCH_SIZE", "WRITE_SIZE"):
                hbm[k][key + "_mean"] = sum(v) / len(v)
                hbm[k][c + "_n"] = len(v)
    import subprocess
    meta = {"commit": subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=root, capture_output=True, text=True).stdout.strip(),
            "command": f"rocprofv3 --pmc <one counter group per pass> --kernel-trace -- python bench.py --config {cfg} --steps 20 --warmup 5 --no-cpu-baseline --no-parity-check"}
    allc["_meta"] = meta; hbm["_meta"] = meta
    if any(len(v) > 2 for k, v in allc.items() if k != "_meta"):
        json.dump(allc, open(os.path.join(dst, f"pmc_all_summary{suffix}.json"), "w"), indent=1)
    json.dump(hbm, open(os.path.join(dst, f"pmc_hbm_summary{suffix}.json"), "w"), indent=1)
    line = open(src + ".bench.json").read().strip().splitlines()[-1]
    open(os.path.join(dst, f"bench_line{suffix}.json"), "w").write(line + "\n")
    print("==", cfg)
    print(open(os.path.join(dst, f"kernel_stats{suffix}.csv")).read()[:420])
    print(json.dumps({k: {kk: round(vv) for kk, vv in v.items() if kk.endswith("_mean")} for k, v in hbm.items() if k != "_meta"}))
