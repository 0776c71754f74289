#!/usr/bin/env python3
"""Turn the rocprofv3 CSVs merged back under gpurun_out/ into the small, tracked
summaries under profiles/ (per round):

  profiles/rNN_kernel_stats.csv   rocprofv3 --kernel-trace --stats of `python3 bench.py ...`
                                  (rows of this library's kernels; full file kept alongside)
  profiles/rNN_pmc_summary.json   mean PMC counters per launch for the scan and decode kernels
  profiles/rNN_traffic.json       HBM bytes per launch of the dominant kernel, corrected as
                                  /opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes:
                                  traffic = 2 * FETCH_SIZE + WRITE_SIZE (KiB -> bytes); FETCH_SIZE
                                  reads half of a wide (16 B/lane, full cache line) coalesced
                                  stream on gfx950, WRITE_SIZE is exact for 16 B/lane stores.
usage: tools/summarize_profiles.py <round> <prof_dir> <pmc_dir> [texts] [length]
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd, prof_dir, pmc_dir = sys.argv[1], sys.argv[2], sys.argv[3]
texts = int(sys.argv[4]) if len(sys.argv) > 4 else 1 << 20
length = int(sys.argv[5]) if len(sys.argv) > 5 else 1024
out = os.path.join(ROOT, "profiles")
os.makedirs(out, exist_ok=True)

stats = glob.glob(os.path.join(prof_dir, "*", "*kernel_stats.csv"))[0]
shutil.copy(stats, os.path.join(out, "%s_kernel_stats_full.csv" % rnd))
rows = list(csv.DictReader(open(stats)))
with open(os.path.join(out, "%s_kernel_stats.csv" % rnd), "w", newline="") as fh:
    w = csv.DictWriter(fh, fieldnames=rows[0].keys())
    w.writeheader()
    for r in rows:
        if "(anonymous namespace)::k_" in r["Name"]:
            w.writerow(r)

summary = {}
for kn, key in (("k_stream_findall<0", "k_stream_findall"), ("k_decode", "k_decode")):
    agg = collections.defaultdict(list)
    for f in glob.glob(os.path.join(pmc_dir, "set*", "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if kn in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    summary[key] = {k: sum(v) / len(v) for k, v in sorted(agg.items())}
    summary[key]["launches_sampled"] = max((len(v) for v in agg.values()), default=0)
json.dump(summary, open(os.path.join(out, "%s_pmc_summary.json" % rnd), "w"), indent=1, sort_keys=True)

s = summary["k_stream_findall"]
fetch_raw = s["FETCH_SIZE"] * 1024.0
write = s["WRITE_SIZE"] * 1024.0
traffic = {
    "kernel": "k_stream_findall", "config": {"texts_per_gpu": texts, "text_bytes": length},
    "fetch_size_raw_bytes": fetch_raw, "fetch_size_corrected_bytes": 2.0 * fetch_raw,
    "write_size_bytes": write, "traffic_bytes_per_launch": 2.0 * fetch_raw + write,
    "correction": "FETCH_SIZE x2 (gfx950 wide coalesced streaming read), WRITE_SIZE exact; "
                  "separate --pmc passes (tools/pmc_pass.sh)",
}
# round 4: the whole findall STEP, not only its dominant kernel -- scan + decode, same correction
d = summary.get("k_decode", {})
if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
    d_fetch, d_write = d["FETCH_SIZE"] * 1024.0, d["WRITE_SIZE"] * 1024.0
    traffic["step"] = {
        "k_stream_findall_bytes": 2.0 * fetch_raw + write,
        "k_decode_fetch_size_raw_bytes": d_fetch, "k_decode_write_size_bytes": d_write,
        "k_decode_bytes": 2.0 * d_fetch + d_write,
        "step_bytes": 2.0 * fetch_raw + write + 2.0 * d_fetch + d_write,
        "note": "scan + decode (the prefix-sum launch between them moves 0.2 MB); FETCH_SIZE x2 + WRITE_SIZE per kernel",
    }
json.dump(traffic, open(os.path.join(out, "%s_traffic.json" % rnd), "w"), indent=1, sort_keys=True)
print(json.dumps(traffic))
