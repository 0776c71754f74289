#!/usr/bin/env python3
"""Copy the outputs of tools/r02_profiles.sh (gpurun_out/r02/) into profiles/r02_* and rebuild the three tables
(configs, reference benchmark list, moved-off-generic).  usage: python tools/r02_collect.py"""
import csv, glob, json, os, shutil, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out", "r02")
P = os.path.join(ROOT, "profiles")


def newest(pat):
    f = glob.glob(pat)
    f.sort(key=os.path.getmtime)
    return f[-1]


subprocess.run([sys.executable, os.path.join(ROOT, "tools", "summarize_profiles.py"), "r02", os.path.join(G, "prof"), os.path.join(G, "pmc")],
               check=True, stdout=subprocess.DEVNULL)
rows = list(csv.DictReader(open(newest(os.path.join(G, "prof_s1", "*", "*kernel_stats.csv")))))
with open(os.path.join(P, "r02_kernel_stats_streams1.csv"), "w", newline="") as fh:
    w = csv.DictWriter(fh, fieldnames=rows[0].keys())
    w.writeheader()
    for r in rows:
        if "(anonymous namespace)::k_" in r["Name"]:
            w.writerow(r)
for a, b in (("bench_line.json", "r02_bench_line.json"), ("bench_line_streams1.json", "r02_bench_line_streams1.json"),
             ("bench_line_under_rocprofv3.json", "r02_bench_line_under_rocprofv3.json"),
             ("bench_line_streams1_under_rocprofv3.json", "r02_bench_line_streams1_under_rocprofv3.json"),
             ("suite.jsonl", "r02_bench_engine_suite.jsonl"), ("cfg.jsonl", "r02_configs.jsonl"),
             ("moved.jsonl", "r02_moved_off_generic.jsonl"), ("bench_gpus2_shared.json", "r02_bench_gpus2_shared_gpu.json")):
    shutil.copy(os.path.join(G, a), os.path.join(P, b))

# ---- configs
rows = [json.loads(l) for l in open(os.path.join(P, "r02_configs.jsonl"))]
out = ["# Round 2: all BASELINE configs on one MI355X (`tools/bench_configs.py`, `... sub`, `... ragged`; final build)", "",
       "Whole calls, inputs resident in HBM; GB/s = input bytes / wall time.  Raw lines: `r02_configs.jsonl`.", "",
       "| config | texts x B | matches | findall kernel | findall ms | findall GB/s | search GB/s | count GB/s | match_first ms |",
       "|---|---|---|---|---|---|---|---|---|"]
for r in rows:
    if "findall_ms" in r:
        out.append("| %s | %d x %d | %d | %s | %.3f | %.0f | %.0f | %.0f | %.3f |" % (
            r["config"].replace("|", "\\|"), r["texts"], r["bytes_per_text"], r["matches"], r["findall_kernel"], r["findall_ms"],
            r["findall_GBps"], r["search_GBps"], r["count_GBps"], r["match_first_ms"]))
out += ["", "c5 search reads only the first chunk of every text (every text matches within it).  `c4 bitset NFA`: findall is count + "
        "emit on the walks (k_bstep_*), search profits from the union pass.", "",
        "| sub | in bytes | out bytes | kernel | ms | GB/s of input |", "|---|---|---|---|---|---|"]
for r in rows:
    if "sub_ms" in r:
        out.append("| %s | %d | %d | %s | %.3f | %.0f |" % (r["config"], r["in_bytes"], r["out_bytes"], r["kernel"], r["sub_ms"], r["sub_GBps"]))
out += ["", "| ragged CSR (config 2's texts cut to U[64, 1024], packed) | bytes | findall kernel | findall GB/s | search GB/s | match_first ms |",
        "|---|---|---|---|---|---|"]
for r in rows:
    if "findall_GBps_stream" in r:
        out.append("| %s | %d | %s | %.0f | %.0f | %.3f |" % (r["pattern"].replace("|", "\\|"), r["bytes"], r["findall_kernel_stream"],
                                                           r["findall_GBps_stream"], r["search_GBps_stream"], r["match_first_ms_stream"]))
open(os.path.join(P, "r02_configs.md"), "w").write("\n".join(out) + "\n")

# ---- the reference's benchmark list
rows = [json.loads(l) for l in open(os.path.join(P, "r02_bench_engine_suite.jsonl"))]
r1 = {json.loads(l)["case"]: json.loads(l) for l in open(os.path.join(P, "r01_bench_engine_suite.jsonl"))}
g = [r["GBps"] for r in rows if "GBps" in r]
md = open(os.path.join(P, "r02_bench_engine_suite.md")).read()
head = md[:md.index("All ")] if "All " in md else ""
out = [head + "All %d cases run; median %.0f GB/s, %d above 1 TB/s, slowest %.0f GB/s.\nParity of every case: `tests/test_gpu_bench_suite.py`.\n" % (
    len(g), statistics.median(g), sum(1 for x in g if x > 1000), min(g)),
    "| case | op | text B | texts | kernel | ms | GB/s | r01 kernel | r01 ms |", "|---|---|---|---|---|---|---|---|---|"]
for r in rows:
    o = r1.get(r["case"], {})
    out.append("| %s | %s | %d | %d | %s | %.3f | %.0f | %s | %s |" % (r["case"], r["op"], r["text_bytes"], r["texts"], r["kernel"], r["ms"],
                                                                  r["GBps"], o.get("kernel", "refused"), o.get("ms", "")))
open(os.path.join(P, "r02_bench_engine_suite.md"), "w").write("\n".join(out) + "\n")

# ---- moved off the generic kernels
rows = [json.loads(l) for l in open(os.path.join(P, "r02_moved_off_generic.jsonl"))]
md = open(os.path.join(P, "r02_moved_off_generic.md")).read()
head, tail = md[:md.index("| pattern | op |")], md[md.index("\nNotes."):]
t = ["| pattern | op | engine | kernel now | ms | GB/s | literal restatement | ms | GB/s | speed-up |", "|---|---|---|---|---|---|---|---|---|---|"]
for r in rows:
    if "new" in r:
        t.append("| `%s` | %s | %s | %s | %.3f | %.0f | %s | %.3f | %.0f | %.1fx |" % (
            r["pattern"].replace("|", "\\|"), r["op"], r["engine"], r["new"]["kernel"], r["new"]["ms"], r["new"]["GBps"],
            r["restatement"]["kernel"], r["restatement"]["ms"], r["restatement"]["GBps"], r["speedup"]))
open(os.path.join(P, "r02_moved_off_generic.md"), "w").write(head + "\n".join(t) + "\n" + tail)

for f in ("r02_bench_line.json", "r02_bench_line_streams1.json", "r02_bench_line_streams1_under_rocprofv3.json", "r02_bench_gpus2_shared_gpu.json"):
    d = json.loads(open(os.path.join(P, f)).read().strip().split("\n")[-1])
    print(f, d["n_gpus"], d["ms_per_step"], d["value"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["config"].get("streams"),
          d.get("serial_one_stream", {}).get("ms_per_step"))
print("suite: %d cases, median %.0f GB/s, %d above 1 TB/s, slowest %.0f" % (len(g), statistics.median(g), sum(1 for x in g if x > 1000), min(g)))
