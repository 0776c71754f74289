#!/usr/bin/env python3
"""profiles/r04_bench_engine_suite.md and profiles/r04_configs.md from the round's jsonl files (tools/bench_suite.py,
tools/bench_configs.py [sub | ragged] on the final build) next to round 3's.  usage: python tools/r04_collect.py"""
import json
import os
import statistics

P = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")


def load(name):
    return [json.loads(l) for l in open(os.path.join(P, name)) if l.startswith("{")]


rows = load("r04_bench_engine_suite.jsonl")
r3 = {r["case"]: r for r in load("r03_bench_engine_suite.jsonl")}
g = [r["GBps"] for r in rows if "GBps" in r]
g3 = [r["GBps"] for r in r3.values() if "GBps" in r]
fa = [r["GBps"] for r in rows if r["op"] == "findall" and "GBps" in r]
fa3 = [r["GBps"] for r in r3.values() if r["op"] == "findall" and "GBps" in r]
out = ["# The reference's benchmark list on one MI355X (round 4, final build)", "",
       "`tools/bench_suite.py`: every case of `benchmarks/bench_engine.mojo` (restated as data in",
       "`tests/bench_engine_cases.py`), the case's text turned into a batch of rotations of itself (about 256 MiB, at most",
       "2^20 texts, device resident), the case's operation enqueued 5 times (after six untimed calls: a required-byte plan's",
       "route tuner has settled by then) and timed as a whole.  Boxes differ by 1.3-2x on these short calls, so the `r03`",
       "columns (round 3's table, another box) are a guide, not an A/B; rows whose KERNEL changed are the round-4 work",
       "(`k_mwalk_pieces` where round 3 had `k_req_wave`: the route tuner of `profiles/r04_suite_routes.md`; `sub` rows: no",
       "host synchronisation in front of the scan or the assembly, `profiles/r04_sub.md`; `sub_group_word_swap`: the chain sub,",
       "`profiles/r04_sub_chain.md`; `single_quantifier_alpha` re-measured alone).  All %d cases run; median %.0f GB/s"
       % (len(g), statistics.median(g)),
       "(round 3: %.0f), %d above 1 TB/s (%d), %d below 300 GB/s (%d), slowest %.0f GB/s.  findall rows: median %.0f (%.0f),"
       % (statistics.median(g3), sum(1 for x in g if x > 1000), sum(1 for x in g3 if x > 1000), sum(1 for x in g if x < 300),
          sum(1 for x in g3 if x < 300), min(g), statistics.median(fa), statistics.median(fa3)),
       "%d of %d below 500 GB/s (%d).  Parity of every case: `tests/test_gpu_bench_suite.py`."
       % (sum(1 for x in fa if x < 500), len(fa), sum(1 for x in fa3 if x < 500)), "",
       "| case | op | text B | texts | kernel | ms | GB/s | r03 kernel | r03 GB/s |", "|---|---|---|---|---|---|---|---|---|"]
for r in rows:
    o = r3.get(r["case"], {})
    out.append("| %s | %s | %d | %d | %s | %.3f | %.0f | %s | %s |" % (
        r["case"], r["op"], r["text_bytes"], r["texts"], r["kernel"], r["ms"], r["GBps"],
        o.get("kernel", "") if o.get("kernel") != r["kernel"] else "=", "%.0f" % o["GBps"] if "GBps" in o else ""))
open(os.path.join(P, "r04_bench_engine_suite.md"), "w").write("\n".join(out) + "\n")
print("\n".join(out[:14]))

cfg = load("r04_configs.jsonl")
c3 = {}
for r in load("r03_configs.jsonl"):
    c3[(r["config"], r.get("pattern", ""))] = r
out = ["# BASELINE.json configs on one MI355X, round 4 final build (`tools/bench_configs.py`, `sub`, `ragged`)", "",
       "Device-resident batches of each config's full per-GPU size; GB/s of input, the call as a caller makes it (findall",
       "enqueued with `findall_async`, the batch timed as a whole).  r03 = round 3's table (another box).  Changed this round:",
       "config 5 (event rows: `profiles/r04_config5_rows.md`), `sub` (`profiles/r04_sub.md`), ragged CSR (line frames:",
       "`profiles/r04_ragged.md`).", "",
       "| config | findall kernel | findall GB/s (r03) | search (r03) | count (r03) | match_first ms |", "|---|---|---|---|---|---|"]
for r in cfg:
    if "findall_GBps" not in r:
        continue
    o = c3.get((r["config"], ""), {})
    out.append("| %s | %s | %.0f (%s) | %.0f (%s) | %.0f (%s) | %.3f |" % (
        r["config"].replace("|", "\\|"), r["findall_kernel"], r["findall_GBps"], "%.0f" % o["findall_GBps"] if o else "",
        r["search_GBps"], "%.0f" % o["search_GBps"] if o else "", r["count_GBps"], "%.0f" % o["count_GBps"] if o else "",
        r["match_first_ms"]))
out += ["", "| config | sub ms | sub GB/s (r03) |", "|---|---|---|"]
for r in cfg:
    if "sub_ms" in r:
        o = c3.get((r["config"], ""), {})
        out.append("| %s | %.3f | %.0f (%s) |" % (r["config"], r["sub_ms"], r["sub_GBps"], "%.0f" % o["sub_GBps"] if o else ""))
out += ["", "Ragged CSR (config 2's texts cut to U[64, 1024] bytes, packed back to back, 0.57 GB):", "",
        "| pattern | findall GB/s, `mrx_findall_dev` (r03) | known totals | known totals, async | search (r03) | match_first ms |",
        "|---|---|---|---|---|---|"]
for r in cfg:
    if "findall_GBps_stream" in r:
        o = c3.get((r["config"], r["pattern"]), {})
        out.append("| `%s` | %.0f (%s) | %.0f | %.0f | %.0f (%s) | %.3f |" % (
            r["pattern"], r["findall_GBps_stream"], "%.0f" % o["findall_GBps_stream"] if o else "",
            r["findall_GBps_known_totals"], r["findall_async_GBps_known_totals"], r["search_GBps_stream"],
            "%.0f" % o["search_GBps_stream"] if o else "", r["match_first_ms_stream"]))
open(os.path.join(P, "r04_configs.md"), "w").write("\n".join(out) + "\n")
print("\n".join(out))
