#!/usr/bin/env python3
"""profiles/r03_bench_engine_suite.md from profiles/r03_bench_engine_suite.jsonl (tools/bench_suite.py on the final
build) next to round 2's table.  usage: python tools/r03_collect.py"""
import json
import os
import statistics

P = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")


def load(name):
    return [json.loads(l) for l in open(os.path.join(P, name)) if l.startswith("{")]


rows = load("r03_bench_engine_suite.jsonl")
r2 = {r["case"]: r for r in load("r02_bench_engine_suite.jsonl")}
g = [r["GBps"] for r in rows if "GBps" in r]
g2 = [r["GBps"] for r in r2.values() if "GBps" in r]
out = ["# The reference's benchmark list on one MI355X (round 3, final build)", "",
       "`tools/bench_suite.py`: every case of `benchmarks/bench_engine.mojo` (restated as data in",
       "`tests/bench_engine_cases.py`), the case's text turned into a batch of rotations of itself (about 256 MiB, at most",
       "2^20 texts, device resident), the case's operation enqueued 5 times (after three untimed calls) and timed as a",
       "whole.  Boxes differ by 1.3-2x on these short calls, so the `r02` columns (round 2's table, another box) are a",
       "guide, not an A/B; rows whose KERNEL changed are the round-3 work (`k_mwalk`: several walks in one pass;",
       "`k_backscan+...`: marks where matches begin in front of the stepper; `sub` rows: the span buffer sized by the last",
       "batch's match density).  All %d cases run; median %.0f GB/s (round 2: %.0f), %d above 1 TB/s (%d), %d below 300 GB/s (%d),"
       % (len(g), statistics.median(g), statistics.median(g2), sum(1 for x in g if x > 1000), sum(1 for x in g2 if x > 1000),
          sum(1 for x in g if x < 300), sum(1 for x in g2 if x < 300)),
       "slowest %.0f GB/s.  Parity of every case: `tests/test_gpu_bench_suite.py`." % min(g), "",
       "| case | op | text B | texts | kernel | ms | GB/s | r02 kernel | r02 GB/s |", "|---|---|---|---|---|---|---|---|---|"]
for r in rows:
    o = r2.get(r["case"], {})
    out.append("| %s | %s | %d | %d | %s | %.3f | %.0f | %s | %s |" % (
        r["case"], r["op"], r["text_bytes"], r["texts"], r["kernel"], r["ms"], r["GBps"],
        o.get("kernel", "") if o.get("kernel") != r["kernel"] else "=", "%.0f" % o["GBps"] if "GBps" in o else ""))
open(os.path.join(P, "r03_bench_engine_suite.md"), "w").write("\n".join(out) + "\n")
print("\n".join(out[:12]))
