#!/usr/bin/env python3
"""Which kernel family serves findall of each generated pattern (tests/pattern_gen.py, the seeds of the GPU parity tests),
read from the compiled plans' describe() -- no GPU needed.  Mirrors FindallJob::choose_route (mrx_kernels.hip).
usage: python tools/r04_route_census.py"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import mojo_regex_amd as M  # noqa: E402
from pattern_gen import patterns  # noqa: E402


def route(d):
    if "support.search=yes" not in d:
        return "refused"
    if "device.streamable=yes" in d:
        return "streaming automaton (single-walk proof)"
    if "streamable=no: anchored" in d:
        return "anchored: one walk per text"
    if "lazy_end_cache=yes" in d:
        return "'$' on the LazyDFA search (per-text cache, stepper LZ form / generic)"
    if "empty_matches=1" in d:
        return "empty matches: one pass on k_mwalk" if "empty_walk=1" in d else "empty matches: stepper (count + emit)"
    if "steppable=required-byte route" in d:
        return "required-byte route on k_mwalk / k_req_wave" if "multiwalk_req=yes" in d else "required-byte route: stepper / k_req_wave"
    if " multiwalk=yes" in d:
        return "multi-walk table (k_mwalk)"
    if "tries_walk=yes" in d:
        return "pending tries beside the oldest walk (k_mwalk, round 4)"
    if "backset=yes" in d and "device.steppable=yes" in d:
        return "backward marks + stepper (k_backscan + k_wstep)"
    if "bitset=1" in d:
        return "bitset NFA (k_bscan / k_bstep)"
    if "device.steppable=yes" in d:
        return "stepper, plain"
    return "backtracker / generic"


c = collections.Counter()
seen = set()
for seed in (20260503, 20260504, 20260505, 20260506):
    for ps in patterns(seed, 300):
        if ps in seen:
            continue
        seen.add(ps)
        try:
            rx = M.compile_regex(ps.encode())
        except Exception:
            c["syntax error (as the reference)"] += 1
            continue
        c[route(rx.describe())] += 1
tot = sum(v for k, v in c.items() if not k.startswith("syntax"))
print("| route | patterns | share |\n|---|---|---|")
for k, v in sorted(c.items(), key=lambda kv: -kv[1]):
    print("| %s | %d | %.1f %% |" % (k, v, 100.0 * v / tot if not k.startswith("syntax") else 0))
print("compiled:", tot)
