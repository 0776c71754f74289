import sys, zlib
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
import mojo_regex_amd as M
import mrx_ref as O
from mrx_ref.hybrid import UnsupportedByOracle
from pattern_gen import patterns
def rtexts(rng, n, max_len, alphabet):
    al = np.frombuffer(alphabet, dtype=np.uint8)
    lens = rng.integers(0, max_len + 1, size=n)
    return [bytes(rng.choice(al, size=int(k)).tolist()) for k in lens]
import os
# MRX_LONG_TEXT_MODE=1: force the wavefront-per-text stepper kernels and the pieces path (200-byte
# pieces) on every plan that has them, with texts long enough to span several blocks / pieces
MODE = int(os.environ.get("MRX_LONG_TEXT_MODE", "0"))
M.load_library().mrx_debug_long_text_kernels(MODE)
bad = 0; checked = 0
# MRX_FUZZ_SEEDS=first:count picks other generator seeds (default 30000:12, the set profiles/rNN_fuzz.txt quotes)
GROUPS = os.environ.get("MRX_FUZZ_GROUPS", "0") == "1"
SEED0, NSEEDS = (int(x) for x in os.environ.get("MRX_FUZZ_SEEDS", "30000:12").split(":"))
for seed in range(SEED0, SEED0 + NSEEDS):
    rng = np.random.default_rng(seed)
    texts = rtexts(rng, 30, 60, b"abcxyz019 -@.") + rtexts(rng, 12, 220, b"abcfoobarhellocatdog0123456789 xyz@.-") + [b"", b"a", b"foobar", b"hello", b"abc123", b"cat dog", b"http://id.no", b"q"*150+b"1"]
    if MODE:
        texts += rtexts(rng, 6, 2600, b"abcfoobarhellocatdog0123456789 xyz@.-") + [b"ab" * 700 + b"12 " + b"7" * 1300 + b"-5 x@y.z"]
    for p in patterns(seed, 300):
        pb = p.encode()
        try: rx = M.compile_regex(pb)
        except M.RegexSyntaxError: continue
        # MRX_FUZZ_GROUPS=1: capture groups instead -- sub with \\1 / \\2 and the group spans of the first match
        # (the backtracking matcher's flat program, or the fixed-width form) against the oracle
        if GROUPS:
            if rx.num_groups < 1: continue
            repl = b"<\\1>" if rx.num_groups == 1 else b"<\\2|\\1>"
            orx = O.compile_regex(pb)
            try: got = rx.sub(repl, texts, 0)
            except M.UnsupportedPattern: got = None
            if got is not None:
                for t, g in zip(texts, got):
                    try: w = orx.sub(repl, t, 0)
                    except (UnsupportedByOracle, O.ReferenceDoesNotTerminate): continue
                    checked += 1
                    if g != w:
                        bad += 1
                        if bad < 10: print("MISMATCH group sub", repr(p), t, g, w)
            try: caps = rx.captures(texts)
            except M.UnsupportedPattern: caps = None
            if caps is not None and orx.fixed_total_width < 0:
                ng = rx.num_groups
                bt = getattr(getattr(orx.matcher, "nfa_matcher", None), "backtrack", None)
                for t, c in zip(texts, caps if bt is not None else []):
                    try: m, groups = bt.match_next_with_groups(t, 0)
                    except (UnsupportedByOracle, O.ReferenceDoesNotTerminate): continue
                    want = [(-1, -1)] * (ng + 1)
                    if m is not None:
                        for gid, gs, ge in groups:
                            if 1 <= gid <= ng: want[gid - 1] = (gs, ge)
                        want[ng] = m
                    checked += 1
                    if [tuple(int(x) for x in r) for r in c] != want:
                        bad += 1
                        if bad < 10: print("MISMATCH captures", repr(p), t, [tuple(int(x) for x in r) for r in c], want)
            continue
        for op in ("findall", "search", "match_first"):
            try:
                if op == "findall": got = rx.findall_lists(texts)
                else:
                    s, e = (rx.match_next if op == "search" else rx.match_first)(texts)
                    got = [(int(a), int(b)) if a >= 0 else None for a, b in zip(s, e)]
            except M.UnsupportedPattern:
                continue
            for t, g in zip(texts, got):
                try: w = getattr(O, op)(pb, t)
                except (UnsupportedByOracle, O.ReferenceDoesNotTerminate): continue
                checked += 1
                if g != w:
                    bad += 1
                    if bad < 10: print("MISMATCH", repr(p), op, t, g, w)
        # regex.sub with a literal replacement (spans route where the plan allows it)
        try:
            got = rx.sub(b"#", texts, 0)
        except M.UnsupportedPattern:
            got = None
        if got is not None:
            for t, g in zip(texts, got):
                try: w = O.sub(pb, b"#", t, 0)
                except (UnsupportedByOracle, O.ReferenceDoesNotTerminate): continue
                checked += 1
                if g != w:
                    bad += 1
                    if bad < 10: print("MISMATCH sub", repr(p), t, g, w)
    print("seed", seed, "checked", checked, "bad", bad, flush=True)
