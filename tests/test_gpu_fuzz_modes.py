"""One generator seed (300 patterns) of every mode of tests/big_fuzz.py -- product against the oracle on the GPU.
The long runs of the same script are recorded in profiles/rNN_fuzz.txt; this keeps a slice of each mode in the
suite: capture groups (sub templates, mrx_captures), the "extra" mode (texts of arbitrary bytes, fixed-pitch
layouts, the `start` argument, sub with a count, is_match, count), every pattern forced onto the NFA route and
the bitset-NFA kernels (plus split), and the default mode with the long-text kernels forced."""
import os
import re
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"MRX_FUZZ_GROUPS": "1"}, {"MRX_FUZZ_EXTRA": "1"}, {"MRX_FUZZ_NFA": "1"},
                                 {"MRX_LONG_TEXT_MODE": "1", "MRX_FUZZ_NFA": "1"}, {}],
                         ids=["groups", "extra", "nfa", "nfa_long_texts", "default"])
def test_big_fuzz_mode_agrees_with_the_oracle(env):
    if not _gpu():
        pytest.skip("needs a GPU")
    e = dict(os.environ)
    e.update(env)
    e["MRX_FUZZ_SEEDS"] = "45000:1"
    r = subprocess.run([sys.executable, os.path.join(HERE, "big_fuzz.py")], capture_output=True, text=True, timeout=900, env=e)
    assert r.returncode == 0, r.stderr[-2000:]
    last = [ln for ln in r.stdout.splitlines() if ln.startswith("seed ")][-1]
    m = re.match(r"seed 45000 checked (\d+) bad (\d+)", last)
    assert m, last
    assert int(m.group(1)) > 1000 and int(m.group(2)) == 0, r.stdout[-3000:]
