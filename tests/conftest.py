import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    # The HIP library and the oracle's C half are build products (git-ignored).  Build them when
    # they are missing or older than their sources, so that a fresh checkout can run the suite;
    # both builds are no-ops when up to date.  (hipcc cross-compiles gfx950 without a GPU.)
    import __graft_entry__ as g
    g.build_library()
    g.build_oracle()


def has_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def oracle_backend():
    from mrx_ref.backend import OracleBackend
    return OracleBackend()
