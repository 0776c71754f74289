"""The C ABI from plain C: tests/c/abi_example.c is compiled with gcc against include/mrx.h and
linked with libmrx_hip.so (CPU check: it builds and links); on a GPU box it is run and its output
compared with the reference's known answers (SURVEY.md Appendix B pins)."""
import os
import subprocess

import pytest

import mojo_regex_amd as M

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    exe = str(tmp_path / "abi_example")
    libdir = os.path.join(ROOT, "mojo_regex_amd")
    M.load_library()   # builds nothing, but fails loudly if the library is missing
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "c", "abi_example.c"), "-o", exe,
                    "-L", libdir, "-lmrx_hip", "-Wl,-rpath," + libdir], check=True)
    return exe


def test_c_client_compiles_and_links(tmp_path):
    assert os.path.exists(_build(tmp_path))


@pytest.mark.gpu
def test_c_client_output(tmp_path):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("gpu test on a box without a GPU")
    # a standalone process: the library resolves the system HIP runtime it was linked against
    r = subprocess.run([_build(tmp_path)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    out = r.stdout.splitlines()
    assert out[0] == "engine=DFA"
    assert "total=3" in out
    assert "text0 [0,8)" in out and "text0 [9,17)" in out and "text2 [2,6)" in out
    assert "search0 0 8" in out and "search1 -1 -1" in out and "search2 2 6" in out
    assert "sub=# #noneQQ#ZZ" in out
    assert "syntax: Missing closing ']'." in out
