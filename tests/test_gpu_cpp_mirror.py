"""GPU: the C++ mirror of the reference's codec objects (include/pqd_encoding.hpp), through a
small C++ program built by __graft_entry__.build() and linked against libpqd.so and the CPU checker."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "test_encoding_hpp")

pytestmark = pytest.mark.gpu


def test_cpp_mirror_round_trips():
    assert os.path.exists(BIN), "run __graft_entry__.build() first"
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.pathsep.join([os.path.join(ROOT, "parquet-go_amd"), os.path.join(ROOT, "oracle"),
                                              "/opt/rocm/lib", env.get("LD_LIBRARY_PATH", "")])
    r = subprocess.run([BIN], env=env, capture_output=True, text=True, timeout=110)
    assert r.returncode == 0 and "pqd_encoding.hpp ok" in r.stdout, r.stdout + r.stderr
