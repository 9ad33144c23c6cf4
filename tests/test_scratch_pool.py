"""CPU-side unit test of the scratch pool bookkeeping (optrace_amd/csrc/ot_scratch.hpp): the same header the HIP library
compiles, driven with malloc by tests/cpp/scratch_pool_test.cpp -- leases are exclusive, trim and the cap free idle blocks
only, an out-of-memory allocation frees what is idle and retries, eight threads against a trimming thread under
AddressSanitizer (the advisor's use-after-free scenario of round 3)."""
import pathlib
import subprocess

import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent


@pytest.mark.parametrize("flags", [["-O2"], ["-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer"]],
                         ids=["plain", "asan"])
def test_scratch_pool_bookkeeping(tmp_path, flags):
    exe = tmp_path / "scratch_pool_test"
    cmd = ["g++", "-std=c++17", "-pthread", *flags, "-I", str(ROOT / "optrace_amd" / "csrc"),
           str(ROOT / "tests" / "cpp" / "scratch_pool_test.cpp"), "-o", str(exe)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert r.stdout.startswith("ok ")
