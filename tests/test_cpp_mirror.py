"""The C++ mirror of the reference's trait (include/msbwt_hip.hpp): it compiles and links against
the C-ABI library everywhere; on a GPU box the reference's own RleBWT tests run through it."""
import os
import subprocess

import pytest

from conftest import GOLDEN_DIR, ROOT

SRC = os.path.join(ROOT, "tests", "cpp", "test_rle_bwt.cpp")
LIBDIR = os.path.join(ROOT, "rust-msbwt_amd")


def _build(out):
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"), SRC, "-o", out,
           "-L", LIBDIR, "-lmsbwt_hip", "-Wl,-rpath," + LIBDIR, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib",
           "-Wl,--allow-shlib-undefined"]
    subprocess.check_call(cmd)


def test_cpp_mirror_compiles_and_links(tmp_path):
    _build(str(tmp_path / "t"))
    # without arguments it only prints its usage (no GPU touched)
    r = subprocess.run([str(tmp_path / "t")], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stdout


@pytest.mark.gpu
def test_reference_tests_through_cpp_mirror(tmp_path):
    exe = str(tmp_path / "t")
    _build(exe)
    r = subprocess.run([exe, os.path.join(GOLDEN_DIR, "two_string.npy"), str(tmp_path)], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all C++ trait-mirror tests passed" in r.stdout
