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


def _build_c_example(out):
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "count_kmers.c"), "-o", out, "-L", LIBDIR, "-lmsbwt_hip",
                           "-Wl,-rpath," + LIBDIR, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-Wl,--allow-shlib-undefined"])


def test_c_example_compiles_as_plain_c(tmp_path):
    """include/msbwt_hip.h is a C header: a C11 host builds against it."""
    exe = str(tmp_path / "count_kmers")
    _build_c_example(exe)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr


@pytest.mark.gpu
def test_c_example_counts_two_string(tmp_path):
    exe = str(tmp_path / "count_kmers")
    _build_c_example(exe)
    r = subprocess.run([exe, os.path.join(GOLDEN_DIR, "two_string.npy"), "ACGT", "TGCA", "CCCC"], capture_output=True,
                       text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "total symbols: 10" in r.stdout and "ACGT\t1" in r.stdout and "TGCA\t1" in r.stdout and "CCCC\t0" in r.stdout


def _build_rank_gather(out):
    subprocess.check_call(["gcc", "-std=gnu11", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"), "-I/opt/rocm/include",
                           os.path.join(ROOT, "examples", "rank_gather.c"), "-o", out, "-L", LIBDIR, "-lmsbwt_hip",
                           "-Wl,-rpath," + LIBDIR, "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"])


def test_rank_gather_example_compiles(tmp_path):
    """examples/rank_gather.c: a C host of the one-process-per-GPU form (msbwt_comm_*, msbwt_rle_allgather_counts)."""
    exe = str(tmp_path / "rank_gather")
    _build_rank_gather(exe)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr


@pytest.mark.gpu
def test_rank_gather_example_one_rank(tmp_path):
    exe = str(tmp_path / "rank_gather")
    _build_rank_gather(exe)
    r = subprocess.run([exe, os.path.join(GOLDEN_DIR, "two_string.npy"), "0", "1", str(tmp_path / "id"), "ACGT", "TGCA", "CCCC"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ACGT\t1" in r.stdout and "TGCA\t1" in r.stdout and "CCCC\t0" in r.stdout
