"""Builds libmsbwt_hip.so (the C-ABI library: host C++ + gfx950 kernels) in-tree with hipcc."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmsbwt_hip.so")
SOURCES = ["capi.cpp", "kernels.hip", "lanes.hip", "device_build.hip", "pair_index.hip", "gather.hip", "order.hip", "run_build.hip", "plane_index.cpp", "run_index.cpp", "npy_io.cpp", "rle_codec.cpp"]
HEADERS = ["kernels.hpp", "device_build.hpp", "pair_index.hpp", "gather.hpp", "order.hpp", "run_build.hpp", "rank_ops.hpp", "search_common.hpp", "host_pipeline.hpp", "plane_index.hpp", "run_index.hpp", "table_policy.hpp", "npy_io.hpp", "rle_codec.hpp", os.path.join("..", "..", "include", "msbwt_hip.h")]


def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not stale():
        return LIB
    objs = []
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    common = ["-O3", "-std=c++17", "-fPIC", "-Wall", "-Wextra"]
    for src in SOURCES:
        obj = os.path.join(objdir, src + ".o")
        cmd = [hipcc()] + common
        if src.endswith(".hip"):
            cmd += ["--offload-arch=gfx950"]
        cmd += ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [hipcc(), "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB] + objs + ["-lpthread", "-ldl"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
