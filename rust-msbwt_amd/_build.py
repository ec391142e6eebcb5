"""Builds libmsbwt_hip.so (the C-ABI library: host C++ + gfx950 kernels) in-tree with hipcc."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmsbwt_hip.so")
SOURCES = ["capi.cpp", "kernels.hip", "lanes.hip", "lanes_tier.hip", "lanes_wide.hip", "lanes_xwide.hip", "sparse_table.hip", "device_build.hip", "pair_index.hip", "gather.hip", "order.hip", "run_build.hip", "plane_index.cpp", "run_index.cpp", "npy_io.cpp", "rle_codec.cpp"]
HEADERS = ["kernels.hpp", "lanes_kernel.hpp", "sparse_table.hpp", "sparse_build.hpp", "sparse_policy.hpp", "device_build.hpp", "pair_index.hpp", "gather.hpp", "order.hpp", "run_build.hpp", "rank_ops.hpp", "search_common.hpp", "host_pipeline.hpp", "plane_index.hpp", "run_index.hpp", "table_policy.hpp", "npy_io.hpp", "rle_codec.hpp", os.path.join("..", "..", "include", "msbwt_hip.h")]


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
    from concurrent.futures import ThreadPoolExecutor

    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    common = ["-O3", "-std=c++17", "-fPIC", "-Wall", "-Wextra"]
    newest_header = max(os.path.getmtime(os.path.join(CSRC, f)) for f in HEADERS)
    newest_header = max(newest_header, os.path.getmtime(os.path.abspath(__file__)))

    def compile_one(src):
        obj = os.path.join(objdir, src + ".o")
        path = os.path.join(CSRC, src)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(path), newest_header):
            return obj  # this object is newer than its source and every header
        cmd = [hipcc()] + common
        if src.endswith(".hip"):
            cmd += ["--offload-arch=gfx950"]
        cmd += ["-c", path, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        return obj

    # the translation units are independent: compiled side by side (the lanes kernel's instantiations alone take most of a minute)
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        objs = list(pool.map(compile_one, SOURCES))
    cmd = [hipcc(), "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB] + objs + ["-lpthread", "-ldl"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
