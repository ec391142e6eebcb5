"""Plane blocks (the product's index) against the run-block layout BASELINE.json's north_star sketches
(tools/runblock_lab.hip), on the same RLE streams, same queries, same launch shape (8 lanes per rank,
one 128-byte line per rank): bytes per symbol, ranks per second, and bit-exact agreement.
   python tools/runblock_bench.py [symbols of the synthetic stream] [c4]
"""
import ctypes as C
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import rust_msbwt_amd as msbwt  # noqa: E402
import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402

SO = os.path.join(ROOT, "tools", "librunblock_lab.so")
SRC = os.path.join(ROOT, "tools", "runblock_lab.hip")
if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(SRC):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-o", SO, SRC])
msbwt._lib.lib()  # maps torch's HIP runtime first (one runtime per process)
L = C.CDLL(SO)
L.rb_build.restype = C.c_void_p
L.rb_build.argtypes = [C.c_void_p, C.c_size_t]
L.rb_info.argtypes = [C.c_void_p] + [C.POINTER(C.c_uint64)] * 4
L.rb_rank.restype = C.c_double
L.rb_rank.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int]
L.rb_free.argtypes = [C.c_void_p]


def measure(name, rle, nq=1 << 26, iters=5):
    dev = torch.device("cuda", 0)
    t0 = time.time()
    lab = L.rb_build(rle.ctypes.data_as(C.c_void_p), rle.size)
    assert lab, "rb_build failed"
    t_build = time.time() - t0
    total, nblocks, nover, pieces = (C.c_uint64() for _ in range(4))
    L.rb_info(lab, C.byref(total), C.byref(nblocks), C.byref(nover), C.byref(pieces))
    total, nblocks, nover, pieces = total.value, nblocks.value, nover.value, pieces.value
    b = msbwt.RleBWT(device=0)
    b.set_table_depth(0)
    b.set_pair_index(0)
    b.load_vector(rle)
    assert b.get_total_size() == total
    gen = torch.Generator(device=dev)
    gen.manual_seed(1)
    pos = (torch.rand(nq, device=dev, dtype=torch.float64, generator=gen) * (total + 1)).to(torch.int64).clamp_(0, total)
    syms = torch.tensor([1, 2, 3, 5, 0, 4], dtype=torch.uint8, device=dev)[torch.randint(0, 6, (nq,), device=dev, generator=gen)]
    out_p = torch.zeros(nq, dtype=torch.int64, device=dev)
    out_h = torch.zeros(nq, dtype=torch.int64, device=dev)
    out_r = torch.zeros(nq, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream

    def planes():
        b.constrain_ranges_device(syms.data_ptr(), pos.data_ptr(), pos.data_ptr(), nq, out_p.data_ptr(), out_h.data_ptr(), stream)
    planes()
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        planes()
    e1.record()
    torch.cuda.synchronize(dev)
    ms_planes = e0.elapsed_time(e1) / iters
    b.device_status(stream)
    ms_runs = L.rb_rank(lab, syms.data_ptr(), pos.data_ptr(), nq, out_r.data_ptr(), iters)
    torch.cuda.synchronize(dev)
    assert ms_runs > 0
    same = bool(torch.equal(out_p, out_r))
    # a sample against the CPU oracle (restated RleBWT::constrain_range)
    ref = orc.OracleRleBWT()
    ref.load_vector(rle)
    ids = np.random.default_rng(3).choice(nq, size=min(nq, 200_000), replace=False)
    h_pos = pos[torch.from_numpy(ids).to(dev)].cpu().numpy().astype(np.uint64)
    h_sym = syms[torch.from_numpy(ids).to(dev)].cpu().numpy()
    ol, _ = ref.constrain_ranges(h_sym, h_pos, h_pos)
    ok_oracle = bool(np.array_equal(out_r[torch.from_numpy(ids).to(dev)].cpu().numpy().astype(np.uint64), ol))
    run_bytes = nblocks * 128 + nover * 256
    plane_bytes = ((total >> 8) + 1) * 128
    print("%s: %d symbols, %d RLE bytes (%.3f B/symbol on disk)" % (name, total, rle.size, rle.size / total))
    print("  run blocks R512: %.3f B/symbol (%d blocks, %.2f%% overflow, mean %.1f pieces/block), host build %.1fs"
          % (run_bytes / total, nblocks, 100.0 * nover / nblocks, pieces / nblocks, t_build))
    print("  plane blocks   : %.3f B/symbol" % (plane_bytes / total))
    print("  ranks/s, %d random (symbol, position) pairs, 8 lanes and one 128-byte line per rank:" % nq)
    print("    plane blocks (k_constrain_ranges, both bounds = 2 ranks/query): %.3e ranks/s (%.2f ms)" % (2 * nq / ms_planes * 1e3, ms_planes))
    print("    run blocks   (k_rank_runs, one rank/query)                   : %.3e ranks/s (%.2f ms)" % (nq / ms_runs * 1e3, ms_runs))
    print("  identical results: %s; sample equals the oracle: %s" % (same, ok_oracle))
    print("  a range of width 30 needs a second line with probability %.3f (plane) / %.3f (R512)" % (30 / 256, 30 / 512))
    L.rb_free(lab)
    assert same and ok_oracle


def selftest():
    """small streams that force every path: run length 1 (every block overflows), long runs spanning
    many blocks, mixed, multi-byte on-disk runs"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from rle_random import random_stream
    for kind in ("ones", "short", "long", "mixed"):
        rle = np.ascontiguousarray(random_stream(11, 40000, kind))
        measure("selftest " + kind, rle, nq=1 << 16, iters=1)


if __name__ == "__main__":
    selftest()
    nsym = int(float(sys.argv[1])) if len(sys.argv) > 1 else 4_500_000_000
    rle, _ = synth.rle_stream(nsym, 6.0, 77)
    measure("synthetic stream (mean run 6)", np.ascontiguousarray(rle))
    if "c4" in sys.argv[2:]:
        npy, _ = synth.workload_index("c4")
        measure("C4 real MSBWT", np.fromfile(npy, dtype=np.uint8, offset=96))
