#!/usr/bin/env python3
"""Does the time of one launch follow WHERE the index sits in HBM?  (round 4: the c4r line ran 18.0 or 20.5 ms, per process,
on one box.)  One process builds the same index several times -- plainly, behind a spacer allocation that pushes it
elsewhere, plainly again -- and times the same batch on each.   python tools/placement_probe.py [workload] [spacer GiB ...]
A NEGATIVE number is a small pad in KiB instead (hipMalloc'ed directly, kept during the trial): it shifts the library's SMALL
arrays (superblock table, side array, ticket counters -- carved from the runtime's 2 MiB fragments) and leaves the big ones alone."""
import ctypes
import gc
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import rust_msbwt_amd as msbwt  # noqa: E402
import synth  # noqa: E402


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "c4r"
    # "p:a,b,c,d" = MSBWT_PROBE_PADS for that trial (KiB in front of the superblock table, the side array, the ticket counters, the status block)
    specs = sys.argv[2:] or ["0", "0", "40", "0", "100", "0"]
    spacers = [0.0 if x.startswith("p:") else float(x) for x in specs]
    cfg = synth.CONFIGS[workload] if hasattr(synth, "CONFIGS") else None
    npy, reads = synth.workload_index(workload, 1.0)
    dev = torch.device("cuda:0")
    seed = cfg["qseed"] if cfg else 43
    q = torch.from_numpy(synth.read_kmers(reads, 31, limit=100_000_000, seed=seed)).to(dev)
    n = q.shape[0]
    out = torch.zeros(n, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    hip = ctypes.CDLL("libamdhip64.so")
    first = None
    for trial, gib in enumerate(spacers):
        os.environ.pop("MSBWT_PROBE_PADS", None)
        if specs[trial].startswith("p:"):
            os.environ["MSBWT_PROBE_PADS"] = specs[trial][2:]
        spacer = torch.empty(int(gib * 2 ** 30), dtype=torch.uint8, device=dev) if gib > 0 else None
        pad = ctypes.c_void_p()
        if gib < 0:
            assert hip.hipMalloc(ctypes.byref(pad), ctypes.c_size_t(int(-gib * 1024))) == 0
        b = msbwt.RleBWT()
        t0 = time.time()
        b.load_numpy_file(npy)
        load_s = time.time() - t0
        for _ in range(2):
            b.count_kmers_device(q.data_ptr(), 31, n, out.data_ptr(), stream)
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(8):
            b.count_kmers_device(q.data_ptr(), 31, n, out.data_ptr(), stream)
        ev1.record()
        torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1) / 8
        same = True if first is None else bool(torch.equal(first, out))
        if first is None:
            first = out.clone()
        print("trial %d (%s): spacer %7.1f GiB (< 0: KiB pad) at %#x  %.3f ms/launch  %.3e q/s  (load %.1f s, counts equal the first trial's: %s)"
              % (trial, specs[trial], gib, spacer.data_ptr() if spacer is not None else (pad.value or 0), ms, n / ms * 1e3, load_s, same), flush=True)
        del b, spacer
        gc.collect()
        if pad:
            hip.hipFree(pad)
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
