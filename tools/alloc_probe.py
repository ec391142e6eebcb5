#!/usr/bin/env python3
"""(Historical: the library no longer reads MSBWT_BIG_ALLOC / MSBWT_SUPER_COPIES / MSBWT_PLACEMENT_CANDIDATES -- the experiments this tool
drove are recorded in profiles/r05_lab/two_modes.log; it still times instances of one index side by side.)
Round 5, the two modes of the C4-sized lines: does the KIND of allocation of the big random-access arrays (pair blocks, sparse
table) decide the mode?  Several instances of the same index side by side in ONE process, alternately with plain hipMalloc and
with physically contiguous memory (MSBWT_BIG_ALLOC, read per allocation), the same batch timed on each in turn.
Since the second experiment the settings to alternate are a list of ALLOC:COPIES pairs (MSBWT_BIG_ALLOC, MSBWT_SUPER_COPIES; 0 copies = the
library's own choice; an optional third field = MSBWT_PLACEMENT_CANDIDATES): does the number of copies of the small, hot superblock table decide it?
   python tools/alloc_probe.py [workload] [instances per setting] [rounds] [sparse: auto|0] [settings, e.g. malloc:1,malloc:32]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import rust_msbwt_amd as msbwt  # noqa: E402
import synth  # noqa: E402


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "c4r"
    ninst = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    os.environ["MSBWT_SPARSE_TABLE"] = sys.argv[4] if len(sys.argv) > 4 else "auto"
    npy, reads = synth.workload_index(workload, 1.0)
    dev = torch.device("cuda:0")
    q = torch.from_numpy(synth.read_kmers(reads, 31, limit=100_000_000, seed=synth.CONFIGS[workload]["qseed"])).to(dev)
    n = q.shape[0]
    out = torch.zeros(n, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream

    def time_on(b, launches=5):
        b.count_kmers_device(q.data_ptr(), 31, n, out.data_ptr(), stream)
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(launches):
            b.count_kmers_device(q.data_ptr(), 31, n, out.data_ptr(), stream)
        ev1.record()
        torch.cuda.synchronize()
        return ev0.elapsed_time(ev1) / launches

    settings = (sys.argv[5] if len(sys.argv) > 5 else "malloc:0,contiguous:0").split(",")
    inst = []
    for i in range(len(settings) * ninst):
        mode = settings[i % len(settings)]
        alloc, copies = mode.split(":")[:2]
        os.environ["MSBWT_PLACEMENT_CANDIDATES"] = mode.split(":")[2] if mode.count(":") > 1 else "1"
        os.environ["MSBWT_BIG_ALLOC"] = alloc
        if int(copies):
            os.environ["MSBWT_SUPER_COPIES"] = copies
        else:
            os.environ.pop("MSBWT_SUPER_COPIES", None)
        b = msbwt.RleBWT()
        b.load_numpy_file(npy)
        inst.append((mode, b))
    ref = None
    for r in range(rounds):
        for i, (mode, b) in enumerate(inst):
            t = time_on(b)
            got = out.sum().item()
            ref = got if ref is None else ref
            assert got == ref, "counts differ between instances"
            rates = ""
            if os.environ.get("ALLOC_PROBE_RATES"):
                rates = " ".join("%s %.3g" % (w, b.probe_line_rate(w)) for w in ("pair_blocks", "sparse_table"))
            b.set_search_counters(True)
            b.search_counters(stream)
            b.count_kmers_device(q.data_ptr(), 31, n, out.data_ptr(), stream)
            c = b.search_counters(stream)
            b.set_search_counters(False)
            rates += " waves that worked %d, wave steps %d" % (c["waves_worked"], c["wave_steps"])
            print("round %d  instance %d  %-14s %.2f ms per launch  (%.0f MB in HBM, sparse depth %d; random lines/s: %s)" % (r, i, mode, t, b.device_bytes() / 1e6, b.get_sparse_table(), rates), flush=True)


if __name__ == "__main__":
    main()
