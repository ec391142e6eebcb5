#!/usr/bin/env python3
"""Which array's rebuild flips the mode of the C4-sized lines?  (round 4: the same launch takes 18 or 20.7 ms per build of the
index.)  One instance: timed, then the TABLE alone is rebuilt a few times (set_table_depth: the planes and the pair blocks stay
where they are), then the pair blocks + table (set_pair_index), the batch timed after each.
   python tools/rebuild_probe.py [workload] [table rebuilds] [pair rebuilds]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import rust_msbwt_amd as msbwt  # noqa: E402
import synth  # noqa: E402


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "c4r"
    ntable = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    npair = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    npy, reads = synth.workload_index(workload, 1.0)
    dev = torch.device("cuda:0")
    q = torch.from_numpy(synth.read_kmers(reads, 31, limit=100_000_000, seed=synth.CONFIGS[workload]["qseed"])).to(dev)
    n = q.shape[0]
    out = torch.zeros(n, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    b = msbwt.RleBWT()
    b.load_numpy_file(npy)

    def timed(what):
        b.count_kmers_device(q.data_ptr(), 31, n, out.data_ptr(), stream)
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(5):
            b.count_kmers_device(q.data_ptr(), 31, n, out.data_ptr(), stream)
        ev1.record()
        torch.cuda.synchronize()
        print("%-28s %.2f ms per launch" % (what, ev0.elapsed_time(ev1) / 5), flush=True)

    timed("as loaded")
    for i in range(ntable):
        b.set_table_depth(-1)
        timed("table rebuilt (%d)" % (i + 1))
    for i in range(npair):
        b.set_pair_index(1)
        timed("pair blocks + table (%d)" % (i + 1))
    for i in range(2):
        b.set_table_depth(-1)
        timed("table rebuilt again (%d)" % (i + 1))


if __name__ == "__main__":
    main()
