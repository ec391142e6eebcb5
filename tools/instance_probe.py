#!/usr/bin/env python3
"""Is the speed of a launch a property of the INDEX INSTANCE?  (round 4: the same c4r batch took 18 or 20.7 ms per launch, per
build of the index, with every array at the same address.)  Builds several instances of the same index side by side, times the
same batch on each of them in turn, several rounds; then the same without the packed table's side array -- the one thing the
build does not lay out deterministically (escape lines take their side groups through an atomic cursor).
   python tools/instance_probe.py [workload] [instances] [rounds] [sides, e.g. 1 or 10]"""
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
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    npy, reads = synth.workload_index(workload, 1.0)
    dev = torch.device("cuda:0")
    q = torch.from_numpy(synth.read_kmers(reads, 31, limit=100_000_000, seed=synth.CONFIGS[workload]["qseed"])).to(dev)
    n = q.shape[0]
    out = torch.zeros(n, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream

    def time_on(b, launches=6):
        b.count_kmers_device(q.data_ptr(), 31, n, out.data_ptr(), stream)
        torch.cuda.synchronize()
        times = []
        for _ in range(launches):
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            b.count_kmers_device(q.data_ptr(), 31, n, out.data_ptr(), stream)
            ev1.record()
            torch.cuda.synchronize()
            times.append(ev0.elapsed_time(ev1))
        return times

    sides = [int(c) for c in (sys.argv[4] if len(sys.argv) > 4 else "10")]
    for side in sides:
        inst = []
        for _ in range(ninst):
            b = msbwt.RleBWT()
            b.set_table_side(side)
            b.load_numpy_file(npy)
            inst.append(b)
        for r in range(rounds):
            for i, b in enumerate(inst):
                t = time_on(b)
                print("side array %d  round %d  instance %d: %s ms per launch" % (side, r, i, " ".join("%.2f" % x for x in t)), flush=True)
        del inst, b
        import gc
        gc.collect()


if __name__ == "__main__":
    main()
