"""What a memory budget costs (msbwt_rle_set_memory_budget): queries/s of a config's batch with the index held to 100 %, 50 %,
25 % ... of what the automatic plan would take.  usage: python tools/budget_bench.py <c2|c3|c4|c4r> [fractions ...]"""
import json
import sys
import time

sys.path.insert(0, '.')
import numpy as np
import torch

import rust_msbwt_amd as m
import synth

name = sys.argv[1] if len(sys.argv) > 1 else "c2"
fractions = [float(x) for x in sys.argv[2:]] or [1.0, 0.5, 0.25, 0.1, 0.03]
cfg = synth.CONFIGS[name]
npy, reads = synth.workload_index(name)
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream(dev).cuda_stream
b = m.RleBWT(device=0)
b.load_numpy_file(npy)
full = b.device_bytes()
k = cfg["k"]
nq = cfg["nq"] or 20_000_000
batches = {"read-derived": torch.from_numpy(synth.read_kmers(reads, k, limit=nq, seed=cfg["qseed"])).to(dev),
           "random": torch.from_numpy(synth.random_kmers(nq, k, cfg["qseed"])).to(dev)}
rows = []
for frac in fractions:
    budget = 0 if frac >= 1.0 else int(full * frac)
    b.set_memory_budget(budget)
    row = {"budget_fraction": frac, "budget_bytes": budget, "index_bytes": b.device_bytes(), "table_depth": b.get_table_depth(), "table_packed": b.get_table_packed(),
           "pair_index": bool(b.get_pair_index()), "pair_stride": b.get_pair_stride()}
    for kind, d_q in batches.items():
        out = torch.zeros(d_q.shape[0], dtype=torch.int64, device=dev)
        for _ in range(2):
            b.count_kmers_device(d_q.data_ptr(), k, d_q.shape[0], out.data_ptr(), stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            b.count_kmers_device(d_q.data_ptr(), k, d_q.shape[0], out.data_ptr(), stream)
        torch.cuda.synchronize()
        row[kind + "_qps"] = d_q.shape[0] * 10 / (time.perf_counter() - t0)
        row[kind + "_checksum"] = int(out.sum().item())
    rows.append(row)
    print(json.dumps(row), flush=True)
assert len({(r["read-derived_checksum"], r["random_checksum"]) for r in rows}) == 1, "counts changed with the budget"
print("counts identical under every budget")
