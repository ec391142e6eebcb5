#!/usr/bin/env python3
"""Throughput of the batched constrain_range entry point (device-resident inputs)."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import torch
import rust_msbwt_amd as m
import synth
from oracle import oracle as orc
name = sys.argv[1] if len(sys.argv) > 1 else 'c3'
npy, rd = synth.workload_index(name)
b = m.RleBWT(); b.load_numpy_file(npy)
T = b.get_total_size()
n = 50_000_000
rng = np.random.default_rng(1)
l = rng.integers(0, T + 1, size=n, dtype=np.int64)
w = rng.integers(0, 64, size=n, dtype=np.int64)          # narrow ranges, like the late search steps
h = np.minimum(l + w, T)
s = np.array([1, 2, 3, 5], dtype=np.uint8)[rng.integers(0, 4, size=n)]
dev = torch.device('cuda:0')
dl, dh, ds = torch.from_numpy(l).to(dev), torch.from_numpy(h).to(dev), torch.from_numpy(s).to(dev)
ol, oh = torch.empty_like(dl), torch.empty_like(dl)
st = torch.cuda.current_stream(dev).cuda_stream
for it in range(4):
    torch.cuda.synchronize(); t = time.time()
    b.constrain_ranges_device(ds.data_ptr(), dl.data_ptr(), dh.data_ptr(), n, ol.data_ptr(), oh.data_ptr(), st)
    torch.cuda.synchronize(); dt = time.time() - t
    print("constrain_ranges: %.3e ranges/s (%.2f ms for %d)" % (n / dt, dt * 1e3, n))
b.device_status(st)
o = orc.OracleRleBWT(); o.load_numpy_file(npy)
k = 200000
el, eh = o.constrain_ranges(s[:k], l[:k].astype(np.uint64), h[:k].astype(np.uint64))
assert np.array_equal(ol[:k].cpu().numpy().view(np.uint64), el) and np.array_equal(oh[:k].cpu().numpy().view(np.uint64), eh)
print("parity ok on", k)
