import sys, time, numpy as np
sys.path.insert(0, '.')
import rust_msbwt_amd as m
import synth
npy, rd = synth.workload_index('c2')
b = m.RleBWT(); b.load_numpy_file(npy)
q = synth.random_kmers(10_000_000, 21, 3)
b.count_kmers(q[:1000])
for _ in range(3):
    t = time.time(); c = b.count_kmers(q); dt = time.time() - t
    print("host API count_kmers: %.3e q/s (%.1f ms, %.2f GB/s in+out)" % (len(q) / dt, dt * 1e3, (q.nbytes + c.nbytes) / dt / 1e9))
reads = rd[:200000]
for _ in range(2):
    t = time.time(); f, r = b.count_read_kmers(reads, 31, ascii=False, revcomp=True); dt = time.time() - t
    print("host API count_read_kmers both strands: %.3e windows/s (%.1f ms)" % (2 * f.size / dt, dt * 1e3))
