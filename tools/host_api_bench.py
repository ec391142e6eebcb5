"""Throughput of the HOST-pointer batch entry points (PCIe copies included): msbwt_rle_count_kmers and
msbwt_rle_count_read_kmers through the pinned three-stage pipeline (csrc/host_pipeline.hpp).
The reported `value` of bench.py never includes PCIe; these are the numbers DESIGN.md section 5 quotes."""
import sys, time, numpy as np
sys.path.insert(0, '.')
import rust_msbwt_amd as m
import synth
from oracle import oracle as orc
npy, rd = synth.workload_index('c2')
b = m.RleBWT(); b.load_numpy_file(npy)
ref = orc.OracleRleBWT(); ref.load_numpy_file(npy)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
q = synth.random_kmers(n, 21, 3)
out = np.zeros(n, dtype=np.uint64)  # touched: no first-touch page faults inside the timed region
b.count_kmers(q[:1000])
for _ in range(4):
    t = time.time(); c = b.count_kmers(q, out=out); dt = time.time() - t
    print("host API count_kmers: %.3e 21-mers/s (%.1f ms, %.2f GB/s in+out)" % (len(q) / dt, dt * 1e3, (q.nbytes + c.nbytes) / dt / 1e9), flush=True)
sel = np.random.default_rng(1).choice(n, size=200_000, replace=False)
assert np.array_equal(out[sel], ref.count_kmers(q[sel], nthreads=8)), "host path differs from the oracle"
# the compact form: 2-bit packed 31-mers, 8 bytes in and 8 or 4 bytes out per query (msbwt_rle_count_kmers_packed)
q31 = synth.random_kmers(n, 31, 5)
w31 = m.rle_bwt.pack_2bit(q31)
for bits in (64, 32):
    o31 = np.zeros(n, dtype=np.uint64 if bits == 64 else np.uint32)
    for _ in range(4):
        t = time.time(); c = b.count_kmers_packed(w31, 31, count_bits=bits, out=o31); dt = time.time() - t
        print("host API count_kmers_packed, %d-bit counts: %.3e 31-mers/s (%.1f ms, %.2f GB/s in+out)" % (bits, len(w31) / dt, dt * 1e3, (w31.nbytes + c.nbytes) / dt / 1e9), flush=True)
    assert np.array_equal(c[sel].astype(np.uint64), ref.count_kmers(q31[sel], nthreads=8)), "packed host path differs from the oracle"
for _ in range(3):
    t = time.time(); c = b.count_kmers(q31, out=out); dt = time.time() - t
    print("host API count_kmers (bytes), the same 31-mers: %.3e 31-mers/s (%.1f ms)" % (len(q31) / dt, dt * 1e3), flush=True)
del q31, w31
reads = rd
of = np.zeros((reads.shape[0], reads.shape[1] - 30), dtype=np.uint64)
oc = np.zeros_like(of)
for _ in range(4):
    t = time.time(); f, r = b.count_read_kmers(reads, 31, ascii=False, revcomp=True, out_fwd=of, out_rc=oc); dt = time.time() - t
    print("host API count_read_kmers both strands: %.3e windows/s (%.1f ms, %d reads)" % (2 * f.size / dt, dt * 1e3, len(reads)), flush=True)
k = 31
ids = np.random.default_rng(2).choice(f.size, size=100_000, replace=False)
w = reads.shape[1] - k + 1
win = np.ascontiguousarray(reads[(ids // w)[:, None], (ids % w)[:, None] + np.arange(k)[None, :]])
assert np.array_equal(f.reshape(-1)[ids], ref.count_kmers(win, nthreads=8)), "fused host path differs from the oracle"
print("parity ok")
# single-call latency through the trait-shaped entry points (a C host: examples/call_latency.c) next to the CPU restatement
import os, subprocess
exe = "/tmp/call_latency"
subprocess.check_call(["gcc", "-O2", "-Iinclude", "examples/call_latency.c", "-Lrust-msbwt_amd", "-lmsbwt_hip",
                       "-Wl,-rpath," + os.path.abspath("rust-msbwt_amd"), "-o", exe])
print(subprocess.run([exe, npy, "21"], capture_output=True, text=True).stdout, flush=True)
qs = q[:200_000]
t = time.time(); ref.count_kmers(qs, nthreads=1); dt = time.time() - t
print("CPU restatement (oracle, 1 thread), same index: %.2f us per 21-mer query" % (dt / len(qs) * 1e6))
