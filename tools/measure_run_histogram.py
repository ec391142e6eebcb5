"""Measures the per-symbol run-length histogram of a real MSBWT (default: config C4's, 1.95e9 symbols, built
here if it is not cached) and writes it as synth/c4_run_histogram.json -- the data the human-scale stand-in
stream draws its runs from.  Needs ~70 GB of host memory for C4: run it on the GPU box."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import synth

name = sys.argv[1] if len(sys.argv) > 1 else "c4"
out = sys.argv[2] if len(sys.argv) > 2 else synth.HISTOGRAM_FILE
t0 = time.time()
npy, reads = synth.workload_index(name)
rle = np.load(npy, mmap_mode="r")
hist = synth.run_histogram(rle)
lengths = {}
for s in range(6):
    nz = np.nonzero(hist[s])[0]
    lengths[str(s)] = {str(int(l)): int(hist[s, l]) for l in nz}
runs = int(hist.sum())
symbols = int((hist * np.arange(hist.shape[1], dtype=np.uint64)[None, :]).sum())
cfg = synth.CONFIGS[name]
data = {"source": "config %s: MSBWT of %d synthetic %d-bp reads (%.1fx of a %d-bp random genome, %.1f%% substitutions), synth.workload_index"
                  % (name, cfg["nreads"], cfg["rlen"], cfg["nreads"] * cfg["rlen"] / cfg["genome"], cfg["genome"], cfg["err"] * 100),
        "symbols": symbols, "runs": runs, "mean_run": symbols / runs, "rle_bytes": int(rle.size),
        "runs_per_symbol": {str(s): int(hist[s].sum()) for s in range(6)}, "lengths": lengths}
with open(out, "w") as f:
    json.dump(data, f, indent=0, sort_keys=True)
print("histogram of %s: %d symbols, %d runs (mean %.2f), %.1fs -> %s" % (name, symbols, runs, symbols / runs, time.time() - t0, out))
