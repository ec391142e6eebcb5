import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rust_msbwt_amd as msbwt
from oracle import oracle as orc
os.environ["MSBWT_SEARCH"] = "lanes"
npy = "tests/golden/two_string.npy"
b = msbwt.RleBWT(); b.load_numpy_file(npy)
o = orc.OracleRleBWT(); o.load_numpy_file(npy)
allk = np.array(list(itertools.product(range(6), repeat=4)), dtype=np.uint8)
lo, hi = int(sys.argv[1]), int(sys.argv[2])
q = np.ascontiguousarray(allk[lo:hi])
try:
    got = b.count_kmers(q)
    exp = o.count_kmers(q)
    print("range", lo, hi, "mismatches", int((got != exp).sum()), flush=True)
except Exception as e:
    print("error:", e, flush=True)
