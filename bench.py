#!/usr/bin/env python3
"""bench.py -- batched count_kmer throughput of the MI355X path (BASELINE.json metric).

One "step" = one pass of the hot path (msbwt_rle_count_kmers_device) over one batch of synthetic
queries that is already resident in HBM.

Default workload ("human") = the configuration BASELINE.json quotes its metric on, on ONE GPU:
k = 31 on a 30x-human-scale index -- 9e10 symbols, the EXACT multi-string BWT of 5.96e8 error-free 150-bp
reads of a random 2.98e9-bp genome, built on the GPU in the run (synth/bwt_reads.py; --stream histogram /
geometric select the independent-symbol stand-ins of rounds 1-2) -- and 3e8 PRESENT 31-mers per step
(LF-walk; every query runs all 31 steps -- the read-corrector case).  Extra keys of the default run:
  c5_random_1e9  the literal configs[4] line: 1e9 random 31-mers generated in HBM, sharded over the ranks
                 and gathered when N > 1;
  c4_repeats     (N = 1) the REAL multi-string BWT of a C4-sized read set (12.9 M reads with 0.5 % substitutions, 1.95e9
                 symbols, suffix-sorted on the host in this run) of a genome WITH REPEATS (synth.REPEAT_FAMILIES): 1e8
                 read-derived 31-mers, with parity, roofline and the kernel's own counters (lines per query, second-line rate,
                 share of queries on escape lines of the packed table); `library_ordered` = the same with the library's own
                 batch-ordering pass forced on; `pair_blocks_rebuilt` = the same batch timed again after the pair blocks (+ table)
                 were freed and rebuilt: launches on a C4-sized index run in one of two modes 15 % apart, decided by the memory
                 the pair blocks were given -- a ratio far from 1 says the line was measured in the other one;
  c4_real_reads  the same on the random genome of rounds 2-3.
Other workloads: c2, c3 (--fused = configs[2]), c4, big.  --genome repeats: the default index over a genome WITH repeats
(synth.repeat_genome at 2.98e9 bp: the human repeat classes at human copy numbers; exact MSBWT by synth/bwt_reads.py msbwt_rle_repeats)
-- a lab line beside the metric's, profiles/r05_lab/human_repeats.json.

--gpus N: one rank per GPU.  Launched plainly (no torchrun) the script starts torch.distributed.run
itself as a child process and relays its JSON line.  The index is replicated; ONE fixed batch is
sharded across the ranks (strong scaling, BASELINE configs[3]'s shape) and each step ends with the final
gather of the counts to rank 0 over RCCL (--exchange gather, the default since round 6: the root receives every
shard over that rank's own xGMI link at once) or with an all_gather after which every rank holds all counts
(--exchange allgather, rounds 1-5).  The weak-scaling figure (every rank the whole batch) is
reported in the extra key `weak_scaling`; `ranks` carries per-rank kernel and exchange times;
`native_gather` repeats the measurement with the library's own RCCL call (msbwt_rle_allgather_counts).

Prints ONE JSON line (rank 0).  DESIGN.md section 5 defines the roofline figures.
"""
import argparse
import datetime
import hashlib
import importlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s is achievable
RANDOM_LINE_PEAK = 4.86e10  # random 128-byte lines/s, measured: independent gathers over 8-250 GiB (tools/ubench_granule.hip; 64-, 32- and 16-byte
# granules are served at the same rate), dependent LDS-DMA gathers over 100 GB 4.8e10 (tools/ubench_lds_gather.hip)
HUMAN_SYMBOLS = 9e10
# everything that decides which bytes a query makes the kernel move: the kernels, the block layouts and their
# builders, and the policies that pick table depth and pair spacing
# (order.hip -- the library's batch-ordering passes -- is not among them: they are off unless forced, and no default line runs them)
KERNEL_SOURCES = ["kernels.hip", "lanes.hip", "lanes_kernel.hpp", "lanes_tier.hip", "lanes_wide.hip", "lanes_xwide.hip", "search_common.hpp", "rank_ops.hpp", "kernels.hpp", "plane_index.hpp",
                  "pair_index.hip", "device_build.hip", "table_policy.hpp", "sparse_table.hpp", "sparse_table.hip"]
NARROW_MAX = 32767


def log(msg):
    print("[bench] " + msg, file=sys.stderr, flush=True)


def kernel_stamp():
    """Identifies the query kernels a PMC summary was taken with: profiles/traffic.json entries
    carry it, and an entry whose stamp differs from the sources in the tree is refused.  Comments
    and white space do not count (a reworded comment does not invalidate a profile)."""
    import re
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "rust-msbwt_amd", "csrc", name), "r") as f:
            text = f.read()
        text = re.sub(r"//[^\n]*", "", text)          # the sources use // comments only
        h.update(" ".join(text.split()).encode())
    return h.hexdigest()[:16]


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def lookup_depth(bwt, k):
    """Symbols the suffix-table lookup of a k-symbol query stands for on this index: the sparse table's depth when it serves the
    query (k >= its depth, lanes kernel), else the direct table's (0 = none applies)."""
    sparse = bwt.get_sparse_table()
    if sparse and bwt.search_kernel_for(k) == "lanes":
        if k >= sparse:
            return sparse
        second = bwt.sparse_table_info()["second_depth"]   # round 6: a second, shallower level for the queries the first is too deep for
        if second and k >= second:
            return second
    direct = bwt.get_table_depth()
    return direct if k >= direct else 0


def kernel_label(bwt, k, fused):
    """Name (as in the rocprofv3 kernel trace) of the kernel the library runs for this index and k:
    k_count_kmers_lanes<kReads, kPair, kWords, kStride96, kPacked, kSparse> (csrc/lanes_kernel.hpp; kSparse: 0 = direct table, sparse table with
    1 = 24-bit tags, 3 = 32-bit tags (depths 25..29), 4 = 40-bit tags (30..31), 2 = the two-tier form)."""
    which = bwt.search_kernel_for(k)
    reads, words = ("true" if fused else "false"), (3 if k <= 32 else 6)
    if which == "lanes":
        pair, s96 = bwt.get_pair_index(), bwt.get_pair_index() and bwt.get_pair_stride() == 96
        info = bwt.sparse_table_info() if bwt.get_sparse_table() else None
        served = bool(info) and (k >= info["depth"] or (info["second_depth"] and k >= info["second_depth"]))
        served_depth = 0 if not served else (info["depth"] if k >= info["depth"] else info["second_depth"])
        sparse = 0 if not served else (2 if info["two_tier"] else 4 if served_depth >= 30 else 3 if served_depth >= 25 else 1)
        return "k_count_kmers_lanes<%s,%s,%d,%s,false,%d>" % (reads, "true" if pair else "false", words, "true" if s96 else "false", sparse)
    return "k_count_kmers_tiled<%s,%d>" % (reads, words) if which == "groups" else "k_count_kmers_generic"


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="human", choices=["human", "c2", "c3", "c4", "c4r", "c4x3", "c4x3r", "big"])
    ap.add_argument("--big-symbols", type=float, default=2.0**33, help="workload big: BWT length")
    ap.add_argument("--big-mean-run", type=float, default=6.0)
    ap.add_argument("--stream", default="reads", choices=["reads", "histogram", "geometric"],
                    help="human / big: what the index is built from.  reads (default) = the EXACT multi-string BWT of an error-free 30x read "
                         "set (150-bp reads of a random genome), built on the GPU without suffix-sorting the reads (synth/bwt_reads.py); "
                         "histogram / geometric = a stream of independent symbols with run lengths from C4's measured histogram / a geometric law")
    ap.add_argument("--genome", default="random", choices=["random", "repeats"],
                    help="human / big with --stream reads: the genome the error-free reads come from.  random (default: the metric's line) or repeats = "
                         "synth.repeat_genome, the human repeat classes at their own proportions and copy numbers (SINE-, LINE-, LTR-, DNA-element-like "
                         "families, segmental duplications, satellite arrays, microsatellites); its exact MSBWT takes the genome's suffix order to the "
                         "read length (synth/bwt_reads.py, msbwt_rle_repeats).  A lab line: never the default, no committed PMC summary")
    ap.add_argument("--stats-sample", type=int, default=2_000_000, help="queries used for the algorithmic-byte counters (0 = all)")
    ap.add_argument("--queries", type=int, default=0, help="queries per step, whole job (0 = the config's)")
    ap.add_argument("--k", type=int, default=0, help="override k")
    ap.add_argument("--query-kind", default="", choices=["", "random", "reads", "walk"])
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the index (tests)")
    ap.add_argument("--table-depth", type=int, default=-2, help="-2 = library default")
    ap.add_argument("--query-length-hint", type=int, default=-1, help="msbwt_rle_set_query_length before the load: the k the index is built for (default: this run's k; 0 = "
                                                                      "unknown, i.e. the automatic sparse table stops at depth 23 whatever k is)")
    ap.add_argument("--sparse-depth", type=int, default=-2, help="msbwt_rle_set_sparse_table before the load: 0 = off, 16..31 = that depth (default: the library's automatic choice)")
    ap.add_argument("--sparse-tiers", type=int, default=-2, help="msbwt_rle_set_sparse_tiers before the load: 1 = the two-tier form of the sparse table (entries for the "
                                                                  "suffixes that occur at least twice, filter bits for the rest), 0 = complete tables only (default: automatic)")
    ap.add_argument("--blocks", default="planes", choices=["planes", "runs"],
                    help="index block format: planes (default) or the memory-lean run blocks (no pair index)")
    ap.add_argument("--fused", action="store_true",
                    help="c2/c3: queries = every k-mer window of every read, prepared in-kernel from the reads "
                         "(msbwt_rle_count_read_kmers_device); BASELINE.json configs[2] is --workload c3 --fused")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL over xGMI, the real thing) or gloo (rehearsal of the N>1 path on fewer GPUs: "
                         "counts are gathered through host memory, ranks may share a GPU)")
    ap.add_argument("--device-queries", action="store_true",
                    help="random queries: generate the batch in HBM (torch PRNG) instead of uploading it")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="N>1: strong = one fixed batch sharded over the ranks (default), weak = every rank the whole batch")
    ap.add_argument("--exchange", default="gather", choices=["gather", "allgather"],
                    help="N > 1: the step's exchange -- `gather`: the final gather of the counts to rank 0 (what the north-star names; the root receives "
                         "every other rank's shard over its own xGMI link at once); `allgather`: every rank ends each step holding ALL counts (rounds 1-5)")
    ap.add_argument("--payload", default="auto", choices=["auto", "int64"],
                    help="N>1: auto = int16 counts on the wire when exact (falls back to int64), int64 = always wide")
    ap.add_argument("--sort-queries", action="store_true",
                    help="hand the batch over in the order of its batch-order keys (msbwt_rle_kmer_order_keys_device: by the last 17 symbols, then "
                         "leftwards): the timed `value` is then that of an ordered batch.  The default human run reports it as the extra key "
                         "`sorted_batch` instead")
    ap.add_argument("--sort-bits", type=int, default=0,
                    help="with --sort-queries: order by the top B bits of the key only (stable; 0 = the whole key) -- how coarse a bucket pass may be")
    ap.add_argument("--no-sorted", action="store_true", help="default workload: skip the extra `sorted_batch` measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-oracle", action="store_true",
                    help="profiling passes only: skip parity, algorithmic-byte counters and the CPU baseline (prints a lean line)")
    ap.add_argument("--no-c5", action="store_true", help="skip the extra 1e9-random-31-mer line of the default workload")
    ap.add_argument("--no-c4", action="store_true", help="skip the extra real-MSBWT (config C4, read-derived 31-mers) line of the default workload")
    ap.add_argument("--no-c4-random", action="store_true", help="default workload: only the repeat-bearing C4 line (c4_repeats), not the random-genome one (c4_real_reads)")
    ap.add_argument("--c4-scale", type=float, default=0.0, help="tests: run the extra C4 line on a shrunk C4 (0 = full size, only with --scale 1)")
    ap.add_argument("--c4-queries", type=int, default=100_000_000)
    ap.add_argument("--c4-lab", action="store_true", help="the C4 lines' lab extras: `library_ordered` (the library's batch-ordering pass forced on) and "
                                                          "`copy_number_bins` (cost by copy number, repeat-bearing line) -- round 4/5 material, off by default")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="default workload, N = 1, full scale: do NOT re-measure the kernel's HBM traffic in this run (two rocprofv3 --pmc child "
                         "passes of a shortened copy of the run, after the GPU has been handed back); roofline.traffic then comes from the committed "
                         "summary under profiles/ only")
    ap.add_argument("--no-weak", action="store_true", help="N>1: skip the extra weak-scaling measurement")
    ap.add_argument("--no-native-gather", action="store_true", help="N>1: skip the extra measurement with the library's own RCCL all-gather")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the N>1 code path (process group, all_gather of the counts, barriers) even with one rank: "
                         "a one-GPU rehearsal of the RCCL calls themselves")
    ap.add_argument("--c5-queries", type=int, default=0, help="queries of the extra random-31-mer line (0 = 1e9 at full scale, none on a shrunk index)")
    ap.add_argument("--counters", action="store_true",
                    help="one extra, untimed pass with the library's search counters on (msbwt_rle_set_search_counters): steps, second lines, "
                         "escape lines per query -> `search_counters` in the JSON line")
    ap.add_argument("--no-table-side", action="store_true", help="packed table without its side array: queries of escape lines search from scratch (round 3)")
    ap.add_argument("--extras-file", default=os.path.join(ROOT, "bench_extras.json"),
                    help="where the FULL record goes (notes, telemetry, counters, sub-lines); stdout carries one compact line (< 8 KB)")
    ap.add_argument("--no-variants", action="store_true",
                    help="default workload: skip the extra lines on the same index rebuilt with k undeclared (`undeclared_k`), without the sparse "
                         "table (`headline_sparse_off`) and their short-k timings (`short_k`)")
    ap.add_argument("--parity-sample", type=int, default=2_000_000)
    ap.add_argument("--cpu-sample", type=int, default=1_000_000)
    return ap.parse_args(argv)


def self_launch(args):
    """`python3 bench.py --gpus N` without torchrun: start one rank per GPU as a CHILD process
    (never exec: nothing here has touched the GPU, but the rule is cheap to keep) and relay its
    single JSON line and return code."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("starting %d ranks: %s" % (args.gpus, " ".join(cmd)))
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, cwd=ROOT)
    out, _ = child.communicate()
    lines = [l for l in out.splitlines() if l.strip().startswith("{")]
    for l in out.splitlines():
        if not l.strip().startswith("{"):
            print(l, file=sys.stderr)
    if lines:
        print(lines[-1], flush=True)
    return child.returncode if child.returncode else (0 if lines else 1)


def walk_kmers(torch, np, bwt, dev, total, n, k, seed, chunk=40_000_000):
    """k-mers that are present in the index, for streams that are not the BWT of known reads:
    start at a random row r with the one-row range [r, r+1) and prepend, k times, the symbol
    stored at that row (the one symbol whose constrain_range keeps the row) -- an LF walk run
    with the product's own batched constrain_ranges.  Walks that meet '$' or 'N' are dropped
    (read k-mers never contain them).  Workload generation only; outside every timed region.
    Returns an (n, k) uint8 tensor on the device."""
    rng = np.random.default_rng(seed)
    stream = torch.cuda.current_stream(dev).cuda_stream
    out = torch.empty((n, k), dtype=torch.uint8, device=dev)
    have = 0
    while have < n:
        m = min(chunk, int((n - have) * 1.3) + 4096)
        l = torch.from_numpy(rng.integers(0, total, size=m, dtype=np.int64)).to(dev)
        h = l + 1
        kmers = torch.zeros((m, k), dtype=torch.uint8, device=dev)
        ok = torch.ones(m, dtype=torch.bool, device=dev)
        ol = torch.empty_like(l)
        oh = torch.empty_like(l)
        for step in range(k):
            nl = torch.zeros_like(l)
            nh = torch.zeros_like(l)
            sym_at = torch.zeros(m, dtype=torch.uint8, device=dev)
            for s in range(6):
                syms = torch.full((m,), s, dtype=torch.uint8, device=dev)
                bwt.constrain_ranges_device(syms.data_ptr(), l.data_ptr(), h.data_ptr(), m, ol.data_ptr(), oh.data_ptr(), stream)
                hit = (oh - ol) == 1
                nl = torch.where(hit, ol, nl)
                nh = torch.where(hit, oh, nh)
                sym_at = torch.where(hit, torch.full_like(sym_at, s), sym_at)
            kmers[:, k - 1 - step] = sym_at
            ok &= (sym_at != 0) & (sym_at != 4)
            l, h = nl, nh
        bwt.device_status(stream)
        good = kmers[ok]
        take = min(n - have, good.shape[0])
        out[have:have + take] = good[:take]
        have += take
        del l, h, kmers, ok, ol, oh, nl, nh, sym_at, good
    return out


def device_random_kmers(torch, dev, lo, hi, k, seed, chunk=50_000_000):
    """Rows [lo, hi) of a batch of uniform ACGT k-mers, generated in HBM (BASELINE configs[4] asks for 1e9
    queries: 31 GB that never need to exist on the host).  The batch is defined chunk by chunk (chunk c =
    rows [c * chunk, (c + 1) * chunk), its own generator seeded seed + c), so a rank that owns a shard
    generates only the chunks its shard touches and every rank agrees on every row."""
    d_q = torch.empty((hi - lo, k), dtype=torch.uint8, device=dev)
    for c in range(lo // chunk, (max(hi, lo + 1) - 1) // chunk + 1):
        c_lo, c_hi = c * chunk, (c + 1) * chunk
        gen = torch.Generator(device=dev)
        gen.manual_seed(seed * 1_000_003 + c)
        a, b = max(lo, c_lo), min(hi, c_hi)
        if b <= a:
            continue
        whole = a == c_lo and b == c_hi
        part = d_q[a - lo:b - lo] if whole else torch.empty((chunk, k), dtype=torch.uint8, device=dev)
        part.random_(0, 4, generator=gen)      # 0..3
        part.add_(1)                           # A C G -> 1 2 3
        part.masked_fill_(part == 4, 5)        # T -> 5
        if not whole:
            d_q[a - lo:b - lo] = part[a - c_lo:b - c_lo]
            del part
    return d_q


def index_shape(bwt, k, fused, n):
    """what decides the kernel and the bytes of a launch on this index, read NOW (later table rebuilds do not change a record made of it)"""
    return {"lookup_depth": lookup_depth(bwt, k), "kernel": kernel_label(bwt, k, fused), "pair_index": bwt.get_pair_index(), "pair_stride": bwt.get_pair_stride(),
            "two_tier": bool(bwt.get_sparse_table()) and bwt.get_sparse_tiers(),
            "ordered": (not fused) and bwt.batch_order_for(k, n), "sparse_depth": bwt.get_sparse_table()}


def lookup_traffic(workload, k, shape, kind, total, fused, full_size):
    """HBM traffic per query of one launch of this configuration: PMC counters from separate rocprofv3 passes
    (FETCH_SIZE x2 on gfx950, WRITE_SIZE), committed under profiles/ and keyed by configuration AND kernel sources.
    Returns (bytes per query | None, source, note, stamp)."""
    per_query, src, note = None, None, "no PMC summary for this configuration under profiles/"
    stamp = kernel_stamp()
    try:
        for ent in json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))["entries"]:
            same = (ent["workload"] == workload and ent["k"] == k and ent["table_depth"] == shape["lookup_depth"]
                    and ent.get("pair_index", False) == shape["pair_index"] and ent.get("pair_stride", shape["pair_stride"]) == shape["pair_stride"]
                    and ent.get("query_kind") == kind and ent.get("bwt_symbols", total) == total and full_size
                    and bool(ent.get("fused", False)) == bool(fused) and bool(ent.get("two_tier", False)) == bool(shape.get("two_tier", False)))
            if same and ent.get("kernel_stamp") != stamp:
                note = "stale: %s was taken with kernel sources %s, the tree has %s" % (ent["source"], ent.get("kernel_stamp"), stamp)
            elif same:
                per_query, src, note = ent["traffic_bytes_per_query"], ent["source"], None
    except (OSError, KeyError, ValueError):
        pass
    return per_query, src, note, stamp


def live_pmc_traffic(extra_args, queries, kernel_substr="k_count_kmers", counters=("FETCH_SIZE", "WRITE_SIZE"), patience=300):
    """HBM-side bytes per query of the count kernel, measured NOW: two rocprofv3 --pmc child passes (FETCH_SIZE and
    WRITE_SIZE separately, as MI355X_MICROARCH.md prescribes; the program goes directly after `--`) over a shortened copy
    of this run -- same index (same seeds), `queries` present k-mers, one warm-up and two timed launches.  The caller
    has released its GPU memory.  Returns (bytes per query, detail dict) or (None, reason)."""
    import csv
    import glob
    import shutil
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None, "rocprofv3 not found"
    means = {}
    for counter in counters:
        out_dir = tempfile.mkdtemp(prefix="msbwt_pmc_", dir="/tmp")
        cmd = [prof, "--pmc", counter, "--kernel-include-regex", kernel_substr, "--kernel-trace", "--output-format", "csv", "-d", out_dir, "-o", "run",
               "--", sys.executable, os.path.abspath(__file__), "--no-oracle", "--no-c5", "--no-c4", "--no-live-pmc", "--no-sorted", "--no-variants", "--queries", str(queries),
               "--steps", "2", "--warmup", "1"] + list(extra_args)
        try:
            done = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=patience)
        except (OSError, subprocess.TimeoutExpired) as e:
            shutil.rmtree(out_dir, ignore_errors=True)
            return None, "%s pass failed: %r" % (counter, e)
        rows = []
        for f in glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if kernel_substr in row.get("Kernel_Name", "") and row.get("Counter_Name") == counter:
                    rows.append((int(row.get("Dispatch_Id", 0)), float(row["Counter_Value"])))
        vals = [v for _, v in sorted(rows)][1:3]   # the child's own two timed launches: not its warm-up dispatch, nothing that may follow
        shutil.rmtree(out_dir, ignore_errors=True)
        if done.returncode != 0 or not vals:
            return None, "%s pass: rc %d, %d counter rows; %s" % (counter, done.returncode, len(vals), done.stderr[-300:].replace("\n", " | "))
        means[counter] = (sum(vals) / len(vals), len(vals))
    # FETCH_SIZE / WRITE_SIZE are in KiB; gfx950 tallies 128-byte read requests at 64 bytes (MI355X_MICROARCH.md, HBM section)
    fetch_b = means["FETCH_SIZE"][0] * 1024 * 2
    # (lines measured with the read pass alone: the writes of a count launch are its counts, 8 bytes per query, and little else -- 8.3 measured)
    write_b = means["WRITE_SIZE"][0] * 1024 if "WRITE_SIZE" in means else 8.0 * queries
    return (fetch_b + write_b) / queries, {"queries_per_launch": queries, "launches": means["FETCH_SIZE"][1],
                                           "fetch_bytes_per_query_x2_gfx950": fetch_b / queries, "write_bytes_per_query": write_b / queries,
                                           "write_bytes": "measured (WRITE_SIZE pass)" if "WRITE_SIZE" in means else "taken as 8 bytes per query (the counts): no WRITE_SIZE pass for this line"}


def roofline_block(orc, ref, queries, k, ncpu, per_launch_q, kern_s, kernel_ms, launches, traffic_per_query, traffic_src, traffic_note,
                   stamp, label, stats_sample, table_depth=0, pair_index=True, ordered=False, sparse=False):
    """roofline object of one measured line ("Bytes"): counter traffic / kernel time / 8 TB/s, next to
    (a) layout_algorithmic -- the bytes the RUNNING layout must move for this query set: one 128-byte table line (k >= table
        depth), then one 128-byte line per search step -- a pair step for every two remaining symbols, a plane step for an odd
        last one -- for the share of steps the queries execute (exact step counters of the instrumented oracle), + k + 8
        bytes of query and count; and
    (b) reference_algorithm -- SURVEY 8(d)'s figure, the bytes the REFERENCE's algorithm touches for the same queries."""
    st = orc.Stats()
    nst = min(len(queries), stats_sample) if stats_sample else len(queries)
    ref.count_kmers(queries[:nst], nthreads=ncpu, stats=st)
    alg_bytes = st.algorithmic_bytes(k) / nst * per_launch_q  # exact when nst == nq, else scaled from the sample
    traffic = None if traffic_per_query is None else traffic_per_query * per_launch_q
    achieved = None if traffic is None else traffic / kern_s / 1e9
    # (a): the layout's own minimum.  mean_steps = reference steps executed per query (a query ends at its first empty range)
    mean_steps = st.steps / nst
    use_table = table_depth > 0 and k >= table_depth
    after = k - table_depth if use_table else k
    step_lines_full = ((after + 1) // 2) if pair_index else after          # pair steps (+ one plane step for an odd remainder)
    executed = 1.0 if after == 0 else max(0.0, min(1.0, (mean_steps - (table_depth if use_table else 0)) / after))
    lines = (1.0 if use_table else 0.0) + step_lines_full * executed
    layout_bytes = lines * 128.0 + k + 8
    layout = {
        "bytes_per_query": layout_bytes, "lines_per_query": lines, "table_line": 1 if use_table else 0,
        "search_lines_if_all_steps_run": step_lines_full, "share_of_steps_executed": executed,
        "frac_algorithmic": layout_bytes * per_launch_q / kern_s / 1e9 / HBM_PEAK_GBS,
        "traffic_over_algorithmic": None if traffic_per_query is None else traffic_per_query / layout_bytes,
        "note": "bytes the layout that RUNS must move per query: (1 table line if k >= depth %d) + ceil((k - depth) / 2) pair lines "
                "(+ 1 plane line for an odd remainder) x 128 B x the share of post-table steps the queries execute (oracle step counters: "
                "%.2f of %d reference steps per query) + k + 8; frac_algorithmic = that / kernel time / 8 TB/s; traffic_over_algorithmic = "
                "counter bytes / it (> 1: ranges straddling two lines, superblock words; < 1: lines served by L2 / Infinity Cache)"
                % (table_depth, mean_steps, k)}
    if sparse:
        layout["note"] += ("; the table line is a bucket of the SPARSE suffix table (depth %d: the ranges of the suffixes that occur, hashed; "
                           "about 1 %% of the lookups go on to a second bucket)" % table_depth)
    if ordered:
        layout["note"] += "; this launch includes the library's batch-ordering pass, whose streaming traffic is not part of the minimum"
    return {
        "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": None if achieved is None else achieved / HBM_PEAK_GBS,
        "traffic": traffic, "traffic_source": traffic_src, "traffic_note": traffic_note, "kernel_stamp": stamp,
        "kernel": label, "kernel_ms": kernel_ms, "kernel_launches": launches,
        "layout_algorithmic": layout,
        "reference_algorithm": {
            "bytes_per_launch": int(alg_bytes), "bytes_per_query": alg_bytes / per_launch_q,
            "GBps": alg_bytes / kern_s / 1e9, "speedup_over_hbm_peak_at_reference_bytes": alg_bytes / kern_s / 1e9 / HBM_PEAK_GBS,
            "mean_steps_per_query": mean_steps, "mean_bin_visits_per_query": st.visits / nst, "stats_queries": int(nst),
            "note": "bytes the REFERENCE algorithm touches for this query set (SURVEY 8d: 56 B of samples + scanned RLE bytes per "
                    "bin visit, + k + 8 per query) / kernel time.  The suffix table and pair steps make this kernel move fewer "
                    "bytes than that: a speed-up measure (it exceeds 1 x peak), never a bandwidth fraction"},
        "note": "achieved/frac = measured HBM-side traffic of one launch (rocprofv3 PMC: FETCH_SIZE x 2 + WRITE_SIZE, separate passes) / "
                "kernel time measured live with HIP events on the launch stream / 8 TB/s",
        "random_lines": None if traffic is None else {
            "per_s": traffic / 128.0 / kern_s, "peak_per_s": RANDOM_LINE_PEAK, "frac": traffic / 128.0 / kern_s / RANDOM_LINE_PEAK,
            "note": "peak_per_s = random 128-byte lines/s of independent gathers over tables of 8-250 GiB (tools/ubench_granule.hip): an "
                                                              "index whose hot arrays are a few GB gets part of its lines from the 256 MB Infinity Cache, which FETCH_SIZE counts "
                                                              "too, so such a line can come close to 1"},
    }

COMPACT_LIMIT = 8192  # bytes of the stdout line (the driver's parser lost round 5's 25 KB line)


def compact_record(full, extras_path):
    """The ONE stdout line: the contract's keys, the roofline and CPU-baseline objects cut to their figures, parity, and scalar
    figures of the extra lines.  Everything else -- notes, telemetry, counters, sub-lines -- is in the extras file (and on stderr)."""
    def pick(d, keys):
        return {kk: d[kk] for kk in keys if isinstance(d, dict) and kk in d}

    def short(text, n):
        text = str(text)
        return text if len(text) <= n else text[:n - 3] + "..."

    out = pick(full, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"))
    cfg = full.get("config", {})
    out["config"] = pick(cfg, ("k", "query_kind", "queries_per_step", "queries_per_gpu", "bwt_symbols", "index_bytes", "table_depth", "direct_table_depth", "sparse_table_depth",
                               "sparse_table_tiers", "query_length_hint", "pair_index", "pair_stride", "block_format"))
    out["config"]["workload"] = short(str(cfg.get("workload", "")).split(".  ")[0], 420)
    if "parallelism" in cfg:
        out["config"]["parallelism"] = short(cfg["parallelism"], 200)
    roof = full.get("roofline")
    if isinstance(roof, dict):
        r = pick(roof, ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "kernel_launches", "kernel_stamp"))
        if roof.get("traffic_source") is not None:
            r["traffic_source"] = short(roof["traffic_source"], 160)
        lay, refalg, lines = roof.get("layout_algorithmic") or {}, roof.get("reference_algorithm") or {}, roof.get("random_lines") or {}
        r.update({"layout_bytes_per_query": lay.get("bytes_per_query"), "layout_lines_per_query": lay.get("lines_per_query"),
                  "frac_layout_algorithmic": lay.get("frac_algorithmic"), "traffic_over_layout": lay.get("traffic_over_algorithmic"),
                  "reference_bytes_per_query": refalg.get("bytes_per_query"), "random_lines_frac": lines.get("frac")})
        out["roofline"] = r
    cpu = full.get("cpu_baseline")
    if isinstance(cpu, dict):
        c = pick(cpu, ("value", "unit", "cores", "kind", "nproc", "cpu_model"))
        c["sample"] = short(cpu.get("sample", ""), 200)
        c["all_cores_value"] = (cpu.get("all_cores") or {}).get("value")
        out["cpu_baseline"] = c
    if isinstance(full.get("parity"), dict):
        out["parity"] = pick(full["parity"], ("checked", "mismatches"))
    extras = {}
    for key in ("c5_random_1e9", "c4_repeats", "c4_real_reads", "c4_budgeted", "undeclared_k", "headline_sparse_off", "sorted_batch", "weak_scaling", "native_gather"):
        line = full.get(key)
        if not isinstance(line, dict):
            continue
        if "error" in line and "value" not in line:
            extras[key + "_error"] = short(line["error"], 120)
            continue
        extras[key + "_qps"] = line.get("value")
        if isinstance(line.get("roofline"), dict) and "frac" in line["roofline"]:
            extras[key + "_frac"] = line["roofline"]["frac"]
        if isinstance(line.get("parity"), dict):
            extras[key + "_mismatches"] = line["parity"].get("mismatches")
        for flag in ("counts_equal_headline", "counts_equal_unordered_run", "equals_torch_path"):
            if flag in line:
                extras[key + "_" + flag] = line[flag]
        for num in ("sparse_table_depth", "second_level_depth", "direct_table_depth", "lines_per_query"):
            if num in line:
                extras[key + "_" + num] = line[num]
    if isinstance(full.get("c4_budgeted"), dict):
        for mode, rec in (full["c4_budgeted"].get("modes") or {}).items():
            if isinstance(rec, dict):
                extras["c4_budgeted_%s_qps" % mode] = rec.get("value")
                extras["c4_budgeted_%s_lines_per_query" % mode] = rec.get("lines_per_query")
    if isinstance(full.get("short_k"), dict):
        extras["short_k_worst_ratio_to_sparse_off"] = full["short_k"].get("worst_ratio_to_sparse_off")
    if isinstance(full.get("host_api"), dict):
        for kk in ("bytes_qps", "packed_u64_qps", "packed_u32_qps", "pcie_both_ways_GBps"):
            if kk in full["host_api"]:
                extras["host_api_" + kk] = full["host_api"][kk]
    if isinstance(full.get("ranks"), dict):
        extras.update({"ranks_kernel_ms_min": full["ranks"].get("kernel_ms_min"), "ranks_kernel_ms_max": full["ranks"].get("kernel_ms_max"),
                       "ranks_exchange_ms_alone": full["ranks"].get("exchange_ms_alone")})
    if isinstance(full.get("native_gather"), dict):
        for kk in ("single_batch_latency_ms", "single_batch_pipelined_ms", "single_batch_pipelined_narrow_destination_ms"):
            if kk in full["native_gather"]:
                extras["native_gather_" + kk] = full["native_gather"][kk]
    if full.get("consistency_errors"):
        extras["consistency_errors"] = len(full["consistency_errors"])
    out["extras"] = extras
    out["extras_file"] = os.path.basename(extras_path) if extras_path else None
    line = json.dumps(out)
    if len(line) >= COMPACT_LIMIT:  # (cannot happen with the fields above; the contract's keys always survive)
        out["extras"] = {kk: vv for kk, vv in extras.items() if kk.endswith("_qps")}
        out["config"]["workload"] = short(out["config"]["workload"], 160)
    return out


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    # stdout carries ONE line, the JSON result: everything else that writes to fd 1 (RCCL prints a
    # version banner there, libraries may follow) is sent to stderr
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        """the full record to the extras file and stderr; ONE compact line (compact_record, < 8 KB) to stdout"""
        blob = json.dumps(obj)
        try:
            with open(args.extras_file, "w") as f:
                f.write(blob + "\n")
        except OSError as e:
            log("could not write %s: %r" % (args.extras_file, e))
        log("full record: " + blob)
        line = json.dumps(compact_record(obj, args.extras_file))
        assert len(line) < COMPACT_LIMIT, "compact line of %d bytes" % len(line)
        os.write(result_fd, (line + "\n").encode())

    import gc

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log("warning: WORLD_SIZE=%d but --gpus %d; using WORLD_SIZE" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    multi = world > 1 or args.force_dist  # the exchange step exists
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        import torch.distributed as dist
        # rank 0 may spend a minute building an index file while the others wait at a barrier
        patience = datetime.timedelta(minutes=30)
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=patience)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=patience)

    msbwt = importlib.import_module("rust-msbwt_amd")
    import synth

    # host threads of this rank: the ranks of one node share its cores (generators, oracle)
    nproc = os.cpu_count() or 1
    host_threads = max(1, min(16, nproc // max(1, world)))
    synth.set_threads(host_threads)
    torch.set_num_threads(host_threads)

    human = args.workload == "human"
    big = human or args.workload == "big"

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gpu_telemetry

    card_pci = None
    try:
        pr = torch.cuda.get_device_properties(dev)
        card_pci = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
    except Exception:  # noqa: BLE001
        pass

    def telemetry():
        """clocks / power / temperature / memory of this rank's card, now (tools/gpu_telemetry.py: sysfs) -- what tells a slow box from a regression"""
        snap = gpu_telemetry.snapshot(local_rank, card_pci)
        try:
            free_b, total_b = torch.cuda.mem_get_info(dev)
            snap["hip_free_bytes"], snap["hip_total_bytes"] = int(free_b), int(total_b)
        except Exception:  # noqa: BLE001
            pass
        return snap

    class EarlyIndexes:
        """The C4-sized real MSBWTs of the default run's extra lines are suffix-sorted on the HOST (about a minute each): started
        now, on a thread of their own, they are ready by the time the GPU has finished the human-scale lines."""

        def __init__(self):
            self.done, self.thread = {}, None

        def start(self, names, scale):
            import threading

            def work():
                for nm in names:
                    t_b = time.time()
                    try:
                        self.done[(nm, scale)] = synth.workload_index(nm, scale)
                        log("background: %s index built in %.1fs" % (nm, time.time() - t_b))
                    except Exception as e:  # noqa: BLE001
                        self.done[(nm, scale)] = e
            self.thread = threading.Thread(target=work, daemon=True)
            self.thread.start()

        def get(self, nm, scale):
            while (nm, scale) not in self.done and self.thread is not None and self.thread.is_alive():
                self.thread.join(timeout=5.0)
            got_it = self.done.get((nm, scale))
            if isinstance(got_it, Exception):
                raise got_it
            return got_it if got_it is not None else synth.workload_index(nm, scale)

    early_indexes = EarlyIndexes()
    early_scale = args.c4_scale if args.c4_scale > 0 else (1.0 if args.scale == 1.0 else 0.0)
    if human and world == 1 and not args.force_dist and not args.no_c4 and early_scale > 0 and rank == 0:
        early_indexes.start(["c4r"] + ([] if args.no_c4_random else ["c4"]), early_scale)
    cfg = dict(synth.CONFIGS["c3" if big else args.workload])
    k = args.k or cfg["k"]
    kind = args.query_kind or ("walk" if big else cfg["queries"])
    nq = args.queries or (300_000_000 if human else 20_000_000 if big else cfg["nq"]) or 20_000_000
    if args.scale < 1.0 and not args.queries:
        nq = max(1000, int(nq * min(1.0, args.scale * (4 if not human else 40))))
    symbols = int((HUMAN_SYMBOLS if human else args.big_symbols) * (args.scale if human else 1.0))
    hist_file = synth.HISTOGRAM_FILE if (args.stream == "histogram" and os.path.exists(synth.HISTOGRAM_FILE)) else None
    exact_bwt = big and args.stream == "reads"
    read_len, coverage = 150, 30.0

    bwt = msbwt.RleBWT(device=local_rank)
    bwt.set_block_format(args.blocks)
    if args.table_depth > -2:
        bwt.set_table_depth(args.table_depth)
    if args.sparse_depth > -2:
        bwt.set_sparse_table(args.sparse_depth)
    if args.sparse_tiers > -2:
        bwt.set_sparse_tiers(args.sparse_tiers)
    # the index is built for the k it will be asked about (a deployment knows its k; results never depend on it): the automatic sparse
    # table then goes as deep as min(k, 31) where it fits -- a table of d-mers serves k >= d only
    bwt.set_query_length(k if args.query_length_hint < 0 else args.query_length_hint)
    if args.no_table_side:
        bwt.set_table_side(0)
    t0 = time.time()
    rle = npy = reads = None
    if big:
        if exact_bwt:
            # the exact MSBWT of an error-free read set, built on this GPU from the genome's own suffix order (no suffix
            # sort of the 9e10 read suffixes: synth/bwt_reads.py).  Same seed, same device type => identical replicas.
            from synth import bwt_reads
            genome_len = max(2000, int(symbols * read_len / (coverage * (read_len + 1))))
            if args.genome == "repeats":
                genome, cnt = bwt_reads.repeat_read_set(genome_len, read_len, coverage, 77, device=dev)
                log("rank %d: repeat-bearing genome of %d bases in %.1fs" % (rank, genome_len, time.time() - t0))
                rle, _, n_reads_total = bwt_reads.msbwt_rle_repeats(genome, cnt, read_len, log=lambda m: log("rank %d: bwt: %s" % (rank, m)))
            else:
                genome, cnt = bwt_reads.read_set(genome_len, read_len, coverage, 77, device=dev)
                rle, _, n_reads_total = bwt_reads.msbwt_rle(genome, cnt, read_len, log=lambda m: log("rank %d: bwt: %s" % (rank, m)))
            del genome, cnt
            torch.cuda.empty_cache()
            log("rank %d: exact MSBWT of %d error-free %d-bp reads: %d RLE bytes in %.1fs" % (rank, n_reads_total, read_len, len(rle), time.time() - t0))
        else:
            # structure-equivalent synthetic RLE stream (NOT a real BWT): independent symbols.  Same seed on every rank => identical replicas.
            rle, _ = synth.rle_stream(symbols, args.big_mean_run, 77, histogram=hist_file)
            log("rank %d: synthetic RLE stream of %d bytes in %.1fs (%d host threads)" % (rank, len(rle), time.time() - t0, host_threads))
        t0 = time.time()
        bwt.load_vector(rle)
        if rank != 0:
            rle = None  # only rank 0 feeds the oracle
    else:
        # rank 0 builds (and caches) the index file, everyone loads it
        if rank == 0:
            npy, reads = synth.workload_index(args.workload, args.scale)
        if multi:
            dist.barrier()
        if rank != 0:
            npy, reads = synth.workload_index(args.workload, args.scale)
        log("rank %d: index file %s ready in %.1fs" % (rank, os.path.basename(npy), time.time() - t0))
        t0 = time.time()
        bwt.load_numpy_file(npy)
    total = bwt.get_total_size()
    log("rank %d: %d symbols on the GPU (%.1f MB index, table depth %d, pair index %s stride %d, typical range width %.1f) in %.1fs"
        % (rank, total, bwt.device_bytes() / 1e6, bwt.get_table_depth(), bwt.get_pair_index(), bwt.get_pair_stride(),
           bwt.get_typical_range_width(), time.time() - t0))

    # ---- the batch: identical on every rank (same seeds); d_q holds ALL nq queries in HBM ---------
    fused = args.fused and not big
    t0 = time.time()
    d_reads = None
    nread = rlen = wins = 0
    if fused:
        kind = "reads"
        nread, rlen = reads.shape
        wins = rlen - k + 1
        nq = nread * wins
        d_reads = torch.from_numpy(reads).to(dev)
        d_q = None
    elif kind == "walk":
        d_q = walk_kmers(torch, np, bwt, dev, total, nq, k, 4242)
    elif kind == "random" and (args.device_queries or big):
        d_q = device_random_kmers(torch, dev, 0, nq, k, 4242)
    elif kind == "random":
        d_q = torch.from_numpy(synth.random_kmers(nq, k, cfg["qseed"])).to(dev)
    else:
        d_q = torch.from_numpy(synth.read_kmers(reads, k, limit=nq, seed=cfg["qseed"])).to(dev)
        nq = d_q.shape[0]
    def table_order(q):
        """the permutation that puts the rows of q into the order of their batch-order keys (msbwt_rle_kmer_order_keys_device)"""
        keys = torch.empty(q.shape[0], dtype=torch.int64, device=dev)
        bwt.kmer_order_keys_device(q.data_ptr(), q.shape[1], q.shape[0], keys.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
        if args.sort_bits > 0:  # the key's table index ends at bit 61
            keys = (keys >> (62 - args.sort_bits)) & ((1 << args.sort_bits) - 1)
            return torch.argsort(keys, stable=True)
        keys ^= -(2 ** 63)  # u64 order on int64 storage
        return torch.argsort(keys)

    def gathered(q, order, chunk=50_000_000):
        out = torch.empty_like(q)
        for lo_s in range(0, q.shape[0], chunk):
            out[lo_s:lo_s + chunk] = q[order[lo_s:lo_s + chunk]]
        return out

    if args.sort_queries and d_q is not None:
        d_q = gathered(d_q, table_order(d_q))
        log("rank %d: batch ordered by its order keys" % rank)
    torch.cuda.synchronize(dev)
    log("rank %d: %d %s %d-mers in HBM in %.1fs" % (rank, nq, kind, k, time.time() - t0))
    stream = torch.cuda.current_stream(dev).cuda_stream

    strong = multi and args.scaling == "strong" and not fused

    def fence():
        if multi:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def gather_floats(values):
        """[values of rank 0, values of rank 1, ...] on every rank"""
        if not multi:
            return [list(values)]
        t = torch.tensor(values, dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
        out = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(out, t)
        return [o.tolist() for o in out]

    # ---- the exchange step (N > 1): every rank ends each step holding ALL counts ------------------
    # 8 bytes per query over xGMI can cost more than the search itself, so the counts travel as
    # int16 whenever that is exact (a device-side overflow flag is kept per step and checked after
    # the loop; any overflow re-runs the whole measurement with 64-bit payloads), the all_gather
    # runs asynchronously on RCCL's stream and overlaps the next step's kernel (two buffers in
    # flight), and each rank widens what it received back to u64 -- inside the timed region.
    class Batch:
        """One batch in HBM and how a step counts this rank's part of it."""

        def __init__(self, the_bwt, q=None, row0=0, fused_reads=None, kk=k):
            self.bwt, self.q, self.row0, self.reads, self.k = the_bwt, q, row0, fused_reads, kk

        def count_into(self, out, a, b):
            if self.reads is not None:
                self.bwt.count_read_kmers_device(self.reads.data_ptr(), rlen, nread, self.k, False, out.data_ptr(), 0, stream)
            elif b > a:  # self.q holds the batch's rows from row0 on
                self.bwt.count_kmers_device(self.q.data_ptr() + (a - self.row0) * self.k, self.k, b - a, out.data_ptr(), stream)

    def run_steps(batch, nsteps, narrow, a, b, per_rank, native=None, exchange=True):
        """nsteps passes over queries [a, b) of the batch; N > 1: + all_gather of `per_rank` counts per rank"""
        outs = [torch.zeros(per_rank, dtype=torch.int64, device=dev) for _ in range(2)]
        if os.environ.get("MSBWT_VERBOSE"):  # (with the library's own load lines: where everything a step touches sits)
            log("buffers: queries %#x counts %#x" % ((batch.reads if batch.reads is not None else batch.q).data_ptr(), outs[0].data_ptr()))
        if not multi:
            for _ in range(nsteps):
                batch.count_into(outs[0], a, b)
            return outs[0], None, False
        if native is not None:  # the library's own RCCL call: narrow, ncclAllGather, widen -- all on this stream
            d_all = torch.empty(per_rank * world, dtype=torch.int64, device=dev)
            for i in range(nsteps):
                batch.count_into(outs[i & 1], a, b)
                batch.bwt.allgather_counts(native, outs[i & 1].data_ptr(), per_rank, d_all.data_ptr(), 16 if narrow else 64, stream)
            return outs[(nsteps - 1) & 1], d_all, False  # (an overflow shows up in device_status)
        pay_dtype = torch.int16 if narrow else torch.int64
        to_root = args.exchange == "gather"
        have_all = (not to_root) or rank == 0     # who ends a step holding every rank's counts
        sends = [torch.zeros(per_rank, dtype=pay_dtype, device=dev) for _ in range(2)]
        recvs = [torch.empty(per_rank * world, dtype=pay_dtype, device=dev) if have_all else None for _ in range(2)]
        d_all = torch.empty(per_rank * world, dtype=torch.int64, device=dev) if have_all else None
        overflow = torch.zeros((), dtype=torch.bool, device=dev)
        works = [None, None]

        def finish(j):  # widen what slot j received (the collective is done once wait() returns)
            if works[j] is not None:
                works[j].wait()
                if have_all:
                    d_all.copy_(recvs[j])
                works[j] = None

        for i in range(nsteps):
            j = i & 1
            finish(j)
            if exchange is not None:
                batch.count_into(outs[j], a, b)
            if narrow and outs[j].numel():  # one reduction pass for both ends (u64 counts >= 2^63 look negative)
                mn, mx = torch.aminmax(outs[j])
                overflow |= (mx > NARROW_MAX) | (mn < 0)
            sends[j].copy_(outs[j])
            # the payload crosses as raw bytes: neither NCCL/RCCL nor gloo has a 16-bit integer type
            if args.dist_backend == "nccl" and to_root:
                # the final gather of the counts: rank 0 receives every shard over that rank's own link (grouped send / recv inside RCCL)
                chunks = [recvs[j][r * per_rank:(r + 1) * per_rank].view(torch.uint8) for r in range(world)] if rank == 0 else None
                works[j] = dist.gather(sends[j].view(torch.uint8), gather_list=chunks, dst=0, async_op=True)
            elif args.dist_backend == "nccl":
                works[j] = dist.all_gather_into_tensor(recvs[j].view(torch.uint8), sends[j].view(torch.uint8),
                                                       async_op=True)  # RCCL over xGMI
            elif to_root:  # rehearsal: same logic, collective through host memory
                host_send = sends[j].cpu().view(torch.uint8)
                host_chunks = [torch.empty_like(host_send) for _ in range(world)] if rank == 0 else None
                dist.gather(host_send, gather_list=host_chunks, dst=0)
                if rank == 0:
                    recvs[j].copy_(torch.cat(host_chunks).view(pay_dtype))
                    d_all.copy_(recvs[j])
            else:
                host = torch.empty(per_rank * world, dtype=pay_dtype)
                dist.all_gather_into_tensor(host.view(torch.uint8), sends[j].cpu().view(torch.uint8))
                recvs[j].copy_(host)
                d_all.copy_(recvs[j])
        finish(nsteps & 1)        # the older of the two outstanding steps first
        finish((nsteps - 1) & 1)
        return outs[(nsteps - 1) & 1], d_all, overflow

    def timed(batch, narrow, a, b, per_rank, native=None, exchange=True):
        run_steps(batch, args.warmup, narrow, a, b, per_rank, native, exchange)
        fence()
        batch.bwt.device_status(stream)
        batch.bwt.set_kernel_timing(True)
        t_start = time.perf_counter()
        d_mine, d_everything, ovf = run_steps(batch, args.steps, narrow, a, b, per_rank, native, exchange)
        fence()
        dt = time.perf_counter() - t_start
        batch.bwt.set_kernel_timing(False)
        k_ms, n_launch = batch.bwt.kernel_time_ms()
        overflowed = bool(ovf) if native is None else False
        try:
            batch.bwt.device_status(stream)
        except msbwt.MsbwtError as e:
            if native is None or e.code != msbwt._lib.ERR_OVERFLOW:
                raise
            overflowed = True
        if multi:
            flag = torch.tensor([1.0 if overflowed else 0.0, dt], dtype=torch.float64,
                                device=dev if args.dist_backend == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            return d_mine, d_everything, float(flag[1].item()), k_ms, n_launch, flag[0].item() > 0
        return d_mine, d_everything, dt, k_ms, n_launch, False

    last_sampled = {}

    def measure(batch, a, b, per_rank, native=None, exchange=True):
        narrow = multi and args.payload == "auto"
        with gpu_telemetry.Sampler(local_rank, card_pci) as sampler:   # what the card did DURING these steps (warm-up included)
            res = timed(batch, narrow, a, b, per_rank, native, exchange)
        last_sampled.clear()
        last_sampled.update(sampler.summary())
        if res[5]:  # some count did not fit int16: measure again with u64 payloads (always exact)
            log("counts exceed int16: re-running with 64-bit payloads")
            narrow = False
            res = timed(batch, False, a, b, per_rank, native, exchange)
        return res[:5] + (narrow,)

    def counted_pass(the_bwt, batch, a, b, per_rank):
        """what one launch does to the index, per query: one untimed pass with the library's search counters on"""
        the_bwt.search_counters(stream)
        the_bwt.set_search_counters(True)
        run_steps(batch, 1, False, a, b, per_rank)
        c = the_bwt.search_counters(stream)
        the_bwt.set_search_counters(False)
        nqs = max(1, b - a)
        info = the_bwt.table_info()
        return {"queries": b - a, "raw": c,
                # (sparse-table lookups are search steps of their own: their bucket lines are among first_lines already)
                "lines_per_query": (c["first_lines"] + c["second_lines"]) / nqs + (0.0 if c.get("table_steps") else (1.0 if the_bwt.get_table_depth() else 0.0)),
                "table_lines_per_query": c.get("table_steps", 0) / nqs, "sparse_table": the_bwt.sparse_table_info(),
                "search_lines_per_query": (c["first_lines"] + c["second_lines"]) / nqs,
                "steps_per_searched_query": c["lane_steps"] / max(1, c["searched"]),
                "second_line_rate": c["second_lines"] / max(1, c["lane_steps"]),
                "sat_out_per_step": c["sat_out"] / max(1, c["lane_steps"]),
                "lanes_busy_per_wave_step": c["lane_steps"] / max(1, c["wave_steps"]),
                "escape_query_fraction": c["escape_queries"] / nqs, "escape_restart_fraction": c["escape_restarts"] / nqs,
                "decided_by_table_fraction": c["table_decided"] / nqs,
                "table": info, "escape_line_fraction": info["escape_lines"] / max(1, info["lines"]),
                "note": "lines_per_query = lines the search asked for (first + second bound lines, side-array lines included) + one table line; "
                        "counted by the kernel itself in an extra pass (msbwt_rle_set_search_counters)"}

    def shard(n_all):
        if strong:
            a, b = msbwt.sharded.shard_bounds(n_all, world, rank)  # 16-query aligned: every shard keeps the fast kernels
            return a, b, msbwt.sharded.shard_capacity(n_all, world)
        return 0, n_all, n_all

    def stitch(d_all, d_out, n_all, cap):
        """the whole batch's counts as one vector (strong: stitched from the gathered shards)"""
        if not strong:
            return d_out
        if d_all is None:   # (gather to the root: only rank 0 holds the whole batch's counts)
            return None
        spans = [msbwt.sharded.shard_bounds(n_all, world, r) for r in range(world)]
        return torch.cat([d_all[r * cap:r * cap + (b - a)] for r, (a, b) in enumerate(spans)])

    main_batch = Batch(bwt, d_q, 0, d_reads)
    lo, hi, cap = shard(nq)
    mine_n = hi - lo
    telemetry_before = telemetry()
    def copy_number_bins(the_bwt, d_batch, d_batch_counts, kk, ranges, want_b=10_000_000):
        """What a k-mer costs by COPY NUMBER: sub-batches of a line's own queries picked by their count in the read set, each timed
        (5 steps) and counted by the kernel's own counters.  -> (record, counts equal the main run's)"""
        bins, ok = [], True
        for lo_c, hi_c in ranges:
            ids = torch.nonzero((d_batch_counts >= lo_c) & (d_batch_counts < hi_c)).flatten()
            have_b = int(ids.numel())
            head = {"count_from": lo_c, "count_below": None if hi_c > 10**12 else hi_c, "queries_in_batch": have_b}
            if have_b < 1000:
                bins.append(head)
                continue
            pick = ids[torch.randint(0, have_b, (want_b,), device=dev, generator=torch.Generator(device=dev).manual_seed(lo_c))] if have_b < want_b else ids[:want_b]
            d_qb = d_batch[pick].contiguous()
            bb = Batch(the_bwt, d_qb, 0, None, kk)
            saved = args.steps
            args.steps = min(args.steps, 5)
            try:
                ob, _, elb, kmsb, _, _ = measure(bb, 0, want_b, want_b)
                cb = counted_pass(the_bwt, bb, 0, want_b, want_b)
                same = bool(torch.equal(ob, d_batch_counts[pick]))
                bins.append(dict(head, queries_timed=want_b, value=want_b * args.steps / elb, ms_per_step=elb / args.steps * 1e3, kernel_ms=kmsb,
                                 lines_per_query=cb["lines_per_query"], second_line_rate=cb["second_line_rate"], escape_query_fraction=cb["escape_query_fraction"],
                                 steps_per_searched_query=cb.get("steps_per_searched_query"), mean_count=float(ob.float().mean().item()), counts_equal_main_run=same))
                ok = ok and same
            finally:
                args.steps = saved
            del d_qb, ob, pick, ids
        return {"bins": bins, "note": "sub-batches of the line's own queries by their count in the read set (10^7 queries each, sampled with repetition "
                                      "where the batch holds fewer), each timed over 5 steps and counted by the kernel's own counters; their counts are "
                                      "those of the main run, whose parity sample the oracle checks"}, ok

    d_out, d_all, elapsed, kernel_ms, launches, narrow = measure(main_batch, lo, hi, cap)
    telemetry_after, telemetry_during = telemetry(), dict(last_sampled)
    # cross-rank consistency: a failed check does not abort the run (a crashed rank leaves no record at all); it is
    # reported in the JSON line and voids `value`
    inconsistent = []
    if multi and d_all is not None and not torch.equal(d_all[rank * cap:rank * cap + mine_n], d_out[:mine_n]):  # the gathered vector must contain this rank's own counts where they belong
        inconsistent.append("rank %d: gathered counts differ from the local ones" % rank)
    if multi and args.exchange == "gather":
        # gather to the root: every rank reports the sum and the length of its shard's counts (exact in float64: below 2^53), the root compares
        # them with what it received from that rank
        reported = gather_floats([float(d_out[:mine_n].sum().item()), float(mine_n)])
        if rank == 0:
            for r, (shard_sum, shard_len) in enumerate(reported):
                a_r, b_r = msbwt.sharded.shard_bounds(nq, world, r) if strong else (0, nq)
                got_sum = float(d_all[r * cap:r * cap + (b_r - a_r)].sum().item())
                if int(shard_len) != b_r - a_r or got_sum != shard_sum:
                    inconsistent.append("root: the counts received from rank %d sum to %.0f over %d queries, that rank reports %.0f over %d" % (r, got_sum, b_r - a_r, shard_sum, int(shard_len)))
    ms_per_step = elapsed / args.steps * 1e3
    counters = None
    if args.counters and not multi:
        counters = counted_pass(bwt, main_batch, lo, hi, cap)
    job_queries = nq if (strong or not multi) else nq * world
    value = job_queries * args.steps / elapsed
    # what the index looked like when the headline ran (the variant lines further down rebuild its tables)
    headline_shape = index_shape(bwt, k, fused, mine_n if strong else nq)
    d_counts = stitch(d_all, d_out, nq, cap)
    # the repeat-genome lab line: what a k-mer costs by copy number AT THIS SIZE (human copy numbers: 10^5 and more occurrences)
    main_bins = None
    if exact_bwt and args.genome == "repeats" and not multi and d_q is not None and not args.no_variants:
        main_bins, main_bins_ok = copy_number_bins(bwt, d_q, d_counts, k, ((1, 100), (100, 1000), (1000, 10_000), (10_000, 100_000), (100_000, 1_000_000), (1_000_000, 1 << 62)))

    # per-rank view of the timed region: the count kernel alone (HIP events on the launch stream), and the exchange
    # step alone (the same all_gather + widening, timed without kernels) -- so that a sub-linear curve can be read
    ranks_info = None
    if multi:
        _, _, ex_elapsed, _, _, _ = measure(main_batch, lo, hi, cap, exchange=None)
        per = gather_floats([kernel_ms, ex_elapsed / args.steps * 1e3])
        ranks_info = {"kernel_ms": [p[0] for p in per], "kernel_ms_min": min(p[0] for p in per), "kernel_ms_max": max(p[0] for p in per),
                      "exchange_ms_alone": max(p[1] for p in per), "step_ms": ms_per_step,
                      "step_minus_slowest_kernel_ms": ms_per_step - max(p[0] for p in per),
                      "note": "kernel_ms: average count-kernel duration per rank; exchange_ms_alone: narrow + gather (or all_gather) + widen of one step "
                              "with no kernel running (max over ranks); inside a step the exchange overlaps the next step's kernel"}

    # the same batch handed over in the order of its batch-order keys (include/msbwt_hip.h, "batch order"): what a caller
    # that holds a sorted k-mer list, or counts a batch more than once, gets.  Counts must equal the unordered run's.
    sorted_batch = None
    if human and not multi and not args.sort_queries and not args.no_sorted and d_q is not None and rank == 0:
        order = table_order(d_q)
        d_sorted = gathered(d_q, order)
        saved_steps, args.steps = args.steps, min(args.steps, 10)
        s_out, _, s_elapsed, s_kms, _, _ = measure(Batch(bwt, d_sorted, 0), 0, nq, nq)
        same = bool(torch.equal(s_out, d_counts[order]))
        sorted_batch = {"value": nq * args.steps / s_elapsed, "unit": "queries/s", "ms_per_step": s_elapsed / args.steps * 1e3, "kernel_ms": s_kms,
                        "steps": args.steps, "counts_equal_unordered_run": same,
                        "note": "the same %d queries ordered by msbwt_rle_kmer_order_keys_device (sorting is outside the timed region: the caller's)" % nq}
        args.steps = saved_steps
        if not same:
            log("PARITY FAILURE: the ordered batch's counts differ from the unordered run's")
        del order, d_sorted, s_out
        torch.cuda.empty_cache()

    weak = None
    if strong and not args.no_weak:
        # every rank the same (whole or 1e8-query) batch, all N x n counts gathered each step; bounded so that
        # the N x n gathered vectors fit next to a 200 GB index at N = 8
        nw = min(nq, 100_000_000)
        was_strong, strong = strong, False
        w_out, w_all, w_elapsed, _, _, w_narrow = measure(main_batch, 0, nw, nw)
        strong = was_strong
        if w_all is not None:   # every rank counted the SAME queries: each received block must equal this rank's own counts
            for r in range(world):
                if not torch.equal(w_all[r * nw:(r + 1) * nw], w_out):
                    inconsistent.append("rank %d: the weak-scaling counts of rank %d differ from the local ones" % (rank, r))
        if d_counts is not None and not torch.equal(w_out, d_counts[:nw]):  # (also what would show if the replicas of the index differed between ranks)
            inconsistent.append("rank %d: sharded counts differ from this GPU's own counts of the same queries" % rank)
        weak = {"value": nw * world * args.steps / w_elapsed, "unit": "queries/s", "ms_per_step": w_elapsed / args.steps * 1e3,
                "queries_per_gpu": nw, "payload": "int16" if w_narrow else "int64",
                "note": "every rank runs the same %d queries and all N x n counts are all_gathered each step" % nw}
        del w_out, w_all

    # the same strong-scaling step with the LIBRARY's own RCCL call site (msbwt_rle_allgather_counts over a
    # communicator made through msbwt_comm_*; the id travels through torch.distributed) -- what a Rust host with one
    # process per GPU would run.  Extra, fenced: a failure here never costs the main line.
    native = None
    if multi and strong and args.dist_backend == "nccl" and not args.no_native_gather:
        try:
            id_t = torch.zeros(msbwt._lib.COMM_ID_BYTES, dtype=torch.uint8, device=dev)
            if rank == 0:
                id_t.copy_(torch.frombuffer(bytearray(msbwt.RankComm.unique_id()), dtype=torch.uint8))
            dist.broadcast(id_t, 0)
            # the communicator is set up on a side thread with a deadline: a bootstrap that never returns must cost this
            # extra, not the run (every rank applies the same rule, and all must have succeeded for the step to be taken)
            import threading
            box = {}

            def make_comm():
                try:
                    torch.cuda.set_device(local_rank)
                    box["comm"] = msbwt.RankComm(world, bytes(id_t.cpu().numpy().tobytes()), rank)
                except Exception as e:  # noqa: BLE001
                    box["error"] = repr(e)

            th = threading.Thread(target=make_comm, daemon=True)
            th.start()
            th.join(timeout=120.0)
            ok_t = torch.tensor([1.0 if "comm" in box else 0.0], dtype=torch.float64, device=dev)
            dist.all_reduce(ok_t, op=dist.ReduceOp.MIN)
            if ok_t.item() < 1.0:
                raise RuntimeError("communicator set-up failed or timed out on some rank: %s" % box.get("error", "no answer within 120 s"))
            comm = box["comm"]
            n_out, n_all, n_elapsed, n_kms, _, n_narrow = measure(main_batch, lo, hi, cap, native=comm)
            # (the library's own call is an all-gather: every rank holds n_all; with --exchange gather only the root holds the torch path's vector)
            same = bool(torch.equal(n_all, d_all)) if d_all is not None else bool(torch.equal(n_all[rank * cap:rank * cap + mine_n], d_out[:mine_n]))
            ref_all = d_all if d_all is not None else n_all.clone()
            native = {"value": nq * args.steps / n_elapsed, "unit": "queries/s", "ms_per_step": n_elapsed / args.steps * 1e3,
                      "kernel_ms": n_kms, "payload": "uint16" if n_narrow else "uint64", "equals_torch_path": same,
                      "note": "msbwt_rle_allgather_counts (ncclAllGather bound at run time inside libmsbwt_hip.so), in stream order "
                              "after the kernel -- no overlap with the next step"}
            # What a caller with ONE batch sees (the timed steps above hide each gather behind the next step's kernel): kernel + gather +
            # widening one after the other, against the library's pipelined form -- the shard counted in 4 pieces while the finished
            # pieces' counts travel on a second stream (msbwt_rle_count_kmers_allgather_device) -- and that with the counts left narrow
            # at the destination.  Best of 3, max over the ranks; the main batch's fused-reads form has no shard pointer: matrix only.
            if hi - lo == cap and cap > 0 and main_batch.q is not None:
                try:
                    wire = 16 if n_narrow else 64
                    qptr = main_batch.q.data_ptr() + (lo - main_batch.row0) * k
                    d_m = torch.zeros(cap, dtype=torch.int64, device=dev)
                    d_a = torch.empty(cap * world, dtype=torch.int64, device=dev)
                    d_n = torch.empty(cap * world, dtype=torch.int16 if n_narrow else torch.int64, device=dev)

                    def latency(fn):
                        fn()
                        best = None
                        for _ in range(3):
                            fence()
                            t0 = time.perf_counter()
                            fn()
                            fence()
                            dt1 = time.perf_counter() - t0
                            best = dt1 if best is None else min(best, dt1)
                        t = torch.tensor([best], dtype=torch.float64, device=dev)
                        dist.all_reduce(t, op=dist.ReduceOp.MAX)
                        return t.item() * 1e3

                    def serial():
                        bwt.count_kmers_device(qptr, k, cap, d_m.data_ptr(), stream)
                        bwt.allgather_counts(comm, d_m.data_ptr(), cap, d_a.data_ptr(), wire, stream)

                    native["single_batch_latency_ms"] = latency(serial)
                    native["single_batch_pipelined_ms"] = latency(lambda: bwt.count_kmers_allgather_device(comm, qptr, k, cap, d_m.data_ptr(), d_a.data_ptr(), wire, 64, 4, stream))
                    native["single_batch_pipelined_equals_torch_path"] = bool(torch.equal(d_a, ref_all))
                    native["single_batch_pipelined_narrow_destination_ms"] = latency(lambda: bwt.count_kmers_allgather_device(comm, qptr, k, cap, d_m.data_ptr(), d_n.data_ptr(), wire, wire, 4, stream))
                    native["single_batch_narrow_destination_equals_torch_path"] = bool(torch.equal(d_n.to(torch.int64), ref_all))
                    bwt.device_status(stream)
                    native["single_batch_note"] = ("one batch, not a stream of them: kernel, all-gather and widening in stream order (single_batch_latency_ms) against the "
                                                   "shard counted in 4 pieces while the finished pieces' counts travel on a second stream "
                                                   "(msbwt_rle_count_kmers_allgather_device), and the same with the counts left %d bits wide at the destination" % wire)
                    del d_m, d_a, d_n
                except Exception as e:  # noqa: BLE001
                    native["single_batch_error"] = repr(e)
            comm.close()
            del n_out, n_all
        except Exception as e:  # noqa: BLE001
            native = {"error": repr(e)}
            log("native gather failed: %r" % (e,))

    kind_text = {"walk": "present (LF-walk)", "random": "random", "reads": "read-derived"}[kind]
    if exact_bwt and args.genome == "repeats":
        wl = ("%s: EXACT multi-string BWT of %d error-free synthetic %d-bp reads (%.0fx of a REPEAT-BEARING %d-bp genome: synth.repeat_genome, the human "
              "repeat classes at their own proportions and copy numbers -- SINE-, LINE-, LTR-, DNA-element-like families, segmental duplications, satellite "
              "arrays, microsatellites; built on the GPU from the genome's suffix order to the read length, synth/bwt_reads.py msbwt_rle_repeats), "
              "%d symbols; %d %s %d-mers per step (drawn like read windows: a k-mer's chance follows its number of occurrences).  A lab line beside the "
              "metric's repeat-free one; forward strand, no read errors"
              % (args.workload, n_reads_total, read_len, coverage, genome_len, total, nq, kind_text, k))
    elif exact_bwt:
        wl = ("%s: EXACT multi-string BWT of %d error-free synthetic %d-bp reads (%.0fx of a random %d-bp genome; built on the GPU from the genome's "
              "suffix order, synth/bwt_reads.py), %d symbols; %d %s %d-mers per step.  The genome is REPEAT-FREE and the reads carry no errors: what "
              "repeat families and read errors do to this path is in `c4_repeats` (a real MSBWT of reads with errors; cost by copy number: its "
              "`copy_number_bins`) and, at THIS size with human copy numbers, in the lab line `bench.py --genome repeats` "
              "(profiles/r05_lab/human_repeats.json: 7.12e9 q/s at the same 0.72 of the HBM peak)"
              % (args.workload, n_reads_total, read_len, coverage, genome_len, total, nq, kind_text, k))
    elif big:
        wl = ("%s: structure-equivalent synthetic RLE stream (NOT a real BWT; 30x-human-scale stand-in), %d symbols, run lengths %s; "
              "%d %s %d-mers per step" % (args.workload, total, "drawn from the run-length histogram of config C4's real MSBWT (synth/c4_run_histogram.json)"
                                          if hist_file else "geometric, mean %.1f" % args.big_mean_run, nq, kind_text, k))
    else:
        wl = ("%s: %d synthetic %d-bp reads (%.0fx of a %d-bp %s genome, %.1f%% subst.) -> MSBWT of %d symbols; %d %s %d-mers per step"
              % (args.workload, len(reads), reads.shape[1], len(reads) * reads.shape[1] / max(1, int(cfg["genome"] * args.scale)),
                 int(cfg["genome"] * args.scale), "repeat-bearing (synth.REPEAT_FAMILIES)" if cfg.get("repeats") else "random", cfg["err"] * 100, total, nq,
                 "read-derived (ALL windows of ALL reads, prepared in-kernel)" if fused else kind_text, k))
    result = {
        "metric": "k-mer count queries/sec (whole node)",
        "value": value,
        "unit": "queries/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "strong" if (strong or not multi) else "weak",
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
        "config": {
            "workload": wl, "k": k, "query_kind": kind, "queries_per_step": job_queries, "queries_per_gpu": mine_n if strong else nq,
            "bwt_symbols": total, "index_bytes": bwt.device_bytes(),
            "table_depth": lookup_depth(bwt, k), "direct_table_depth": bwt.get_table_depth(), "sparse_table_depth": bwt.get_sparse_table(),
            "sparse_table_tiers": 2 if bwt.get_sparse_tiers() else (1 if bwt.get_sparse_table() else 0),
            "sparse_table": {kk: vv for kk, vv in bwt.sparse_table_info().items() if kk != "wide"}, "query_length_hint": bwt.get_query_length(),
            "pair_index": bwt.get_pair_index(), "pair_stride": bwt.get_pair_stride(),
            "typical_range_width": bwt.get_typical_range_width(), "block_format": bwt.get_block_format(),
            "parallelism": ("index replicated x%d; %s; per step one %s %s of all counts (%s payload, widened to u64 on "
                            "arrival), overlapped with the next step's kernel"
                            % (world, "ONE fixed batch sharded over the ranks" if strong else "every rank its own whole batch",
                               "RCCL" if args.dist_backend == "nccl" else "gloo (rehearsal)",
                               "gather to rank 0" if args.exchange == "gather" else "all_gather",
                               "int16" if narrow else "int64")) if multi else "1 GPU",
        },
    }
    if weak is not None:
        result["weak_scaling"] = weak
    if ranks_info is not None:
        result["ranks"] = ranks_info
    if native is not None:
        result["native_gather"] = native
    if multi:  # any rank's failed consistency check reaches rank 0
        bad = torch.tensor([float(len(inconsistent))], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(bad, op=dist.ReduceOp.SUM)
        if bad.item() > 0:
            result["value"] = None
            result["consistency_errors"] = inconsistent or ["%d check(s) failed on other ranks (see their stderr)" % int(bad.item())]
            for msg in inconsistent:
                log("CONSISTENCY FAILURE: " + msg)
    result["telemetry"] = {"during_timed_region": telemetry_during, "after_timed_region": telemetry_after, "before_timed_region": telemetry_before,
                           "note": "this card's clocks / power / temperatures from sysfs (amdgpu hwmon): sampled every 20 ms while the warm-up and timed steps "
                                   "of the main line ran, and snapshots right before and after; the extra lines carry their own"}
    if counters is not None:
        result["search_counters"] = counters
    if main_bins is not None:
        result["copy_number_bins"] = main_bins
        if not main_bins_ok:
            log("PARITY FAILURE: a copy-number sub-batch counts differently from the main run")
            result["value"] = None
    if sorted_batch is not None:
        result["sorted_batch"] = sorted_batch
        if not sorted_batch["counts_equal_unordered_run"]:
            result["value"] = None

    # the rows the oracle will check travel to the host now; then the batch leaves HBM (the extra lines
    # below need up to 39 GB next to a 200 GB index)
    queries = got = None
    if rank == 0 and not args.no_oracle:
        rng = np.random.default_rng(5)
        want = min(nq, max(args.parity_sample, args.stats_sample or nq, 0 if args.no_cpu_baseline else args.cpu_sample))
        sample_ids = np.sort(rng.choice(nq, size=want, replace=False)) if want < nq else np.arange(nq)
        if fused:
            queries = np.ascontiguousarray(reads[(sample_ids // wins)[:, None], (sample_ids % wins)[:, None] + np.arange(k)[None, :]])
        else:
            queries = d_q[torch.from_numpy(sample_ids).to(dev)].cpu().numpy()
        got = d_counts[torch.from_numpy(sample_ids).to(dev)].cpu().numpy().astype(np.uint64)
    # the variant lines (same index, tables rebuilt) need the batch and its counts again after the 1e9-query line: parked on the host meanwhile
    want_variants = human and not multi and not args.no_variants and not fused and d_q is not None and rank == 0 and not args.sort_queries
    h_q = h_counts = None
    if want_variants:
        h_q, h_counts = d_q.cpu(), d_counts.cpu()
    del d_counts, d_out, d_all, d_q, main_batch
    torch.cuda.empty_cache()

    # ---- BASELINE configs[4] in its literal shape: 1e9 random 31-mers generated in HBM, sharded over the ranks ----
    c5 = None
    if human and not args.no_c5 and (args.scale == 1.0 or args.c5_queries > 0):
        t0 = time.time()
        n5 = args.c5_queries or 1_000_000_000
        lo5, hi5, cap5 = shard(n5)
        d_q5 = device_random_kmers(torch, dev, lo5, hi5, k, 99)
        b5 = Batch(bwt, d_q5, lo5)
        saved_steps, saved_warm = args.steps, args.warmup
        args.steps, args.warmup = 3, 1
        o5, all5, el5, kms5, _, narrow5 = measure(b5, lo5, hi5, cap5)
        args.steps, args.warmup = saved_steps, saved_warm
        dt5 = el5 / 3
        c5 = {"queries": n5, "queries_per_gpu": hi5 - lo5, "ms_per_pass": dt5 * 1e3, "kernel_ms": kms5, "value": n5 / dt5, "unit": "queries/s",
              "kind": "uniform random ACGT 31-mers generated in HBM (BASELINE configs[4]'s query shape)%s"
                      % ("; sharded over %d ranks, all counts all_gathered (%s payload) each pass" % (world, "int16" if narrow5 else "int64") if multi else ", one GPU")}
        if rank == 0:  # parity on a sample of THIS rank's rows (rank 0 holds the first shard)
            ids5 = torch.from_numpy(np.sort(np.random.default_rng(5).choice(hi5 - lo5, size=min(hi5 - lo5, args.parity_sample), replace=False))).to(dev)
            c5["_q"] = d_q5[ids5].cpu().numpy()
            c5["_got"] = o5[ids5].cpu().numpy().astype(np.uint64)
        del d_q5, o5, all5, b5
        torch.cuda.empty_cache()
        log("c5 line: %d random %d-mers in %.1f ms per pass (%.1fs incl. generation)" % (n5, k, dt5 * 1e3, time.time() - t0))

    # ---- the same index with its tables rebuilt: what the headline owes to the declared k and to the sparse table --------------------
    #   undeclared_k         msbwt_rle_set_query_length(0): the automatic sparse table stops at depth 23 (serves every k >= 23);
    #   headline_sparse_off  msbwt_rle_set_sparse_table(0): no sparse table, the deep direct table of round 4 -- what a read set whose error
    #                        k-mers outgrow HBM gets when not even the two-tier table fits;
    #   short_k              present k-mers of k = 17, 19, 21 (shorter than the sparse table's entries: the direct table serves them, and it is
    #                        shallower beside a sparse table) on both, the last k symbols of the headline's own queries.
    short_k_samples = {}
    if want_variants:
        t0 = time.time()
        # (the tables are rebuilt BEFORE the batch returns to HBM: the loader leaves an eighth of the device free for the caller's batches, and
        # counted on top of a batch that is already resident that reserve is what the second sparse level would have fitted in)
        undeclared_error = None
        try:
            if bwt.get_query_length() != 0:
                bwt.set_query_length(0)
        except msbwt.MsbwtError as e:
            undeclared_error = e
        d_q, d_counts = h_q.to(dev), h_counts.to(dev)
        del h_q, h_counts

        def variant_line():
            vb = Batch(bwt, d_q, 0)
            saved = args.steps
            args.steps = min(args.steps, 10)
            try:
                o, _, el, kms, _, _ = measure(vb, 0, nq, nq)
                cp = counted_pass(bwt, vb, 0, nq, nq)
                line = {"value": nq * args.steps / el, "unit": "queries/s", "ms_per_step": el / args.steps * 1e3, "kernel_ms": kms, "steps": args.steps,
                        "counts_equal_headline": bool(torch.equal(o, d_counts)), "sparse_table_depth": bwt.get_sparse_table(),
                        "direct_table_depth": bwt.get_table_depth(), "index_bytes": bwt.device_bytes(), "kernel": kernel_label(bwt, k, False),
                        "lines_per_query": cp["lines_per_query"], "second_line_rate": cp["second_line_rate"]}
            finally:
                args.steps = saved
            del o
            return line

        def short_k_lines(which, into):
            ns = min(nq, 100_000_000)
            saved = args.steps
            args.steps = min(args.steps, 5)
            try:
                for kk in (17, 19, 21):
                    if kk >= k:
                        continue
                    d_s = d_q[:ns, k - kk:].contiguous()
                    o, _, el, kms, _, _ = measure(Batch(bwt, d_s, 0, None, kk), 0, ns, ns)
                    rec = into.setdefault("k%d" % kk, {"queries": ns})
                    rec[which + "_qps"] = ns * args.steps / el
                    rec[which + "_kernel_ms"] = kms
                    rec[which + "_kernel"] = kernel_label(bwt, kk, False)
                    if kk not in short_k_samples:  # the oracle checks a sample of the first variant's counts; the second must equal the first everywhere
                        ids = torch.from_numpy(np.sort(np.random.default_rng(kk).choice(ns, size=min(ns, 100_000), replace=False))).to(dev)
                        short_k_samples[kk] = (d_s[ids].cpu().numpy(), o[ids].cpu().numpy().astype(np.uint64), o)
                    else:
                        rec["counts_equal"] = bool(torch.equal(o, short_k_samples[kk][2]))
                        short_k_samples[kk] = short_k_samples[kk][:2]
                    del d_s
            finally:
                args.steps = saved

        short_k = {}
        try:
            if undeclared_error is not None:
                raise undeclared_error
            result["undeclared_k"] = dict(variant_line(), second_level_depth=bwt.sparse_table_info()["second_depth"],
                                          note="the same index after msbwt_rle_set_query_length(0): the automatic sparse table stops at depth 23 (+ a second level of "
                                               "17-symbol suffixes for 17 <= k < 23 where it fits)")
            short_k_lines("default", short_k)
            bwt.set_sparse_table(0)
            result["headline_sparse_off"] = dict(variant_line(), note="the same index after msbwt_rle_set_sparse_table(0): no sparse table, the deep direct table")
            short_k_lines("sparse_off", short_k)
            ratios = [rec["default_qps"] / rec["sparse_off_qps"] for rec in short_k.values() if "default_qps" in rec and "sparse_off_qps" in rec]
            short_k["worst_ratio_to_sparse_off"] = min(ratios) if ratios else None
            short_k["note"] = ("present k-mers shorter than the sparse table's entries (the last k symbols of the headline's queries, 1e8 of them): the default index "
                               "(sparse table + shallow direct table) against the same index without the sparse table (deep direct table)")
            result["short_k"] = short_k
            for key in ("undeclared_k", "headline_sparse_off"):
                if not result[key]["counts_equal_headline"]:
                    log("PARITY FAILURE: %s counts differ from the headline's" % key)
                    result["value"] = None
            if any(rec.get("counts_equal") is False for rec in short_k.values() if isinstance(rec, dict)):
                log("PARITY FAILURE: short-k counts differ between the two indexes")
                result["value"] = None
        except msbwt.MsbwtError as e:  # (a variant that cannot be built -- no room for the deep direct table -- is recorded, not fatal)
            result.setdefault("undeclared_k", {"error": repr(e)})
            result.setdefault("headline_sparse_off", {"error": repr(e)})
        short_k_samples = {kk: vv[:2] for kk, vv in short_k_samples.items()}
        del d_q, d_counts
        torch.cuda.empty_cache()
        log("variant lines (undeclared k, sparse table off, short k) in %.1fs" % (time.time() - t0))

    rc = 0
    ncpu = host_threads
    ref = orc = None
    if rank == 0 and args.no_oracle:
        result["roofline"] = {"kernel_ms": kernel_ms, "kernel_launches": launches, "note": "--no-oracle: lean line of a profiling pass"}
        if c5 is not None:
            c5.pop("_q", None), c5.pop("_got", None)
            result["c5_random_1e9"] = c5
    elif rank == 0:
        # ---- parity + algorithmic bytes (the oracle is the checker, never the thing timed as `value`) ----
        from oracle import oracle as orc
        t0 = time.time()
        ref = orc.OracleRleBWT(8)
        if big:
            ref.load_vector(rle)
        else:
            ref.load_numpy_file(npy)
        log("oracle loaded in %.1fs" % (time.time() - t0))
        ns = min(len(queries), args.parity_sample)
        sel = np.linspace(0, len(queries) - 1, ns).astype(np.int64)
        exp = ref.count_kmers(queries[sel], nthreads=ncpu)
        mism = int((exp != got[sel]).sum())
        result["parity"] = {"checked": int(ns), "mismatches": mism, "vs": "CPU oracle (restated RleBWT::count_kmer)",
                            "nonzero_counts_in_sample": int((got > 0).sum())}
        if c5 is not None:
            exp5 = ref.count_kmers(c5["_q"], nthreads=ncpu)
            m5 = int((exp5 != c5["_got"]).sum())
            c5["parity"] = {"checked": int(len(exp5)), "mismatches": m5}
            mism += m5
            del c5["_q"], c5["_got"]
            result["c5_random_1e9"] = c5
        if short_k_samples and "short_k" in result:
            checked = bad = 0
            for kk, (qs_k, got_k) in short_k_samples.items():
                exp_k = ref.count_kmers(qs_k, nthreads=ncpu)
                checked += len(exp_k)
                bad += int((exp_k != got_k).sum())
            result["short_k"]["parity"] = {"checked": checked, "mismatches": bad}
            mism += bad
        if mism:
            log("PARITY FAILURE: %d sampled counts differ from the oracle" % mism)
            result["value"] = None
            rc = 1
        per_launch_q = mine_n if strong else nq
        kern_s = kernel_ms / 1e3 if launches else elapsed / args.steps
        tq, tsrc, tnote, stamp = lookup_traffic(args.workload + ("+repeats" if args.genome == "repeats" else ""), k, headline_shape, kind, total, fused, args.scale == 1.0)
        result["roofline"] = roofline_block(orc, ref, queries, k, ncpu, per_launch_q, kern_s, kernel_ms, launches, tq, tsrc, tnote, stamp,
                                            headline_shape["kernel"], args.stats_sample, headline_shape["lookup_depth"], headline_shape["pair_index"],
                                            headline_shape["ordered"], sparse=headline_shape["lookup_depth"] == headline_shape["sparse_depth"] != 0)
        if world == 1 and not args.no_cpu_baseline:
            ncs = min(len(queries), args.cpu_sample)
            t0 = time.time()
            ref.count_kmers(queries[:ncs], nthreads=1)
            t1 = time.time() - t0
            t0 = time.time()
            ref.count_kmers(queries, nthreads=nproc)   # no byte counters: the plain loop on every hardware thread
            t_all = time.time() - t0
            result["cpu_baseline"] = {
                "value": ncs / t1, "unit": "queries/s", "cores": 1, "kind": "port",
                "cpu_model": cpu_model(), "nproc": nproc,
                "sample": "%d queries sampled from the same batch, same RLE stream / comp_msbwt.npy, 1 thread (the reference is single-threaded), -O3 C restatement" % ncs,
                "all_cores": {"value": len(queries) / t_all, "cores": nproc,
                              "note": "%d sampled queries, static partition over all %d hardware threads of this box's share, same un-instrumented build" % (len(queries), nproc)},
            }

    # ---- REAL 30x MSBWTs at config C4's size (12.9 M reads WITH substitutions, 1.95e9 symbols, suffix-sorted on the host while
    # the GPU was busy with the lines above), read-derived 31-mers, one GPU:
    #   c4_repeats     genome with repeat families, satellites and microsatellites (synth.REPEAT_FAMILIES): wide ranges, high-copy
    #                  17-mers -- the packed table's escape lines --, with the kernel's own counters (lines, second lines, escapes);
    #   c4_real_reads  the random genome of rounds 2-3 (the series those rounds quote).
    # `library_ordered` = the same launch with the library's own batch-ordering pass forced on (automatic = off: it does not pay).
    c4_scale = args.c4_scale if args.c4_scale > 0 else (1.0 if args.scale == 1.0 else 0.0)
    if human and world == 1 and not multi and not args.no_c4 and c4_scale > 0 and rank == 0:
        del bwt, ref, rle, queries, got
        torch.cuda.empty_cache()

        def real_msbwt_line(name, key):
            nonlocal rc
            t0 = time.time()
            cfg4 = synth.CONFIGS[name]
            npy4, reads4 = early_indexes.get(name, c4_scale)
            log("%s: real MSBWT %s ready after %.1fs of waiting" % (key, os.path.basename(npy4), time.time() - t0))
            t0 = time.time()
            bwt4 = msbwt.RleBWT(device=local_rank)
            bwt4.set_query_length(31 if args.query_length_hint < 0 else args.query_length_hint)
            bwt4.load_numpy_file(npy4)
            total4 = bwt4.get_total_size()
            n4 = max(1000, int(args.c4_queries * min(1.0, c4_scale * 4))) if c4_scale < 1.0 else args.c4_queries
            # the queries of synth.read_kmers (same seed, same draws), gathered on the GPU instead of the host
            per = reads4.shape[1] - 31 + 1
            if n4 >= len(reads4) * per:
                d_q4 = torch.from_numpy(synth.read_kmers(reads4, 31)).to(dev)
            else:
                rng4 = np.random.default_rng(cfg4["qseed"])
                r4 = torch.from_numpy(rng4.integers(0, len(reads4), size=n4)).to(dev)
                p4 = torch.from_numpy(rng4.integers(0, per, size=n4)).to(dev)
                d_reads4 = torch.from_numpy(reads4).to(dev)
                d_q4 = torch.empty((n4, 31), dtype=torch.uint8, device=dev)
                for lo_s in range(0, n4, 20_000_000):
                    sl = slice(lo_s, lo_s + 20_000_000)
                    d_q4[sl] = d_reads4[r4[sl, None], p4[sl, None] + torch.arange(31, device=dev)[None, :]]
                del d_reads4, r4, p4
            n4 = d_q4.shape[0]
            log("%s: %d symbols on the GPU (%.1f MB, table depth %d, pair stride %d, typical range width %.1f), %d read-derived 31-mers in HBM in %.1fs"
                % (key, total4, bwt4.device_bytes() / 1e6, bwt4.get_table_depth(), bwt4.get_pair_stride(), bwt4.get_typical_range_width(), n4, time.time() - t0))
            b4 = Batch(bwt4, d_q4, 0, None, 31)
            shape4 = index_shape(bwt4, 31, False, n4)
            ordered = shape4["ordered"]
            o4, _, el4, kms4, launches4, _ = measure(b4, 0, n4, n4)
            sampled4 = dict(last_sampled)
            line = {"value": n4 * args.steps / el4, "unit": "queries/s", "ms_per_step": el4 / args.steps * 1e3, "steps": args.steps,
                    "batch_ordered_by_the_library": ordered,
                    "config": {"workload": "%s: the REAL multi-string BWT of %d synthetic %d-bp reads (%.0fx of a %d-bp %s genome, %.1f%% substitutions), "
                                           "%d symbols, built on the host in this run; %d read-derived 31-mers per step"
                                           % (name, len(reads4), reads4.shape[1], len(reads4) * reads4.shape[1] / max(1, int(cfg4["genome"] * c4_scale)),
                                              int(cfg4["genome"] * c4_scale),
                                              "repeat-bearing (synth.REPEAT_FAMILIES: SINE-, LINE-, LTR-, DNA-element-like families, segmental duplications, "
                                              "satellite arrays, microsatellites)" if cfg4.get("repeats") else "random", cfg4["err"] * 100, total4, n4),
                               "k": 31, "queries_per_step": n4, "bwt_symbols": total4, "index_bytes": bwt4.device_bytes(), "table_depth": lookup_depth(bwt4, 31),
                               "direct_table_depth": bwt4.get_table_depth(), "sparse_table_depth": bwt4.get_sparse_table(), "sparse_table": bwt4.sparse_table_info(),
                               "pair_stride": bwt4.get_pair_stride(), "typical_range_width": bwt4.get_typical_range_width(), "table": bwt4.table_info()}}
            line["search_counters"] = counted_pass(bwt4, b4, 0, n4, n4)
            # the same launch with the library's own batch-ordering pass forced on (msbwt_rle_set_batch_order(1); automatic = off, on
            # these very numbers): pack + two bucket passes + ordered search + counts back to the caller's order, all timed
            if not ordered and args.c4_lab:
                bwt4.set_batch_order(1)
                if bwt4.batch_order_for(31, n4):
                    saved = args.steps
                    args.steps = min(args.steps, 10)
                    u4, _, uel4, ukms4, _, _ = measure(b4, 0, n4, n4)
                    line["library_ordered"] = {"value": n4 * args.steps / uel4, "ms_per_step": uel4 / args.steps * 1e3, "kernels_ms": ukms4,
                                               "counts_equal_unordered_run": bool(torch.equal(u4, o4)),
                                               "note": "msbwt_rle_set_batch_order(1): the batch is packed, bucket-ordered by 22 key bits, counted in index order and "
                                                       "its counts are returned to the caller's order inside the launch; kernels_ms = all of it"}
                    if not line["library_ordered"]["counts_equal_unordered_run"]:
                        log("PARITY FAILURE on %s: the library-ordered batch's counts differ from the unordered run's" % key)
                        result["value"] = None
                        rc = 1
                    args.steps = saved
                    del u4
                bwt4.set_batch_order(-1)
            # What a k-mer costs by COPY NUMBER (the repeat-bearing line only): the batch's own counts pick sub-batches of read-derived 31-mers
            # that occur 1-99, 100-999, 1 000-9 999 and >= 10 000 times in the read set (a young-SINE or satellite 31-mer occurs tens of
            # thousands of times here; at human scale copy numbers are another ~50 x higher), each timed and counted on its own.
            if cfg4.get("repeats") and args.c4_lab:
                line["copy_number_bins"], bins_ok = copy_number_bins(bwt4, d_q4, o4, 31, ((1, 100), (100, 1000), (1000, 10_000), (10_000, 1 << 62)))
                if not bins_ok:
                    log("PARITY FAILURE on %s: a copy-number sub-batch counts differently from the main run" % key)
                    result["value"] = None
                    rc = 1
            # Which of this line's two modes the build is in (the time of a launch on a C4-sized index follows the physical
            # memory its pair blocks were given, 15 % apart): the pair blocks (+ table) rebuilt once, the same batch timed again -- a slow
            # line can then be told from a regression by its own record.
            try:
                saved = args.steps
                args.steps = min(args.steps, 5)
                try:
                    t_rebuild = time.time()
                    bwt4.set_pair_index(1 if bwt4.get_pair_index() else 0)
                    t_rebuild = time.time() - t_rebuild
                    p4, _, pel4, pkms4, _, _ = measure(Batch(bwt4, d_q4, 0, None, 31), 0, n4, n4)
                    line["pair_blocks_rebuilt"] = {"ms_per_step": pel4 / args.steps * 1e3, "kernel_ms": pkms4, "rebuild_s": t_rebuild,
                                                   "counts_equal_first_run": bool(torch.equal(p4, o4)),
                                                   "ratio_to_the_line": (pel4 / args.steps) / (el4 / saved),
                                                   "note": "msbwt_rle_set_pair_index on the loaded index frees and rebuilds the pair blocks and the table; the same "
                                                           "batch timed again.  A ratio far from 1 means the two builds sit in different modes of this line"}
                    del p4
                finally:
                    args.steps = saved
            except Exception as e:  # (a diagnostic: never the reason a bench line is lost)
                line["pair_blocks_rebuilt"] = {"error": repr(e)}
            # What a read set WITH errors gets from the sparse table when its complete form does not fit (round 6; the human-scale case: 1.3e10
            # distinct 23-mers, 185 GB): this index (0.5 % substitutions: 3.7 x as many distinct 23-mers as its genome has) with k undeclared --
            #   complete            the complete depth-23 table (fits here: the index is small);
            #   two_tier            msbwt_rle_set_sparse_tiers(1): entries only for the suffixes that occur at least twice, filter bits for the
            #                       rest, whose queries go on through the direct table;
            #   two_tier_budgeted   the AUTOMATIC choice under msbwt_rle_set_memory_budget set between the two forms' sizes;
            #   fallback            msbwt_rle_set_sparse_table(0): the deep direct table alone (what round 5 fell back to).
            if name == "c4" and not args.no_variants:
                modes = {}

                def timed_mode(tag):
                    saved_m = args.steps
                    args.steps = min(args.steps, 10)
                    try:
                        om, _, elm, kmsm, _, _ = measure(b4, 0, n4, n4)
                        cpm = counted_pass(bwt4, b4, 0, n4, n4)
                        inf = bwt4.sparse_table_info()
                        modes[tag] = {"value": n4 * args.steps / elm, "ms_per_step": elm / args.steps * 1e3, "kernel_ms": kmsm, "lines_per_query": cpm["lines_per_query"],
                                      "filter_fallbacks_per_query": cpm["raw"].get("tier_fallbacks", 0) / n4, "sparse_table_depth": bwt4.get_sparse_table(),
                                      "two_tier": bwt4.get_sparse_tiers(), "sparse_table_entries": inf["entries"], "sparse_table_filtered": inf["filtered"],
                                      "sparse_table_bytes": inf["bytes"] + inf["side_bytes"], "direct_table_depth": bwt4.get_table_depth(),
                                      "index_bytes": bwt4.device_bytes(), "pair_stride": bwt4.get_pair_stride(), "counts_equal_the_line": bool(torch.equal(om, o4))}
                        del om
                    finally:
                        args.steps = saved_m
                    return modes[tag]

                try:
                    bwt4.set_query_length(0)
                    complete = timed_mode("complete")
                    bwt4.set_sparse_tiers(1)
                    tier = timed_mode("two_tier")
                    bwt4.set_sparse_tiers(-1)
                    # a budget that the complete table of NO depth fits but a two-tier one does: plane blocks + overlapping pair blocks + the packed
                    # depth-15 direct table (what a budgeted index keeps beside a sparse table) + 0.6 of the complete table's bytes.  (On this index
                    # the complete depth-23 table sits at the least bucket count its 24-bit tags allow -- 2^25, 4.3 GB; depth 21's needs 3.3 GB --
                    # while the two-tier one, whose buckets hold a third of the entries, gets by with a probe limit of 3 and half the buckets: 2.1 GB.)
                    rest = (total4 // 256 + 1) * 128 + (total4 // 96 + 1) * 128 + ((4 ** 15 + 29) // 30) * 128 + 150_000_000
                    budget = rest + int(complete["sparse_table_bytes"] * 0.6)
                    bwt4.set_memory_budget(budget)
                    timed_mode("two_tier_budgeted")["budget_bytes"] = budget
                    bwt4.set_memory_budget(0)
                    bwt4.set_sparse_table(0)
                    timed_mode("fallback")
                    bwt4.set_sparse_table(-1)
                    bwt4.set_query_length(31 if args.query_length_hint < 0 else args.query_length_hint)
                    line_b = {"modes": modes, "value": modes["two_tier"]["value"], "unit": "queries/s",
                              "note": "the same index and batch, k undeclared: complete depth-23 sparse table / its two-tier form (forced) / the automatic choice under a "
                                      "memory budget between the two / no sparse table.  On this index the complete table costs a few GB; on a 30x HUMAN read set "
                                      "with errors it would cost 185 GB and not fit -- the two-tier line is what such a set gets, the fallback line what round 5 gave it"}
                    if not all(m["counts_equal_the_line"] for m in modes.values()):
                        log("PARITY FAILURE on c4_budgeted: a mode counts differently from the line")
                        result["value"] = None
                        rc = 1
                    result["c4_budgeted"] = line_b
                except msbwt.MsbwtError as e:
                    result["c4_budgeted"] = {"error": repr(e), "modes": modes}
            # The drop-in caller's path (a Rust caller behind the trait hands over HOST slices: msbwt_rle_count_kmers, ..._packed): the same
            # 31-mers from host memory and their counts back to it, PCIe copies inside the timed region -- never `value` -- next to what this
            # box's PCIe link moves when both directions are busy (pinned buffers of the same sizes, two streams).
            if name == "c4" and not args.no_variants:
                try:
                    nh = min(n4, 30_000_000)
                    h_q = d_q4[:nh].cpu().numpy()
                    exp_h = o4[:nh].cpu().numpy().astype(np.uint64)
                    out_h = np.zeros(nh, dtype=np.uint64)     # touched: no first-touch page faults inside the timed region

                    def best_of(fn, reps=3):
                        fn()
                        best = None
                        for _ in range(reps):
                            t_h = time.perf_counter()
                            fn()
                            dt_h = time.perf_counter() - t_h
                            best = dt_h if best is None else min(best, dt_h)
                        return best

                    host = {"queries": nh, "k": 31}
                    t_b = best_of(lambda: bwt4.count_kmers(h_q, out=out_h))
                    host.update({"bytes_qps": nh / t_b, "bytes_GBps": (h_q.nbytes + out_h.nbytes) / t_b / 1e9, "bytes_equal": bool(np.array_equal(out_h, exp_h))})
                    w_h = msbwt.rle_bwt.pack_2bit(h_q)
                    for bits, tag in ((64, "packed_u64"), (32, "packed_u32")):
                        o_p = np.zeros(nh, dtype=np.uint64 if bits == 64 else np.uint32)
                        t_p = best_of(lambda: bwt4.count_kmers_packed(w_h, 31, count_bits=bits, out=o_p))
                        host.update({tag + "_qps": nh / t_p, tag + "_GBps": (w_h.nbytes + o_p.nbytes) / t_p / 1e9,
                                     tag + "_equal": bool(np.array_equal(o_p.astype(np.uint64), exp_h))})
                    # the link itself: H2D of the query bytes and D2H of the counts at the same time, pinned
                    for tag, n_in, n_out in (("pcie_bytes_shape", h_q.nbytes, 8 * nh), ("pcie_packed_u32_shape", w_h.nbytes, 4 * nh)):
                        hin, hout = torch.empty(n_in, dtype=torch.uint8).pin_memory(), torch.empty(n_out, dtype=torch.uint8).pin_memory()
                        din, dout = torch.empty(n_in, dtype=torch.uint8, device=dev), torch.zeros(n_out, dtype=torch.uint8, device=dev)
                        s_in, s_out = torch.cuda.Stream(dev), torch.cuda.Stream(dev)

                        def both():
                            with torch.cuda.stream(s_in):
                                din.copy_(hin, non_blocking=True)
                            with torch.cuda.stream(s_out):
                                hout.copy_(dout, non_blocking=True)
                            s_in.synchronize()
                            s_out.synchronize()
                        host[tag + "_GBps"] = (n_in + n_out) / best_of(both) / 1e9
                        del hin, hout, din, dout
                    host["pcie_both_ways_GBps"] = host["pcie_bytes_shape_GBps"]
                    host["fraction_of_the_link"] = {"bytes": host["bytes_GBps"] / host["pcie_bytes_shape_GBps"],
                                                    "packed_u32": host["packed_u32_GBps"] / host["pcie_packed_u32_shape_GBps"]}
                    host["note"] = ("msbwt_rle_count_kmers / msbwt_rle_count_kmers_packed on HOST arrays of %d read-derived 31-mers of this line (pinned three-stage pipeline, "
                                    "csrc/host_pipeline.hpp), best of 3; *_GBps = bytes in + bytes out per second; pcie_*_shape_GBps = what the link moves with "
                                    "pinned buffers of the same sizes copied in both directions at once: the host path is bound by the link (and by the pageable-to-pinned "
                                    "staging copy on the host), not by the kernel" % nh)
                    if not (host["bytes_equal"] and host["packed_u64_equal"] and host["packed_u32_equal"]):
                        log("PARITY FAILURE: the host-pointer entry points count differently from the device path")
                        result["value"] = None
                        rc = 1
                    result["host_api"] = host
                    del h_q, w_h, out_h
                except Exception as e:  # noqa: BLE001  (a measurement aid: never the reason a bench line is lost)
                    result["host_api"] = {"error": repr(e)}
            if not args.no_oracle:
                from oracle import oracle as orc
                ref4 = orc.OracleRleBWT(8)
                ref4.load_numpy_file(npy4)
                ids4 = torch.from_numpy(np.sort(np.random.default_rng(6).choice(n4, size=min(n4, max(args.parity_sample, args.stats_sample)), replace=False))).to(dev)
                got4 = o4[ids4].cpu().numpy().astype(np.uint64)
                qs4 = d_q4[ids4].cpu().numpy()
                ns4 = min(len(qs4), args.parity_sample)
                exp4 = ref4.count_kmers(qs4[:ns4], nthreads=ncpu)
                m4 = int((exp4 != got4[:ns4]).sum())
                line["parity"] = {"checked": int(ns4), "mismatches": m4, "nonzero_counts_in_sample": int((got4 > 0).sum()),
                                  "mean_count_in_sample": float(got4.mean()), "max_count_in_sample": int(got4.max())}
                if m4:
                    log("PARITY FAILURE on %s: %d sampled counts differ from the oracle" % (key, m4))
                    result["value"] = None
                    rc = 1
                tq, tsrc, tnote, stamp = lookup_traffic(name, 31, shape4, "reads", total4, False, c4_scale == 1.0)
                live4 = None
                if c4_scale == 1.0 and not args.no_live_pmc:
                    # this line's HBM-side read traffic measured NOW, like the headline's: one rocprofv3 --pmc FETCH_SIZE child pass over the same
                    # index (its .npy is cached) and 5 x 10^7 of the same kind of queries; the GPU is handed over for it
                    keep = (o4, d_q4)
                    del bwt4
                    gc.collect()
                    torch.cuda.empty_cache()
                    t_live = time.time()
                    per_q4, detail4 = live_pmc_traffic(["--workload", name, "--query-kind", "reads"], 50_000_000, counters=("FETCH_SIZE",))
                    if per_q4 is None:
                        live4 = {"error": detail4}
                    else:
                        live4 = dict(detail4, bytes_per_query=per_q4, committed_bytes_per_query=tq, seconds=time.time() - t_live)
                        tq, tsrc, tnote = per_q4, "live: rocprofv3 --pmc FETCH_SIZE, one child pass of this run (same index, %d read-derived 31-mers per launch)" % detail4["queries_per_launch"], None
                    log("%s: live PMC traffic: %s" % (key, live4))
                line["roofline"] = roofline_block(orc, ref4, qs4, 31, ncpu, n4, kms4 / 1e3 if launches4 else el4 / args.steps, kms4, launches4, tq, tsrc,
                                                  tnote, stamp, shape4["kernel"], args.stats_sample, shape4["lookup_depth"], shape4["pair_index"], ordered,
                                                  sparse=shape4["lookup_depth"] == shape4["sparse_depth"] != 0)
                if live4 is not None:
                    line["roofline"]["traffic_live"] = live4
                if not args.no_cpu_baseline:
                    ncs4 = min(len(qs4), args.cpu_sample // 4)
                    t0 = time.time()
                    ref4.count_kmers(qs4[:ncs4], nthreads=1)
                    line["cpu_baseline"] = {"value": ncs4 / (time.time() - t0), "unit": "queries/s", "cores": 1, "kind": "port",
                                            "sample": "%d of the same queries, 1 thread" % ncs4}
            else:
                line["roofline"] = {"kernel_ms": kms4, "kernel_launches": launches4}
            line["telemetry"] = {"during_timed_region": dict(sampled4), "after": telemetry()}
            result[key] = line
            log("%s: %.3e q/s, %.2f ms per step%s" % (key, line["value"], line["ms_per_step"],
                                                      " (library-ordered: %.3e)" % line["library_ordered"]["value"] if "library_ordered" in line else ""))

        real_msbwt_line("c4r", "c4_repeats")
        torch.cuda.empty_cache()
        if not args.no_c4_random:
            real_msbwt_line("c4", "c4_real_reads")
            torch.cuda.empty_cache()
    # ---- the default line's HBM traffic, measured in THIS run (the committed summary stays beside it) ---------------------
    if (human and rank == 0 and world == 1 and not multi and args.scale == 1.0 and not args.no_oracle and not args.no_live_pmc
            and args.stream == "reads" and not args.queries and "roofline" in result and result.get("value")):
        bwt = ref = None   # hand the GPU (and 30 GB of host memory) to the child passes
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        t0 = time.time()
        child_args = ((["--k", str(args.k)] if args.k else []) + (["--table-depth", str(args.table_depth)] if args.table_depth > -2 else []) +
                      (["--sparse-depth", str(args.sparse_depth)] if args.sparse_depth > -2 else []) +
                      (["--query-length-hint", str(args.query_length_hint)] if args.query_length_hint >= 0 else []) +
                      (["--genome", args.genome] if args.genome != "random" else []))
        if args.genome == "repeats":   # (a lab line whose index takes longer to build: the read pass only, the writes are the counts)
            per_q, detail = live_pmc_traffic(child_args, 100_000_000, counters=("FETCH_SIZE",), patience=900)
        else:
            per_q, detail = live_pmc_traffic(child_args, 100_000_000)
        roof = result["roofline"]
        if per_q is None:
            roof["traffic_live"] = {"error": detail}
            log("live PMC traffic: %s" % detail)
        else:
            kern_s = roof["kernel_ms"] / 1e3
            committed = None if roof["traffic"] is None else roof["traffic"] / result["config"]["queries_per_gpu"]
            traffic = per_q * result["config"]["queries_per_gpu"]
            roof.update({"traffic": traffic, "achieved": traffic / kern_s / 1e9, "frac": traffic / kern_s / 1e9 / HBM_PEAK_GBS,
                         "traffic_source": "live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, two child passes of this run (same index, %d present k-mers per launch)" % detail["queries_per_launch"],
                         "traffic_note": None,
                         "traffic_live": dict(detail, bytes_per_query=per_q, committed_bytes_per_query=committed, seconds=time.time() - t0),
                         "random_lines": {"per_s": traffic / 128.0 / kern_s, "peak_per_s": RANDOM_LINE_PEAK, "frac": traffic / 128.0 / kern_s / RANDOM_LINE_PEAK}})
            log("live PMC traffic: %.1f B per query (committed summary: %s) in %.0fs" % (per_q, "%.1f" % committed if committed else "none", time.time() - t0))
    if rank == 0:
        if result.get("value") is None:
            rc = 1
        emit(result)
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    return rc


if __name__ == "__main__":
    sys.exit(main())
