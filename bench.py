#!/usr/bin/env python3
"""bench.py -- batched count_kmer throughput of the MI355X path (BASELINE.json metric).

One "step" = one pass of the hot path (msbwt_rle_count_kmers_device) over one batch of synthetic
queries that is already resident in HBM.

Default workload ("human") = the configuration BASELINE.json quotes its metric on, on ONE GPU:
k = 31 on a 30x-human-scale index (9e10 symbols; a structure-equivalent synthetic RLE stream,
see DESIGN.md), 3e8 PRESENT 31-mers per step (LF-walk; every query runs all 31 steps -- the
read-corrector case).  The literal configs[4] line (1e9 random 31-mers generated in HBM) is
carried in the extra key `c5_random_1e9`.  Other workloads: c2, c3 (--fused = configs[2]), c4, big.

--gpus N: one rank per GPU.  Launched plainly (no torchrun) the script starts
torch.distributed.run itself as a child process and relays its JSON line.  The index is
replicated; ONE fixed batch is sharded across the ranks (strong scaling, BASELINE configs[3]'s
shape) and every rank ends each step holding all counts after one RCCL all_gather.  The
weak-scaling figure (every rank the whole batch) is reported in the extra key `weak_scaling`.

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
KERNEL_SOURCES = ["kernels.hip", "lanes.hip", "search_common.hpp", "rank_ops.hpp", "kernels.hpp", "plane_index.hpp"]


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


def kernel_label(bwt, k, fused):
    """Name (as in the rocprofv3 kernel trace) of the kernel the library runs for this index and k."""
    which = bwt.search_kernel_for(k)
    reads, words = ("true" if fused else "false"), (3 if k <= 32 else 6)
    if which == "lanes":
        pair, s96 = bwt.get_pair_index(), bwt.get_pair_index() and bwt.get_pair_stride() == 96
        return "k_count_kmers_lanes<%s,%s,%d,%s>" % (reads, "true" if pair else "false", words, "true" if s96 else "false")
    return "k_count_kmers_tiled<%s,%d>" % (reads, words) if which == "groups" else "k_count_kmers_generic"


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="human", choices=["human", "c2", "c3", "c4", "big"])
    ap.add_argument("--big-symbols", type=float, default=2.0**33, help="workload big: BWT length")
    ap.add_argument("--big-mean-run", type=float, default=6.0)
    ap.add_argument("--stats-sample", type=int, default=2_000_000, help="queries used for the algorithmic-byte counters (0 = all)")
    ap.add_argument("--queries", type=int, default=0, help="queries per step, whole job (0 = the config's)")
    ap.add_argument("--k", type=int, default=0, help="override k")
    ap.add_argument("--query-kind", default="", choices=["", "random", "reads", "walk"])
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the index (tests)")
    ap.add_argument("--table-depth", type=int, default=-2, help="-2 = library default")
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
    ap.add_argument("--payload", default="auto", choices=["auto", "int64"],
                    help="N>1: auto = int16 counts on the wire when exact (falls back to int64), int64 = always wide")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-oracle", action="store_true",
                    help="profiling passes only: skip parity, algorithmic-byte counters and the CPU baseline (prints a lean line)")
    ap.add_argument("--no-c5", action="store_true", help="skip the extra 1e9-random-31-mer line of the default workload")
    ap.add_argument("--no-weak", action="store_true", help="N>1: skip the extra weak-scaling measurement")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the N>1 code path (process group, all_gather of the counts, barriers) even with one rank: "
                         "a one-GPU rehearsal of the RCCL calls themselves")
    ap.add_argument("--c5-queries", type=int, default=1_000_000_000)
    ap.add_argument("--parity-sample", type=int, default=200_000)
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


def device_random_kmers(torch, dev, n, k, seed, chunk=50_000_000):
    """uniform ACGT k-mers generated in HBM (BASELINE configs[4] asks for 1e9 queries: 31 GB that
    never need to exist on the host)"""
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    d_q = torch.empty((n, k), dtype=torch.uint8, device=dev)
    for lo_q in range(0, n, chunk):
        part = d_q[lo_q:lo_q + chunk]
        part.random_(0, 4, generator=gen)      # 0..3
        part.add_(1)                           # A C G -> 1 2 3
        part.masked_fill_(part == 4, 5)        # T -> 5
    return d_q


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
        os.write(result_fd, (json.dumps(obj) + "\n").encode())

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

    human = args.workload == "human"
    big = human or args.workload == "big"
    cfg = dict(synth.CONFIGS["c3" if big else args.workload])
    k = args.k or cfg["k"]
    kind = args.query_kind or ("walk" if big else cfg["queries"])
    nq = args.queries or (300_000_000 if human else 20_000_000 if big else cfg["nq"]) or 20_000_000
    if args.scale < 1.0 and not args.queries:
        nq = max(1000, int(nq * min(1.0, args.scale * (4 if not human else 40))))
    symbols = int((HUMAN_SYMBOLS if human else args.big_symbols) * (args.scale if human else 1.0))

    bwt = msbwt.RleBWT(device=local_rank)
    bwt.set_block_format(args.blocks)
    if args.table_depth > -2:
        bwt.set_table_depth(args.table_depth)
    t0 = time.time()
    rle = npy = reads = None
    if big:
        # structure-equivalent synthetic RLE stream (NOT a real BWT): sizes that cannot be
        # suffix-sorted here.  Same seed on every rank => identical replicas.
        rle, _ = synth.rle_stream(symbols, args.big_mean_run, 77)
        log("rank %d: synthetic RLE stream of %d bytes in %.1fs" % (rank, len(rle), time.time() - t0))
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
    log("rank %d: %d symbols on the GPU (%.1f MB index, table depth %d, pair index %s) in %.1fs"
        % (rank, total, bwt.device_bytes() / 1e6, bwt.get_table_depth(), bwt.get_pair_index(), time.time() - t0))

    # ---- the batch: identical on every rank (same seeds); d_q holds ALL nq queries in HBM ---------
    fused = args.fused and not big
    t0 = time.time()
    d_reads = None
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
        d_q = device_random_kmers(torch, dev, nq, k, 4242)
    elif kind == "random":
        d_q = torch.from_numpy(synth.random_kmers(nq, k, cfg["qseed"])).to(dev)
    else:
        d_q = torch.from_numpy(synth.read_kmers(reads, k, limit=nq, seed=cfg["qseed"])).to(dev)
        nq = d_q.shape[0]
    torch.cuda.synchronize(dev)
    log("rank %d: %d %s %d-mers in HBM in %.1fs" % (rank, nq, kind, k, time.time() - t0))
    stream = torch.cuda.current_stream(dev).cuda_stream

    strong = multi and args.scaling == "strong" and not fused
    if strong:
        lo, hi = msbwt.sharded.shard_bounds(nq, world, rank)  # 16-query aligned: every shard keeps the tiled kernel
        cap = msbwt.sharded.shard_capacity(nq, world)
    else:
        lo, hi, cap = 0, nq, nq
    mine_n = hi - lo

    def count_into(out, a=lo, b=hi):
        if fused:
            bwt.count_read_kmers_device(d_reads.data_ptr(), rlen, nread, k, False, out.data_ptr(), 0, stream)
        elif b > a:
            bwt.count_kmers_device(d_q.data_ptr() + a * k, k, b - a, out.data_ptr(), stream)

    def fence():
        if multi:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # ---- the exchange step (N > 1): every rank ends each step holding ALL counts ------------------
    # 8 bytes per query over xGMI can cost more than the search itself, so the counts travel as
    # int16 whenever that is exact (a device-side overflow flag is kept per step and checked after
    # the loop; any overflow re-runs the whole measurement with 64-bit payloads), the all_gather
    # runs asynchronously on RCCL's stream and overlaps the next step's kernel (two buffers in
    # flight), and each rank widens what it received back to u64 -- inside the timed region.
    NARROW_MAX = 32767

    def run_steps(nsteps, narrow, a, b, per_rank):
        """nsteps passes over queries [a, b) of the batch; N > 1: + all_gather of `per_rank` counts per rank"""
        outs = [torch.zeros(per_rank, dtype=torch.int64, device=dev) for _ in range(2)]
        if not multi:
            for _ in range(nsteps):
                count_into(outs[0], a, b)
            return outs[0], None, False
        pay_dtype = torch.int16 if narrow else torch.int64
        sends = [torch.zeros(per_rank, dtype=pay_dtype, device=dev) for _ in range(2)]
        recvs = [torch.empty(per_rank * world, dtype=pay_dtype, device=dev) for _ in range(2)]
        d_all = torch.empty(per_rank * world, dtype=torch.int64, device=dev)
        overflow = torch.zeros((), dtype=torch.bool, device=dev)
        works = [None, None]

        def finish(j):  # widen what slot j received (the collective is done once wait() returns)
            if works[j] is not None:
                works[j].wait()
                d_all.copy_(recvs[j])
                works[j] = None

        for i in range(nsteps):
            j = i & 1
            finish(j)
            count_into(outs[j], a, b)
            if narrow and outs[j].numel():  # one reduction pass for both ends (u64 counts >= 2^63 look negative)
                mn, mx = torch.aminmax(outs[j])
                overflow |= (mx > NARROW_MAX) | (mn < 0)
            sends[j].copy_(outs[j])
            # the payload crosses as raw bytes: neither NCCL/RCCL nor gloo has a 16-bit integer type
            if args.dist_backend == "nccl":
                works[j] = dist.all_gather_into_tensor(recvs[j].view(torch.uint8), sends[j].view(torch.uint8),
                                                       async_op=True)  # RCCL over xGMI
            else:  # rehearsal: same logic, collective through host memory
                host = torch.empty(per_rank * world, dtype=pay_dtype)
                dist.all_gather_into_tensor(host.view(torch.uint8), sends[j].cpu().view(torch.uint8))
                recvs[j].copy_(host)
                d_all.copy_(recvs[j])
        finish(nsteps & 1)        # the older of the two outstanding steps first
        finish((nsteps - 1) & 1)
        return outs[(nsteps - 1) & 1], d_all, overflow

    def timed(narrow, a, b, per_rank):
        run_steps(args.warmup, narrow, a, b, per_rank)
        fence()
        bwt.device_status(stream)
        bwt.set_kernel_timing(True)
        t_start = time.perf_counter()
        d_mine, d_everything, ovf = run_steps(args.steps, narrow, a, b, per_rank)
        fence()
        dt = time.perf_counter() - t_start
        bwt.set_kernel_timing(False)
        k_ms, n_launch = bwt.kernel_time_ms()
        bwt.device_status(stream)
        if multi:
            flag = torch.tensor([1.0 if bool(ovf) else 0.0, dt], dtype=torch.float64,
                                device=dev if args.dist_backend == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            return d_mine, d_everything, float(flag[1].item()), k_ms, n_launch, flag[0].item() > 0
        return d_mine, d_everything, dt, k_ms, n_launch, False

    def measure(a, b, per_rank):
        narrow = multi and args.payload == "auto"
        res = timed(narrow, a, b, per_rank)
        if res[5]:  # some count did not fit int16: measure again with u64 payloads (always exact)
            log("counts exceed int16: re-running with 64-bit payloads")
            narrow = False
            res = timed(False, a, b, per_rank)
        return res[:5] + (narrow,)

    d_out, d_all, elapsed, kernel_ms, launches, narrow = measure(lo, hi, cap)
    if multi:  # the gathered vector must contain this rank's own counts where they belong
        assert torch.equal(d_all[rank * cap:rank * cap + mine_n], d_out[:mine_n]), "gathered counts differ from the local ones"
    ms_per_step = elapsed / args.steps * 1e3
    job_queries = nq if (strong or not multi) else nq * world
    value = job_queries * args.steps / elapsed

    # the whole batch's counts as one vector (strong: stitched from the gathered shards)
    if strong:
        spans = [msbwt.sharded.shard_bounds(nq, world, r) for r in range(world)]
        d_counts = torch.cat([d_all[r * cap:r * cap + (b - a)] for r, (a, b) in enumerate(spans)])
    else:
        d_counts = d_out

    weak = None
    if strong and not args.no_weak:
        # every rank the same (whole or 1e8-query) batch, all N x n counts gathered each step; bounded so that
        # the N x n gathered vectors fit next to a 200 GB index at N = 8
        nw = min(nq, 100_000_000)
        w_out, w_all, w_elapsed, _, _, w_narrow = measure(0, nw, nw)
        assert torch.equal(w_all[rank * nw:(rank + 1) * nw], w_out)
        assert torch.equal(w_out, d_counts[:nw]), "sharded counts differ from one GPU's counts of the same queries"
        weak = {"value": nw * world * args.steps / w_elapsed, "unit": "queries/s", "ms_per_step": w_elapsed / args.steps * 1e3,
                "queries_per_gpu": nw, "payload": "int16" if w_narrow else "int64",
                "note": "every rank runs the same %d queries and all N x n counts are all_gathered each step" % nw}
        del w_out, w_all

    kind_text = {"walk": "present (LF-walk)", "random": "random", "reads": "read-derived"}[kind]
    if big:
        wl = ("%s: structure-equivalent synthetic RLE stream (NOT a real BWT; 30x-human-scale stand-in), %d symbols, mean run %.1f; "
              "%d %s %d-mers per step" % (args.workload, total, args.big_mean_run, nq, kind_text, k))
    else:
        wl = ("%s: %d synthetic %d-bp reads (%.0fx of a %d-bp random genome, %.1f%% subst.) -> MSBWT of %d symbols; %d %s %d-mers per step"
              % (args.workload, len(reads), reads.shape[1], len(reads) * reads.shape[1] / max(1, int(cfg["genome"] * args.scale)),
                 int(cfg["genome"] * args.scale), cfg["err"] * 100, total, nq,
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
        "scaling": "strong" if strong else "weak",
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
        "config": {
            "workload": wl, "k": k, "queries_per_step": job_queries, "queries_per_gpu": mine_n if strong else nq,
            "bwt_symbols": total, "index_bytes": bwt.device_bytes(),
            "table_depth": bwt.get_table_depth(), "pair_index": bwt.get_pair_index(), "block_format": bwt.get_block_format(),
            "parallelism": ("index replicated x%d; %s; per step one %s all_gather of all counts (%s payload, widened to u64 on "
                            "arrival), overlapped with the next step's kernel"
                            % (world, "ONE fixed batch sharded over the ranks" if strong else "every rank its own whole batch",
                               "RCCL" if args.dist_backend == "nccl" else "gloo (rehearsal)",
                               "int16" if narrow else "int64")) if multi else "1 GPU",
        },
    }
    if weak is not None:
        result["weak_scaling"] = weak

    # the rows the oracle will check travel to the host now; then the batch leaves HBM (the extra line
    # below needs 39 GB next to a 200 GB index)
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
    del d_counts, d_out, d_all, d_q
    torch.cuda.empty_cache()

    # ---- BASELINE configs[4] in its literal shape on this GPU: 1e9 random 31-mers generated in HBM ----
    c5 = None
    if human and world == 1 and not args.no_c5 and args.scale == 1.0:
        t0 = time.time()
        n5 = args.c5_queries
        d_q5 = device_random_kmers(torch, dev, n5, k, 99)
        out5 = torch.empty(n5, dtype=torch.int64, device=dev)
        bwt.count_kmers_device(d_q5.data_ptr(), k, n5, out5.data_ptr(), stream)  # warm-up pass
        torch.cuda.synchronize(dev)
        passes = 3
        t1 = time.perf_counter()
        for _ in range(passes):
            bwt.count_kmers_device(d_q5.data_ptr(), k, n5, out5.data_ptr(), stream)
        torch.cuda.synchronize(dev)
        dt5 = (time.perf_counter() - t1) / passes
        bwt.device_status(stream)
        ids5 = torch.from_numpy(np.sort(np.random.default_rng(5).choice(n5, size=min(n5, args.parity_sample), replace=False))).to(dev)
        c5 = {"queries": n5, "ms_per_pass": dt5 * 1e3, "value": n5 / dt5, "unit": "queries/s",
              "kind": "uniform random ACGT 31-mers generated in HBM (BASELINE configs[4]'s query shape, one GPU)",
              "_q": d_q5[ids5].cpu().numpy(), "_got": out5[ids5].cpu().numpy().astype(np.uint64)}
        del d_q5, out5
        log("c5 line: %d random %d-mers in %.1f ms per pass (%.1fs incl. generation)" % (n5, k, dt5 * 1e3, time.time() - t0))

    rc = 0
    if rank == 0 and args.no_oracle:
        result["roofline"] = {"kernel_ms": kernel_ms, "kernel_launches": launches, "note": "--no-oracle: lean line of a profiling pass"}
        if c5 is not None:
            del c5["_q"], c5["_got"]
            result["c5_random_1e9"] = c5
        emit(result)
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
        ncpu = min(os.cpu_count() or 1, 16)
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
        if mism:
            log("PARITY FAILURE: %d sampled counts differ from the oracle" % mism)
            result["value"] = None
            rc = 1
        # algorithmic bytes of the reference algorithm for THIS query set: exact counters
        st = orc.Stats()
        nst = min(len(queries), args.stats_sample) if args.stats_sample else len(queries)
        t0 = time.time()
        ref.count_kmers(queries[:nst], nthreads=ncpu, stats=st)
        t_all = time.time() - t0
        per_launch_q = mine_n if strong else nq
        alg_bytes = st.algorithmic_bytes(k) / nst * per_launch_q  # exact when nst == nq, else scaled from the sample
        kern_s = kernel_ms / 1e3 if launches else elapsed / args.steps
        # HBM traffic of one launch: PMC counters from separate rocprofv3 passes (FETCH_SIZE x2 on gfx950,
        # WRITE_SIZE), committed under profiles/ and keyed by configuration AND kernel sources
        traffic, traffic_src, traffic_note = None, None, "no PMC summary for this configuration under profiles/"
        stamp = kernel_stamp()
        try:
            for ent in json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))["entries"]:
                same = (ent["workload"] == args.workload and ent["k"] == k and ent["table_depth"] == bwt.get_table_depth()
                        and ent.get("pair_index", False) == bwt.get_pair_index() and ent.get("query_kind", cfg["queries"]) == kind
                        and ent.get("bwt_symbols", total) == total and args.scale == 1.0 and bool(ent.get("fused", False)) == bool(fused))
                if same and ent.get("kernel_stamp") != stamp:
                    traffic_note = "stale: %s was taken with kernel sources %s, the tree has %s" % (ent["source"], ent.get("kernel_stamp"), stamp)
                elif same:
                    traffic = ent["traffic_bytes_per_query"] * per_launch_q
                    traffic_src, traffic_note = ent["source"], None
        except (OSError, KeyError, ValueError):
            pass
        achieved = None if traffic is None else traffic / kern_s / 1e9
        result["roofline"] = {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": None if achieved is None else achieved / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_source": traffic_src, "traffic_note": traffic_note, "kernel_stamp": stamp,
            "kernel": kernel_label(bwt, k, fused),
            "kernel_ms": kernel_ms, "kernel_launches": launches,
            "algorithmic": {
                "bytes_per_launch": int(alg_bytes), "bytes_per_query": alg_bytes / per_launch_q,
                "GBps": alg_bytes / kern_s / 1e9, "x_hbm_peak": alg_bytes / kern_s / 1e9 / HBM_PEAK_GBS,
                "mean_steps_per_query": st.steps / nst, "mean_bin_visits_per_query": st.visits / nst, "stats_queries": int(nst),
                "note": "bytes the REFERENCE algorithm touches for this query set (SURVEY 8d: 56 B of samples + scanned RLE bytes per "
                        "bin visit, + k + 8 per query) / kernel time.  The suffix table and pair steps make this kernel move fewer "
                        "bytes than that, so it is a speed-up measure, not a bandwidth fraction"},
            "note": "achieved/frac = measured HBM-side traffic of one launch (rocprofv3 PMC, committed summary) / kernel time measured "
                    "live with HIP events on the launch stream / 8 TB/s",
            "random_lines": None if traffic is None else {
                "per_s": traffic / 128.0 / kern_s, "peak_per_s": RANDOM_LINE_PEAK, "frac": traffic / 128.0 / kern_s / RANDOM_LINE_PEAK},
        }
        if world == 1 and not args.no_cpu_baseline:
            ncs = min(len(queries), args.cpu_sample)
            t0 = time.time()
            ref.count_kmers(queries[:ncs], nthreads=1)
            t1 = time.time() - t0
            result["cpu_baseline"] = {
                "value": ncs / t1, "unit": "queries/s", "cores": 1, "kind": "port",
                "cpu_model": cpu_model(), "nproc": os.cpu_count(),
                "sample": "%d queries sampled from the same batch, same RLE stream / comp_msbwt.npy, 1 thread (the reference is single-threaded), -O3 C restatement" % ncs,
                "all_cores": {"value": nst / t_all, "cores": ncpu, "note": "same sample, static partition, instrumented build"},
            }
        emit(result)
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    return rc


if __name__ == "__main__":
    sys.exit(main())
