#!/usr/bin/env python3
"""bench.py -- batched count_kmer throughput of the MI355X path (BASELINE.json metric).

One "step" = one pass of the hot path (msbwt_rle_count_kmers_device) over one batch of
synthetic queries that is already resident in HBM.  Default workload = BASELINE.json
configs[1] ("c2": 1M synthetic 100-bp reads -> MSBWT, 10M random 21-mers); --workload c3
runs the E. coli-like config (read-derived 31-mers).  With --gpus N (launched by
torch.distributed.run, one rank per GPU) the index is replicated, every rank runs its own
batch (weak scaling) and the per-rank counts are gathered with one RCCL all_gather per step.

Prints ONE JSON line (rank 0).  See DESIGN.md "Measurement" for how the roofline figures are
defined.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s is achievable
RANDOM_LINE_PEAK = 4.4e10  # dependent random 128-byte lines/s, measured (tools/ubench_gather.hip, 1-200 GB tables)


def log(msg):
    print("[bench] " + msg, file=sys.stderr, flush=True)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def make_queries(synth, cfg, reads, nq, k, seed, kind):
    if kind == "random":
        return synth.random_kmers(nq, k, seed)
    return synth.read_kmers(reads, k, limit=nq, seed=seed)


def walk_kmers(torch, bwt, dev, total, n, k, seed):
    """k-mers that are present in the index, for streams that are not the BWT of known reads:
    start at a random row r with the one-row range [r, r+1) and prepend, k times, the symbol
    stored at that row (the one symbol whose constrain_range keeps the row) -- an LF walk run
    with the product's own batched constrain_ranges.  Walks that meet '$' or 'N' are dropped
    (read k-mers never contain them).  Workload generation only; outside every timed region."""
    rng = np.random.default_rng(seed)
    m = int(n * 1.6) + 1024
    stream = torch.cuda.current_stream(dev).cuda_stream
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
            torch.cuda.synchronize(dev)
            hit = (oh - ol) == 1
            nl = torch.where(hit, ol, nl)
            nh = torch.where(hit, oh, nh)
            sym_at = torch.where(hit, torch.full_like(sym_at, s), sym_at)
        kmers[:, k - 1 - step] = sym_at
        ok &= (sym_at != 0) & (sym_at != 4)
        l, h = nl, nh
    bwt.device_status(stream)
    out = kmers[ok][:n].cpu().numpy()
    assert len(out) == n, "LF walk produced too few ACGT-only k-mers (%d of %d)" % (len(out), n)
    return np.ascontiguousarray(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2", choices=["c2", "c3", "c4", "big"])
    ap.add_argument("--big-symbols", type=float, default=2.0**33, help="workload big: BWT length")
    ap.add_argument("--big-mean-run", type=float, default=6.0)
    ap.add_argument("--stats-sample", type=int, default=0, help="queries used for the algorithmic-byte counters (0 = all)")
    ap.add_argument("--queries", type=int, default=0, help="queries per GPU per step (0 = the config's)")
    ap.add_argument("--k", type=int, default=0, help="override k")
    ap.add_argument("--query-kind", default="", choices=["", "random", "reads"])
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the index (tests)")
    ap.add_argument("--table-depth", type=int, default=-2, help="-2 = library default")
    ap.add_argument("--fused", action="store_true",
                    help="c2/c3: queries = every k-mer window of every read, prepared in-kernel from the reads "
                         "(msbwt_rle_count_read_kmers_device); BASELINE.json configs[2] is --workload c3 --fused")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL over xGMI, the real thing) or gloo (rehearsal of the N>1 path on fewer GPUs: "
                         "counts are gathered through host memory, ranks may share a GPU)")
    ap.add_argument("--device-queries", action="store_true",
                    help="workload big, random queries: generate the batch in HBM (torch PRNG) instead of uploading it -- "
                         "BASELINE configs[4] asks for 1e9 queries, 31 GB that never need to exist on the host")
    ap.add_argument("--payload", default="auto", choices=["auto", "int64"],
                    help="N>1: auto = int16 counts on the wire when exact (falls back to int64), int64 = always wide")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--parity-sample", type=int, default=200_000)
    ap.add_argument("--cpu-sample", type=int, default=1_000_000)
    args = ap.parse_args()

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
    if world > 1:
        import torch.distributed as dist
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    msbwt = importlib.import_module("rust-msbwt_amd")
    import synth

    big = args.workload == "big"
    cfg = dict(synth.CONFIGS["c3" if big else args.workload])
    k = args.k or cfg["k"]
    kind = args.query_kind or ("walk" if big else cfg["queries"])
    nq = args.queries or (20_000_000 if big else cfg["nq"]) or 20_000_000
    nq = int(nq * min(1.0, args.scale * 4)) if args.scale < 1.0 and not args.queries else nq

    bwt = msbwt.RleBWT(device=local_rank)
    if args.table_depth > -2:
        bwt.set_table_depth(args.table_depth)
    t0 = time.time()
    if big:
        # structure-equivalent synthetic RLE stream (NOT a real BWT): sizes that cannot be
        # suffix-sorted here.  Same seed on every rank => identical replicas.
        rle, _ = synth.rle_stream(int(args.big_symbols), args.big_mean_run, 77)
        log("rank %d: synthetic RLE stream of %d bytes in %.1fs" % (rank, len(rle), time.time() - t0))
        t0 = time.time()
        bwt.load_vector(rle)
        npy, reads = None, None
    else:
        # rank 0 builds (and caches) the index file, everyone loads it
        if rank == 0:
            npy, reads = synth.workload_index(args.workload, args.scale)
        if world > 1:
            dist.barrier()
        if rank != 0:
            npy, reads = synth.workload_index(args.workload, args.scale)
        log("rank %d: index file %s ready in %.1fs" % (rank, os.path.basename(npy), time.time() - t0))
        t0 = time.time()
        bwt.load_numpy_file(npy)
    total = bwt.get_total_size()
    log("rank %d: %d symbols on the GPU (%.1f MB index, table depth %d) in %.1fs"
        % (rank, total, bwt.device_bytes() / 1e6, bwt.get_table_depth(), time.time() - t0))

    fused = args.fused and not big
    device_sampled = False
    if fused:
        kind = "reads"
        nread, rlen = reads.shape
        wins = rlen - k + 1
        nq = nread * wins
        d_reads = torch.from_numpy(reads).to(dev)
        # the oracle side needs explicit k-mers only for the sampled checks
        rng = np.random.default_rng(5 + rank)
        sample_ids = np.sort(rng.choice(nq, size=min(nq, max(args.parity_sample, args.stats_sample or 2_000_000, args.cpu_sample)), replace=False))
        queries = np.ascontiguousarray(reads[(sample_ids // wins)[:, None], (sample_ids % wins)[:, None] + np.arange(k)[None, :]])
    elif big and kind == "walk":
        t0 = time.time()
        queries = walk_kmers(torch, bwt, dev, total, nq, k, 4242 + rank)
        log("rank %d: %d present %d-mers by LF-walk on the GPU in %.1fs" % (rank, len(queries), k, time.time() - t0))
    elif big and args.device_queries:
        # uniform ACGT k-mers generated on the device; only the rows the oracle checks travel to the host
        gen = torch.Generator(device=dev)
        gen.manual_seed(4242 + 1000 * rank)
        d_q = torch.empty((nq, k), dtype=torch.uint8, device=dev)
        chunk = 50_000_000
        for lo_q in range(0, nq, chunk):
            part = d_q[lo_q:lo_q + chunk]
            part.random_(0, 4, generator=gen)      # 0..3
            part.add_(1)                           # A C G -> 1 2 3
            part.masked_fill_(part == 4, 5)        # T -> 5
        rng = np.random.default_rng(7 + rank)
        want = max(args.parity_sample, args.stats_sample or 1_000_000, args.cpu_sample)
        sample_ids = np.sort(rng.choice(nq, size=min(nq, want), replace=False))
        queries = d_q[torch.from_numpy(sample_ids).to(dev)].cpu().numpy()
        device_sampled = True
    elif big:
        queries = synth.random_kmers(nq, k, 4242 + 1000 * rank)
    else:
        queries = make_queries(synth, cfg, reads, nq, k, cfg["qseed"] + 1000 * rank, kind)
    if not fused and not device_sampled:
        nq = len(queries)
        d_q = torch.from_numpy(queries).to(dev)
    stream = torch.cuda.current_stream(dev).cuda_stream

    def count_into(out):
        if fused:
            bwt.count_read_kmers_device(d_reads.data_ptr(), rlen, nread, k, False, out.data_ptr(), 0, stream)
        else:
            bwt.count_kmers_device(d_q.data_ptr(), k, nq, out.data_ptr(), stream)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # ---- the exchange step (N > 1): every rank ends each step holding ALL N x nq u64 counts ------
    # 8 bytes per query over xGMI would cost more than the search itself (C2: 0.3 ms of kernel per
    # 10 M queries, 80 MB of counts), so the counts travel as int16 whenever that is exact (a
    # device-side overflow flag is kept per step and checked after the loop; any overflow re-runs
    # the whole measurement with 64-bit payloads), the all_gather runs asynchronously on RCCL's
    # stream and overlaps the next step's kernel (two buffers in flight), and each rank widens what
    # it received back to u64 -- inside the timed region.
    NARROW_MAX = 32767

    def run_steps(nsteps, narrow):
        outs = [torch.empty(nq, dtype=torch.int64, device=dev) for _ in range(2)]
        if world == 1:
            for _ in range(nsteps):
                count_into(outs[0])
            return outs[0], None, False
        pay_dtype = torch.int16 if narrow else torch.int64
        sends = [torch.empty(nq, dtype=pay_dtype, device=dev) for _ in range(2)]
        recvs = [torch.empty(nq * world, dtype=pay_dtype, device=dev) for _ in range(2)]
        d_all = torch.empty(nq * world, dtype=torch.int64, device=dev)
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
            count_into(outs[j])
            if narrow:
                overflow |= (outs[j] > NARROW_MAX).any() | (outs[j] < 0).any()
            sends[j].copy_(outs[j])
            # the payload crosses as raw bytes: neither NCCL/RCCL nor gloo has a 16-bit integer type
            if args.dist_backend == "nccl":
                works[j] = dist.all_gather_into_tensor(recvs[j].view(torch.uint8), sends[j].view(torch.uint8),
                                                       async_op=True)  # RCCL over xGMI
            else:  # rehearsal: same logic, collective through host memory
                host = torch.empty(nq * world, dtype=pay_dtype)
                dist.all_gather_into_tensor(host.view(torch.uint8), sends[j].cpu().view(torch.uint8))
                recvs[j].copy_(host)
                d_all.copy_(recvs[j])
        finish(nsteps & 1)        # the older of the two outstanding steps first
        finish((nsteps - 1) & 1)
        return outs[(nsteps - 1) & 1], d_all, overflow

    def timed(narrow):
        run_steps(args.warmup, narrow)
        fence()
        bwt.device_status(stream)
        bwt.set_kernel_timing(True)
        t_start = time.perf_counter()
        d_mine, d_everything, ovf = run_steps(args.steps, narrow)
        fence()
        dt = time.perf_counter() - t_start
        bwt.set_kernel_timing(False)
        k_ms, n_launch = bwt.kernel_time_ms()
        bwt.device_status(stream)
        if world > 1:
            flag = torch.tensor([1.0 if bool(ovf) else 0.0, dt], dtype=torch.float64,
                                device=dev if args.dist_backend == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            return d_mine, d_everything, float(flag[1].item()), k_ms, n_launch, flag[0].item() > 0
        return d_mine, d_everything, dt, k_ms, n_launch, False

    narrow = world > 1 and args.payload == "auto"
    d_out, d_all, elapsed, kernel_ms, launches, overflowed = timed(narrow)
    if overflowed:  # some count did not fit int16: measure again with u64 payloads (always exact)
        log("counts exceed int16: re-running with 64-bit payloads")
        narrow = False
        d_out, d_all, elapsed, kernel_ms, launches, _ = timed(False)
    if world > 1:  # the gathered vector must contain this rank's own counts where they belong
        assert torch.equal(d_all[rank * nq:(rank + 1) * nq], d_out), "gathered counts differ from the local ones"
    ms_per_step = elapsed / args.steps * 1e3
    value = nq * world * args.steps / elapsed

    result = {
        "metric": "k-mer count queries/sec (whole node)",
        "value": value,
        "unit": "queries/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
        "config": {
            "workload": ("big: structure-equivalent synthetic RLE stream (NOT a real BWT), %d symbols, mean run %.1f; "
                         "%d %s %d-mers per GPU per step" % (total, args.big_mean_run, nq,
                                                             "present (LF-walk)" if kind == "walk" else "random, generated in HBM" if device_sampled else "random", k)) if big else
                        "%s: %d synthetic %d-bp reads (%.0fx of a %d-bp random genome, %.1f%% subst.) -> MSBWT of %d symbols; "
                        "%d %s %d-mers per GPU per step" % (
                            args.workload, len(reads), reads.shape[1], len(reads) * reads.shape[1] / max(1, int(cfg["genome"] * args.scale)),
                            int(cfg["genome"] * args.scale), cfg["err"] * 100, total, nq,
                            "random" if kind == "random" else "read-derived (ALL windows of ALL reads, prepared in-kernel)" if fused else "read-derived", k),
            "k": k, "queries_per_gpu": nq, "bwt_symbols": total, "index_bytes": bwt.device_bytes(),
            "table_depth": bwt.get_table_depth(), "pair_index": bwt.get_pair_index(),
            "parallelism": ("query-sharded x%d, index replicated; per step one %s all_gather of all counts (%s payload, "
                            "widened to u64 on arrival), overlapped with the next step's kernel"
                            % (world, "RCCL" if args.dist_backend == "nccl" else "gloo (rehearsal)",
                               "int16" if narrow else "int64")) if world > 1 else "1 GPU",
        },
    }

    if rank == 0:
        # ---- parity + algorithmic bytes (the oracle is the checker, never the thing timed as `value`) ----
        from oracle import oracle as orc
        ref = orc.OracleRleBWT(8)
        if big:
            ref.load_vector(rle)
        else:
            ref.load_numpy_file(npy)
        if fused or device_sampled:  # `queries` holds only sampled rows; sample_ids are their positions in the output
            got = d_out[torch.from_numpy(sample_ids).to(dev)].cpu().numpy().astype(np.uint64)
            nq_all, nq = nq, len(queries)
        else:
            got = d_out.cpu().numpy().astype(np.uint64)
        ns = min(nq, args.parity_sample)
        sel = np.linspace(0, nq - 1, ns).astype(np.int64)
        exp = ref.count_kmers(queries[sel], nthreads=os.cpu_count() or 1)
        mism = int((exp != got[sel]).sum())
        result["parity"] = {"checked": int(ns), "mismatches": mism, "vs": "CPU oracle (restated RleBWT::count_kmer)",
                            "nonzero_counts": int((got > 0).sum())}
        if mism:
            log("PARITY FAILURE: %d of %d sampled counts differ" % (mism, ns))
        # algorithmic bytes of the reference algorithm for THIS query set: exact counters
        st = orc.Stats()
        ncpu = min(os.cpu_count() or 1, 16)
        nst = min(nq, args.stats_sample) if args.stats_sample else nq
        t0 = time.time()
        ref.count_kmers(queries[:nst], nthreads=ncpu, stats=st)
        t_all = time.time() - t0
        if fused or device_sampled:
            nq = nq_all
        alg_bytes = st.algorithmic_bytes(k) * (nq / nst)  # exact when nst == nq, else scaled from the sample
        kern_s = kernel_ms / 1e3 if launches else elapsed / args.steps
        achieved = alg_bytes / kern_s / 1e9
        traffic, traffic_src = None, None
        try:  # PMC counters are collected in separate rocprofv3 passes; their summary is committed
            for ent in json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))["entries"]:
                if ent["workload"] == args.workload and ent["k"] == k and ent["table_depth"] == bwt.get_table_depth() \
                        and ent.get("pair_index", False) == bwt.get_pair_index() \
                        and kind == cfg["queries"] and args.scale == 1.0 and not fused:
                    traffic = ent["traffic_bytes_per_query"] * nq
                    traffic_src = ent["source"]
        except (OSError, KeyError, ValueError):
            pass
        result["roofline"] = {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
            "kernel": "k_count_kmers_tiled<reads>" if fused else "k_count_kmers_tiled" if 1 <= k <= 32 else "k_count_kmers_generic", "kernel_ms": kernel_ms, "kernel_launches": launches,
            "algorithmic_bytes_per_launch": int(alg_bytes),
            "algorithmic_bytes_per_query": alg_bytes / nq,
            "note": "achieved = the REFERENCE algorithm's bytes for this query set / kernel time (SURVEY 8d); the suffix "
                    "table and pair steps make the kernel move fewer bytes than that, so frac can exceed 1 -- "
                    "`traffic` is what it really moved, `random_lines` prices that against the measured "
                    "random-128-byte-line rate of the memory system (tools/ubench_gather.hip)",
            "random_lines": None if traffic is None else {
                "per_s": traffic / 128.0 / kern_s, "peak_per_s": RANDOM_LINE_PEAK, "frac": traffic / 128.0 / kern_s / RANDOM_LINE_PEAK},
            "mean_steps_per_query": st.steps / nst, "mean_bin_visits_per_query": st.visits / nst,
            "stats_queries": int(nst),
        }
        if world == 1 and not args.no_cpu_baseline:
            ncs = min(nq, args.cpu_sample)
            t0 = time.time()
            ref.count_kmers(queries[:ncs], nthreads=1)
            t1 = time.time() - t0
            result["cpu_baseline"] = {
                "value": ncs / t1, "unit": "queries/s", "cores": 1, "kind": "port",
                "cpu_model": cpu_model(), "nproc": os.cpu_count(),
                "sample": "first %d queries of the same batch, same comp_msbwt.npy, 1 thread (the reference is single-threaded), -O3 C restatement" % ncs,
                "all_cores": {"value": nst / t_all, "cores": ncpu, "note": "same batch, static partition, instrumented build"},
            }
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
