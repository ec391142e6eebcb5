"""bench.py end to end at a tiny scale (guards the measurement path; needs the GPU)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


COMPACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config")


def check_compact_line(stdout, extras_path):
    """stdout carries ONE line: json.loads-able, under 8 KB (round 5's 25 KB line could not be parsed by the driver), with the contract's
    keys; the full record is in the extras file.  -> (compact, full)"""
    lines = [l for l in stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "bench.py must print exactly one JSON line: %r" % stdout[:500]
    assert len(lines[0].encode()) < 8192, "the stdout line has %d bytes" % len(lines[0].encode())
    compact = json.loads(lines[-1])
    for key in COMPACT_KEYS:
        assert key in compact, key
    assert "workload" in compact["config"] and "model" not in compact["config"] and len(compact["config"]["workload"]) <= 420
    assert compact["extras_file"] == os.path.basename(extras_path)
    full = json.load(open(extras_path))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "scaling"):
        assert compact[key] == full[key], key
    if "roofline" in full and "frac" in full["roofline"]:
        assert "frac" in compact["roofline"] and compact["roofline"]["frac"] == full["roofline"]["frac"]
        assert compact["roofline"]["kernel_ms"] == full["roofline"]["kernel_ms"] and "layout_bytes_per_query" in compact["roofline"]
    if "cpu_baseline" in full:
        assert compact["cpu_baseline"]["value"] == full["cpu_baseline"]["value"] and compact["cpu_baseline"]["cores"] == 1
    if "parity" in full:
        assert compact["parity"]["mismatches"] == full["parity"]["mismatches"]
    return compact, full


def _run(args, tmp=[0]):
    tmp[0] += 1
    extras = os.path.join(ROOT, "gpurun_out", "bench_extras_test_%d_%d.json" % (os.getpid(), tmp[0]))
    os.makedirs(os.path.dirname(extras), exist_ok=True)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--extras-file", extras] + args, capture_output=True, text=True,
                         timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    compact, full = check_compact_line(out.stdout, extras)
    os.remove(extras)
    full["_compact"] = compact
    return full


@pytest.mark.parametrize("extra", [["--scale", "0.00003"],  # the default workload (human-scale stand-in), shrunk
                                   ["--workload", "c2", "--scale", "0.003"], ["--workload", "c3", "--scale", "0.003"],
                                   ["--workload", "c3", "--fused", "--scale", "0.003"],
                                   ["--workload", "big", "--big-symbols", "3000000", "--queries", "20000"]])
def test_bench_contract(extra):
    r = _run(extra + ["--steps", "2", "--warmup", "1", "--cpu-sample", "2000", "--parity-sample", "5000"])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "parity"):
        assert key in r, key
    assert r["n_gpus"] == 1 and r["steps"] == 2 and r["warmup"] == 1 and r["vs_baseline"] is None
    assert r["parity"]["mismatches"] == 0 and r["parity"]["checked"] > 0
    roof = r["roofline"]
    assert r["value"] > 0 and roof["bound"] == "hbm" and roof["kernel_ms"] > 0
    # frac is counter traffic / kernel time / peak, or null when no PMC summary matches this (shrunk) configuration
    assert roof["frac"] is None or 0 < roof["frac"] <= 1.0
    assert roof["reference_algorithm"]["bytes_per_launch"] > 0 and roof["reference_algorithm"]["GBps"] > 0 and roof["layout_algorithmic"]["bytes_per_query"] > 0
    assert r["cpu_baseline"]["kind"] == "port" and r["cpu_baseline"]["cores"] == 1
    assert "workload" in r["config"] and "model" not in r["config"]


def test_bench_default_run_carries_the_real_msbwt_line():
    """The default run's extra key c4_real_reads (the REAL MSBWT of config C4, read-derived 31-mers), here on a
    shrunk C4 beside a shrunk human-scale index: own value, roofline, parity."""
    r = _run(["--scale", "0.00003", "--c4-scale", "0.002", "--c4-lab", "--steps", "2", "--warmup", "1", "--cpu-sample", "2000", "--parity-sample", "5000"])
    compact = r["_compact"]
    assert compact["roofline"]["kernel_ms"] > 0 and compact["cpu_baseline"]["value"] > 0 and compact["parity"]["mismatches"] == 0
    for key in ("c4_real_reads_qps", "c4_repeats_qps", "undeclared_k_qps", "headline_sparse_off_qps"):
        assert compact["extras"][key] > 0, key
    assert compact["extras"]["c4_real_reads_mismatches"] == 0 and compact["extras"]["undeclared_k_counts_equal_headline"] is True
    assert compact["extras"]["headline_sparse_off_counts_equal_headline"] is True and compact["extras"]["headline_sparse_off_sparse_table_depth"] == 0
    for rec in (r["short_k"]["k17"], r["short_k"]["k19"], r["short_k"]["k21"]):
        assert rec["default_qps"] > 0 and rec["sparse_off_qps"] > 0 and rec["counts_equal"] is True
    assert r["short_k"]["parity"]["mismatches"] == 0 and r["short_k"]["parity"]["checked"] > 0
    # round 6: the C4 line's modes (complete / two-tier / budgeted / no sparse table) and the host-pointer path, all counting like the line
    modes = r["c4_budgeted"]["modes"]
    assert set(modes) == {"complete", "two_tier", "two_tier_budgeted", "fallback"} and all(m["counts_equal_the_line"] and m["value"] > 0 for m in modes.values())
    assert modes["fallback"]["sparse_table_depth"] == 0 and compact["extras"]["c4_budgeted_two_tier_qps"] == modes["two_tier"]["value"]
    host = r["host_api"]
    assert "error" not in host and host["bytes_equal"] and host["packed_u64_equal"] and host["packed_u32_equal"] and host["pcie_both_ways_GBps"] > 0
    assert compact["extras"]["host_api_bytes_qps"] == host["bytes_qps"] > 0
    c4 = r["c4_real_reads"]
    assert c4["value"] > 0 and c4["parity"]["mismatches"] == 0 and c4["parity"]["checked"] > 0
    assert c4["parity"]["mean_count_in_sample"] > 5      # a real 30x read set: present k-mers occur ~ coverage times
    assert c4["roofline"]["kernel_ms"] > 0 and c4["config"]["k"] == 31 and "REAL" in c4["config"]["workload"]
    assert "EXACT multi-string BWT" in r["config"]["workload"] and r["parity"]["mismatches"] == 0
    assert r["config"]["typical_range_width"] > 8        # ... and so does the human-scale index's read set
    # round 4: the repeat-bearing C4 line, with the kernel's own counters and the library's batch order before / after
    rep = r["c4_repeats"]
    assert rep["value"] > 0 and rep["parity"]["mismatches"] == 0 and "repeat-bearing" in rep["config"]["workload"]
    counters = rep["search_counters"]
    assert counters["lines_per_query"] > 1 and 0 <= counters["escape_query_fraction"] <= 1 and 0 <= counters["second_line_rate"] <= 1
    assert rep["parity"]["max_count_in_sample"] > 4 * rep["parity"]["mean_count_in_sample"]   # repeats: some k-mers occur far more often than the coverage
    assert not rep["batch_ordered_by_the_library"] and rep["library_ordered"]["counts_equal_unordered_run"] and rep["library_ordered"]["value"] > 0
    assert r["roofline"]["layout_algorithmic"]["lines_per_query"] > 1 and "telemetry" in r
    # ... and which of the line's two modes the build is in: the pair blocks rebuilt, the same batch timed again
    for key in ("c4_repeats", "c4_real_reads"):
        again = r[key]["pair_blocks_rebuilt"]
        assert "error" not in again and again["counts_equal_first_run"] and again["ms_per_step"] > 0 and again["ratio_to_the_line"] > 0


def test_bench_default_is_the_metric_config():
    """The driver runs `python3 bench.py --gpus 1 --steps K --warmup W`: that must be k = 31 on the
    human-scale index (checked on the argument defaults; the full-size run is the driver's)."""
    sys.path.insert(0, ROOT)
    import bench
    a = bench.parse_args([])
    assert a.workload == "human" and a.gpus == 1 and bench.HUMAN_SYMBOLS == 9e10
    import synth
    assert synth.CONFIGS["c3"]["k"] == 31  # the k the human workload inherits


@pytest.mark.parametrize("scaling", ["strong", "weak"])
def test_bench_self_launch_two_ranks(scaling):
    """`python3 bench.py --gpus 2` launched plainly must start its own ranks (child process) and
    print one JSON line with n_gpus = 2."""
    r = _run(["--gpus", "2", "--dist-backend", "gloo", "--scale", "0.00003", "--steps", "2", "--warmup", "1",
              "--parity-sample", "5000", "--scaling", scaling])
    assert r["n_gpus"] == 2 and r["scaling"] == scaling and r["parity"]["mismatches"] == 0 and r["value"] > 0
    assert r["config"]["k"] == 31
    if scaling == "strong":
        assert r["weak_scaling"]["value"] > 0
        assert r["config"]["queries_per_gpu"] * 2 >= r["config"]["queries_per_step"]


def test_bench_two_ranks_sharded_random_line():
    """The literal configs[4] shape at N = 2 (gloo rehearsal on the shared GPU): the device-generated random 31-mers are
    defined chunk by chunk, each rank generates only its shard, all counts are gathered; plus the per-rank timing block."""
    r = _run(["--gpus", "2", "--dist-backend", "gloo", "--scale", "0.00003", "--steps", "2", "--warmup", "1", "--parity-sample", "5000",
              "--c5-queries", "300000", "--no-weak"])
    c5 = r["c5_random_1e9"]
    assert c5["queries"] == 300000 and c5["queries_per_gpu"] in (150000, 150016) and c5["parity"]["mismatches"] == 0 and c5["value"] > 0
    assert "sharded over 2 ranks" in c5["kind"]
    ranks = r["ranks"]
    assert len(ranks["kernel_ms"]) == 2 and ranks["kernel_ms_min"] > 0 and ranks["exchange_ms_alone"] > 0


@pytest.mark.parametrize("payload", ["auto", "int64"])
def test_bench_two_ranks_rehearsal(payload):
    """The N>1 code path of bench.py (barriers, per-rank batches, gather of all counts, max over
    ranks) with two ranks sharing this box's one GPU and gloo carrying the collective -- RCCL
    itself needs one GPU per rank and is exercised by the driver's multi-GPU runs."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo",
           "--workload", "c2", "--scale", "0.003", "--steps", "3", "--warmup", "1", "--parity-sample", "5000", "--payload", payload,
           "--scaling", "weak", "--exchange", "gather" if payload == "auto" else "allgather"]   # (both forms of the step's exchange)
    extras = os.path.join(ROOT, "gpurun_out", "bench_extras_test_ranks_%d_%s.json" % (os.getpid(), payload))
    os.makedirs(os.path.dirname(extras), exist_ok=True)
    out = subprocess.run(cmd + ["--extras-file", extras], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    compact, r = check_compact_line(out.stdout, extras)
    os.remove(extras)
    assert "ranks_kernel_ms_max" in compact["extras"] and compact["n_gpus"] == 2
    assert r["n_gpus"] == 2 and r["steps"] == 3 and r["scaling"] == "weak" and r["parity"]["mismatches"] == 0
    assert "cpu_baseline" not in r            # reported at N=1 only
    assert r["value"] > 0 and "x2" in r["config"]["parallelism"]
    assert ("int16" if payload == "auto" else "int64") in r["config"]["parallelism"]
    assert ("gather to rank 0" if payload == "auto" else "all_gather") in r["config"]["parallelism"]


@pytest.mark.parametrize("payload", ["auto", "int64"])
def test_bench_rccl_calls_on_one_rank(payload):
    """The N>1 code path on REAL RCCL (nccl backend), with the only communicator this box allows: one
    rank.  Process-group init with a device id, the asynchronous all_gather_into_tensor of int16/int64
    counts shipped as raw bytes, its overlap with the next step, widening, the float64 all_reduce of the
    timing, barriers -- everything but a second GPU on the other end of xGMI."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    extras = os.path.join(ROOT, "gpurun_out", "bench_extras_test_rccl_%d_%s.json" % (os.getpid(), payload))
    os.makedirs(os.path.dirname(extras), exist_ok=True)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--dist-backend", "nccl", "--scale", "0.00003",
                          "--steps", "4", "--warmup", "1", "--parity-sample", "5000", "--payload", payload, "--no-c5", "--extras-file", extras,
                          "--exchange", "gather" if payload == "auto" else "allgather"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    compact, r = check_compact_line(out.stdout, extras)   # (stdout carries the JSON line only: RCCL's banner belongs on stderr)
    os.remove(extras)
    for key in ("native_gather_qps", "weak_scaling_qps", "native_gather_single_batch_latency_ms", "native_gather_single_batch_pipelined_ms", "ranks_kernel_ms_max"):
        assert compact["extras"][key] > 0, key
    assert r["n_gpus"] == 1 and r["scaling"] == "strong" and r["parity"]["mismatches"] == 0 and r["value"] > 0
    assert "RCCL" in r["config"]["parallelism"] and ("int16" if payload == "auto" else "int64") in r["config"]["parallelism"]
    assert r["weak_scaling"]["value"] > 0
    # the library's own RCCL call site, and what a caller with ONE batch sees: serial against the pipelined form (pieces counted while
    # earlier pieces travel), counts widened or left narrow at the destination -- all equal to the torch path's vector
    ng = r["native_gather"]
    assert "error" not in ng and ng["equals_torch_path"], ng
    assert "single_batch_error" not in ng and ng["single_batch_latency_ms"] > 0 and ng["single_batch_pipelined_ms"] > 0, ng
    assert ng["single_batch_pipelined_equals_torch_path"] is True and ng["single_batch_narrow_destination_equals_torch_path"] is True, ng
