"""bench.py end to end at a tiny scale (guards the measurement path; needs the GPU)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(args):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                         timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "bench.py must print exactly one JSON line"
    return json.loads(lines[0])


@pytest.mark.parametrize("extra", [["--workload", "c2"], ["--workload", "c3"], ["--workload", "c3", "--fused"],
                                   ["--workload", "big", "--big-symbols", "3000000", "--queries", "20000"]])
def test_bench_contract(extra):
    r = _run(extra + ["--scale", "0.003", "--steps", "2", "--warmup", "1", "--cpu-sample", "2000",
                      "--parity-sample", "5000"])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "parity"):
        assert key in r, key
    assert r["n_gpus"] == 1 and r["steps"] == 2 and r["warmup"] == 1 and r["vs_baseline"] is None
    assert r["parity"]["mismatches"] == 0 and r["parity"]["checked"] > 0
    assert r["value"] > 0 and r["roofline"]["achieved"] > 0 and r["roofline"]["bound"] == "hbm"
    assert abs(r["roofline"]["frac"] - r["roofline"]["achieved"] / r["roofline"]["peak"]) < 1e-9
    assert r["cpu_baseline"]["kind"] == "port" and r["cpu_baseline"]["cores"] == 1
    assert "workload" in r["config"] and "model" not in r["config"]


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
           "--scale", "0.003", "--steps", "3", "--warmup", "1", "--parity-sample", "5000", "--payload", payload]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 3 and r["scaling"] == "weak" and r["parity"]["mismatches"] == 0
    assert "cpu_baseline" not in r            # reported at N=1 only
    assert r["value"] > 0 and "x2" in r["config"]["parallelism"]
    assert ("int16" if payload == "auto" else "int64") in r["config"]["parallelism"]
