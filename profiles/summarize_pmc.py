#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (one counter group per pass) for one kernel and register the
result in profiles/traffic.json.

usage: summarize_pmc.py <dir with pmc_*/...counter_collection.csv> <kernel substring> <out.json>
                        <bench line of one of the PMC passes> <path of out.json relative to the repo root>
Applies the gfx950 correction of MI355X_MICROARCH.md (HBM section): FETCH_SIZE is in KiB and
tallies 128-byte requests at 64 bytes, so read bytes = 2 x FETCH_SIZE x 1024.  FETCH_SIZE counts the
L2's memory-side requests, i.e. it includes reads served by the Infinity Cache; for an index far
larger than 256 MiB that share is negligible and the figure is HBM traffic.
The traffic.json entry carries the kernel-source stamp of the tree it was measured with
(bench.kernel_stamp()); bench.py refuses entries whose stamp differs from the sources it runs.
"""
import collections
import csv
import glob
import json
import os
import sys

root, kernel, out_path, bench_line, rel_path = sys.argv[1:6]
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
out = {}
# only the launches of the bench line's MAIN measurement count: the first warmup + steps dispatches of the kernel (a
# default run goes on to other lines -- the C4 extra, the 1e9-query line -- that may use the same kernel instantiation)
main = json.loads(open(bench_line).read().strip().splitlines()[-1])
n_main = int(main["steps"]) + int(main["warmup"])
for f in sorted(glob.glob(root + "/pmc_*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(list)
    for r in sorted((r for r in csv.DictReader(open(f)) if kernel in r["Kernel_Name"]), key=lambda r: int(r["Dispatch_Id"])):
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        v = v[:n_main]
        out[k] = {"launches": len(v), "mean_per_launch": sum(v) / len(v)}
d = {}
if "FETCH_SIZE" in out:
    d["fetch_bytes_raw"] = out["FETCH_SIZE"]["mean_per_launch"] * 1024
    d["fetch_bytes_corrected_x2_gfx950"] = 2 * d["fetch_bytes_raw"]
if "WRITE_SIZE" in out:
    d["write_bytes"] = out["WRITE_SIZE"]["mean_per_launch"] * 1024
if "FETCH_SIZE" in out and "WRITE_SIZE" in out:
    d["traffic_bytes_per_launch"] = d["fetch_bytes_corrected_x2_gfx950"] + d["write_bytes"]
if "TCC_HIT_sum" in out and "TCC_MISS_sum" in out:
    hit, miss = out["TCC_HIT_sum"]["mean_per_launch"], out["TCC_MISS_sum"]["mean_per_launch"]
    d["l2_hit_rate"] = hit / (hit + miss)
    d["tcc_miss_x128_bytes"] = miss * 128
if "SQ_WAVE_CYCLES" in out:
    wc = out["SQ_WAVE_CYCLES"]["mean_per_launch"]
    for name in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
        if name in out:
            d[name.lower() + "_frac_of_wave_cycles"] = out[name]["mean_per_launch"] / wc
bench = json.loads(open(bench_line).read().strip().splitlines()[-1])
cfg = bench["config"]
nq = cfg["queries_per_gpu"]
for name in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS"):
    if name in out:
        d[name.lower() + "_per_query"] = out[name]["mean_per_launch"] / nq
d["kernel"] = kernel
d["bench_config"] = cfg
out["_derived"] = d
json.dump(out, open(out_path, "w"), indent=1)
print(json.dumps(d, indent=1))

if "traffic_bytes_per_launch" in d:
    import bench as bench_mod
    workload = cfg["workload"].split(":", 1)[0] + ("+repeats" if "REPEAT-BEARING" in cfg["workload"] else "")   # (bench.py --genome repeats)
    kind = cfg.get("query_kind") or ("walk" if "LF-walk" in cfg["workload"] else "reads" if "read-derived" in cfg["workload"] else "random")
    entry = {
        "workload": workload, "k": cfg["k"], "table_depth": cfg["table_depth"], "pair_index": cfg["pair_index"], "pair_stride": cfg.get("pair_stride"),
        "query_kind": kind, "fused": "prepared in-kernel" in cfg["workload"], "two_tier": cfg.get("sparse_table_tiers") == 2, "bwt_symbols": cfg["bwt_symbols"], "queries_per_launch": nq,
        "traffic_bytes_per_launch": d["traffic_bytes_per_launch"], "traffic_bytes_per_query": d["traffic_bytes_per_launch"] / nq,
        "l2_hit_rate": d.get("l2_hit_rate"), "kernel_stamp": bench_mod.kernel_stamp(), "source": rel_path,
    }
    tj = os.path.join(REPO, "profiles", "traffic.json")
    doc = json.load(open(tj))
    key = lambda e: (e["workload"], e["k"], e["table_depth"], e.get("pair_index"), e.get("pair_stride"), e.get("query_kind"), e.get("bwt_symbols"), bool(e.get("fused", False)),
                     bool(e.get("two_tier", False)))
    doc["entries"] = [e for e in doc["entries"] if key(e) != key(entry)] + [entry]
    json.dump(doc, open(tj, "w"), indent=1)
    print("registered in profiles/traffic.json:", json.dumps(entry))
