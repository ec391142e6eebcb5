#!/bin/bash
# round 5: HBM-side traffic of the sparse-table kernel at human scale (FETCH_SIZE / WRITE_SIZE / TCC hit-miss passes), then an 8-waves-per-CU run
export PROF_PASSES="FETCH_SIZE WRITE_SIZE TCC_HIT_sum"
bash tools/profile_bench.sh r05_lab human 2>&1 | tail -2
python3 - <<'PY'
import csv, glob, collections, json
root="gpurun_out/prof_r05_lab/human"
for f in sorted(glob.glob(root + "/pmc_*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_count_kmers" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(k, len(v), sum(v)/len(v))
for name in ("FETCH_SIZE","WRITE_SIZE","TCC_HIT_sum"):
    try:
        d=json.loads(open(root+"/bench_pmc_%s.json"%name).read().strip().splitlines()[-1]); print(name, "kernel_ms under pmc", d["roofline"]["kernel_ms"])
    except Exception as e: print(name, e)
PY
for w in 8 12; do
MSBWT_LANES_WAVES_PER_CU=$w timeout -k 10 400 python bench.py --no-c5 --no-c4 --no-live-pmc --no-sorted --no-oracle --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r5f_w$w.json 2> gpurun_out/r5f_w$w.err || exit 1
echo "waves/CU $w: $(python -c "import json;d=json.load(open('gpurun_out/r5f_w$w.json'));print(d['value'], d['roofline']['kernel_ms'])")"
done
