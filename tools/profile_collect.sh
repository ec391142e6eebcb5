#!/bin/bash
# Turns the raw rocprofv3 output of tools/profile_bench.sh (gpurun_out/prof_<tag>/<name>/) into the
# committed summaries: profiles/<tag>/kernel_stats_<name>.csv, bench_<name>_under_rocprof.json,
# pmc_summary_<name>.json, and the profiles/traffic.json entry bench.py reads.
#   usage: tools/profile_collect.sh <tag> <name> [kernel substring]
set -e -o pipefail
TAG=$1; NAME=$2; KERNEL=${3:-k_count_kmers}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_$TAG/$NAME
DST=$ROOT/profiles/$TAG
mkdir -p "$DST"
cp "$(find "$OUT/stats" -name "*kernel_stats.csv" | head -1)" "$DST/kernel_stats_$NAME.csv"
cp "$OUT/bench_under_rocprof.json" "$DST/bench_${NAME}_under_rocprof.json"
python3 "$ROOT/profiles/summarize_pmc.py" "$OUT" "$KERNEL" "$DST/pmc_summary_$NAME.json" "$OUT/bench_pmc_FETCH_SIZE.json" "profiles/$TAG/pmc_summary_$NAME.json"
