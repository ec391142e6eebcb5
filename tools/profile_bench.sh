#!/bin/bash
# rocprofv3 evidence for one bench.py configuration, in separate passes (MI355X_MICROARCH.md, HBM /
# rocprofv3 sections): kernel-trace + stats, then one PMC pass per counter group.  The program goes
# directly after `--` (no env/bash hop).  Writes everything under gpurun_out/prof_<tag>/<name>/ and
# nothing else (only gpurun_out/ travels back from the GPU box); afterwards, in the repo:
#   tools/profile_collect.sh <tag> <name>   -> profiles/<tag>/ + profiles/traffic.json, to be committed.
#   usage: tools/profile_bench.sh <tag> <name> [bench.py arguments...]
#   e.g.   tools/profile_bench.sh r02_v1 human
#          tools/profile_bench.sh r02_v1 c3 --workload c3
set -e -o pipefail
TAG=$1; NAME=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG/$NAME
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
LEAN="--no-cpu-baseline --no-c5 --no-c4 --no-live-pmc --no-sorted --no-variants --parity-sample 20000 --stats-sample 200000"

# PROF_PASSES (optional): which passes to run, by first word -- default "stats FETCH_SIZE WRITE_SIZE TCC_HIT_sum SQ_WAVE_CYCLES"
PASSES=${PROF_PASSES:-stats FETCH_SIZE WRITE_SIZE TCC_HIT_sum SQ_WAVE_CYCLES}
wanted() { case " $PASSES " in *" $1 "*) return 0;; *) return 1;; esac; }

if wanted stats; then
echo "[profile] kernel-trace + stats" >&2
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- python3 "$ROOT/bench.py" "$@" $LEAN --steps 10 --warmup 2 \
    > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.err"
fi

for group in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS"; do
    first=${group%% *}
    wanted "$first" || continue
    echo "[profile] pmc $group" >&2
    rocprofv3 --pmc $group --kernel-include-regex "k_count_kmers" --kernel-trace --output-format csv -d "$OUT/pmc_$first" -o run -- python3 "$ROOT/bench.py" "$@" $LEAN --no-oracle --steps 3 --warmup 1 \
        > "$OUT/bench_pmc_$first.json" 2> "$OUT/pmc_$first.err"
done
# only gpurun_out/ travels back (64 MiB at most): keep the aggregated stats, the counter rows of the
# query kernels, the bench lines and the logs; everything else (per-dispatch traces of the workload
# generator's thousands of launches, databases) stays behind
for f in $(find "$OUT" -name "*counter_collection.csv"); do
    (head -1 "$f"; grep "k_count_kmers" "$f" || true) > "$f.tmp" && mv "$f.tmp" "$f"
done
find "$OUT" -type f ! -name "*kernel_stats.csv" ! -name "*counter_collection.csv" ! -name "*.json" ! -name "*.err" -size +64k -print -delete | sed 's/^/[profile] dropped /' >&2
find "$OUT" -type f -size +2M -print -delete | sed 's/^/[profile] dropped (too big) /' >&2
du -sh "$OUT" >&2
echo "[profile] done: $OUT" >&2
