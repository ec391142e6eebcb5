#!/bin/bash
# c4r's two modes under the counters: three instances side by side (tools/instance_probe.py), per-dispatch L2 hits / misses, fetch size and wave
# cycles beside each dispatch's duration (kernel trace of the same pass) -- separate passes per counter group
out=$PWD/gpurun_out/r4at; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
ROOT=$GRAFT_REPO_ROOT
for group in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS"; do
  first=${group%% *}
  timeout -k 10 400 rocprofv3 --pmc $group --kernel-include-regex "k_count_kmers" --kernel-trace --output-format csv -d $out/pmc_$first -o run -- python3 $ROOT/tools/instance_probe.py c4r 3 2 1 > $out/inst_$first.log 2> $out/inst_$first.err || { tail -5 $out/inst_$first.err; exit 1; }
  echo "pass $first"; cat $out/inst_$first.log
done
for f in $(find $out -name "*counter_collection.csv"); do (head -1 "$f"; grep "k_count_kmers" "$f" || true) > "$f.tmp" && mv "$f.tmp" "$f"; done
for f in $(find $out -name "*kernel_trace.csv"); do (head -1 "$f"; grep "k_count_kmers" "$f" || true) > "$f.tmp" && mv "$f.tmp" "$f"; done
find $out -type f -size +2M -print -delete
du -sh $out
