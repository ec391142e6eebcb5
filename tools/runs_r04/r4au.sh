#!/bin/bash
# c4r's two modes under the L2 counters: three instances side by side per process; up to three processes, until one holds a fast AND a slow instance
out=$PWD/gpurun_out/r4au; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
ROOT=$GRAFT_REPO_ROOT
for try in 1 2 3; do
  timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-include-regex "k_count_kmers" --kernel-trace --output-format csv -d $out/pmc_$try -o run -- python3 $ROOT/tools/instance_probe.py c4r 3 1 1 > $out/inst_$try.log 2> $out/inst_$try.err || { tail -5 $out/inst_$try.err; exit 1; }
  echo "process $try"; cat $out/inst_$try.log
  if python3 -c "
import re,sys
t=[float(l.split(':')[1].split()[2]) for l in open('$out/inst_$try.log')]
sys.exit(0 if min(t) < 19.3 and max(t) > 20.0 else 1)"; then echo "both modes in process $try"; break; fi
done
for f in $(find $out -name "*counter_collection.csv"); do (head -1 "$f"; grep "k_count_kmers" "$f" || true) > "$f.tmp" && mv "$f.tmp" "$f"; done
for f in $(find $out -name "*kernel_trace.csv"); do (head -1 "$f"; grep "k_count_kmers" "$f" || true) > "$f.tmp" && mv "$f.tmp" "$f"; done
find $out -type f -size +2M -print -delete
