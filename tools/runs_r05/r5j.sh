#!/bin/bash
# round 5, the two modes, one last look: do the slow instances run with fewer WORKING waves (workgroups of the persistent kernel that became
# resident late find no tile left)?  And do the modes exist at 8 waves per CU?
out=$PWD/gpurun_out/r5j; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || exit 1
for w in 12 8; do
  MSBWT_LANES_WAVES_PER_CU=$w timeout -k 10 400 python tools/alloc_probe.py c4r 6 2 auto malloc:0:1 > $out/waves_$w.log 2> $out/waves_$w.err || { tail -5 $out/waves_$w.err; exit 1; }
  echo "== $w waves per CU"; cat $out/waves_$w.log
done
