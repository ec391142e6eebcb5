#!/bin/bash
# round-3 GPU call 17: MSBWT_LANES_MIN_BUSY variants on C3 fused and the human line
for V in default tools/_variants/minbusy56.so tools/_variants/minbusy48.so; do
  echo "== $V: c3 fused"; tools/sweep_variants.sh "--workload c3 --fused --no-live-pmc --no-sorted" $V 2>/dev/null | grep -E "q/s"
  echo "== $V: human";    tools/sweep_variants.sh "--no-c4 --no-live-pmc --no-sorted" $V 2>/dev/null | grep -E "q/s"
done
