#!/bin/bash
# c4r ran bimodal between PROCESSES on one box (18.0 / 20.3 ms, r4ag): is it where the index lands in HBM?  The same line four times with
# torch's caching allocator as it is, four times with it off (every freed torch block goes back to the driver before the index is built)
out=$PWD/gpurun_out/r4ak; mkdir -p $out
line() { python -c "import json,sys;d=json.load(open(sys.argv[1]));print('%.4g q/s  %.3f ms/step kernel %.3f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))" $1; }
for rep in 1 2 3 4; do
  timeout -k 10 400 python bench.py --workload c4r --no-oracle --no-c5 --no-cpu-baseline --steps 10 --warmup 2 > $out/c4r_cache_$rep.json 2> $out/c4r_cache_$rep.err || exit 1
  echo "c4r caching rep$rep $(line $out/c4r_cache_$rep.json)"
  PYTORCH_NO_CUDA_MEMORY_CACHING=1 timeout -k 10 400 python bench.py --workload c4r --no-oracle --no-c5 --no-cpu-baseline --steps 10 --warmup 2 > $out/c4r_nocache_$rep.json 2> $out/c4r_nocache_$rep.err || exit 1
  echo "c4r no-caching rep$rep $(line $out/c4r_nocache_$rep.json)"
done
