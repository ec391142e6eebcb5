#!/bin/bash
# round 5: where do the cycles of the sparse-table kernel go at human scale?  (SQ counters only)
export PROF_PASSES="SQ_WAVE_CYCLES"
bash tools/profile_bench.sh r05_lab human 2>&1 | tail -3
python3 profiles/summarize_pmc.py gpurun_out/prof_r05_lab/human k_count_kmers gpurun_out/prof_r05_lab/human/summary.json gpurun_out/prof_r05_lab/human/bench_pmc_SQ_WAVE_CYCLES.json x || true
cat gpurun_out/prof_r05_lab/human/summary.json 2>/dev/null | head -60
