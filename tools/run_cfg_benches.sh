#!/bin/bash
# Bench lines of BASELINE.json configs 4 and 5 (1 GPU, B = 32, train mode and p = 0), written under gpurun_out/ on the GPU
# box; copied to profiles/r02_bench_cfg{4,5}.jsonl by hand after the run.    gpurun -- bash tools/run_cfg_benches.sh
mkdir -p gpurun_out/r2
for cfg in 4 5; do
  : > gpurun_out/r2/bench_cfg$cfg.jsonl
  for mode in train eval; do
    python tools/gpu_bench_cfg.py $cfg 32 5 $mode 2>/dev/null | tail -1 >> gpurun_out/r2/bench_cfg$cfg.jsonl
  done
  SMX_PAD_FFN=0 python tools/gpu_bench_cfg.py $cfg 32 5 train 2>/dev/null | tail -1 >> gpurun_out/r2/bench_cfg$cfg.jsonl
  cat gpurun_out/r2/bench_cfg$cfg.jsonl
done
