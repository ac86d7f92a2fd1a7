#!/bin/bash
# Bench lines of BASELINE.json configs 4 and 5 (1 GPU, B = 32, train mode and p = 0), written under gpurun_out/ on the GPU
# box; copied to profiles/rNN_bench_cfg{4,5}.jsonl by hand after the run.    gpurun -- bash tools/run_cfg_benches.sh
# Config 4 additionally with the LM head streamed in row chunks (SMX_HEAD_STREAM=1: the [B*L, 250 054] fp32 logits and
# their gradient are never materialised) and with it forced off (=0); the default is the engine's size-based choice.
O=gpurun_out/cfgbench
mkdir -p $O
for cfg in 4 5; do
  : > $O/bench_cfg$cfg.jsonl
  for mode in train eval; do
    python tools/gpu_bench_cfg.py $cfg 32 5 $mode 2>/dev/null | tail -1 >> $O/bench_cfg$cfg.jsonl
  done
  if [ $cfg = 4 ]; then
    for hs in 0 1; do
      SMX_HEAD_STREAM=$hs python tools/gpu_bench_cfg.py 4 32 5 train 2>/dev/null | tail -1 | sed "s/^{/{\"SMX_HEAD_STREAM\": $hs, /" >> $O/bench_cfg$cfg.jsonl
    done
  fi
  cat $O/bench_cfg$cfg.jsonl
done
