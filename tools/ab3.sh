#!/bin/bash
# same-box comparison of N environment settings with warm per-setting tuner files:  bash tools/ab3.sh "A=1 B=2" "A=0" ...   (gpurun_out/ab3/)
cd "$(dirname "$0")/.."
O=gpurun_out/ab3
mkdir -p $O
rm -f $O/*.jsonl $O/tune_*.json
export TMPDIR=/tmp
ARGS="--seed 1 --no-cpu-baseline --no-profile --no-eval-leg --no-trainer-leg --no-fresh-leg --steps 40"
i=0
for cfg in "$@"; do
  env SMX_STEP_GRAPHS=0 SMX_TUNE_FILE=$PWD/$O/tune_$i.json $cfg timeout 400 python bench.py $ARGS --steps 5 > /dev/null 2>>$O/ab.err
  i=$((i+1))
done
for rep in 1 2 3; do
  i=0
  for cfg in "$@"; do
    env SMX_STEP_GRAPHS=0 SMX_TUNE_FILE=$PWD/$O/tune_$i.json $cfg timeout 400 python bench.py $ARGS 2>>$O/ab.err >> $O/run_$i.jsonl
    i=$((i+1))
  done
done
i=0
for cfg in "$@"; do
  python -c "
import sys, json
v = [json.loads(l) for l in open(sys.argv[1]) if l.startswith('{')]
ms = [d['ms_per_step'] for d in v]
print(sys.argv[2], '->', ms, 'median', sorted(ms)[len(ms)//2], 'enc_frac', [d.get('roofline',{}).get('encoder_gemms_frac') for d in v])
" $O/run_$i.jsonl "$cfg"
  i=$((i+1))
done
