#!/bin/bash
# same-box A/B of one environment switch:  bash tools/ab_env.sh VAR OFF_VALUE ON_VALUE [pairs] [extra pytest -k expr]
cd "$(dirname "$0")/.."
VAR=$1; OFF=$2; ON=$3; PAIRS=${4:-3}
O=gpurun_out/r5env_$VAR
mkdir -p $O
export TMPDIR=/tmp
ARGS="--seed 1 --no-cpu-baseline --no-profile --no-eval-leg --no-trainer-leg --no-fresh-leg --steps 40"
for i in $(seq $PAIRS); do
  env SMX_STEP_GRAPHS=0 $VAR=$OFF timeout 300 python bench.py $ARGS 2>>$O/ab.err >> $O/ab_off.jsonl
  env SMX_STEP_GRAPHS=0 $VAR=$ON timeout 300 python bench.py $ARGS 2>>$O/ab.err >> $O/ab_on.jsonl
done
for f in $O/ab_off.jsonl $O/ab_on.jsonl; do python -c "
import sys, json
v = [json.loads(l)['ms_per_step'] for l in open(sys.argv[1])]
print(sys.argv[1], v, 'median', sorted(v)[len(v)//2], 'loss', [json.loads(l)['final_loss'] for l in open(sys.argv[1])])
" $f; done
