#!/bin/bash
# full GPU test suite + alternating eager / graph A/B (4 pairs x 40 steps)
cd "$(dirname "$0")/.."
O=gpurun_out/r5full
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -30 > $O/tests.log
ARGS="--seed 1 --no-cpu-baseline --no-profile --no-eval-leg --steps 40"
for i in 1 2 3 4; do
  SMX_STEP_GRAPHS=0 timeout 300 python bench.py $ARGS 2>>$O/ab.err >> $O/ab_eager.jsonl
  SMX_STEP_GRAPHS=1 timeout 300 python bench.py $ARGS 2>>$O/ab.err >> $O/ab_graph.jsonl
done
tail -15 $O/tests.log
for f in $O/ab_eager.jsonl $O/ab_graph.jsonl; do python -c "
import sys, json
print(sys.argv[1], [json.loads(l)['ms_per_step'] for l in open(sys.argv[1])])
" $f; done
