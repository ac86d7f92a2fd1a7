#!/bin/bash
# captured-step tests, then eager vs graph-replayed bench (same box, seeded), host time per step, kernel-trace timelines
cd "$(dirname "$0")/.."
O=gpurun_out/r5ab
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_r5.py -x -q 2>&1 | tail -25 > $O/tests.log
ARGS="--seed 1 --no-cpu-baseline --no-profile --no-eval-leg"
for i in 1 2; do
  SMX_STEP_GRAPHS=0 timeout 300 python bench.py $ARGS 2>>$O/ab.err >> $O/ab_eager.jsonl
  SMX_STEP_GRAPHS=1 timeout 300 python bench.py $ARGS 2>>$O/ab.err >> $O/ab_graph.jsonl
done
for m in 0 1; do
  SMX_STEP_GRAPHS=$m timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace$m -o t -- python3 bench.py $ARGS --steps 6 > $O/trace$m.log 2>&1
  T=$(find $O/trace$m -name "*kernel_trace.csv" | head -1)
  if [ -n "$T" ]; then
    python3 tools/rocprof_steps.py "$T" 9 > $O/kernel_steps_graphs$m.txt 2>&1
    python3 tools/step_gaps.py "$T" 2 > $O/gaps_graphs$m.txt 2>&1; python3 tools/step_gaps.py "$T" 2 --list > $O/list_graphs$m.txt 2>&1
  fi
  rm -rf $O/trace$m
done
grep -v amdgpu.ids $O/ab.err | tail -5
cat $O/tests.log
cat $O/ab_eager.jsonl $O/ab_graph.jsonl | python -c "
import sys, json
for l in sys.stdin:
    try:
        d = json.loads(l)
        print(d['ms_per_step'], d['host']['enqueue_ms_per_step'], d['host']['step_graphs'], d['host']['graphs_per_step'], d['final_loss'])
    except Exception as e:
        print('ERR', l[:300])
"
head -3 $O/kernel_steps_graphs0.txt $O/kernel_steps_graphs1.txt; head -30 $O/gaps_graphs0.txt $O/gaps_graphs1.txt
