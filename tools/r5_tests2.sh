#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r5t2
mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_r5b.py tests/test_full_dimension_r4.py -x -q -s -m gpu 2>&1 | grep -v "Warning\|warn\|^$" | tail -40 > $O/tests.log
timeout 600 python bench.py --no-cpu-baseline --no-eval-leg > $O/bench.json 2> $O/bench.err
cat $O/tests.log
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5t2/bench.json"))
print(d["ms_per_step"], d.get("trainer_path"), d["host"]["enqueue_ms_per_step"], d["host"].get("trial_fwd_bwd_ms"))
print(json.dumps(d["roofline"])[:1500])
PY
tail -3 $O/bench.err
