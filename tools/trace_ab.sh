#!/bin/bash
# kernel tables of two settings of one environment switch on ONE box:  bash tools/trace_ab.sh VAR OFF ON [steps]
# -> gpurun_out/trace_ab_VAR/{off,on}_kernel_steps.txt  (rocprofv3 --kernel-trace of bench.py, eager steps, per-step table by tools/rocprof_steps.py)
cd "$(dirname "$0")/.."
VAR=$1; OFF=$2; ON=$3; STEPS=${4:-6}
O=gpurun_out/trace_ab_$VAR
mkdir -p $O
export TMPDIR=/tmp SMX_STEP_GRAPHS=0
for side in off on; do
  if [ $side = off ]; then export $VAR=$OFF; else export $VAR=$ON; fi
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/tr_$side -o t -- python3 bench.py --seed 1 --no-cpu-baseline --no-profile --no-eval-leg --no-trainer-leg --no-fresh-leg --steps $STEPS > $O/$side.log 2>&1
  T=$(find $O/tr_$side -name "*kernel_trace.csv" | head -1)
  [ -n "$T" ] && python3 tools/rocprof_steps.py "$T" 9 > $O/${side}_kernel_steps.txt 2>&1
  rm -rf $O/tr_$side
  head -1 $O/${side}_kernel_steps.txt
done
