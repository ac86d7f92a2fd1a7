#!/bin/bash
# Round-6 profile artefacts, one gpurun call:   gpurun --timeout 2700 -- bash tools/run_profiles_r6.sh
# Writes under gpurun_out/r6prof/ ; the summaries are copied to profiles/r06_* by hand after the run.
# (rocprofv3: the program itself after `--`, never a wrapper; PMC passes are separate runs with --kernel-trace only.
#  The profiled passes run with SMX_STEP_GRAPHS=0: under the profiler the eager host is slower than the GPU, so the `auto` trial
#  would pick the replayed step there and the kernel table would describe another schedule than the bench line's.)
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r6prof
mkdir -p $O
export TMPDIR=/tmp SMX_TUNE_FILE=$PWD/$O/tune.json
# 1. the bench line (also fills the tuner file so that every later pass launches the same kernels)
timeout 900 python3 bench.py > $O/bench.json 2> $O/bench.err
tail -c 600 $O/bench.json
# 1b. the same line with the step replayed from captured graphs (host time, trial numbers)
SMX_STEP_GRAPHS=1 timeout 600 python3 bench.py --no-cpu-baseline --no-profile --no-eval-leg --no-trainer-leg > $O/bench_graphs.json 2>> $O/bench.err
# 2. kernel trace + stats of the same command (timed steps + event-instrumented steps)
SMX_STEP_GRAPHS=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 bench.py --no-cpu-baseline --no-trainer-leg > $O/trace.log 2>&1
T=$(find $O/trace -name "*kernel_trace.csv" | head -1)
S=$(find $O/trace -name "*kernel_stats.csv" | head -1)
[ -n "$T" ] && python3 tools/rocprof_steps.py "$T" 3 > $O/kernel_steps.txt 2>&1
[ -n "$S" ] && cp "$S" $O/kernel_stats.csv
# 2b. train-mode steps only (no eval leg, no instrumented pass) for both schedules: per-step kernel table + GPU idle accounting
for m in 0 1; do
  SMX_STEP_GRAPHS=$m timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace_m$m -o t -- python3 bench.py --seed 1 --no-cpu-baseline --no-profile --no-eval-leg --no-trainer-leg --steps 6 > $O/trace_m$m.log 2>&1
  T2=$(find $O/trace_m$m -name "*kernel_trace.csv" | head -1)
  if [ -n "$T2" ]; then
    python3 tools/rocprof_steps.py "$T2" 9 > $O/train_mode_kernel_steps_graphs$m.txt 2>&1
    python3 tools/step_gaps.py "$T2" 2 > $O/gaps_graphs$m.txt 2>&1
  fi
  rm -rf $O/trace_m$m
done
# 3. host profile + stage-by-stage comparison of the two schedules (no profiler attached)
timeout 600 python3 tools/gpu_host_profile.py 5 > $O/host_profile.txt 2>&1
timeout 600 python3 tools/gpu_stage_compare.py 12 > $O/stage_compare.txt 2>&1
# 4. PMC passes (short run: 2 warm-up + 2 steps, eager)
ARGS="bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-profile --no-eval-leg --no-trainer-leg"
export SMX_STEP_GRAPHS=0
export SMX_GEMM_BYTES_LOG=$PWD/$O/gemm_bytes_log.json      # (algorithmic bytes of every GEMM launch of the FETCH pass, in launch order)
timeout 900 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/pmc_fetch -o p -- python3 $ARGS > $O/pmc_fetch.log 2>&1
unset SMX_GEMM_BYTES_LOG
timeout 900 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/pmc_write -o p -- python3 $ARGS > $O/pmc_write.log 2>&1
timeout 900 rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $O/pmc_sq -o p -- python3 $ARGS > $O/pmc_sq.log 2>&1
F=$(find $O/pmc_fetch -name "*counter_collection.csv" | head -1)
W=$(find $O/pmc_write -name "*counter_collection.csv" | head -1)
Q=$(find $O/pmc_sq -name "*counter_collection.csv" | head -1)
[ -n "$F" ] && [ -n "$W" ] && python3 tools/pmc_traffic.py "$F" "$W" $O/gemm_bytes_log.json > $O/pmc.json 2> $O/pmc.err
[ -s $O/pmc.json ] && python3 -c "
import json,sys,os
p=sys.argv[1]; d=json.load(open(p)); d['tree']='tree '+os.environ.get('SMX_TREE','unknown')+' (round 6)'; json.dump(d,open(p,'w'),indent=1)" $O/pmc.json
[ -n "$Q" ] && python3 tools/pmc_mfma.py "$Q" > $O/pmc_mfma.json 2> $O/pmc_mfma.err
# the raw traces are large: keep the summaries only
rm -rf $O/trace $O/pmc_fetch $O/pmc_write $O/pmc_sq
# 5. round-6 kernels: GEMM timelines (free-running vs wave-specialised), attention phase trace and microbench
tools/lab/build_variant.sh frtrace gemm_fr.hip "-DSMX_FR_TRACE=1 -DSMX_TU=gemm_fr" > /dev/null 2>&1
SMX_LIB=tools/lab/libsmx_frtrace.so SMX_TL_MODES=13:192 timeout 300 python3 tools/gpu_fr_timeline.py > $O/fr_timeline.txt 2>&1
SMX_ATTN_V3=0 timeout 200 python3 tools/gpu_attn_bench.py > $O/attn_v2.txt 2>&1
SMX_ATTN_V3=1 timeout 200 python3 tools/gpu_attn_bench.py > $O/attn_v3.txt 2>&1
timeout 300 python3 tools/gpu_ws_check.py time > $O/ws_time.txt 2>&1
ls -la $O
