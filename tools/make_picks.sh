#!/bin/bash
# Regenerate the shipped kernel picks on an MI355X:   gpurun --timeout 1500 -- bash tools/make_picks.sh
# Live-tunes (medians of interleaved launches, speechmix_amd/ops.py measure_candidates) every GEMM key of the benchmarked
# configurations - config 2 through bench.py, configs 4 / 5 through tools/gpu_bench_cfg.py - and writes the union to
# gpurun_out/picks/mi355x_picks.json; copy it to speechmix_amd/tune/mi355x_picks.json and commit.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/picks
mkdir -p $O
export SMX_TUNE=live SMX_TUNE_FILE=$PWD/$O/mi355x_picks.json
rm -f $SMX_TUNE_FILE
timeout 600 python3 bench.py --no-cpu-baseline --no-eval-leg > $O/bench_cfg2.json 2> $O/bench_cfg2.err
timeout 600 python3 bench.py --no-cpu-baseline --no-profile --eval-mode --steps 3 --warmup 3 > $O/bench_cfg2_eval.json 2>> $O/bench_cfg2.err
# the backward of a data-parallel run leaves 40 CUs to RCCL (ops.PP_BACKWARD_CUS = 216): its keys carry that reserve
SMX_PP_BACKWARD_CUS=216 timeout 600 python3 bench.py --no-cpu-baseline --no-profile --no-eval-leg --steps 3 --warmup 3 > $O/bench_cfg2_cus216.json 2>> $O/bench_cfg2.err
for c in 4 5; do
  timeout 900 python3 tools/gpu_bench_cfg.py $c > $O/bench_cfg$c.txt 2>&1
done
python3 - <<'PY'
import json, os
p = os.environ["SMX_TUNE_FILE"]
print(len(json.load(open(p))), "picks in", p)
PY
