#!/bin/bash
# The long-running GPU screens DESIGN.md cites, in one place (minutes of GPU time, so they are tools, not tests):
#     gpurun --timeout 3000 -- bash tools/screens.sh [quick]
# Writes gpurun_out/screens/*.txt and prints one PASS / FAIL line per screen; exit code = number of failed screens.
#   gemm_fuzz    randomised launches of the 256-wide GEMM kernels (ragged M / N / K, all layouts and epilogue classes) - bit-identical
#                to the 128x128 kernel
#   attn_fuzz    random attention configurations with per-clip key lengths against fp32 torch (zero gradient on padded keys)
#   fr_stress_*  the race screen of the free-running (tr_mode 12 / 13), wave-specialised (14) and ping-pong (8) kernels: repeated launches, bit for bit
#   train        train-mode steps of config 2 on the fixed synthetic batch (bench.py --steps N --seed 5): the loss must fall below 1.0
# (inputs at the ends of the reference's length filter - 20 s clips, 3-frame clips, one label token - are tests now:
#  tests/test_gpu_r4.py)
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/screens
mkdir -p $O
fail=0
check() {   # name, pattern that the LAST "TOTAL" line must match
  local name=$1 pat=$2 last
  last=$(grep -a "TOTAL" $O/$name.txt | tail -1)
  if echo "$last" | grep -aqE "$pat"; then echo "PASS $name: $last"; else echo "FAIL $name: ${last:-$(tail -2 $O/$name.txt | cut -c1-300)}"; fail=$((fail+1)); fi
}
if [ "${1:-full}" = quick ]; then G=200; A=60; T=60; else G=2000; A=500; T=300; fi
timeout 1500 python3 tools/gpu_gemm_fuzz.py $G 11 > $O/gemm_fuzz.txt 2>&1; check gemm_fuzz " 0 mismatches"
timeout 1500 python3 tools/gpu_attn_fuzz.py $A 12 > $O/attn_fuzz.txt 2>&1; check attn_fuzz "TOTAL bad 0 "
for tr in 12 13 14 8; do
  SMX_DEBUG_TR=$tr timeout 600 python3 tools/gpu_fr_stress.py > $O/fr_stress_$tr.txt 2>&1; check fr_stress_$tr "TOTAL 0$"
done
timeout 900 python3 bench.py --steps $T --warmup 5 --seed 5 --no-cpu-baseline --no-profile --no-eval-leg > $O/train.json 2> $O/train.err
python3 - <<PY
import json, sys
try:
    l = json.loads(open("$O/train.json").read().strip().splitlines()[-1])
    ok = l["final_loss"] < 1.0
    print(("PASS" if ok else "FAIL"), "train: $T steps at", l["ms_per_step"], "ms, final loss", l["final_loss"])
    sys.exit(0 if ok else 1)
except Exception as e:
    print("FAIL train:", e)
    sys.exit(1)
PY
[ $? -eq 0 ] || fail=$((fail+1))
exit $fail
