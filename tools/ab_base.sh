#!/bin/bash
# Same-box A/B support: export commit $1 (default HEAD) into .ab_base/ and build its library there (in the build container).
# A later `gpurun -- 'python3 .ab_base/bench.py ...; python3 bench.py ...'` then times both trees on ONE box - boxes of the pool
# differ by several per cent, so only same-box pairs mean anything.  .ab_base/ is git-ignored and travels with the snapshot.
set -eu
cd "$(dirname "$0")/.."
rev=${1:-HEAD}
rm -rf .ab_base && mkdir .ab_base
git archive "$rev" | tar -x -C .ab_base
rm -rf .ab_base/tests/golden .ab_base/profiles .ab_base/gpurun_out
python3 .ab_base/speechmix_amd/csrc/build.py > /dev/null
echo "exported $(git rev-parse --short "$rev") to .ab_base/"
