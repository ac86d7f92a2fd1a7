#!/bin/bash
# A/B build of one kernel file: tools/lab/build_variant.sh NAME FILE.hip "-DX=1 -DY=2"  ->  tools/lab/libsmx_NAME.so (use with SMX_LIB=...)
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"; CS="$HERE/../../speechmix_amd/csrc"
NAME="$1"; FILE="$2"; DEFS="$3"
mkdir -p "$HERE/_obj"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-result $DEFS -I "$CS" -c "$CS/$FILE" -o "$HERE/_obj/${NAME}_${FILE%.hip}.o"
OBJS=""
for f in "$CS"/_obj/*.o; do b="$(basename "$f")"; if [ "$b" != "${FILE%.hip}.o" ]; then OBJS="$OBJS $f"; fi; done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$HERE/libsmx_$NAME.so" $OBJS "$HERE/_obj/${NAME}_${FILE%.hip}.o"
echo "built $HERE/libsmx_$NAME.so"
