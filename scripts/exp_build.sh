#!/bin/bash
# A/B build of the library with extra -D flags on some sources (timing experiments; the product build is _native.build()):
#   bash scripts/exp_build.sh "-DTSGNN_EXP_X=1" layer_fwd rowgemm   ->  scripts/_build/libtsgnn_exp.so   (use with TSGNN_LIB_PATH)
set -e
R=$(cd "$(dirname "$0")/.." && pwd); D=/tmp/tsgnn_expb; mkdir -p $D "$R/scripts/_build"
DEFS="$1"; shift
OBJS=""
for o in "$R"/two-stage-gnn_amd/build/*.o; do b=$(basename $o .o); skip=0; for f in "$@"; do [ "$f" = "$b" ] && skip=1; done; [ $skip = 0 ] && OBJS="$OBJS $o"; done
for f in "$@"; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $DEFS -c "$R/two-stage-gnn_amd/csrc/$f.hip" -o $D/$f.o & done; wait
for f in "$@"; do OBJS="$OBJS $D/$f.o"; done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$R/scripts/_build/libtsgnn_exp.so" $OBJS
echo "built scripts/_build/libtsgnn_exp.so with $DEFS on: $*"
