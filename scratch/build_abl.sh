#!/bin/bash
# usage: scratch/build_abl.sh <file-stem> <TAG> [extra -D flags]: builds scratch/libs/libtgnn_<TAG>.so = HEAD with
# csrc/<file-stem>.hip recompiled with -DTGNN_ABL_<TAG> (+ extra flags)
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
stem=$1; tag=$2; shift 2
make -C $REPO/tilingnn_amd/csrc -j8 > /dev/null
d=$REPO/scratch/tmp/abl_$tag; rm -rf $d; mkdir -p $d $REPO/scratch/libs
cp $REPO/build/csrc/*.o $d/
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DTGNN_ABL_$tag "$@" \
    -c $REPO/tilingnn_amd/csrc/$stem.hip -o $d/$stem.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $REPO/scratch/libs/libtgnn_$tag.so $d/*.o -ldl
echo built scratch/libs/libtgnn_$tag.so
