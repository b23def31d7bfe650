#!/bin/bash
# Build an experiment variant of libsfk with extra compiler flags: tools/build_variant.sh NAME -DSFK_NT=1 ...
# -> video-classification_amd/libsfk_NAME.so (objects under build_NAME/); select it at run time with SFK_LIB=<path>.
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
PKG=$ROOT/video-classification_amd
mkdir -p $PKG/build_$NAME
for f in $PKG/csrc/*.hip; do
  o=$PKG/build_$NAME/$(basename ${f%.hip}).o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I$ROOT/include -I$PKG/csrc "$@" -c $f -o $o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $PKG/libsfk_$NAME.so $PKG/build_$NAME/*.o
echo $PKG/libsfk_$NAME.so
