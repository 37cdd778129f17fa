#!/bin/bash
# Dev: build a variant of libmassfuse.so with extra compiler flags for fuse.hip (kernel experiments):
#   tools/dev/build_variant.sh NAME -DCELLS_EB_DEF=1 ...   ->  tools/dev/variants/libmassfuse_NAME.so
# Run a program against it with MASSFUSE_LIB=tools/dev/variants/libmassfuse_NAME.so.
set -e
name=$1; shift
cd "$(dirname "$0")/../../mass_amd/csrc"
make -s -j4 >/dev/null
mkdir -p ../../tools/dev/variants
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -munsafe-fp-atomics -fvisibility=hidden -Wall -Wno-unused-function -I../../include -I."
/opt/rocm/bin/hipcc $FLAGS "$@" -c fuse.hip -o /tmp/fuse_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/dev/variants/libmassfuse_$name.so /tmp/fuse_$name.o match.o reduce.o lsa.o contours.o api.o
echo built tools/dev/variants/libmassfuse_$name.so
