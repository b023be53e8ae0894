#!/bin/bash
# A/B builds of libisx.so with extra -D flags: tools/build_variant.sh <name> [-DISX_CONV_CHUNK_AB=128 ...]  ->  build_ab/<name>/libisx.so
# (in-tree so that the library travels to the GPU box; select it with ISX_LIB=build_ab/<name>/libisx.so)
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/build_ab/$NAME
rm -rf $OUT && mkdir -p $OUT/a/csrc $OUT/include
cp $ROOT/instance-search_amd/csrc/*.hip $ROOT/instance-search_amd/csrc/*.hpp $ROOT/instance-search_amd/csrc/*.cpp $ROOT/instance-search_amd/csrc/Makefile $OUT/a/csrc/
cp $ROOT/include/isx.h $OUT/include/          # the sources include ../../include/isx.h relative to csrc/
make -s -j8 -C $OUT/a/csrc CXXFLAGS="-O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function $*"
cp $OUT/a/csrc/libisx.so $OUT/libisx.so
rm -rf $OUT/a $OUT/include
echo "built $OUT/libisx.so with: $*"
