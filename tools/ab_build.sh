#!/bin/bash
# Builds a second libpcx.so with extra compiler flags for same-box A/B runs:
#   tools/ab_build.sh noepi -DPCX_NO_EPILOGUE   ->  gpurun_variants/libpcx_noepi.so
# Select it at run time with PCX_LIB=gpurun_variants/libpcx_<tag>.so.
set -e
TAG=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
cp $ROOT/pycolab_amd/csrc/*.hip $ROOT/pycolab_amd/csrc/*.cpp $ROOT/pycolab_amd/csrc/*.h $ROOT/pycolab_amd/csrc/Makefile $TMP/
mkdir -p $TMP/../../include 2>/dev/null || true
sed -i "s#../../include/pcx.h#$ROOT/include/pcx.h#g" $TMP/*.h $TMP/Makefile
make -s -C $TMP -j8 CXXFLAGS="-O3 -std=c++17 -fPIC -Wall -Wno-unused-result $*" 2>&1 | grep -E "error" || true
mkdir -p $ROOT/gpurun_variants
cp $TMP/libpcx.so $ROOT/gpurun_variants/libpcx_$TAG.so
rm -rf $TMP
ls -la $ROOT/gpurun_variants/libpcx_$TAG.so
