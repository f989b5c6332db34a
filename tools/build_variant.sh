#!/bin/bash
# Development tool: builds an A/B variant of libacgpu.so with extra compiler flags into ahocorasick_amd/lib_<name>/
# (select it at run time with ACGPU_LIB=ahocorasick_amd/lib_<name>/libacgpu.so).   usage: build_variant.sh name -DACGPU_X=1 ...
set -e
name=$1; shift
cd "$(dirname "$0")/../ahocorasick_amd/csrc"
make -j8 OUT=../lib_$name CXXFLAGS="-O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-pass-failed --offload-arch=gfx950 -I../../include $*"
