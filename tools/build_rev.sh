#!/bin/bash
# Development tool: builds libacgpu.so from the kernel sources of a git revision into ahocorasick_amd/lib_<name>/ for
# same-box A/B runs (ACGPU_LIB=ahocorasick_amd/lib_<name>/libacgpu.so).   usage: build_rev.sh name rev
set -e
name=$1; rev=$2
root="$(cd "$(dirname "$0")/.." && pwd)"
tmp=$(mktemp -d)
mkdir -p $tmp/ahocorasick_amd $tmp/include
git -C "$root" archive $rev ahocorasick_amd/csrc include | tar -x -C $tmp
make -C $tmp/ahocorasick_amd/csrc -j8 OUT="$root/ahocorasick_amd/lib_$name" >/dev/null
rm -rf $tmp
ls -la "$root/ahocorasick_amd/lib_$name/libacgpu.so"
