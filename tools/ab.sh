#!/bin/bash
# Development tool: same-box A/B of library variants (ahocorasick_amd/lib_<name>/libacgpu.so; "cur" = ahocorasick_amd/lib),
# interleaved, N passes.  usage: tools/ab.sh N "kbench args" name1 name2 ...
n=$1; args=$2; shift 2
for i in $(seq $n); do
  for v in "$@"; do
    if [ "$v" = cur ]; then lib=ahocorasick_amd/lib/libacgpu.so; else lib=ahocorasick_amd/lib_$v/libacgpu.so; fi
    ACGPU_LIB=$lib timeout -k 10 150 python tools/kbench.py $args 2>&1 | grep median | sed "s/^/$v: /"
  done
done
