#!/bin/bash
# Development tool: compile one kernel file to gfx950 assembly and print, for the kernels whose mangled name matches
# $2 (a regex), registers, spills and the static instruction mix.  usage: tools/isa_stats.sh acgpu_tile.hip 'k_ac_tileILi4ELb1ELb0ELb0ELb0ELb1ELb1' [extra -D flags]
set -e
cd "$(dirname "$0")/../ahocorasick_amd/csrc"
src=$1; pat=$2; shift 2
out=/tmp/isa_$(basename $src .hip).s
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wno-pass-failed -x hip $src --cuda-device-only -S -o $out "$@" 2>/dev/null
python3 - "$out" "$pat" <<'PY'
import re, sys
lines = open(sys.argv[1]).read().split('\n')
pat = re.compile(sys.argv[2])
i = 0
while i < len(lines):
    m = re.match(r'^(_Z\w+):', lines[i])
    if m and pat.search(m.group(1)):
        name = m.group(1); j = i + 1; body = []
        while j < len(lines) and not lines[j].startswith('\t.section') and not lines[j].startswith('.Lfunc_end'):
            body.append(lines[j]); j += 1
        ins = [l.split()[0] for l in body if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
        from collections import Counter
        c = Counter(ins)
        valu = sum(v for k, v in c.items() if k.startswith('v_'))
        print(name)
        print('  instructions %d  VALU %d  SALU %d  DS %d  VMEM %d  v_mov %d  v_cndmask %d  v_readlane/writelane %d' % (
            len(ins), valu, sum(v for k, v in c.items() if k.startswith('s_')), sum(v for k, v in c.items() if k.startswith('ds_')),
            sum(v for k, v in c.items() if k.startswith(('global_', 'buffer_', 'flat_', 'scratch_'))), c['v_mov_b32_e32'] + c['v_mov_b64_e32'],
            c['v_cndmask_b32_e32'] + c['v_cndmask_b32_e64'], c['v_readlane_b32'] + c['v_writelane_b32']))
        # metadata
        k = j
        meta = {}
        while k < len(lines) and k < j + 400:
            mm = re.match(r'\s*\.amdhsa_(next_free_vgpr|next_free_sgpr|accum_offset)\s+(\S+)', lines[k])
            if mm: meta[mm.group(1)] = mm.group(2)
            mm = re.match(r'\s*;\s*(ScratchSize|SGPRBlocks|NumSgprs|NumVgprs|Occupancy|LDSByteSize|sgpr_spill_count|vgpr_spill_count).*?:\s*(\S+)', lines[k])
            if mm: meta[mm.group(1)] = mm.group(2)
            if '.end_amdhsa_kernel' in lines[k]: break
            k += 1
        print('  ', meta)
        i = j
    else:
        i += 1
PY
