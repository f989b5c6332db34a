#!/usr/bin/env python3
"""Development tool: register / spill / LDS numbers of the kernels in a device assembly file (hipcc --cuda-device-only -S)."""
import re, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for blk in s.split('  - .agpr_count')[1:]:
    name = re.search(r'\.name:\s+(\S+)', blk).group(1)
    if pat in name:
        g = lambda k: re.search(r'\.%s:\s+(\d+)' % k, blk).group(1)
        print(name[:75], 'vgpr', g('vgpr_count'), 'spill', g('vgpr_spill_count'), 'sgpr', g('sgpr_count'), 'sspill', g('sgpr_spill_count'), 'lds', g('group_segment_fixed_size'), 'scratch', g('private_segment_fixed_size'))
