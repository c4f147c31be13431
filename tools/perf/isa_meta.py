#!/usr/bin/env python3
"""Kernel resource table from a -save-temps gfx950 assembly file:
name, VGPRs, AGPRs, SGPRs, scratch bytes per lane, static LDS bytes.
usage: isa_meta.py file.s [substring ...]"""
import re
import subprocess
import sys


def demangle(n):
    try:
        return subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt', n],
                              capture_output=True, text=True).stdout.strip()
    except OSError:
        return n


def main():
    txt = open(sys.argv[1]).read()
    pats = sys.argv[2:]
    body = txt[txt.find('amdhsa.kernels'):]
    rows = []
    for blk in re.split(r'\n  - \.agpr_count', body)[1:]:
        blk = '.agpr_count' + blk
        g = lambda k: re.search(r'\.%s:\s+(\S+)' % k, blk)
        name = g('name').group(1)
        rows.append((demangle(name).split('(')[0],
                     int(g('vgpr_count').group(1)),
                     int(g('agpr_count').group(1)),
                     int(g('sgpr_count').group(1)),
                     int(g('private_segment_fixed_size').group(1)),
                     int(g('group_segment_fixed_size').group(1))))
    print('%-60s %5s %5s %5s %8s %7s' % ('kernel', 'vgpr', 'agpr', 'sgpr',
                                         'scratch', 'lds'))
    for r in rows:
        if pats and not any(p in r[0] for p in pats):
            continue
        print('%-60s %5d %5d %5d %8d %7d' % r)


if __name__ == '__main__':
    main()
