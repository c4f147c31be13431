#!/usr/bin/env python3
"""The assembly of ONE kernel out of a -save-temps gfx950 .s file (label to
.Lfunc_end), optionally summarised as runs of opcodes.
usage: isa_kernel.py file.s <substring of the mangled name> [--ops [first last]]"""
import re
import sys


def main():
    txt = open(sys.argv[1]).read().split('\n')
    pat = sys.argv[2]
    start = next(i for i, l in enumerate(txt)
                 if l.startswith('_Z') and pat in l.split(':')[0] and '; @' in l)
    end = next(i for i in range(start, len(txt)) if txt[i].startswith('.Lfunc_end'))
    body = txt[start:end]
    if '--ops' not in sys.argv:
        print('\n'.join(body))
        return
    k = sys.argv.index('--ops')
    lo = int(sys.argv[k + 1]) if len(sys.argv) > k + 1 else 0
    hi = int(sys.argv[k + 2]) if len(sys.argv) > k + 2 else len(body)
    prev, n = None, 0
    for i, l in enumerate(body[lo:hi]):
        m = re.match(r'\s+([a-z_0-9]+)', l)
        if not m:
            continue
        op = m.group(1)
        if op != prev:
            if prev:
                print('%5d x %s' % (n, prev))
            prev, n = op, 0
        n += 1
    if prev:
        print('%5d x %s' % (n, prev))


if __name__ == '__main__':
    main()
