#!/usr/bin/env python3
"""Basic-block census of one kernel in hipcc's -S output: per block the number of VALU / SALU / LDS / VMEM / SMEM
instructions, waits, lane-spill traffic (v_readlane / v_writelane), and where each block branches to.

    hipcc ... --cuda-device-only -S -o engine.s csrc/engine.hip
    tools/isa_blocks.py engine.s 'batch_kernelILi4ELi1024ELi7ELb0ELb0E' [--dump LBBx_y ...]
"""
import re
import sys


def classify(op):
    if op.startswith('v_readlane') or op.startswith('v_writelane'):
        return 'lane'
    if op.startswith('v_readfirstlane'):
        return 'valu'
    if op.startswith('v_'):
        return 'valu'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith('global_') or op.startswith('buffer_') or op.startswith('flat_') or op.startswith('scratch_'):
        return 'vmem'
    if op.startswith('s_load') or op.startswith('s_buffer_load') or op.startswith('s_memtime') or op.startswith('s_memrealtime') or op.startswith('s_dcache'):
        return 'smem'
    if op.startswith('s_waitcnt'):
        return 'wait'
    if op.startswith('s_nop'):
        return 'nop'
    if op.startswith('s_cbranch') or op.startswith('s_branch') or op.startswith('s_endpgm') or op.startswith('s_setpc'):
        return 'branch'
    if op.startswith('s_'):
        return 'salu'
    return 'other'


def parse(path, needle):
    lines = open(path).read().split('\n')
    start = None
    for i, l in enumerate(lines):
        if l.startswith('_ZN') and needle in l and l.rstrip().split(':')[0].endswith('E') and ':' in l:
            start = i
            break
    if start is None:
        raise SystemExit('kernel not found: ' + needle)
    blocks = []
    cur = {'label': 'entry', 'ins': []}
    for l in lines[start + 1:]:
        s = l.strip()
        if s.startswith('.Lfunc_end'):
            break
        if not s or s.startswith(';') or s.startswith('.'):
            m = re.match(r'^(\.LBB\d+_\d+):', s)
            if m:
                blocks.append(cur)
                cur = {'label': m.group(1)[1:], 'ins': []}
            continue
        m = re.match(r'^(\.LBB\d+_\d+):', s)
        if m:
            blocks.append(cur)
            cur = {'label': m.group(1)[1:], 'ins': []}
            continue
        cur['ins'].append(s.split(';')[0].strip())
    blocks.append(cur)
    return blocks


def main():
    path, needle = sys.argv[1], sys.argv[2]
    dump = set()
    if '--dump' in sys.argv:
        dump = set(sys.argv[sys.argv.index('--dump') + 1:])
    blocks = parse(path, needle)
    index = {b['label']: i for i, b in enumerate(blocks)}
    tot = {}
    print('%-12s %5s %5s %5s %4s %4s %4s %4s %4s %4s  branches' % ('block', 'n', 'valu', 'salu', 'lds', 'vmem', 'smem', 'wait', 'nop', 'lane'))
    for b in blocks:
        c = {}
        targets = []
        for ins in b['ins']:
            op = ins.split()[0]
            k = classify(op)
            c[k] = c.get(k, 0) + 1
            tot[k] = tot.get(k, 0) + 1
            if k == 'branch':
                m = re.search(r'\.(LBB\d+_\d+)', ins)
                if m:
                    t = m.group(1)
                    back = '^' if index.get(t, 1 << 30) <= index[b['label']] else ''
                    targets.append(op.replace('s_cbranch_', '').replace('s_branch', 'jmp') + '->' + t + back)
        print('%-12s %5d %5d %5d %4d %4d %4d %4d %4d %4d  %s' % (b['label'], len(b['ins']), c.get('valu', 0), c.get('salu', 0), c.get('lds', 0), c.get('vmem', 0),
                                                               c.get('smem', 0), c.get('wait', 0), c.get('nop', 0), c.get('lane', 0), ' '.join(targets)))
        if b['label'] in dump:
            for ins in b['ins']:
                print('        ' + ins)
    print('total', tot)


if __name__ == '__main__':
    main()
