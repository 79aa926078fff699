#!/usr/bin/env python3
"""Instruction mix of kernels in build/asm/soccer_hip.s (make -C gym_soccer_littman94_amd/csrc asm).
Usage: tools/isa_count.py [substring of the mangled kernel name ...]"""
import collections, re, sys
pats = sys.argv[1:] or ['step_kernel_swar', 'step_kernel_hot']
s = open('build/asm/soccer_hip.s').read().split('\n')
name, body = None, []
def report(name, body):
    ins = [l.strip() for l in body]
    ins = [l for l in ins if l and not l.startswith(('.', ';', '//')) and not l.endswith(':')]
    v = [l for l in ins if l.startswith('v_')]; sa = [l for l in ins if l.startswith('s_')]
    mem = [l for l in ins if l.startswith(('global_', 'flat_', 'buffer_', 'ds_', 'scratch_'))]
    c = collections.Counter(l.split()[0] for l in v)
    print("%s\n  total %d  VALU %d  SALU %d  MEM %d" % (name, len(ins), len(v), len(sa), len(mem)))
    print("  " + ", ".join("%s %d" % kv for kv in c.most_common(18)))
for l in s:
    m = re.match(r'^(_Z\w+):\s*(;.*)?$', l)
    if m:
        name, body = m.group(1), []
        continue
    if name is not None:
        if l.strip().startswith('s_endpgm'):
            body.append(l)
        if l.startswith('.Lfunc_end') or l.strip().startswith('.section'):
            if any(p in name for p in pats):
                report(name, body)
            name = None
            continue
        body.append(l)
