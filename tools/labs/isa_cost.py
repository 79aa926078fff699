#!/usr/bin/env python3
"""Weighted vector-issue cost of an asm fragment (stdin or file), using the issue classes tools/labs/valu_rate_lab.hip
measured on gfx950 with DEPENDENT CHAINS of one instruction: plain VOP1/VOP2 integer ops on VGPR operands ~1 unit,
v_bitop3_b32 on VGPRs ~1.4, everything else (shifts, multiplies, v_perm, v_bfi, packed ops, VOP3 forms, any vector op
reading an SGPR) ~2.
CAVEAT: a rough guide only.  The model did NOT predict kernel-level results in round 2: moving the rollout loop's
constants and Philox keys from SGPRs into VGPRs cut the modelled cost by 8 % (335 -> 307 units) but measured +2 % on the
fused rollout and +-0 on the self-play rollout (profiles/r02_e_summary.md).  Trust an A/B of the real kernel
(tools/labs/lib_ab.sh), not these weights."""
import re, sys, collections
FAST = {'v_xor_b32', 'v_add_u32', 'v_sub_u32', 'v_subrev_u32', 'v_and_b32', 'v_or_b32', 'v_mov_b32', 'v_not_b32'}
src = open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()
fast = slow = mid = 0
by = collections.Counter()
for l in src.split('\n'):
    l = l.split(';')[0].strip()
    if not l.startswith('v_'): continue
    op = l.split()[0]
    base = re.sub(r'_e32$|_e64$', '', op)
    ops = l[len(op):]
    sgpr = bool(re.search(r'\bs\d+\b|\bs\[\d+:\d+\]|vcc|exec', ops))
    if base == 'v_bitop3_b32' and not sgpr: mid += 1; by[op + ' (vgpr)'] += 1
    elif base in FAST and not sgpr: fast += 1; by[op] += 1
    else: slow += 1; by[op + (' +sgpr' if sgpr and base in FAST | {'v_bitop3_b32'} else '')] += 1
print("fast %d  bitop3(vgpr) %d  slow %d   -> issue units (fast 1, bitop3 1.4, slow 2): %.0f" % (fast, mid, slow, fast + 1.4 * mid + 2 * slow))
print(", ".join("%s %d" % kv for kv in by.most_common(40)))
