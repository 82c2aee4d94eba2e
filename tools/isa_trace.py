#!/usr/bin/env python3
"""One-line opcode trace of a kernel's ISA (from hipcc -save-temps=obj): M mfma, t ds_read_b64_tr_b16, r other ds_read,
d other LDS op, G LDS-DMA, g other global/buffer op, |B| s_barrier, W(...) s_waitcnt ('*' marks the kernel's own inline-asm
waits), <br> branch.    usage: tools/isa_trace.py file.s mangled-name-substring [max-chars]"""
import re, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2]
lim = int(sys.argv[3]) if len(sys.argv) > 3 else 4000
for m in re.finditer(r'^(_Z\w+):[^\n]*\n', s, re.M):
    if pat not in m.group(1): continue
    i = m.end(); j = s.find('.end_amdhsa_kernel', i)
    out = []; in_asm = False
    for t in s[i:j].splitlines():
        t = t.strip()
        if t.startswith(';;#ASMSTART'): in_asm = True; continue
        if t.startswith(';;#ASMEND'): in_asm = False; continue
        if not t or t.startswith(';') or t.startswith('.'): continue
        op = t.split()[0]
        if op.startswith('v_mfma'): c = 'M'
        elif op.startswith('ds_read_b64_tr'): c = 't'
        elif op.startswith('ds_read'): c = 'r'
        elif op.startswith('ds_'): c = 'd'
        elif op.startswith('s_waitcnt'): c = 'W' + ('*' if in_asm else '') + '(' + t.split(None, 1)[1].replace('cnt', '') + ')'
        elif 'load_lds' in op: c = 'G'
        elif op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): c = 'g'
        elif op.startswith('s_barrier'): c = '|B|'
        elif op.startswith(('s_cbranch', 's_branch')): c = '<br>'
        elif op.startswith('v_exp'): c = 'e'
        else: c = ''
        out.append(c)
    print(m.group(1)); print(''.join(out)[:lim]); print()
