#!/usr/bin/env python3
"""List the `s_waitcnt vmcnt(...)` the COMPILER put into kernels that use LDS-DMA (global_load_lds): a wait it adds in front of
an LDS read it cannot prove disjoint from the DMA destinations drains the whole ring at every stage.  The kernels' own counted
waits are inline asm (between #ASMSTART / #ASMEND) and are not listed.
usage: tools/isa_dma_waits.py file.s [...]   (hipcc -save-temps=obj gives <name>-hip-amdgcn-amd-amdhsa-gfx950.s)"""
import re, sys
for path in sys.argv[1:]:
    s = open(path).read()
    for m in re.finditer(r'^(_Z\w+):[^\n]*\n', s, re.M):
        i = m.end(); j = s.find('.end_amdhsa_kernel', i)
        if j < 0: continue
        body = s[i:j].splitlines()
        if not any('load_lds' in l or ('lds' in l and l.strip().startswith(('global_load', 'buffer_load'))) for l in body): continue
        in_asm = False; hits = []
        dma = [k for k, l in enumerate(body) if 'load_lds' in l]
        for k, l in enumerate(body):
            t = l.strip()
            if t.startswith(';;#ASMSTART'): in_asm = True
            elif t.startswith(';;#ASMEND'): in_asm = False
            elif t.startswith('s_waitcnt') and 'vmcnt' in t and not in_asm:
                nxt = next((x.strip().split()[0] for x in body[k + 1:k + 6] if x.strip() and not x.strip().startswith(';')), '')
                hits.append((k, t, nxt + ('      <-- between LDS-DMA issues: inside the ring loop' if dma and dma[0] < k < dma[-1] else '')))
        name = m.group(1)
        print(f'{name[:100]}: {len(hits)} compiler vmcnt waits')
        for k, t, nxt in hits: print(f'    line {k}: {t}   -> next: {nxt}')
