"""Per kernel of a hipcc -S listing: the order of loads (L global, F flat, S scratch), s_waitcnt vmcnt(N) ([N]), branches (b),
barriers (|), MFMA groups (m) and atomics (A) -- shows at a glance whether loads are in flight across the matrix instructions
or waited for where they were issued (loads behind branches are).  usage: isa_waits.py <file.s> <regex on the mangled name>"""
import re, sys
s=open(sys.argv[1]).read()
pat=sys.argv[2]
lines=s.split('\n')
cur=None; seq=[]
for l in lines:
    m=re.match(r'^(_Z\w+):', l)
    if m:
        if cur and re.search(pat,cur): print(cur[:70], ''.join(seq)[:600]); print()
        cur=m.group(1); seq=[]; continue
    t=l.strip()
    if t.startswith(('global_load','buffer_load','flat_load','scratch_load')): seq.append('S' if t.startswith('scratch') else ('F' if t.startswith('flat') else 'L'))
    elif t.startswith('s_waitcnt') and 'vmcnt' in t: seq.append('['+re.search(r'vmcnt\((\d+)\)',t).group(1)+']')
    elif t.startswith('s_cbranch'): seq.append('b')
    elif t.startswith('s_barrier'): seq.append('|')
    elif t.startswith('v_mfma'): 
        if not seq or seq[-1]!='m': seq.append('m')
    elif t.startswith('global_atomic'): seq.append('A')
if cur and re.search(pat,cur): print(cur[:70], ''.join(seq)[:600])
