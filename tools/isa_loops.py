"""Instruction mix of the smallest MFMA-containing loop of the named kernels in a hipcc -S listing.
usage: isa_loops.py <file.s> <mangled kernel name>..."""
import re,sys
from collections import Counter
s=open(sys.argv[1]).read()
for name in sys.argv[2:]:
    i=s.index(name+':'); j=s.index('s_endpgm',i)
    body=s[i:j].split('\n')
    labels={}
    for n,l in enumerate(body):
        m=re.match(r'^(\.LBB\d+_\d+):',l)
        if m: labels[m.group(1)]=n
    best=None
    for n,l in enumerate(body):
        m=re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)',l)
        if m and m.group(1) in labels and labels[m.group(1)]<n:
            a=labels[m.group(1)]
            ins=[x.strip().split()[0] for x in body[a:n] if x.strip() and not x.strip().startswith(('.',';'))]
            if any(x.startswith('v_mfma') for x in ins):
                if best is None or len(ins)<best[0]: best=(len(ins),Counter(re.sub(r'_e32|_e64','',x) for x in ins))
    print(name[-50:], best[0] if best else None)
    if best: print('   ', best[1].most_common(14))
