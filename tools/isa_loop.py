#!/usr/bin/env python3
"""Dev tool: print the innermost store loop(s) of a kernel from lib/asm/*.s with instruction counts by kind.
    python tools/isa_loop.py <mangled-name-substring> [max block length]"""
import re, sys, glob, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
t = "\n".join(open(f).read() for f in sorted(glob.glob(os.path.join(ROOT, "raycastworlds.jl_amd", "lib", "asm", "*.s"))))
want = sys.argv[1]; maxlen = int(sys.argv[2]) if len(sys.argv) > 2 else 70
for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)\.Lfunc_end', t, re.S | re.M):
    if want not in m.group(1):
        continue
    blocks, cur, name = [], [], None
    for l in m.group(2).split('\n'):
        s = l.split(';')[0].rstrip()
        if re.match(r'^\.LBB\d+_\d+:', s):
            blocks.append((name, cur)); name, cur = s, []
        elif s.strip() and not s.strip().startswith('.'):
            cur.append(s.strip())
    blocks.append((name, cur))
    print(m.group(1))
    for name, b in blocks:
        if name and any('global_store' in x for x in b) and any(re.search(r's_cbranch\w+ ' + re.escape(name[:-1]) + r'\b', x) for x in b) and len(b) <= maxlen:
            kinds = {"vector": sum(x.startswith('v_') for x in b), "scalar": sum(x.startswith('s_') for x in b),
                     "lds": sum(x.startswith('ds_') for x in b), "vmem": sum(x.startswith(('global_', 'buffer_', 'flat_')) for x in b)}
            sel = sum(x.startswith(('v_cmp', 'v_cndmask')) for x in b)
            print(f"  loop {name} {len(b)} instructions: {kinds}; compares + selects: {sel}; stores: {sum('global_store' in x for x in b)}")
            if "-v" in sys.argv:
                for x in b: print("      " + x)
    break
