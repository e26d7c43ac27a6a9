"""Where a kernel spills, relative to its barriers and fences (dev aid):  python tools/isa_events.py file.s 'k_filtfiltILi2ELi2ELi4E'"""
import re, sys
t = open(sys.argv[1]).read().split('\n')
start = next(i for i, l in enumerate(t) if l.startswith('_Z') and sys.argv[2] in l and l.rstrip().split(';')[0].strip().endswith(':'))
end = next(i for i in range(start, len(t)) if t[i].startswith('.Lfunc_end'))
ev = []
for i in range(start, end):
    ls = t[i].strip()
    for key, name in (('s_barrier', 'BAR'), ('scratch_store', 'st'), ('scratch_load', 'ld'), ('s_memrealtime', 'CLK'), ('buffer_wbl2', 'WBL2'), ('buffer_inv', 'INV'),
                      ('global_store', 'gst'), ('s_sleep', 'SLEEP')):
        if ls.startswith(key):
            ev.append((i - start, name))
print(end - start, 'lines')
out = []; prev = None; cnt = 0; first = 0
for i, e in ev:
    if e == prev: cnt += 1
    else:
        if prev: out.append(f"{prev}x{cnt}@{first}")
        prev, cnt, first = e, 1, i
out.append(f"{prev}x{cnt}@{first}")
print(' '.join(out))
