"""Basic-block VALU / SALU / memory / LDS instruction counts of one kernel in a hipcc -S listing: python tools/bbcount.py file.s <mangled-name-substring>"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = [i for i, l in enumerate(lines) if l.startswith('_Z') and key in l and l.rstrip().endswith(':') or (l.startswith('_Z') and key in l and ':' in l and not l.startswith('\t'))][0]
end = [i for i in range(start, len(lines)) if 's_endpgm' in lines[i]][-1] if False else next(i for i in range(start, len(lines)) if lines[i].strip().startswith('.section') or lines[i].strip().startswith('.end_amdhsa_kernel') or 'amdhsa_kernel' in lines[i])
blocks = []; cur = {'name': 'entry', 'v': 0, 's': 0, 'm': 0, 'l': 0, 'sc': 0, 'cmt': ''}; blocks.append(cur)
for l in lines[start + 1:end]:
    m = re.match(r'^(\.LBB\d+_\d+):(.*)', l)
    if m:
        cur = {'name': m.group(1), 'v': 0, 's': 0, 'm': 0, 'l': 0, 'sc': 0, 'cmt': m.group(2).strip()}; blocks.append(cur); continue
    t = l.strip()
    if not t or t[0] in ';.': continue
    op = t.split()[0]
    if op.startswith('v_'): cur['v'] += 1
    elif op.startswith('s_'): cur['s'] += 1
    elif op.startswith('ds_'): cur['l'] += 1
    elif op.startswith('scratch_'): cur['sc'] += 1
    elif op.startswith(('global_', 'buffer_', 'flat_')): cur['m'] += 1
tot = sum(b['v'] for b in blocks)
print('total VALU', tot)
for b in blocks:
    if b['v'] + b['m'] + b['l'] + b['sc'] >= int(sys.argv[3]) if len(sys.argv) > 3 else True:
        print('%-12s V %4d S %4d M %3d L %3d SC %2d  %s' % (b['name'], b['v'], b['s'], b['m'], b['l'], b['sc'], b['cmt'][:70]))
