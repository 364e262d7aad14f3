"""List the wait / barrier / DMA / branch skeleton of one kernel in a hipcc -save-temps .s file.
usage: isa_waits.py file.s mangled-name-substring [extra-regex]"""
import re, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2]
extra = sys.argv[3] if len(sys.argv) > 3 else None
for k in re.split(r'\n(?=_Z\w+:)', s):
    name = k.split(':')[0]
    if pat not in name:
        continue
    lines = k.split('\n')
    print(name, len(lines), "lines")
    for i, l in enumerate(lines):
        t = l.strip()
        if (t.startswith('global_load_lds') or t.startswith('buffer_load') and ' lds' in t or ('s_waitcnt' in t and 'vmcnt' in t) or t.startswith('s_barrier')
                or t.startswith('s_cbranch') or t.startswith('s_branch') or re.match(r'\.LBB', t) or (extra and re.search(extra, t))):
            print(i, t)
    for key in ['NumVgprs', 'NumAgprs', 'TotalNumVgprs', 'ScratchSize', 'Occupancy', 'NumSgprs']:
        mm = re.search(r'; ' + key + r': (\d+)', k)
        if mm:
            print(key, mm.group(1))
