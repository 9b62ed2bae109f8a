"""Static instruction census of compiled kernels (no GPU): MFMA / vector / scalar / LDS / memory instructions of the whole kernel and
of every loop, from `hipcc -S --cuda-device-only` output.  usage: python tools/isa_census.py <file.s> [kernel-name-substring ...]
On gfx950 the vector instructions of an f32-MFMA kernel are paid in matrix-pipe time (tools/probe/mfma_valu_*.hip), so the
VALU column of a K loop / epilogue is the number to shrink."""
import re,sys,collections
def kernels(path):
    cur=None; body=[]
    for l in open(path):
        m=re.match(r'^(_Z\w+):',l)
        if m and ('kernel' in m.group(1)):
            cur=m.group(1); body=[]; continue
        if cur is not None:
            body.append(l)
            if 's_endpgm' in l:
                yield cur, body; cur=None
def classify(op):
    if op.startswith('v_mfma'): return 'MFMA'
    if op.startswith('v_'): return 'VALU'
    if op.startswith('s_waitcnt'): return 'wait'
    if op.startswith('s_barrier'): return 'barrier'
    if op.startswith('s_'): return 'SALU'
    if op.startswith('ds_'): return 'LDS'
    if op.startswith(('buffer_','global_','scratch_','flat_')): return 'VMEM'
    return 'other'
want=sys.argv[2:]
for name,body in kernels(sys.argv[1]):
    if want and not any(w in name for w in want): continue
    # split into regions: loops (by "Loop Header" comments) -> report whole-kernel and innermost-loop totals
    ops=[(i,l.split()[0]) for i,l in enumerate(body) if l.startswith('\t') and not l.strip().startswith((';','.')) and l.split()]
    tot=collections.Counter(classify(o) for _,o in ops)
    # loops: find label lines with Loop Header and the backward branch to them
    labels={}
    for i,l in enumerate(body):
        m=re.match(r'^(\.LBB\d+_\d+):',l)
        if m: labels[m.group(1)]=i
    loops=[]
    for i,l in enumerate(body):
        m=re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)',l) or re.search(r's_branch\s+(\.LBB\d+_\d+)',l)
        if m and m.group(1) in labels and labels[m.group(1)]<i:
            loops.append((labels[m.group(1)],i))
    print(name[:110]); print('   whole kernel:', dict(tot))
    for a,b in sorted(set(loops)):
        c=collections.Counter(classify(o) for i,o in ops if a<=i<=b)
        if c['MFMA'] or c['VALU']>40: print('   loop lines %d-%d:'%(a,b), dict(c))
