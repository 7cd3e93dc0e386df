"""dev tool: per kernel of a hipcc -S listing, the loops that contain MFMAs or global loads, with what they wait on.
    hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -S csrc/<file>.hip -o /tmp/k.s && python scripts/isa_loops.py /tmp/k.s [name filter] [drains]
('drains': only loops with a vmcnt(0) AND memory operations of their own)
Columns per loop: instructions, MFMAs, global / buffer loads (lds = LDS-DMA), stores, `s_waitcnt vmcnt(0)`, counted vmcnt waits, barriers.
A vmcnt(0) inside a loop that also issues loads or stores is a full drain per iteration: intended (one tile ahead) or the round-5
conv_rows3x3 disease (`__syncthreads()` = full fence; the compiler's vmcnt(0) in front of plain LDS reads while LDS-DMA is pending)."""
import re
import sys

lines = open(sys.argv[1]).read().split('\n')
flt = sys.argv[2] if len(sys.argv) > 2 else ''
kern, cur = {}, None
for ln in lines:
    m = re.match(r'^(_Z\S+):', ln)
    if m and not ln.startswith('.'):
        cur = m.group(1)
        kern[cur] = []
    elif cur is not None:
        kern[cur].append(ln)
        if 's_endpgm' in ln:
            cur = None
for k, body in kern.items():
    if flt not in k:
        continue
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m:
            labels[m.group(1)] = i
    rows = []
    for i, l in enumerate(body):
        m = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            b = body[labels[m.group(1)]:i]
            cnt = lambda f: sum(1 for x in b if f(x))
            nm = cnt(lambda x: 'v_mfma' in x)
            nl = cnt(lambda x: ('buffer_load' in x or 'global_load' in x) and ' lds' not in x)
            nd = cnt(lambda x: 'buffer_load' in x and ' lds' in x)
            ns = cnt(lambda x: 'buffer_store' in x or 'global_store' in x)
            if (nm or nl or nd) and not (len(sys.argv) > 3 and sys.argv[3] == 'drains' and not (cnt(lambda x: 'vmcnt(0)' in x) and (nl or nd or ns))):
                rows.append('    loop %5d instr: mfma %3d  loads %3d  lds-dma %3d  stores %3d  vmcnt(0) %d  vmcnt(n) %d  barriers %d'
                            % (len(b), nm, nl, nd, ns, cnt(lambda x: 'vmcnt(0)' in x), cnt(lambda x: 'vmcnt(' in x and 'vmcnt(0)' not in x),
                               cnt(lambda x: 's_barrier' in x)))
    if rows:
        print(k[:110])
        print('\n'.join(rows))
