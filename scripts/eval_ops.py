"""per-op milliseconds of the eval (RUN-mode) forward program: python scripts/eval_ops.py [B]
NOTE: ops on different lanes overlap, so a bracket also contains the contention with the op running beside it --
run with IFCBK_LANES=1 for isolated per-kernel times (the sum then equals the single-stream wall time)."""
import sys, json, collections, ctypes as C, torch
sys.path.insert(0, '.')
from ifcb_classifier_amd import graph, _lib
from ifcb_classifier_amd.engine import Engine
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
eng = Engine(graph.build('inception_v3', 100, pretrained=False), device=0, max_batch=B, train_batch=1)
eng.init_weights(seed=1)
rois, _ = bench.synth_rois(B, 1, eng.dev)
eng.load_rois(**rois)
eng.graph_eval = False
pl = eng.forward_eval(B)
prog = pl.fwd_eval
acc = [0.0] * prog.n
R = 5
for _ in range(R):
    ms = (C.c_float * prog.n)()
    eng.run(prog, ms)
    for i in range(prog.n):
        acc[i] += ms[i] / R
rows = []
name = C.create_string_buffer(128)
for i in range(prog.n):
    o = prog.arr[i]
    eng.ctx.lib.ifcbk_op_kernel(C.byref(o), name, 128)
    fl, by = C.c_double(), C.c_double()
    eng.ctx.lib.ifcbk_op_cost(C.byref(o), C.byref(fl), C.byref(by))
    rows.append({'i': i, 'tag': prog.tags[i], 'op': _lib.OP_NAMES.get(o.kind, str(o.kind)), 'kernel': name.value.decode(),
                 'ms': acc[i], 'gflop': fl.value / 1e9, 'mbytes': by.value / 1e6})
json.dump(rows, open('gpurun_out/eval_ops_%d.json' % B, 'w'))
by = collections.defaultdict(lambda: [0, 0, 0, 0])
for r in rows:
    b = by[r['op']]; b[0] += r['ms']; b[1] += r['gflop']; b[2] += r['mbytes']; b[3] += 1
print('total %.3f ms' % sum(r['ms'] for r in rows))
for k, v in sorted(by.items(), key=lambda kv: -kv[1][0]):
    print('%-18s n=%3d %7.3f ms %8.1f TF/s %6.2f TB/s' % (k, v[3], v[0], v[1] / v[0] if v[0] else 0, v[2] / v[0] / 1e3 if v[0] else 0))
for r in sorted(rows, key=lambda r: -r['ms'])[:25]:
    print('%-40s %-14s %-42s %6.3f ms %7.1f TF/s %5.2f TB/s' % (r['tag'][:40], r['op'], r['kernel'][:42], r['ms'], r['gflop'] / r['ms'], r['mbytes'] / r['ms'] / 1e3))
