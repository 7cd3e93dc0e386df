"""dev tool: RUN-mode batch sweep of the engine in the tree this script is started from (ROOT = argv[1] or the repo root): ROI
preprocess + hipGraph-replayed eval forward + softmax, 100 k ROIs per batch size.  Used to A/B two checkouts on one box."""
import os
import sys
import time

ROOT = os.path.abspath(sys.argv[1]) if len(sys.argv) > 1 else os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from ifcb_classifier_amd import graph
from ifcb_classifier_amd.engine import Engine

POOL, BMAX = 8192, int(os.environ.get("SWEEP_BMAX", "1024"))
eng = Engine(graph.build('inception_v3', 100, pretrained=False), device=0, max_batch=BMAX, train_batch=1)
eng.init_weights(seed=1234)
rois, _ = bench.synth_rois(POOL + BMAX, 4321, eng.dev)
for B in [int(v) for v in os.environ.get("SWEEP_BATCHES", "256,512,768,1024,512,1024").split(",")]:
    nb = 100000 // B + 1

    def batch(k):
        s0 = (k * B) % POOL
        eng.load_rois(rois['pixels'], rois['offs'][s0:s0 + B], rois['hs'][s0:s0 + B], rois['ws'][s0:s0 + B], rois['max_h'], rois['max_w'])
        p = eng.forward_eval(B)
        eng.run(p.softmax)
    for k in range(3):
        batch(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(nb):
        batch(k)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print('%s batch %4d: %.3f ms/batch %.0f img/s' % (os.path.basename(ROOT), B, 1e3 * dt / nb, nb * B / dt), flush=True)
