"""dev tool: RUN-mode throughput with ONE engine (batches back to back on one stream) vs TWO engines (two batches in flight on two
streams: eval samples are independent, so the HBM-bound stem of one batch can run beside the MFMA-bound 17x17 stage of the other).
    python scripts/run_two_engines.py [batch] [batches]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ifcb_classifier_amd import graph
from ifcb_classifier_amd.engine import Engine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 60
POOL = 4096
engs = []
for k in range(2):
    e = Engine(graph.build('inception_v3', 100, pretrained=False), device=0, max_batch=B, train_batch=1)
    e.init_weights(seed=1234)
    engs.append(e)
rois, _ = bench.synth_rois(POOL + B, 4321, engs[0].dev)
streams = [torch.cuda.Stream() for _ in range(2)]


def batch(e, k):
    s0 = (k * B) % POOL
    e.load_rois(rois['pixels'], rois['offs'][s0:s0 + B], rois['hs'][s0:s0 + B], rois['ws'][s0:s0 + B], rois['max_h'], rois['max_w'])
    p = e.forward_eval(B)
    e.run(p.softmax)


def one(nb):
    for k in range(nb):
        batch(engs[0], k)


def two(nb):
    for k in range(nb):
        with torch.cuda.stream(streams[k & 1]):
            batch(engs[k & 1], k)


for name, fn in (('one engine', one), ('two engines', two), ('one engine', one), ('two engines', two)):
    fn(4)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(NB)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print('%-12s batch %d: %.3f ms/batch  %.0f img/s' % (name, B, 1e3 * dt / NB, NB * B / dt), flush=True)
