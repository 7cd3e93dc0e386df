import sys, time, torch
sys.path.insert(0, '.')
from ifcb_classifier_amd import graph
from ifcb_classifier_amd.engine import Engine
import bench
# one launch addresses each tensor through a 2 GiB buffer descriptor: 776 inception_v3 images at most (engine.window_batch);
# the RUN loop feeds larger --batch values as chunks of that size
for B in (256, 512, 768):
    eng = Engine(graph.build('inception_v3', 100, pretrained=False), device=0, max_batch=B)
    eng.init_weights(seed=1)
    rois, _ = bench.synth_rois(B, 1, eng.dev)
    def step():
        eng.load_rois(**rois); p = eng.forward_eval(B); eng.run(p.softmax)
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print('B=%d: %.2f ms/batch  %.0f img/s' % (B, dt * 1e3, B / dt), flush=True)
    del eng; torch.cuda.empty_cache()
