"""dev tool: hipGraph stream capture of a program with N lanes (IFCBK_LANES_CAP, default 4), with the library's fatal-signal
backtrace armed -- the configuration that crashed in round 1 (gpurun_out/b_l3.err, b_l4.err).  Prints where it dies, or
replays the captured graph and compares it with the plain launch list.
    IFCBK_SEGV_BACKTRACE=1 timeout -k 10 300 python scripts/capture_lanes.py [eval|train] [model]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault('IFCBK_SEGV_BACKTRACE', '1')
nl = os.environ.get('IFCBK_LANES_CAP', '4')
os.environ['IFCBK_LANES'] = nl
os.environ['IFCBK_LANES_EVAL'] = nl
import torch                                                    # noqa: E402
from ifcb_classifier_amd import graph                           # noqa: E402
from ifcb_classifier_amd.engine import Engine                   # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else 'eval'
name = sys.argv[2] if len(sys.argv) > 2 else 'inception_v3'
B, S = 6, (299 if name == 'inception_v3' else 224)
eng = Engine(graph.build(name, 5), 0, max_batch=B)
eng.init_weights(seed=1)
x = torch.rand(B, 3, S, S).cuda()
eng.load_input_nchw(x)
eng.target[:B].zero_()
pl = eng.plan(B)
eng.ensure_packed(pl)
eng.run(pl.evalprep)
prog = pl.fwd_eval if which == 'eval' else pl.fwd_bwd
eng.make_dropout_mask(B)
print('program', which, 'ops', prog.n, 'lanes', prog.lanes, flush=True)
eng.run(prog)
torch.cuda.synchronize()
head = [h for h in eng.heads if not h.aux][0]
ref = head.logits[:B].clone()
refg = eng.G.clone()
print('plain launch ok; capturing ...', flush=True)
g = eng.ctx.capture(prog.arr, prog.n)
print('captured; replaying ...', flush=True)
head.logits.zero_()
eng.ctx.graph_launch(g, eng.stream())
torch.cuda.synchronize()
print('replay == plain launches:', torch.equal(head.logits[:B], ref), (torch.equal(eng.G, refg) if which != 'eval' else ''), flush=True)
print('CAPTURE_OK lanes', prog.lanes)
