#!/usr/bin/env python
"""Headline benchmark: inception_v3 100-class bf16 TRAIN on synthetic IFCB ROIs (BASELINE.json configs[1]).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus N --steps K --warmup W          # starts its N ranks itself (self_launch below), or, equivalently,
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one per-GPU batch of 256 synthetic ROIs that are already
resident in HBM as ragged u8 images: on-GPU PIL-exact resize/normalise -> forward -> CE(+0.4 aux) ->
backward -> (gradient all-reduce over RCCL when N>1, overlapped with backward) -> Adam -> bf16 weight
repack.  Rank 0 prints ONE JSON line.  ``roofline`` is measured with HIP events recorded around every
launch of the timed steps themselves (ifcbk_run_program_ev, no extra synchronisation); ``cpu_baseline``
times the CPU oracle (the reference's torch-CPU arithmetic) on the host cores, rank 0, N=1 only.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0       # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"
HBM_PEAK_GBPS = 8000.0
TRAIN_GFLOP_PER_IMG = 34.26          # fwd + dgrad + wgrad, conv1a has no dgrad


def synth_rois(n, seed, device):
    """u8 grayscale ROIs, h,w ~ U{32..299}, concatenated; returns device tensors."""
    g = torch.Generator().manual_seed(seed)
    hs = torch.randint(32, 300, (n,), generator=g, dtype=torch.int32)
    ws = torch.randint(32, 300, (n,), generator=g, dtype=torch.int32)
    sizes = hs.long() * ws.long()
    offs = torch.zeros(n, dtype=torch.int64)
    offs[1:] = torch.cumsum(sizes, 0)[:-1]
    total = int(sizes.sum())
    pix = torch.randint(0, 256, (total,), generator=g, dtype=torch.uint8)
    return dict(pixels=pix.to(device), offs=offs.to(device), hs=hs.to(device), ws=ws.to(device),
                max_h=int(hs.max()), max_w=int(ws.max()), in_channels=1), (hs, ws, offs, pix)


def host_cores():
    """threads this process may really use (cgroup quota and affinity, not the machine's core count)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        q, p = open('/sys/fs/cgroup/cpu.max').read().split()
        if q != 'max':
            n = min(n, max(1, int(float(q) / float(p) + 0.5)))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline_eval(batch=32, budget_s=10.0, seed=1234):
    """the reference's CPU RUN path restated by the oracle ("train and eval"): PIL L->RGB->resize->ToTensor per ROI,
    eval forward of the fp32 torch-CPU backbone, softmax (neuston_models.py:152-157); bounded to ~budget_s of CPU work."""
    import numpy as np
    from PIL import Image
    from oracle import tv_models
    cores = host_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(seed)
    model = tv_models.get_namebrand_model('inception_v3', 100, storage='fp32')
    model.eval()
    _, (hs, ws, offs, pix) = synth_rois(batch, seed, 'cpu')

    def step():
        imgs = []
        for i in range(batch):
            a = pix[offs[i]:offs[i] + int(hs[i]) * int(ws[i])].view(int(hs[i]), int(ws[i])).numpy()
            im = Image.fromarray(a, 'L').convert('RGB').resize((299, 299), Image.BILINEAR)
            imgs.append(torch.from_numpy(np.asarray(im).copy()).permute(2, 0, 1).float().div(255))
        with torch.no_grad():
            return torch.softmax(model(torch.stack(imgs)), 1)

    step()
    steps, t0 = 0, time.perf_counter()
    while True:
        step()
        steps += 1
        dt = time.perf_counter() - t0
        if dt >= 0.6 * budget_s or steps >= 64 or dt + dt / steps > budget_s:
            break
    return dict(value=round(batch * steps / dt, 3), unit='images/s', cores=cores, kind='port',
                sample='%d eval batches of %d after 1 warm-up (PIL resize + fp32 torch-CPU oracle eval forward + softmax), %.1f s'
                       % (steps, batch, dt))


def cpu_baseline(batch=32, budget_s=25.0, seed=1234):
    """the reference's CPU path restated by the oracle: PIL L->RGB->resize->ToTensor per ROI, then the
    fp32 torch-CPU train step (loss = CE + 0.4 CE_aux, Adam 1e-3; neuston_models.py:63-86).  Bounded: one
    warm-up step, then steps until ~15 s of timed CPU work (at least 1 step; total bounded by budget_s)."""
    import numpy as np
    from PIL import Image
    from oracle import tv_models
    cores = host_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(seed)
    model = tv_models.get_namebrand_model('inception_v3', 100, storage='fp32')
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    crit = torch.nn.CrossEntropyLoss()
    _, (hs, ws, offs, pix) = synth_rois(batch, seed, 'cpu')
    y = torch.randint(0, 100, (batch,), generator=torch.Generator().manual_seed(seed))

    def step():
        imgs = []
        for i in range(batch):
            a = pix[offs[i]:offs[i] + int(hs[i]) * int(ws[i])].view(int(hs[i]), int(ws[i])).numpy()
            im = Image.fromarray(a, 'L').convert('RGB').resize((299, 299), Image.BILINEAR)
            imgs.append(torch.from_numpy(np.asarray(im).copy()).permute(2, 0, 1).float().div(255))
        x = torch.stack(imgs)
        out = model(x)
        loss = crit(out.logits, y) + 0.4 * crit(out.aux_logits, y)
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss.item()

    t0 = time.perf_counter()
    step()
    warm = time.perf_counter() - t0
    print('[cpu_baseline] warm-up step of batch %d on %d threads: %.1f s' % (batch, cores, warm), file=sys.stderr, flush=True)
    target = min(12.0, max(1.0, budget_s - warm))      # ~12 s of timed CPU work, never more than budget_s in total
    steps = 0
    t0 = time.perf_counter()
    while True:
        step()
        steps += 1
        dt = time.perf_counter() - t0
        print('[cpu_baseline] step %d, %.1f s' % (steps, dt), file=sys.stderr, flush=True)
        if dt >= target or steps >= 64 or dt + dt / steps > budget_s:
            break
    return dict(value=round(batch * steps / dt, 3), unit='images/s', cores=cores, kind='port',
                sample='%d train steps of batch %d after 1 warm-up (PIL resize + fp32 torch-CPU oracle fwd/bwd/Adam), %.1f s'
                       % (steps, batch, dt))


def run_mode_leg(args, local):
    """BASELINE configs[3] / SURVEY 8(d) config 4: RUN-mode inference on 1 M synthetic IFCB ROIs, one GPU, fixed-size batches,
    the eval forward replayed as a hipGraph.  Per batch: on-GPU PIL-exact resize + normalise of ragged u8 ROIs -> eval forward
    (BatchNorm folded into the conv epilogues) -> softmax; file writing excluded.  The ROIs are windows of a pool of 8,192
    distinct synthetic ROIs resident in HBM (1 M distinct ones would be 27 GB of host-generated random bytes; the kernels' work
    does not depend on the pixel values).  Batch sweep on 100 k ROIs each, then the full count at the best batch size."""
    from ifcb_classifier_amd import graph
    from ifcb_classifier_amd.engine import Engine
    POOL, BMAX = 8192, 2048
    # (an inference engine, as neuston_net RUN builds it: activations for BMAX images, gradient-side buffers for one)
    eng = Engine(graph.build('inception_v3', args.classes, pretrained=False), device=local, max_batch=BMAX, train_batch=1)
    eng.init_weights(seed=1234)
    rois, _ = synth_rois(POOL + BMAX, 4321, eng.dev)

    def run(B, n_rois):
        nb = (n_rois + B - 1) // B
        pl = eng.plan(B)

        def stage(k):
            s0 = (k * B) % POOL
            slot, side = eng.prefetch_begin()
            with torch.cuda.stream(side):
                eng.load_rois(rois['pixels'], rois['offs'][s0:s0 + B], rois['hs'][s0:s0 + B], rois['ws'][s0:s0 + B], rois['max_h'],
                              rois['max_w'], slot=slot)
            eng.prefetch_end(slot)

        pipelined = os.environ.get('IFCBK_BENCH_PIPELINE', '0') != '0'     # default off: see the training leg

        def batch(k):
            if pipelined:
                # batch k was staged by the previous call; batch k+1 is preprocessed on the side stream beside this forward
                eng.use_prefetched()
                stage(k + 1)
            else:
                s0 = (k * B) % POOL
                eng.load_rois(rois['pixels'], rois['offs'][s0:s0 + B], rois['hs'][s0:s0 + B], rois['ws'][s0:s0 + B], rois['max_h'],
                              rois['max_w'])
            p = eng.forward_eval(B)
            eng.run(p.softmax)
        if pipelined:
            stage(0)
        for k in range(3):
            batch(k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(nb):
            batch(k)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        return nb * B / dt, 1e3 * dt / nb, nb * B

    sweep = {}
    for B in (256, 512, 768, 1024, 2048):      # from 1024 on beyond the 776-image descriptor window of one launch (convolutions run over image groups)
        ips, ms, n = run(B, 100000)
        sweep[str(B)] = dict(images_per_s=round(ips, 1), ms_per_batch=round(ms, 3), rois=n)
    best = max(sweep, key=lambda b: sweep[b]['images_per_s'])
    ips, ms, n = run(int(best), args.infer_rois)
    tf = ips * 11.423e-3
    cpu_eval = None
    hipgraph, lanes_eval = bool(eng.graph_eval), eng.NL_eval
    if not args.no_cpu_baseline:
        del eng
        torch.cuda.empty_cache()
        cpu_eval = cpu_baseline_eval()
    return dict(cpu_baseline=cpu_eval, workload='RUN-mode inference, inception_v3 %d-class bf16, %d synthetic ROIs h,w~U{32..299} (windows of a pool of %d '
                         'in HBM), batch %s, hipGraph-replayed eval forward, preprocess + softmax included, file writes excluded '
                         '(BASELINE.json configs[3])' % (args.classes, n, POOL, best),
                images_per_s=round(ips, 1), ms_per_batch=round(ms, 3), batch=int(best), rois=n, seconds=round(n / ips, 2),
                batch_sweep_100k=sweep, hipgraph=hipgraph, program_lanes=lanes_eval,
                roofline=dict(bound='mfma', achieved=round(tf, 1), peak=MFMA_BF16_PEAK_TFLOPS, unit='TFLOP/s',
                              frac=round(tf / MFMA_BF16_PEAK_TFLOPS, 4), flops_per_image=11.423e9,
                              note='whole RUN step (preprocess + 94 convs with folded BatchNorm + pools + head + softmax) against the '
                                   'dense bf16 MFMA peak; algorithmic flops = conv + fc MACs x 2 (SURVEY 8(d))'))


def fp32_leg(args, local):
    """the fp32 PARITY mode (fp32 storage, v_mfma_f32_16x16x4_f32: the mode that meets the north-star 1e-3 logit tolerance,
    tests/test_gpu_model.py::test_fp32_*) timed on the same workload, so that nobody reads the bf16 speed with the fp32 parity"""
    from ifcb_classifier_amd import graph
    from ifcb_classifier_amd.engine import Engine
    B = args.batch
    eng = Engine(graph.build('inception_v3', args.classes, pretrained=False), device=local, max_batch=B, dtype='fp32')
    if eng.max_batch < B:
        return dict(skipped='batch %d exceeds the fp32 descriptor window (%d images)' % (B, eng.window_batch))
    eng.init_weights(seed=1234)
    rois, _ = synth_rois(B, 1234, eng.dev)
    eng.target[:B].copy_(torch.randint(0, args.classes, (B,), generator=torch.Generator().manual_seed(99)))
    for _ in range(2):
        eng.load_rois(**rois)
        eng.train_step(B)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.fp32_steps):
        eng.load_rois(**rois)
        eng.train_step(B)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.fp32_steps
    return dict(dtype='f32', ms_per_step=round(1e3 * dt, 2), images_per_s=round(B / dt, 1), steps=args.fp32_steps,
                peak_tflops=157.3, frac_of_fp32_mfma_peak=round(B / dt * TRAIN_GFLOP_PER_IMG * 1e-3 / 157.3, 4),
                note='parity mode: logits within 1e-3 of the fp32 CPU oracle (train 4e-5, eval 3e-7); the headline value is the bf16 mode, '
                     'whose random-init logits sit inside the bf16-storage envelope of the oracle (DESIGN.md section 4)')


def lib_sha16():
    import hashlib
    from ifcb_classifier_amd import _lib
    return hashlib.sha256(open(_lib.LIB_PATH, 'rb').read()).hexdigest()[:16]


def stage_of(tag):
    """network stage of an op by its layer tag (the spatial size the layer's convs work at)"""
    t = tag.split('+')[0]
    if t.startswith('Conv2d_') or t.startswith('maxpool') or t.startswith('input'):
        return 'stem'
    if t.startswith(('Mixed_5', 'Mixed_6a')):
        return '35x35'
    if t.startswith(('Mixed_6', 'Mixed_7a', 'AuxLogits')):
        return '17x17'
    if t.startswith('Mixed_7'):
        return '8x8'
    return 'head+update'


def layer_roofline(eng, pl, nsteps, hbm_tbps=6.29):
    """per-op roofline of the survey pass (every op alone on the GPU, one lane): MAC-weighted achieved / bound of the conv ops --
    bound = min(MFMA peak, arithmetic intensity x measured-copy HBM rate), as scripts/layer_roofline.py prints per layer -- and
    the step's isolated-op time by network stage (comparable across rounds)"""
    n = pl.step.n
    ms = (C.c_float * n)()
    tot = [0.0] * n
    for k in range(nsteps):
        eng.ctx.call('ifcbk_program_times', k, n, ms)
        for j in range(n):
            tot[j] += ms[j] / nsteps
    stages, conv_ms, conv_bound_ms, conv_fl = {}, 0.0, 0.0, 0.0
    for j in range(n):
        if tot[j] <= 0.0:
            continue
        op = pl.step.arr[j]
        fl, by = C.c_double(), C.c_double()
        eng.ctx.lib.ifcbk_op_cost(C.byref(op), C.byref(fl), C.byref(by))
        st = stages.setdefault(stage_of(pl.step.tags[j]), dict(ms=0.0, conv_ms=0.0, conv_gflop=0.0))
        st['ms'] += tot[j]
        if fl.value > 0 and by.value > 0 and op.kind in (1, 2, 3, 20, 22, 25, 29, 31, 39):
            bound = min(MFMA_BF16_PEAK_TFLOPS * 1e12, fl.value / by.value * hbm_tbps * 1e12)
            conv_ms += tot[j]
            conv_bound_ms += 1e3 * fl.value / bound
            conv_fl += fl.value
            st['conv_ms'] += tot[j]
            st['conv_gflop'] += fl.value / 1e9
    return {'conv_mac_weighted_achieved_over_bound': round(conv_bound_ms / conv_ms, 4) if conv_ms else None,
            'conv_ms_at_roofline': round(conv_bound_ms, 3), 'conv_ms_measured': round(conv_ms, 3),
            'isolated_ms_by_stage': {k: dict(ms=round(v['ms'], 3), conv_ms=round(v['conv_ms'], 3),
                                             conv_tflops=round(v['conv_gflop'] / v['conv_ms'], 1) if v['conv_ms'] else None)
                                     for k, v in sorted(stages.items())},
            'note': 'survey pass (every op bracketed, one lane, each kernel alone on the GPU); bound = min(2.5 PF, AI x 6.29 TB/s)'}


def launch_argv(n, argv, port):
    """the command that starts the N ranks of this benchmark: what the driver itself runs for N > 1 (one process per GPU over
    RCCL).  The reference gets its N processes the same way without a launcher on the command line: Lightning's ddp_spawn from
    ``gpus=len(args.gpus)`` (neuston_net.py:101-107,430-432)."""
    return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
            '--master-port', str(port), os.path.abspath(__file__)] + list(argv)


def free_port():
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(n, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks as CHILD processes (this process has not touched the
    GPU and never will: no torch.cuda call, no library load -- nothing re-execs a process that initialised HIP), relay rank 0's
    JSON line unchanged as the LAST stdout line, everything else to stderr, and return the children's exit code.
    The launcher and its ranks run in a process group of their own: if this process is told to stop (SIGTERM / SIGINT / any
    exception) the whole group is terminated and reaped -- no GPU rank is left behind as an orphan.  The rendezvous port is
    picked by bind-and-close, which another job on the box can win in between: a launcher that dies on "address already in use"
    is started again on a fresh port (at most 3 times); any other failure is final."""
    import signal
    import subprocess
    import threading
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', str(max(1, host_cores() // n)))

    def on_term(signum, frame):
        raise SystemExit(128 + signum)
    old = {sg: signal.signal(sg, on_term) for sg in (signal.SIGTERM, signal.SIGINT)}
    try:
        for attempt in range(3):
            cmd = launch_argv(n, argv, free_port())
            print('[bench] --gpus %d without WORLD_SIZE: starting the ranks: %s' % (n, ' '.join(cmd)), file=sys.stderr, flush=True)
            proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, start_new_session=True)
            port_taken = [False]

            def relay_err():
                for ln in proc.stderr:
                    if 'EADDRINUSE' in ln or 'ddress already in use' in ln:
                        port_taken[0] = True
                    sys.stderr.write(ln)
                    sys.stderr.flush()
            th = threading.Thread(target=relay_err, daemon=True)
            th.start()
            line = None
            try:
                for ln in proc.stdout:
                    t = ln.strip()
                    if t.startswith('{') and '"metric"' in t:
                        try:
                            json.loads(t)
                            line = t
                            continue
                        except ValueError:
                            pass
                    sys.stderr.write(ln)
                    sys.stderr.flush()
                rc = proc.wait()
                th.join(timeout=5)
            finally:
                if proc.poll() is None:                      # we are being stopped: take the ranks with us
                    try:
                        os.killpg(proc.pid, signal.SIGTERM)
                        try:
                            proc.wait(timeout=15)
                        except subprocess.TimeoutExpired:
                            os.killpg(proc.pid, signal.SIGKILL)
                            proc.wait()
                    except ProcessLookupError:
                        pass
            if rc != 0 and line is None and port_taken[0] and attempt < 2:
                print('[bench] the rendezvous port was taken by another job; starting the ranks again on a fresh port',
                      file=sys.stderr, flush=True)
                continue
            break
    finally:
        for sg, h in old.items():
            signal.signal(sg, h)
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        print('[bench] the ranks exited 0 but rank 0 printed no result line', file=sys.stderr, flush=True)
        rc = 1
    return rc


def rehearse(args, rank, world):
    """IFCBK_BENCH_REHEARSE=1: the N > 1 control flow WITHOUT a GPU -- rendezvous on gloo, the engine's real backward op list and
    bucket plan (Engine(plan_only=True)), the configured bucket exchange (IFCBK_DP_EXCHANGE) over the real bucket sizes on CPU
    tensors, the barrier / max-over-ranks timing and the rank-0 line.  Not a measurement: `value` is null and the line says so.
    What the CPU test of the self-launch path runs (tests/test_bench_launch_cpu.py)."""
    import torch.distributed as dist
    dist.init_process_group('gloo')
    from ifcb_classifier_amd import graph
    from ifcb_classifier_amd.dp import make_exchange, run_overlapped
    from ifcb_classifier_amd.engine import Engine
    torch.set_num_threads(2)
    eng = Engine(graph.build('inception_v3', args.classes, pretrained=False), max_batch=2, plan_only=True)
    pl = eng.plan(2)
    segs = eng.ddp_segments(pl)
    exchange, mode = make_exchange(dist)
    eng.G.fill_(float(rank + 1))
    dist.barrier()
    if os.environ.get('IFCBK_REHEARSE_SLEEP'):          # test hook: ranks that live long enough to be killed (launcher clean-up test)
        time.sleep(float(os.environ['IFCBK_REHEARSE_SLEEP']))
    t0 = time.perf_counter()
    for _ in range(max(1, min(args.steps, 3))):
        eng.G.fill_(float(rank + 1))
        run_overlapped(segs, lambda seg: None, eng.G, exchange)
    dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    want = world * (world + 1) / 2.0
    ok = bool((eng.G == want).all())
    devs = [None] * world
    dist.all_gather_object(devs, 'cpu:%d' % rank)
    if rank == 0:
        print(json.dumps({
            'metric': 'train images/sec, inception_v3 299^2 IFCB ROIs', 'value': None, 'unit': 'images/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': None, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': 'REHEARSAL on CPU (IFCBK_BENCH_REHEARSE=1): rendezvous, bucket plan and exchange only, no GPU work',
                       'global_batch': world * args.batch, 'parallelism': 'dp%d' % world, 'rank_devices': devs,
                       'exchange': mode, 'buckets': [int(s[3] - s[2]) for s in segs], 'IFCBK_LANES': eng.NL,
                       'program_lanes': eng.NL, 'weight_gradient_lanes': eng.wgrad_lane},
            'single_gpu_same_lanes_ms': None,
            'single_gpu_same_lanes_note': 'rehearsal: nothing is measured (a GPU run times 10 local steps at the DP lane count here)',
            'rehearsal': True, 'exchange_sums_ok': ok, 'cpu_baseline': None,
            'cpu_baseline_reason': 'rehearsal: nothing is measured'}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return 0 if ok else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)      # 100 timed steps after 20 warm-up steps (2.3 s + 0.5 s)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--batch', type=int, default=256, help='per-GPU batch (BASELINE config: 256)')
    ap.add_argument('--classes', type=int, default=100)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-events', action='store_true', help='do not record per-launch HIP events in the timed region')
    ap.add_argument('--dump-ops', default=None, help='write the per-op timing table (JSON) here')
    ap.add_argument('--infer-rois', type=int, default=1000000, help='RUN-mode leg (BASELINE configs[3]): ROIs classified at the '
                    'best batch size; 0 = the short leg only')
    ap.add_argument('--fp32-steps', type=int, default=3, help='train steps of the fp32 parity mode timed next to the bf16 headline (0 = skip)')
    ap.add_argument('--train-only', action='store_true', help='skip the secondary RUN-mode (inference) measurement: '
                    'the process then launches training kernels only, so a rocprofv3 --stats summary of it is comparable '
                    'launch for launch with the per-kernel HIP-event times')
    args = ap.parse_args()

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        # started the way the driver starts the N = 1 run, with --gpus N: become the launcher (before any GPU call)
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d: start %d ranks (or none: bench.py starts them itself)' % (args.gpus, world, args.gpus))
    if os.environ.get('IFCBK_BENCH_REHEARSE', '0') != '0':
        raise SystemExit(rehearse(args, rank, world))
    if 'IFCBK_BENCH_DEVICE' in os.environ:      # rehearsal of the N>1 path on a one-GPU box: every rank on the same device
        local = int(os.environ['IFCBK_BENCH_DEVICE'])
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get('IFCBK_BENCH_BACKEND', 'nccl')         # 'gloo' for the one-GPU rehearsal
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend)

    from ifcb_classifier_amd import graph, _lib
    from ifcb_classifier_amd.engine import Engine
    B = args.batch
    net = graph.build('inception_v3', args.classes, pretrained=False)
    eng = Engine(net, device=local, max_batch=B)
    eng.init_weights(seed=1234)
    if world > 1:
        dist.broadcast(eng.P, 0)
        eng.params_changed()
    rois, _ = synth_rois(B, 1234 + rank, eng.dev)
    eng.target[:B].copy_(torch.randint(0, args.classes, (B,), generator=torch.Generator().manual_seed(99 + rank)))
    eng.load_rois(**rois)          # (the plan is per input kind: grey ROIs arrive as the resized u8 plane)
    pl = eng.plan(B)
    do_survey = (not args.no_events) and args.steps <= 256
    use_ev = do_survey and world == 1        # events in the timed region: single-GPU runs only (the N>1 step is several programs)

    allred, exchange_name = None, None
    if world > 1:
        from ifcb_classifier_amd.dp import make_exchange
        allred, exchange_name = make_exchange(dist)      # IFCBK_DP_EXCHANGE=allreduce|rsag
    # exposed part of the exchange: an event pair around the waits behind the last backward segment, every timed step
    ex_events = []

    def ex_mark(name):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        ex_events.append(ev)

    # input pipelining (the product's Trainer does the same, neuston_net.Trainer._lookahead): the on-GPU preprocessing of batch
    # k+1 runs on a side stream into the engine's other input slot while step k computes.  Every timed step still contains
    # exactly one preprocess and one train step.  Default OFF here: with the inputs already in HBM there is no upload to hide and
    # a fifth stream next to the four program lanes costs 0.5 ms per step on this runtime (4 hardware queues per process;
    # GPU_MAX_HW_QUEUES=8 is far worse) -- measured 24.40 vs 23.92 ms.  What it buys is in `pcie_inclusive` below.
    pipelined = os.environ.get('IFCBK_BENCH_PIPELINE', '0') != '0'
    timing_on = [False]
    for tb in eng.tgt_bufs:
        tb[:B].copy_(eng.tgt_bufs[0][:B])

    def stage():
        slot, side = eng.prefetch_begin()
        with torch.cuda.stream(side):
            eng.load_rois(slot=slot, **rois)
        eng.prefetch_end(slot)

    stage_first = os.environ.get('IFCBK_BENCH_STAGE_FIRST', '0') != '0'

    def step(k=None, ev_arrs=None):
        if pipelined:
            eng.use_prefetched()
            if stage_first:
                stage()
        else:
            eng.load_rois(**rois)
        if world > 1:
            eng.train_step_ddp(B, world, allred, mark=ex_mark if timing_on[0] else None)
        else:
            eng.train_step(B, ev_slot=k, ev_arr=None if ev_arrs is None else ev_arrs[eng.in_slot])
        if pipelined and not stage_first:
            stage()          # enqueued behind the step's ~800 launches: reaches the GPU while the backward pass runs

    def op_table(nsteps):
        """per-kernel sums of the event-bracketed ops of event slots 0..nsteps-1"""
        n = pl.step.n
        ms = (C.c_float * n)()
        agg = {}
        for k in range(nsteps):
            eng.ctx.call('ifcbk_program_times', k, n, ms)
            for j in range(n):
                if ms[j] <= 0.0:
                    continue
                op = pl.step.arr[j]
                nm = C.create_string_buffer(64)
                eng.ctx.lib.ifcbk_op_kernel(C.byref(op), nm, 64)
                fl, by = C.c_double(), C.c_double()
                eng.ctx.lib.ifcbk_op_cost(C.byref(op), C.byref(fl), C.byref(by))
                key = nm.value.decode() or _lib.OP_NAMES[op.kind]
                a = agg.setdefault(key, dict(ms=0.0, flops=0.0, bytes=0.0, launches=0))
                a['ms'] += ms[j]
                a['flops'] += fl.value
                a['bytes'] += by.value
                a['launches'] += 1
        return agg

    def kernel_of(j):
        nm = C.create_string_buffer(64)
        eng.ctx.lib.ifcbk_op_kernel(C.byref(pl.step.arr[j]), nm, 64)
        return nm.value.decode() or _lib.OP_NAMES[pl.step.arr[j].kind]

    if pipelined:
        stage()
    for _ in range(args.warmup):
        step()
    # untimed survey pass: every op bracketed by HIP events, all ops back to back on ONE lane -> the per-kernel table
    # (each kernel alone on the GPU) and the dominant conv kernel.  Bracketing all ~530 ops costs ~1.6 ms per step, so the
    # timed region below brackets the dominant kernel only -- there on all lanes, i.e. as the step really runs.
    survey, ev_dom, dom, layer_rf = None, None, None, None
    if do_survey:
        NS = 3
        ev_all = eng.plan(B).step.timed(single_lane=True)     # (the plan of the input slot the warm-up left current)
        for k in range(NS):
            eng.load_rois(**rois)
            eng.train_step(B, ev_slot=k, ev_arr=ev_all)      # N>1: three LOCAL steps (no all-reduce) ...
        torch.cuda.synchronize()
        if world > 1:                                         # ... then every replica is put back on rank 0's state
            for buf in (eng.P, eng.M, eng.V, eng.RB):
                dist.broadcast(buf, 0)
            eng.params_changed()
        survey = op_table(NS)
        layer_rf = layer_roofline(eng, pl, NS)            # (reads the survey's event slots: before the timed region reuses them)
        conv = {k: v for k, v in survey.items() if k.startswith('conv_')}
        # dominant kernel = the conv kernel instantiation (the name rocprofv3 lists) with the most time per step
        dom = max(conv, key=lambda k: conv[k]['ms'])
        if use_ev:
            domops = [j for j in range(pl.step.n) if kernel_of(j) == dom]
            cur = eng.in_slot
            ev_dom = {}
            for sl in (0, 1):                                 # one event-flagged op table per input slot
                eng._select_slot(sl)
                eng.in_kind[sl] = eng.in_kind[cur]            # (both slots will hold ROIs of the same kind: the plan is per kind)
                ev_dom[sl] = eng.plan(B).step.timed(domops)
            eng._select_slot(cur)
    torch.cuda.synchronize()
    # N > 1: the single-GPU step at THIS job's program-lane count (a DP job defaults to 2 lanes, the N = 1 bench to 4), timed on
    # every rank before the DP loop -- 10 local steps, no exchange -- so that (single_gpu_same_lanes_ms / DP ms per step) separates
    # the cost of communication from the cost of the lane change in the first scaling curve.  Every replica is then put back on
    # rank 0's state.
    single_same_lanes_ms = None
    if world > 1:
        for k in range(13):
            if k == 3:
                torch.cuda.synchronize()
                dist.barrier()
                torch.cuda.synchronize()
                tl = time.perf_counter()
            eng.load_rois(**rois)
            eng.train_step(B)
        torch.cuda.synchronize()
        single_same_lanes_ms = 1e2 * (time.perf_counter() - tl)
        for buf in (eng.P, eng.M, eng.V, eng.RB):
            dist.broadcast(buf, 0)
        eng.params_changed()
        torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    timing_on[0] = True
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k if use_ev else None, ev_dom)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    timing_on[0] = False
    exposed_ms, rank_devices = None, [local]
    if world > 1:
        tt = torch.tensor([dt], device=eng.dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        exposed_ms = sum(a.elapsed_time(b) for a, b in zip(ex_events[0::2], ex_events[1::2])) / max(1, len(ex_events) // 2)
        rank_devices = [None] * world
        dist.all_gather_object(rank_devices, int(local))
    loss = float(eng.loss.item())

    # secondary number (BASELINE metric is "train + infer"): RUN-mode inference on the same ROIs -- on-GPU
    # preprocess -> eval forward (BN from running statistics) -> softmax; replicas only, no collective
    def infer_step():
        if pipelined:
            eng.use_prefetched()
            if stage_first:
                stage()
        else:
            eng.load_rois(**rois)
        p = eng.forward_eval(B)
        eng.run(p.softmax)
        if pipelined and not stage_first:
            stage()

    n_inf = 0 if args.train_only else args.steps
    for _ in range(2 if n_inf else 0):
        infer_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    ti = time.perf_counter()
    for _ in range(n_inf):
        infer_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dti = time.perf_counter() - ti
    if world > 1:
        tt = torch.tensor([dti], device=eng.dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dti = float(tt.item())

    # PCIe-inclusive rate (the boundary as the product's DataLoader feeds it: a pinned host batch per step -- one ragged u8 blob
    # plus offset / size tables, <= 23 MB): upload -> on-GPU preprocess -> train step, on one stream and with one batch of
    # look-ahead on the side stream (neuston_net.Trainer's loop).  Never the headline `value`.
    pcie = None
    if world == 1 and not args.train_only:
        hostb = {k: (v.cpu().pin_memory() if torch.is_tensor(v) else v) for k, v in rois.items()}
        up = lambda: {k: (v.to(eng.dev, non_blocking=True) if torch.is_tensor(v) else v) for k, v in hostb.items()}

        def one_stream():
            eng.load_rois(**up())
            eng.train_step(B)

        def stage_host():
            slot, side = eng.prefetch_begin()
            with torch.cuda.stream(side):
                eng.load_rois(slot=slot, **up())
            eng.prefetch_end(slot)

        def looked_ahead():
            eng.use_prefetched()
            stage_host()
            eng.train_step(B)

        pcie = {}
        for name, fn, pre in (('one_stream', one_stream, None), ('look_ahead', looked_ahead, stage_host)):
            if pre:
                pre()
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            tp = time.perf_counter()
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
            pcie[name + '_ms_per_step'] = round(1e2 * (time.perf_counter() - tp), 3)
        pcie['upload_mb_per_step'] = round(sum(v.numel() * v.element_size() for v in hostb.values() if torch.is_tensor(v)) / 1e6, 2)
        pcie['note'] = 'train step fed from a pinned host batch every step (10 timed steps each); look_ahead = the product Trainer loop'

    out = None
    if rank == 0:
        ips = world * B * args.steps / dt
        out = {
            'metric': 'train images/sec, inception_v3 299^2 IFCB ROIs', 'value': round(ips, 1), 'unit': 'images/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(1e3 * dt / args.steps, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': 'inception_v3 100-class bf16 TRAIN, batch %d per GPU, synthetic u8 ROIs h,w~U{32..299} '
                                   'resized on-GPU to 299x299 (BASELINE.json configs[1])' % B,
                       'global_batch': world * B, 'parallelism': 'dp%d' % world, 'program_lanes': eng.NL, 'IFCBK_LANES': eng.NL,
                       'weight_gradient_lanes': eng.wgrad_lane, 'optimizer': 'adam lr=1e-3',
                       'rank_devices': rank_devices,
                       'loss': 'CE + 0.4*CE_aux',
                       'input_pipeline': ('preprocess of batch k+1 on a side stream beside step k (two input slots); one preprocess + '
                                          'one train step per timed step') if pipelined else 'preprocess then step on one stream'},
            'train_tflops': round(ips * TRAIN_GFLOP_PER_IMG * 1e-3, 2),
            'mfma_frac_whole_step': round(ips * TRAIN_GFLOP_PER_IMG * 1e-3 / (world * MFMA_BF16_PEAK_TFLOPS), 4),
            'final_loss': round(loss, 4),
        }
        if world > 1:
            segs = eng.ddp_segments(pl)
            out['exchange'] = {'algorithm': exchange_name, 'backend': dist.get_backend(), 'buckets': len(segs),
                               'bucket_bytes': [int(4 * (sg[3] - sg[2])) for sg in segs],
                               'exposed_ms_per_step': round(exposed_ms, 4),
                               'note': 'fp32 sum of the flat gradient, one bucket per backward segment, launched behind the segment '
                                       'that completes it; exposed = HIP-event pair around the waits behind the last segment, rank 0, '
                                       'mean over the timed steps (IFCBK_DP_EXCHANGE=allreduce|rsag)'}
            out['single_gpu_same_lanes_ms'] = round(single_same_lanes_ms, 3)
            out['single_gpu_same_lanes_note'] = ('rank 0, 10 local train steps (preprocess + fused step, no exchange) at this job\'s %d program '
                                                 'lanes, timed before the DP loop: efficiency of the exchange alone = this / ms_per_step' % eng.NL)
            out['cpu_baseline'] = None
            out['cpu_baseline_reason'] = 'N > 1: the CPU reference is timed on rank 0 of the N = 1 run only (bench contract)'
        if pcie:
            out['pcie_inclusive'] = pcie
        if n_inf:
            out.update({'infer_images_per_s': round(world * B * n_inf / dti, 1),
                        'infer_ms_per_batch': round(1e3 * dti / n_inf, 3),
                        'infer_mfma_frac': round(world * B * n_inf / dti * 11.423e-3 / (world * MFMA_BF16_PEAK_TFLOPS), 4)})
        if do_survey:
            # N=1: the dominant kernel's launches INSIDE the timed region; N>1: its launches in the survey pass
            d = op_table(args.steps)[dom] if use_ev else survey[dom]
            ach = d['flops'] / (d['ms'] * 1e-3) / 1e12
            traffic = None
            mfma_util, prof_src = None, None

            lib_sha = lib_sha16()

            def newest(suffix):
                # the newest committed PMC summary (profiles/r<round><letter>_<suffix>.json, ordered by round number, then letter)
                # that lists this kernel AND was collected with this very build of libifcbk.so (the collectors record its hash):
                # a kernel whose body changed under an unchanged name must not inherit stale counters
                import glob
                import re

                def order(f):
                    m = re.match(r'r(\d+)([a-z]*)_', os.path.basename(f))
                    return (int(m.group(1)), m.group(2)) if m else (-1, '')
                stale = None
                for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_%s.json' % suffix)), key=order, reverse=True):
                    try:
                        j = json.load(open(f))
                        if dom in j['kernels']:
                            if j.get('lib_sha16') == lib_sha:
                                return j['kernels'][dom], os.path.basename(f), None
                            stale = stale or os.path.basename(f)
                    except Exception:
                        pass
                return None, None, stale
            ent, src, stale_t = newest('traffic')    # HBM bytes per launch from two separate --pmc passes (scripts/collect_traffic.py)
            if ent:
                traffic, prof_src = round(ent['hbm_bytes_per_launch']), src
            ent, src, stale_m = newest('mfma')       # MFMA-pipe busy fraction from its own --pmc pass (scripts/collect_mfma.py)
            if ent:
                mfma_util = round(ent['mfma_pipe_utilisation'], 4)
            out['roofline'] = {'bound': 'mfma', 'kernel': dom, 'achieved': round(ach, 2), 'peak': MFMA_BF16_PEAK_TFLOPS,
                               'unit': 'TFLOP/s', 'frac': round(ach / MFMA_BF16_PEAK_TFLOPS, 4), 'traffic': traffic,
                               'algorithmic_bytes_per_launch': round(d['bytes'] / d['launches']),
                               'avg_launch_ms': round(d['ms'] / d['launches'], 5), 'launches': d['launches'],
                               'flops_per_launch': d['flops'] / d['launches'],
                               'note': ('HIP events around every launch of this kernel in the timed region, on the lane it runs '
                                        'on: with %d program lanes another branch\'s kernel usually shares the GPU during a '
                                        'launch; isolated_* = the same launches back to back on one lane (survey pass)' % eng.NL)
                               if use_ev else 'N > 1: rank 0, untimed survey pass (3 local steps, every op bracketed, one lane)',
                               'isolated_tflops': round(survey[dom]['flops'] / (survey[dom]['ms'] * 1e-3) / 1e12, 2),
                               'isolated_avg_launch_ms': round(survey[dom]['ms'] / survey[dom]['launches'], 5),
                               'pmc_mfma_pipe_utilisation': mfma_util, 'pmc_source': prof_src,
                               'field_sources': {'measured_in_this_run': ['achieved', 'frac', 'avg_launch_ms', 'launches', 'isolated_tflops',
                                                                          'isolated_avg_launch_ms', 'flops_per_launch',
                                                                          'algorithmic_bytes_per_launch'],
                                                 'from_committed_rocprofv3_pmc_passes': ['traffic', 'pmc_mfma_pipe_utilisation'],
                                                 'pmc_profile_of_this_library_build': prof_src is not None,
                                                 'stale_profiles_ignored': [x for x in (stale_t, stale_m) if x]}}
            if dom.startswith('conv_wgrad'):
                # one weight-gradient op = the split-K MFMA kernel + its fixed-order fp32 reduction (wgrad_reduce): the event
                # bracket, avg_launch_ms and achieved cover BOTH; rocprofv3 lists them as two kernels (their averages add up)
                out['roofline']['bracket_includes'] = 'wgrad_reduce'
            conv = {k: v for k, v in survey.items() if k.startswith('conv_')}
            call = sum(v['flops'] for v in conv.values()) / (sum(v['ms'] for v in conv.values()) * 1e-3) / 1e12
            out['conv_all'] = {'achieved_tflops': round(call, 2), 'frac': round(call / MFMA_BF16_PEAK_TFLOPS, 4),
                               'ms_per_step': round(sum(v['ms'] for v in conv.values()) / 3, 3),
                               'note': 'survey pass: all ops bracketed, one lane, untimed'}
            out['ms_per_step_by_kernel'] = {k: round(v['ms'] / 3, 3) for k, v in
                                            sorted(survey.items(), key=lambda kv: -kv[1]['ms'])}
            out['layer_roofline'] = layer_rf
            n = pl.step.n
            ms = (C.c_float * n)()
            if args.dump_ops:
                rows = []
                eng.load_rois(**rois)        # one more fully bracketed single-lane step for the per-op table
                eng.train_step(B, ev_slot=0, ev_arr=eng.plan(B).step.timed(single_lane=True))
                torch.cuda.synchronize()
                eng.ctx.call('ifcbk_program_times', 0, n, ms)
                for j in range(n):
                    op = pl.step.arr[j]
                    nm = C.create_string_buffer(64)
                    eng.ctx.lib.ifcbk_op_kernel(C.byref(op), nm, 64)
                    fl, by = C.c_double(), C.c_double()
                    eng.ctx.lib.ifcbk_op_cost(C.byref(op), C.byref(fl), C.byref(by))
                    rows.append(dict(i=j, tag=pl.step.tags[j], op=_lib.OP_NAMES[op.kind], kernel=nm.value.decode(),
                                     ms=ms[j], gflop=fl.value / 1e9, mbytes=by.value / 1e6))
                with open(args.dump_ops, 'w') as f:
                    json.dump(rows, f)
        if world == 1 and not args.train_only:
            del eng
            torch.cuda.empty_cache()
            if args.infer_rois > 0:
                out['run_mode'] = run_mode_leg(args, local)
            if args.fp32_steps > 0:
                out['fp32_parity_mode'] = fp32_leg(args, local)
        if world == 1 and not args.no_cpu_baseline:
            eng = None
            torch.cuda.empty_cache()
            out['cpu_baseline'] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
