"""`python bench.py --gpus N` started without a launcher (the way the driver starts the N = 1 run) must start its N ranks
itself, relay rank 0's JSON line as the last stdout line and exit with the children's code (the reference
gets its processes from ``gpus=len(args.gpus)`` with no launcher on the command line, ref neuston_net.py:101-107,430-432).
CPU only: the ranks run bench.py's rehearsal leg (gloo, Engine(plan_only=True): the real bucket plan and exchange, no GPU)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(IFCBK_BENCH_REHEARSE='1', OMP_NUM_THREADS='2', **kw)
    return env


def test_launch_argv_is_the_drivers_command():
    sys.path.insert(0, ROOT)
    import bench
    argv = bench.launch_argv(8, ['--gpus', '8', '--steps', '20', '--warmup', '5'], 29511)
    assert argv[:2] == [sys.executable, '-m'] and argv[2] == 'torch.distributed.run'
    assert argv[3:11] == ['--nnodes=1', '--nproc-per-node', '8', '--master-addr', '127.0.0.1', '--master-port', '29511', BENCH]
    assert argv[11:] == ['--gpus', '8', '--steps', '20', '--warmup', '5']


@pytest.mark.parametrize('exchange', ['allreduce', 'rsag'])
def test_gpus2_without_launcher_starts_two_ranks_and_relays_one_line(exchange):
    p = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--steps', '2', '--warmup', '0'], env=_env(IFCBK_DP_EXCHANGE=exchange),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout                     # nothing but rank 0's line on stdout
    out = json.loads(lines[-1])
    assert out['n_gpus'] == 2 and out['config']['global_batch'] == 512 and out['config']['parallelism'] == 'dp2'
    assert out['config']['rank_devices'] == ['cpu:0', 'cpu:1'] and out['config']['exchange'] == exchange
    assert out['rehearsal'] is True and out['value'] is None and out['exchange_sums_ok'] is True
    assert out['cpu_baseline'] is None and out['cpu_baseline_reason']
    assert sum(out['config']['buckets']) > 24_000_000 and 2 <= len(out['config']['buckets']) <= 9
    assert 'torch.distributed.run' in p.stderr           # the launcher said what it started
    # the first scaling curve must be readable: the DP lane count and the single-GPU step at that lane count are in the line
    # (a rehearsal measures nothing, so the value is null -- the FIELD is what the driver's parser relies on)
    assert 'single_gpu_same_lanes_ms' in out and out['single_gpu_same_lanes_note']
    assert out['config']['IFCBK_LANES'] == 2 and out['config']['weight_gradient_lanes'] == 1      # WORLD_SIZE=2 -> the DP default


def test_the_launcher_takes_its_ranks_with_it_when_it_is_terminated():
    """a driver that kills `bench.py --gpus N` must not leave the N ranks running as orphans"""
    import signal
    import time
    p = subprocess.Popen([sys.executable, BENCH, '--gpus', '2', '--steps', '2', '--warmup', '0'], env=_env(IFCBK_REHEARSE_SLEEP='120'),
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT)
    pids = set()
    t0 = time.time()
    while time.time() - t0 < 120 and len(pids) < 2:          # wait until both ranks exist (children of torch.distributed.run)
        time.sleep(1.0)
        out = subprocess.run(['ps', '-eo', 'pid,args'], stdout=subprocess.PIPE, text=True).stdout
        pids = {int(ln.split()[0]) for ln in out.splitlines() if BENCH in ln and 'torch.distributed.run' not in ln
                and int(ln.split()[0]) != p.pid}
    assert len(pids) >= 2, 'the ranks never started'
    p.send_signal(signal.SIGTERM)
    p.wait(timeout=60)
    assert p.returncode != 0
    time.sleep(2.0)
    alive = [q for q in pids if os.path.exists('/proc/%d' % q) and 'Z' not in open('/proc/%d/stat' % q).read().split(')')[-1].split()[0]]
    assert not alive, 'orphaned ranks: %s' % alive


def test_a_failing_rank_fails_the_launcher():
    p = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--steps', '1', '--warmup', '0'], env=_env(IFCBK_DP_EXCHANGE='bogus'),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, cwd=ROOT)
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.splitlines() if ln.strip().startswith('{')]
