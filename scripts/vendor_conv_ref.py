"""dev tool (context only, not a product dependency): what torch's own backends (MIOpen for convs, hipBLASLt for the 1x1 GEMM) reach
on the benchmark's layer shapes on the same box, bf16 channels_last, forward / input gradient / weight gradient -- a yardstick for
the hand-written kernels' TF/s."""
import time
import torch
import torch.nn.functional as F

L = {'6e_7x1 192->192': (256, 192, 17, 17, 192, 7, 1, 1, (3, 0)), '6e_1x7 192->192': (256, 192, 17, 17, 192, 1, 7, 1, (0, 3)),
     '6e_1x1 group 768->768': (256, 768, 17, 17, 768, 1, 1, 1, (0, 0)), '4a_3x3 80->192': (256, 80, 73, 73, 192, 3, 3, 1, (0, 0)),
     '5c_3x3 96->96': (256, 96, 35, 35, 96, 3, 3, 1, (1, 1)), '5c_5x5 48->64': (256, 48, 35, 35, 64, 5, 5, 1, (2, 2)),
     '7c_3x3 448->384': (256, 448, 8, 8, 384, 3, 3, 1, (1, 1)), '6a_3x3s2 288->384': (256, 288, 35, 35, 384, 3, 3, 2, (0, 0))}
torch.backends.cudnn.benchmark = True


def bench(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best


for name, (N, C, H, W, K, R, S, st, pad) in L.items():
    x = torch.randn(N, C, H, W, device='cuda', dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(K, C, R, S, device='cuda', dtype=torch.bfloat16) * 0.05).contiguous(memory_format=torch.channels_last)
    y = F.conv2d(x, w, None, st, pad)
    dy = torch.randn_like(y)
    flops = 2.0 * y.numel() * C * R * S
    res = []
    try:
        t = bench(lambda: F.conv2d(x, w, None, st, pad))
        res.append('fwd %.0f us %.0f TF/s' % (t * 1e3, flops / t / 1e9))
        t = bench(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, (st, st), pad, (1, 1), False, (0, 0), 1, (True, False, False)))
        res.append('dgrad %.0f us %.0f TF/s' % (t * 1e3, flops / t / 1e9))
        t = bench(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, (st, st), pad, (1, 1), False, (0, 0), 1, (False, True, False)))
        res.append('wgrad %.0f us %.0f TF/s' % (t * 1e3, flops / t / 1e9))
    except Exception as e:        # noqa
        res.append('failed: %s' % str(e)[:80])
    print('%-24s %s' % (name, ' | '.join(res)), flush=True)
a = torch.randn(73984, 768, device='cuda', dtype=torch.bfloat16)
b = torch.randn(768, 768, device='cuda', dtype=torch.bfloat16)
t = bench(lambda: a @ b)
print('GEMM 73984x768x768 (hipBLASLt): %.0f us %.0f TF/s' % (t * 1e3, 2.0 * 73984 * 768 * 768 / t / 1e9))
a = torch.randn(8192, 8192, device='cuda', dtype=torch.bfloat16)
t = bench(lambda: a @ a)
print('GEMM 8192^3 (hipBLASLt): %.0f us %.0f TF/s' % (t * 1e3, 2.0 * 8192 ** 3 / t / 1e9))
