"""Walk the HIP plan node by node after one forward+backward and check every node against the node-level
CPU oracle (oracle/ops.py) on the HIP path's OWN inputs (teacher forcing).  Returns a dict of worst errors."""
import torch

from oracle import ops as O


def rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def _nchw(t, view, N):
    return t[:N, :, :, view.coff:view.coff + view.C].float().cpu().permute(0, 3, 1, 2).contiguous()


def check_plan(hip, N, mask=None, verbose=False):
    eng = hip.engine
    net = eng.net
    act = lambda v: _nchw(eng.act[v.buf.id], v, N)
    grd = lambda v: _nchw(eng.grad[v.buf.id], v, N)
    P = {k: p.detach().cpu() for k, p in hip.named_parameters()}
    G = {k: p.grad.detach().cpu() for k, p in hip.named_parameters()}
    expected = {}      # buf id -> expected total gradient (NCHW fp32)
    worst = dict(raw=0.0, raw_cp=0.0, y=0.0, dW=0.0, dW_cp=0.0, dgamma=0.0, dbeta=0.0, pool=0.0, head=0.0, dx=0.0, stats=0.0)

    relu_out = {}      # buf id -> [(coff, C, activation)]: conv(+bias)+ReLU outputs whose stored gradient is already masked in place

    def add_expected(view, g):
        b = view.buf
        if b.id not in expected:
            expected[b.id] = torch.zeros(N, b.C, b.H, b.W)
        expected[b.id][:, view.coff:view.coff + view.C] += g

    def upd(k, e, name):
        if verbose:
            print('%-8s %-34s %.3e' % (k, name, e))
        worst[k] = max(worst[k], e)

    def upd_dgamma(name, g_hip, dg32, args, gy, y_h):
        """dgamma = sum(dz * xhat) can cancel to 1e-5 of its terms (a deep random-init network's first BatchNorm): the fp32 oracle's own
        summation order then shows at the 1e-2 level.  Where the two disagree by more than 1e-4 in fp32 storage, an fp64 evaluation of the
        SAME node arbitrates: 'dgamma64' = distance of the HIP result from the fp64 value, 'dgamma64_oracle' = of the fp32 oracle's."""
        e = rel(g_hip, dg32)
        upd('dgamma', e, name)
        if O.STORAGE == 'fp32' and e > 1e-4:
            raw, gamma, beta, eps, relu, res = args
            _, dg64, _, _ = O.bn_act_bwd(raw.double(), gamma.double(), beta.double(), eps, relu, None if res is None else res.double(),
                                         gy.double(), y_for_mask=y_h.double())
            worst['dgamma64'] = max(worst.get('dgamma64', 0.0), rel(g_hip.double(), dg64))
            worst['dgamma64_oracle'] = max(worst.get('dgamma64_oracle', 0.0), rel(dg32.double(), dg64))
            if verbose:
                print('dgamma64 %-34s hip %.3e  fp32 oracle %.3e' % (name, rel(g_hip.double(), dg64), rel(dg32.double(), dg64)))

    fused = getattr(eng, 'fused_pool', {})            # conv node -> (max-pool node, index): activation never materialised
    fused_y = {}                                      # pool node -> the oracle's activation of the HIP raw output
    absorbed = getattr(eng, 'absorbed_pools', set())    # avg pools that run BEHIND their 1x1 conv in training (engine.__init__)
    for k, n in enumerate(net.nodes):
        if n.kind == 'conv':
            cp = getattr(n, 'cpool', None)
            if cp is not None:
                # the engine computed avgpool(conv1x1(x)); the reference order conv1x1(avgpool(x)) is the same linear map:
                # the oracle takes the reference order on the block input the HIP path read
                x0 = act(cp.x)
                x = O.pool_fwd('avg', x0, (cp.R, cp.S), (cp.sh, cp.sw), (cp.ph, cp.pw))
            else:
                x = act(n.x)
            if n.x.buf.is_input:
                x = x[:, :3]
            w = P[n.conv_key + '.weight']
            gamma, beta = P[n.bn_key + '.weight'], P[n.bn_key + '.bias']
            stride, pad = (n.sh, n.sw), (n.ph, n.pw)
            raw_h = eng.act[n.raw.id][:N].float().cpu().permute(0, 3, 1, 2).contiguous()
            # a commuted pool branch rounds to the storage type at a different place than the reference order (after the conv
            # and after the pool, instead of after the pool and after the conv): same count of roundings, not the same bits
            upd('raw_cp' if cp is not None else 'raw', rel(raw_h, O.conv_raw(x, w, stride, pad)), n.name)
            res = act(n.residual) if n.residual is not None else None
            y_ref, mean, var = O.bn_act_fwd(raw_h, gamma, beta, n.eps, n.relu, res)
            if n in fused:
                pn = fused[n][0]
                fused_y[pn] = y_ref
                y_h = y_ref
                gy = O.pool_bwd('max', y_ref, (pn.R, pn.S), (pn.sh, pn.sw), (pn.ph, pn.pw), grd(pn.y))
            else:
                y_h = act(n.y)
                gy = grd(n.y)
                upd('y', rel(y_h, y_ref), n.name)
            st_mean = eng.stats[n.st_off:n.st_off + n.K].cpu()
            st_is = eng.stats[n.st_off + n.st_ld:n.st_off + n.st_ld + n.K].cpu()
            upd('stats', max(rel(st_mean, mean), rel(st_is, 1.0 / torch.sqrt(var + n.eps))), n.name)
            d_raw, dg, db, dres = O.bn_act_bwd(raw_h, gamma, beta, n.eps, n.relu, res, gy, y_for_mask=y_h)
            upd_dgamma(n.name, G[n.bn_key + '.weight'], dg, (raw_h, gamma, beta, n.eps, n.relu, res), gy, y_h)
            upd('dbeta', rel(G[n.bn_key + '.bias'], db), n.name)
            need_dx = not n.x.buf.is_input
            dw, dx = O.conv_bwd(x, w, d_raw, stride, pad, need_dx)
            upd('dW_cp' if cp is not None else 'dW', rel(G[n.conv_key + '.weight'], dw), n.name)      # (same remark as raw_cp)
            if need_dx and cp is not None:
                add_expected(cp.x, O.pool_bwd('avg', x0, (cp.R, cp.S), (cp.sh, cp.sw), (cp.ph, cp.pw), dx))
            elif need_dx:
                add_expected(n.x, dx)
            if dres is not None:
                add_expected(n.residual, dres)
        elif n.kind in ('max', 'avg'):
            if n in absorbed:
                continue                          # checked together with its conv above
            x = fused_y[n] if n in fused_y else act(n.x)
            upd('pool', rel(act(n.y), O.pool_fwd(n.kind, x, (n.R, n.S), (n.sh, n.sw), (n.ph, n.pw))), n.name)
            if n not in fused_y:
                add_expected(n.x, O.pool_bwd(n.kind, x, (n.R, n.S), (n.sh, n.sw), (n.ph, n.pw), grd(n.y)))
        elif n.kind == 'cb':
            # conv (+bias) (+ReLU) without BatchNorm, nn.Linear as a 1x1 conv: ONE stored tensor; its gradient buffer holds
            # dz = dy * (y > 0) (masked in place by ifcbk_bias_relu_bwd)
            x = act(n.x)
            if n.x.buf.is_input:
                x = x[:, :3]
            w = P[n.key + '.weight']
            if w.dim() == 2:
                w = w[:, :, None, None]
            kr = n.K_real
            bias = P[n.key + '.bias'] if n.bias else None
            stride, pad = (n.sh, n.sw), (n.ph, n.pw)
            y_h = act(n.y)
            y_ref = torch.nn.functional.conv2d(x, O.bf16_round(w), bias, stride, pad)
            y_ref = O.bf16_round(torch.relu(y_ref) if n.relu else y_ref)
            upd('y', rel(y_h[:, :kr], y_ref), n.name)
            if kr != n.K:
                assert float(y_h[:, kr:].abs().max()) == 0.0, n.name       # padded output channels stay zero
            dz = grd(n.y)
            if n.relu:
                assert float((dz * (y_h <= 0)).abs().max()) == 0.0, n.name + ': gradient not masked'
                relu_out.setdefault(n.y.buf.id, []).append((n.y.coff, n.y.C, y_h))
            if bias is not None:
                upd('dbeta', rel(G[n.key + '.bias'], dz[:, :kr].sum((0, 2, 3))), n.name)
            need_dx = not n.x.buf.is_input
            dw, dx = O.conv_bwd(x, w, dz[:, :kr], stride, pad, need_dx)
            upd('dW', rel(G[n.key + '.weight'].reshape(dw.shape), dw), n.name)
            if need_dx:
                add_expected(n.x, dx)
        elif n.kind == 'drop':
            x = act(n.x)
            m = n.mask[:N].reshape(N, n.x.H, n.x.W, n.x.C).permute(0, 3, 1, 2).float().cpu() * (1.0 / (1.0 - n.p))
            upd('pool', rel(act(n.y), O.bf16_round(x * m)), n.name)
            add_expected(n.x, grd(n.y) * m)
        elif n.kind == 'flat':
            x = act(n.x)
            upd('pool', rel(act(n.y).reshape(N, -1), torch.flatten(x, 1)), n.name)
            add_expected(n.x, grd(n.y).reshape(x.shape))
        elif n.kind == 'bnr':
            # BatchNorm -> ReLU in front of a conv (densenet), on a channel slice of the block's concatenation
            x = act(n.x)
            gamma, beta = P[n.bn_key + '.weight'], P[n.bn_key + '.bias']
            y_ref, mean, var = O.bn_act_fwd(x, gamma, beta, n.eps, n.relu)
            y_h = act(n.y)
            upd('y', rel(y_h, y_ref), n.name)
            st_mean = eng.stats[n.st_off:n.st_off + n.K].cpu()
            st_is = eng.stats[n.st_off + n.st_ld:n.st_off + n.st_ld + n.K].cpu()
            upd('stats', max(rel(st_mean, mean), rel(st_is, 1.0 / torch.sqrt(var + n.eps))), n.name)
            d_x, dg, db, _ = O.bn_act_bwd(x, gamma, beta, n.eps, n.relu, None, grd(n.y), y_for_mask=y_h)
            upd_dgamma(n.name, G[n.bn_key + '.weight'], dg, (x, gamma, beta, n.eps, n.relu, None), grd(n.y), y_h)
            upd('dbeta', rel(G[n.bn_key + '.bias'], db), n.name)
            add_expected(n.x, d_x)
        elif n.kind == 'head' and not n.fc:
            # squeezenet: the pooled channels of the classifier conv ARE the logits
            x = act(n.x)
            logits = x.mean((2, 3))[:, :n.NC]
            upd('head', rel(n.logits[:N].cpu(), logits), n.name + ':logits')
            dl = n.dlogits[:N].cpu()
            dx = torch.zeros_like(x)
            dx[:, :n.NC] = (dl / float(n.HW))[:, :, None, None]
            add_expected(n.x, O.bf16_round(dx))
        elif n.kind == 'head':
            x = act(n.x)
            W, b = P[n.key + '.weight'], P[n.key + '.bias']
            m = mask[:N] if (n.dropout and mask is not None) else None
            feat, logits = O.head_fwd(x, W, b, m)
            upd('head', rel(n.logits[:N].cpu(), logits), n.name + ':logits')
            dl = n.dlogits[:N].cpu()
            dx, dW, db = O.head_bwd(x, W, b, m, dl)
            upd('head', rel(G[n.key + '.weight'], dW), n.name + ':dW')
            upd('head', rel(G[n.key + '.bias'], db), n.name + ':db')
            add_expected(n.x, dx)
    for bid, g in expected.items():
        b = net.bufs[bid]
        for coff, Cc, y_h in relu_out.get(bid, ()):
            g[:, coff:coff + Cc] *= (y_h > 0).to(g.dtype)
        upd('dx', rel(_nchw(eng.grad[bid], b.full(), N), g), b.name)
    return worst
