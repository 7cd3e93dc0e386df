"""Plan/executor for one backbone on one GPU: owns the flat parameter / gradient / optimizer-state buffers,
the NHWC activation and gradient buffers, and the op tables that ``ifcbk_run_program`` launches.

No autograd, no tracing: the graph is static (``graph.py``), so forward, backward and update are three
pre-built op tables; one FFI crossing launches a whole step.  PyTorch is used only for device memory and
streams.  The HIP library is mandatory -- there is no CPU fallback in this module.
"""
import ctypes as C
import math
import os

import torch

from . import _lib
from ._lib import Op, ConvDesc, BnDesc, PoolDesc, HeadDesc, RoiDesc


# environment switches the library's conv dispatch reads per launch (csrc/conv_igemm.hip, conv_big.hip, conv_flat.hip, conv_wgrad*.hip)
_DISPATCH_SWITCHES = ('IFCBK_FWD_LANES', 'IFCBK_CONV_PP3', 'IFCBK_CONV_PP3_GRID', 'IFCBK_WGRAD_LANE', 'IFCBK_WGRAD_GROUP', 'IFCBK_WGRAD_GROUP_MINKH', 'IFCBK_CONV_BIG', 'IFCBK_CONV_SLAB', 'IFCBK_CONV_BIG_MT', 'IFCBK_CONV_BIG_TN', 'IFCBK_CONV_FLAT', 'IFCBK_CONV_NT',
                      'IFCBK_CONV_WM', 'IFCBK_CONV_MQ', 'IFCBK_CONV_WS', 'IFCBK_CONV_WS_TILES', 'IFCBK_CONV_ROWS', 'IFCBK_WGRAD_PP',
                      'IFCBK_WGRAD_PP_KH', 'IFCBK_WGRAD_COLS', 'IFCBK_WGRAD_STEM', 'IFCBK_WGRAD_ROUNDS')


def _dispatch_env():
    return tuple(os.environ.get(k) for k in _DISPATCH_SWITCHES)


def dtype_is_bf16(eng):
    return eng.dtype == 'bf16'


def _vp(t, byte_off=0):
    return None if t is None else C.c_void_p(t.data_ptr() + byte_off)


class OpList:
    """ops + tags + scheduling metadata: (lane, reads, writes) per op.  A resource is (key..., lo, hi): a channel range
    of a tensor.  An op without annotations is a full barrier on lane 0 (loss, head, Adam, pack ...)."""

    def __init__(self):
        self.ops = []
        self.tags = []
        self.meta = []

    def add(self, kind, tag, p=(), i=(), f=(), flags=0, conv=None, bn=None, pool=None, head=None, lane=0, reads=None,
            writes=None):
        o = Op()
        o.kind, o.flags = kind, flags
        self.meta.append((lane, reads, writes))
        for k, v in enumerate(p):
            o.p[k] = v if (v is None or isinstance(v, int)) else v.value
        for k, v in enumerate(i):
            o.i[k] = int(v)
        for k, v in enumerate(f):
            o.f[k] = float(v)
        if conv is not None:
            o.u.conv = conv
        elif bn is not None:
            o.u.bn = bn
        elif pool is not None:
            o.u.pool = pool
        elif head is not None:
            o.u.head = head
        self.ops.append(o)
        self.tags.append(tag)
        return o

    def extend(self, other):
        self.ops.extend(other.ops)
        self.tags.extend(other.tags)
        self.meta.extend(other.meta)

    def slice(self, b0, b1):
        sub = OpList()
        sub.ops, sub.tags, sub.meta = self.ops[b0:b1], self.tags[b0:b1], self.meta[b0:b1]
        return sub

    def freeze(self):
        arr = (Op * len(self.ops))(*self.ops)
        for k, (lane, wait) in enumerate(schedule_lanes(self.meta)):
            arr[k].flags = (arr[k].flags & 0xff) | (lane << 8) | (wait << 12)
        return arr


def _overlap(a, b):
    return a[:-2] == b[:-2] and a[-2] < b[-1] and b[-2] < a[-1]


def schedule_lanes(meta):
    """(lane, wait mask) per op from the static data flow: an op waits for the lanes that hold an unfinished producer of
    something it reads, or an unfinished reader / writer of something it writes.  A wait covers everything queued on the
    waited lane so far (and, transitively, whatever that lane had waited for), which is remembered to skip redundant waits."""
    NL = 8
    count = [0] * NL                              # ops queued per lane
    synced = [[0] * NL for _ in range(NL)]        # synced[L][j]: lane L is ordered after the first synced[L][j] ops of lane j
    table = {}                                    # resource -> [writer (lane, pos) or None, readers [(lane, pos)]]
    star = ('*', 0, 1)
    out = []
    for lane, reads, writes in meta:
        if reads is None and writes is None:
            lane, reads, writes = 0, [star], [star]           # barrier
        else:
            reads, writes = list(reads or ()) + [star], list(writes or ())
        deps = set()
        for r in reads:
            for res, (w, _rd) in table.items():
                if w is not None and _overlap(r, res):
                    deps.add(w)
        for wres in writes:
            for res, (w, rd) in table.items():
                if _overlap(wres, res):
                    if w is not None:
                        deps.add(w)
                    deps.update(rd)
        wait = 0
        for (lj, pj) in deps:
            if lj != lane and pj >= synced[lane][lj]:
                wait |= 1 << lj
        for lj in range(NL):
            if wait >> lj & 1:
                synced[lane][lj] = count[lj]
                for k in range(NL):
                    if k != lane:
                        synced[lane][k] = max(synced[lane][k], synced[lj][k])
        me = (lane, count[lane])
        count[lane] += 1
        for r in reads:
            table.setdefault(r, [None, []])[1].append(me)
        for wres in writes:
            for res in [res for res in table if _overlap(wres, res)]:
                table[res] = [me, []]
            table[wres] = [me, []]
        out.append((lane, wait))
    return out


class Program:
    def __init__(self, oplist):
        self.tags = list(oplist.tags)
        self.n = len(oplist.ops)
        self.arr = oplist.freeze()
        self.lanes = sorted({m[0] for m in oplist.meta if not (m[1] is None and m[2] is None)} | {0})

    def find(self, kind):
        return [k for k in range(self.n) if self.arr[k].kind == kind]

    def timed(self, which=None, single_lane=False):
        """a copy of the op table in which the ops `which` (indices; None = all) carry flags bit 7: ifcbk_run_program_ev
        brackets exactly those with HIP events.  single_lane: drop the lane / wait bits, i.e. run every op back to back
        on the caller's stream (per-kernel durations without another lane's kernel sharing the GPU)"""
        arr = (Op * self.n)(*self.arr)
        for k in (range(self.n) if which is None else which):
            arr[k].flags |= 0x80
        if single_lane:
            for k in range(self.n):
                arr[k].flags &= 0xff
        return arr


# The largest activation an engine may hold: that of the largest batch an eval program was verified on, bit for bit against its
# 256-image parts over every activation (scripts/big_batch_bisect.py: 4,096 inception_v3 images, 147 x 147 x 64 bf16 each).
VERIFIED_ACTIVATION_BYTES = 4096 * 147 * 147 * 64 * 2


def _index_bytes():
    # development switch (scripts/big_batch_bisect.py): another cap, to look beyond the verified one
    e = os.environ.get('IFCBK_DEV_INDEX_GIB')
    return int(float(e) * (1 << 30)) if e else VERIFIED_ACTIVATION_BYTES


class Engine:
    """Device state of one model replica."""

    @staticmethod
    def capacity_limit(net, dtype='bf16'):
        """the largest batch one engine serves (see __init__: largest activation <= VERIFIED_ACTIVATION_BYTES)"""
        per_img = max(b.H * b.W * b.C for b in net.bufs) * (2 if dtype == 'bf16' else 4)
        return max(1, _index_bytes() // per_img)

    def __init__(self, net, device=0, max_batch=32, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, dtype='bf16', optimizer='adam',
                 momentum=0.0, plan_only=False, train_batch=None, dp_world=None):
        # dp_world: world size of the data-parallel job this replica belongs to (None: WORLD_SIZE of the launcher, else an
        # initialised torch.distributed group, else 1) -- it picks the program-lane default, and train_step_ddp checks it
        self._dp_world_arg = None if dp_world is None else int(dp_world)
        # plan_only: build buffers on the host and the op tables only (no HIP context, nothing can run) -- the CPU tests
        # of the data-parallel bucket plan read the REAL backward list of a network this way
        self.plan_only = bool(plan_only)
        if not self.plan_only and not torch.cuda.is_available():
            raise RuntimeError('ifcb_classifier_amd needs a HIP device (MI355X); none is visible and there is no CPU path')
        if str(optimizer).lower() not in ('adam', 'sgd'):
            raise ValueError("optimizer must be 'adam' (the reference's only behaviour, neuston_models.py:63-64) or 'sgd'")
        self.optimizer, self.momentum = str(optimizer).lower(), float(momentum)
        self.net = net
        if dtype not in ('bf16', 'fp32'):
            raise ValueError("dtype must be 'bf16' (performance) or 'fp32' (parity mode)")
        self.dtype = dtype
        self.tdtype = torch.bfloat16 if dtype == 'bf16' else torch.float32
        self.esize = 2 if dtype == 'bf16' else 4
        self.cdtype = _lib.BF16 if dtype == 'bf16' else _lib.F32
        if self.plan_only:
            self.dev = torch.device('cpu')
            self.ctx = _lib.PlanOnlyContext()
        else:
            self.dev = torch.device('cuda', device)
            torch.cuda.set_device(self.dev)
            self.ctx = _lib.Context(device)
        # The conv kernels address every tensor through a 32-bit buffer descriptor (2 GiB window): the largest per-image tensor
        # (inception_v3: 147x147x64 bf16 = 2.77 MB) bounds the images ONE LAUNCH can take (776 in bf16, 388 in fp32).  Programs
        # without batch statistics (the eval forward) go beyond it: the library cuts such a convolution into launches over image
        # groups (csrc/conv_igemm.hip conv_fwd_impl), so the engine's capacity is what was asked for; a TRAINING step beyond the
        # window is refused (BatchNorm statistics and the split-K weight gradient are per launch).
        self.requested_batch = int(max_batch)
        per_img = max(b.H * b.W * b.C for b in net.bufs) * (2 if dtype == 'bf16' else 4)
        self.window_batch = max(1, ((1 << 31) - 1) // per_img)
        # ... and an eval batch beyond that window is verified bit for bit against its parts up to 4,096 inception images, every
        # activation compared (scripts/big_batch_bisect.py; tests/test_gpu_model.py at 1024 and across the 2 GiB offset of the stem
        # output).  Round 4 had capped this at 3 GiB after batch 1536 faulted and batch 2048 returned wrong probabilities: the u8
        # stem kernel sign-extended the low half of its 64-bit row offset (readfirstlane returns int), so the rows of images
        # >= 1,511 were written 4 GiB in front of the tensor -- fixed in csrc/conv_stem_u8.hip.  The engine still refuses a
        # capacity beyond what was verified instead of finding out on the device; RUN chunks a larger --batch itself.
        self.index_batch = max(1, _index_bytes() // per_img)
        if int(max_batch) > self.index_batch:
            raise RuntimeError('max_batch %d: the largest activation of %s would exceed the verified %.1f GiB; batches beyond %d '
                               'images are not supported in one program (run them in chunks)'
                               % (int(max_batch), net.name, _index_bytes() / (1 << 30), self.index_batch))
        self.max_batch = int(max_batch)
        # capacity of the TRAINING-side buffers (activation gradients, d(raw) scratch, pool arg-max, split-K workspace): a training
        # step beyond the window is refused anyway, so they never need more than it; an inference-only engine (neuston_net RUN:
        # train_batch=1) keeps to activations -- 'RUN --batch 2048' must not allocate gradients for 2048 images
        self.train_batch = max(1, min(self.max_batch, self.window_batch if train_batch is None else int(train_batch)))
        self.lr, self.betas, self.eps = lr, betas, eps
        self.step_count = 0
        self.packed = False
        self.eval_stats_ready = False
        self.dropout_seed = 0x1234
        self.fuse_siblings = os.environ.get('IFCBK_FUSE_SIBLINGS', '1') != '0'
        self.dropout_calls = 0
        self.external_mask = None
        self._plans = {}
        self._alloc_params()
        self._alloc_acts()

    # ------------------------------------------------------------------ parameters
    def _alloc_params(self):
        net, dev = self.net, self.dev
        off = 0
        self.poff = {}
        self.palloc = {}          # flat offset -> padded element count of the tensor stored there
        for key, shape, kind, node in net.params:
            n = na = int(math.prod(shape))
            if getattr(node, 'kind', '') == 'cb' and node.K != node.K_real:
                na = n // shape[0] * node.K       # output channels padded to a whole 16-byte chunk: the extra rows stay zero
            self.poff[key] = (off, n, shape, kind, node)
            self.palloc[off] = (na + 3) // 4 * 4
            off += (na + 3) // 4 * 4          # keep every tensor 16-byte aligned
        self.nparam_padded = off
        self.cbias_after = {}
        for key, (o, n, shape, kind, node) in self.poff.items():
            if kind == 'bn_cbias':
                self.cbias_after[self.poff[node.bn_key + '.weight'][0]] = [o]
        self.P = torch.zeros(off, dtype=torch.float32, device=dev)
        self.G = torch.zeros(off, dtype=torch.float32, device=dev)
        self.M = torch.zeros(off, dtype=torch.float32, device=dev)
        self.V = torch.zeros(off, dtype=torch.float32, device=dev)
        self.pviews, self.gviews = {}, {}
        for key, (o, n, shape, kind, node) in self.poff.items():
            if kind == 'conv':
                K, Cw, R, S = shape
                self.pviews[key] = self.P[o:o + n].view(K, R, S, Cw).permute(0, 3, 1, 2)   # OIHW view of KRSC memory
                self.gviews[key] = self.G[o:o + n].view(K, R, S, Cw).permute(0, 3, 1, 2)
            else:
                self.pviews[key] = self.P[o:o + n].view(shape)
                self.gviews[key] = self.G[o:o + n].view(shape)
        # BN running statistics
        boff = 0
        self.boff = {}
        for key, shape, node in net.buffers:
            self.boff[key] = (boff, shape[0])
            boff += shape[0]
        self.RB = torch.zeros(boff, dtype=torch.float32, device=dev)
        self.bviews = {k: self.RB[o:o + n] for k, (o, n) in self.boff.items()}
        for k, v in self.bviews.items():
            if k.endswith('running_var'):
                v.fill_(1.0)
        convs = [n for n in net.nodes if getattr(n, 'kind', '') == 'conv']
        self.convs = convs
        self.plains = [n for n in net.nodes if n.kind == 'cb']       # conv (+bias) (+ReLU) without BatchNorm, nn.Linear layers
        self.bnrs = [n for n in net.nodes if n.kind == 'bnr']        # BatchNorm -> ReLU in front of a conv (densenet)
        self.bn_nodes = [n for n in net.nodes if n.kind in ('conv', 'bnr')]
        self.bn_index = {n: k for k, n in enumerate(self.bn_nodes)}
        self.nbt = torch.zeros(len(self.bn_nodes), dtype=torch.int64, device=dev)        # num_batches_tracked (all BNs)
        self.ones = torch.ones(max([n.K for n in self.plains] + [8]), dtype=torch.float32, device=dev)
        # bf16 shadows + per-BN statistics
        # horizontal fusion: sibling 1x1 convs that read the same tensor become ONE GEMM (filters concatenated along K)
        # in the training forward / dgrad / wgrad; their BatchNorms stay per branch on channel slices.
        class Group:
            pass
        self.groups = []
        bykey = {}
        # An Inception block's pool branch is avgpool3x3(s1,p1) -> 1x1 conv -> BN -> ReLU.  Both operators are linear and the
        # conv has no bias, so conv1x1(avgpool(x)) == avgpool(conv1x1(x)) (zero padding included: count_include_pad averages
        # zeros, and conv1x1(0) = 0).  In TRAINING the engine runs the conv first -- as one more member of the block's
        # sibling GEMM on x -- and pools its pf = 32..192 output channels instead of the block's 192..2048 input channels:
        # the pool's traffic shrinks 4-10x in both directions and the branch needs no conv launches of its own.  BatchNorm
        # then normalises the pooled tensor, whose batch statistics come from ifcbk_bn_stats; in eval mode the pool applies the
        # folded BatchNorm affine + ReLU in its epilogue (ifcbk_avgpool3x3_affine).
        self.commute_pool = os.environ.get('IFCBK_COMMUTE_POOL', '1') != '0' and self.fuse_siblings
        self.eval_groups = os.environ.get('IFCBK_EVAL_GROUPS', '1') != '0'      # eval forward: sibling 1x1 convs as one GEMM
        nread = {}
        for m in net.nodes:
            for v in ((m.x, getattr(m, 'residual', None)) if m.kind == 'conv' else (m.x,)):
                if v is not None:
                    nread[v.buf.id] = nread.get(v.buf.id, 0) + 1
        for n in convs:
            n.cpool = None
            if (self.commute_pool and n.R == 1 and n.S == 1 and n.sh == 1 and n.sw == 1 and n.ph == 0 and n.pw == 0
                    and n.x.is_full and n.residual is None and n.relu and not n.aux and nread.get(n.x.buf.id, 0) == 1):
                prod = [m for m in net.nodes if m.kind == 'avg' and m.y.buf.id == n.x.buf.id]
                if (len(prod) == 1 and (prod[0].R, prod[0].S, prod[0].sh, prod[0].sw, prod[0].ph, prod[0].pw) == (3, 3, 1, 1, 1, 1)
                        and prod[0].x.is_full and prod[0].y.is_full and not prod[0].x.buf.is_input and not prod[0].aux):
                    n.cpool = prod[0]
        if self.fuse_siblings:
            for n in convs:
                n.group = None
                src = n.cpool.x if n.cpool is not None else n.x
                if (n.R == 1 and n.S == 1 and n.sh == 1 and n.sw == 1 and n.ph == 0 and n.pw == 0 and src.is_full
                        and not src.buf.is_input and n.residual is None and n.relu):
                    bykey.setdefault(src.buf.id, []).append(n)
        for members in bykey.values():
            plain = [m for m in members if m.cpool is None]
            if not (2 <= len(members) <= 4) or not plain:
                for m in members:
                    m.cpool = None                      # no sibling GEMM to join: keep the reference order
                members = plain
            members = plain + [m for m in members if m.cpool is not None]      # the first member launches the group's GEMMs
            if 2 <= len(members) <= 4:
                g = Group()
                g.members, g.x = members, members[0].x
                g.Ktot = sum(m.K for m in members)
                off = 0
                for m in members:
                    m.group, m.koff = g, off
                    off += m.K
                self.groups.append(g)
        for n in convs:
            if not hasattr(n, 'group'):
                n.group = None
            if n.group is None:
                n.cpool = None
        self.absorbed_pools = {n.cpool for n in convs if n.cpool is not None}
        soff = 0
        stoff = 0
        done = set()
        for n in convs:
            g = n.group
            if g is None:
                n.w_off = soff
                soff += n.K * n.R * n.S * n.x.C
                n.wT_off = soff
                soff += n.K * n.R * n.S * n.x.C
                n.wT_ld = 0
                n.st_off, n.st_ld = stoff, n.K
                stoff += 6 * n.K          # mean, invstd, scale, shift, eval_scale, eval_shift
            elif id(g) not in done:
                done.add(id(g))
                Cin = g.x.C
                g.w_off = soff                      # [Ktot][Cin]: member rows are contiguous
                soff += g.Ktot * Cin
                g.wT_off = soff                     # [Cin][Ktot]: members own column slices
                soff += g.Ktot * Cin
                g.st_off = stoff                    # 6 arrays of Ktot
                stoff += 6 * g.Ktot
                for m in g.members:
                    m.w_off = g.w_off + m.koff * Cin
                    m.wT_off = g.wT_off + m.koff
                    m.wT_ld = g.Ktot
                    m.st_off, m.st_ld = g.st_off + m.koff, g.Ktot
        for n in self.plains:
            n.group = None
            n.w_off = soff
            soff += n.K * n.R * n.S * n.x.C
            n.wT_off = soff
            soff += n.K * n.R * n.S * n.x.C
            n.wT_ld = 0
        for n in self.bnrs:
            n.group = None
            n.st_off, n.st_ld = stoff, n.K
            stoff += 6 * n.K
        self.Wsh = torch.zeros(soff, dtype=self.tdtype, device=dev)
        self.stats = torch.zeros(stoff, dtype=torch.float32, device=dev)

    def init_weights(self, seed=None):
        """[TV] initialisation: truncated normal (inception) / kaiming fan_out (resnet); the replaced
        ``fc`` heads use ``nn.Linear``'s default init (neuston_models.py:25-26,39)."""
        g = torch.Generator(device='cpu')
        if seed is None:        # draw from torch's global RNG so that seed_everything(--seed) makes the init reproducible
            seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
        g.manual_seed(int(seed))
        incep = self.net.name == 'inception_v3'
        for key, (o, n, shape, kind, node) in self.poff.items():
            v = self.pviews[key]
            how = getattr(node, 'init', None)
            if kind in ('conv', 'lin_w') and how is not None:
                # [TV] alexnet: torch defaults; vgg: kaiming_normal(fan_out) convs, N(0, 0.01) Linears, zero biases; squeezenet:
                # kaiming_uniform convs, zero biases; densenet: kaiming_normal (fan_in) convs; the layers the reference replaces
                # (neuston_models.py:27-42) keep torch's default init
                rf = shape[2] * shape[3] if len(shape) == 4 else 1
                fan_in, fan_out = shape[1] * rf, shape[0] * rf
                w = torch.empty(shape)
                if how == 'default':
                    w.uniform_(-1.0 / math.sqrt(fan_in), 1.0 / math.sqrt(fan_in), generator=g)
                elif how == 'kaiming_uniform':
                    w.uniform_(-math.sqrt(6.0 / fan_in), math.sqrt(6.0 / fan_in), generator=g)
                elif how == 'kaiming_out':
                    w.normal_(0, math.sqrt(2.0 / fan_out), generator=g)
                elif how == 'kaiming_in':
                    w.normal_(0, math.sqrt(2.0 / fan_in), generator=g)
                elif how == 'normal01':
                    w.normal_(0, 0.01, generator=g)
                else:
                    raise ValueError(how)
                v.copy_(w.to(self.dev))
            elif kind == 'cbias':
                wshape = self.poff[key[:-4] + 'weight'][2]
                fan_in = int(math.prod(wshape[1:]))
                if node.init == 'default':
                    v.copy_(((torch.rand(shape, generator=g) * 2 - 1) / math.sqrt(fan_in)).to(self.dev))
                else:
                    v.zero_()
            elif kind == 'bn_cbias':
                v.zero_()
            elif kind == 'conv':
                w = torch.empty(shape)
                if incep:
                    std = 0.01 if key.startswith('AuxLogits.conv1') else 0.1
                    torch.nn.init.trunc_normal_(w, 0.0, std, -2 * std, 2 * std, generator=g)
                else:
                    fan_out = shape[0] * shape[2] * shape[3]
                    w.normal_(0, math.sqrt(2.0 / fan_out), generator=g)
                v.copy_(w.to(self.dev))
            elif kind == 'bn_w':
                v.fill_(1.0)
            elif kind == 'bn_b':
                v.zero_()
            elif kind == 'fc_w':
                bound = 1.0 / math.sqrt(shape[1])
                v.copy_(((torch.rand(shape, generator=g) * 2 - 1) * bound).to(self.dev))
            elif kind == 'fc_b':
                fan_in = self.poff[key[:-4] + 'weight'][2][1]
                bound = 1.0 / math.sqrt(fan_in)
                v.copy_(((torch.rand(shape, generator=g) * 2 - 1) * bound).to(self.dev))
        self.params_changed()

    def params_changed(self):
        self.packed = False
        self.eval_stats_ready = False

    # ------------------------------------------------------------------ activations
    def _alloc_acts(self):
        net, dev, N = self.net, self.dev, self.max_batch
        Nt = self.train_batch
        bf = self.tdtype
        self.act = {}
        self.grad = {}
        grouped_raw = {m.raw.id: m for g in self.groups for m in g.members if m.cpool is None}
        for b in net.bufs:
            if b.id in grouped_raw:
                continue                              # lives inside the group's merged raw tensor
            self.act[b.id] = torch.zeros(N, b.H, b.W, b.C, dtype=bf, device=dev)
        for g in self.groups:
            g.raw = torch.zeros(N, g.x.H, g.x.W, g.Ktot, dtype=bf, device=dev)
            for m in g.members:
                if m.cpool is None:
                    self.act[m.raw.id] = g.raw[..., m.koff:m.koff + m.K]        # strided view (tests / debugging)
                # a commuted pool branch: its slice of g.raw is the UNPOOLED conv output, m.raw (own buffer) the pooled one
        need_grad = set()
        for n in net.nodes:
            if getattr(n, 'kind', '') == 'conv':
                need_grad.add(n.y.buf.id)
                if n.residual is not None:
                    need_grad.add(n.residual.buf.id)
            elif n.kind in ('max', 'avg', 'cb', 'drop', 'flat', 'bnr'):
                need_grad.add(n.y.buf.id)
        need_grad.discard(net.input.id)
        for bid in need_grad:
            b = net.bufs[bid]
            if b.name.endswith(':raw'):
                continue
            self.grad[bid] = torch.zeros(Nt, b.H, b.W, b.C, dtype=bf, device=dev)
        # program lanes (branch-parallel streams) of the training programs: 4 on a single GPU, 2 in a data-parallel job (world > 1)
        # until a run with one process per GPU on >= 2 GPUs shows 4 is no slower THERE.  Round 1 saw 1.3-6.5 s per step with 3-4
        # lanes on a 2-rank rehearsal (two processes' 5 streams each on ONE device's hardware queues); round 3 measured the DP step
        # on ProcessGroupNCCL at world 1 (scripts/dp_lanes.py: 2 / 3 / 4 lanes = 25.57 / 24.15 / 23.98 ms, no stall) -- but a
        # world-1 all-reduce launches no RCCL ring kernels, so that run cannot show the lanes competing with them for CUs and
        # hardware queues (main + 4 lanes + RCCL = 6 streams on 4 queues).  No multi-GPU node was available to this build: the
        # conservative count ships, IFCBK_LANES overrides it, bench.py prints the count in `config.program_lanes`.
        # world size: the explicit argument, else the launcher's WORLD_SIZE (set before any rank builds its engine, so every rank
        # agrees whether or not init_process_group has run yet), else an initialised process group
        dp_world = self._dp_world_arg
        if dp_world is None:
            try:
                dp_world = int(os.environ.get('WORLD_SIZE', '0')) or None
            except ValueError:
                dp_world = None
        if dp_world is None:
            dp_world = 1
            try:
                import torch.distributed as _d
                if _d.is_available() and _d.is_initialized():
                    dp_world = _d.get_world_size()
            except Exception:
                dp_world = 1
        self.dp_world = max(1, int(dp_world))
        dp_world = self.dp_world
        self.NL = max(1, min(8, int(os.environ.get('IFCBK_LANES', '2' if dp_world > 1 else '4'))))
        self.NL_eval = max(1, min(self.NL, int(os.environ.get('IFCBK_LANES_EVAL', '2'))))     # ... of the eval forward (measured best)
        # hipGraph replay of the static programs.  Measured on MI355X (B=256): the eval forward replays 1.8 % faster than its
        # launch list (6.73 vs 6.85 ms); the train fwd+bwd graph is 4 % SLOWER (28.7 vs 27.6 ms per step: the graph's own
        # branch scheduling loses to the lane assignment below) -- so the default is 'eval'.  IFCBK_GRAPH=0 | eval | all
        gm = os.environ.get('IFCBK_GRAPH', 'eval')
        # (any lane count can be captured: ctx.hip records lane-to-lane edges through the origin stream, see run_lanes)
        self.graph_eval = gm not in ('0', 'off', 'none')
        self.graph_train = gm in ('1', 'all', 'train')
        self.wgrad_side_lane = os.environ.get('IFCBK_WGRAD_SIDE', '0') != '0' and self.NL > 1
        # IFCBK_WGRAD_LANE=1 (round 4): EVERY weight gradient on the last lane, the branch chains on the others, and every layer keeps
        # its d(raw) in a buffer of its own (4.6 GB at batch 256) -- the backward critical path is then BN-backward -> input gradient
        # -> BN-backward ..., and the MFMA-bound weight gradients (6.7 ms of a step when each runs alone) fill in beside the
        # HBM-bound BatchNorm / pool kernels instead of standing in front of every input gradient on its lane
        # default: ONE weight-gradient lane (the last) when there are at least 3 lanes.  Measured at batch 256, same box, ms/step:
        # 4 chain lanes 22.67 | 3 chains + 1 weight-gradient lane 21.72 | the same with that lane's stream at low priority 21.60 (off, below) |
        # 2 + 1 lanes 22.34 | FIVE streams (4 + 1, 3 + 2) 27.7, six 31.8: this runtime has four hardware queues per process.
        # Two lanes (the data-parallel default): 2 chains 22.90 | 1 chain + 1 weight-gradient lane 22.52.
        self.wgrad_lane = int(os.environ.get('IFCBK_WGRAD_LANE', '1' if self.NL >= 2 else '0'))       # number of weight-gradient lanes
        if self.NL - self.wgrad_lane < 1 or self.wgrad_side_lane:
            self.wgrad_lane = 0
        # (round 4's least-priority stream for that lane -- 0.0-0.1 ms per step -- is gone: round 5)
        if not self.plan_only:
            self.ctx.call('ifcbk_ctx_set_lanes', self.NL)     # one workspace arena per lane in use (was: always 8)
        max_raw = max([n.P * n.Q * n.K for n in self.convs] + [8])
        # d(raw) scratch: per lane; two per lane when the weight gradient runs on the side lane (it keeps reading one while
        # the next node's BN backward already fills the other)
        self.draw = [torch.zeros(Nt * max_raw, dtype=bf, device=dev) for _ in range(self.NL * (2 if self.wgrad_side_lane else 1))]
        gmax = max([g.x.H * g.x.W * g.Ktot for g in self.groups] + [0])
        self.draw_group = torch.zeros(max(1, Nt * gmax), dtype=bf, device=dev)
        mb = max([self.ctx.lib.ifcbk_conv2d_fwd_mblocks(C.byref(self._conv_desc(n, N))) * 2 * n.K for n in self.convs] +
                 [self.ctx.lib.ifcbk_conv2d_fwd_mblocks(C.byref(self._group_desc(g, N))) * 2 * g.Ktot for g in self.groups] +
                 [self.ctx.lib.ifcbk_bn_stats_rows(N * n.x.H * n.x.W) * 2 * n.K for n in self.bnrs] + [16])
        self.bn_part = [torch.zeros(mb, dtype=torch.float32, device=dev) for _ in range(self.NL)]
        self.argmax = {}
        for k, n in enumerate(net.nodes):
            if n.kind == 'max':
                self.argmax[k] = torch.zeros(Nt, n.P, n.Q, n.x.C, dtype=torch.uint8, device=dev)
        self.heads = [n for n in net.nodes if n.kind == 'head']
        for h in self.heads:
            h.feat = torch.zeros(N, h.C, dtype=torch.float32, device=dev)
            h.logits = torch.zeros(N, h.NC, dtype=torch.float32, device=dev)
            h.dlogits = torch.zeros(N, h.NC, dtype=torch.float32, device=dev)
            h.mask = torch.ones(N, h.C, dtype=torch.uint8, device=dev) if h.dropout else None
        self.drops = [n for n in net.nodes if n.kind == 'drop']
        for n in self.drops:
            n.mask = torch.ones(N, n.x.H * n.x.W * n.x.C, dtype=torch.uint8, device=dev)
        self.probs = torch.zeros(N, net.NC, dtype=torch.float32, device=dev)
        # Two input slots (network input tensor + labels): while a step runs on slot s, the next batch is uploaded and
        # preprocessed into slot 1-s on a side stream (prefetch_begin / prefetch_end / use_prefetched).  Programs are built per
        # slot (only the pointers of the first conv, its weight gradient and the loss ops differ).
        self.tgt_bufs = [torch.zeros(N, dtype=torch.int64, device=dev) for _ in range(2)]
        self.in_bufs = [self.act[net.input.id], torch.zeros_like(self.act[net.input.id])]
        self.in_slot = 0
        # Grey ROIs into inception's Conv2d_1a: the resize writes only the u8 plane and the stem conv reads it directly
        # (csrc/conv_stem_u8.hip) -- the [N,S,S,8] tensor of the slot is then not written at all.  in_kind[slot]: 'nhwc' (the dense
        # tensor is current: load_input_nchw, RGB images) or 'u8' (the plane is current); programs are built per (slot, kind).
        self.stem_u8 = None
        first = [n for n in self.convs if n.x.buf.is_input]
        if (os.environ.get('IFCBK_STEM_U8', '1') != '0' and len(first) == 1 and first[0].group is None
                and self.ctx.lib.ifcbk_stem_u8_rows(C.byref(self._conv_desc(first[0], N))) > 0):
            self.stem_u8 = first[0]
            self.in_u8 = [torch.zeros(N, net.S, net.S, dtype=torch.uint8, device=dev) for _ in range(2)]
            self.in_ab = [torch.zeros(6, dtype=torch.float32, device=dev) for _ in range(2)]
            self._in_ab_host = [None, None]
        self.in_kind = ['nhwc', 'nhwc']
        self.pre_stream = self.pre_ctx = None
        self.ev_ready, self.ev_free, self.prefetched = [None, None], [None, None], None
        self.loss = torch.zeros(1, dtype=torch.float32, device=dev)
        self.loss_sum = torch.zeros(1, dtype=torch.float32, device=dev)
        # workspace: wgrad split-K slabs / bn_bwd partials
        ws = 1 << 20
        for n in self.convs:
            d = self._conv_desc(n, Nt)
            ws = max(ws, self.ctx.lib.ifcbk_conv2d_wgrad_workspace(C.byref(d)))
            M = Nt * n.P * n.Q
            ws = max(ws, (((M + 255) // 256) * 2 * n.K + 2 * n.K) * 4)
        for g in self.groups:
            ws = max(ws, self.ctx.lib.ifcbk_conv2d_wgrad_workspace(C.byref(self._group_desc(g, Nt))))
        for n in self.plains:
            ws = max(ws, self.ctx.lib.ifcbk_conv2d_wgrad_workspace(C.byref(self._conv_desc(n, Nt))),
                     self.ctx.lib.ifcbk_bias_relu_bwd_workspace(Nt * n.P * n.Q, n.K))
        for n in self.bnrs:
            M = Nt * n.x.H * n.x.W
            ws = max(ws, (((M + 255) // 256) * 2 * n.K + 2 * n.K) * 4)
        if self.stem_u8 is not None:
            ws = max(ws, self.ctx.lib.ifcbk_stem_u8_wgrad_workspace(C.byref(self._conv_desc(self.stem_u8, Nt))))
        self.ctx.reserve(ws)

    @property
    def target(self):
        return self.tgt_bufs[self.in_slot]

    # ------------------------------------------------------------------ input pipelining
    def _select_slot(self, slot):
        self.in_slot = slot
        self.act[self.net.input.id] = self.in_bufs[slot]

    def prefetch_stream(self):
        """the side stream of the input pipeline (created on first use, with its own library context / workspace arena)"""
        if self.plan_only:
            raise RuntimeError('prefetch: this engine was built with plan_only=True')
        if self.pre_stream is None:
            self.pre_stream = torch.cuda.Stream(self.dev)
            self.pre_ctx = _lib.Context(self.dev.index)          # its own workspace arena: the resize tables of the prefetch
        return self.pre_stream

    def prefetch_begin(self):
        """-> (slot, stream): everything the caller enqueues on `stream` until prefetch_end() -- host-to-device copies of the
        next batch, load_rois(..., slot=slot), the label copy into tgt_bufs[slot] -- runs beside the step in flight.  The side
        stream first waits until the last step that read this slot has finished."""
        self.prefetch_stream()
        slot = 1 - self.in_slot
        if self.ev_free[slot] is not None:
            self.pre_stream.wait_event(self.ev_free[slot])
        return slot, self.pre_stream

    def prefetch_end(self, slot):
        ev = torch.cuda.Event()
        ev.record(self.pre_stream)
        self.ev_ready[slot] = ev
        self.prefetched = slot

    def use_prefetched(self):
        """make the prefetched slot the current one: the caller's stream waits for its preprocessing; the slot just left is
        marked busy until everything enqueued so far (the step that read it) has finished."""
        slot = self.prefetched
        if slot is None:
            raise RuntimeError('use_prefetched: nothing was prefetched')
        cur = torch.cuda.current_stream(self.dev)
        ev = torch.cuda.Event()
        ev.record(cur)
        self.ev_free[self.in_slot] = ev
        cur.wait_event(self.ev_ready[slot])
        self._select_slot(slot)
        self.prefetched = None
        return slot

    def activation_bytes(self):
        tot = sum(t.numel() * t.element_size() for t in self.act.values())
        tot += sum(t.numel() * t.element_size() for t in self.grad.values())
        return tot + sum(t.numel() for t in self.draw) * self.esize

    # ------------------------------------------------------------------ descriptors
    def _conv_desc(self, n, N):
        if (n.R * n.S > 1 and n.R == n.x.H and n.S == n.x.W and n.ph == 0 and n.pw == 0 and n.P == 1 and n.Q == 1
                and n.x.is_full and n.Cw == n.x.C):
            # the filter covers the whole input (inception AuxLogits.conv1, 5x5 on 5x5): a plain GEMM.  Lower it as a
            # 1x1 conv over the flattened pixel -- KRSC weights and NHWC activations already have that layout -- so
            # dgrad does not walk 25 taps of which one is valid per pixel
            CC = n.R * n.S * n.x.C
            return ConvDesc(N, 1, 1, CC, CC, n.K, 1, 1, 1, 1, 0, 0, 1, 1, n.y.buf.C, CC, self.cdtype)
        return ConvDesc(N, n.x.H, n.x.W, n.x.C, n.x.buf.C, n.K, n.R, n.S, n.sh, n.sw, n.ph, n.pw, n.P, n.Q,
                        n.y.buf.C, n.Cw, self.cdtype)

    def zeros_k(self, K):
        z = getattr(self, '_zeros_k', None)
        if z is None or z.numel() < K:
            z = self._zeros_k = torch.zeros(max(K, 4096), dtype=torch.float32, device=self.dev)
        return z

    def _pack_desc(self, n, d):
        """weight_pack of a layer whose output channels were padded (squeezenet's classifier conv): only the true rows exist in
        the fp32 master; the shadow rows behind them stay zero"""
        if n.K_real == n.K:
            return d
        dp = ConvDesc.from_buffer_copy(d)
        dp.K = n.K_real
        return dp

    def _group_desc(self, g, N):
        x = g.x
        return ConvDesc(N, x.H, x.W, x.C, x.buf.C, g.Ktot, 1, 1, 1, 1, 0, 0, x.H, x.W, g.Ktot, x.C, self.cdtype)

    def _aptr(self, view, grad=False):
        t = (self.grad if grad else self.act)[view.buf.id]
        return _vp(t, self.esize * view.coff)

    def _pptr(self, key, which='P'):
        o = self.poff[key][0]
        return _vp(getattr(self, which), 4 * o)

    def _stat(self, n, k):
        return _vp(self.stats, 4 * (n.st_off + k * n.st_ld))

    def _raw_ptr(self, n):
        """(pointer, pixel stride) of a conv's raw output -- a channel slice of the merged tensor for fused siblings"""
        if n.group is not None and n.cpool is None:
            return _vp(n.group.raw, self.esize * n.koff), n.group.Ktot
        return _vp(self.act[n.raw.id]), n.K

    def _pre_ptr(self, n):
        """a commuted pool branch's unpooled 1x1-conv output: its channel slice of the sibling GEMM's merged tensor"""
        return _vp(n.group.raw, self.esize * n.koff), n.group.Ktot

    def _bs_table(self, grp, gd, readers, fused_pool):
        """per-chunk producer table (ifcbk_bs_chunk) of the block input a sibling GEMM reads, when that GEMM is the input's only
        consumer and every conv that wrote a slice of it is a plain conv -> BN -> ReLU; None otherwise"""
        import numpy as np
        x = grp.x
        members = set(grp.members)
        for r in readers.get(x.buf.id, ()):
            if r not in members and not any(m.cpool is r for m in grp.members):
                return None
        prods = [m for m in self.net.nodes if m.kind != 'head' and m.y.buf.id == x.buf.id]
        convs = [m for m in prods if m.kind == 'conv']
        if not convs or not x.is_full:
            return None
        for m in convs:
            if not m.relu or m.residual is not None or m in fused_pool or m.aux:
                return None
        nrow = self.ctx.lib.ifcbk_conv2d_dgrad_bnstat_mblocks(C.byref(gd))
        if nrow <= 0:
            return None
        nchunk = x.C // 8
        tab = (_lib.BsChunk * nchunk)()
        for m in convs:
            rawp, rawld = self._raw_ptr(m)
            for k in range(m.K // 8):
                e = tab[m.y.coff // 8 + k]
                e.raw = rawp.value + k * 8 * self.esize
                e.stat = self.stats.data_ptr() + 4 * (m.st_off + k * 8)
                e.raw_ld, e.stat_ld = rawld, m.st_ld
        table = torch.from_numpy(np.frombuffer(bytes(tab), dtype=np.uint8).copy()).to(self.dev)
        part = torch.zeros(nrow * 2 * x.C, dtype=torch.float32, device=self.dev)
        return table, part, nrow, convs

    # ------------------------------------------------------------------ programs
    def plan(self, N):
        if N > self.max_batch:
            raise RuntimeError('batch %d exceeds the engine capacity %d' % (N, self.max_batch))
        key = (N, self.in_slot, self.in_kind[self.in_slot])
        if key not in self._plans:
            self._plans[key] = self._build(N)
            self._plans[key].dispatch_env = _dispatch_env()
        pl = self._plans[key]
        if pl.dispatch_env != _dispatch_env():
            # the library picks its conv kernels per launch from these switches, while the rows of the BatchNorm partial sums (and
            # the workspace) were sized when the plan was built: a change in between would make bn_finalize sum the wrong rows
            raise RuntimeError('a kernel-dispatch switch (%s) changed after the plan for batch %d was built: build a new model/engine '
                               'instead' % (', '.join(k for k, v in zip(_DISPATCH_SWITCHES, pl.dispatch_env) if os.environ.get(k) != v), N))
        return pl

    def _build(self, N):
        net = self.net
        fwd_t, fwd_e, bwd = OpList(), OpList(), OpList()
        pack, evalprep = OpList(), OpList()
        written = set()       # grad buffers already written in this backward pass

        def acc_flag(buf):
            a = 1 if buf.id in written else 0
            written.add(buf.id)
            return a

        # conv -> BN -> ReLU whose activation feeds ONLY a 3x3/stride-2 max pool (inception Conv2d_2b / Conv2d_4a, the
        # resnet stem): in training the activation and its gradient are never materialised (bn_apply_maxpool / bn_bwd_maxpool)
        readers = {}
        for m in net.nodes:
            for v in ((m.x, getattr(m, 'residual', None)) if m.kind == 'conv' else (m.x,)):
                if v is not None:
                    readers.setdefault(v.buf.id, []).append(m)
        pool_cand = {}        # conv node -> (pool node, its index): the candidates, whatever the switches say
        for kk, m in enumerate(net.nodes):
            if (m.kind == 'max' and m.R == 3 and m.S == 3 and m.sh == 2 and m.sw == 2 and m.ph <= 1 and m.pw <= 1
                    and m.x.is_full and len(readers.get(m.x.buf.id, ())) == 1):
                prod = [c for c in net.nodes if c.kind == 'conv' and c.y.buf.id == m.x.buf.id]
                if (len(prod) == 1 and prod[0].y.is_full and prod[0].relu and prod[0].residual is None
                        and prod[0].group is None and bool(prod[0].aux) == bool(m.aux)):
                    pool_cand[prod[0]] = (m, kk)
        # IFCBK_FUSE_POOL: the TRAINING fusion (bn_apply_maxpool / bn_bwd_maxpool); IFCBK_FUSE_POOL_EVAL: the eval one below --
        # two switches, two decisions
        fused_pool = dict(pool_cand) if os.environ.get('IFCBK_FUSE_POOL', '1') != '0' else {}
        fused_pool_nodes = {v[0] for v in fused_pool.values()}
        self.fused_pool = fused_pool
        # eval twin: conv + folded BatchNorm + ReLU + that max pool in ONE kernel where the row-streaming conv serves the layer
        # (inception Conv2d_2b -> maxpool1): the 708 MB activation of a batch of 256 is neither written nor read back
        fused_pool_eval = {}
        if os.environ.get('IFCBK_FUSE_POOL_EVAL', '1') != '0' and not self.plan_only:
            for cn, (pn, pk) in pool_cand.items():
                if (pn.ph == 0 and pn.pw == 0 and cn.y.buf.C == cn.K
                        and self.ctx.lib.ifcbk_conv2d_fwd_affine_maxpool_ok(C.byref(self._conv_desc(cn, N)))):
                    fused_pool_eval[cn] = pn
        fused_pool_eval_nodes = set(fused_pool_eval.values())
        # conv c whose input is the private BN+ReLU activation of conv n: c's input-gradient kernel also reduces n's BN
        # backward sums in its epilogue (ifcbk_conv2d_dgrad_bnstat) and n's BN backward skips its reduction pass
        bnstat_of = {}        # consumer conv -> producer conv
        # 0 off, 1 one-producer layers, 2 + block outputs through a chunk table.  Level 2 is correct (tests run it) but measured
        # SLOWER at batch 256 (the wide-tile dgrads run one block per CU, nothing hides their epilogue): default 1
        fuse_level = int(os.environ.get('IFCBK_FUSE_BNSTAT', '1'))
        fuse_bnstat = fuse_level >= 1
        keep = []             # device tables / buffers the op tables point into
        if fuse_bnstat:
            for cnode in net.nodes:
                if cnode.kind != 'conv' or cnode.group is not None or cnode.x.buf.is_input or not cnode.x.is_full:
                    continue
                if len(readers.get(cnode.x.buf.id, ())) != 1:
                    continue
                prod = [c for c in net.nodes if c.kind == 'conv' and c.y.buf.id == cnode.x.buf.id]
                if (len(prod) == 1 and prod[0].y.is_full and prod[0].relu and prod[0].residual is None
                        and prod[0] not in fused_pool and bool(prod[0].aux) == bool(cnode.aux)
                        and self._conv_desc(cnode, N).C == cnode.x.C      # not the flattened full-cover form: its 'channels' are (tap, c)
                        and self.ctx.lib.ifcbk_conv2d_dgrad_bnstat_mblocks(C.byref(self._conv_desc(cnode, N))) > 0):
                    bnstat_of[cnode] = prod[0]
        bnstat_done = {}      # producer conv -> (partials pointer, row count, lane of the partial buffer)

        # ---- lanes: every chain of nodes that hangs off a shared tensor (an Inception block input, a resnet block
        # input) gets a lane, round-robin; a node fed by a private tensor stays on its producer's lane
        NL = self.NL
        producers = {}
        for m in net.nodes:
            if m.kind != 'head':
                producers.setdefault(m.y.buf.id, []).append(m)

        # Which chain gets which lane: chains in order of appearance -- lane 0 (the caller's stream, where the sibling GEMMs and their
        # gradients run) takes the light 1x1 branch, the heavy chains land on lanes 1 and 2.  Putting the HEAVIEST chain of every
        # block on lane 0 (round 2) or rotating the lanes so that a block's GEMM follows the previous block's heaviest chain on the
        # same stream (round 3) both lose ~1 ms per step: the event wait they save is cheaper than what they serialise.

        def follows_of():
            f = {}
            for m in net.nodes:
                if m.aux or m.kind == 'head':
                    continue
                prods = producers.get(m.x.buf.id, [])
                if (len(prods) == 1 and prods[0].y.is_full and m.x.is_full and len(readers.get(m.x.buf.id, ())) == 1
                        and not prods[0].aux):
                    f[m] = prods[0]
            return f

        def assign_lanes(nl):
            follows = follows_of()
            head_of = {}
            for m in net.nodes:                       # forward order: a follower's producer is already resolved
                if m.aux or m.kind == 'head':
                    continue
                head_of[m] = head_of[follows[m]] if m in follows and follows[m] in head_of else m
            by_tensor = {}
            for m in net.nodes:
                if not m.aux and m.kind != 'head' and head_of.get(m) is m:
                    by_tensor.setdefault(m.x.buf.id, []).append(m)
            rank = {}
            for heads in by_tensor.values():
                for k, h in enumerate(heads):
                    rank[h] = k
            lanes = {}
            for m in net.nodes:
                if m.aux:
                    lanes[m] = 1 % nl     # the auxiliary classifier (pool, 2 convs, fc) runs beside Mixed_7a..7c
                elif m.kind == 'head':
                    lanes[m] = 0
                else:
                    lanes[m] = rank[head_of[m]] % nl
            return lanes

        # training and eval programs get their own lane counts: measured (B=256) 2 / 3 / 4 lanes = 25.9 / 25.5 / 25.3 ms per
        # train step but 6.12 / 6.63 / 6.78 ms per eval forward (its kernels are few and wide: more lanes only add waits)
        WLs = list(range(NL - self.wgrad_lane, NL))        # the weight-gradient lanes
        wl_next = [0]
        lane_of = assign_lanes(NL - self.wgrad_lane)
        # The TRUNK: the single-chain prefix of the network (inception's stem Conv2d_1a .. 4a + its pools).  Its backward is the end
        # of the step -- one chain lane busy, the other chain lanes idle -- and its weight gradients (1.0 ms of kernels when each
        # runs alone) used to queue up on the one weight-gradient lane behind each other.  Round 5: they go round-robin over ALL
        # lanes but the trunk's own, so that the weight gradients of 4a / 2b / 2a overlap each other beside the trunk's chain
        # (IFCBK_WGRAD_SPREAD=1; measured on one box, three interleaved rounds: 21.44 vs 21.45 ms per step -- no gain: those kernels cost the step their CU-time, not their queueing -- so the default stays 0: one weight-gradient lane for everything, as in round 4)
        trunk = set()
        if self.wgrad_lane and NL - self.wgrad_lane >= 2 and os.environ.get('IFCBK_WGRAD_SPREAD', '0') != '0':
            body = [m for m in net.nodes if not m.aux and m.kind != 'head']
            if body:
                for m in body:
                    if lane_of[m] != lane_of[body[0]]:
                        break
                    trunk.add(m)
        TLs = [l for l in range(NL - 1, -1, -1) if not trunk or l != lane_of[next(iter(trunk))]] if trunk else []
        tl_next = [0]
        # the forward has no weight gradients: IFCBK_FWD_LANES lets its branches use the weight-gradient lane(s) too
        fwd_lanes = max(1, min(NL, int(os.environ.get('IFCBK_FWD_LANES', str(NL - self.wgrad_lane)))))
        lane_fwd = assign_lanes(fwd_lanes) if fwd_lanes != NL - self.wgrad_lane else lane_of
        lane_eval = assign_lanes(self.NL_eval)
        if not hasattr(self, 'draw_own'):
            self.draw_own = {}

        def _carve(nelem):
            """a d(raw) buffer out of ONE pooled allocation per engine (78 tensors of an inception plan: one hipMalloc instead of 78,
            not zero-filled -- BatchNorm backward writes every element before anything reads it)"""
            if not self.wgrad_lane:
                # (without a weight-gradient lane only the members of grouped launches keep a buffer of their own: a handful of
                # tensors -- a pool sized for EVERY conv, 4.6 GB at batch 256, would be mostly unused)
                return torch.empty(nelem, dtype=self.tdtype, device=self.dev)
            if getattr(self, '_draw_pool', None) is None:
                tot = 0
                for m in self.convs:
                    if m.group is None or m.cpool is not None:
                        tot += (self.train_batch * m.P * m.Q * m.K + 127) // 128 * 128
                for gq in self.groups:
                    tot += (self.train_batch * gq.x.H * gq.x.W * gq.Ktot + 127) // 128 * 128
                self._draw_pool = torch.empty(max(tot, 128), dtype=self.tdtype, device=self.dev)
                self._draw_used = 0
            n = (nelem + 127) // 128 * 128
            if self._draw_used + n > self._draw_pool.numel():
                return torch.empty(nelem, dtype=self.tdtype, device=self.dev)
            v = self._draw_pool[self._draw_used:self._draw_used + nelem]
            self._draw_used += n
            return v

        def own_draw(m):
            if m not in self.draw_own:
                self.draw_own[m] = _carve(self.train_batch * m.P * m.Q * m.K)
            return self.draw_own[m]

        def group_draw(gq):
            """the merged d(raw) tensor of a sibling GEMM: the shared scratch, or the group's own when weight gradients have a lane"""
            if not self.wgrad_lane:
                return self.draw_group
            if getattr(gq, 'draw_own', None) is None:
                gq.draw_own = _carve(self.train_batch * gq.x.H * gq.x.W * gq.Ktot)
            return gq.draw_own

        # ---- resources for the lane scheduler: channel ranges of tensors
        def ra(v):
            return ('a', v.buf.id, v.coff, v.coff + v.C)

        def rg(v):
            return ('g', v.buf.id, v.coff, v.coff + v.C)

        def rraw(m):
            if m.group is not None and m.cpool is None:
                return ('gr', id(m.group), m.koff, m.koff + m.K)
            return ('r', m.raw.id, 0, m.K)

        absorbed = self.absorbed_pools

        def rdg(m):
            return ('dg', id(m.group), m.koff, m.koff + m.K)

        rst = lambda m: ('s', id(m), 0, 1)
        rbp = lambda L: ('bp', L, 0, 1)
        ram = lambda kk: ('am', kk, 0, 1)

        bwd_groups = []
        for k, n in enumerate(net.nodes):
            grp = OpList()
            L = lane_fwd[n]
            if n.kind == 'conv':
                M = N * n.P * n.Q
                d = self._conv_desc(n, N)
                # forward conv writes the raw output (own buffer, ld = K)
                raw, ldraw = self._raw_ptr(n)
                dfw = ConvDesc.from_buffer_copy(d)
                dfw.ldy = ldraw
                wk = _vp(self.Wsh, self.esize * n.w_off)
                wT = _vp(self.Wsh, self.esize * n.wT_off)
                ckey, bkey = n.conv_key + '.weight', n.bn_key
                mb = self.ctx.lib.ifcbk_conv2d_fwd_mblocks(C.byref(d))
                # the stem conv on the resized u8 plane of this slot (conv_stem_u8.hip) instead of the [N,S,S,8] tensor
                u8 = self.stem_u8 is n and self.in_kind[self.in_slot] == 'u8'
                if u8:
                    mb = self.ctx.lib.ifcbk_stem_u8_rows(C.byref(d))
                    gu8, gab = _vp(self.in_u8[self.in_slot]), _vp(self.in_ab[self.in_slot])
                bnd = BnDesc(M, n.K, ldraw, n.y.buf.C, 1 if n.relu else 0, self.cdtype, n.eps, 0.1)
                g = n.group
                first_of_group = g is not None and g.members[0] is n
                res = self._aptr(n.residual) if n.residual is not None else None
                ldr = n.residual.buf.C if n.residual is not None else 0
                for lst, train in ((fwd_t, True), (fwd_e, False)):
                    if n.aux and not train:
                        continue
                    Le = lane_eval[n]
                    if not train and g is not None and self.eval_groups:
                        # inference: the sibling 1x1 convs of a block as ONE GEMM with per-segment destinations
                        # (ifcbk_conv2d_fwd_affine_segments): members with their folded BatchNorm + ReLU straight into their output
                        # tensors, a commuted pool branch raw into its slice of the sibling tensor (its pool applies the affine)
                        if first_of_group:
                            gd = self._group_desc(g, N)
                            ys, iv, wr = [], [], []
                            for m in g.members:
                                if m.cpool is not None:
                                    pre, ldpre = self._pre_ptr(m)
                                    ys.append(pre); iv.append(m.K | (ldpre << 20)); wr.append(('gr', id(g), m.koff, m.koff + m.K))
                                else:
                                    ys.append(self._aptr(m.y)); iv.append(m.K | (m.y.buf.C << 20) | (1 << 40)); wr.append(ra(m.y))
                            ys += [None] * (4 - len(ys))
                            iv += [0] * (4 - len(iv))
                            lst.add(_lib.OP_CONV_FWD_AFFINE_SEG, '+'.join(m.name for m in g.members),
                                    p=(self._aptr(g.x), _vp(self.Wsh, self.esize * g.w_off), ys[0], ys[1], ys[2], ys[3],
                                       self._stat(g.members[0], 4), self._stat(g.members[0], 5)),
                                    i=iv, conv=gd, lane=Le, reads=[ra(g.x)], writes=wr)
                        if n.cpool is not None:
                            cp = n.cpool
                            pre, ldpre = self._pre_ptr(n)
                            ppd = PoolDesc(N, cp.x.H, cp.x.W, n.K, ldpre, 3, 3, 1, 1, 1, 1, cp.P, cp.Q, n.y.buf.C, self.cdtype)
                            lst.add(_lib.OP_AVGPOOL_AFFINE, cp.name + '(' + n.name + ')',
                                    p=(pre, self._stat(n, 4), self._stat(n, 5), self._aptr(n.y)), flags=4 if n.relu else 0, pool=ppd,
                                    lane=Le, reads=[('gr', id(g), n.koff, n.koff + n.K)], writes=[ra(n.y)])
                        continue
                    if not train and n.cpool is not None:
                        # inference twin of the commuted pool branch: plain 1x1 conv of the block input into the branch's slice of
                        # the sibling tensor, then avgpool with the eval-BN affine + ReLU in its epilogue
                        cp = n.cpool
                        pre, ldpre = self._pre_ptr(n)
                        dpre = ConvDesc(N, cp.x.H, cp.x.W, cp.x.C, cp.x.buf.C, n.K, 1, 1, 1, 1, 0, 0, n.P, n.Q, ldpre, n.Cw, self.cdtype)
                        rpre = ('gr', id(g), n.koff, n.koff + n.K)
                        lst.add(_lib.OP_CONV_FWD, n.name, p=(self._aptr(cp.x), wk, pre, None), conv=dpre, lane=Le,
                                reads=[ra(cp.x)], writes=[rpre])
                        ppd = PoolDesc(N, cp.x.H, cp.x.W, n.K, ldpre, 3, 3, 1, 1, 1, 1, cp.P, cp.Q, n.y.buf.C, self.cdtype)
                        lst.add(_lib.OP_AVGPOOL_AFFINE, cp.name + '(' + n.name + ')',
                                p=(pre, self._stat(n, 4), self._stat(n, 5), self._aptr(n.y)), flags=4 if n.relu else 0, pool=ppd,
                                lane=Le, reads=[rpre], writes=[ra(n.y)])
                        continue
                    if not train and u8:
                        lst.add(_lib.OP_STEM_U8_FWD, n.name,
                                p=(gu8, self._pptr(ckey), gab, self._aptr(n.y), None, self._stat(n, 4), self._stat(n, 5)),
                                flags=4 if n.relu else 0, conv=d, lane=Le, reads=[ra(n.x)], writes=[ra(n.y)])
                        continue
                    if not train and n in fused_pool_eval:
                        pn = fused_pool_eval[n]
                        lst.add(_lib.OP_CONV_FWD_AFFINE_MAXPOOL, n.name + '+' + pn.name,
                                p=(self._aptr(n.x), wk, self._aptr(pn.y), self._stat(n, 4), self._stat(n, 5)),
                                i=(pn.y.buf.C,), flags=4 if n.relu else 0, conv=d, lane=Le, reads=[ra(n.x)], writes=[ra(pn.y)])
                        continue
                    if not train:
                        # inference: eval-BN affine (+residual) + ReLU fused into the conv epilogue; no raw tensor
                        lst.add(_lib.OP_CONV_FWD_AFFINE, n.name,
                                p=(self._aptr(n.x), wk, self._aptr(n.y), self._stat(n, 4), self._stat(n, 5), res),
                                i=(ldr,), flags=4 if n.relu else 0, conv=d, lane=Le,
                                reads=[ra(n.x)] + ([ra(n.residual)] if n.residual is not None else []), writes=[ra(n.y)])
                        continue
                    if g is not None:
                        if not first_of_group:
                            continue                      # emitted with the group's first member
                        # ONE GEMM for all sibling 1x1 convs, then each branch's BN on its channel slice
                        gd = self._group_desc(g, N)
                        gmb = self.ctx.lib.ifcbk_conv2d_fwd_mblocks(C.byref(gd))
                        lst.add(_lib.OP_CONV_FWD, '+'.join(m.name for m in g.members),
                                p=(self._aptr(g.x), _vp(self.Wsh, self.esize * g.w_off), _vp(g.raw), _vp(self.bn_part[0])),
                                conv=gd, lane=0, reads=[ra(g.x)], writes=[('gr', id(g), 0, g.Ktot), rbp(0)])
                        for m in g.members:
                            mbk = m.bn_key
                            mraw, mld = self._raw_ptr(m)
                            mbnd = BnDesc(M, m.K, mld, m.y.buf.C, 1, self.cdtype, m.eps, 0.1)
                            if m.cpool is not None:
                                # commuted pool branch: pool the conv's slice of the merged tensor, then BatchNorm the pooled tensor
                                Lm, pn = lane_fwd[m], m.cpool
                                pre, ldpre = self._pre_ptr(m)
                                ppd = PoolDesc(N, pn.x.H, pn.x.W, m.K, ldpre, 3, 3, 1, 1, 1, 1, pn.P, pn.Q, mld, self.cdtype)
                                rpre, rr2 = ('gr', id(g), m.koff, m.koff + m.K), ('r', m.raw.id, 0, m.K)
                                lst.add(_lib.OP_AVGPOOL_FWD, pn.name + '(' + m.name + ')', p=(pre, mraw), pool=ppd, lane=Lm,
                                        reads=[rpre], writes=[rr2])
                                lst.add(_lib.OP_BN_STATS, m.name, p=(mraw, _vp(self.bn_part[Lm])), bn=mbnd, lane=Lm,
                                        reads=[rr2], writes=[rbp(Lm)])
                                lst.add(_lib.OP_BN_FINALIZE, m.name,
                                        p=(_vp(self.bn_part[Lm]), self._pptr(mbk + '.weight'), self._pptr(mbk + '.bias'),
                                           _vp(self.bviews[mbk + '.running_mean']), _vp(self.bviews[mbk + '.running_var']),
                                           self._stat(m, 0), self._stat(m, 1), self._stat(m, 2), self._stat(m, 3)),
                                        i=(self.ctx.lib.ifcbk_bn_stats_rows(M), m.K), bn=mbnd, lane=Lm, reads=[rbp(Lm)], writes=[rst(m)])
                                lst.add(_lib.OP_BN_APPLY, m.name,
                                        p=(mraw, self._stat(m, 2), self._stat(m, 3), None, self._aptr(m.y)), i=(0,), bn=mbnd,
                                        lane=Lm, reads=[rr2, rst(m)], writes=[ra(m.y)])
                                continue
                            # (one finalize per member, not per group: with all 29 member finalizes of the training forward REMOVED
                            # -- timing only -- the step gains 0.15-0.22 ms; merging them into 11 launches could recover part of that)
                            lst.add(_lib.OP_BN_FINALIZE, m.name,
                                    p=(_vp(self.bn_part[0], 4 * m.koff), self._pptr(mbk + '.weight'), self._pptr(mbk + '.bias'),
                                       _vp(self.bviews[mbk + '.running_mean']), _vp(self.bviews[mbk + '.running_var']),
                                       self._stat(m, 0), self._stat(m, 1), self._stat(m, 2), self._stat(m, 3)),
                                    i=(gmb, g.Ktot), bn=mbnd, lane=lane_fwd[m], reads=[rbp(0)], writes=[rst(m)])
                            lst.add(_lib.OP_BN_APPLY, m.name,
                                    p=(mraw, self._stat(m, 2), self._stat(m, 3), None, self._aptr(m.y)), i=(0,), bn=mbnd,
                                    lane=lane_fwd[m], reads=[rraw(m), rst(m)], writes=[ra(m.y)])
                        continue
                    if u8:
                        lst.add(_lib.OP_STEM_U8_FWD, n.name, p=(gu8, self._pptr(ckey), gab, raw, _vp(self.bn_part[L]), None, None),
                                conv=dfw, lane=L, reads=[ra(n.x)], writes=[rraw(n), rbp(L)])
                    else:
                        lst.add(_lib.OP_CONV_FWD, n.name, p=(self._aptr(n.x), wk, raw, _vp(self.bn_part[L]) if train else None),
                                conv=dfw, lane=L, reads=[ra(n.x)], writes=[rraw(n), rbp(L)])
                    if train:
                        lst.add(_lib.OP_BN_FINALIZE, n.name,
                                p=(_vp(self.bn_part[L]), self._pptr(bkey + '.weight'), self._pptr(bkey + '.bias'),
                                   _vp(self.bviews[bkey + '.running_mean']), _vp(self.bviews[bkey + '.running_var']),
                                   self._stat(n, 0), self._stat(n, 1), self._stat(n, 2), self._stat(n, 3)),
                                i=(mb,), bn=bnd, lane=L, reads=[rbp(L)], writes=[rst(n)])
                    if train and n in fused_pool:
                        pn, pk = fused_pool[n]
                        ppd = PoolDesc(N, pn.x.H, pn.x.W, pn.x.C, ldraw, 3, 3, 2, 2, pn.ph, pn.pw, pn.P, pn.Q, pn.y.buf.C,
                                       self.cdtype)
                        lst.add(_lib.OP_BN_APPLY_MAXPOOL, n.name + '+' + pn.name,
                                p=(raw, self._stat(n, 2), self._stat(n, 3), self._aptr(pn.y), _vp(self.argmax[pk])),
                                i=(1,), pool=ppd, lane=L, reads=[rraw(n), rst(n)], writes=[ra(pn.y), ram(pk)])
                        continue
                    lst.add(_lib.OP_BN_APPLY, n.name,
                            p=(raw, self._stat(n, 2 if train else 4), self._stat(n, 3 if train else 5), res, self._aptr(n.y)),
                            i=(ldr,), bn=bnd, lane=L, reads=[rraw(n), rst(n)] + ([ra(n.residual)] if n.residual is not None else []),
                            writes=[ra(n.y)])
                evalprep.add(_lib.OP_BN_FINALIZE, n.name,
                             p=(None, self._pptr(bkey + '.weight'), self._pptr(bkey + '.bias'),
                                _vp(self.bviews[bkey + '.running_mean']), _vp(self.bviews[bkey + '.running_var']),
                                None, None, self._stat(n, 4), self._stat(n, 5)), i=(0,), bn=bnd)
                needs_dgrad = not n.x.buf.is_input
                pack.add(_lib.OP_WEIGHT_PACK, n.name, p=(self._pptr(ckey), wk, wT if needs_dgrad else None), i=(n.wT_ld,), conv=d)
                # ---- backward of this node
                di = L * 2 + (k & 1) if self.wgrad_side_lane else L
                draw = _vp(self.draw[di]) if (g is None or n.cpool is not None) else _vp(self.draw_group, self.esize * n.koff)
                dres, lddres, dres_acc = None, 0, 0
                # (flags resolved later, in reverse order) -> store a closure
                bwd_groups.append(('conv', n, d, bnd, draw, wT, needs_dgrad, di))
            elif n.kind in ('max', 'avg'):
                pd = PoolDesc(N, n.x.H, n.x.W, n.x.C, n.x.buf.C, n.R, n.S, n.sh, n.sw, n.ph, n.pw, n.P, n.Q, n.y.buf.C,
                              self.cdtype)
                for lst, train in ((fwd_t, True), (fwd_e, False)):
                    if n.aux and not train:
                        continue
                    if n in absorbed:
                        continue                  # runs behind its 1x1 conv, on that conv's output (see __init__)
                    Lx = L if train else lane_eval[n]
                    if n.kind == 'max':
                        if train and n in fused_pool_nodes:
                            continue              # done by the producing conv's bn_apply_maxpool
                        if not train and n in fused_pool_eval_nodes:
                            continue              # done in the producing conv's epilogue
                        lst.add(_lib.OP_MAXPOOL_FWD, n.name, p=(self._aptr(n.x), self._aptr(n.y), _vp(self.argmax[k]) if train else None), pool=pd,
                                lane=Lx, reads=[ra(n.x)], writes=[ra(n.y), ram(k)])
                    else:
                        lst.add(_lib.OP_AVGPOOL_FWD, n.name, p=(self._aptr(n.x), self._aptr(n.y)), pool=pd,
                                lane=Lx, reads=[ra(n.x)], writes=[ra(n.y)])
                bwd_groups.append(('pool', n, pd, k))
            elif n.kind == 'head':
                hd = HeadDesc(N, n.HW, n.C, n.x.buf.C, n.NC, self.cdtype, 2.0)
                wkey, bkey = n.key + '.weight', n.key + '.bias'
                for lst, train in ((fwd_t, True), (fwd_e, False)):
                    if n.aux and not train:
                        continue
                    mask = _vp(n.mask) if (train and n.dropout) else None
                    lst.add(_lib.OP_HEAD_FWD, n.name,
                            p=(self._aptr(n.x), mask, self._pptr(wkey) if n.fc else None, self._pptr(bkey) if n.fc else None,
                               _vp(n.feat), _vp(n.logits)), head=hd,
                            lane=L if train else lane_eval[n], reads=[ra(n.x)], writes=[('hd', id(n), 0, 1)])
                bwd_groups.append(('head', n, hd))
            elif n.kind == 'cb':
                # conv (+bias) (+ReLU) without BatchNorm: the bias is the epilogue's shift (scale = 1), read from the fp32 master
                d = self._conv_desc(n, N)
                wk = _vp(self.Wsh, self.esize * n.w_off)
                wT = _vp(self.Wsh, self.esize * n.wT_off)
                for lst, train in ((fwd_t, True), (fwd_e, False)):
                    Lx = L if train else lane_eval[n]
                    if n.bias or n.relu:
                        lst.add(_lib.OP_CONV_FWD_AFFINE, n.name,
                                p=(self._aptr(n.x), wk, self._aptr(n.y), _vp(self.ones),
                                   self._pptr(n.key + '.bias') if n.bias else _vp(self.zeros_k(n.K)), None),
                                i=(0,), flags=4 if n.relu else 0, conv=d, lane=Lx, reads=[ra(n.x)], writes=[ra(n.y)])
                    else:
                        lst.add(_lib.OP_CONV_FWD, n.name, p=(self._aptr(n.x), wk, self._aptr(n.y), None), conv=d, lane=Lx,
                                reads=[ra(n.x)], writes=[ra(n.y)])
                needs_dgrad = not n.x.buf.is_input
                pack.add(_lib.OP_WEIGHT_PACK, n.name, p=(self._pptr(n.key + '.weight'), wk, wT if needs_dgrad else None),
                         i=(n.K if n.K != n.K_real else 0,), conv=self._pack_desc(n, d))      # wT rows keep the PADDED stride
                bwd_groups.append(('cb', n, d, wT, needs_dgrad))
            elif n.kind == 'drop':
                cnt = N * n.x.H * n.x.W * n.x.C
                for lst, train in ((fwd_t, True), (fwd_e, False)):
                    lst.add(_lib.OP_DROPOUT, n.name, p=(self._aptr(n.x), _vp(n.mask) if train else None, self._aptr(n.y)),
                            i=(cnt, self.cdtype), f=(1.0 / (1.0 - n.p),), lane=L if train else lane_eval[n],
                            reads=[ra(n.x), ('dm', id(n), 0, 1)], writes=[ra(n.y)])
                bwd_groups.append(('drop', n, cnt))
            elif n.kind == 'flat':
                fi = (N, n.x.H * n.x.W, n.x.C, n.x.buf.C | (self.cdtype << 32))
                for lst, train in ((fwd_t, True), (fwd_e, False)):
                    lst.add(_lib.OP_FLATTEN_CHW, n.name, p=(self._aptr(n.x), self._aptr(n.y)), i=fi, flags=4,
                            lane=L if train else lane_eval[n], reads=[ra(n.x)], writes=[ra(n.y)])
                bwd_groups.append(('flat', n, fi))
            elif n.kind == 'bnr':
                # BatchNorm -> ReLU in FRONT of a conv, on a channel slice of a concatenation (densenet)
                M = N * n.x.H * n.x.W
                bkey = n.bn_key
                bnd = BnDesc(M, n.K, n.x.buf.C, n.K, 1 if n.relu else 0, self.cdtype, n.eps, 0.1)
                xp = self._aptr(n.x)
                fwd_t.add(_lib.OP_BN_STATS, n.name, p=(xp, _vp(self.bn_part[L])), bn=bnd, lane=L, reads=[ra(n.x)], writes=[rbp(L)])
                fwd_t.add(_lib.OP_BN_FINALIZE, n.name,
                          p=(_vp(self.bn_part[L]), self._pptr(bkey + '.weight'), self._pptr(bkey + '.bias'),
                             _vp(self.bviews[bkey + '.running_mean']), _vp(self.bviews[bkey + '.running_var']),
                             self._stat(n, 0), self._stat(n, 1), self._stat(n, 2), self._stat(n, 3)),
                          i=(self.ctx.lib.ifcbk_bn_stats_rows(M), n.K), bn=bnd, lane=L, reads=[rbp(L)], writes=[rst(n)])
                for lst, train in ((fwd_t, True), (fwd_e, False)):
                    lst.add(_lib.OP_BN_APPLY, n.name,
                            p=(xp, self._stat(n, 2 if train else 4), self._stat(n, 3 if train else 5), None, self._aptr(n.y)),
                            i=(0,), bn=bnd, lane=L if train else lane_eval[n], reads=[ra(n.x), rst(n)], writes=[ra(n.y)])
                evalprep.add(_lib.OP_BN_FINALIZE, n.name,
                             p=(None, self._pptr(bkey + '.weight'), self._pptr(bkey + '.bias'),
                                _vp(self.bviews[bkey + '.running_mean']), _vp(self.bviews[bkey + '.running_var']),
                                None, None, self._stat(n, 4), self._stat(n, 5)), i=(0,), bn=bnd)
                bwd_groups.append(('bnr', n, bnd))

        # ---- grouped weight gradients (round 4): the wide-tile wgrads of one block that share a channel tile run as ONE split-K
        # grid (ifcbk_conv2d_wgrad_group) behind the block's LAST such layer -- 1/n of the slab traffic and reduce work per layer,
        # n times the K-steps per block.  Each member keeps its d(raw) in a buffer of its own until the group has run (the per-lane
        # scratch is reused by the next layer's BatchNorm backward).  IFCBK_WGRAD_GROUP=0 switches it off.
        wg_of, wg_groups = {}, []
        if os.environ.get('IFCBK_WGRAD_GROUP', '1') != '0' and self.dtype == 'bf16' and not self.wgrad_side_lane:
            buckets = {}
            for gsp in reversed(bwd_groups):
                if gsp[0] != 'conv':
                    continue
                n = gsp[1]
                if n.group is not None or n.aux or (self.stem_u8 is n and self.in_kind[self.in_slot] == 'u8'):
                    continue
                dbw = ConvDesc.from_buffer_copy(gsp[2])
                dbw.ldy = n.K
                khv = self.ctx.lib.ifcbk_conv2d_wgrad_group_member_kh(C.byref(dbw))
                if khv <= 0:
                    continue
                key = n.name.split('.')[0] if net.name == 'inception_v3' else n.name.rsplit('.', 1)[0]
                buckets.setdefault((key, khv), []).append((n, dbw))
            for (key, khv), mem in buckets.items():
                for c0 in range(0, len(mem), 6):
                    part = mem[c0:c0 + 6]
                    if len(part) < 2:
                        continue
                    descs = (ConvDesc * len(part))(*[m[1] for m in part])
                    need = self.ctx.lib.ifcbk_conv2d_wgrad_group_workspace(len(part), descs)
                    if need == 0:
                        continue                      # the library keeps such layers on their single launches
                    grp_obj = dict(members=[m[0] for m in part], descs=[m[1] for m in part], ws=need, kh=khv, key=key)
                    wg_groups.append(grp_obj)
                    for m in part:
                        wg_of[m[0]] = grp_obj
            if wg_groups:
                need = max(gq['ws'] for gq in wg_groups)
                if not self.plan_only and need > self.ctx.lib.ifcbk_ctx_workspace_bytes(self.ctx.h):
                    self.ctx.reserve(need)
                for gq in wg_groups:
                    for m in gq['members']:
                        own_draw(m)
        self.wgrad_groups = wg_groups

        # backward in reverse node order, resolving first-writer / accumulate flags
        for g in reversed(bwd_groups):
            if g[0] == 'head':
                _, n, hd = g
                wkey, bkey = n.key + '.weight', n.key + '.bias'
                assert n.x.is_full
                acc = acc_flag(n.x.buf)
                assert acc == 0, 'head must be the first writer of its input gradient'
                bwd.add(_lib.OP_HEAD_BWD, n.name,
                        p=(_vp(n.dlogits), _vp(n.feat), _vp(n.mask) if n.dropout else None, self._pptr(wkey) if n.fc else None,
                           self._pptr(wkey, 'G') if n.fc else None, self._pptr(bkey, 'G') if n.fc else None, self._aptr(n.x, True)),
                        i=(n.x.buf.C,), head=hd, lane=lane_of[n], reads=[('hd', id(n), 0, 1)], writes=[rg(n.x)])
            elif g[0] == 'cb':
                _, n, d, wT, needs_dgrad = g
                L = lane_of[n]
                M = N * n.P * n.Q
                dy, ldy = self._aptr(n.y, True), n.y.buf.C
                if n.relu or n.bias:
                    # dz = dy * (y > 0) in place, dbias = column sums of dz
                    bnd = BnDesc(M, n.K, ldy, ldy, 1 if n.relu else 0, self.cdtype, 0.0, 0.0)
                    bwd.add(_lib.OP_BIAS_RELU_BWD, n.name,
                            p=(self._aptr(n.y), dy, dy if n.relu else None, self._pptr(n.key + '.bias', 'G') if n.bias else None),
                            i=(ldy,), bn=bnd, lane=L, reads=[ra(n.y), rg(n.y)], writes=[rg(n.y)])
                bwd.add(_lib.OP_CONV_WGRAD, n.name, p=(self._aptr(n.x), dy, self._pptr(n.key + '.weight', 'G')), conv=d, lane=L,
                        reads=[ra(n.x), rg(n.y)], writes=[])
                if needs_dgrad:
                    assert n.x.is_full
                    acc = acc_flag(n.x.buf)
                    bwd.add(_lib.OP_CONV_DGRAD, n.name, p=(dy, wT, self._aptr(n.x, True)), flags=acc, conv=d, lane=L,
                            reads=[rg(n.y)], writes=[rg(n.x)])
            elif g[0] == 'drop':
                _, n, cnt = g
                acc = acc_flag(n.x.buf)
                bwd.add(_lib.OP_DROPOUT, n.name, p=(self._aptr(n.y, True), _vp(n.mask), self._aptr(n.x, True)), i=(cnt, self.cdtype),
                        f=(1.0 / (1.0 - n.p),), flags=acc, lane=lane_of[n], reads=[rg(n.y), ('dm', id(n), 0, 1)], writes=[rg(n.x)])
            elif g[0] == 'flat':
                _, n, fi = g
                assert n.x.is_full
                acc = acc_flag(n.x.buf)
                bwd.add(_lib.OP_FLATTEN_CHW, n.name, p=(self._aptr(n.x, True), self._aptr(n.y, True)), i=fi, flags=acc,
                        lane=lane_of[n], reads=[rg(n.y)], writes=[rg(n.x)])
            elif g[0] == 'bnr':
                _, n, bnd = g
                bkey = n.bn_key
                # the slice's gradient: the FIRST backward op that touches the concatenation must cover all of it (densenet: the
                # transition / norm5 that reads the whole block output); later (= earlier-in-forward) layers accumulate
                first = n.x.buf.id not in written
                if first:
                    assert n.x.is_full, 'the first gradient written into a concatenation must cover all of it: ' + n.name
                acc = acc_flag(n.x.buf)
                bwd.add(_lib.OP_BN_BWD, n.name,
                        p=(self._aptr(n.x), self._aptr(n.y), self._aptr(n.y, True), self._pptr(bkey + '.weight'),
                           self._stat(n, 0), self._stat(n, 1), self._aptr(n.x, True), None, self._pptr(bkey + '.weight', 'G'),
                           self._pptr(bkey + '.bias', 'G'), self._stat(n, 2), self._stat(n, 3)),
                        i=(n.K, n.x.buf.C, 0), flags=8 if acc else 0, bn=bnd, lane=lane_of[n],
                        reads=[ra(n.x), ra(n.y), rg(n.y), rst(n)], writes=[rg(n.x)])
            elif g[0] == 'pool':
                _, n, pd, k = g
                assert n.x.is_full
                if n in fused_pool_nodes:
                    continue                      # gathered inside the producer's bn_bwd_maxpool
                if n in absorbed:
                    continue                      # its gradient reaches x through the sibling GEMM's input gradient
                acc = acc_flag(n.x.buf)
                if n.kind == 'max':
                    bwd.add(_lib.OP_MAXPOOL_BWD, n.name, p=(self._aptr(n.y, True), _vp(self.argmax[k]), self._aptr(n.x, True)), flags=acc, pool=pd,
                            lane=lane_of[n], reads=[rg(n.y), ram(k)], writes=[rg(n.x)])
                else:
                    bwd.add(_lib.OP_AVGPOOL_BWD, n.name, p=(self._aptr(n.y, True), self._aptr(n.x, True)), flags=acc, pool=pd,
                            lane=lane_of[n], reads=[rg(n.y)], writes=[rg(n.x)])
            else:
                _, n, d, bnd, draw, wT, needs_dgrad, di = g
                wgq = wg_of.get(n)
                own = wgq is not None or (self.wgrad_lane and (n.group is None or n.cpool is not None))
                if own:
                    draw = _vp(own_draw(n))               # kept until the (grouped / side-lane) weight gradient has read it
                elif self.wgrad_lane:
                    draw = _vp(group_draw(n.group), self.esize * n.koff)
                LW = None                                 # lane of this node's weight gradient (None: the node's own lane)
                if self.wgrad_lane and n in trunk:
                    LW = TLs[tl_next[0] % len(TLs)]       # (the trunk's weight gradients: every lane but the trunk's own)
                    tl_next[0] += 1
                elif self.wgrad_lane:
                    LW = WLs[wl_next[0] % len(WLs)]       # (round-robin over the weight-gradient lanes, per NODE)
                    wl_next[0] += 1
                ckey, bkey = n.conv_key + '.weight', n.bn_key
                dres, lddres, dres_acc = None, 0, 0
                if n.residual is not None:
                    assert n.residual.is_full
                    dres_acc = acc_flag(n.residual.buf)
                    dres, lddres = self._aptr(n.residual, True), n.residual.buf.C
                grp = n.group
                L = lane_of[n]
                cp = n.cpool is not None
                rdraw = [rdg(n)] if (grp is not None and not cp) else [('draw', di, 0, 1)]     # d(raw): merged-group slice or the lane's scratch
                if own:
                    rdraw = [('drawn', id(n), 0, 1)]
                rawp, _ld = self._raw_ptr(n)
                if n in fused_pool:
                    pn, pk = fused_pool[n]
                    ppd = PoolDesc(N, pn.x.H, pn.x.W, pn.x.C, _ld, 3, 3, 2, 2, pn.ph, pn.pw, pn.P, pn.Q, pn.y.buf.C, self.cdtype)
                    bwd.add(_lib.OP_BN_BWD_MAXPOOL, n.name + '+' + pn.name,
                            p=(rawp, self._aptr(pn.y, True), _vp(self.argmax[pk]), self._pptr(bkey + '.weight'),
                               self._stat(n, 0), self._stat(n, 1), self._stat(n, 2), self._stat(n, 3), draw,
                               self._pptr(bkey + '.weight', 'G'), self._pptr(bkey + '.bias', 'G')),
                            i=(1, n.K), pool=ppd, lane=L, reads=[rraw(n), rg(pn.y), ram(pk), rst(n)], writes=rdraw)
                elif n in bnstat_done:
                    ppart, nrow, pres, pld = bnstat_done[n]
                    bwd.add(_lib.OP_BN_BWD_PARTIALS, n.name,
                            p=(rawp, self._aptr(n.y, True), self._pptr(bkey + '.weight'), self._stat(n, 0), self._stat(n, 1),
                               self._stat(n, 2), self._stat(n, 3), ppart, draw, self._pptr(bkey + '.weight', 'G'),
                               self._pptr(bkey + '.bias', 'G')),
                            i=(n.y.buf.C, nrow, n.K if (grp is None or cp) else grp.Ktot, pld), bn=bnd, lane=L,
                            reads=[rraw(n), rg(n.y), rst(n), pres], writes=rdraw)
                else:
                    bwd.add(_lib.OP_BN_BWD, n.name,
                            p=(rawp, self._aptr(n.y), self._aptr(n.y, True), self._pptr(bkey + '.weight'),
                               self._stat(n, 0), self._stat(n, 1), draw, dres, self._pptr(bkey + '.weight', 'G'),
                               self._pptr(bkey + '.bias', 'G'), self._stat(n, 2), self._stat(n, 3)),
                            i=(n.y.buf.C, n.K if (grp is None or cp) else grp.Ktot, lddres), flags=dres_acc, bn=bnd, lane=L,
                            reads=[rraw(n), ra(n.y), rg(n.y), rst(n)],
                            writes=rdraw + ([rg(n.residual)] if n.residual is not None else []))
                if cp:
                    # d(pooled conv output) -> d(conv output): the average pool is its own transpose; into the branch's slice of
                    # the merged d(raw) scratch, from where the sibling GEMM's weight / input gradients pick it up
                    pn = n.cpool
                    ppd = PoolDesc(N, pn.x.H, pn.x.W, n.K, grp.Ktot, 3, 3, 1, 1, 1, 1, pn.P, pn.Q, n.K, self.cdtype)
                    bwd.add(_lib.OP_AVGPOOL_BWD, pn.name + '(' + n.name + ')',
                            p=(draw, _vp(group_draw(grp), self.esize * n.koff)), flags=0, pool=ppd, lane=L,
                            reads=rdraw, writes=[rdg(n)])
                if grp is not None:
                    # fused siblings: every member's d(raw) lands in its slice of the merged scratch; the member that
                    # comes FIRST in forward order is the last one here and launches the single wgrad + dgrad
                    if grp.members[0] is n:
                        gd = self._group_desc(grp, N)
                        gdraw = group_draw(grp)
                        gp = [_vp(gdraw)] + [self._pptr(m.conv_key + '.weight', 'G') for m in grp.members]
                        gres = ('dg', id(grp), 0, grp.Ktot)
                        bwd.add(_lib.OP_CONV_WGRAD_SEG, '+'.join(m.name for m in grp.members),
                                p=[self._aptr(grp.x), gp[0]] + gp[1:], i=[m.K for m in grp.members], conv=gd,
                                lane=0 if LW is None else LW, reads=[ra(grp.x), gres], writes=[])
                        acc = acc_flag(grp.x.buf)
                        tab = self._bs_table(grp, gd, readers, fused_pool) if (acc == 0 and fuse_level >= 2) else None
                        if tab is not None:
                            # the sibling GEMM is the ONLY consumer of this block input: its input-gradient kernel is the sole writer
                            # of the block-output gradient and reduces the BatchNorm-backward sums of every producer of that
                            # concatenation in its epilogue (per-chunk producer table); the producers skip their reduction pass
                            table, part, nrow, prods = tab
                            keep.extend([table, part])
                            rpt = ('bpt', id(grp), 0, 1)
                            bwd.add(_lib.OP_CONV_DGRAD_BNSTAT_TAB, '+'.join(m.name for m in grp.members),
                                    p=(_vp(gdraw), _vp(self.Wsh, self.esize * grp.wT_off), self._aptr(grp.x, True), _vp(table),
                                       _vp(part)), conv=gd, lane=0,
                                    reads=[gres] + [rraw(m) for m in prods] + [rst(m) for m in prods], writes=[rg(grp.x), rpt])
                            for m in prods:
                                bnstat_done[m] = (_vp(part, 4 * m.y.coff), nrow, rpt, grp.x.C)
                        else:
                            bwd.add(_lib.OP_CONV_DGRAD, '+'.join(m.name for m in grp.members),
                                    p=(_vp(gdraw), _vp(self.Wsh, self.esize * grp.wT_off), self._aptr(grp.x, True)),
                                    flags=acc, conv=gd, lane=0, reads=[gres], writes=[rg(grp.x)])
                    continue
                dbw = ConvDesc.from_buffer_copy(d)
                dbw.ldy = n.K                       # dy of the conv = the dense d(raw) scratch
                # wgrad and dgrad only share their input d(raw); IFCBK_WGRAD_SIDE=1 puts the weight gradient on the neighbouring
                # lane -- measured slower (31.5 vs 29.1 ms/step): the next node's bn_bwd must wait for it to release the scratch
                if wgq is not None:
                    wgq.setdefault('seen', []).append(n)
                    if len(wgq['seen']) == len(wgq['members']):
                        # the block's last member in backward order: every member's d(raw) exists now
                        mem = wgq['members']
                        items = (_lib.WgradItem * len(mem))()
                        for kq, (m, dq) in enumerate(zip(mem, wgq['descs'])):
                            items[kq].d = dq
                            items[kq].x = self._aptr(m.x).value
                            items[kq].dy = self.draw_own[m].data_ptr()
                            items[kq].dw = self._pptr(m.conv_key + '.weight', 'G').value
                        keep.append(items)
                        bwd.add(_lib.OP_CONV_WGRAD_GROUP, '+'.join(m.name for m in mem),
                                p=[C.addressof(items)] + [items[kq].dw for kq in range(len(mem))], i=(len(mem),), lane=L if LW is None else LW,
                                reads=[ra(m.x) for m in mem] + [('drawn', id(m), 0, 1) for m in mem], writes=[])
                elif self.stem_u8 is n and self.in_kind[self.in_slot] == 'u8':
                    bwd.add(_lib.OP_STEM_U8_WGRAD, n.name,
                            p=(_vp(self.in_u8[self.in_slot]), draw, _vp(self.in_ab[self.in_slot]), self._pptr(ckey, 'G')), conv=dbw,
                            lane=L if LW is None else LW, reads=[ra(n.x)] + rdraw, writes=[])
                else:
                    bwd.add(_lib.OP_CONV_WGRAD, n.name, p=(self._aptr(n.x), draw, self._pptr(ckey, 'G')), conv=dbw,
                            lane=LW if LW is not None else ((L + 1) % NL if self.wgrad_side_lane else L),
                            reads=[ra(n.x)] + rdraw, writes=[])
                if needs_dgrad:
                    assert n.x.is_full
                    acc = acc_flag(n.x.buf)
                    if n in bnstat_of and acc == 0:
                        pn = bnstat_of[n]
                        prow, pld = self._raw_ptr(pn)
                        nrow = self.ctx.lib.ifcbk_conv2d_dgrad_bnstat_mblocks(C.byref(dbw))
                        bwd.add(_lib.OP_CONV_DGRAD_BNSTAT, n.name,
                                p=(draw, wT, self._aptr(n.x, True), prow, self._stat(pn, 0), self._stat(pn, 1),
                                   self._stat(pn, 2), self._stat(pn, 3), _vp(self.bn_part[L])), i=(pld,), conv=dbw,
                                lane=L, reads=rdraw + [rraw(pn), rst(pn)], writes=[rg(n.x), rbp(L)])
                        bnstat_done[pn] = (_vp(self.bn_part[L]), nrow, rbp(L), 0)
                    else:
                        bwd.add(_lib.OP_CONV_DGRAD, n.name, p=(draw, wT, self._aptr(n.x, True)), flags=acc, conv=dbw,
                                lane=L, reads=rdraw, writes=[rg(n.x)])

        class PlanObj:
            pass
        pl = PlanObj()
        pl.N = N
        pl.fwd_train, pl.fwd_eval, pl.bwd = Program(fwd_t), Program(fwd_e), Program(bwd)
        pack_ops = pack                                # one OP_WEIGHT_PACK per conv (the bucketed update cuts this list)
        pack = self._pack_multi(pack)
        pl.pack, pl.evalprep = Program(pack), Program(evalprep)
        # loss ops: CE(main) + 0.4*CE(aux)   (neuston_models.py:70-78)
        lossl = OpList()
        main = [h for h in self.heads if not h.aux][0]
        auxh = [h for h in self.heads if h.aux]
        lossl.add(_lib.OP_SOFTMAX_XENT, 'loss', p=(_vp(main.logits), _vp(self.target), _vp(self.loss), _vp(main.dlogits)),
                  i=(N, net.NC), f=(1.0,))
        for h in auxh:
            lossl.add(_lib.OP_SOFTMAX_XENT, 'loss_aux', p=(_vp(h.logits), _vp(self.target), _vp(self.loss), _vp(h.dlogits)),
                      i=(N, net.NC), f=(0.4,), flags=1)
        pl.loss = Program(lossl)
        evl = OpList()
        evl.add(_lib.OP_SOFTMAX_XENT, 'val_loss', p=(_vp(main.logits), _vp(self.target), _vp(self.loss), None), i=(N, net.NC), f=(1.0,))
        pl.eval_loss = Program(evl)
        sm = OpList()
        sm.add(_lib.OP_SOFTMAX, 'softmax', p=(_vp(main.logits), _vp(self.probs)), i=(N, net.NC))
        pl.softmax = Program(sm)
        opt = OpList()
        if self.optimizer == 'sgd':
            # additive option (north_star "SGD/Adam step"; upstream only has Adam): torch.optim.SGD's update, momentum buffer = M
            opt.add(_lib.OP_SGD, 'sgd', p=(_vp(self.P), _vp(self.G), _vp(self.M) if self.momentum else None),
                    i=(self.nparam_padded, 1), f=(self.lr, self.momentum, 0.0, 1.0))
        else:
            opt.add(_lib.OP_ADAM, 'adam', p=(_vp(self.P), _vp(self.G), _vp(self.M), _vp(self.V)),
                    i=(self.nparam_padded, 1), f=(self.lr, self.betas[0], self.betas[1], self.eps, 0.0, 1.0))
        pl.adam = Program(opt)
        # the step's bookkeeping in the step's own op table: num_batches_tracked += 1 of every BatchNorm, loss_sum += loss
        cnt = OpList()
        cnt.add(_lib.OP_STEP_COUNTERS, 'step_counters', p=(_vp(self.nbt), _vp(self.loss_sum), _vp(self.loss)), i=(self.nbt.numel(),))
        # fused train step = fwd + loss + bwd + adam + pack (+ counters); round 5: the optimizer and the repack run PER BUCKET of the
        # flat gradient buffer, on the weight-gradient lane, as soon as backward has finished the bucket (_bucketed_update)
        allops = OpList()
        for prog_ops in (fwd_t, lossl) + self._bucketed_update(bwd, opt, pack_ops) + (cnt,):
            allops.extend(prog_ops)
        pl.step = Program(allops)
        pl.step_adam_idxs = pl.step.find(_lib.OP_SGD if self.optimizer == 'sgd' else _lib.OP_ADAM)
        pl.step_adam_idx = pl.step_adam_idxs[-1]
        # data-parallel variant: fwd+loss | backward segments (all-reduce launched after each) | adam+pack
        fl = OpList()
        fl.extend(fwd_t)
        fl.extend(lossl)
        pl.fwd_loss = Program(fl)
        fb = OpList()
        for prog_ops in (fwd_t, lossl, bwd):
            fb.extend(prog_ops)
        pl.fwd_bwd = Program(fb)          # the part of a train step whose launch arguments never change: hipGraph-capturable
        pl.graphs = {}
        ap = OpList()
        ap.extend(opt)
        ap.extend(pack)
        ap.extend(cnt)
        pl.adam_pack = Program(ap)
        pl.bwd_list = bwd
        pl.ddp_segs = None
        pl.keep = keep
        return pl

    def _bucketed_update(self, bwd, opt, pack_ops):
        """-> (backward ops with the optimizer + repack of every finished gradient bucket inserted, the last bucket's optimizer, its
        repack).  Backward finishes the flat gradient buffer from its tail (dp.segment_plan: the cuts of the data-parallel exchange);
        a bucket's optimizer op and the repack of its conv weights go onto the weight-gradient lane right behind the ops that
        complete it -- and behind the last input-gradient op that reads a bf16 shadow of the bucket (the shadow is rewritten by the
        repack) -- so that only the last bucket (the stem: 4 % of the parameters) is updated behind the end of backward.  Same
        arithmetic per element as one launch over the whole buffer (Adam / SGD are element-wise; the step count is a launch
        argument of every bucket's op).  IFCBK_OPT_BUCKETS: target number of buckets (6), <= 1 = one launch at the end."""
        nb = int(os.environ.get('IFCBK_OPT_BUCKETS', '6'))
        whole = (bwd, opt, self._pack_multi(pack_ops))
        if nb <= 1 or self.wgrad_side_lane:
            return whole
        from .dp import segment_plan
        offs = [self._op_param_offsets(o) for o in bwd.ops]
        try:
            segs = segment_plan(offs, self.palloc, self.nparam_padded, nb)
        except RuntimeError:
            return whole
        if len(segs) < 2:
            return whole
        # which parameters' shadows does an input-gradient op read?  (a sibling group's op: every member's)
        pbase, sbase = self.P.data_ptr(), self.Wsh.data_ptr()
        regions = [(g.wT_off, g.wT_off + g.Ktot * g.x.C) for g in self.groups]

        def skey(elem_off):
            for (r0, r1) in regions:
                if r0 <= elem_off < r1:
                    return r0
            return elem_off
        owners = {}
        for o in pack_ops.ops:
            if o.p[2]:
                owners.setdefault(skey((o.p[2] - sbase) // self.esize), []).append((o.p[0] - pbase) // 4)
        dkinds = (_lib.OP_CONV_DGRAD, _lib.OP_CONV_DGRAD_BNSTAT, _lib.OP_CONV_DGRAD_BNSTAT_TAB)
        last_reader = {}                       # parameter offset -> index of the last backward op that reads its shadow
        shreads = {}                           # backward op index -> [('SH', off, off + size)]
        for k, o in enumerate(bwd.ops):
            if o.kind in dkinds:
                own = owners.get(skey((o.p[1] - sbase) // self.esize))
                if own is None:
                    return whole               # an input gradient whose filter this plan cannot attribute: keep the single update
                shreads[k] = [('SH', off, off + self.palloc[off]) for off in own]
                for off in own:
                    last_reader[off] = k
        LW = self.NL - 1 if self.wgrad_lane else 0
        inserts = {}                           # op index -> OpList to emit BEFORE that op (i.e. behind op index - 1)
        prev_ins = 0
        for (b0, b1, lo, hi) in segs[:-1]:
            ins = max([b1, prev_ins] + [1 + k for off, k in last_reader.items() if lo <= off < hi])
            prev_ins = ins
            ol = inserts.setdefault(ins, OpList())
            o0 = opt.ops[0]
            ptrs = [(o0.p[j] + 4 * lo) if o0.p[j] else None for j in range(4 if self.optimizer != 'sgd' else 3)]
            ol.add(o0.kind, opt.tags[0] + '[%d:%d]' % (lo, hi), p=ptrs, i=(hi - lo, int(o0.i[1])), f=tuple(o0.f[j] for j in range(8)),
                   lane=LW, reads=[('G', lo, hi)], writes=[('P', lo, hi)])
            sub = OpList()
            for o, tg, mt in zip(pack_ops.ops, pack_ops.tags, pack_ops.meta):
                if lo <= (o.p[0] - pbase) // 4 < hi:
                    sub.ops.append(o); sub.tags.append(tg); sub.meta.append(mt)
            if sub.ops:
                pm = self._pack_multi(sub, key=(lo, hi))
                pm.meta[0] = (LW, [('P', lo, hi)], [('SH', lo, hi)])
                ol.extend(pm)
        out = OpList()
        for k, (o, tg, mt) in enumerate(zip(bwd.ops, bwd.tags, bwd.meta)):
            if k in inserts:
                out.extend(inserts[k])
            lane, reads, writes = mt
            if not (reads is None and writes is None):
                extra_w = [('G', off, off + self.palloc[off]) for off in offs[k]]
                mt = (lane, list(reads or ()) + shreads.get(k, []), list(writes or ()) + extra_w)
            out.ops.append(o); out.tags.append(tg); out.meta.append(mt)
        for k in sorted(inserts):
            if k >= len(bwd.ops):
                out.extend(inserts[k])
        # the last bucket: behind the end of backward, as full barriers (as the single update was)
        lo, hi = segs[-1][2], segs[-1][3]
        last_opt, last_pack = OpList(), OpList()
        o0 = opt.ops[0]
        ptrs = [(o0.p[j] + 4 * lo) if o0.p[j] else None for j in range(4 if self.optimizer != 'sgd' else 3)]
        last_opt.add(o0.kind, opt.tags[0] + '[%d:%d]' % (lo, hi), p=ptrs, i=(hi - lo, int(o0.i[1])), f=tuple(o0.f[j] for j in range(8)))
        sub = OpList()
        for o, tg, mt in zip(pack_ops.ops, pack_ops.tags, pack_ops.meta):
            if lo <= (o.p[0] - pbase) // 4 < hi:
                sub.ops.append(o); sub.tags.append(tg); sub.meta.append(mt)
        if sub.ops:
            last_pack = self._pack_multi(sub, key=(lo, hi))
        return (out, last_opt, last_pack)

    def _pack_multi(self, pack, key=None):
        """fold the per-conv weight_pack ops into ONE multi-tensor launch (device-side item table, built once per ``key``:
        None = every conv of the network, (lo, hi) = the convs of one optimizer bucket)."""
        import numpy as np
        cache = self.__dict__.setdefault('_pack_tables', {})
        if key not in cache:
            items = (_lib.PackItem * len(pack.ops))()
            blk = 0
            for k, o in enumerate(pack.ops):
                d = o.u.conv
                it = items[k]
                it.w_master, it.w, it.wT = o.p[0], o.p[1], o.p[2]
                it.K, it.RS, it.C, it.Cw = d.K, d.R * d.S, d.C, d.Cw
                it.wT_ld = int(o.i[0])
                it.first_block = blk
                blk += ((d.K + 31) // 32) * d.R * d.S * ((d.C + 31) // 32)      # 32x32 (k, c) tiles per filter tap
            raw = np.frombuffer(bytes(items), dtype=np.uint8).copy()
            cache[key] = (torch.from_numpy(raw).to(self.dev), len(pack.ops), blk)
        tab, n, blocks = cache[key]
        if key is None:
            self._pack_items, self._pack_n, self._pack_blocks = tab, n, blocks
        out = OpList()
        out.add(_lib.OP_WEIGHT_PACK_MULTI, 'weight_pack' + ('' if key is None else '[%d:%d]' % key), p=(_vp(tab),),
                i=(n, blocks, self.cdtype))
        return out

    def _op_param_offsets(self, o):
        """element offsets (into the flat gradient buffer) of the parameter tensors a backward op writes."""
        base = self.G.data_ptr()
        if o.kind == _lib.OP_CONV_WGRAD:
            return [(o.p[2] - base) // 4]
        if o.kind == _lib.OP_STEM_U8_WGRAD:
            return [(o.p[3] - base) // 4]
        if o.kind == _lib.OP_CONV_WGRAD_GROUP:
            return [(o.p[1 + k] - base) // 4 for k in range(int(o.i[0]))]
        if o.kind == _lib.OP_CONV_WGRAD_SEG:
            return [(o.p[2 + k] - base) // 4 for k in range(4) if o.i[k] > 0]
        # (a vgg*_bn conv's bias rides with its BatchNorm's gradients: nothing writes it -- the batch mean absorbs the bias, its
        # gradient is identically zero -- but the bucket plan must see the whole flat buffer covered)
        if o.kind == _lib.OP_BN_BWD:
            g0 = (o.p[8] - base) // 4
            return [g0, (o.p[9] - base) // 4] + self.cbias_after.get(g0, [])
        if o.kind in (_lib.OP_BN_BWD_MAXPOOL, _lib.OP_BN_BWD_PARTIALS):
            g0 = (o.p[9] - base) // 4
            return [g0, (o.p[10] - base) // 4] + self.cbias_after.get(g0, [])
        if o.kind == _lib.OP_HEAD_BWD:
            return [(o.p[4] - base) // 4, (o.p[5] - base) // 4] if o.p[4] else []
        if o.kind == _lib.OP_BIAS_RELU_BWD:
            return [(o.p[3] - base) // 4] if o.p[3] else []
        return []

    def ddp_segments(self, pl, nseg=8):
        """split the backward list into <= nseg runs whose finished gradients form a contiguous tail
        [lo, prev_lo) of the flat buffer, so each run's all-reduce can start while later runs compute."""
        if pl.ddp_segs is not None:
            return pl.ddp_segs
        from .dp import segment_plan
        padded = self.palloc
        ops = pl.bwd_list.ops
        segs = []
        for (b0, b1, lo, hi) in segment_plan([self._op_param_offsets(o) for o in ops], padded, self.nparam_padded, nseg):
            segs.append((Program(pl.bwd_list.slice(b0, b1)), b1, lo, hi))
        pl.ddp_segs = segs
        return segs

    def train_step_ddp(self, N, world, all_reduce, mark=None):
        """data-parallel step: gradient all-reduce (sum) of each finished tail bucket is launched right after
        the backward segment that completes it and overlaps the remaining backward; Adam divides by world."""
        self._check_train_batch(N)
        if int(world) != self.dp_world and 'IFCBK_LANES' not in os.environ:
            raise RuntimeError('train_step_ddp(world=%d): this engine chose its program lanes for world %d (built before the process '
                               'group existed and without WORLD_SIZE?); pass dp_world=%d to Engine or set IFCBK_LANES'
                               % (int(world), self.dp_world, int(world)))
        pl = self.plan(N)
        self.ensure_packed(pl)
        self.make_dropout_mask(N)
        self.run(pl.fwd_loss)
        from .dp import run_overlapped
        run_overlapped(self.ddp_segments(pl), lambda seg: self.run(seg[0]), self.G, all_reduce, mark)
        self.step_count += 1
        self._set_update(pl.adam_pack.arr[0], 1.0 / world)
        self.run(pl.adam_pack)                    # (Adam, repack, and the step's counters: nbt += 1, loss_sum += loss)
        self.eval_stats_ready = False
        return pl

    # ------------------------------------------------------------------ lifetime
    def close(self):
        """destroy every captured graph of this engine, then its library context (arenas, lane streams, events).  The library
        enforces the same order by itself (a ctx owns its graphs); doing it here keeps the handles in ``pl.graphs`` from dangling."""
        ctx = getattr(self, 'ctx', None)
        if ctx is None or getattr(ctx, 'h', None) is None:
            return
        for pl in getattr(self, '_plans', {}).values():
            for g in pl.graphs.values():
                ctx.lib.ifcbk_graph_destroy(ctx.h, g)
            pl.graphs.clear()
        for c in (getattr(self, 'pre_ctx', None), ctx):
            if c is not None:
                c.close()

    def plan_graph_handles(self):
        return [g for pl in self._plans.values() for g in pl.graphs.values()]

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ execution
    def stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)

    def run(self, prog, op_ms=None):
        self.ctx.run_program(prog.arr, prog.n, self.stream(), op_ms)

    def replay(self, pl, name):
        """run program ``name`` of plan ``pl`` as a hipGraph (captured on first use: all lanes, with their fork / wait /
        join edges); buffers are static per plan, so the baked pointers stay valid."""
        g = pl.graphs.get(name)
        if g is None:
            prog = getattr(pl, name)
            g = pl.graphs[name] = self.ctx.capture(prog.arr, prog.n)
        self.ctx.graph_launch(g, self.stream())

    def ensure_packed(self, pl):
        if not self.packed:
            self.run(pl.pack)
            self.packed = True

    def load_input_nchw(self, x):
        """fp32 NCHW [N,3,S,S] (device) -> the plan's NHWC bf16 input buffer (+ [TV] transform_input)."""
        N, Cc, H, W = x.shape
        S = self.net.S
        if Cc != 3 or H != S or W != S:
            raise RuntimeError('expected input [N,3,%d,%d], got %s' % (S, S, tuple(x.shape)))
        x = x.to(device=self.dev, dtype=torch.float32).contiguous()
        sc = sh = None
        if self.net.transform_input:
            sc = (C.c_float * 3)(0.229 / 0.5, 0.224 / 0.5, 0.225 / 0.5)
            sh = (C.c_float * 3)((0.485 - 0.5) / 0.5, (0.456 - 0.5) / 0.5, (0.406 - 0.5) / 0.5)
        self.ctx.call('ifcbk_nchw_to_nhwc', _vp(x), N, 3, S, S, 8, self.cdtype, sc, sh, _vp(self.act[self.net.input.id]),
                      self.stream())
        self.in_kind[self.in_slot] = 'nhwc'
        return N

    def load_rois(self, pixels, offs, hs, ws, max_h, max_w, in_channels=1, flips=None, mean=None, std=None, slot=None):
        """ragged u8 ROIs (device tensors) -> input buffer via the PIL-exact resize kernel.  slot: the input slot of a
        prefetch (runs on the prefetch stream with the prefetch context's workspace); None = the current slot, current stream."""
        if slot is not None:
            return self._load_rois_into(self.pre_ctx, C.c_void_p(self.pre_stream.cuda_stream), slot, False,
                                        pixels, offs, hs, ws, max_h, max_w, in_channels, flips, mean, std)
        return self._load_rois_into(self.ctx, self.stream(), self.in_slot, True, pixels, offs, hs, ws, max_h,
                                    max_w, in_channels, flips, mean, std)

    def _load_rois_into(self, ctx, stream, slot, main, pixels, offs, hs, ws, max_h, max_w, in_channels, flips, mean, std):
        n = hs.numel()
        dst = self.in_bufs[slot]
        d = RoiDesc()
        d.n_img, d.S, d.in_channels, d.out_channels = n, self.net.S, in_channels, 8
        d.flip_bits_valid = 1 if flips is not None else 0
        d.dtype = self.cdtype
        for k in range(3):
            d.mean[k] = 0.0 if mean is None else float(mean[k])
            d.std[k] = 1.0 if std is None else float(std[k])
            d.tin_scale[k], d.tin_shift[k] = 1.0, 0.0
        if self.net.transform_input:
            for k, (s, m) in enumerate(((0.229, 0.485), (0.224, 0.456), (0.225, 0.406))):
                d.tin_scale[k], d.tin_shift[k] = s / 0.5, (m - 0.5) / 0.5
        need = ctx.lib.ifcbk_roi_preprocess_workspace(C.byref(d), int(max_h), int(max_w))
        if need > ctx.lib.ifcbk_ctx_workspace_bytes(ctx.h):
            if not main:
                self.pre_stream.synchronize()        # an earlier prefetch may still read the arena that is about to move
            ctx.reserve(need)
            if main:
                for pl in self._plans.values():          # the workspace moved: graphs captured so far hold stale pointers
                    for g in pl.graphs.values():
                        self.ctx.lib.ifcbk_graph_destroy(self.ctx.h, g)
                    pl.graphs.clear()
        if self.stem_u8 is not None and in_channels == 1:
            # grey ROIs: only the resized u8 plane is written; x_c = a_c * g + b_c (ToTensor, Normalize, transform_input) goes to the
            # stem conv as six floats
            ab = tuple(d.tin_scale[k] / (255.0 * d.std[k]) for k in range(3)) + \
                tuple(d.tin_shift[k] - d.tin_scale[k] * d.mean[k] / d.std[k] for k in range(3))
            if self._in_ab_host[slot] != ab:
                with torch.cuda.stream(torch.cuda.current_stream(self.dev) if main else self.pre_stream):
                    self.in_ab[slot].copy_(torch.tensor(ab, dtype=torch.float32), non_blocking=False)
                self._in_ab_host[slot] = ab
            ctx.call('ifcbk_roi_preprocess', C.byref(d), _vp(pixels), _vp(offs), _vp(hs), _vp(ws), _vp(flips),
                     int(max_h), int(max_w), None, _vp(self.in_u8[slot]), stream)
            self.in_kind[slot] = 'u8'
            return n
        ctx.call('ifcbk_roi_preprocess', C.byref(d), _vp(pixels), _vp(offs), _vp(hs), _vp(ws), _vp(flips),
                 int(max_h), int(max_w), _vp(dst), None, stream)
        self.in_kind[slot] = 'nhwc'
        return n

    def make_dropout_mask(self, N):
        ext = self.external_mask
        for h in self.heads:
            if h.dropout:
                if ext is not None and not isinstance(ext, dict):
                    h.mask[:N].copy_(ext[:N].to(torch.uint8))
                elif isinstance(ext, dict) and h.name in ext:
                    h.mask[:N].copy_(ext[h.name][:N].to(torch.uint8))
                else:
                    self.ctx.call('ifcbk_dropout_mask', _vp(h.mask), N * h.C, 0.5, self.dropout_seed,
                                  self.dropout_calls * (1 << 24), self.stream())
        for k, n in enumerate(self.drops):
            # nn.Dropout layers of a classifier stack (alexnet / vgg / squeezenet): one keep-mask per layer and step; a parity
            # test hands masks over as {node name: [B, H*W*C] in NHWC element order}
            if isinstance(ext, dict) and n.name in ext:
                n.mask[:N].copy_(ext[n.name][:N].reshape(N, -1).to(torch.uint8))
            else:
                self.ctx.call('ifcbk_dropout_mask', _vp(n.mask), N * n.mask.shape[1], float(n.p),
                              self.dropout_seed + 0x9E3779B1 * (k + 1), self.dropout_calls * (1 << 26), self.stream())
        self.dropout_calls += 1

    def _check_train_batch(self, N):
        if N > self.train_batch and N <= self.window_batch:
            raise RuntimeError('batch %d > %d: this engine was built for inference (train_batch=%d): its gradient buffers hold %d image(s)'
                               % (N, self.train_batch, self.train_batch, self.train_batch))
        if N > self.window_batch:
            raise RuntimeError('batch %d > %d: the 2 GiB buffer-descriptor window holds %d images of this network per launch, and '
                               'BatchNorm batch statistics cannot be taken over chunks: use a smaller --batch per GPU (more GPUs)'
                               % (N, self.window_batch, self.window_batch))

    def forward_train(self, N):
        self._check_train_batch(N)
        pl = self.plan(N)
        self.ensure_packed(pl)
        self.make_dropout_mask(N)
        self.run(pl.fwd_train)
        self.nbt += 1
        self.eval_stats_ready = False
        return pl

    def forward_eval(self, N):
        pl = self.plan(N)
        self.ensure_packed(pl)
        if not self.eval_stats_ready:
            self.run(pl.evalprep)
            self.eval_stats_ready = True
        if self.graph_eval:
            self.replay(pl, 'fwd_eval')
        else:
            self.run(pl.fwd_eval)
        return pl

    def backward(self, N):
        self._check_train_batch(N)
        self.run(self.plan(N).bwd)

    def _set_update(self, op, grad_scale=1.0):
        """per-step scalars of the optimizer op: the step count (Adam's bias correction) and the gradient scale (1/world)"""
        op.i[1] = self.step_count
        op.f[3 if self.optimizer == 'sgd' else 5] = grad_scale

    def adam_step(self, N):
        pl = self.plan(N)
        self.step_count += 1
        self._set_update(pl.adam.arr[0])
        self.run(pl.adam)
        self.packed = False
        self.eval_stats_ready = False

    def train_step(self, N, op_ms=None, ev_slot=None, ev_arr=None):
        """one fused launch list: fwd + CE(+0.4 aux) + bwd + Adam + weight repack; loss stays on device."""
        self._check_train_batch(N)
        pl = self.plan(N)
        self.ensure_packed(pl)
        self.make_dropout_mask(N)
        self.step_count += 1
        for j in pl.step_adam_idxs:
            self._set_update(pl.step.arr[j])
        if ev_slot is not None:
            # ev_arr: pl.step.timed(...) -- which ops to bracket with HIP events (default: all)
            if ev_arr is None:
                if getattr(pl, 'step_timed_all', None) is None:
                    pl.step_timed_all = pl.step.timed()
                ev_arr = pl.step_timed_all
            for j in pl.step_adam_idxs:
                self._set_update(ev_arr[j])
            self.ctx.call('ifcbk_run_program_ev', ev_arr, pl.step.n, self.stream(), int(ev_slot))
        elif self.graph_train and op_ms is None:
            # fwd + loss + bwd replayed as one hipGraph; Adam (its step count is a launch argument) + repack stay plain launches
            self.replay(pl, 'fwd_bwd')
            self._set_update(pl.adam_pack.arr[0])
            self.run(pl.adam_pack)
        else:
            self.ctx.run_program(pl.step.arr, pl.step.n, self.stream(), op_ms)
        # (nbt += 1 and loss_sum += loss are the step's last op: no framework kernel between load_rois and the end of a step)
        self.eval_stats_ready = False
        return pl
