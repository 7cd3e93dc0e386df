"""Static layer graphs of the backbones ``get_namebrand_model`` serves (reference:
``/root/reference/neuston_models.py:22-45``), expressed as a flat node list over NHWC buffers.

Concatenation never materialises: every branch of an Inception block writes into its channel slice of
one block-output buffer ("concat-free").  Node creation order is torchvision's module registration order,
so the parameter list (and therefore ``state_dict`` / optimizer-state order) matches [TV] 0.8.2.
"""


class Buf:
    """An NHWC activation buffer [N,H,W,C] (N is bound by the plan)."""

    def __init__(self, name, H, W, C, is_input=False):
        self.name, self.H, self.W, self.C, self.is_input = name, H, W, C, is_input
        self.id = None

    def view(self, coff=0, C=None):
        return View(self, coff, self.C if C is None else C)

    def full(self):
        return View(self, 0, self.C)


class View:
    """Channel slice [coff, coff+C) of a Buf."""

    def __init__(self, buf, coff, C):
        assert coff % 8 == 0 and C % 8 == 0 and coff + C <= buf.C, (buf.name, coff, C, buf.C)
        self.buf, self.coff, self.C = buf, coff, C

    @property
    def H(self):
        return self.buf.H

    @property
    def W(self):
        return self.buf.W

    @property
    def is_full(self):
        return self.coff == 0 and self.C == self.buf.C


class ConvNode:
    kind = 'conv'

    def __init__(self, **kw):
        self.__dict__.update(kw)


class PoolNode:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class HeadNode:
    kind = 'head'

    def __init__(self, **kw):
        self.__dict__.update(kw)


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


def _out(h, k, s, p):
    return (h + 2 * p - k) // s + 1


class Net:
    def __init__(self, name, in_size, num_classes, bn_eps):
        self.name, self.S, self.NC, self.bn_eps = name, in_size, num_classes, bn_eps
        self.nodes = []
        self.bufs = []
        self.params = []      # (key, shape, kind, node) in [TV] registration order
        self.buffers = []     # (key, shape)  BN running stats
        self.transform_input = False
        self.has_aux = False
        self.input = self.new_buf('input', in_size, in_size, 8, is_input=True)   # 3 channels padded to 8

    def new_buf(self, name, H, W, C, is_input=False):
        b = Buf(name, H, W, C, is_input)
        b.id = len(self.bufs)
        self.bufs.append(b)
        return b

    def conv_bn(self, x, cout, k, stride=1, pad=0, conv_key=None, bn_key=None, relu=True, out=None, residual=None,
                cin_real=None, aux=False):
        R, S = _pair(k)
        sh, sw = _pair(stride)
        ph, pw = _pair(pad)
        P, Q = _out(x.H, R, sh, ph), _out(x.W, S, sw, pw)
        raw = self.new_buf(conv_key + ':raw', P, Q, cout)
        if out is None:
            out = self.new_buf(conv_key + ':y', P, Q, cout).full()
        assert out.C == cout and out.H == P and out.W == Q, (conv_key, out.C, cout, out.H, P)
        cw = cin_real if cin_real is not None else x.C
        node = ConvNode(name=conv_key, x=x, raw=raw, y=out, K=cout, R=R, S=S, sh=sh, sw=sw, ph=ph, pw=pw, P=P, Q=Q,
                        relu=relu, residual=residual, Cw=cw, conv_key=conv_key, bn_key=bn_key, aux=aux,
                        eps=self.bn_eps)
        self.nodes.append(node)
        self.params.append((conv_key + '.weight', (cout, cw, R, S), 'conv', node))
        self.params.append((bn_key + '.weight', (cout,), 'bn_w', node))
        self.params.append((bn_key + '.bias', (cout,), 'bn_b', node))
        self.buffers.append((bn_key + '.running_mean', (cout,), node))
        self.buffers.append((bn_key + '.running_var', (cout,), node))
        return out

    def basic(self, x, cout, k, name, stride=1, pad=0, out=None, aux=False, cin_real=None):
        """[TV] inception.BasicConv2d: <name>.conv / <name>.bn"""
        return self.conv_bn(x, cout, k, stride, pad, name + '.conv', name + '.bn', True, out, None, cin_real, aux)

    def pool(self, kind, x, k, stride, pad=0, out=None, name='pool', aux=False):
        R, S = _pair(k)
        sh, sw = _pair(stride)
        ph, pw = _pair(pad)
        P, Q = _out(x.H, R, sh, ph), _out(x.W, S, sw, pw)
        if out is None:
            out = self.new_buf(name, P, Q, x.C).full()
        assert out.C == x.C and out.H == P and out.W == Q
        node = PoolNode(kind=kind, name=name, x=x, y=out, R=R, S=S, sh=sh, sw=sw, ph=ph, pw=pw, P=P, Q=Q, aux=aux)
        self.nodes.append(node)
        return out

    def head(self, x, key, dropout, aux=False):
        node = HeadNode(name=key, x=x, C=x.C, HW=x.H * x.W, NC=self.NC, dropout=dropout, aux=aux, key=key)
        self.nodes.append(node)
        self.params.append((key + '.weight', (self.NC, x.C), 'fc_w', node))
        self.params.append((key + '.bias', (self.NC,), 'fc_b', node))
        return node


# ------------------------------------------------------------------------------------------ inception_v3
def _inception_a(net, x, name, pf):
    out = net.new_buf(name, x.H, x.W, 64 + 64 + 96 + pf)
    net.basic(x, 64, 1, name + '.branch1x1', out=out.view(0, 64))
    t = net.basic(x, 48, 1, name + '.branch5x5_1')
    net.basic(t, 64, 5, name + '.branch5x5_2', pad=2, out=out.view(64, 64))
    t = net.basic(x, 64, 1, name + '.branch3x3dbl_1')
    t = net.basic(t, 96, 3, name + '.branch3x3dbl_2', pad=1)
    net.basic(t, 96, 3, name + '.branch3x3dbl_3', pad=1, out=out.view(128, 96))
    p = net.pool('avg', x, 3, 1, 1, name=name + ':avgpool')
    net.basic(p, pf, 1, name + '.branch_pool', out=out.view(224, pf))
    return out.full()


def _inception_b(net, x, name):
    H2 = _out(x.H, 3, 2, 0)
    out = net.new_buf(name, H2, H2, 384 + 96 + x.C)
    net.basic(x, 384, 3, name + '.branch3x3', stride=2, out=out.view(0, 384))
    t = net.basic(x, 64, 1, name + '.branch3x3dbl_1')
    t = net.basic(t, 96, 3, name + '.branch3x3dbl_2', pad=1)
    net.basic(t, 96, 3, name + '.branch3x3dbl_3', stride=2, out=out.view(384, 96))
    net.pool('max', x, 3, 2, 0, out=out.view(480, x.C), name=name + ':maxpool')
    return out.full()


def _inception_c(net, x, name, c7):
    out = net.new_buf(name, x.H, x.W, 768)
    net.basic(x, 192, 1, name + '.branch1x1', out=out.view(0, 192))
    t = net.basic(x, c7, 1, name + '.branch7x7_1')
    t = net.basic(t, c7, (1, 7), name + '.branch7x7_2', pad=(0, 3))
    net.basic(t, 192, (7, 1), name + '.branch7x7_3', pad=(3, 0), out=out.view(192, 192))
    t = net.basic(x, c7, 1, name + '.branch7x7dbl_1')
    t = net.basic(t, c7, (7, 1), name + '.branch7x7dbl_2', pad=(3, 0))
    t = net.basic(t, c7, (1, 7), name + '.branch7x7dbl_3', pad=(0, 3))
    t = net.basic(t, c7, (7, 1), name + '.branch7x7dbl_4', pad=(3, 0))
    net.basic(t, 192, (1, 7), name + '.branch7x7dbl_5', pad=(0, 3), out=out.view(384, 192))
    p = net.pool('avg', x, 3, 1, 1, name=name + ':avgpool')
    net.basic(p, 192, 1, name + '.branch_pool', out=out.view(576, 192))
    return out.full()


def _inception_d(net, x, name):
    H2 = _out(x.H, 3, 2, 0)
    out = net.new_buf(name, H2, H2, 320 + 192 + x.C)
    t = net.basic(x, 192, 1, name + '.branch3x3_1')
    net.basic(t, 320, 3, name + '.branch3x3_2', stride=2, out=out.view(0, 320))
    t = net.basic(x, 192, 1, name + '.branch7x7x3_1')
    t = net.basic(t, 192, (1, 7), name + '.branch7x7x3_2', pad=(0, 3))
    t = net.basic(t, 192, (7, 1), name + '.branch7x7x3_3', pad=(3, 0))
    net.basic(t, 192, 3, name + '.branch7x7x3_4', stride=2, out=out.view(320, 192))
    net.pool('max', x, 3, 2, 0, out=out.view(512, x.C), name=name + ':maxpool')
    return out.full()


def _inception_e(net, x, name):
    out = net.new_buf(name, x.H, x.W, 2048)
    net.basic(x, 320, 1, name + '.branch1x1', out=out.view(0, 320))
    t = net.basic(x, 384, 1, name + '.branch3x3_1')
    net.basic(t, 384, (1, 3), name + '.branch3x3_2a', pad=(0, 1), out=out.view(320, 384))
    net.basic(t, 384, (3, 1), name + '.branch3x3_2b', pad=(1, 0), out=out.view(704, 384))
    t = net.basic(x, 448, 1, name + '.branch3x3dbl_1')
    t = net.basic(t, 384, 3, name + '.branch3x3dbl_2', pad=1)
    net.basic(t, 384, (1, 3), name + '.branch3x3dbl_3a', pad=(0, 1), out=out.view(1088, 384))
    net.basic(t, 384, (3, 1), name + '.branch3x3dbl_3b', pad=(1, 0), out=out.view(1472, 384))
    p = net.pool('avg', x, 3, 1, 1, name=name + ':avgpool')
    net.basic(p, 192, 1, name + '.branch_pool', out=out.view(1856, 192))
    return out.full()


def inception_v3(num_classes, transform_input=False, in_size=299):
    net = Net('inception_v3', in_size, num_classes, 1e-3)
    net.transform_input = transform_input
    net.has_aux = True
    x = net.input.full()
    x = net.basic(x, 32, 3, 'Conv2d_1a_3x3', stride=2, cin_real=3)
    x = net.basic(x, 32, 3, 'Conv2d_2a_3x3')
    x = net.basic(x, 64, 3, 'Conv2d_2b_3x3', pad=1)
    x = net.pool('max', x, 3, 2, name='maxpool1')
    x = net.basic(x, 80, 1, 'Conv2d_3b_1x1')
    x = net.basic(x, 192, 3, 'Conv2d_4a_3x3')
    x = net.pool('max', x, 3, 2, name='maxpool2')
    x = _inception_a(net, x, 'Mixed_5b', 32)
    x = _inception_a(net, x, 'Mixed_5c', 64)
    x = _inception_a(net, x, 'Mixed_5d', 64)
    x = _inception_b(net, x, 'Mixed_6a')
    x = _inception_c(net, x, 'Mixed_6b', 128)
    x = _inception_c(net, x, 'Mixed_6c', 160)
    x = _inception_c(net, x, 'Mixed_6d', 160)
    x = _inception_c(net, x, 'Mixed_6e', 192)
    # AuxLogits (train only): avg_pool2d(5,3) -> conv0 1x1 -> conv1 5x5 -> GAP -> fc
    a = net.pool('avg', x, 5, 3, name='AuxLogits:avgpool', aux=True)
    a = net.basic(a, 128, 1, 'AuxLogits.conv0', aux=True)
    a = net.basic(a, 768, 5, 'AuxLogits.conv1', aux=True)
    net.head(a, 'AuxLogits.fc', dropout=False, aux=True)
    x = _inception_d(net, x, 'Mixed_7a')
    x = _inception_e(net, x, 'Mixed_7b')
    x = _inception_e(net, x, 'Mixed_7c')
    net.head(x, 'fc', dropout=True)
    return net


# ------------------------------------------------------------------------------------------ resnet
_RESNETS = {'resnet18': ('basic', [2, 2, 2, 2]), 'resnet34': ('basic', [3, 4, 6, 3]),
            'resnet50': ('bottleneck', [3, 4, 6, 3]), 'resnet101': ('bottleneck', [3, 4, 23, 3]),
            'resnet152': ('bottleneck', [3, 8, 36, 3])}


def resnet(name, num_classes, in_size=224):
    block, layers = _RESNETS[name]
    exp = 1 if block == 'basic' else 4
    net = Net(name, in_size, num_classes, 1e-5)
    x = net.input.full()
    x = net.conv_bn(x, 64, 7, 2, 3, 'conv1', 'bn1', cin_real=3)
    x = net.pool('max', x, 3, 2, 1, name='maxpool')
    inplanes = 64
    for li, (planes, n) in enumerate(zip([64, 128, 256, 512], layers)):
        for bi in range(n):
            stride = 2 if (bi == 0 and li > 0) else 1
            pre = 'layer%d.%d' % (li + 1, bi)
            need_ds = bi == 0 and (stride != 1 or inplanes != planes * exp)
            if block == 'basic':
                t = net.conv_bn(x, planes, 3, stride, 1, pre + '.conv1', pre + '.bn1')
                y = net.conv_bn(t, planes, 3, 1, 1, pre + '.conv2', pre + '.bn2', relu=True, residual=None)
                last = net.nodes[-1]
            else:
                t = net.conv_bn(x, planes, 1, 1, 0, pre + '.conv1', pre + '.bn1')
                t = net.conv_bn(t, planes, 3, stride, 1, pre + '.conv2', pre + '.bn2')
                y = net.conv_bn(t, planes * 4, 1, 1, 0, pre + '.conv3', pre + '.bn3', relu=True, residual=None)
                last = net.nodes[-1]
            idt = x
            if need_ds:
                idt = net.conv_bn(x, planes * exp, 1, stride, 0, pre + '.downsample.0', pre + '.downsample.1',
                                  relu=False)
                # execution order: the downsample branch must run before the residual add -> move it in front of `last`
                ds = net.nodes.pop()
                net.nodes.insert(net.nodes.index(last), ds)
            last.residual = idt
            inplanes = planes * exp
            x = y
    net.head(x, 'fc', dropout=False)
    return net


def build(model_name, num_classes, pretrained=False):
    """Graph twin of ``get_namebrand_model`` (neuston_models.py:22-45).  Unknown names -> KeyError."""
    if model_name == 'inception_v3':
        return inception_v3(num_classes, transform_input=bool(pretrained))
    if model_name in _RESNETS:
        return resnet(model_name, num_classes)
    raise KeyError("model unknown!")
