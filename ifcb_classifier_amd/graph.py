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


class PlainConvNode:
    """conv (+bias) (+ReLU) with no BatchNorm behind it (alexnet / vgg / squeezenet convs, every nn.Linear of a classifier
    stack -- a 1x1 conv on a [N,1,1,C] tensor --, densenet's growth convs): writes its activation directly"""
    kind = 'cb'

    def __init__(self, **kw):
        self.__dict__.update(kw)


class DropNode:
    kind = 'drop'

    def __init__(self, **kw):
        self.__dict__.update(kw)


class FlatNode:
    """torch.flatten(x, 1) of the NCHW tensor: [N,H,W,C] -> [N,1,1,C*H*W] in (c, h, w) order"""
    kind = 'flat'

    def __init__(self, **kw):
        self.__dict__.update(kw)


class BnNode:
    """BatchNorm2d -> ReLU in front of a conv (densenet's pre-activation order), on a channel slice of a concatenation"""
    kind = 'bnr'

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
                cin_real=None, aux=False, conv_bias=False, init=None):
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
                        eps=self.bn_eps, conv_bias=conv_bias, init=init)
        self.nodes.append(node)
        self.params.append((conv_key + '.weight', (cout, cw, R, S), 'conv', node))
        if conv_bias:
            # vgg*_bn: Conv2d(bias=True) -> BatchNorm2d.  The batch mean absorbs the bias: it only shifts running_mean
            self.params.append((conv_key + '.bias', (cout,), 'bn_cbias', node))
        self.params.append((bn_key + '.weight', (cout,), 'bn_w', node))
        self.params.append((bn_key + '.bias', (cout,), 'bn_b', node))
        self.buffers.append((bn_key + '.running_mean', (cout,), node))
        self.buffers.append((bn_key + '.running_var', (cout,), node))
        return out

    def basic(self, x, cout, k, name, stride=1, pad=0, out=None, aux=False, cin_real=None):
        """[TV] inception.BasicConv2d: <name>.conv / <name>.bn"""
        return self.conv_bn(x, cout, k, stride, pad, name + '.conv', name + '.bn', True, out, None, cin_real, aux)

    def pool(self, kind, x, k, stride, pad=0, out=None, name='pool', aux=False, ceil_mode=False):
        R, S = _pair(k)
        sh, sw = _pair(stride)
        ph, pw = _pair(pad)
        P, Q = _out(x.H, R, sh, ph), _out(x.W, S, sw, pw)
        if ceil_mode:          # torch: ceil the division, then drop a last window that would START in the right/bottom padding
            P, Q = -(-(x.H + 2 * ph - R) // sh) + 1, -(-(x.W + 2 * pw - S) // sw) + 1
            if (P - 1) * sh >= x.H + ph:
                P -= 1
            if (Q - 1) * sw >= x.W + pw:
                Q -= 1
        if out is None:
            out = self.new_buf(name, P, Q, x.C).full()
        assert out.C == x.C and out.H == P and out.W == Q
        node = PoolNode(kind=kind, name=name, x=x, y=out, R=R, S=S, sh=sh, sw=sw, ph=ph, pw=pw, P=P, Q=Q, aux=aux)
        self.nodes.append(node)
        return out

    def head(self, x, key, dropout, aux=False, fc=True):
        """global average pool (+dropout) + Linear; fc=False: the pooled channels ARE the logits (squeezenet)"""
        node = HeadNode(name=key, x=x, C=x.C, HW=x.H * x.W, NC=self.NC, dropout=dropout, aux=aux, key=key, fc=fc)
        self.nodes.append(node)
        if fc:
            self.params.append((key + '.weight', (self.NC, x.C), 'fc_w', node))
            self.params.append((key + '.bias', (self.NC,), 'fc_b', node))
        return node

    def conv_plain(self, x, cout, k, key, stride=1, pad=0, relu=True, bias=True, out=None, cin_real=None, linear=False,
                   init='default', k_real=None):
        """conv (+bias) (+ReLU), no BatchNorm.  linear: an nn.Linear (its weight is [out, in] in the state_dict).
        k_real: the layer's true output channels when `cout` was padded up to a whole 16-byte chunk"""
        R, S = _pair(k)
        sh, sw = _pair(stride)
        ph, pw = _pair(pad)
        P, Q = _out(x.H, R, sh, ph), _out(x.W, S, sw, pw)
        if out is None:
            out = self.new_buf(key + ':y', P, Q, cout).full()
        assert out.C == cout and out.H == P and out.W == Q, (key, out.C, cout, out.H, P)
        cw = cin_real if cin_real is not None else x.C
        kr = cout if k_real is None else k_real
        node = PlainConvNode(name=key, x=x, y=out, K=cout, K_real=kr, R=R, S=S, sh=sh, sw=sw, ph=ph, pw=pw, P=P, Q=Q, relu=relu,
                             bias=bias, Cw=cw, key=key, aux=False, linear=linear, init=init, residual=None)
        self.nodes.append(node)
        self.params.append((key + '.weight', (kr, cw) if linear else (kr, cw, R, S), 'lin_w' if linear else 'conv', node))
        if bias:
            self.params.append((key + '.bias', (kr,), 'cbias', node))
        return out

    def dropout(self, x, name, p=0.5):
        assert x.is_full
        out = self.new_buf(name, x.H, x.W, x.C).full()
        self.nodes.append(DropNode(name=name, x=x, y=out, p=p, aux=False))
        return out

    def flatten(self, x, name):
        out = self.new_buf(name, 1, 1, x.H * x.W * x.C).full()
        self.nodes.append(FlatNode(name=name, x=x, y=out, aux=False))
        return out

    def bn_relu(self, x, bn_key, relu=True):
        out = self.new_buf(bn_key + ':y', x.H, x.W, x.C).full()
        node = BnNode(name=bn_key, x=x, y=out, K=x.C, bn_key=bn_key, relu=relu, eps=self.bn_eps, aux=False)
        self.nodes.append(node)
        self.params.append((bn_key + '.weight', (x.C,), 'bn_w', node))
        self.params.append((bn_key + '.bias', (x.C,), 'bn_b', node))
        self.buffers.append((bn_key + '.running_mean', (x.C,), node))
        self.buffers.append((bn_key + '.running_var', (x.C,), node))
        return out


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


# ------------------------------------------------------------------------------------------ alexnet / vgg
def _classifier_stack(net, x, c_hidden, keys, drop_first):
    """[TV] AlexNet.classifier = Dropout, Linear, ReLU, Dropout, Linear, ReLU, Linear  (drop_first)
       [TV] VGG.classifier     = Linear, ReLU, Dropout, Linear, ReLU, Dropout, Linear
    on torch.flatten(avgpool(features), 1); the AdaptiveAvgPool2d in front is the identity at the 224-pixel input the
    reference feeds these backbones (neuston_net.py: resize 224 for everything but inception)."""
    x = net.flatten(x, 'flatten')
    k1, k2, k3 = keys
    if drop_first:
        x = net.dropout(x, 'classifier.drop0')
        x = net.conv_plain(x, c_hidden, 1, k1, linear=True)
        x = net.dropout(x, 'classifier.drop1')
        x = net.conv_plain(x, c_hidden, 1, k2, linear=True)
    else:
        x = net.conv_plain(x, c_hidden, 1, k1, linear=True, init='normal01')
        x = net.dropout(x, 'classifier.drop0')
        x = net.conv_plain(x, c_hidden, 1, k2, linear=True, init='normal01')
        x = net.dropout(x, 'classifier.drop1')
    net.head(x, k3, dropout=False)           # the replaced Linear(4096, num_o_classes): nn.Linear default init


def alexnet(num_classes, in_size=224):
    """[TV] alexnet.py: features 0 3 6 8 10 (conv+bias+ReLU), MaxPool2d(3, 2) after 0, 3 and 10"""
    net = Net('alexnet', in_size, num_classes, 1e-5)
    x = net.input.full()
    x = net.conv_plain(x, 64, 11, 'features.0', stride=4, pad=2, cin_real=3)
    x = net.pool('max', x, 3, 2, name='features.2')
    x = net.conv_plain(x, 192, 5, 'features.3', pad=2)
    x = net.pool('max', x, 3, 2, name='features.5')
    x = net.conv_plain(x, 384, 3, 'features.6', pad=1)
    x = net.conv_plain(x, 256, 3, 'features.8', pad=1)
    x = net.conv_plain(x, 256, 3, 'features.10', pad=1)
    x = net.pool('max', x, 3, 2, name='features.12')
    assert (x.H, x.W) == (6, 6), 'AdaptiveAvgPool2d((6, 6)) is only the identity for 224-pixel inputs'
    _classifier_stack(net, x, 4096, ('classifier.1', 'classifier.4', 'classifier.6'), drop_first=True)
    return net


_VGG = {'11': [64, 'M', 128, 'M', 256, 256, 'M', 512, 512, 'M', 512, 512, 'M'],
        '13': [64, 64, 'M', 128, 128, 'M', 256, 256, 'M', 512, 512, 'M', 512, 512, 'M'],
        '16': [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M'],
        '19': [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512, 512, 512, 512, 'M']}
VGG_NAMES = tuple('vgg' + d + b for d in _VGG for b in ('', '_bn'))


def vgg(name, num_classes, in_size=224):
    """[TV] vgg.py make_layers: Conv2d(3x3, pad 1) [, BatchNorm2d], ReLU per entry, MaxPool2d(2, 2) per 'M'; module indices of
    ``features`` count every layer (conv, bn, relu, pool), as the state_dict keys do"""
    bn = name.endswith('_bn')
    net = Net(name, in_size, num_classes, 1e-5)
    x = net.input.full()
    idx, first = 0, True
    for v in _VGG[name[3:5]]:
        if v == 'M':
            x = net.pool('max', x, 2, 2, name='features.%d' % idx)
            idx += 1
            continue
        if bn:
            x = net.conv_bn(x, v, 3, 1, 1, 'features.%d' % idx, 'features.%d' % (idx + 1), cin_real=3 if first else None,
                            conv_bias=True, init='kaiming_out')
            idx += 3
        else:
            x = net.conv_plain(x, v, 3, 'features.%d' % idx, pad=1, cin_real=3 if first else None, init='kaiming_out')
            idx += 2
        first = False
    assert (x.H, x.W) == (7, 7), 'AdaptiveAvgPool2d((7, 7)) is only the identity for 224-pixel inputs'
    _classifier_stack(net, x, 4096, ('classifier.0', 'classifier.3', 'classifier.6'), drop_first=False)
    return net


# ------------------------------------------------------------------------------------------ squeezenet1_1
def squeezenet1_1(num_classes, in_size=224):
    """[TV] squeezenet.py, version 1_1; Fire = squeeze 1x1 -> ReLU -> cat(expand1x1 -> ReLU, expand3x3 -> ReLU).  The reference
    replaces classifier[1] with Conv2d(512, num_o_classes, 1) (neuston_models.py:30-33)"""
    net = Net('squeezenet', in_size, num_classes, 1e-5)
    x = net.input.full()
    x = net.conv_plain(x, 64, 3, 'features.0', stride=2, cin_real=3, init='kaiming_uniform')
    x = net.pool('max', x, 3, 2, name='features.2', ceil_mode=True)

    def fire(x, idx, sq, e1, e3):
        pre = 'features.%d' % idx
        s = net.conv_plain(x, sq, 1, pre + '.squeeze', init='kaiming_uniform')
        out = net.new_buf(pre + ':cat', s.H, s.W, e1 + e3)
        net.conv_plain(s, e1, 1, pre + '.expand1x1', out=out.view(0, e1), init='kaiming_uniform')
        net.conv_plain(s, e3, 3, pre + '.expand3x3', pad=1, out=out.view(e1, e3), init='kaiming_uniform')
        return out.full()

    x = fire(x, 3, 16, 64, 64)
    x = fire(x, 4, 16, 64, 64)
    x = net.pool('max', x, 3, 2, name='features.5', ceil_mode=True)
    x = fire(x, 6, 32, 128, 128)
    x = fire(x, 7, 32, 128, 128)
    x = net.pool('max', x, 3, 2, name='features.8', ceil_mode=True)
    x = fire(x, 9, 48, 192, 192)
    x = fire(x, 10, 48, 192, 192)
    x = fire(x, 11, 64, 256, 256)
    x = fire(x, 12, 64, 256, 256)
    x = net.dropout(x, 'classifier.0')
    ncp = (num_classes + 7) // 8 * 8
    x = net.conv_plain(x, ncp, 1, 'classifier.1', k_real=num_classes)        # the replaced conv: nn.Conv2d default init
    net.head(x, 'classifier', dropout=False, fc=False)
    return net


# ------------------------------------------------------------------------------------------ densenet
_DENSENETS = {'densenet121': (32, (6, 12, 24, 16), 64), 'densenet161': (48, (6, 12, 36, 24), 96),
              'densenet169': (32, (6, 12, 32, 32), 64), 'densenet201': (32, (6, 12, 48, 32), 64)}


def densenet(name, num_classes, in_size=224, bn_size=4):
    """[TV] densenet.py: conv0/norm0/relu0/pool0, dense blocks of (norm1, relu1, conv1 1x1, norm2, relu2, conv2 3x3) layers whose
    outputs are concatenated, transitions (norm, relu, conv 1x1, AvgPool2d(2, 2)), norm5, relu, global average pool, classifier.
    Concatenation never materialises: every layer's conv2 writes its growth-rate channels into the block's one buffer."""
    growth, blocks, c0 = _DENSENETS[name]
    net = Net(name, in_size, num_classes, 1e-5)
    x = net.input.full()
    x = net.conv_bn(x, c0, 7, 2, 3, 'features.conv0', 'features.norm0', cin_real=3, init='kaiming_in')
    nf = c0
    pooled_hw = _out(x.H, 3, 2, 1)
    for bi, nl in enumerate(blocks):
        ctot = nf + nl * growth
        cat = net.new_buf('features.denseblock%d:cat' % (bi + 1), pooled_hw, pooled_hw, ctot)
        if bi == 0:
            net.pool('max', x, 3, 2, 1, out=cat.view(0, nf), name='features.pool0')
        else:
            net.pool('avg', x, 2, 2, out=cat.view(0, nf), name='features.transition%d.pool' % bi)
        for li in range(nl):
            pre = 'features.denseblock%d.denselayer%d' % (bi + 1, li + 1)
            cin = nf + li * growth
            t = net.bn_relu(cat.view(0, cin), pre + '.norm1')
            t = net.conv_bn(t, bn_size * growth, 1, 1, 0, pre + '.conv1', pre + '.norm2', init='kaiming_in')
            net.conv_plain(t, growth, 3, pre + '.conv2', pad=1, relu=False, bias=False, out=cat.view(cin, growth), init='kaiming_in')
        nf = ctot
        if bi != len(blocks) - 1:
            tp = 'features.transition%d' % (bi + 1)
            t = net.bn_relu(cat.full(), tp + '.norm')
            x = net.conv_plain(t, nf // 2, 1, tp + '.conv', relu=False, bias=False, init='kaiming_in')
            nf //= 2
            pooled_hw = _out(x.H, 2, 2, 0)
        else:
            x = net.bn_relu(cat.full(), 'features.norm5')
    net.head(x, 'classifier', dropout=False)
    return net


def build(model_name, num_classes, pretrained=False):
    """Graph twin of ``get_namebrand_model`` (neuston_models.py:22-45).  Unknown names -> KeyError."""
    if model_name == 'inception_v3':
        return inception_v3(num_classes, transform_input=bool(pretrained))
    if model_name in _RESNETS:
        return resnet(model_name, num_classes)
    if model_name == 'alexnet':
        return alexnet(num_classes)
    if model_name == 'squeezenet':
        return squeezenet1_1(num_classes)
    if model_name in VGG_NAMES:
        return vgg(model_name, num_classes)
    if model_name in _DENSENETS:
        return densenet(model_name, num_classes)
    if model_name.startswith(('vgg', 'densenet')):
        # the reference does getattr(torchvision.models, model_name): an AttributeError for names torchvision does not have
        raise AttributeError("module 'torchvision.models' has no attribute '%s'" % model_name)
    raise KeyError("model unknown!")
