"""Drop-in twin of ``/root/reference/neuston_models.py``: ``get_namebrand_model`` (:22-45) and
``NeustonModel`` (:48-190), with the torchvision backbone replaced by the MI355X engine.

``get_namebrand_model`` returns an ``nn.Module`` honouring the reference's module contract: callable on
fp32 NCHW ``[B,3,S,S]``; train mode returns ``InceptionOutputs(logits, aux_logits)`` for inception_v3 and a
plain tensor otherwise; ``state_dict()`` keys/shapes are torchvision's (OIHW fp32 views of the engine's flat
KRSC master buffer); ``parameters()`` in [TV] registration order; ``loss.backward()`` drives the HIP backward
through one autograd node and leaves ``param.grad`` as views of the flat gradient buffer.
"""
import argparse
from collections import namedtuple

import numpy as np
import torch
import torch.nn as nn

from . import graph
from .engine import Engine

import os
import pickle
import types

InceptionOutputs = namedtuple('InceptionOutputs', ['logits', 'aux_logits'])
InceptionOutputs.__annotations__ = {'logits': torch.Tensor, 'aux_logits': torch.Tensor}


class _NetFn(torch.autograd.Function):
    """whole-network autograd node: forward already ran; backward runs the HIP backward program."""

    @staticmethod
    def forward(ctx, hook, module, N, *outs):
        ctx.module, ctx.N = module, N
        return tuple(o.clone() for o in outs)

    @staticmethod
    def backward(ctx, *gouts):
        m, N = ctx.module, ctx.N
        eng = m.engine
        for h, g in zip(m._train_heads, gouts):
            if g is None:
                h.dlogits[:N].zero_()
            else:
                h.dlogits[:N].copy_(g)
        eng.backward(N)
        for key, p in m._pmap.items():
            gv = eng.gviews[key]
            if p.grad is None:
                p.grad = gv
            elif p.grad.data_ptr() != gv.data_ptr():
                p.grad.add_(gv)
        return (None, None, None) + (None,) * len(gouts)


class HipBackbone(nn.Module):
    def __init__(self, net, device=0, max_batch=None, dtype='bf16', **engine_kw):
        super().__init__()
        max_batch = max_batch or 32
        cap = Engine.capacity_limit(net, dtype)
        if max_batch > cap:
            # (neuston_net RUN --batch 2048: results are per image, the RUN loop cuts its batches to what the engine holds)
            print('%s: batch capacity %d -> %d images per program (largest verified activation)' % (net.name, max_batch, cap))
            max_batch = cap
        object.__setattr__(self, 'engine', Engine(net, device, max_batch, dtype=dtype, **engine_kw))
        object.__setattr__(self, 'net', net)
        eng = self.engine
        pmap = {}
        for key, shape, kind, node in net.params:
            mod, leaf = self._submodule(key)
            p = nn.Parameter(eng.pviews[key], requires_grad=True)
            mod.register_parameter(leaf, p)
            pmap[key] = p
        object.__setattr__(self, '_pmap', pmap)
        for k, (key, shape, node) in enumerate(net.buffers):
            mod, leaf = self._submodule(key)
            mod.register_buffer(leaf, eng.bviews[key])
            if leaf == 'running_var':
                mod.register_buffer('num_batches_tracked', eng.nbt[eng.bn_index[node]])
        object.__setattr__(self, '_hook', torch.zeros(1, device=eng.dev, requires_grad=True))
        object.__setattr__(self, '_train_heads', sorted(eng.heads, key=lambda h: h.aux))
        self.num_classes = net.NC
        if self._cbias_pairs():
            self._register_load_state_dict_pre_hook(self._cbias_pre_hook)
        eng.init_weights()

    def _submodule(self, key):
        parts = key.split('.')
        mod = self
        for p in parts[:-1]:
            if p not in mod._modules:
                mod.add_module(p, nn.Module())
            mod = mod._modules[p]
        return mod, parts[-1]

    # parameters may be modified behind the engine's back (optimizer.step, load_state_dict)
    def _load_from_state_dict(self, *a, **k):
        super()._load_from_state_dict(*a, **k)
        self.engine.params_changed()

    # vgg*_bn: Conv2d(bias=True) -> BatchNorm2d.  The batch mean absorbs the bias, so the engine convolves without it and its
    # running_mean tracks the bias-free conv output; the state_dict shows torchvision's tensor (mean of conv + bias).
    def _cbias_pairs(self):
        return [(n.conv_key + '.bias', n.bn_key + '.running_mean') for n in self.engine.convs if getattr(n, 'conv_bias', False)]

    def state_dict(self, *a, **k):
        sd = super().state_dict(*a, **k)
        prefix = k.get('prefix', a[1] if len(a) > 1 else '')
        for bk, rk in self._cbias_pairs():
            if prefix + rk in sd and prefix + bk in sd:
                sd[prefix + rk] = sd[prefix + rk] + sd[prefix + bk].to(sd[prefix + rk].device)
        return sd

    def _cbias_pre_hook(self, state_dict, prefix, *unused):
        # runs inside load_state_dict (direct or through a parent module) before this module's tensors are copied; the dict is
        # load_state_dict's own shallow copy
        for bk, rk in self._cbias_pairs():
            if prefix + bk in state_dict and prefix + rk in state_dict:
                rm = state_dict[prefix + rk]
                state_dict[prefix + rk] = rm - state_dict[prefix + bk].to(rm.device)

    def load_state_dict(self, state_dict, strict=True):
        r = super().load_state_dict(state_dict, strict)
        self.engine.params_changed()
        return r

    def set_dropout_mask(self, mask):
        """parity hook: fix the Bernoulli keep-mask of the train-mode dropout (None = generate): a [B, 2048] tensor for
        inception_v3's head, or {node name: [B, features]} for the nn.Dropout layers of alexnet / vgg / squeezenet."""
        self.engine.external_mask = mask

    def forward(self, x):
        eng = self.engine
        if x.shape[0] > eng.max_batch:
            raise RuntimeError('batch %d > capacity %d: construct with a larger max_batch' % (x.shape[0], eng.max_batch))
        if self.training and x.shape[0] > eng.window_batch:
            # (an EVAL batch beyond the window is one program all the same: the library cuts its convolutions into launches over
            # image groups -- samples are independent)
            raise RuntimeError('batch %d > %d: the 2 GiB buffer-descriptor window holds %d images of this network per launch, and BatchNorm '
                               'batch statistics cannot be taken over chunks: use a smaller --batch per GPU (more GPUs)'
                               % (x.shape[0], eng.window_batch, eng.window_batch))
        N = eng.load_input_nchw(x)
        if self.training:
            if torch.is_grad_enabled():
                eng.params_changed()          # an external optimizer may have stepped since the last call
            eng.forward_train(N)
            outs = [h.logits[:N] for h in self._train_heads]
            if torch.is_grad_enabled():
                outs = _NetFn.apply(self._hook, self, N, *outs)
            else:
                outs = [o.clone() for o in outs]
            if self.net.has_aux:
                return InceptionOutputs(outs[0], outs[1])
            return outs[0]
        eng.params_changed()
        eng.forward_eval(N)
        return self._train_heads[0].logits[:N].clone()


class _MissingClass(dict):
    """stand-in for a class whose module is not installed here (pytorch_lightning's AttributeDict, callback classes ...):
    keeps dict items, attribute state and constructor arguments so that the rest of the checkpoint can be read"""

    def __init__(self, *a, **k):
        dict.__init__(self)
        self._args, self._kwargs = a, k

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.__dict__.update(state)
        else:
            self.__dict__['_state'] = state

    def __reduce__(self):
        return (dict, (dict(self),))


def _tolerant_pickle():
    """``pickle_module`` for torch.load: a [PL] 1.3.x checkpoint keys ``checkpoint['callbacks']`` by callback CLASS objects and
    stores ``hyper_parameters`` as ``pytorch_lightning.utilities.parsing.AttributeDict`` -- unpickling a reference-trained
    ``.ptl`` therefore imports pytorch_lightning (absent here).  Unknown classes resolve to dict-like stand-ins."""
    made = {}

    class Unpickler(pickle.Unpickler):
        def find_class(self, module, name):
            try:
                return super().find_class(module, name)
            except (ImportError, AttributeError):
                key = module + '.' + name
                if key not in made:
                    made[key] = type(name, (_MissingClass,), {'__module__': module})
                return made[key]

    mod = types.ModuleType('ifcbk_tolerant_pickle')
    mod.Unpickler = Unpickler
    mod.load = lambda f, **kw: Unpickler(f, **kw).load()
    mod.loads = pickle.loads
    mod.__name__ = 'pickle'
    return mod


def load_checkpoint_file(path):
    """read a ``.ptl`` / ``.ckpt`` written by this package OR by the reference under Lightning 1.3.8 (f-2)"""
    try:
        return torch.load(path, map_location='cpu', weights_only=False)
    except (ImportError, AttributeError, ModuleNotFoundError):
        return torch.load(path, map_location='cpu', weights_only=False, pickle_module=_tolerant_pickle())


def load_pretrained_weights(backbone, path):
    """``--weights PATH``: a torchvision ``state_dict`` (e.g. inception_v3_google-1a9a5a14.pth) or a checkpoint holding one,
    standing in for the download ``pretrained=True`` triggers upstream (neuston_models.py:23-42).  As upstream, the
    classifier heads were replaced for ``num_o_classes`` AFTER the ImageNet weights were loaded: tensors whose shape differs
    (fc, AuxLogits.fc) keep their fresh initialisation.  Returns (loaded, skipped) key lists."""
    try:
        # a plain torchvision state_dict (the usual third-party download) needs no arbitrary unpickling
        sd = torch.load(path, map_location='cpu', weights_only=True)
    except pickle.UnpicklingError as e:
        # the safe loader refused a global: the file is a full checkpoint (Lightning-style .ptl / .ckpt holding a state_dict).
        # Unpickling that runs whatever the file says, so it needs the caller's word that the file is trusted; I/O errors and
        # everything else propagate unchanged
        if os.environ.get('IFCBK_TRUST_WEIGHTS', '0') in ('', '0'):
            raise RuntimeError('--weights %s is not a plain state_dict (%s); loading a full checkpoint unpickles arbitrary '
                               'objects: set IFCBK_TRUST_WEIGHTS=1 if you trust this file' % (path, str(e).splitlines()[0])) from e
        sd = load_checkpoint_file(path)        # the permissive loader of f-2
    if isinstance(sd, dict) and 'state_dict' in sd:
        sd = sd['state_dict']
    sd = {(k[len('model.'):] if k.startswith('model.') else k): v for k, v in sd.items()}
    own = backbone.state_dict()
    take = {k: v for k, v in sd.items() if k in own and tuple(v.shape) == tuple(own[k].shape)}
    skipped = [k for k in own if k not in take]
    heads = ('fc.', 'AuxLogits.fc.')
    bad = [k for k in skipped if not k.startswith(heads) and not k.endswith('num_batches_tracked')]
    if bad:
        raise RuntimeError('--weights %s does not fit this backbone: missing or mis-shaped tensors %s%s'
                           % (path, bad[:5], ' ...' if len(bad) > 5 else ''))
    own.update(take)
    backbone.load_state_dict(own)
    return sorted(take), skipped


def get_namebrand_model(model_name, num_o_classes, pretrained=False, device=0, max_batch=None, dtype='bf16', **engine_kw):
    """``neuston_models.py:22-45``.  Backbones on the HIP path: every family the reference accepts -- inception_v3, alexnet,
    squeezenet (1_1), vgg11/13/16/19(_bn), resnet18/34/50/101/152, densenet121/161/169/201.
    ``pretrained=True`` cannot download ImageNet weights here (no torchvision / network): it switches on
    inception's ``transform_input`` exactly as torchvision does and expects a ``load_state_dict`` to follow.
    Names the reference rejects raise its ``KeyError("model unknown!")`` (``AttributeError`` for a vgg* / densenet* name
    torchvision does not have, as ``getattr(torchvision.models, name)`` would)."""
    net = graph.build(model_name, num_o_classes, pretrained)
    return HipBackbone(net, device, max_batch, dtype, **engine_kw)


class NeustonModel(nn.Module):
    """``neuston_models.py:48-180`` without Lightning: same hooks, same loss, same optimizer, same
    aggregation.  ``training_step`` / ``validation_step`` / ``test_step`` accept the reference's batch tuples.
    The fused fast path (``fit_batch``) runs forward+loss+backward+Adam as one HIP program."""

    def __init__(self, hparams, device=0, max_batch=None, train_batch=None):
        super().__init__()
        if isinstance(hparams, dict):
            hparams = argparse.Namespace(**hparams)
        self.hparams = hparams
        self.criterion = nn.CrossEntropyLoss()
        mb = max_batch or getattr(hparams, 'batch_size', None) or 32
        opt = str(getattr(hparams, 'optimizer', None) or 'Adam').lower()
        self.model = get_namebrand_model(hparams.MODEL, len(hparams.classes), hparams.pretrained, device, mb,
                                         getattr(hparams, 'precision', 'bf16') or 'bf16', optimizer=opt,
                                         lr=float(getattr(hparams, 'learning_rate', None) or 0.001),
                                         momentum=float(getattr(hparams, 'momentum', None) or 0.0), train_batch=train_batch)
        self.best_val_loss = np.inf
        self.best_epoch = 0
        self.agg_train_loss = 0.0
        self.current_epoch = 0
        self.logged = {}

    def configure_optimizers(self):
        eng = self.model.engine
        if eng.optimizer == 'sgd':            # additive option; the reference's only behaviour is Adam(lr=0.001) (:63-64)
            from torch.optim import SGD
            return SGD(self.parameters(), lr=eng.lr, momentum=eng.momentum)
        from torch.optim import Adam
        return Adam(self.parameters(), lr=eng.lr)

    def forward(self, inputs):
        return self.model(inputs)

    def loss(self, inputs, outputs):
        if isinstance(outputs, tuple) and len(outputs) == 2:  # inception_v3
            outputs, aux_outputs = outputs
            loss1 = self.criterion(outputs, inputs)
            loss2 = self.criterion(aux_outputs, inputs)
            batch_loss = loss1 + 0.4 * loss2
        else:
            batch_loss = self.criterion(outputs, inputs)
        return batch_loss

    # TRAINING (reference hook: autograd through the HIP backward, any torch optimizer) #
    def training_step(self, batch, batch_nb):
        input_data, input_classes, input_src = batch
        outputs = self.forward(input_data)
        batch_loss = self.loss(input_classes.to(outputs[0].device if isinstance(outputs, tuple) else outputs.device), outputs)
        self.agg_train_loss += batch_loss.item()
        return dict(loss=batch_loss)

    # TRAINING (fast path: one fused HIP program; loss accumulated on device, read once per epoch) #
    def fit_batch(self, input_data, input_classes):
        eng = self.model.engine
        if torch.is_tensor(input_data):
            N = eng.load_input_nchw(input_data)
        else:
            N = eng.load_rois(**input_data)
        eng.target[:N].copy_(input_classes, non_blocking=True)
        self.model.train()
        eng.train_step(N)
        return N

    # ---- pipelined form of the three batch calls: the NEXT batch is uploaded and preprocessed on a side stream into the
    # engine's other input slot while the current step runs (Engine.prefetch_begin / prefetch_end / use_prefetched):
    #     n = model.stage_batch(first);  for each batch: model.use_staged(); n2 = model.stage_batch(next); model.fit_current(n)
    def stage_batch(self, rois, transform=None, input_classes=None):
        from .neuston_data import rois_to_device
        eng = self.model.engine
        slot, side = eng.prefetch_begin()
        with torch.cuda.stream(side):
            kw = rois_to_device(rois, eng.dev, transform)
            n = eng.load_rois(slot=slot, **kw)
            if input_classes is not None:
                eng.tgt_bufs[slot][:n].copy_(input_classes, non_blocking=True)
        eng.prefetch_end(slot)
        return n

    def stage_resident(self, res, i0, i1, transform=None):
        """stage ROIs i0..i1-1 of a bin whose .roi blob and ADC table already live on the device (``res`` from
        ``upload_bin``): the preprocess kernel reads them where they are -- no per-batch slicing, concatenation or upload"""
        eng = self.model.engine
        slot, side = eng.prefetch_begin()
        with torch.cuda.stream(side):
            side.wait_event(res['ready'])
            kw = dict(pixels=res['pixels'], offs=res['offs'][i0:i1], hs=res['hs'][i0:i1], ws=res['ws'][i0:i1],
                      max_h=res['max_h'], max_w=res['max_w'], in_channels=1)
            if transform is not None and transform.img_norm is not None:
                kw['mean'], kw['std'] = transform.img_norm
            n = eng.load_rois(slot=slot, **kw)
        eng.prefetch_end(slot)
        return n

    def upload_bin(self, blob, offs, hs, ws):
        """one upload per bin: the raw .roi bytes and the offset / size table of its ROIs"""
        eng = self.model.engine
        if len(hs) and (int(hs.min()) < 1 or int(ws.min()) < 1 or int(offs.min()) < 0
                        or int((offs.astype('int64') + hs.astype('int64') * ws).max()) > blob.size):
            raise ValueError('upload_bin: the ROI table has empty / negative entries or points past the %d-byte blob' % blob.size)
        slot_stream = eng.prefetch_stream()
        with torch.cuda.stream(slot_stream):
            host = torch.from_numpy(blob).pin_memory()
            res = dict(host=host, pixels=host.to(eng.dev, non_blocking=True),      # (the pinned copy lives as long as the bin's batches)
                       offs=torch.from_numpy(offs).to(eng.dev, non_blocking=True),
                       hs=torch.from_numpy(hs).to(eng.dev, non_blocking=True), ws=torch.from_numpy(ws).to(eng.dev, non_blocking=True),
                       max_h=int(hs.max()), max_w=int(ws.max()))
            res['ready'] = torch.cuda.Event()
            res['ready'].record(slot_stream)
        return res

    def use_staged(self):
        self.model.engine.use_prefetched()

    def fit_current(self, N, world=1, all_reduce=None):
        eng = self.model.engine
        self.model.train()
        if world > 1:
            eng.train_step_ddp(N, world, all_reduce)
        else:
            eng.train_step(N)
        return N

    def eval_current(self, N, with_loss=False):
        eng = self.model.engine
        self.model.eval()
        pl = eng.forward_eval(N)
        loss = None
        if with_loss:
            eng.run(pl.eval_loss)
            loss = eng.loss[0].clone()
        eng.run(pl.softmax)
        return eng.probs[:N].clone(), loss

    def fit_batch_ddp(self, input_data, input_classes, world, all_reduce):
        """data-parallel twin of fit_batch: per-rank local BatchNorm statistics (no SyncBN upstream), gradient
        all-reduce over RCCL launched per finished bucket and overlapped with the rest of backward."""
        eng = self.model.engine
        N = eng.load_input_nchw(input_data) if torch.is_tensor(input_data) else eng.load_rois(**input_data)
        eng.target[:N].copy_(input_classes, non_blocking=True)
        self.model.train()
        eng.train_step_ddp(N, world, all_reduce)
        return N

    def eval_batch(self, input_data, input_classes=None):
        """fast path of validation_step / test_step (:94-103, :152-157): eval forward -> [CE] -> softmax, all on
        device, no host sync.  Returns (probs[N,NC] device tensor, loss 0-d device tensor or None)."""
        eng = self.model.engine
        N = eng.load_input_nchw(input_data) if torch.is_tensor(input_data) else eng.load_rois(**input_data)
        self.model.eval()
        pl = eng.forward_eval(N)
        loss = None
        if input_classes is not None:
            eng.target[:N].copy_(input_classes, non_blocking=True)
            eng.run(pl.eval_loss)
            loss = eng.loss[0].clone()
        eng.run(pl.softmax)
        return eng.probs[:N].clone(), loss

    def epoch_train_loss(self):
        eng = self.model.engine
        v = float(eng.loss_sum.item())
        eng.loss_sum.zero_()
        return v

    # Validation #
    def validation_step(self, batch, batch_idx):
        input_data, input_classes, input_src = batch
        with torch.no_grad():
            outputs = self.forward(input_data)
            input_classes = input_classes.to(outputs.device)
            val_batch_loss = self.loss(input_classes, outputs)
            outputs = outputs.logits if isinstance(outputs, InceptionOutputs) else outputs
            outputs = torch.softmax(outputs, dim=1)
        return dict(val_batch_loss=val_batch_loss, val_outputs=outputs, val_input_classes=input_classes,
                    val_input_srcs=input_src)

    def validation_epoch_end(self, steps):
        from sklearn import metrics
        if self.current_epoch == 0:
            self.best_val_loss = np.inf
        validation_loss = torch.stack([batch['val_batch_loss'] for batch in steps]).sum()
        if validation_loss.item() < self.best_val_loss:
            self.best_val_loss = validation_loss.item()
            self.best_epoch = self.current_epoch
        outputs = torch.cat([batch['val_outputs'] for batch in steps], dim=0).detach().cpu().numpy()
        output_classes = np.argmax(outputs, axis=1)
        input_classes = torch.cat([batch['val_input_classes'] for batch in steps], dim=0).detach().cpu().numpy()
        input_srcs = [item for sublist in [batch['val_input_srcs'] for batch in steps] for item in sublist]
        f1_weighted = metrics.f1_score(input_classes, output_classes, average='weighted')
        f1_macro = metrics.f1_score(input_classes, output_classes, average='macro')
        eoe = 'Best Epoch: {}, train_loss: {:.3f}, val_loss: {:.3f}, val_f1_w={:02.1f}%, val_f1_m={:02.1f}%'
        eoe = eoe.format(True if self.current_epoch == self.best_epoch else self.best_epoch + 1, self.agg_train_loss,
                         validation_loss, 100 * f1_weighted, 100 * f1_macro)
        print(eoe, flush=True, end='\n\n')
        self.logged = dict(epoch=self.current_epoch, best=self.best_epoch == self.current_epoch,
                           train_loss=self.agg_train_loss, val_loss=validation_loss.item(),
                           input_classes=input_classes, output_classes=output_classes, input_srcs=input_srcs,
                           outputs=outputs, f1_macro=f1_macro, f1_weighted=f1_weighted)
        self.agg_train_loss = 0.0
        return dict(hiddens=dict(outputs=outputs))

    # RUNNING the model #
    def test_step(self, batch, batch_idx, dataloader_idx=None):
        input_data, input_srcs = batch
        with torch.no_grad():
            outputs = self.forward(input_data)
            outputs = outputs.logits if isinstance(outputs, InceptionOutputs) else outputs
            outputs = torch.softmax(outputs, dim=1)
        return dict(test_outputs=outputs, test_srcs=input_srcs)

    def test_epoch_end(self, steps, input_obj=None):
        outputs = torch.cat([batch['test_outputs'] for batch in steps], dim=0).detach().cpu().numpy()
        images = [item for batch in steps for item in batch['test_srcs']]
        rr = self.RunResults(inputs=images, outputs=outputs, input_obj=input_obj)
        self.logged = dict(RunResults=[rr])
        return rr

    class RunResults:
        def __init__(self, inputs, outputs, input_obj, bin_pid=False):
            self.inputs = inputs
            self.outputs = outputs
            self.input_obj = input_obj
            self.type = 'Bin' if bin_pid else 'ImgDir'

        def __repr__(self):
            rep = '{}: {} ({} imgs)'.format(self.type, self.input_obj, len(self.inputs))
            return repr(rep)

    # checkpoints: the [PL] dict layout the reference reads back (neuston_net.py:173,443)
    def optimizer_state(self):
        """``torch.optim`` state_dict of the engine's fused optimizer in parameters() order ([PL] ``optimizer_states[0]``)"""
        eng = self.model.engine
        state, ids = {}, []
        for i, (key, (o, n, shape, kind, node)) in enumerate(eng.poff.items()):
            ids.append(i)
            if eng.step_count == 0:
                continue

            def view(buf):
                t = buf[o:o + n].detach().cpu()
                return (t.view(shape[0], shape[2], shape[3], shape[1]).permute(0, 3, 1, 2) if kind == 'conv' else t.view(shape)).contiguous()
            if eng.optimizer == 'sgd':
                if eng.momentum:
                    state[i] = dict(momentum_buffer=view(eng.M))
            else:
                state[i] = dict(step=eng.step_count, exp_avg=view(eng.M), exp_avg_sq=view(eng.V))
        if eng.optimizer == 'sgd':
            group = dict(lr=eng.lr, momentum=eng.momentum, dampening=0, weight_decay=0, nesterov=False, params=ids)
        else:
            group = dict(lr=eng.lr, betas=tuple(eng.betas), eps=eng.eps, weight_decay=0, amsgrad=False, params=ids)
        return dict(state=state, param_groups=[group])

    def checkpoint_dict(self, epoch=0, global_step=0):
        """the dict [PL] 1.3.8 ``trainer.save_checkpoint`` writes and ``LightningModule.load_from_checkpoint`` reads back
        (reference: neuston_net.py:98-100,118-120,173,443): ``hparams_name`` / ``hparams_type`` tell Lightning to rebuild the
        module as ``NeustonModel(hparams=Namespace(**hyper_parameters))``; ``callbacks`` is empty (upstream keys it by callback
        class objects, which would force every reader to import Lightning)."""
        hp = dict(vars(self.hparams))
        out = dict(epoch=epoch, global_step=global_step, state_dict={k: v.detach().cpu().contiguous() for k, v in self.state_dict().items()},
                   hyper_parameters=hp, hparams_name='hparams', hparams_type='Namespace', optimizer_states=[self.optimizer_state()],
                   lr_schedulers=[], callbacks={})
        out['pytorch-lightning_version'] = '1.3.8'
        return out

    @classmethod
    def load_from_checkpoint(cls, path, device=0, max_batch=None, inference=False):
        """never triggers the pretrained download/--weights requirement: the state_dict in the file is the model.
        inference=True (neuston_net RUN): activations for max_batch images, gradient-side buffers for one"""
        ckpt = load_checkpoint_file(path)
        hp = dict(ckpt['hyper_parameters'])
        obj = cls(hp, device=device, max_batch=max_batch, train_batch=1 if inference else None)
        obj.load_state_dict(ckpt['state_dict'])
        st = (ckpt.get('optimizer_states') or [None])[0]
        if st and st.get('state'):
            obj.load_optimizer_state(st)
        return obj

    def load_optimizer_state(self, st):
        eng = self.model.engine
        for i, (key, (o, n, shape, kind, node)) in enumerate(eng.poff.items()):
            ps = st['state'].get(i)
            if not ps:
                continue

            def put(buf, t):
                t = torch.as_tensor(t, dtype=torch.float32)
                buf[o:o + n].copy_((t.permute(0, 2, 3, 1) if kind == 'conv' else t).reshape(-1))
            if 'exp_avg' in ps:
                put(eng.M, ps['exp_avg']); put(eng.V, ps['exp_avg_sq'])
                eng.step_count = max(eng.step_count, int(ps.get('step', 0)))
            elif 'momentum_buffer' in ps:
                put(eng.M, ps['momentum_buffer'])
