"""CPU ORACLE for the Isensee-2017 models (3-D, and the 2-D twin with ndim=2) — test infrastructure only.

Restates reference fetal_net/model/unet3d/isensee2017.py:15-111 on torch-CPU with Keras/TF semantics:
  conv block      = Conv3D(k, 'same', strides) -> keras-contrib InstanceNormalization(axis=1) -> LeakyReLU(0.3)   (:12)
  context module  = conv block -> SpatialDropout3D(rate) -> conv block                                             (:107-111)
  level           = in_conv (stride 1 at level 0, stride 2 below) ; out = in_conv + context(in_conv)              (:44-57)
  up-sampling     = UpSampling3D(2) -> conv block                                                                  (:101-104)
  localisation    = conv block 3x3x3 -> conv block 1x1x1 on concatenate([skip, up])  (skip FIRST)                  (:62,95-98)
  heads           = Conv3D(n_labels,1x1x1) at the n_segmentation_levels shallowest levels, summed bottom-up through
                    UpSampling3D (:64-77), then Activation(sigmoid)
Stride-2 'same' uses TensorFlow's asymmetric padding (even extent: 0 before / 1 after).  PARITY UNPINNED for the arithmetic
(see oracle/unet_oracle.py header); the layer graph is pinned by tests/golden/topology_golden.json (isensee3d_* cases).

ndim=2 restates reference fetal_net/model/unet/isensee.py:14-105: channels-LAST input (X,Y,C) wrapped in Permute layers, Conv2D /
UpSampling2D / SpatialDropout2D, and the heads are summed only with summation=True (:62-75) - by default the output is the level-0
head alone; the deeper heads are created (Keras' conv2d_N counter advances) but belong to no model: they get no weights here.
Layer graph pinned by the isensee2d_* cases of the same fixture.
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from .unet_oracle import IN_EPS, LEAKY_ALPHA, glorot_uniform


class IsenseeSpec:
    def __init__(self, input_shape=(1, 128, 128, 128), n_base_filters=16, depth=5, dropout_rate=0.3, n_segmentation_levels=1,
                 n_labels=1, ndim=3, summation=None):
        self.input_shape = tuple(input_shape)
        self.base, self.depth, self.dropout_rate = n_base_filters, depth, dropout_rate
        self.nseg, self.n_labels, self.ndim = n_segmentation_levels, n_labels, ndim
        self.summation = (ndim == 3) if summation is None else bool(summation)      # the 3-D builder always sums its heads
        ck = nk = 0
        self.blocks = []          # every conv block in Keras creation order: dict(name, norm, cin, cout, k, s)
        cname = "conv%dd_%%d" % ndim

        def block(cin, cout, k=3, s=1):
            nonlocal ck, nk
            ck += 1
            nk += 1
            b = dict(name=cname % ck, norm="instance_normalization_%d" % nk, cin=cin, cout=cout, k=k, s=s)
            self.blocks.append(b)
            return b

        self.levels = []
        cin = input_shape[0] if ndim == 3 else input_shape[-1]
        for lv in range(depth):
            n = (2 ** lv) * n_base_filters
            inc = block(cin, n, 3, 1 if lv == 0 else 2)
            c1 = block(n, n)
            c2 = block(n, n)
            self.levels.append(dict(in_conv=inc, ctx=(c1, c2), filters=n))
            cin = n
        self.decoder = []
        self.heads = {}
        for lv in range(depth - 2, -1, -1):
            n = self.levels[lv]["filters"]
            up = block(cin, n)
            loc1 = block(2 * n, n)
            loc2 = block(n, n, 1, 1)
            self.decoder.append(dict(level=lv, up=up, loc=(loc1, loc2)))
            cin = n
            if lv < n_segmentation_levels:
                ck += 1
                if self.summation or lv == 0:                  # a head that does not reach the output owns no weights (2-D default)
                    self.heads[lv] = dict(name=cname % ck, cin=n, cout=n_labels)

    def init_weights(self, seed=42):
        rs = np.random.RandomState(seed)
        W = OrderedDict()
        # Keras creation order: blocks interleaved with heads as the builder creates them
        order = []
        for lv in self.levels:
            order += [lv["in_conv"], lv["ctx"][0], lv["ctx"][1]]
        for d in self.decoder:
            order += [d["up"], d["loc"][0], d["loc"][1]]
            if d["level"] in self.heads:
                order.append(self.heads[d["level"]])
        for b in order:
            k = b.get("k", 1)
            W[b["name"] + "/kernel"] = glorot_uniform(rs, (k,) * self.ndim + (b["cin"], b["cout"]))
            W[b["name"] + "/bias"] = np.zeros(b["cout"], np.float32)
            if "norm" in b:
                W[b["norm"] + "/gamma"] = np.ones(b["cout"], np.float32)
                W[b["norm"] + "/beta"] = np.zeros(b["cout"], np.float32)
        return W


def _conv_same(x, k, b, stride):
    """Keras kernel (k..., Cin, Cout), TensorFlow 'same' padding; 2-D or 3-D by the rank of x"""
    ks = k.shape[0]
    nd = x.dim() - 2
    pads = []
    for n in reversed(x.shape[2:]):
        out = -(-n // stride)
        tot = max((out - 1) * stride + ks - n, 0)
        pads += [tot // 2, tot - tot // 2]
    if nd == 2:
        return F.conv2d(F.pad(x, pads), k.permute(3, 2, 0, 1), b, stride=stride)
    return F.conv3d(F.pad(x, pads), k.permute(4, 3, 0, 1, 2), b, stride=stride)


def _inorm_leaky(x, gamma, beta):
    ax = tuple(range(2, x.dim()))
    mean = x.mean(dim=ax, keepdim=True)
    std = x.std(dim=ax, unbiased=False, keepdim=True) + IN_EPS
    shp = (1, -1) + (1,) * (x.dim() - 2)
    return F.leaky_relu((x - mean) / std * gamma.view(shp) + beta.view(shp), LEAKY_ALPHA)


def _up(x):
    for ax in range(2, x.dim()):
        x = torch.repeat_interleave(x, 2, dim=ax)
    return x


def forward(spec, Wt, x, dropout_masks=None):
    """x (N,C,X,Y,Z) - ndim=2: (N,X,Y,C) channels-last, logits / probabilities come back as (N,X,Y,labels).
    dropout_masks: {level: (N,C) tensor of 0 | 1/(1-p)} for training mode, None = inference (identity)."""
    if spec.ndim == 2:
        x = x.permute(0, 3, 1, 2)                              # Permute((3,1,2)), reference unet/isensee.py:38

    def block(h, b):
        h = _conv_same(h, Wt[b["name"] + "/kernel"], Wt[b["name"] + "/bias"], b["s"])
        return _inorm_leaky(h, Wt[b["norm"] + "/gamma"], Wt[b["norm"] + "/beta"])

    outs = []
    h = x
    for lv, L in enumerate(spec.levels):
        inc = block(h, L["in_conv"])
        c = block(inc, L["ctx"][0])
        if dropout_masks is not None:
            c = c * dropout_masks[lv].view((c.shape[0], c.shape[1]) + (1,) * (c.dim() - 2))
        c = block(c, L["ctx"][1])
        h = inc + c
        outs.append(h)
    segs = {}
    for d in spec.decoder:
        up = block(_up(h), d["up"])
        cat = torch.cat([outs[d["level"]], up], dim=1)          # skip first (reference isensee2017.py:62)
        h = block(block(cat, d["loc"][0]), d["loc"][1])
        if d["level"] in spec.heads:
            hd = spec.heads[d["level"]]
            segs[d["level"]] = _conv_same(h, Wt[hd["name"] + "/kernel"], Wt[hd["name"] + "/bias"], 1)
    if spec.summation:
        out = None
        for lv in reversed(range(spec.nseg)):
            out = segs[lv] if out is None else out + segs[lv]
            if lv > 0:
                out = _up(out)
    else:
        out = segs[0]
    if spec.ndim == 2:
        out = out.permute(0, 2, 3, 1)                          # Permute((2,3,1)) behind the activation, reference unet/isensee.py:78
    return out, torch.sigmoid(out)


def loss_and_grads(spec, W, x, y, dropout_masks=None, dtype=torch.float64):
    from .unet_oracle import dice_coefficient_t, to_torch
    Wt = to_torch(W, dtype, requires_grad=True)
    masks = None if dropout_masks is None else {k: torch.tensor(v, dtype=dtype) for k, v in dropout_masks.items()}
    logits, probs = forward(spec, Wt, torch.tensor(np.asarray(x), dtype=dtype), masks)
    dice = dice_coefficient_t(torch.tensor(np.asarray(y), dtype=dtype), probs)
    (-dice).backward()
    return dict(loss=-float(dice.detach()), dice=float(dice.detach()), logits=logits.detach().numpy(), probs=probs.detach().numpy(),
                grads=OrderedDict((k, v.grad.numpy().copy()) for k, v in Wt.items()))
