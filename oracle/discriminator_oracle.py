"""CPU ORACLE for the PatchGAN discriminator and the adversarial generator step — test infrastructure only (tests/, smoke(), bench's
cpu_baseline leg may import it; nothing under fetal-mri-segmentation_amd/ does).

Restates on torch-CPU (fp64 in the tests, autograd for every gradient), with Keras 2.2 / TensorFlow semantics:
  reference fetal_net/model/discriminator/all_dis_3d.py
    :59-64  mini_conv_block = Conv3D(n, 3, 'same', strides) -> keras-contrib InstanceNormalization(axis=1) -> LeakyReLU(0.3)
    :67-72  conv_block      = mini block (strides) -> SpatialDropout3D(rate) -> mini block -> AveragePooling3D()      n = min(128, 2^level * base)
    :31-38  level 0 with strides (2, 2, 1), levels 1 .. depth-1 with stride 1; stop after a level whose output has shape[-2] < 3, every
            level not built becomes one Dense(128, LeakyReLU) after the pooling
    :40-44  GlobalAveragePooling3D -> Dense(128, LeakyReLU) x fc_layers -> Dense(1, 'sigmoid')
    :47-54  loss = binary_crossentropy over the flattened batch (Keras: probabilities clipped to [1e-7, 1 - 1e-7]), metric mae,
            Adam(lr, beta_1 = 0.5)
  reference fetal/experiments/train_adv.py
    :173-180  combined model: total = gd_loss_ratio * BCE(D(Concatenate(axis=1)([G(x), x])), valid) + seg_loss(G(x), segs); D frozen
    :92-115   discriminator batches: mul_merge_maps / soft labels (host side, restated in fetal_net/adversarial.py itself)
Stride-2 'same' uses TensorFlow's asymmetric padding (even extent: 0 before / 1 after); AveragePooling3D is 'valid' (floor).

PARITY UNPINNED for the arithmetic: Keras / TensorFlow / keras-contrib are not installable here, so no reference output exists to pin
against.  The layer graph (names, shapes, parameter counts, the early stop and fc_layers rule) IS pinned: tests/golden/topology_golden.json
holds the discriminator cases recorded from the reference builder running over a recording Keras stub (tests/golden/make_fixtures.py).
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from .unet_oracle import IN_EPS, LEAKY_ALPHA, glorot_uniform


class DiscriminatorSpec:
    def __init__(self, input_shape, n_base_filters=16, depth=5, dropout_rate=0.3):
        self.input_shape = tuple(int(v) for v in input_shape)          # (C, X, Y, Z)
        self.base, self.depth, self.dropout_rate = n_base_filters, depth, dropout_rate
        self.blocks, self.fc_layers = [], 0
        cin, sp, ck = self.input_shape[0], list(self.input_shape[1:]), 0
        for level in range(depth):
            n = min(128, (2 ** level) * n_base_filters)
            strides = (2, 2, 1) if level == 0 else (1, 1, 1)
            ck += 2
            self.blocks.append(dict(level=level, a="conv3d_%d" % (ck - 1), na="instance_normalization_%d" % (ck - 1), b="conv3d_%d" % ck,
                                    nb="instance_normalization_%d" % ck, cin=cin, cout=n, strides=strides))
            sp = [-(-d // s) for d, s in zip(sp, strides)]
            sp = [d // 2 for d in sp]
            cin = n
            if level >= 1 and sp[-2] < 3:
                self.fc_layers = depth - level - 1
                break
        self.gap_channels, self.final_spatial = cin, tuple(sp)
        self.dense = ["dense_%d" % (i + 1) for i in range(self.fc_layers + 1)]

    def init_weights(self, seed=42):
        rs = np.random.RandomState(seed)
        W = OrderedDict()
        for b in self.blocks:
            for conv, norm, cin in ((b["a"], b["na"], b["cin"]), (b["b"], b["nb"], b["cout"])):
                W[conv + "/kernel"] = glorot_uniform(rs, (3, 3, 3, cin, b["cout"]))
                W[conv + "/bias"] = np.zeros(b["cout"], np.float32)
                W[norm + "/gamma"] = np.ones(b["cout"], np.float32)
                W[norm + "/beta"] = np.zeros(b["cout"], np.float32)
        k = self.gap_channels
        for i, name in enumerate(self.dense):
            m = 128 if i < self.fc_layers else 1
            lim = math.sqrt(6.0 / (k + m))
            W[name + "/kernel"] = rs.uniform(-lim, lim, size=(k, m)).astype(np.float32)
            W[name + "/bias"] = np.zeros(m, np.float32)
            k = m
        return W


def conv_same(x, k, b, strides):
    """Keras kernel (3,3,3,Cin,Cout), TensorFlow 'same' padding, per-axis strides"""
    pads = []
    for n, s in zip(reversed(x.shape[2:]), reversed(strides)):
        out = -(-n // s)
        tot = max((out - 1) * s + k.shape[0] - n, 0)
        pads += [tot // 2, tot - tot // 2]
    return F.conv3d(F.pad(x, pads), k.permute(4, 3, 0, 1, 2), b, stride=tuple(strides))


def inorm_leaky(x, gamma, beta):
    mean = x.mean(dim=(2, 3, 4), keepdim=True)
    std = x.std(dim=(2, 3, 4), unbiased=False, keepdim=True) + IN_EPS
    return F.leaky_relu((x - mean) / std * gamma.view(1, -1, 1, 1, 1) + beta.view(1, -1, 1, 1, 1), LEAKY_ALPHA)


def forward(spec, Wt, x, dropout_masks=None):
    """x (N, C, X, Y, Z) -> (logits (N, 1), probabilities (N, 1)).  dropout_masks: {level: (N, C) tensor of 0 | 1 / (1 - p)} = training mode"""
    h = x
    for b in spec.blocks:
        h = inorm_leaky(conv_same(h, Wt[b["a"] + "/kernel"], Wt[b["a"] + "/bias"], b["strides"]), Wt[b["na"] + "/gamma"], Wt[b["na"] + "/beta"])
        if dropout_masks is not None:
            h = h * dropout_masks[b["level"]].view(h.shape[0], h.shape[1], 1, 1, 1)
        h = inorm_leaky(conv_same(h, Wt[b["b"] + "/kernel"], Wt[b["b"] + "/bias"], (1, 1, 1)), Wt[b["nb"] + "/gamma"], Wt[b["nb"] + "/beta"])
        h = F.avg_pool3d(h, 2)
    h = h.mean(dim=(2, 3, 4))
    for i, name in enumerate(spec.dense):
        h = h @ Wt[name + "/kernel"] + Wt[name + "/bias"]
        if i < spec.fc_layers:
            h = F.leaky_relu(h, LEAKY_ALPHA)
    return h, torch.sigmoid(h)


def keras_bce(p, t, eps=1e-7):
    """keras.losses.binary_crossentropy on probabilities, mean over everything (the last axis has one element)"""
    pc = torch.clamp(p, eps, 1 - eps)
    return -(t * torch.log(pc) + (1 - t) * torch.log(1 - pc)).mean()


def discriminator_step(spec, W, x, target, dropout_masks=None, dtype=torch.float64):
    """one discriminator training batch: (loss, mae, probabilities, {weight name: gradient})"""
    Wt = OrderedDict((k, torch.tensor(np.asarray(v), dtype=dtype, requires_grad=True)) for k, v in W.items())
    _, p = forward(spec, Wt, torch.as_tensor(x, dtype=dtype), dropout_masks)
    t = torch.as_tensor(target, dtype=dtype).reshape(p.shape)
    loss = keras_bce(p, t)
    loss.backward()
    return float(loss.detach()), float((p - t).abs().mean().detach()), p.detach(), OrderedDict((k, v.grad) for k, v in Wt.items())


def adversarial_term(spec, W, probs, x, valid, dropout_masks=None, dtype=torch.float64):
    """BCE(D(concat([probs, x])), valid) as a differentiable function of the generator's probabilities (N, L, X, Y, Z): returns
    (loss value, d loss / d probs) - what the frozen discriminator hands back to the generator (train_adv.py:175-180)"""
    Wt = OrderedDict((k, torch.tensor(np.asarray(v), dtype=dtype)) for k, v in W.items())
    pr = torch.as_tensor(probs, dtype=dtype).clone().requires_grad_(True)
    _, p = forward(spec, Wt, torch.cat([pr, torch.as_tensor(x, dtype=dtype)], dim=1), dropout_masks)
    loss = keras_bce(p, torch.as_tensor(valid, dtype=dtype).reshape(p.shape))
    loss.backward()
    return float(loss.detach()), pr.grad


def combined_loss_and_grads(gen_forward, gen_W, spec, dis_W, x, segs, valid, gd_loss_ratio, dropout_masks=None, x_semi=None,
                            dtype=torch.float64):
    """The generator step through the frozen discriminator.  gen_forward(Wt, x) -> probabilities (N, L, X, Y, Z), differentiable.
    train_adv.py:173-180:   total = ratio * BCE(D(cat[G(x), x]), valid) + (-Dice(segs, G(x)))
    train_semi.py:174-184:  total = (-Dice(segs, G(x))) + ratio * BCE(D(cat[G(x_semi), x_semi]), valid)         (x_semi given)
    -> dict(total, seg_loss, dis_loss, grads{generator weight: ndarray})"""
    Wg = OrderedDict((k, torch.tensor(np.asarray(v), dtype=dtype, requires_grad=True)) for k, v in gen_W.items())
    Wd = OrderedDict((k, torch.tensor(np.asarray(v), dtype=dtype)) for k, v in dis_W.items())
    xt = torch.as_tensor(np.asarray(x), dtype=dtype)
    yt = torch.as_tensor(np.asarray(segs), dtype=dtype)
    probs = gen_forward(Wg, xt)
    dice = (2.0 * (yt * probs).sum() + 1.0) / (yt.sum() + probs.sum() + 1.0)
    seg_loss = -dice
    if x_semi is not None:
        xa = torch.as_tensor(np.asarray(x_semi), dtype=dtype)
        pa = gen_forward(Wg, xa)
    else:
        xa, pa = xt, probs
    _, p = forward(spec, Wd, torch.cat([pa, xa], dim=1), dropout_masks)
    dis_loss = keras_bce(p, torch.as_tensor(np.asarray(valid), dtype=dtype).reshape(p.shape))
    total = seg_loss + gd_loss_ratio * dis_loss
    total.backward()
    return dict(total=float(total.detach()), seg_loss=float(seg_loss.detach()), dis_loss=float(dis_loss.detach()),
                grads=OrderedDict((k, v.grad.detach().numpy().copy()) for k, v in Wg.items()))
