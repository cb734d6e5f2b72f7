"""CPU ORACLE — test infrastructure only (never imported by the product path).

Restates, with Keras-2.2/TF-1.x semantics, the arithmetic of the reference hot path:

  * unet_model_3d / create_convolution_block / get_up_convolution     reference fetal_net/model/unet3d/unet.py:17-138
  * unet_model_2d                                                      reference fetal_net/model/unet/unet.py:22-141
  * dice / vod / weighted-dice losses                                  reference fetal_net/metrics.py:7-100
  * Keras Adam (model.compile(optimizer=Adam(lr=...)))                 reference fetal_net/model/unet3d/unet.py:85

PARITY STATUS: the conv/pool/norm/optimizer arithmetic lives in un-vendored third-party code (Keras>=2,
TensorFlow 1.x, keras-contrib — reference requirements.txt:7, README.md:18) that can be neither imported nor built
here, and the reference's own tests hold no numeric vector for it => for those ops this oracle is
**parity unpinned**: it follows the published Keras/TF algorithms (from memory, each behind a named switch) and is
cross-checked against an independent slow numpy direct convolution (oracle/numpy_ref.py).  The metrics and the
builder topology ARE pinned against fixtures generated from the reference itself (tests/golden/make_fixtures.py).

Layout here is the reference's: channels-first (N,C,X,Y,Z), fp32 or fp64, Keras kernel tensors (kD,kH,kW,Cin,Cout).
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

# ---- named semantic switches (all [3p, from memory]) -----------------------------------
KERAS_EPSILON = 1e-7            # K.epsilon(); Adam epsilon default in Keras 2.2 (None -> K.epsilon())
ADAM_BETA1, ADAM_BETA2 = 0.9, 0.999
BN_MOMENTUM, BN_EPS = 0.99, 1e-3
IN_EPS = 1e-3                   # keras-contrib InstanceNormalization: (x-mean)/(std+eps)  (eps added to std)
LEAKY_ALPHA = 0.3               # keras.layers.LeakyReLU default


def glorot_uniform(rs, shape):
    """Keras glorot_uniform for a conv kernel (k..., Cin, Cout): U(+-sqrt(6/(fan_in+fan_out))), fan = prod(k)*C."""
    rf = int(np.prod(shape[:-2]))
    fan_in, fan_out = rf * shape[-2], rf * shape[-1]
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return rs.uniform(-lim, lim, size=shape).astype(np.float32)


class Spec:
    """Layer list of unet_model_3d / unet_model_2d in Keras creation order with Keras auto-names."""

    def __init__(self, input_shape, ndim=3, pool_size=2, n_labels=1, deconvolution=False, depth=4, n_base_filters=32,
                 batch_normalization=False, instance_normalization=False, activation_name="sigmoid", dropout_rate=0.0):
        self.ndim = ndim
        self.input_shape = tuple(input_shape)
        self.depth, self.n_base_filters, self.n_labels = depth, n_base_filters, n_labels
        self.deconvolution, self.batch_normalization = deconvolution, batch_normalization
        self.instance_normalization = instance_normalization
        self.activation_name = activation_name
        self.pool = pool_size
        self.in_channels = input_shape[0] if ndim == 3 else input_shape[-1]
        cnt = {}

        def nm(base):
            cnt[base] = cnt.get(base, 0) + 1
            return "%s_%d" % (base, cnt[base])

        d = "3d" if ndim == 3 else "2d"
        self.layers = []  # (kind, name, params)
        nm("input")
        if ndim == 2:
            nm("permute")
        cin = self.in_channels
        self.enc = []
        for ld in range(depth):
            blocks = []
            for mult in (1, 2):
                cout = n_base_filters * (2 ** ld) * mult
                c = nm("conv" + d)
                bn = nm("batch_normalization") if batch_normalization else None
                inn = nm("instance_normalization") if (instance_normalization and not batch_normalization) else None
                nm("activation")
                blocks.append(dict(name=c, cin=cin, cout=cout, bn=bn, inn=inn))
                cin = cout
                if ndim == 2 and mult == 1 and dropout_rate > 0:
                    nm("spatial_dropout2d")
            if ld < depth - 1:
                nm("max_pooling" + d)
            self.enc.append(blocks)
        self.dec = []
        for ld in range(depth - 2, -1, -1):
            up = None
            if deconvolution:
                up = dict(name=nm("conv%s_transpose" % d), cin=cin, cout=cin)
            else:
                nm("up_sampling" + d)
            nm("concatenate")
            skip_c = self.enc[ld][1]["cout"]
            blocks = []
            ccat = cin + skip_c
            for i in range(2):
                c = nm("conv" + d)
                bn = nm("batch_normalization") if batch_normalization else None
                inn = nm("instance_normalization") if (instance_normalization and not batch_normalization) else None
                nm("activation")
                blocks.append(dict(name=c, cin=ccat if i == 0 else skip_c, cout=skip_c, bn=bn, inn=inn))
                if ndim == 2 and i == 0 and dropout_rate > 0:
                    nm("spatial_dropout2d")
            cin = skip_c
            self.dec.append(dict(level=ld, up=up, blocks=blocks))
        self.final = dict(name=nm("conv" + d), cin=cin, cout=n_labels)

    def conv_blocks(self):
        for lv in self.enc:
            for b in lv:
                yield b
        for dlv in self.dec:
            for b in dlv["blocks"]:
                yield b

    def init_weights(self, seed=42):
        """Keras defaults: glorot_uniform kernels, zero bias, BN/IN gamma=1 beta=0.  Keras kernel layout."""
        rs = np.random.RandomState(seed)
        k = (3,) * self.ndim
        W = OrderedDict()
        order = []
        for lv in self.enc:
            order += lv
        for dlv in self.dec:
            if dlv["up"] is not None:
                order.append(dict(dlv["up"], transpose=True))
            order += dlv["blocks"]
        for b in order:
            if b.get("transpose"):
                # Keras Conv3DTranspose kernel: (k,k,k, Cout, Cin)
                W[b["name"] + "/kernel"] = glorot_uniform(rs, (2,) * self.ndim + (b["cout"], b["cin"]))
                W[b["name"] + "/bias"] = np.zeros(b["cout"], np.float32)
                continue
            W[b["name"] + "/kernel"] = glorot_uniform(rs, k + (b["cin"], b["cout"]))
            W[b["name"] + "/bias"] = np.zeros(b["cout"], np.float32)
            for nk in ("bn", "inn"):
                if b.get(nk):
                    W[b[nk] + "/gamma"] = np.ones(b["cout"], np.float32)
                    W[b[nk] + "/beta"] = np.zeros(b["cout"], np.float32)
        f = self.final
        W[f["name"] + "/kernel"] = glorot_uniform(rs, (1,) * self.ndim + (f["cin"], f["cout"]))
        W[f["name"] + "/bias"] = np.zeros(f["cout"], np.float32)
        return W

    def n_params(self, W=None):
        W = W or self.init_weights()
        return int(sum(v.size for v in W.values()))


# ------------------------------------------------------------------------------------------------------------------
# forward / backward (torch-CPU autograd does the differentiation; ops chosen to mirror Keras/TF semantics)
# ------------------------------------------------------------------------------------------------------------------
def _conv(x, k, b, ndim):
    # Keras kernel (k..., Cin, Cout) -> torch (Cout, Cin, k...); 'same' at stride 1 with k=3 is symmetric pad 1
    if ndim == 3:
        w = k.permute(4, 3, 0, 1, 2)
        pad = (k.shape[0] - 1) // 2
        return F.conv3d(x, w, b, padding=pad)
    w = k.permute(3, 2, 0, 1)
    pad = (k.shape[0] - 1) // 2
    return F.conv2d(x, w, b, padding=pad)


def _deconv(x, k, b, ndim):
    # Keras Conv3DTranspose kernel (k,k,k,Cout,Cin), stride 2, 'valid': torch weight is (Cin, Cout, k,k,k)
    if ndim == 3:
        return F.conv_transpose3d(x, k.permute(4, 3, 0, 1, 2), b, stride=2)
    return F.conv_transpose2d(x, k.permute(3, 2, 0, 1), b, stride=2)


def _upsample(x, ndim):
    for ax in range(2, 2 + ndim):
        x = torch.repeat_interleave(x, 2, dim=ax)
    return x


def _batchnorm_train(x, gamma, beta):
    ax = [0] + list(range(2, x.dim()))
    mean = x.mean(dim=ax, keepdim=True)
    var = x.var(dim=ax, unbiased=False, keepdim=True)
    shp = [1, -1] + [1] * (x.dim() - 2)
    return (x - mean) / torch.sqrt(var + BN_EPS) * gamma.view(shp) + beta.view(shp)


def _instancenorm(x, gamma, beta):
    ax = list(range(2, x.dim()))
    mean = x.mean(dim=ax, keepdim=True)
    std = x.std(dim=ax, unbiased=False, keepdim=True) + IN_EPS
    shp = [1, -1] + [1] * (x.dim() - 2)
    return (x - mean) / std * gamma.view(shp) + beta.view(shp)


def forward(spec, Wt, x, return_intermediates=False):
    """x: torch (N,C,X,Y,Z) [3D] or (N,X,Y,C) [2D, channels-last like the reference].  Returns (logits, probs)."""
    nd = spec.ndim
    inter = OrderedDict()
    if nd == 2:
        x = x.permute(0, 3, 1, 2)

    def block(h, b):
        h = _conv(h, Wt[b["name"] + "/kernel"], Wt[b["name"] + "/bias"], nd)
        if return_intermediates:
            inter[b["name"] + "/pre"] = h
            if h.requires_grad:
                h.retain_grad()
        if b.get("bn"):
            h = _batchnorm_train(h, Wt[b["bn"] + "/gamma"], Wt[b["bn"] + "/beta"])
        elif b.get("inn"):
            h = _instancenorm(h, Wt[b["inn"] + "/gamma"], Wt[b["inn"] + "/beta"])
        if return_intermediates:
            inter[b["name"] + "/z"] = h.detach()           # pre-activation (tests use it to avoid ReLU-boundary ties)
        h = F.relu(h)
        inter[b["name"]] = h
        if return_intermediates and h.requires_grad:
            h.retain_grad()
        return h

    h = x
    skips = []
    for ld, lv in enumerate(spec.enc):
        h = block(h, lv[0])
        h = block(h, lv[1])
        skips.append(h)
        if ld < spec.depth - 1:
            h = F.max_pool3d(h, 2) if nd == 3 else F.max_pool2d(h, 2)
    for dlv in spec.dec:
        if dlv["up"] is not None:
            u = dlv["up"]
            h = _deconv(h, Wt[u["name"] + "/kernel"], Wt[u["name"] + "/bias"], nd)
        else:
            h = _upsample(h, nd)
        h = torch.cat([h, skips[dlv["level"]]], dim=1)   # up first (reference unet.py:61)
        for b in dlv["blocks"]:
            h = block(h, b)
    f = spec.final
    logits = _conv(h, Wt[f["name"] + "/kernel"], Wt[f["name"] + "/bias"], nd)
    probs = torch.sigmoid(logits) if spec.activation_name == "sigmoid" else logits
    if nd == 2:
        logits, probs = logits.permute(0, 2, 3, 1), probs.permute(0, 2, 3, 1)
    if return_intermediates:
        return logits, probs, inter
    return logits, probs


def dice_coefficient_t(y, p, smooth=1.0):
    """reference metrics.py:11-15 (whole-batch flatten)."""
    yf, pf = y.reshape(-1), p.reshape(-1)
    inter = (yf * pf).sum()
    return (2.0 * inter + smooth) / (yf.sum() + pf.sum() + smooth)


def to_torch(W, dtype=torch.float32, requires_grad=False):
    return OrderedDict((k, torch.tensor(np.asarray(v), dtype=dtype, requires_grad=requires_grad)) for k, v in W.items())


def loss_and_grads(spec, W, x, y, dtype=torch.float64):
    """-> dict(loss, dice, logits, probs, grads{name: ndarray in Keras layout})"""
    Wt = to_torch(W, dtype, requires_grad=True)
    xt = torch.tensor(np.asarray(x), dtype=dtype)
    yt = torch.tensor(np.asarray(y), dtype=dtype)
    logits, probs = forward(spec, Wt, xt)
    dice = dice_coefficient_t(yt, probs)
    loss = -dice                                             # reference metrics.py:31-32
    loss.backward()
    grads = OrderedDict((k, v.grad.detach().numpy().copy()) for k, v in Wt.items())
    return dict(loss=float(loss.detach()), dice=float(dice.detach()), logits=logits.detach().numpy(), probs=probs.detach().numpy(),
                grads=grads)


class KerasAdam:
    """Keras 2.2 Adam.get_updates: lr_t = lr*sqrt(1-b2^t)/(1-b1^t); p -= lr_t*m/(sqrt(v)+eps)."""

    def __init__(self, W, lr, beta1=ADAM_BETA1, beta2=ADAM_BETA2, eps=KERAS_EPSILON, dtype=np.float64):
        self.lr, self.b1, self.b2, self.eps = lr, beta1, beta2, eps
        self.t = 0
        self.m = OrderedDict((k, np.zeros_like(v, dtype=dtype)) for k, v in W.items())
        self.v = OrderedDict((k, np.zeros_like(v, dtype=dtype)) for k, v in W.items())

    def step(self, W, grads):
        self.t += 1
        lr_t = self.lr * math.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t)
        for k in W:
            g = grads[k].astype(self.m[k].dtype)
            self.m[k] = self.b1 * self.m[k] + (1 - self.b1) * g
            self.v[k] = self.b2 * self.v[k] + (1 - self.b2) * g * g
            W[k] = (W[k].astype(self.m[k].dtype) - lr_t * self.m[k] / (np.sqrt(self.v[k]) + self.eps)).astype(W[k].dtype)
        return W


def train_step(spec, W, opt, x, y, dtype=torch.float64):
    r = loss_and_grads(spec, W, x, y, dtype)
    opt.step(W, r["grads"])
    return r


# ------------------------------------------------------------------------------------------------------------------
# synthetic data: z-scored volumes, smooth-blob labels with ~30 % foreground
# ------------------------------------------------------------------------------------------------------------------
def synthetic_batch(shape, seed_x=1234, seed_y=1235, fg=0.30):
    """shape = (N, C, X, Y, Z).  x ~ N(0,1) fp32; y = (gaussian-filtered noise > its (1-fg) quantile) uint8."""
    from scipy.ndimage import gaussian_filter
    rs = np.random.RandomState(seed_x)
    x = rs.randn(*shape).astype(np.float32)
    rs = np.random.RandomState(seed_y)
    N = shape[0]
    sp = shape[2:]
    y = np.zeros((N, 1) + tuple(sp), np.uint8)
    for n in range(N):
        f = gaussian_filter(rs.randn(*sp), sigma=[min(4.0, s / 8.0) for s in sp])
        y[n, 0] = (f > np.quantile(f, 1.0 - fg)).astype(np.uint8)
    return x, y
