"""Execution engine for the reference's U-Net topology on the C ABI (explicit forward + backward, no autograd).

Mirrors the graph built by reference fetal_net/model/unet3d/unet.py:17-86 (and the 2-D twin model/unet/unet.py:22-88,
run as D == 1 volumes): per level two Conv(3x3x3,'same')+ReLU blocks, MaxPooling(2), then per decoder level
UpSampling(2) -> concatenate([up, skip]) -> two conv blocks, final 1x1x1 conv + sigmoid; loss = -Dice
(metrics.py:11-32); optimizer = Keras Adam (unet.py:85).

MI355X layout decisions
  * activations channels-last [N][D][H][W][C] in the compute dtype (bf16 for speed, fp32 for parity);
  * all parameters live in ONE flat fp32 buffer ordered in BACKWARD-completion order (final layer first, first
    encoder conv last) so that gradient buckets for the RCCL all-reduce are contiguous prefixes, Adam is one launch
    and gradient zeroing is one memset;
  * the up-sampled tensor and the concat are never materialised (the conv kernels read two sources);
  * every tensor is allocated once at build time (288 GB HBM: no re-computation, no allocator traffic in the step).
"""
import math
from collections import OrderedDict

import numpy as np
import torch

from . import ops
from ._lib import ACT_NONE, ACT_RELU, IMPL_AUTO


class UNetPlan:
    """Topology in Keras creation order (names match Keras auto-naming)."""

    def __init__(self, in_channels, spatial, depth=4, n_base_filters=32, n_labels=1, ndim=3):
        self.in_channels, self.spatial = in_channels, tuple(spatial)
        self.depth, self.n_base_filters, self.n_labels, self.ndim = depth, n_base_filters, n_labels, ndim
        d = "3d" if ndim == 3 else "2d"
        k = 0

        def conv_name():
            nonlocal k
            k += 1
            return "conv%s_%d" % (d, k)

        self.enc = []
        cin = in_channels
        for ld in range(depth):
            lv = []
            for mult in (1, 2):
                cout = n_base_filters * (2 ** ld) * mult
                lv.append(dict(name=conv_name(), cin=cin, cout=cout, level=ld))
                cin = cout
            self.enc.append(lv)
        self.dec = []
        for ld in range(depth - 2, -1, -1):
            skip_c = self.enc[ld][1]["cout"]
            a = dict(name=conv_name(), cin=cin + skip_c, cout=skip_c, level=ld, c_up=cin, c_skip=skip_c)
            b = dict(name=conv_name(), cin=skip_c, cout=skip_c, level=ld)
            self.dec.append([a, b])
            cin = skip_c
        self.final = dict(name=conv_name(), cin=cin, cout=n_labels)
        for s in self.spatial:
            if s % (2 ** (depth - 1)):
                raise ValueError("spatial dims %s must be divisible by 2^(depth-1)" % (self.spatial,))

    def convs_forward_order(self):
        out = []
        for lv in self.enc:
            out += lv
        for lv in self.dec:
            out += lv
        return out

    def level_dims(self, level, slices=1):
        """(D,H,W) of the activations at `level`.  2-D plans are planar: D = number of slices in the batch, never pooled."""
        if self.ndim == 3:
            return tuple(s >> level for s in self.spatial)
        return (slices,) + tuple(s >> level for s in self.spatial)


class UNetEngine:
    def __init__(self, plan, batch, dtype=torch.bfloat16, device="cuda", seed=42, training=True, dist_ctx=None):
        from ._lib import lib
        lib()  # fail loudly if the HIP library is missing
        self.plan, self.N, self.dtype, self.dev, self.training = plan, batch, dtype, torch.device(device), training
        self.planar = plan.ndim == 2      # 2-D: tensors are [1][slices][H][W][C], every op is planar (no coupling along D)
        self.dist = dist_ctx
        self.t = 0                       # Adam step counter
        self._bufsets = {}
        self._build_params(seed)
        self.set_batch(batch)

    # ------------------------------------------------------------------------------------------------ parameters
    def _build_params(self, seed):
        p = self.plan
        # backward-completion order: final, dec (shallowest level first, b before a), enc (deepest level first, b before a)
        order = [("final", p.final)]
        for lv in reversed(p.dec):
            order += [("conv", lv[1]), ("conv", lv[0])]
        for lv in reversed(p.enc):
            order += [("conv", lv[1]), ("conv", lv[0])]
        self.layout = OrderedDict()
        off = 0
        for kind, c in order:
            nw = (27 if kind == "conv" else 1) * c["cout"] * c["cin"]
            off = (off + 3) & ~3
            self.layout[c["name"]] = dict(kind=kind, w=(off, nw), b=(((off + nw + 3) & ~3), c["cout"]), cin=c["cin"], cout=c["cout"])
            off = ((off + nw + 3) & ~3) + c["cout"]
        self.n_flat = (off + 3) & ~3
        dev = self.dev
        self.P = torch.zeros(self.n_flat, dtype=torch.float32, device=dev)
        if self.training:
            self.G = torch.zeros_like(self.P)
            self.M = torch.zeros_like(self.P)
            self.V = torch.zeros_like(self.P)
        # compute-dtype copies of the 3x3x3 filters
        self.Wf, self.Wd = {}, {}
        first = p.enc[0][0]["name"]
        for name, L in self.layout.items():
            if L["kind"] != "conv":
                continue
            self.Wf[name] = torch.empty((27, L["cout"], L["cin"]), dtype=self.dtype, device=dev)
            if self.training and name != first:
                self.Wd[name] = torch.empty((27, L["cin"], L["cout"]), dtype=self.dtype, device=dev)
        self.init_glorot(seed)

    def w_view(self, name, buf=None):
        L = self.layout[name]
        buf = self.P if buf is None else buf
        o, n = L["w"]
        shape = (27, L["cout"], L["cin"]) if L["kind"] == "conv" else (L["cout"], L["cin"])
        return buf[o:o + n].view(shape)

    def b_view(self, name, buf=None):
        L = self.layout[name]
        buf = self.P if buf is None else buf
        o, n = L["b"]
        return buf[o:o + n]

    def init_glorot(self, seed):
        """Keras defaults (glorot_uniform kernels, zero bias), drawn in Keras layer-creation order."""
        rs = np.random.RandomState(seed)
        W = OrderedDict()
        k3, k1 = ((3, 3, 3), (1, 1, 1)) if self.plan.ndim == 3 else ((3, 3), (1, 1))
        for c in self.plan.convs_forward_order():
            W[c["name"] + "/kernel"] = _glorot(rs, k3 + (c["cin"], c["cout"]))
            W[c["name"] + "/bias"] = np.zeros(c["cout"], np.float32)
        f = self.plan.final
        W[f["name"] + "/kernel"] = _glorot(rs, k1 + (f["cin"], f["cout"]))
        W[f["name"] + "/bias"] = np.zeros(f["cout"], np.float32)
        self.load_keras_weights(W)

    def load_keras_weights(self, W):
        """W: {'<layer>/kernel': (kD,kH,kW,Cin,Cout) ndarray, '<layer>/bias': (Cout,)} in Keras layout."""
        host = np.zeros(self.n_flat, np.float32)
        for name, L in self.layout.items():
            k = np.asarray(W[name + "/kernel"], np.float32)
            if k.ndim == 4:  # 2-D kernel (kH,kW,Cin,Cout) -> centre plane of a 3x3x3 / 1x1x1 kernel
                k = _embed_2d_kernel(k)
            o, n = L["w"]
            if L["kind"] == "conv":
                assert k.shape == (3, 3, 3, L["cin"], L["cout"]), (name, k.shape)
                host[o:o + n] = k.transpose(0, 1, 2, 4, 3).reshape(-1)        # -> [27][Cout][Cin]
            else:
                assert k.shape[-2:] == (L["cin"], L["cout"]), (name, k.shape)
                host[o:o + n] = k.reshape(L["cin"], L["cout"]).T.reshape(-1)  # -> [L][C]
            ob, nb = L["b"]
            host[ob:ob + nb] = np.asarray(W[name + "/bias"], np.float32)
        self.P.copy_(torch.from_numpy(host))
        self.refresh_weight_copies()

    def export_keras_weights(self):
        host = self.P.detach().cpu().numpy()
        W = OrderedDict()
        order = [c["name"] for c in self.plan.convs_forward_order()] + [self.plan.final["name"]]
        for name in order:
            L = self.layout[name]
            o, n = L["w"]
            if L["kind"] == "conv":
                k = host[o:o + n].reshape(3, 3, 3, L["cout"], L["cin"]).transpose(0, 1, 2, 4, 3)
                W[name + "/kernel"] = (k if self.plan.ndim == 3 else k[1]).copy()          # 2-D: the centre kd plane is the 3x3 kernel
            else:
                k = host[o:o + n].reshape(L["cout"], L["cin"]).T
                W[name + "/kernel"] = k.reshape(((1, 1, 1) if self.plan.ndim == 3 else (1, 1)) + (L["cin"], L["cout"])).copy()
            ob, nb = L["b"]
            W[name + "/bias"] = host[ob:ob + nb].copy()
        return W

    def refresh_weight_copies(self):
        for name in self.Wf:
            ops.pack_weights(self.w_view(name), self.Wf[name], self.Wd.get(name))

    # ------------------------------------------------------------------------------------------------ buffers
    def set_batch(self, N):
        """switch to (and lazily allocate) the activation / gradient buffer set for batch size N"""
        key = (N, self.training)
        if key not in self._bufsets:
            self.N = N
            self._build_buffers()
            self._bufsets[key] = dict(act=self.act, grad=getattr(self, "grad", None), logits=self.logits, probs=self.probs,
                                      dlogits=getattr(self, "dlogits", None),
                                      dummy_y=torch.zeros(self.logits.numel(), dtype=torch.uint8, device=self.dev))
        b = self._bufsets[key]
        self.N, self.act, self.grad, self.logits, self.probs, self.dlogits = N, b["act"], b["grad"], b["logits"], b["probs"], b["dlogits"]
        self._dummy_y = b["dummy_y"]       # per buffer set and never freed: captured hipGraphs keep raw pointers to it

    def _dims(self, level):
        """leading (N, D, H, W) of an activation at `level` for the current batch"""
        if self.planar:
            return (1,) + self.plan.level_dims(level, self.N)
        return (self.N,) + self.plan.level_dims(level)

    def _build_buffers(self):
        p, N, dt, dev = self.plan, self.N, self.dtype, self.dev
        A = self.act = {}
        for lv in p.enc:
            for c in lv:
                A[c["name"]] = torch.empty(self._dims(c["level"]) + (c["cout"],), dtype=dt, device=dev)
        for ld in range(p.depth - 1):
            A["pool_%d" % ld] = torch.empty(self._dims(ld + 1) + (p.enc[ld][1]["cout"],), dtype=dt, device=dev)
        for lv in p.dec:
            for c in lv:
                A[c["name"]] = torch.empty(self._dims(c["level"]) + (c["cout"],), dtype=dt, device=dev)
        nvox0 = int(np.prod(self._dims(0)))
        self.logits = torch.empty((nvox0, p.n_labels), dtype=torch.float32, device=dev)
        self.probs = torch.empty_like(self.logits)
        self.sums = torch.zeros(8, dtype=torch.float64, device=dev)
        if not self.training:
            self.grad, self.dlogits = None, None
            return
        Gd = self.grad = {}
        for name, t in A.items():
            if name.startswith("pool_"):
                Gd[name] = torch.empty_like(t)        # gradient w.r.t. the pooled tensor (un-masked)
            else:
                Gd[name] = torch.empty_like(t)        # gradient w.r.t. the conv's pre-activation
        for lv in p.dec:
            a = lv[0]
            Gd["cat_%d" % a["level"]] = torch.empty(self._dims(a["level"]) + (a["cin"],), dtype=dt, device=dev)
        self.dlogits = torch.empty_like(self.logits)

    # ------------------------------------------------------------------------------------------------ forward
    def forward(self, x):
        """x: [N,D,H,W,Cin] compute dtype, device.  Leaves logits (fp32 [nvox, L]) in self.logits."""
        p, A = self.plan, self.act
        assert tuple(x.shape) == self._dims(0) + (p.in_channels,), (x.shape,)
        self.x_in = x
        h = x
        for ld, lv in enumerate(p.enc):
            for c in lv:
                ops.conv3d_fwd(h, None, self.Wf[c["name"]], self.b_view(c["name"]), A[c["name"]], act=ACT_RELU, planar=self.planar)
                h = A[c["name"]]
            if ld < p.depth - 1:
                h = ops.maxpool_fwd(h, A["pool_%d" % ld], planar=self.planar)
        for lv in p.dec:
            a, b = lv
            skip = A[p.enc[a["level"]][1]["name"]]
            ops.conv3d_fwd(h, skip, self.Wf[a["name"]], self.b_view(a["name"]), A[a["name"]], up0=True, act=ACT_RELU, planar=self.planar)
            ops.conv3d_fwd(A[a["name"]], None, self.Wf[b["name"]], self.b_view(b["name"]), A[b["name"]], act=ACT_RELU, planar=self.planar)
            h = A[b["name"]]
        f = p.final
        ops.conv1x1_fwd(h, self.w_view(f["name"]), self.b_view(f["name"]), self.logits)
        return self.logits

    def loss_forward(self, y_true):
        """y_true uint8 [nvox*L] device.  probs + the 8 metric sums (accumulated into zeroed self.sums)."""
        self.sums.zero_()
        ops.sigmoid_dice_fwd(self.logits, y_true, self.probs, self.sums)
        if self.dist is not None and self.dist.world > 1 and self.dist.global_dice:
            self.dist.all_reduce_sums(self.sums)
        return self.sums

    def predict(self, x):
        self.forward(x)
        self.sums.zero_()
        # sigmoid only (y_true is irrelevant for probs): reuse the fused kernel with an all-zero label buffer
        ops.sigmoid_dice_fwd(self.logits, self._dummy_y, self.probs, self.sums)
        return self.probs

    # ------------------------------------------------------------------------------------------------ backward
    def backward(self, y_true, grad_scale=1.0):
        p, A, Gd = self.plan, self.act, self.grad
        self.G.zero_()
        ops.sigmoid_dice_bwd(self.probs, y_true, self.sums, self.dlogits, smooth=1.0, grad_scale=grad_scale)
        f = p.final
        last = p.dec[-1][1] if p.dec else p.enc[-1][1]
        ops.conv1x1_bwd(A[last["name"]], self.w_view(f["name"]), self.dlogits, Gd[last["name"]], self.w_view(f["name"], self.G),
                        self.b_view(f["name"], self.G), relu_mask=True)
        self._grad_ready(f["name"])
        # decoder, shallowest level first
        for lv in reversed(p.dec):
            a, b = lv
            ld = a["level"]
            # block b: input = A[a]
            ops.conv3d_wgrad(A[a["name"]], None, Gd[b["name"]], self.w_view(b["name"], self.G), self.b_view(b["name"], self.G), planar=self.planar)
            self._grad_ready(b["name"])
            ops.conv3d_dgrad(Gd[b["name"]], self.Wd[b["name"]], Gd[a["name"]], mask=A[a["name"]], planar=self.planar)
            # block a: input = [up(low) | skip]
            low = self._dec_input_name(ld)
            skip = A[p.enc[ld][1]["name"]]
            ops.conv3d_wgrad(A[low], skip, Gd[a["name"]], self.w_view(a["name"], self.G), self.b_view(a["name"], self.G), up0=True, planar=self.planar)
            self._grad_ready(a["name"])
            cat = Gd["cat_%d" % ld]
            ops.conv3d_dgrad(Gd[a["name"]], self.Wd[a["name"]], cat, planar=self.planar)
            ops.upsample_bwd(cat, Gd[low], dy_off=0, xmask=A[low], planar=self.planar)
        # encoder, deepest level first
        for ld in range(p.depth - 1, -1, -1):
            ca, cb = p.enc[ld]
            if ld < p.depth - 1:
                # gradient of enc[ld].b output = pooled path + skip path, masked by its ReLU
                cat = Gd["cat_%d" % ld]
                ops.maxpool_bwd(A[cb["name"]], Gd["pool_%d" % ld], Gd[cb["name"]], add=cat, add_off=cat.shape[-1] - cb["cout"],
                                relu_mask=True, planar=self.planar)
            ops.conv3d_wgrad(A[ca["name"]], None, Gd[cb["name"]], self.w_view(cb["name"], self.G), self.b_view(cb["name"], self.G),
                             planar=self.planar)
            self._grad_ready(cb["name"])
            ops.conv3d_dgrad(Gd[cb["name"]], self.Wd[cb["name"]], Gd[ca["name"]], mask=A[ca["name"]], planar=self.planar)
            xin = self.x_in if ld == 0 else A["pool_%d" % (ld - 1)]
            ops.conv3d_wgrad(xin, None, Gd[ca["name"]], self.w_view(ca["name"], self.G), self.b_view(ca["name"], self.G), planar=self.planar)
            self._grad_ready(ca["name"])
            if ld > 0:
                ops.conv3d_dgrad(Gd[ca["name"]], self.Wd[ca["name"]], Gd["pool_%d" % (ld - 1)], planar=self.planar)
        if self.dist is not None:
            self.dist.finish(self)

    def _dec_input_name(self, ld):
        """name of the activation that is up-sampled into decoder level ld"""
        p = self.plan
        if ld == p.depth - 2:
            return p.enc[-1][1]["name"]
        for lv in p.dec:
            if lv[0]["level"] == ld + 1:
                return lv[1]["name"]
        raise KeyError(ld)

    def _grad_ready(self, name):
        if self.dist is not None:
            self.dist.grad_ready(self, name)

    # ------------------------------------------------------------------------------------------------ optimizer
    def adam_step(self, lr, beta1=0.9, beta2=0.999, eps=1e-7, grad_scale=1.0):
        self.t += 1
        lr_t = lr * math.sqrt(1.0 - beta2 ** self.t) / (1.0 - beta1 ** self.t)
        ops.adam_step(self.P, self.G, self.M, self.V, lr_t, beta1, beta2, eps, grad_scale)
        self.refresh_weight_copies()

    def train_step(self, x, y_true, lr):
        """one full step: forward, Dice, backward, (all-reduce), Adam.  Returns the device tensor of metric sums."""
        self.forward(x)
        self.loss_forward(y_true)
        self.backward(y_true)
        self.adam_step(lr)
        return self.sums

    @staticmethod
    def metrics_from_sums(s, smooth=1.0):
        s = [float(v) for v in s]
        dice = (2.0 * s[0] + smooth) / (s[1] + s[2] + smooth)
        vod = (s[3] + smooth) / (s[4] + s[5] - s[3] + smooth)
        return dict(loss=-dice, dice_coefficient=dice, vod_coefficient=vod, binary_accuracy=s[6] / max(s[7], 1.0))


def _glorot(rs, shape):
    rf = int(np.prod(shape[:-2]))
    lim = math.sqrt(6.0 / (rf * shape[-2] + rf * shape[-1]))
    return rs.uniform(-lim, lim, size=shape).astype(np.float32)


def _embed_2d_kernel(k):
    kh, kw, ci, co = k.shape
    if kh == 1:
        return k.reshape(1, 1, 1, ci, co)
    out = np.zeros((3, 3, 3, ci, co), np.float32)
    out[1] = k
    return out
