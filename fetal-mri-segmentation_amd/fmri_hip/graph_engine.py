"""Generic executor for a recorded Keras-style layer graph (fetal_net.model.graph.Graph) on the C ABI.

The hand-scheduled `UNetEngine` stays the path of the headline U-Net; this interpreter runs any graph made of the layer
classes the reference's builders emit — Conv3D (3x3x3 stride 1|2, 1x1x1), InstanceNormalization / BatchNormalization,
LeakyReLU / Activation('relu'|'sigmoid'), MaxPooling3D, UpSampling3D, Concatenate, Add, SpatialDropout3D — i.e. the Isensee
model of reference fetal_net/model/unet3d/isensee2017.py:15-111 (and, for cross-checking, unet_model_3d) — and their 2-D twins
(Conv2D, UpSampling2D, SpatialDropout2D, MaxPooling2D between two Permute layers: reference fetal_net/model/unet/isensee.py:14-105).
A graph may also end in Dense(1, 'sigmoid') on a GlobalAveragePooling3D (the PatchGAN discriminator of reference
fetal_net/model/discriminator/all_dis_3d.py:11-72: + AveragePooling3D, Dense, anisotropic strides (2, 2, 1)); its loss is the binary
cross-entropy on float targets, and with `input_grad=True` the backward pass also delivers dL/d(input) - what the generator of the
adversarial experiments is trained with (reference fetal/experiments/train_adv.py:165-180).
2-D graphs run PLANAR: the batch of slices (N,X,Y,C) is one tensor [1][N][X][Y][C], every kernel gets planar = 1 (no coupling along
the slice axis, pooling / up-sampling / stride in X and Y only), a 2-D filter is the centre kd plane of a 27-tap image.

Compile-time fusions (nothing of the fused kind is ever materialised):
  * UpSampling3D -> Conv3D(3x3x3)          => conv reads its source through the fused nearest x2 (up0)
  * Concatenate([a, b]) -> Conv3D(3x3x3)   => dual-source conv
  * Norm -> LeakyReLU | Activation('relu') => fmri_norm_act_fwd/bwd
  * Conv3D -> Activation('relu')           => activation in the conv epilogue
Backward walks the op list in reverse; a tensor with several consumers receives its gradient contributions through
`first writes, later ones accumulate`.  Parameters/gradients/Adam state are flat fp32 buffers like in UNetEngine.
"""
import math
import os
from collections import OrderedDict

import numpy as np
import torch

from . import ops
from ._lib import ACT_LEAKY, ACT_NONE, ACT_RELU, lib

LEAKY_ALPHA = 0.3      # keras.layers.LeakyReLU default (reference isensee2017.py:12)


class _Dims(object):
    def __init__(self, spatial, n_labels):
        self.spatial, self.n_labels, self.ndim = tuple(spatial), n_labels, len(spatial)

    def level_dims(self, level, slices=1):
        return tuple(s >> level for s in self.spatial)


class LayerGraphEngine(object):
    def __init__(self, layers, batch, dtype=torch.bfloat16, device="cuda", seed=42, training=True, dist_ctx=None, input_grad=False):
        lib()
        self.input_grad = bool(input_grad)      # keep dL/d(input) (channel-padded in bf16 mode: the caller hands over cp(C) channels)
        self.beta1 = 0.9
        self.layers = list(layers)
        self.by_name = OrderedDict((l.name, l) for l in self.layers)
        self.dtype, self.dev, self.training, self.dist = dtype, torch.device(device), training, dist_ctx
        self.planar = any(l.class_name == "Conv2D" for l in self.layers)      # 2-D graph: slices ride the kernels' D axis
        self.nd = 2 if self.planar else 3
        # bf16: every tensor carries its channels padded to a multiple of 32 (zero weights / zero activations in the padding) so that
        # all convolutions - including the 16- and 32-channel levels, the stride-2 and the 1x1x1 ones - run on the MFMA kernels:
        # stride 2 = the stride-1 conv sampled at every second voxel, 1x1x1 = the centre tap of a 27-tap filter.  fp32 (parity mode)
        # keeps the exact-size tensors and the fp32 VALU kernels.  FMRI_GRAPH_PAD=0 switches the padding off.
        import os
        self.pad = dtype == torch.bfloat16 and os.environ.get("FMRI_GRAPH_PAD", "1") != "0"
        self.t = 0
        self._wg_stream = (torch.cuda.Stream(device=self.dev) if (training and self.dev.type == "cuda" and os.environ.get("FMRI_WGRAD_STREAM", "1") != "0")
                           else None)
        self._fixed_drop = None
        self.loss_kind, self.loss_param = 0, 1.0
        self.sums = torch.zeros(16, dtype=torch.float64, device=self.dev)
        self._compile()
        self._build_params(seed)
        self._bufsets = {}
        self.set_batch(batch)

    # ------------------------------------------------------------------------------------------------ compile
    def _compile(self):
        L = self.layers
        absorbed = set()          # layers that never get a tensor of their own
        self.ops = []             # dicts: kind, out, ins, + params
        nd = self.nd
        CONV, UPS, DROP, POOL = "Conv%dD" % nd, "UpSampling%dD" % nd, "SpatialDropout%dD" % nd, "MaxPooling%dD" % nd
        k3, s1 = (3,) * nd, (1,) * nd
        self.input_name = L[0].name
        # Permute layers (the 2-D builders wrap a channels-first graph in two of them) own no tensor: the engine's tensors are
        # channels-last throughout, a Permute is just another name for its source
        self.alias = {}           # layer name -> name of the layer whose tensor holds its value
        for l in L:
            if l.class_name == "Permute":
                self.alias[l.name] = self.alias.get(l.inbound[0], l.inbound[0])
                absorbed.add(l.name)
        res = lambda n: self.alias.get(n, n)
        consumers = {l.name: [] for l in L}
        for l in L:
            if l.class_name == "Permute":
                continue
            for i in l.inbound:
                consumers[res(i)].append(l.name)
        self.consumers = consumers
        inb = {l.name: [res(i) for i in l.inbound] for l in L}
        out_layer = L[-1] if L[-1].class_name != "Permute" else self.by_name[L[-1].inbound[0]]
        self.head = "dense" if (out_layer.class_name == "Dense" and out_layer.config.get("activation") == "sigmoid") else "seg"
        if self.head == "seg" and not (out_layer.class_name == "Activation" and out_layer.config.get("activation") == "sigmoid"):
            raise NotImplementedError("the graph must end in Activation('sigmoid') or Dense(n, 'sigmoid')")
        # seg head: the sigmoid layer owns no tensor, its source holds the logits; dense head: the Dense op writes the logits itself
        self.logits_src = inb[out_layer.name][0] if self.head == "seg" else out_layer.name
        in_shape = L[0].output_shape                      # 3-D: (None, C, X, Y, Z); 2-D: (None, X, Y, C) channels-last
        if self.planar:
            self.in_channels = in_shape[-1]
            self.plan = _Dims(in_shape[1:3], out_layer.output_shape[1])
        else:
            self.in_channels = in_shape[1]
            self.plan = _Dims(in_shape[2:], out_layer.output_shape[1])
        self.shape = {l.name: (l.output_shape[1],) + tuple(l.output_shape[2:]) for l in L if l.class_name != "Permute"}   # (C, spatial...)
        self.flat = set(l.name for l in L if l.class_name in ("GlobalAveragePooling%dD" % nd, "Dense"))   # fp32 [N, C] tensors
        if self.planar:
            self.shape[self.input_name] = (self.in_channels,) + tuple(in_shape[1:3])
        self.convs, self.norms = OrderedDict(), OrderedDict()

        def single_consumer(name, cls=None):
            c = consumers[name]
            return len(c) == 1 and (cls is None or self.by_name[c[0]].class_name in cls)

        # producers that a later 3x3x3 stride-1 conv reads through (fused up-sampling / concatenation): decided up front because
        # they precede their consumer in the layer list
        for l in L:
            if l.class_name == CONV and tuple(l.config["kernel_size"]) == k3 and tuple(l.config.get("strides") or s1) == s1:
                src = self.by_name[inb[l.name][0]]
                if src.class_name == UPS and single_consumer(src.name):
                    absorbed.add(src.name)
                elif src.class_name == "Concatenate" and len(src.inbound) == 2 and single_consumer(src.name):
                    absorbed.add(src.name)
        APOOL, GAP = "AveragePooling%dD" % nd, "GlobalAveragePooling%dD" % nd
        self.denses = OrderedDict()
        for l in L:
            if l.name in absorbed or (l is out_layer and self.head == "seg"):
                continue
            cn = l.class_name
            if cn == "InputLayer":
                continue
            if cn == CONV:
                k, s = tuple(l.config["kernel_size"]), tuple(l.config.get("strides") or s1)
                src = self.by_name[inb[l.name][0]]
                if len(set(k)) != 1 or not set(s) <= {1, 2}:
                    raise NotImplementedError("convolution %s: kernel %s strides %s" % (l.name, k, s))
                # strides other than 1: in the channel-padded mode, and for anisotropic strides such as the discriminator's (2, 2, 1),
                # the convolution runs at stride 1 and every second voxel along the strided axes is kept (`sub`)
                sub = s if (s != s1 and (self.pad or len(set(s)) != 1)) else None
                if sub is not None and k != k3:
                    raise NotImplementedError("strided %s convolution %s" % (k, l.name))
                op = dict(kind="conv", name=l.name, out=l.name, k=k[0], s=(1 if sub is not None else s[0]), sub=sub, act=ACT_NONE, up0=False,
                          ins=[inb[l.name][0]])
                if k == k3 and s == s1:
                    if src.class_name == UPS and single_consumer(src.name):
                        absorbed.add(src.name)
                        op["ins"], op["up0"] = [inb[src.name][0]], True
                    elif src.class_name == "Concatenate" and len(src.inbound) == 2 and single_consumer(src.name):
                        absorbed.add(src.name)
                        op["ins"] = list(inb[src.name])
                nxt = self.by_name[consumers[l.name][0]] if single_consumer(l.name) else None
                if nxt is not None and nxt.class_name == "Activation" and nxt.config.get("activation") == "relu":
                    absorbed.add(nxt.name)
                    op["act"], op["out"] = ACT_RELU, nxt.name
                self.convs[l.name] = op
                self.ops.append(op)
            elif cn in ("InstanceNormalization", "BatchNormalization"):
                op = dict(kind="norm", name=l.name, out=l.name, ins=[inb[l.name][0]], instance=(cn == "InstanceNormalization"), act=ACT_NONE)
                nxt = self.by_name[consumers[l.name][0]] if single_consumer(l.name) else None
                if nxt is not None and (nxt.class_name == "LeakyReLU" or (nxt.class_name == "Activation" and nxt.config.get("activation") == "relu")):
                    absorbed.add(nxt.name)
                    op["act"], op["out"] = (ACT_LEAKY if nxt.class_name == "LeakyReLU" else ACT_RELU), nxt.name
                self.norms[l.name] = op
                self.ops.append(op)
            elif cn == "Add":
                self.ops.append(dict(kind="add", out=l.name, ins=list(inb[l.name])))
            elif cn == DROP:
                self.ops.append(dict(kind="dropout", out=l.name, ins=[inb[l.name][0]], rate=float(l.config.get("rate", 0.0))))
            elif cn == UPS:
                self.ops.append(dict(kind="upsample", out=l.name, ins=[inb[l.name][0]]))
            elif cn == POOL:
                self.ops.append(dict(kind="maxpool", out=l.name, ins=[inb[l.name][0]]))
            elif cn == APOOL:
                if tuple(l.config.get("pool_size")) != (2,) * nd:
                    raise NotImplementedError("AveragePooling with pool_size %s" % (l.config.get("pool_size"),))
                self.ops.append(dict(kind="avgpool", out=l.name, ins=[inb[l.name][0]]))
            elif cn == GAP:
                self.ops.append(dict(kind="gap", out=l.name, ins=[inb[l.name][0]]))
            elif cn == "Dense":
                a = l.config.get("activation")
                if a not in (None, "leaky_relu", "relu") and not (l is out_layer and a == "sigmoid"):
                    raise NotImplementedError("Dense activation %r (%s)" % (a, l.name))
                op = dict(kind="dense", name=l.name, out=l.name, ins=[inb[l.name][0]], units=int(l.config["units"]),
                          act={None: ACT_NONE, "sigmoid": ACT_NONE, "relu": ACT_RELU, "leaky_relu": ACT_LEAKY}[a])
                self.denses[l.name] = op
                self.ops.append(op)
            elif cn == "Concatenate":
                raise NotImplementedError("Concatenate is only supported in front of a 3x3(x3) convolution (layer %s)" % l.name)
            else:
                raise NotImplementedError("layer class %s (%s) is not executable on the engine yet" % (cn, l.name))
        self.clog = {n: sh[0] for n, sh in self.shape.items()}          # logical channel counts
        if self.pad:
            for n, sh in self.shape.items():
                if n in self.flat and self.by_name[n].class_name == "Dense":
                    continue                                  # dense outputs keep their logical width (fp32, no MFMA tiling)
                if n != self.input_name or self.input_grad:
                    self.shape[n] = (self._cp(sh[0]),) + tuple(sh[1:])
        for o in self.ops:                                   # shapes of op outputs follow the layer whose name they carry
            o["shape"] = self.shape[o["out"]]

    def _cp(self, c):
        """physical channel count: the MFMA forward / input-gradient kernels tile channels by 32"""
        return ((c + 31) // 32) * 32 if self.pad else c

    # ------------------------------------------------------------------------------------------------ parameters
    def _build_params(self, seed):
        self.layout = OrderedDict()
        off = 0
        for name, op in self.convs.items():
            cin = sum(self.clog[i] for i in op["ins"])
            cout = self.clog[name]
            nw = op["k"] ** 3 * cout * cin
            off = (off + 3) & ~3
            self.layout[name] = dict(kind="conv", w=(off, nw), b=((off + nw + 3) & ~3, cout), cin=cin, cout=cout, k=op["k"])
            off = ((off + nw + 3) & ~3) + cout
        for name, op in self.norms.items():
            c = self.clog[name]
            off = (off + 3) & ~3
            self.layout[name] = dict(kind="norm", gamma=(off, c), beta=(off + ((c + 3) & ~3), c), c=c)
            off = off + ((c + 3) & ~3) + c
        for name, op in self.denses.items():
            K, M = self.clog[op["ins"][0]], op["units"]
            off = (off + 3) & ~3
            self.layout[name] = dict(kind="dense", w=(off, K * M), b=((off + K * M + 3) & ~3, M), K=K, M=M)
            off = ((off + K * M + 3) & ~3) + M
        self.n_flat = (off + 3) & ~3
        self.P = torch.zeros(self.n_flat, dtype=torch.float32, device=self.dev)
        if self.training:
            self.G, self.M, self.V = torch.zeros_like(self.P), torch.zeros_like(self.P), torch.zeros_like(self.P)
        self.Wf, self.Wd, self.Wup, self._dy64_buf = {}, {}, {}, {}
        self.Ws2, self._s2 = {}, None
        if self.pad:
            self._build_padded_params()
        else:
            for name, op in self.convs.items():
                Lc = self.layout[name]
                self.Wf[name] = torch.empty((op["k"] ** 3, Lc["cout"], Lc["cin"]), dtype=self.dtype, device=self.dev)
                if self.training and op["k"] == 3 and op["s"] == 1 and not self._is_input(op["ins"]):
                    self.Wd[name] = torch.empty((27, Lc["cin"], Lc["cout"]), dtype=self.dtype, device=self.dev)
        self.init_glorot(seed)

    def _build_padded_params(self):
        """physical (channel-padded, always 27-tap) images of the parameters and of their gradients; the flat fp32 buffers P/G/M/V keep
        the logical Keras-shaped parameters, `refresh_weight_copies` scatters them in, backward gathers the gradients out"""
        dev, f32 = self.dev, torch.float32
        self.Wp32, self.bp, self.dWp, self.dbp, self.cin_map = {}, {}, {}, {}, {}
        self.gp, self.betap, self.dgp, self.dbetap, self.dDense = {}, {}, {}, {}, {}
        # All parameter images live in ONE flat buffer Pp and all gradient images in ONE flat buffer Gp (16-byte aligned pieces): a step
        # then costs one scatter P -> Pp, one fill of Gp and one gather Gp -> G instead of ~300 per-layer fills / index copies / adds
        # (isensee2017_model_3d defaults: 147 fills + 110 adds + 54 index kernels per step, ~2 ms of launch-latency-bound work).
        # map_p / map_g: for every element of the logical flat buffers (P / G) the position of its twin in Pp / Gp; alignment holes of
        # the logical layout point at a spare last element.
        segs_p, segs_g = [], []                               # (key, kind, shape)
        geo = {}
        for name, op in self.convs.items():
            coutp = self.shape[name][0]
            idx, base = [], 0
            for i in op["ins"]:
                c = self.clog[i]
                idx += list(range(base, base + c))
                base += self.shape[i][0] if (i != self.input_name or self.input_grad) else c
            cinp = base
            c64 = ((coutp + 63) // 64) * 64                  # the weight-gradient kernel may be fed a dy zero-extended to 64 channels
            # ... unless the kd-sharing weight-gradient kernel (32 x 32 blocks) takes the 32-channel dy as it is: then the gradient image
            # has the filter's own 32 rows per tap.  Decided once, for batch 1 (larger batches only raise the launch's FLOPs, the threshold)
            self._c32 = getattr(self, "_c32", {})
            self._c32[name] = False
            if coutp % 64 and not self.planar and self.dtype == torch.bfloat16 and op["k"] == 3:
                from ._lib import lib, BF16
                sp = tuple(self.shape[op["ins"][0]][1:]) if op.get("sub") is not None else tuple(self.shape[name][1:])
                if op["up0"] and op.get("sub") is None:
                    pass                                      # (fused up-sampling: the per-kd kernel)
                elif len(sp) == 3 and lib().fmri_conv3d_wgrad_cout32_ok(cinp, 0, coutp, 1, sp[0], sp[1], sp[2], BF16, 0, 0):
                    self._c32[name] = True
                    c64 = coutp
            geo[name] = (coutp, cinp, c64, np.asarray(idx, np.int64))
            self.cin_map[name] = torch.tensor(idx, dtype=torch.long, device=dev)
            segs_p += [((name, "w"), (27, coutp, cinp)), ((name, "b"), (coutp,))]
            segs_g += [((name, "w"), (27, c64, cinp)), ((name, "b"), (c64,))]
        for name in self.norms:
            cp = self.shape[name][0]
            segs_p += [((name, "gamma"), (cp,)), ((name, "beta"), (cp,))]
            segs_g += [((name, "gamma"), (cp,)), ((name, "beta"), (cp,))]
        for name in self.denses:
            Lc = self.layout[name]
            segs_g += [((name, "w"), (Lc["K"], Lc["M"])), ((name, "b"), (Lc["M"],))]

        def carve(segs):
            offs, off = {}, 0
            for key, shape in segs:
                offs[key] = (off, shape)
                off += (int(np.prod(shape)) + 3) & ~3
            return offs, off

        offs_p, n_p = carve(segs_p)
        offs_g, n_g = carve(segs_g)
        self.Pp = torch.zeros(n_p + 4, dtype=f32, device=dev)
        self.Gp = torch.zeros(n_g + 4, dtype=f32, device=dev) if self.training else None

        def view(buf, offs, key):
            off, shape = offs[key]
            return buf[off:off + int(np.prod(shape))].view(shape)

        map_p = np.full(self.n_flat, n_p, np.int64)
        map_g = np.full(self.n_flat, n_g, np.int64)
        for name, op in self.convs.items():
            Lc = self.layout[name]
            coutp, cinp, c64, cmap = geo[name]
            self.Wp32[name], self.bp[name] = view(self.Pp, offs_p, (name, "w")), view(self.Pp, offs_p, (name, "b"))
            self.Wf[name] = torch.empty((27, coutp, cinp), dtype=self.dtype, device=dev)
            k3 = Lc["k"] ** 3
            taps = (np.arange(27) if k3 == 27 else np.array([13]))[:, None, None]           # 1x1x1 = the centre tap of the 27-tap image
            co = np.arange(Lc["cout"])[None, :, None]
            ow, nw = Lc["w"]
            ob, nb = Lc["b"]
            map_p[ow:ow + nw] = (offs_p[(name, "w")][0] + (taps * coutp + co) * cinp + cmap[None, None, :]).reshape(-1)
            map_p[ob:ob + nb] = offs_p[(name, "b")][0] + np.arange(nb)
            if self.training:
                self.dWp[name], self.dbp[name] = view(self.Gp, offs_g, (name, "w")), view(self.Gp, offs_g, (name, "b"))
                map_g[ow:ow + nw] = (offs_g[(name, "w")][0] + (taps * c64 + co) * cinp + cmap[None, None, :]).reshape(-1)
                map_g[ob:ob + nb] = offs_g[(name, "b")][0] + np.arange(nb)
                if not self._is_input(op["ins"]):
                    self.Wd[name] = torch.empty((27, cinp, coutp), dtype=self.dtype, device=dev)
        for name in self.norms:
            for key in ("gamma", "beta"):
                o_, n_ = self.layout[name][key]
                map_p[o_:o_ + n_] = offs_p[(name, key)][0] + np.arange(n_)
                map_g[o_:o_ + n_] = offs_g[(name, key)][0] + np.arange(n_)
        for name in self.denses:
            Lc = self.layout[name]
            for key in ("w", "b"):
                o_, n_ = Lc[key]
                map_g[o_:o_ + n_] = offs_g[(name, key)][0] + np.arange(n_)
            if self.training:
                self.dDense[name] = (view(self.Gp, offs_g, (name, "w")), view(self.Gp, offs_g, (name, "b")))
        self.map_p = torch.from_numpy(map_p).to(dev)
        self.map_g = torch.from_numpy(map_g).to(dev)
        # Conv3D(3x3x3, strides 2) (reference isensee2017.py:51): on the parity kernels (fmri_hip/strided_parity.py) where the shapes allow -
        # forward + input gradient (bit 0), weight gradient (bit 1: 64-wide blocks of the input's channels); FMRI_S2_PARITY=0: A/B
        self.Ws2, self._s2 = {}, None
        if self.pad and not self.planar and self.dtype == torch.bfloat16 and os.environ.get("FMRI_S2_PARITY", "1") != "0":
            for name, op in self.convs.items():
                if op.get("sub") != (2, 2, 2) or op["k"] != 3 or op["act"] != ACT_NONE or len(op["ins"]) != 1:
                    continue
                coutp, cinp = self.shape[name][0], self.Wp32[name].shape[2]
                fine = tuple(self.shape[op["ins"][0]][1:])
                if any(d % 2 for d in fine) or self._is_input(op["ins"]):
                    continue
                ok = ops.conv3d_upcat_ok(coutp, 0, cinp, fine[0], fine[1], fine[2], self.dtype)
                if ok & 1:
                    if self._s2 is None:
                        from .strided_parity import StridedParity
                        self._s2 = StridedParity(dev)
                    self.Ws2[name] = dict(fwd=torch.zeros((8, 8, coutp, cinp), dtype=self.dtype, device=dev),
                                          dgrad=torch.zeros((8, 8, cinp, coutp), dtype=self.dtype, device=dev) if self.training else None,
                                          wgrad=bool(ok & 2) and self.training)
                    if self.Ws2[name]["wgrad"]:
                        self.Ws2[name].update(dw27=torch.zeros((27, cinp, coutp), dtype=f32, device=dev), db=torch.zeros(cinp, dtype=f32, device=dev))
        # UpSampling3D -> Conv3D (reference isensee2017.py:101-104): parity form, 8 pre-summed 2x2x2 filters on the low-res tensor
        self.Wup, self.dwc_scratch = {}, None
        for name, op in self.convs.items():
            if op["up0"] and len(op["ins"]) == 1 and op["s"] == 1 and op["k"] == 3 and not self.planar:     # (2-D: fused-upsample 9-tap kernels)
                coutp, cinp = self.shape[name][0], self.Wp32[name].shape[2]
                ok = ops.conv3d_upcat_ok(cinp, 0, coutp, *self.shape[name][1:], self.dtype)
                if ok & 1:
                    W = dict(up_f=torch.empty((8, 8, coutp, cinp), dtype=self.dtype, device=dev), up_d=None, wgrad=bool(ok & 2))
                    if self.training:
                        W["up_d"] = torch.empty((8, 8, cinp, coutp), dtype=self.dtype, device=dev)
                    self.Wup[name] = W
        need = [64 * self.shape[n][0] * self.Wp32[n].shape[2] for n, W in list(self.Wup.items()) + list(self.Ws2.items()) if W["wgrad"]]
        if self.training and need:
            self.dwc_scratch = torch.empty(max(need), dtype=f32, device=dev)
        for name in self.norms:
            self.gp[name], self.betap[name] = view(self.Pp, offs_p, (name, "gamma")), view(self.Pp, offs_p, (name, "beta"))
            if self.training:
                self.dgp[name], self.dbetap[name] = view(self.Gp, offs_g, (name, "gamma")), view(self.Gp, offs_g, (name, "beta"))

    def _is_input(self, ins):
        """a conv that reads the graph's input needs no input-gradient filters - unless the caller wants dL/d(input)"""
        return len(ins) == 1 and ins[0] == self.input_name and not self.input_grad

    def _v(self, name, which, buf=None):
        buf = self.P if buf is None else buf
        o, n = self.layout[name][which]
        return buf[o:o + n]

    def w_view(self, name, buf=None):
        Lc = self.layout[name]
        return self._v(name, "w", buf).view(Lc["k"] ** 3, Lc["cout"], Lc["cin"])

    def init_glorot(self, seed):
        rs = np.random.RandomState(seed)
        W = OrderedDict()
        for l in self.layers:                                # Keras creation order
            if l.name in self.convs:
                Lc = self.layout[l.name]
                k = Lc["k"]
                lim = math.sqrt(6.0 / (k ** self.nd * (Lc["cin"] + Lc["cout"])))
                W[l.name + "/kernel"] = rs.uniform(-lim, lim, size=(k,) * self.nd + (Lc["cin"], Lc["cout"])).astype(np.float32)
                W[l.name + "/bias"] = np.zeros(Lc["cout"], np.float32)
            elif l.name in self.norms:
                W[l.name + "/gamma"] = np.ones(self.layout[l.name]["c"], np.float32)
                W[l.name + "/beta"] = np.zeros(self.layout[l.name]["c"], np.float32)
            elif l.name in self.denses:
                Lc = self.layout[l.name]
                lim = math.sqrt(6.0 / (Lc["K"] + Lc["M"]))
                W[l.name + "/kernel"] = rs.uniform(-lim, lim, size=(Lc["K"], Lc["M"])).astype(np.float32)
                W[l.name + "/bias"] = np.zeros(Lc["M"], np.float32)
        self.load_keras_weights(W)

    def keras_to_flat(self, W):
        host = np.zeros(self.n_flat, np.float32)
        for name, Lc in self.layout.items():
            if Lc["kind"] == "conv":
                k = np.asarray(W[name + "/kernel"], np.float32)
                assert k.shape == (Lc["k"],) * self.nd + (Lc["cin"], Lc["cout"]), (name, k.shape)
                if self.nd == 2:                          # (kh,kw,Cin,Cout) -> centre kd plane of the k^3 image (k = 1: the single tap)
                    k3 = np.zeros((Lc["k"],) * 3 + (Lc["cin"], Lc["cout"]), np.float32)
                    k3[Lc["k"] // 2] = k
                    k = k3
                o, n = Lc["w"]
                host[o:o + n] = k.transpose(0, 1, 2, 4, 3).reshape(-1)
                ob, nb = Lc["b"]
                host[ob:ob + nb] = np.asarray(W[name + "/bias"], np.float32)
            elif Lc["kind"] == "dense":
                k = np.asarray(W[name + "/kernel"], np.float32)
                assert k.shape == (Lc["K"], Lc["M"]), (name, k.shape)
                o, n = Lc["w"]
                host[o:o + n] = k.reshape(-1)
                ob, nb = Lc["b"]
                host[ob:ob + nb] = np.asarray(W[name + "/bias"], np.float32)
            else:
                for key in ("gamma", "beta"):
                    o, n = Lc[key]
                    host[o:o + n] = np.asarray(W[name + "/" + key], np.float32)
        return host

    def load_keras_weights(self, W):
        self.P.copy_(torch.from_numpy(self.keras_to_flat(W)))
        self.refresh_weight_copies()

    def flat_to_keras(self, host, moving=True):
        W = OrderedDict()
        for l in self.layers:
            name = l.name
            if name in self.convs:
                Lc = self.layout[name]
                o, n = Lc["w"]
                k = host[o:o + n].reshape((Lc["k"],) * 3 + (Lc["cout"], Lc["cin"])).transpose(0, 1, 2, 4, 3)
                W[name + "/kernel"] = (k[Lc["k"] // 2] if self.nd == 2 else k).copy()
                ob, nb = Lc["b"]
                W[name + "/bias"] = host[ob:ob + nb].copy()
            elif name in self.norms:
                for key in ("gamma", "beta"):
                    o, n = self.layout[name][key]
                    W[name + "/" + key] = host[o:o + n].copy()
            elif name in self.denses:
                Lc = self.layout[name]
                o, n = Lc["w"]
                W[name + "/kernel"] = host[o:o + n].reshape(Lc["K"], Lc["M"]).copy()
                ob, nb = Lc["b"]
                W[name + "/bias"] = host[ob:ob + nb].copy()
        return W

    def export_keras_weights(self):
        return self.flat_to_keras(self.P.detach().cpu().numpy())

    def refresh_weight_copies(self):
        if self.pad:
            self.Pp.index_copy_(0, self.map_p, self.P)       # every logical parameter into its place in the padded fp32 images (one kernel)
            for name, op in self.convs.items():
                if name in self.Wup:
                    W = self.Wup[name]
                    ops.conv3d_pack_up_weights(self.Wp32[name], self.Wp32[name].shape[2], 0, W["up_f"], W["up_d"], None, None)
                    continue                                  # forward and input gradient use the parity filters only
                if name in self.Ws2:
                    W = self.Ws2[name]
                    self._s2.pack(self.Wp32[name], W["fwd"], W["dgrad"])
                    if W["wgrad"]:
                        continue                              # no stride-1 image is read any more
                    ops.pack_weights(self.Wp32[name], self.Wf[name], None)      # (the stride-1 weight-gradient path reads neither image; kept for load/save symmetry)
                    continue
                ops.pack_weights(self.Wp32[name], self.Wf[name], self.Wd.get(name))
            return
        for name, op in self.convs.items():
            if op["k"] == 3 and op["s"] == 1:
                ops.pack_weights(self.w_view(name), self.Wf[name], self.Wd.get(name))
            else:
                ops.cast(self.w_view(name), self.Wf[name])

    # ------------------------------------------------------------------------------------------------ buffers
    def set_batch(self, N):
        if N not in self._bufsets:
            self.N = N
            T, pre, stats = {}, {}, {}
            for o in self.ops:
                C, sp = o["shape"][0], tuple(o["shape"][1:])
                if o["out"] in self.flat:                   # GlobalAveragePooling / Dense outputs: fp32 [samples][C]
                    T[o["out"]] = torch.zeros((N, C), dtype=torch.float32, device=self.dev)
                    continue
                T[o["out"]] = torch.empty(self._lead(N) + sp + (C,), dtype=self.dtype, device=self.dev)
                if o["kind"] == "norm":
                    stats[o["name"]] = torch.zeros((N if o["instance"] else 1, C, 3), dtype=torch.float32, device=self.dev)
            Gd, tmp = {}, {}
            if self.training:
                for name, t in T.items():
                    Gd[name] = torch.empty_like(t)
                if self.input_grad:
                    Gd[self.input_name] = torch.empty(self._lead(N) + tuple(self.shape[self.input_name][1:]) + (self.shape[self.input_name][0],),
                                                      dtype=self.dtype, device=self.dev)
            full, gfull = {}, {}
            for o in self.ops:                            # strided convs run as `sub`: the stride-1 result at the input resolution (+ gradient)
                if o["kind"] == "conv" and o["sub"] is not None:
                    s2 = self.Ws2.get(o["name"])
                    if s2 is not None and (s2["wgrad"] or not self.training):
                        continue                          # stride 2 on the parity kernels: no stride-1 temporaries
                    sp_in = tuple(self.shape[o["ins"][0]][1:])
                    if s2 is None:
                        full[o["name"]] = torch.empty(self._lead(N) + sp_in + (o["shape"][0],), dtype=self.dtype, device=self.dev)
                    if self.training:
                        gfull[o["name"]] = torch.zeros(self._lead(N) + sp_in + (o["shape"][0],), dtype=self.dtype, device=self.dev)
            cmax = max([self.shape[n][0] for n in self.norms] + [1])
            nvox = N * int(np.prod(self.plan.spatial)) if self.head == "seg" else N
            Lb = self.plan.n_labels
            self._bufsets[N] = dict(T=T, G=Gd, tmp=tmp, stats=stats, ws=torch.zeros((N, cmax, 2), dtype=torch.float64, device=self.dev),
                                    logits=torch.empty((nvox, Lb), dtype=torch.float32, device=self.dev),
                                    probs=torch.empty((nvox, Lb), dtype=torch.float32, device=self.dev),
                                    dlogits=torch.empty((nvox, Lb), dtype=torch.float32, device=self.dev),
                                    dummy_y=torch.zeros(nvox * Lb, dtype=torch.uint8, device=self.dev), drop={}, cat={}, full=full, gfull=gfull,
                                    dense_in={})
        b = self._bufsets[N]
        self.N, self.T, self.Gt, self.tmp, self.stats, self.norm_ws = N, b["T"], b["G"], b["tmp"], b["stats"], b["ws"]
        self.logits, self.probs, self.dlogits, self._dummy_y, self.drop, self.cat = b["logits"], b["probs"], b["dlogits"], b["dummy_y"], b["drop"], b["cat"]
        self.full, self.gfull, self.dense_in = b["full"], b["gfull"], b["dense_in"]

    def _t(self, name):
        return self.x_in if name == self.input_name else self.T[name]

    def _lead(self, N):
        """leading dims of an activation: 3-D (N,) + (D,H,W); 2-D planar (1, N) + (H,W) - the slices are the kernels' D axis"""
        return (1, N) if self.planar else (N,)

    def _smp(self, t):
        """[samples][voxels...][C] view for the per-sample kernels (instance norm, spatial dropout): 2-D slices are the samples"""
        return t.reshape(tuple(t.shape[1:])) if self.planar else t

    def _sub(self, t, strides):
        """the voxels of a stride-1 'same' result that a strided 'same' convolution computes.  TF pads pad_before = 0 for an even size
        (output o = the stride-1 result at 2o + 1) and 1 for an odd one (at 2o); axes with stride 1 are kept whole."""
        first = 2 if self.planar else 1
        sl = [slice(None)] * t.dim()
        for a, st in enumerate(strides):
            if st == 2:
                sl[first + a] = slice(1 if t.shape[first + a] % 2 == 0 else 0, None, 2)
        return t[tuple(sl)]

    # ------------------------------------------------------------------------------------------------ forward
    def forward(self, x, bn_training=None):
        training = self.training if bn_training is None else bn_training
        assert tuple(x.shape) == self._lead(self.N) + self.plan.spatial + (self.shape[self.input_name][0],), x.shape
        self.x_in = x
        for o in self.ops:
            kind = o["kind"]
            out = self.T[o["out"]]
            if kind == "conv":
                name = o["name"]
                if name in self.Ws2:
                    # stride 2 on the parity kernels: the gather launch over the input, the fp32 master bias in its accumulators
                    ops.conv3d_stride2_fwd(self._t(o["ins"][0]), self.Ws2[name]["fwd"], self.bp[name], out)
                    continue
                dst = out if o["sub"] is None else self.full[name]
                if self.pad:
                    s0 = self._t(o["ins"][0])
                    s1 = self._t(o["ins"][1]) if len(o["ins"]) > 1 else None
                    if name in self.Wup:
                        ops.conv3d_upcat_fwd(s0, None, self.Wup[name]["up_f"], None, self.bp[name], out, act=o["act"], alpha=LEAKY_ALPHA)
                    else:
                        ops.conv3d_fwd(s0, s1, self.Wf[name], self.bp[name], dst, up0=o["up0"], act=o["act"], planar=self.planar)
                elif o["k"] == 3 and o["s"] == 1:
                    s0 = self._t(o["ins"][0])
                    s1 = self._t(o["ins"][1]) if len(o["ins"]) > 1 else None
                    ops.conv3d_fwd(s0, s1, self.Wf[name], self._v(name, "b"), dst, up0=o["up0"], act=o["act"], planar=self.planar)
                else:
                    ops.conv_direct_fwd(self._t(o["ins"][0]), self.Wf[name], self._v(name, "b"), out, o["k"], o["s"], act=o["act"],
                                        planar=self.planar)
                if o["sub"] is not None:
                    out.copy_(self._sub(dst, o["sub"]))
            elif kind == "norm":
                name = o["name"]
                gam, bet = (self.gp[name], self.betap[name]) if self.pad else (self._v(name, "gamma"), self._v(name, "beta"))
                ops.norm_act_fwd(self._smp(self._t(o["ins"][0])), gam, bet, self._smp(out), self.stats[name], self.norm_ws,
                                 1 if o["instance"] else 0, eps=1e-3, eps_on_std=o["instance"], act=o["act"], alpha=LEAKY_ALPHA)
            elif kind == "add":
                ops.add(self._t(o["ins"][0]), self._t(o["ins"][1]), out)
            elif kind == "dropout":
                src = self._t(o["ins"][0])
                if training and o["rate"] > 0:
                    keep = 1.0 - o["rate"]
                    if self._fixed_drop is not None:
                        sc = self._fixed_drop[o["out"]]
                        if sc.shape[1] != src.shape[-1]:
                            sc = torch.nn.functional.pad(sc, (0, src.shape[-1] - sc.shape[1]))
                    else:                                   # whole channels of a sample are dropped, survivors scaled by 1/(1-p)
                        sc = (torch.rand((self.N, src.shape[-1]), device=self.dev) < keep).float() / keep
                    self.drop[o["out"]] = sc
                    ops.channel_scale(self._smp(src), sc, self._smp(out))
                else:
                    self.drop[o["out"]] = None
                    ops.cast(src, out)                      # identity copy keeps the tensor table simple
            elif kind == "upsample":
                ops.upsample_fwd(self._t(o["ins"][0]), out, planar=self.planar)
            elif kind == "maxpool":
                ops.maxpool_fwd(self._t(o["ins"][0]), out, planar=self.planar)
            elif kind == "avgpool":
                ops.avgpool_fwd(self._t(o["ins"][0]), out, planar=self.planar)
            elif kind == "gap":
                ops.global_avgpool_fwd(self._smp(self._t(o["ins"][0])), out)
            elif kind == "dense":
                name = o["name"]
                ops.dense_fwd(self._dense_x(o), self._v(name, "w").view(self.layout[name]["K"], o["units"]), self._v(name, "b"), out,
                              act=o["act"], alpha=LEAKY_ALPHA)
        src = self.T[self.logits_src]
        if src.shape[-1] != self.plan.n_labels:                # channel-padded: the logits are the first n_labels channels
            self.logits.copy_(src.reshape(-1, src.shape[-1])[:, :self.plan.n_labels])
        else:
            ops.cast(src.reshape(-1), self.logits.reshape(-1))
        return self.logits

    def _dense_x(self, o):
        """the Dense layer's input [N, K] fp32: its source tensor, or the logical channels of a channel-padded one"""
        src, K = self.T[o["ins"][0]], self.layout[o["name"]]["K"]
        if src.shape[1] == K:
            return src
        buf = self.dense_in.get(o["name"])
        if buf is None:
            buf = self.dense_in[o["name"]] = torch.empty((src.shape[0], K), dtype=torch.float32, device=self.dev)
        buf.copy_(src[:, :K])
        return buf

    def set_dropout_masks(self, masks):
        """testing hook: fix the SpatialDropout3D masks ({layer name: [N,C] fp32 tensor}) instead of drawing them"""
        self._fixed_drop = masks

    def loss_forward(self, y_true, weight=None):
        self.sums.zero_()
        ops.sigmoid_dice_fwd(self.logits, y_true, self.probs, self.sums, weight=weight)
        if self.loss_kind == ops.LOSS_WEIGHTED_DICE:
            ns, nl = self._wdice_groups()
            if getattr(self, "_gsums", None) is None or self._gsums.numel() < 3 * ns * nl:
                self._gsums = torch.zeros(3 * ns * nl, dtype=torch.float64, device=self.dev)
            ops.weighted_dice_fwd(self.probs, y_true, self._gsums, self.sums, ns, nl)
        if self.dist is not None and self.dist.world > 1 and self.dist.global_dice:
            self.dist.all_reduce_sums(self.sums)
        return self.sums

    def _wdice_groups(self):
        """(groups along the batch axis, labels per group) of weighted_dice_coefficient's axis=(-3,-2,-1) (reference metrics.py:39): the 3-D
        models' (N, labels, X, Y, Z) tensors give one Dice per (sample, label), the 2-D models' (N, X, Y, labels) one per slice"""
        if self.plan.ndim == 2:
            return self.N, 1
        return self.N, self.plan.n_labels

    def bce_forward(self, target):
        """dense head: probs = sigmoid(logits); sums = [sum of binary cross-entropy terms, sum |p - t|, n] (loss = [0] / [2], mae = [1] / [2])"""
        assert self.head == "dense"
        self.sums.zero_()
        ops.sigmoid_bce_fwd(self.logits.reshape(-1), target.reshape(-1), self.probs.reshape(-1), self.sums)
        return self.sums

    def predict(self, x):
        self.forward(x, bn_training=False)
        self.sums.zero_()
        ops.sigmoid_dice_fwd(self.logits, self._dummy_y, self.probs, self.sums)
        return self.probs

    # ------------------------------------------------------------------------------------------------ backward
    def _accum(self, name, write):
        """route a gradient contribution for tensor `name`: write(dst) fills dst; the first contribution writes G directly"""
        if name == self.input_name and not self.input_grad:
            return
        if name not in self._has_grad:
            write(self.Gt[name])
            self._has_grad.add(name)
        else:
            if name not in self.tmp:
                self.tmp[name] = torch.empty_like(self.Gt[name])
            write(self.tmp[name])
            ops.add(self.Gt[name], self.tmp[name], self.Gt[name])

    def grad_streams(self):
        return [st for st in (getattr(self, "_main_stream", None), self._wg_stream) if st is not None]

    def backward(self, y_true, grad_scale=1.0, weight=None, dprobs=None, dprobs_scale=1.0, seg_loss=True, params=True):
        """seg head: y_true = uint8 labels; `dprobs` (optional, [..., ld >= n_labels]) is an extra gradient that arrives on the probabilities
        (the adversarial term of reference train_adv.py:177-180), `seg_loss=False` leaves only that term (train_semi.py:176-183).
        dense head: y_true = float targets, the loss is their mean binary cross-entropy times grad_scale.
        params=False: only the input gradient is wanted (the frozen discriminator inside the combined model)."""
        self._main_stream = torch.cuda.current_stream(self.dev) if self.dev.type == "cuda" else None
        if self.pad:
            if params:
                self.Gp.zero_()                               # every padded gradient image at once; G itself is overwritten by the final gather
        else:
            self.G.zero_()
        self._has_grad = set()
        self._params = params
        if self.dist is not None and params:
            self.dist.begin()
        if self.head == "dense":
            ops.sigmoid_bce_bwd(self.probs.reshape(-1), y_true.reshape(-1), self.dlogits.reshape(-1), grad_scale / self.probs.numel())
        else:
            if seg_loss and self.loss_kind == ops.LOSS_WEIGHTED_DICE:
                ns, nl = self._wdice_groups()
                ops.weighted_dice_bwd(self.probs, y_true, self._gsums, self.sums, self.dlogits, ns, nl, grad_scale=grad_scale)
            elif seg_loss:
                ops.sigmoid_loss_bwd(self.probs, y_true, self.sums, self.dlogits, self.loss_kind, self.loss_param, smooth=1.0, grad_scale=grad_scale,
                                     weight=weight)
            if dprobs is not None:
                ops.sigmoid_chain(self.probs, dprobs, self.dlogits, scale=dprobs_scale, accumulate=seg_loss)
        gsrc = self.Gt[self.logits_src]
        if gsrc.shape[-1] != self.plan.n_labels:
            gsrc.zero_()
            gsrc.reshape(-1, gsrc.shape[-1])[:, :self.plan.n_labels] = self.dlogits
        else:
            ops.cast(self.dlogits.reshape(-1), gsrc.reshape(-1))
        self._has_grad.add(self.logits_src)
        for o in reversed(self.ops):
            kind, out = o["kind"], o["out"]
            if out not in self._has_grad:
                continue                                      # dead branch (no consumer reaches the loss)
            g = self.Gt[out]
            if kind == "conv":
                name = o["name"]
                if o["act"] != ACT_NONE:
                    ops.act_bwd(self.T[out], g, g, o["act"], LEAKY_ALPHA)
                dw, db = self.w_view(name, self.G), self._v(name, "b", self.G)
                ins = o["ins"]
                s2 = self.Ws2.get(name)
                g_out = g                                    # gradient at the conv's own (strided) output
                if o["sub"] is not None and (s2 is None or not s2["wgrad"]):
                    # gradient of "sample every second voxel": scatter into the stride-1 grid
                    gf = self.gfull[name]
                    self._sub(gf, o["sub"]).copy_(g)             # the other voxels stay zero (zeroed at allocation, never written)
                    g = gf
                if self.pad or (o["k"] == 3 and o["s"] == 1):
                    s0 = self._t(ins[0])
                    s1 = self._t(ins[1]) if len(ins) > 1 else None

                    def wgrad(o=o, name=name, g=g, s0=s0, s1=s1, dw=dw, db=db, s2=s2, g_out=g_out):
                        if s2 is not None and s2["wgrad"]:
                            # stride 2 on the parity kernels: the 64 slot gradients of (low-res source = dy, full-resolution tensor = x), 27 read back
                            coutp, cinp = g_out.shape[-1], s0.shape[-1]
                            ops.conv3d_upcat_wgrad(g_out, None, s0, s2["dw27"], s2["db"], self.dwc_scratch)
                            self.dWp[name][:, :coutp, :].add_(self._s2.unpack_wgrad(self.dwc_scratch, coutp, cinp))
                            self.dbp[name][:coutp].add_(g_out.reshape(-1, coutp).sum(0, dtype=torch.float32))
                            return
                        if self.pad:
                            dwp, dbp = self.dWp[name], self.dbp[name]          # views of Gp (zeroed once per backward, gathered at the end)
                            gw = g
                            if g.shape[-1] % 64 and not self._c32.get(name, False):
                                # the per-kd MFMA weight-gradient kernel tiles Cout by 64: hand it a zero-extended copy of dy (the extra rows of
                                # dw stay 0) - unless the kd-sharing kernel (32 x 32 blocks) takes the launch as it is
                                gw = self._dy64(g)
                            if name in self.Wup and self.Wup[name]["wgrad"] and gw is g:
                                ops.conv3d_upcat_wgrad(s0, None, g, dwp, dbp, self.dwc_scratch)
                            else:
                                ops.conv3d_wgrad(s0, s1, gw, dwp, dbp, up0=o["up0"], planar=self.planar)
                        else:
                            ops.conv3d_wgrad(s0, s1, g, dw, db, up0=o["up0"], planar=self.planar)

                    if not params:
                        pass
                    elif self._wg_stream is None:
                        wgrad()
                    else:          # weight gradients beside the input-gradient chain (see UNetEngine._block_bwd)
                        self._wg_stream.wait_stream(torch.cuda.current_stream(self.dev))
                        with torch.cuda.stream(self._wg_stream):
                            wgrad()
                    if name in self.Wup and self.training:
                        self._accum(ins[0], lambda dst: ops.conv3d_upcat_dgrad(g, self.Wup[name]["up_d"], None, None, None, dst, None))
                    elif s2 is not None:
                        self._accum(ins[0], lambda dst: ops.conv3d_upcat_fwd(g_out, None, s2["dgrad"], None, None, dst, act=ACT_NONE))
                    elif name in self.Wd:
                        if len(ins) == 1 and not o["up0"]:
                            self._accum(ins[0], lambda dst: ops.conv3d_dgrad(g, self.Wd[name], dst, planar=self.planar))
                        else:
                            cin = self.Wd[name].shape[1]           # physical (channel-padded) width of the concatenated input
                            if name not in self.cat:
                                self.cat[name] = torch.empty(tuple(g.shape[:-1]) + (cin,), dtype=self.dtype, device=self.dev)
                            cat = self.cat[name]
                            ops.conv3d_dgrad(g, self.Wd[name], cat, planar=self.planar)
                            c0 = self.shape[ins[0]][0]
                            if o["up0"]:
                                self._accum(ins[0], lambda dst: ops.upsample_bwd(cat, dst, dy_off=0, planar=self.planar))
                            else:
                                self._slice_into(ins[0], cat, 0)
                            if len(ins) > 1:
                                self._slice_into(ins[1], cat, c0)
                else:
                    x = self._t(ins[0])
                    if params:
                        ops.conv_direct_bwd(x, self.Wf[name], g, None, dw, db, o["k"], o["s"], planar=self.planar)
                    if ins[0] != self.input_name or self.input_grad:
                        self._accum(ins[0], lambda dst: ops.conv_direct_bwd(x, self.Wf[name], g, dst, None, None, o["k"], o["s"], planar=self.planar))
            elif kind == "norm":
                name = o["name"]
                src = o["ins"][0]
                if self.pad:
                    dg, dbt = self.dgp[name], self.dbetap[name]
                    self._accum(src, lambda dst: ops.norm_act_bwd(self._smp(self._t(src)), None, self._smp(g), self.gp[name],
                                                                  self.stats[name], self._smp(dst), dg, dbt, self.norm_ws,
                                                                  1 if o["instance"] else 0, act=o["act"], alpha=LEAKY_ALPHA, beta=self.betap[name]))
                    continue
                self._accum(src, lambda dst: ops.norm_act_bwd(self._smp(self._t(src)), None, self._smp(g), self._v(name, "gamma"),
                                                              self.stats[name], self._smp(dst), self._v(name, "gamma", self.G),
                                                              self._v(name, "beta", self.G), self.norm_ws,
                                                              1 if o["instance"] else 0, act=o["act"], alpha=LEAKY_ALPHA, beta=self._v(name, "beta")))
            elif kind == "add":
                for i in o["ins"]:
                    self._slice_into(i, g, 0)
            elif kind == "dropout":
                sc = self.drop.get(out)
                if sc is None:
                    self._slice_into(o["ins"][0], g, 0)
                else:
                    self._accum(o["ins"][0], lambda dst: ops.channel_scale(self._smp(g), sc, self._smp(dst)))
            elif kind == "upsample":
                self._accum(o["ins"][0], lambda dst: ops.upsample_bwd(g, dst, dy_off=0, planar=self.planar))
            elif kind == "maxpool":
                src = o["ins"][0]
                self._accum(src, lambda dst: ops.maxpool_bwd(self._t(src), g, dst, relu_mask=False, planar=self.planar))
            elif kind == "avgpool":
                self._accum(o["ins"][0], lambda dst: ops.avgpool_bwd(g, dst, planar=self.planar))
            elif kind == "gap":
                self._accum(o["ins"][0], lambda dst: ops.global_avgpool_bwd(g, self._smp(dst)))
            elif kind == "dense":
                name = o["name"]
                Lc = self.layout[name]
                x, w = self._dense_x(o), self._v(name, "w").view(Lc["K"], Lc["M"])
                if self.pad:
                    dw, db = self.dDense[name] if params else (None, None)
                else:
                    dw = self._v(name, "w", self.G).view(Lc["K"], Lc["M"]) if params else None
                    db = self._v(name, "b", self.G) if params else None

                def write(dst, o=o, x=x, w=w, dw=dw, db=db, g=g):
                    if dst.shape[1] == x.shape[1]:
                        ops.dense_bwd(x, w, self.T[o["out"]], g, dst, dw, db, act=o["act"], alpha=LEAKY_ALPHA)
                    else:                                       # channel-padded source: the padding receives a zero gradient
                        dx = torch.empty_like(x)
                        ops.dense_bwd(x, w, self.T[o["out"]], g, dx, dw, db, act=o["act"], alpha=LEAKY_ALPHA)
                        dst.zero_()
                        dst[:, :x.shape[1]] = dx
                self._accum(o["ins"][0], write)
        if self._wg_stream is not None:
            torch.cuda.current_stream(self.dev).wait_stream(self._wg_stream)
        if self.pad and params:
            torch.index_select(self.Gp, 0, self.map_g, out=self.G)       # the logical gradients out of the padded images (one kernel)
        if self.dist is not None and params:
            self.dist.finish(self)

    def input_gradient(self):
        """dL/d(input) of the last backward pass (engines built with input_grad=True): same layout as the input"""
        return self.Gt[self.input_name]

    def _dy64(self, g):
        C = g.shape[-1]
        c64 = ((C + 63) // 64) * 64
        key = tuple(g.shape[:-1]) + (c64,)
        buf = self._dy64_buf.get(key)
        if buf is None:
            buf = self._dy64_buf[key] = torch.zeros(key, dtype=g.dtype, device=g.device)
        buf[..., :C] = g
        return buf

    def _slice_into(self, name, src, off):
        if name == self.input_name and not self.input_grad:
            return
        first = name not in self._has_grad
        ops.slice_channels(src, off, self.Gt[name], accumulate=not first)
        self._has_grad.add(name)

    # ------------------------------------------------------------------------------------------------ optimizer
    def adam_step(self, lr, beta1=None, beta2=0.999, eps=1e-7, grad_scale=1.0):
        beta1 = self.beta1 if beta1 is None else beta1
        self.t += 1
        lr_t = lr * math.sqrt(1.0 - beta2 ** self.t) / (1.0 - beta1 ** self.t)
        ops.adam_step(self.P, self.G, self.M, self.V, lr_t, beta1, beta2, eps, grad_scale)
        self.refresh_weight_copies()

    def train_step(self, x, y_true, lr, weight=None):
        self.forward(x)
        self.loss_forward(y_true, weight)
        self.backward(y_true, grad_scale=(self.dist.grad_scale if getattr(self, "dist", None) is not None else 1.0), weight=weight)
        self.adam_step(lr)
        return self.sums

    @staticmethod
    def metrics_from_sums(s, smooth=1.0, loss_kind=0, loss_param=1.0):
        from .engine import UNetEngine
        return UNetEngine.metrics_from_sums(s, smooth, loss_kind, loss_param)
