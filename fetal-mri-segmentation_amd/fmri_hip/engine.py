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
import os
from collections import OrderedDict

import numpy as np
import torch

from . import ops
from ._lib import lib, ACT_NONE, ACT_RELU, IMPL_AUTO


class UNetPlan:
    """Topology in Keras creation order (names match Keras auto-naming)."""

    def __init__(self, in_channels, spatial, depth=4, n_base_filters=32, n_labels=1, ndim=3, norm=None, deconvolution=False):
        """norm: None | 'batch' | 'instance' (reference create_convolution_block options); deconvolution: Conv*DTranspose(k=2,s=2)
        instead of nearest up-sampling (reference get_up_convolution)."""
        self.in_channels, self.spatial = in_channels, tuple(spatial)
        self.depth, self.n_base_filters, self.n_labels, self.ndim = depth, n_base_filters, n_labels, ndim
        self.norm, self.deconvolution = norm, deconvolution
        d = "3d" if ndim == 3 else "2d"
        k = 0
        nk = 0
        tk = 0

        def conv_name():
            nonlocal k
            k += 1
            return "conv%s_%d" % (d, k)

        def norm_name():
            nonlocal nk
            nk += 1
            return "%s_%d" % ("batch_normalization" if norm == "batch" else "instance_normalization", nk)

        self.enc = []
        cin = in_channels
        for ld in range(depth):
            lv = []
            for mult in (1, 2):
                cout = n_base_filters * (2 ** ld) * mult
                lv.append(dict(name=conv_name(), cin=cin, cout=cout, level=ld, norm=norm_name() if norm else None))
                cin = cout
            self.enc.append(lv)
        self.dec = []
        self.up = {}          # decoder level -> transposed-conv descriptor (only with deconvolution=True)
        for ld in range(depth - 2, -1, -1):
            skip_c = self.enc[ld][1]["cout"]
            if deconvolution:
                tk += 1
                self.up[ld] = dict(name="conv%s_transpose_%d" % (d, tk), cin=cin, cout=cin, level=ld)
            a = dict(name=conv_name(), cin=cin + skip_c, cout=skip_c, level=ld, c_up=cin, c_skip=skip_c,
                     norm=norm_name() if norm else None)
            b = dict(name=conv_name(), cin=skip_c, cout=skip_c, level=ld, norm=norm_name() if norm else None)
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
        self._pack_stream, self._pack_events, self._pack_event_dec, self._pack_pending = None, None, None, []
        # FMRI_WGRAD_PRIO (A/B): HIP priority of the weight-gradient stream (0 = default, positive = lower): with a lower one the dispatcher hands free
        # CUs to the input-gradient chain first and the weight-gradient backlog fills in behind the HBM-bound kernels of the main stream
        self._wg_stream = (torch.cuda.Stream(device=self.dev, priority=int(os.environ.get("FMRI_WGRAD_PRIO", "0")))
                           if (training and self.dev.type == "cuda" and os.environ.get("FMRI_WGRAD_STREAM", "1") != "0") else None)
        self.loss_kind, self.loss_param = 0, 1.0      # ops.LOSS_KINDS: 0 = dice_coefficient_loss
        # FMRI_DETERMINISTIC=1: bit-reproducible training steps (same weights + same batch -> the same bits in every gradient and metric):
        # gradient partial sums meet as fixed-point integers in a shadow of G (ops.set_deterministic), the parity-form weight gradient -
        # whose scratch is filled by fp32 atomics - gives way to the fused-upsample 27-tap one.  One engine per process at a time.
        self.deterministic = bool(training and self.dev.type == "cuda" and os.environ.get("FMRI_DETERMINISTIC", "0") == "1")
        self._bufsets = {}
        self._build_params(seed)
        self.set_batch(batch)

    # ------------------------------------------------------------------------------------------------ parameters
    def _build_params(self, seed):
        p = self.plan
        # backward-completion order: final, dec (shallowest level first: b, a, transposed conv), enc (deepest first, b before a)
        order = [("final", p.final)]
        for lv in reversed(p.dec):
            order += [("conv", lv[1]), ("conv", lv[0])]
            if lv[0]["level"] in p.up:
                order.append(("deconv", p.up[lv[0]["level"]]))
        for lv in reversed(p.enc):
            order += [("conv", lv[1]), ("conv", lv[0])]
        self.layout = OrderedDict()
        off = 0
        ntap = {"conv": 27, "final": 1, "deconv": 8}
        for kind, c in order:
            nw = ntap[kind] * c["cout"] * c["cin"]
            off = (off + 3) & ~3
            L = dict(kind=kind, w=(off, nw), b=(((off + nw + 3) & ~3), c["cout"]), cin=c["cin"], cout=c["cout"], norm=c.get("norm"))
            off = ((off + nw + 3) & ~3) + c["cout"]
            if c.get("norm"):          # gamma, beta of the block's normalisation layer right behind the bias
                off = (off + 3) & ~3
                L["gamma"] = (off, c["cout"])
                L["beta"] = (off + ((c["cout"] + 3) & ~3), c["cout"])
                off = L["beta"][0] + c["cout"]
            self.layout[c["name"]] = L
        self.n_flat = (off + 3) & ~3
        dev = self.dev
        self.P = torch.zeros(self.n_flat, dtype=torch.float32, device=dev)
        if self.training:
            self.G = torch.zeros_like(self.P)
            self.M = torch.zeros_like(self.P)
            self.V = torch.zeros_like(self.P)
            if self.deterministic:
                if p.norm is not None or p.deconvolution:
                    raise NotImplementedError("FMRI_DETERMINISTIC=1 covers the plain unet_model_3d / unet_model_2d step (no normalisation layers, "
                                              "UpSampling up-convolution): the normalisation statistics and the folded transposed conv still "
                                              "accumulate with floating-point atomics")
                self.G64 = torch.zeros(self.n_flat, dtype=torch.int64, device=dev)
                _register_deterministic(self)
        # compute-dtype copies of the 3x3x3 filters
        self.Wf, self.Wd, self.Wup = {}, {}, {}
        self.upcat = self._upcat_layers()
        self.Wt = {}                      # compute-dtype copies of the transposed-conv filters [8][Cout][Cin]
        self.moving = {}                  # batch-norm moving mean / variance (inference statistics), fp32 [2][C]
        first = p.enc[0][0]["name"]
        for name, L in self.layout.items():
            if L["kind"] == "deconv":
                self.Wt[name] = torch.empty((8, L["cout"], L["cin"]), dtype=self.dtype, device=dev)
            if L.get("norm") and p.norm == "batch":
                mv = torch.zeros((2, L["cout"]), dtype=torch.float32, device=dev)
                mv[1].fill_(1.0)
                self.moving[name] = mv
            if L["kind"] != "conv":
                continue
            if name in self.upcat:
                # up-sample + concat + conv in parity form (fmri_conv3d_upcat_*): pre-summed 2x2x2 filters per output parity for the
                # up-sampled channels, plain 27-tap filters for the skip channels
                c0, c1 = self.upcat[name]
                npar = 4 if self.planar else 8           # parity classes = pre-summed taps per class (2-D: (ph,pw) x 2x2 taps)
                W = dict(up_f=torch.empty((npar, npar, L["cout"], c0), dtype=self.dtype, device=dev),
                         sk_f=torch.empty((27, L["cout"], c1), dtype=self.dtype, device=dev), up_d=None, sk_d=None)
                if self.training:
                    W["up_d"] = torch.empty((npar, npar, c0, L["cout"]), dtype=self.dtype, device=dev)
                    W["sk_d"] = torch.empty((27, c1, L["cout"]), dtype=self.dtype, device=dev)
                self.Wup[name] = W
                if not self.planar:
                    continue                             # 2-D keeps the 9-tap images too: batches whose slice count does not tile fall back
            self.Wf[name] = torch.empty((27, L["cout"], L["cin"]), dtype=self.dtype, device=dev)
            if self.training and name != first:
                self.Wd[name] = torch.empty((27, L["cin"], L["cout"]), dtype=self.dtype, device=dev)
        # Deconvolution3D(k = 2, s = 2) (reference unet.py:135): output voxel 2g+p = W[p] . x[g] + b, i.e. the parity form with ONE
        # non-zero tap per parity class (low-res offset 0) - it rides the same MFMA kernels (fmri_conv3d_upcat_* with C1 = 0)
        self.Wdc = {}
        if not self.planar and self.dtype == torch.bfloat16:
            for lvl, u in p.up.items():
                L = self.layout[u["name"]]
                D, H, W = p.level_dims(lvl)
                if ops.conv3d_upcat_ok(L["cin"], 0, L["cout"], D, H, W, self.dtype) == 3:
                    Wd_ = dict(up_f=torch.zeros((8, 8, L["cout"], L["cin"]), dtype=self.dtype, device=dev), up_d=None, dw27=None)
                    if self.training:
                        Wd_["up_d"] = torch.zeros((8, 8, L["cin"], L["cout"]), dtype=self.dtype, device=dev)
                        Wd_["dw27"] = torch.zeros((27, L["cout"], L["cin"]), dtype=torch.float32, device=dev)   # never read
                    self.Wdc[u["name"]] = Wd_
        # 2-D twin (Deconvolution2D, reference unet/unet.py get_up_convolution): the 4 taps are 4 independent 1x1 convs, run as ONE planar
        # 3x3 MFMA conv to 4*Cout channels whose only non-zero tap is the centre one, followed by a depth-to-space copy
        self.Wd2 = {}
        if self.planar and self.dtype == torch.bfloat16:
            for lvl, u in p.up.items():
                L = self.layout[u["name"]]
                _, S, H, W = (1,) + p.level_dims(lvl + 1, 4)                  # low-res dims (the slice count does not matter here)
                if lib().fmri_conv3d_uses_mfma(L["cin"], 0, 4 * L["cout"], 4, H, W, 1) & 1:
                    Wd_ = dict(w32=torch.zeros((27, 4 * L["cout"], L["cin"]), dtype=torch.float32, device=dev),
                               wf=torch.empty((27, 4 * L["cout"], L["cin"]), dtype=self.dtype, device=dev), wd=None,
                               b4=torch.zeros(4 * L["cout"], dtype=torch.float32, device=dev))
                    if self.training:
                        Wd_["wd"] = torch.empty((27, L["cin"], 4 * L["cout"]), dtype=self.dtype, device=dev)
                        Wd_["dw"] = torch.zeros((27, 4 * L["cout"], L["cin"]), dtype=torch.float32, device=dev)
                        Wd_["db"] = torch.zeros(4 * L["cout"], dtype=torch.float32, device=dev)
                    self.Wd2[u["name"]] = Wd_
        # Deconvolution3D -> concatenate -> Conv3D FOLDED into one parity-form convolution of the low-res tensor (round 3; fmri_hip/deconv_fold.py,
        # fmri_conv3d_upcat_fwd_bias27): the transposed conv's output is never materialised and the decoder 'a' conv does 8 instead of 27 taps
        # on its up-sampled channels, exactly as in the UpSampling3D variant - with pre-MULTIPLIED instead of pre-summed filters.  Keyed by the
        # 'a' conv's name.  FMRI_DECONV_FOLD=0 keeps the two-step form (transposed conv as one-tap parity form, then the plain 27-tap conv).
        self.Wfd, self.fold = {}, None
        if (not self.planar and self.dtype == torch.bfloat16 and p.norm is None and os.environ.get("FMRI_DECONV_FOLD", "1") != "0"
                and os.environ.get("FMRI_FWD_WS", "1") != "0"):
            from .deconv_fold import DeconvFold
            for lv in p.dec:
                a = lv[0]
                if a["level"] not in p.up:
                    continue
                u = p.up[a["level"]]
                Lu = self.layout[u["name"]]
                D, H, W = p.level_dims(a["level"])
                if Lu["cout"] == a["c_up"] and ops.conv3d_upcat_ok(Lu["cin"], a["c_skip"], a["cout"], D, H, W, self.dtype) == 3:
                    if self.fold is None:
                        self.fold = DeconvFold(dev)
                    # weight GEMMs of the largest levels with bf16 operands (fp32 accumulation): 0.3 % relative error on filters that
                    # are rounded to bf16 for the MFMA kernels anyway, 2.3x faster at 256 x 512 x 512 (tools/bench_fold.py); below
                    # 2^25 multiply-adds per block fp32 is as fast
                    F = dict(u=u["name"], cmid=Lu["cout"], cin=Lu["cin"], cs=a["c_skip"],
                             gd=torch.bfloat16 if a["cout"] * Lu["cout"] * Lu["cin"] >= (1 << 25) else None,
                             up_f=torch.empty((8, 8, a["cout"], Lu["cin"]), dtype=self.dtype, device=dev),
                             sk_f=torch.empty((27, a["cout"], a["c_skip"]), dtype=self.dtype, device=dev), up_d=None, sk_d=None,
                             bias27=torch.zeros((27, a["cout"]), dtype=torch.float32, device=dev))
                    if self.training:
                        F["up_d"] = torch.empty((8, 8, Lu["cin"], a["cout"]), dtype=self.dtype, device=dev)
                        F["sk_d"] = torch.empty((27, a["c_skip"], a["cout"]), dtype=self.dtype, device=dev)
                        F["s27"] = torch.zeros((27, a["cout"]), dtype=torch.float32, device=dev)
                    self.Wfd[a["name"]] = F
        self._folded_up = set(F["u"] for F in self.Wfd.values())          # transposed convs that no longer run on their own
        self.dwc_scratch = None
        need = [64 * self.layout[n]["cout"] * self.upcat[n][0] for n in self.upcat_wgrad] + \
               [64 * self.layout[n]["cout"] * self.layout[n]["cin"] for n in self.Wdc] + \
               [64 * self.layout[n]["cout"] * F["cin"] for n, F in self.Wfd.items()]
        if self.training and need:
            self.dwc_scratch = torch.empty(max(need), dtype=torch.float32, device=dev)
        self._par8 = torch.arange(8, device=dev)
        self.init_glorot(seed)

    def w_view(self, name, buf=None):
        L = self.layout[name]
        buf = self.P if buf is None else buf
        o, n = L["w"]
        shape = {"conv": (27, L["cout"], L["cin"]), "deconv": (8, L["cout"], L["cin"]), "final": (L["cout"], L["cin"])}[L["kind"]]
        return buf[o:o + n].view(shape)

    def gb_view(self, name, which, buf=None):
        """gamma / beta of the normalisation layer of conv block `name`"""
        buf = self.P if buf is None else buf
        o, n = self.layout[name][which]
        return buf[o:o + n]

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
        k2 = (2, 2, 2) if self.plan.ndim == 3 else (2, 2)
        p = self.plan
        seq = [c for lv in p.enc for c in lv]
        for lv in p.dec:                                  # Keras creation order: the transposed conv precedes its level's blocks
            if lv[0]["level"] in p.up:
                seq.append(dict(p.up[lv[0]["level"]], transpose=True))
            seq += lv
        for c in seq:
            if c.get("transpose"):
                W[c["name"] + "/kernel"] = _glorot(rs, k2 + (c["cout"], c["cin"]))      # Keras: (k,k,k,Cout,Cin)
                W[c["name"] + "/bias"] = np.zeros(c["cout"], np.float32)
                continue
            W[c["name"] + "/kernel"] = _glorot(rs, k3 + (c["cin"], c["cout"]))
            W[c["name"] + "/bias"] = np.zeros(c["cout"], np.float32)
            if c.get("norm"):
                W[c["norm"] + "/gamma"] = np.ones(c["cout"], np.float32)
                W[c["norm"] + "/beta"] = np.zeros(c["cout"], np.float32)
        f = self.plan.final
        W[f["name"] + "/kernel"] = _glorot(rs, k1 + (f["cin"], f["cout"]))
        W[f["name"] + "/bias"] = np.zeros(f["cout"], np.float32)
        self.load_keras_weights(W)

    def keras_to_flat(self, W):
        """W: {'<layer>/kernel': (kD,kH,kW,Cin,Cout) ndarray, '<layer>/bias': (Cout,), '<norm>/gamma' ...} in Keras layout -> the flat
        fp32 parameter vector of this engine (also the layout of the Adam moments)."""
        host = np.zeros(self.n_flat, np.float32)
        for name, L in self.layout.items():
            k = np.asarray(W[name + "/kernel"], np.float32)
            if k.ndim == 4 and L["kind"] != "deconv":  # 2-D kernel (kH,kW,Cin,Cout) -> centre plane of a 3x3x3 / 1x1x1 kernel
                k = _embed_2d_kernel(k)
            o, n = L["w"]
            if L["kind"] == "deconv":
                kk = np.asarray(W[name + "/kernel"], np.float32)
                if kk.ndim == 4:                                              # (2,2,Cout,Cin) -> taps 0..3 of the 8
                    kk = np.concatenate([kk.reshape(4, L["cout"], L["cin"]), np.zeros((4, L["cout"], L["cin"]), np.float32)])
                host[o:o + n] = kk.reshape(8, L["cout"], L["cin"]).reshape(-1)
            elif L["kind"] == "conv":
                assert k.shape == (3, 3, 3, L["cin"], L["cout"]), (name, k.shape)
                host[o:o + n] = k.transpose(0, 1, 2, 4, 3).reshape(-1)        # -> [27][Cout][Cin]
            else:
                assert k.shape[-2:] == (L["cin"], L["cout"]), (name, k.shape)
                host[o:o + n] = k.reshape(L["cin"], L["cout"]).T.reshape(-1)  # -> [L][C]
            ob, nb = L["b"]
            host[ob:ob + nb] = np.asarray(W[name + "/bias"], np.float32)
            if L.get("norm"):
                host[L["gamma"][0]:L["gamma"][0] + nb] = np.asarray(W[L["norm"] + "/gamma"], np.float32)
                host[L["beta"][0]:L["beta"][0] + nb] = np.asarray(W[L["norm"] + "/beta"], np.float32)
        return host

    def load_keras_weights(self, W):
        self.P.copy_(torch.from_numpy(self.keras_to_flat(W)))
        for name, L in self.layout.items():
            if L.get("norm") and name in self.moving and (L["norm"] + "/moving_mean") in W:
                self.moving[name][0].copy_(torch.from_numpy(np.asarray(W[L["norm"] + "/moving_mean"], np.float32)))
                self.moving[name][1].copy_(torch.from_numpy(np.asarray(W[L["norm"] + "/moving_variance"], np.float32)))
        self.refresh_weight_copies()

    def flat_to_keras(self, host, moving=True):
        """inverse of keras_to_flat for any vector in the parameter layout (weights, Adam m, Adam v)"""
        W = OrderedDict()
        order = [c["name"] for c in self.plan.convs_forward_order()] + [u["name"] for u in self.plan.up.values()] + [self.plan.final["name"]]
        for name in order:
            L = self.layout[name]
            o, n = L["w"]
            if L["kind"] == "deconv":
                k = host[o:o + n].reshape(8, L["cout"], L["cin"])
                W[name + "/kernel"] = (k.reshape(2, 2, 2, L["cout"], L["cin"]) if self.plan.ndim == 3
                                       else k[:4].reshape(2, 2, L["cout"], L["cin"])).copy()
            elif L["kind"] == "conv":
                k = host[o:o + n].reshape(3, 3, 3, L["cout"], L["cin"]).transpose(0, 1, 2, 4, 3)
                W[name + "/kernel"] = (k if self.plan.ndim == 3 else k[1]).copy()          # 2-D: the centre kd plane is the 3x3 kernel
            else:
                k = host[o:o + n].reshape(L["cout"], L["cin"]).T
                W[name + "/kernel"] = k.reshape(((1, 1, 1) if self.plan.ndim == 3 else (1, 1)) + (L["cin"], L["cout"])).copy()
            ob, nb = L["b"]
            W[name + "/bias"] = host[ob:ob + nb].copy()
            if L.get("norm"):
                W[L["norm"] + "/gamma"] = host[L["gamma"][0]:L["gamma"][0] + nb].copy()
                W[L["norm"] + "/beta"] = host[L["beta"][0]:L["beta"][0] + nb].copy()
                if moving and name in self.moving:
                    mv = self.moving[name].cpu().numpy()
                    W[L["norm"] + "/moving_mean"], W[L["norm"] + "/moving_variance"] = mv[0].copy(), mv[1].copy()
        return W

    def export_keras_weights(self):
        return self.flat_to_keras(self.P.detach().cpu().numpy())

    def _upcat_layers(self):
        """decoder 'a' convs (UpSampling3D -> concatenate -> Conv3D, reference unet.py:132-138,61,102) that take the parity form:
        name -> (up-sampled channels, skip channels).  FMRI_UPCAT=0 keeps the 27-tap fused-upsample kernel (A/B switch)."""
        import os
        p = self.plan
        out = {}
        self.upcat_wgrad = set()                           # ... of which the weight gradient takes the parity form too (never in deterministic mode)
        # (fp32, round 6: the parity form runs on the fp32 instantiation of the same MFMA kernels where fmri_conv3d_upcat_ok says so - 3-D,
        # channels in multiples of 16; FMRI_F32_MFMA=0 keeps fp32 on the VALU kernels and with them on the 27-tap fused-upsample form)
        if (self.dtype != torch.bfloat16 and self.planar) or os.environ.get("FMRI_UPCAT", "1") == "0":
            return out
        for lv in p.dec:
            a = lv[0]
            if a["level"] in p.up:
                continue                                   # Deconvolution3D variant: the up-sampled tensor is materialised
            c1 = p.enc[a["level"]][1]["cout"]
            c0 = a["cin"] - c1
            # 2-D: the slices ride the kernels' D axis; 4 of them (one tile) stand in for "a batch that tiles" (see _use_upcat)
            D, H, W = p.level_dims(a["level"], 4) if self.planar else p.level_dims(a["level"])
            ok = ops.conv3d_upcat_ok(c0, c1, a["cout"], D, H, W, self.dtype, planar=self.planar)
            if ok & 1:
                out[a["name"]] = (c0, c1)
                if ok & 2 and not self.deterministic:
                    self.upcat_wgrad.add(a["name"])
        return out

    def _use_upcat(self, name):
        """parity form for this layer at the CURRENT batch size (2-D: the slice count must be a multiple of the 4-slice tile)"""
        return name in self.Wup and (not self.planar or self.N % 4 == 0)

    def refresh_weight_copies(self, overlap=False):
        """compute-dtype images of the fp32 parameters (forward filters, tap-flipped transposed filters for the input gradients, parity
        filters).  Default (round 6): ONE launch for all of them on the current stream (ops.pack_weights_batched).  The scheme it replaced,
        kept behind FMRI_PACK_BATCHED=0: overlap=True (the optimizer step): only the first encoder level is repacked on the current stream; the other layers -
        whose weights are the large ones and are first read a whole level later - are repacked on a side stream, in order of first use,
        with one event per encoder level and one for the decoder: `forward` waits for a level's event in front of that level, i.e. the
        repack runs under the first convolutions of the NEXT step instead of in front of them.  (One event for the whole encoder made the
        level-1 convs wait for the 256 -> 512 image, whose kernel - like every kernel next to a persistent conv launch, which fills all
        CUs - only gets CUs between two conv launches: 74 us of stall behind the first level in the rocprofv3 timeline; un-profiled the
        step does not notice: 13.09-13.15 ms either way.)"""
        if self.dev.type == "cuda" and os.environ.get("FMRI_PACK_BATCHED", "1") != "0":
            # round 6: one launch on the current stream for all plain / parity-form images (FMRI_PACK_BATCHED=0: the per-layer launches, on the
            # side stream when overlap=True - the round-2 ... round-5 scheme below)
            self._join_packs()
            self._repack(lambda name: True, batched=True)
            return
        early = set(c["name"] for c in self.plan.enc[0])
        if overlap and self.dev.type == "cuda" and os.environ.get("FMRI_PACK_OVERLAP", "1") != "0":
            if self._pack_stream is None:
                self._pack_stream = torch.cuda.Stream(device=self.dev)
                self._pack_events = {ld: torch.cuda.Event() for ld in range(1, len(self.plan.enc))}
                self._pack_event_dec = torch.cuda.Event()
            self._repack(lambda name: name in early)
            main = torch.cuda.current_stream(self.dev)
            self._pack_stream.wait_stream(main)
            enc = set(c["name"] for lv in self.plan.enc for c in lv) - early
            with torch.cuda.stream(self._pack_stream):
                # in order of first use: the encoder's images (joined after level 0), then the decoder's (joined in front of the decoder)
                for ld in range(1, len(self.plan.enc)):
                    lvl = set(c["name"] for c in self.plan.enc[ld])
                    self._repack(lambda name: name in lvl)
                    self._pack_events[ld].record(self._pack_stream)
                self._repack(lambda name: name not in early and name not in enc)
                self._pack_event_dec.record(self._pack_stream)
            self._pack_pending = list(range(1, len(self.plan.enc))) + ["dec"]
            return
        self._join_packs()
        self._repack(lambda name: True)

    def _join_packs(self, level=None):
        """make the current stream wait for the side-stream repack: the images of encoder level `level` (and of the levels before it), or
        (level=None) all of them"""
        while self._pack_pending and (level is None or (self._pack_pending[0] != "dec" and self._pack_pending[0] <= level)):
            k = self._pack_pending.pop(0)
            torch.cuda.current_stream(self.dev).wait_event(self._pack_event_dec if k == "dec" else self._pack_events[k])

    def _pack_table(self):
        """the images of _repack's first and third loop (plain layers, parity-form layers) as one table for ops.pack_weights_batched"""
        if getattr(self, "_ptab", None) is None:
            ent = [("plain", self.w_view(n), self.Wf[n], self.Wd.get(n)) for n in self.Wf if n not in self.Wfd]
            for n, W in self.Wup.items():
                c0, c1 = self.upcat[n]
                ent.append(("up", self.w_view(n), c0, c1, W["up_f"], W["up_d"], W["sk_f"], W["sk_d"], self.planar))
            self._ptab = ops.pack_table(ent, self.dev) if ent else (None, 0)
        return self._ptab

    def _repack(self, want, batched=False):
        """batched: every plain / parity-form image in ONE launch (want must then accept every layer); the per-layer launches are
        launch-bound - 14 of them cost the configs[1] step 0.12 ms wherever they run (tools/r06/pack_cost.py)"""
        if batched:
            tab, nb = self._pack_table()
            if tab is not None:
                ops.pack_weights_batched(tab, nb, self.dtype)
        for name in self.Wf:
            if not batched and want(name) and name not in self.Wfd:
                ops.pack_weights(self.w_view(name), self.Wf[name], self.Wd.get(name))
        for name, F in self.Wfd.items():
            if not want(name):
                continue
            w3 = self.w_view(name)
            weff, b27 = self.fold.effective(w3, self.w_view(F["u"]), self.b_view(name), self.b_view(F["u"]), F["cmid"], gemm_dtype=F["gd"])
            F["up_f"].copy_(weff)
            F["bias27"].copy_(b27)
            F["sk_f"].copy_(w3[:, :, F["cmid"]:])
            if F["up_d"] is not None:
                F["up_d"].copy_(weff[:, self.fold.mirror].transpose(-1, -2))            # Wc[p][1 - t']^T (fmri_conv3d_pack_up_weights' w_up_dgrad)
                F["sk_d"].copy_(w3[:, :, F["cmid"]:].flip(0).transpose(1, 2))           # tap-flipped transposed skip filters
        for name, W in self.Wup.items():
            if not batched and want(name):
                c0, c1 = self.upcat[name]
                ops.conv3d_pack_up_weights(self.w_view(name), c0, c1, W["up_f"], W["up_d"], W["sk_f"], W["sk_d"], planar=self.planar)
        for name, wt in self.Wt.items():
            if not want(name) or name in self._folded_up:
                continue
            if name in self.Wd2:
                Wd_, L = self.Wd2[name], self.layout[name]
                Wd_["w32"][13] = self.w_view(name)[:4].reshape(4 * L["cout"], L["cin"])
                Wd_["b4"].copy_(self.b_view(name).repeat(4))
                ops.pack_weights(Wd_["w32"], Wd_["wf"], Wd_["wd"])
                continue
            if name in self.Wdc:                          # parity p reads low-res offset 0 = combined tap 7 - p (mirrored: tap p)
                Wd_, w8 = self.Wdc[name], self.w_view(name)
                Wd_["up_f"][self._par8, 7 - self._par8] = w8.to(self.dtype)
                if Wd_["up_d"] is not None:
                    Wd_["up_d"][self._par8, self._par8] = w8.transpose(1, 2).to(self.dtype)
                continue
            ops.cast(self.w_view(name), wt)

    # ------------------------------------------------------------------------------------------------ buffers
    def set_batch(self, N):
        """switch to (and lazily allocate) the activation / gradient buffer set for batch size N"""
        key = (N, self.training)
        if key not in self._bufsets:
            self.N = N
            self._build_buffers()
            self._bufsets[key] = dict(act=self.act, grad=getattr(self, "grad", None), logits=self.logits, probs=self.probs,
                                      dlogits=getattr(self, "dlogits", None), pre=self.pre, nstats=self.nstats, nss=self.nss, norm_ws=self.norm_ws,
                                      wgrad_ws=getattr(self, "wgrad_ws", None),
                                      dummy_y=torch.zeros(self.logits.numel(), dtype=torch.uint8, device=self.dev))
        b = self._bufsets[key]
        self.N, self.act, self.grad, self.logits, self.probs, self.dlogits = N, b["act"], b["grad"], b["logits"], b["probs"], b["dlogits"]
        self.pre, self.nstats, self.nss, self.norm_ws, self.wgrad_ws = b["pre"], b["nstats"], b["nss"], b["norm_ws"], b["wgrad_ws"]
        self._dummy_y = b["dummy_y"]       # per buffer set and never freed: captured hipGraphs keep raw pointers to it

    def _dims(self, level):
        """leading (N, D, H, W) of an activation at `level` for the current batch"""
        if self.planar:
            return (1,) + self.plan.level_dims(level, self.N)
        return (self.N,) + self.plan.level_dims(level)

    def _build_buffers(self):
        p, N, dt, dev = self.plan, self.N, self.dtype, self.dev
        A = self.act = {}
        self.pre, self.nstats = {}, {}            # conv outputs before normalisation; saved statistics {mean, 1/s, 1/sigma}
        self.nss = {}                             # {scale, shift} of the apply pass, for the normalisation tail of the input-gradient launches
        G = N if (p.norm == "instance" and not self.planar) else 1
        if p.norm == "instance" and self.planar:
            G = N                                  # 2-D: every slice is a sample

        def block_bufs(c):
            A[c["name"]] = torch.empty(self._dims(c["level"]) + (c["cout"],), dtype=dt, device=dev)
            if c.get("norm"):
                self.pre[c["name"]] = torch.empty_like(A[c["name"]])
                self.nstats[c["name"]] = torch.zeros((G, c["cout"], 3), dtype=torch.float32, device=dev)
                self.nss[c["name"]] = torch.zeros((G, c["cout"], 2), dtype=torch.float32, device=dev)

        for lv in p.enc:
            for c in lv:
                block_bufs(c)
        for ld in range(p.depth - 1):
            A["pool_%d" % ld] = torch.empty(self._dims(ld + 1) + (p.enc[ld][1]["cout"],), dtype=dt, device=dev)
        for lv in p.dec:
            if lv[0]["level"] in p.up and lv[0]["name"] not in self.Wfd:      # (a folded transposed conv's output never exists)
                u = p.up[lv[0]["level"]]
                A[u["name"]] = torch.empty(self._dims(u["level"]) + (u["cout"],), dtype=dt, device=dev)
            for c in lv:
                block_bufs(c)
        cmax = max(c["cout"] for c in p.convs_forward_order())
        # scratch of the normalisation sums: [G][C][2] doubles - and, behind them, the per-workgroup blocks of the conv launches' tails
        self.norm_ws = torch.zeros(max(max(G, 1) * cmax * 2, ops.norm_tail_ws_doubles(max(G, 1), cmax) if dev.type == "cuda" else 0),
                                   dtype=torch.float64, device=dev) if p.norm else None
        nvox0 = int(np.prod(self._dims(0)))
        self.logits = torch.empty((nvox0, p.n_labels), dtype=torch.float32, device=dev)
        self.probs = torch.empty_like(self.logits)
        if getattr(self, "sums", None) is None:
            # ONE tensor for the engine's lifetime: captured hipGraphs of other batch sizes keep its address (re-creating it per buffer
            # set left them writing into freed memory, which the allocator then handed to the tile index list of the next volume)
            self.sums = torch.zeros(16, dtype=torch.float64, device=dev)
        if not self.training:
            self.grad, self.dlogits, self.wgrad_ws = None, None, None
            return
        Gd = self.grad = {}
        for name, t in A.items():
            Gd[name] = torch.empty_like(t)            # gradient w.r.t. the tensor (conv blocks: w.r.t. the pre-activation)
        for lv in p.dec:
            a = lv[0]
            # gradient of the concatenated conv input; the parity form writes the up-sampled part straight at low resolution,
            # so only the skip channels remain
            ccat = self.upcat[a["name"]][1] if self._use_upcat(a["name"]) else (a["c_skip"] if a["name"] in self.Wfd else a["cin"])
            Gd["cat_%d" % a["level"]] = torch.empty(self._dims(a["level"]) + (ccat,), dtype=dt, device=dev)
        self.dlogits = torch.empty_like(self.logits)
        # scratch for the slab flush of the MFMA weight-gradient kernel (max over the layers of this plan)
        need = 0
        for c in p.convs_forward_order():
            dims = self._dims(c["level"])
            c0, c1 = (c["c_up"], c["c_skip"]) if "c_up" in c else (c["cin"], 0)
            need = max(need, ops.conv3d_wgrad_workspace_bytes(c0, c1, c["cout"], dims[0], dims[1], dims[2], dims[3], dt, self.planar))
            if c["name"] in self.upcat_wgrad:             # parity form: the plain kernel only sees the skip channels
                need = max(need, ops.conv3d_wgrad_workspace_bytes(c1, 0, c["cout"], dims[0], dims[1], dims[2], dims[3], dt, self.planar))
        self.wgrad_ws = torch.empty(max(need // 4, 1), dtype=torch.float32, device=dev) if need else None

    # ------------------------------------------------------------------------------------------------ forward
    def _norm_mode(self):
        """(per_instance, eps_on_std) of fmri_norm_act_fwd for this plan"""
        return (1, True) if self.plan.norm == "instance" else (0, False)

    def _as_samples(self, t):
        """view [N][D][H][W][C] as [samples][voxels...][C] for the normalisation kernels (2-D: slices are the samples)"""
        return t.reshape((t.shape[1],) + tuple(t.shape[2:])) if self.planar else t

    def _tail_ok(self, c):
        """what the epilogue of conv block `c` can produce besides its output (ops.conv3d_fwd_tail_ok bits): only plain single-source
        blocks without a normalisation layer (the consumer then reads the block's OUTPUT); FMRI_TAIL_FUSE=0 switches it off (A/B).  2-D
        (round 6): MaxPooling2D / the final Conv2D out of the planar kernel's epilogue, FMRI_TAIL_FUSE_2D=0 switches that off."""
        if c.get("norm") or self.dtype != torch.bfloat16 or os.environ.get("FMRI_TAIL_FUSE", "1") == "0":
            return 0
        if self.planar and os.environ.get("FMRI_TAIL_FUSE_2D", "1") == "0":
            return 0
        key = (c["name"], self.N)
        cache = self.__dict__.setdefault("_tail_cache", {})
        if key not in cache:                                   # a host-side query per (layer, batch size), not per step
            d = self._dims(c["level"])
            cache[key] = ops.conv3d_fwd_tail_ok(c["cin"], c["cout"], d[0], d[1], d[2], d[3], self.dtype, planar=self.planar)
        return cache[key]

    def _ntail_ok(self, c0, c1, cout, level, kind=1):
        """does the 3-D MFMA launch (c0 | c1) -> cout at `level` carry a normalisation tail (kind 1: statistics of its output, kind 2: the
        backward reductions - in its asynchronous epilogue: ops.conv3d_fwd_ntail_ok)?  FMRI_NORM_FUSE = bit mask of the kinds in use,
        FMRI_NORM_FUSE_MAXLEVEL = deepest level that uses them.  Defaults from the interleaved A/B inside the configs[1] step
        (tools/ab_norm_tails.py, profiles/r03_norm_tails_ab.log): the statistics tail down to level 1 (batch norm: 19.91 -> 19.55 ms) or on
        level 0 only (instance norm: a workgroup's sums are flushed whenever the sample changes; 20.65 -> 20.47 ms); the backward tail is
        built and tested but off: it saves 170-340 us of reduction passes per full-resolution layer and costs the input-gradient launch
        100-180 us (tools/bench_ntail.py) - a launch that shares the chip with the weight-gradient stream, under which the HBM-bound
        reduction passes were already hidden (step +0.2 ... 0.5 ms with it)."""
        if self.plan.norm is None or self.planar or self.dtype != torch.bfloat16 or not (int(os.environ.get("FMRI_NORM_FUSE", "1")) & kind):
            return False
        if level > int(os.environ.get("FMRI_NORM_FUSE_MAXLEVEL", "1" if self.plan.norm == "batch" else "0")):
            return False
        key = (c0, c1, cout, level, self.N)
        cache = self.__dict__.setdefault("_ntail_cache", {})
        if key not in cache:
            d = self._dims(level)
            cache[key] = ops.conv3d_fwd_ntail_ok(c0, c1, cout, d[0], d[1], d[2], d[3], self.dtype)
        return cache[key]

    def _block_fwd(self, c, src0, src1, up0, bn_training, pool=None, final=None):
        """one [conv -> (norm) -> ReLU] block (reference create_convolution_block, unet.py:89-115).  pool: tensor that receives
        MaxPooling3D(2) of the block's output; final: the final 1x1x1 conv descriptor whose logits the epilogue computes - both only
        when _tail_ok(c) says so (the caller checks)."""
        name = c["name"]
        out, act = (self.pre[name], ACT_NONE) if c.get("norm") else (self.act[name], ACT_RELU)
        if pool is not None or final is not None:
            w1 = self.w_view(final["name"]).reshape(-1) if final is not None else None
            ops.conv3d_fwd_tail(src0, self.Wf[name], self.b_view(name), out, pool=pool, w1=w1,
                                b1=self.b_view(final["name"]) if final is not None else None,
                                logits=self.logits.reshape(-1) if final is not None else None, act=act, planar=self.planar)
            return self.act[name]
        # normalised block with training statistics: the conv sums its own output in its epilogue where the launch allows it
        per, eos = self._norm_mode()
        want_sums = bool(c.get("norm")) and not (self.plan.norm == "batch" and not bn_training)
        c0, c1 = src0.shape[-1], (0 if src1 is None else src1.shape[-1])
        summed = False
        if up0 and self._use_upcat(name):
            W = self.Wup[name]
            if want_sums and src1 is not None and self._ntail_ok(c1, 0, c["cout"], c["level"]):
                ops.conv3d_upcat_fwd_stats(src0, src1, W["up_f"], W["sk_f"], self.b_view(name), out, self.norm_ws, per, act=act)
                summed = True
            else:
                ops.conv3d_upcat_fwd(src0, src1, W["up_f"], W["sk_f"], self.b_view(name), out, act=act, planar=self.planar)
        elif want_sums and self._ntail_ok(c0, c1, c["cout"], c["level"]):
            ops.conv3d_fwd_stats(src0, src1, self.Wf[name], self.b_view(name), out, self.norm_ws, per, up0=up0, act=act)
            summed = True
        else:
            ops.conv3d_fwd(src0, src1, self.Wf[name], self.b_view(name), out, up0=up0, act=act, planar=self.planar)
        if not c.get("norm"):
            return self.act[name]
        st = self.nstats[name]
        if self.plan.norm == "batch" and not bn_training:
            mv = self.moving[name]                                            # inference: moving averages (Keras learning_phase 0)
            st[0, :, 0] = mv[0]
            st[0, :, 1] = torch.rsqrt(mv[1] + 1e-3)
            st[0, :, 2] = st[0, :, 1]
            per = -1
        (ops.norm_act_fwd_pre if summed else ops.norm_act_fwd)(
            self._as_samples(self.pre[name]), self.gb_view(name, "gamma"), self.gb_view(name, "beta"),
            self._as_samples(self.act[name]), st, self.norm_ws, per, eps=1e-3, eps_on_std=eos, act=ACT_RELU)
        if self.plan.norm == "batch" and bn_training:
            # Keras moving statistics (momentum 0.99); the variance fed to the moving average is sample-size corrected
            M = float(self.pre[name].numel() // c["cout"])
            mv = self.moving[name]
            if st.is_cuda:
                ops.norm_moving_update(st, mv[0], mv[1], M)
            else:
                var = (1.0 / (st[0, :, 1] * st[0, :, 1]) - 1e-3) * (M / max(M - (1.0 + 1e-3), 1.0))
                mv[0].mul_(0.99).add_(st[0, :, 0], alpha=0.01)
                mv[1].mul_(0.99).add_(var, alpha=0.01)
        return self.act[name]

    def forward(self, x, bn_training=None):
        """x: [N,D,H,W,Cin] compute dtype, device.  Leaves logits (fp32 [nvox, L]) in self.logits."""
        p, A = self.plan, self.act
        assert tuple(x.shape) == self._dims(0) + (p.in_channels,), (x.shape,)
        bn_training = self.training if bn_training is None else bn_training
        self.x_in = x
        h = x
        for ld, lv in enumerate(p.enc):
            if ld > 0:                                      # this level's weight images were repacked on the side stream (FMRI_PACK_LEVELS=0: A/B,
                self._join_packs(level=ld if os.environ.get("FMRI_PACK_LEVELS", "1") != "0" else len(p.enc))     # wait for the whole encoder's)
            h = self._block_fwd(lv[0], h, None, False, bn_training)
            # MaxPooling3D behind the level's second block comes out of that conv's epilogue when the kernel can do it
            fuse_pool = ld < p.depth - 1 and (self._tail_ok(lv[1]) & 1)
            h = self._block_fwd(lv[1], h, None, False, bn_training, pool=A["pool_%d" % ld] if fuse_pool else None)
            if ld < p.depth - 1:
                h = A["pool_%d" % ld] if fuse_pool else ops.maxpool_fwd(h, A["pool_%d" % ld], planar=self.planar)
        self._join_packs()
        for lv in p.dec:
            a, b = lv
            skip = A[p.enc[a["level"]][1]["name"]]
            if a["name"] in self.Wfd:
                F = self.Wfd[a["name"]]
                ops.conv3d_upcat_fwd_bias27(h, skip, F["up_f"], F["sk_f"], F["bias27"], A[a["name"]], act=ACT_RELU)
            elif a["level"] in p.up:
                u = p.up[a["level"]]
                if u["name"] in self.Wd2 and h.shape[1] % 4 == 0:
                    Wd_ = self.Wd2[u["name"]]
                    t4 = self._d2s_tmp(u["name"], h, Wd_)
                    ops.conv3d_fwd(h, None, Wd_["wf"], Wd_["b4"], t4, act=ACT_NONE, planar=True)
                    self._d2s_views(A[u["name"]], t4)[0].copy_(self._d2s_views(A[u["name"]], t4)[1])
                elif u["name"] in self.Wdc:
                    ops.conv3d_upcat_fwd(h, None, self.Wdc[u["name"]]["up_f"], None, self.b_view(u["name"]), A[u["name"]], act=ACT_NONE)
                else:
                    ops.deconv_fwd(h, self.Wt[u["name"]], self.b_view(u["name"]), A[u["name"]], planar=self.planar)
                self._block_fwd(a, A[u["name"]], skip, False, bn_training)
            else:
                self._block_fwd(a, h, skip, True, bn_training)
            # the last block also produces the logits of the final 1x1x1 conv (one label) in its epilogue when the kernel can do it
            fuse_final = lv is p.dec[-1] and p.n_labels == 1 and (self._tail_ok(b) & 2)
            h = self._block_fwd(b, A[a["name"]], None, False, bn_training, final=p.final if fuse_final else None)
        f = p.final
        if not (p.dec and fuse_final):
            ops.conv1x1_fwd(h, self.w_view(f["name"]), self.b_view(f["name"]), self.logits)
        return self.logits

    def loss_forward(self, y_true, weight=None):
        """y_true uint8 [nvox*L] device.  probs + the 8 metric sums (accumulated into zeroed self.sums)."""
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

    def predict(self, x):
        self.forward(x, bn_training=False)
        self.sums.zero_()
        # sigmoid only (y_true is irrelevant for probs): reuse the fused kernel with an all-zero label buffer
        ops.sigmoid_dice_fwd(self.logits, self._dummy_y, self.probs, self.sums)
        return self.probs

    # ------------------------------------------------------------------------------------------------ backward
    def _block_bwd(self, c, src0, src1, up0):
        """G[c] holds dL/d(output of the block) (already ReLU-masked when the block has no norm).  Leaves dL/d(conv output) in
        G[c] and accumulates the block's parameter gradients."""
        name = c["name"]
        g = self.grad[name]
        if c.get("norm") and name in self._dz_ready:
            # the input-gradient launch that produced g already applied the ReLU mask and left the two reductions in norm_ws (_dgrad_into)
            self._dz_ready.discard(name)
            ops.norm_act_bwd_pre(self.pre[name], g, self.gb_view(name, "gamma"), self.nstats[name], g, self.gb_view(name, "gamma", self.G),
                                 self.gb_view(name, "beta", self.G), self.norm_ws, self._norm_mode()[0])
        elif c.get("norm"):
            per, _ = self._norm_mode()
            # (beta given: the ReLU mask is recomputed from the pre-normalisation tensor, the block's output is not read)
            ops.norm_act_bwd(self._as_samples(self.pre[name]), None, self._as_samples(g),
                             self.gb_view(name, "gamma"), self.nstats[name], self._as_samples(g), self.gb_view(name, "gamma", self.G),
                             self.gb_view(name, "beta", self.G), self.norm_ws, per, act=ACT_RELU, beta=self.gb_view(name, "beta"))
        def wgrad():
            if up0 and name in self.upcat_wgrad and self._use_upcat(name):
                ops.conv3d_upcat_wgrad(src0, src1, g, self.w_view(name, self.G), self.b_view(name, self.G), self.dwc_scratch,
                                       workspace=self.wgrad_ws, planar=self.planar)
            else:
                ops.conv3d_wgrad(src0, src1, g, self.w_view(name, self.G), self.b_view(name, self.G), up0=up0, planar=self.planar,
                                 workspace=self.wgrad_ws)
            self._grad_ready(name)

        if self._wg_stream is None:
            wgrad()
            return
        # the weight gradients are off the critical path of the backward pass (nothing reads them before the optimizer step): they run on
        # their own stream, behind the event "dL/d(conv output) is final", next to the input-gradient chain of the main stream
        # (a third stream for the first conv's weight gradient - HBM-bound, ready last - measured nothing: 13.09-13.13 ms either way)
        self._wg_stream.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(self._wg_stream):
            wgrad()

    def _folded_bwd(self, a, x_low, skip, cat, low):
        """backward of the folded transposed conv + conv of decoder block `a`: G[a] holds dL/d(conv output).  Weight gradients (side
        stream): parity-filter gradients from the MFMA kernel, chained through the transposed conv's weights by fmri_hip.deconv_fold;
        input gradients: w.r.t. the low-res tensor (parity-form launch over the space-to-depth view of dy) and the skip tensor."""
        name, F = a["name"], self.Wfd[a["name"]]
        g = self.grad[name]

        def wgrad():
            Cout = a["cout"]
            dwc = self.dwc_scratch[:64 * Cout * F["cin"]]
            ops.conv3d_upcat_wgrad_parts(x_low, skip, g, self.w_view(name, self.G), self.b_view(name, self.G), dwc, workspace=self.wgrad_ws)
            s27 = F["s27"]
            s27.zero_()
            ops.border_class_sums(g, s27)
            s27[13] = self.b_view(name, self.G) - s27.sum(0)        # interior class = all voxels (the conv's bias gradient) - the 26 border classes
            dw3u, dwt, dbt = self.fold.chain(dwc.view(8, 8, Cout, F["cin"]), self.w_view(name), self.w_view(F["u"]), self.b_view(F["u"]), F["cmid"], s27,
                                             gemm_dtype=F["gd"])
            self.w_view(name, self.G)[:, :, :F["cmid"]].add_(dw3u)
            self.w_view(F["u"], self.G).add_(dwt)
            self.b_view(F["u"], self.G).add_(dbt)
            self._grad_ready(name)
            self._grad_ready(F["u"])

        if self._wg_stream is None:
            wgrad()
        else:
            self._wg_stream.wait_stream(torch.cuda.current_stream(self.dev))
            with torch.cuda.stream(self._wg_stream):
                wgrad()
        ops.conv3d_upcat_dgrad(g, F["up_d"], F["sk_d"], self._mask_of(low), None, self.grad[low], cat)

    def _dgrad_into(self, b, a):
        """dL/d(output of block a) from block b's conv (whose only input is a's output) -> grad[a].  a normalised: the launch's epilogue
        applies a's ReLU mask (recomputed from a's conv output) and sums dz and dz * x for a's normalisation backward where it can."""
        Gd = self.grad
        an, bn = a["name"], b["name"]
        if an in self.pre and self._ntail_ok(b["cout"], 0, a["cout"], a["level"], kind=2):
            per = self._norm_mode()[0]
            ops.norm_scale_shift(self.nstats[an], self.gb_view(an, "gamma"), self.gb_view(an, "beta"), self.nss[an])
            ops.conv3d_dgrad_norm(Gd[bn], self.Wd[bn], self.pre[an], self.nss[an], Gd[an], self.norm_ws, per, act=ACT_RELU)
            self._dz_ready.add(an)
        else:
            ops.conv3d_dgrad(Gd[bn], self.Wd[bn], Gd[an], mask=self._mask_of(an), planar=self.planar)

    def _mask_of(self, name):
        """ReLU mask tensor a consumer applies to the gradient of `name`'s output, or None when the block is normalised (its
        norm backward applies the activation derivative itself) or `name` is not a conv block."""
        if name in self.pre:
            return None
        return self.act.get(name)

    def backward(self, y_true, grad_scale=1.0, weight=None, dprobs=None, dprobs_scale=1.0, seg_loss=True):
        """`dprobs` ([..., ld >= n_labels], optional): a gradient that arrives on the probabilities from outside - the adversarial term of
        reference fetal/experiments/train_adv.py:177-180 (the frozen discriminator's input gradient); `seg_loss=False` leaves only that
        term (the unlabelled pass of train_semi.py:176-183)."""
        p, A, Gd = self.plan, self.act, self.grad
        normed = p.norm is not None
        self._main_stream = torch.cuda.current_stream(self.dev) if self.dev.type == "cuda" else None
        self._dz_ready = set()
        self.G.zero_()
        if seg_loss and self.loss_kind == ops.LOSS_WEIGHTED_DICE:
            ns, nl = self._wdice_groups()
            ops.weighted_dice_bwd(self.probs, y_true, self._gsums, self.sums, self.dlogits, ns, nl, grad_scale=grad_scale)
        elif seg_loss:
            ops.sigmoid_loss_bwd(self.probs, y_true, self.sums, self.dlogits, self.loss_kind, self.loss_param, smooth=1.0, grad_scale=grad_scale,
                                 weight=weight)
        if dprobs is not None:
            ops.sigmoid_chain(self.probs, dprobs, self.dlogits, scale=dprobs_scale, accumulate=seg_loss)
        f = p.final
        last = p.dec[-1][1] if p.dec else p.enc[-1][1]
        ops.conv1x1_bwd(A[last["name"]], self.w_view(f["name"]), self.dlogits, Gd[last["name"]], self.w_view(f["name"], self.G),
                        self.b_view(f["name"], self.G), relu_mask=not normed)
        self._grad_ready(f["name"])
        # decoder, shallowest level first
        for lv in reversed(p.dec):
            a, b = lv
            ld = a["level"]
            low = self._dec_input_name(ld)
            skip = A[p.enc[ld][1]["name"]]
            self._block_bwd(b, A[a["name"]], None, False)
            self._dgrad_into(b, a)
            cat = Gd["cat_%d" % ld]
            if a["name"] in self.Wfd:
                self._folded_bwd(a, A[low], skip, cat, low)
            elif ld in p.up:
                u = p.up[ld]
                self._block_bwd(a, A[u["name"]], skip, False)
                ops.conv3d_dgrad(Gd[a["name"]], self.Wd[a["name"]], cat, planar=self.planar)
                if u["name"] in self.Wd2 and A[low].shape[1] % 4 == 0:
                    Wd_, L = self.Wd2[u["name"]], self.layout[u["name"]]
                    dyc = ops.slice_channels(cat, 0, Gd[u["name"]])
                    t4 = self._d2s_tmp(u["name"], A[low], Wd_)
                    yv, tv = self._d2s_views(dyc, t4)
                    tv.copy_(yv)                                               # space-to-depth of the gradient
                    ops.conv3d_dgrad(t4, Wd_["wd"], Gd[low], mask=self._mask_of(low), planar=True)
                    Wd_["dw"].zero_()
                    Wd_["db"].zero_()
                    ops.conv3d_wgrad(A[low], None, t4, Wd_["dw"], Wd_["db"], planar=True)
                    self.w_view(u["name"], self.G)[:4].add_(Wd_["dw"][13].view(4, L["cout"], L["cin"]))
                    self.b_view(u["name"], self.G).add_(Wd_["db"].view(4, L["cout"]).sum(0))
                elif u["name"] in self.Wdc:
                    Wd_ = self.Wdc[u["name"]]
                    dyc = ops.slice_channels(cat, 0, Gd[u["name"]])          # the transposed conv's own slice of the concat gradient
                    ops.conv3d_upcat_dgrad(dyc, Wd_["up_d"], None, self._mask_of(low), None, Gd[low], None)
                    ops.conv3d_upcat_wgrad(A[low], None, dyc, Wd_["dw27"], self.b_view(u["name"], self.G), self.dwc_scratch)
                    L = self.layout[u["name"]]
                    dwc = self.dwc_scratch[:64 * L["cout"] * L["cin"]].view(8, 8, L["cout"], L["cin"])
                    self.w_view(u["name"], self.G).add_(dwc[self._par8, 7 - self._par8])
                else:
                    ops.deconv_bwd(A[low], self.Wt[u["name"]], cat, Gd[low], self.w_view(u["name"], self.G), self.b_view(u["name"], self.G),
                                   dy_off=0, xmask=self._mask_of(low), planar=self.planar)
                self._grad_ready(u["name"])
            elif self._use_upcat(a["name"]):
                self._block_bwd(a, A[low], skip, True)
                W = self.Wup[a["name"]]
                ops.conv3d_upcat_dgrad(Gd[a["name"]], W["up_d"], W["sk_d"], self._mask_of(low), None, Gd[low], cat, planar=self.planar)
            else:
                self._block_bwd(a, A[low], skip, True)
                ops.conv3d_dgrad(Gd[a["name"]], self.Wd[a["name"]], cat, planar=self.planar)
                ops.upsample_bwd(cat, Gd[low], dy_off=0, xmask=self._mask_of(low), planar=self.planar)
        # encoder, deepest level first
        for ld in range(p.depth - 1, -1, -1):
            ca, cb = p.enc[ld]
            if ld < p.depth - 1:
                # gradient of enc[ld].b output = pooled path + skip path, masked by its ReLU
                cat = Gd["cat_%d" % ld]
                ops.maxpool_bwd(A[cb["name"]], Gd["pool_%d" % ld], Gd[cb["name"]], add=cat, add_off=cat.shape[-1] - cb["cout"],
                                relu_mask=not normed, planar=self.planar)
            self._block_bwd(cb, A[ca["name"]], None, False)
            self._dgrad_into(cb, ca)
            xin = self.x_in if ld == 0 else A["pool_%d" % (ld - 1)]
            self._block_bwd(ca, xin, None, False)
            if ld > 0:
                ops.conv3d_dgrad(Gd[ca["name"]], self.Wd[ca["name"]], Gd["pool_%d" % (ld - 1)], planar=self.planar)
        if self._wg_stream is not None:
            torch.cuda.current_stream(self.dev).wait_stream(self._wg_stream)
        if self.deterministic:
            ops.deterministic_finish(self.G, self.G64)       # G += the fixed-point sums of every gradient kernel of this pass
            if self.dist is not None:
                self.dist.begin()                            # nothing was reduced during this backward: one range, the whole buffer
        if self.dist is not None:
            self.dist.finish(self)

    def _d2s_tmp(self, name, x_low, Wd_):
        """(1, S, H, W, 4*Cout) scratch of the 2-D transposed conv for the current slice count"""
        key = "t4_%d" % x_low.shape[1]
        if key not in Wd_:
            Wd_[key] = torch.empty(tuple(x_low.shape[:-1]) + (Wd_["b4"].numel(),), dtype=self.dtype, device=self.dev)
        return Wd_[key]

    @staticmethod
    def _d2s_views(y, t4):
        """matching 6-D views (S, H, ah, W, aw, Cout) of the full-resolution tensor y (1,S,2H,2W,Cout) and of t4 (1,S,H,W,4*Cout)"""
        _, S, H, W, C4 = t4.shape
        C = C4 // 4
        return y.view(S, H, 2, W, 2, C), t4.view(S, H, W, 2, 2, C).permute(0, 1, 3, 2, 4, 5)

    def _dec_input_name(self, ld):
        """name of the activation that is up-sampled into decoder level ld"""
        p = self.plan
        if ld == p.depth - 2:
            return p.enc[-1][1]["name"]
        for lv in p.dec:
            if lv[0]["level"] == ld + 1:
                return lv[1]["name"]
        raise KeyError(ld)

    def close(self):
        """give up the process-wide deterministic registration (FMRI_DETERMINISTIC=1; also done when the engine is garbage-collected)"""
        fin = self.__dict__.pop("_det_finalizer", None)
        if fin is not None:
            fin()

    def grad_streams(self):
        """streams that enqueue parameter-gradient kernels during backward (a gradient bucket is complete when all of them got there)"""
        return [st for st in (getattr(self, "_main_stream", None), self._wg_stream) if st is not None]

    def _grad_ready(self, name):
        # deterministic mode: the gradients sit in the int64 shadow until deterministic_finish at the end of backward(), so no bucket of
        # G may be reduced before that - the whole buffer goes in dist.finish (early buckets all-reduced zeros)
        if self.dist is not None and not self.deterministic:
            self.dist.grad_ready(self, name)

    # ------------------------------------------------------------------------------------------------ optimizer
    def adam_step(self, lr, beta1=0.9, beta2=0.999, eps=1e-7, grad_scale=1.0):
        self.t += 1
        lr_t = lr * math.sqrt(1.0 - beta2 ** self.t) / (1.0 - beta1 ** self.t)
        ops.adam_step(self.P, self.G, self.M, self.V, lr_t, beta1, beta2, eps, grad_scale)
        self.refresh_weight_copies(overlap=True)

    def train_step(self, x, y_true, lr, weight=None):
        """one full step: forward, Dice, backward, (all-reduce), Adam.  Returns the device tensor of metric sums."""
        self.forward(x)
        self.loss_forward(y_true, weight)
        # data parallel: with the global-batch Dice sums the ranks' gradients are summed (scale 1); with per-rank losses
        # (global_dice=False) the all-reduce sum is turned into the mean by scaling each rank's loss gradient by 1/world
        self.backward(y_true, grad_scale=(self.dist.grad_scale if self.dist is not None else 1.0), weight=weight)
        self.adam_step(lr)
        return self.sums

    @staticmethod
    def metrics_from_sums(s, smooth=1.0, loss_kind=0, loss_param=1.0):
        s = [float(v) for v in s]
        dice = (2.0 * s[0] + smooth) / (s[1] + s[2] + smooth)
        vod = (s[3] + smooth) / (s[4] + s[5] - s[3] + smooth)
        return dict(loss=ops.loss_value_from_sums(s, loss_kind, loss_param, smooth), dice_coefficient=dice, vod_coefficient=vod,
                    binary_accuracy=s[6] / max(s[7], 1.0))


_DET_OWNER = []          # [weakref to the engine that holds the process-wide fmri_set_deterministic registration]


def _release_deterministic(token):
    if _DET_OWNER and _DET_OWNER[0] is token:
        _DET_OWNER.clear()
        try:
            ops.set_deterministic(None, None)
        except Exception:                                    # interpreter shutdown: the library may be gone already
            pass


def _register_deterministic(eng):
    """fmri_set_deterministic keeps raw device pointers process-wide: ONE engine may hold the registration, and it is given back when
    that engine dies or is closed (a stale registration redirected a later engine's gradients into freed memory)"""
    import weakref
    if _DET_OWNER and _DET_OWNER[0]() is not None:
        raise RuntimeError("FMRI_DETERMINISTIC=1: another engine of this process holds the deterministic-gradient registration "
                           "(one deterministic engine per process at a time; call its close() first)")
    _DET_OWNER.clear()
    ref = weakref.ref(eng)
    _DET_OWNER.append(ref)
    ops.set_deterministic(eng.G, eng.G64)
    eng._det_finalizer = weakref.finalize(eng, _release_deterministic, ref)


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
