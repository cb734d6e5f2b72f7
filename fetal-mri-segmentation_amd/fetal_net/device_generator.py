"""Training-patch generator that never leaves the MI355X.

The reference builds each patch on one host thread (fetal_net/generator.py:222-328: random corner -> augment_data or
extract_patch -> convert_data) and Keras copies the batch to the device; with the convolutions at MFMA speed that thread is the
bottleneck.  Here every (padded) volume of the split is uploaded once - 288 GB of HBM holds any dataset of this kind - and a patch
is: the reference's random draws on the host (same order, same generators => same transformation for the same seed), then one
fmri_affine_sample launch for the image (trilinear, cval = volume minimum) and one for the labels (nearest, cval = 0) straight
into the batch tensors, then the elementwise intensity passes.  The generator yields torch CUDA tensors in the reference's logical
layouts ((N,1,X,Y,Z) / (N,X,Y,C) float32 images, uint8 labels); Model.train_on_batch / fit_generator take them as they are.
A data file with a non-empty `mask` array (the distance masks of the mask-weighted loss, reference generator.py:18-21) makes the generator
yield ([x, masks], y) like the reference's convert_data (generator.py:397-401): the mask patch is sampled like the labels (nearest, outside
= 0, same transformation) at the label slices.  As in the reference the masks are NOT padded with the volumes.

`device_data_generator` keeps the keyword arguments of the reference's `data_generator` (generator.py:222-225).
Applied augmenters: flip, scale, iso_scale, rotate, translate, piecewise_affine, elastic_transform (imgaug), contrast,
intensity_multiplication, gaussian_filter (skimage.filters.gaussian), poisson_noise (the reference's shot_noise), speckle_noise,
gaussian_noise, coarse_dropout (imgaug CoarseDropout) - in the reference's order (augment.py:344-375), i.e. everything the reference's
default config switches on (fetal/config_utils.py:81-123).  The three imgaug augmenters follow imgaug 0.4.0 as published (the reference
does not pin a version and the package is not installed here: their oracle says "parity unpinned", oracle/augment_oracle.py); the elastic
warp restates `_map_coordinates`' scipy branch, not the cv2.remap branch imgaug prefers for float images when cv2 is importable.
imgaug's piecewise_affine (commented out in the reference's default config, config_utils.py:101-103) is applied as well when configured,
between the affine sampling and the elastic transform (augment.py:344-347), on the reference's 2 x 2 grid.
The noise fields (normal, uniform and Poisson draws) are made inside the kernels by a counter-based generator (Philox4x32-10 keyed by
`noise_seed`; csrc/augment.hip), not by numpy: same distributions, other streams.  The patches of a batch are PLANNED one after the other
on the host (every draw of the reference, in its order) and then LAUNCHED together, one launch per step of the chain (`_Sampler.plan` /
`launch_batch`); configurations the batch kernels do not cover (previous-slice truth channels, piecewise affine, the Gaussian filter,
drop_easy_patches' interleaved draws) go patch by patch (`launch_one`) - same draws, same batches (tests/test_gpu_augment.py).
"""
import random
import warnings

import numpy as np

from .augment import distort_image, draw_augment_parameters

_UNSUPPORTED = ()                          # every augmenter of reference augment.py:222-377 is applied


class DeviceDataFile(object):
    """The volumes of a data file, padded as the reference pads them for training (generator.py:13-57: DataFileDummy with
    `samples_pad`, then pad_samples), resident in HBM: .data[i] float32 (X,Y,Z), .truth[i] uint8 (X,Y,Z), .stats min / max."""

    def __init__(self, data_file, patch_shape, samples_pad=3, truth_downsample=None, indices=None, device="cuda"):
        import torch
        ds = truth_downsample or 1
        root = data_file.root
        n = len(root.data)
        self.indices = list(range(n)) if indices is None else list(indices)
        self.data, self.truth, self.min, self.max, self.mask = {}, {}, {}, {}, None
        if hasattr(root, "mask") and root.mask is not None and len(root.mask):
            self.mask = {}
        self.subject_ids = [s for s in root.subject_ids] if hasattr(root, "subject_ids") else None
        out_shape = [patch_shape[0] // ds, patch_shape[1] // ds, 1]
        padding = np.ceil(np.subtract(patch_shape, out_shape) / 2).astype(int)
        for i in self.indices:
            d = np.asarray(root.data[i])
            t = np.asarray(root.truth[i])
            d = np.pad(d, samples_pad, "constant", constant_values=d.min())
            t = np.pad(t, samples_pad, "constant", constant_values=0)
            dmin, dmax = float(np.min(d)), float(np.max(d))
            d = np.pad(d, [(p, p) for p in padding], "constant", constant_values=dmin)
            t = np.pad(t, [(p, p) for p in padding], "constant", constant_values=0)
            fit = np.ceil(np.maximum(np.subtract(patch_shape, d.shape) + 1, 0) / 2).astype(int)
            d = np.pad(d, [(p, p) for p in fit], "constant", constant_values=dmin)
            fit = np.ceil(np.maximum(np.subtract(patch_shape, t.shape) + 1, 0) / 2).astype(int)
            t = np.pad(t, [(p, p) for p in fit], "constant", constant_values=0)
            self.data[i] = torch.from_numpy(np.ascontiguousarray(d, dtype=np.float32)).to(device)
            self.truth[i] = torch.from_numpy(np.ascontiguousarray(t).astype(np.uint8)).to(device)
            self.min[i], self.max[i] = dmin, dmax
            if self.mask is not None:
                self.mask[i] = torch.from_numpy(np.ascontiguousarray(np.asarray(root.mask[i]), dtype=np.float32)).to(device)
        self.device = device
        self._mask_edge = {}

    def mask_for_crops(self, i):
        """the mask as the reference's un-augmented path sees it: a crop that runs past the (unpadded) mask is completed with edge values
        (reference utils/patches.py:66-90), so the volume is grown with replicated borders up to the padded label volume's extent"""
        if i not in self._mask_edge:
            import torch
            m = self.mask[i]
            grow = [max(int(t) - int(s), 0) for t, s in zip(self.truth[i].shape, m.shape)]
            if any(grow):
                m = torch.nn.functional.pad(m[None, None], (0, grow[2], 0, grow[1], 0, grow[0]), mode="replicate")[0, 0].contiguous()
            self._mask_edge[i] = m
        return self._mask_edge[i]

    def nbytes(self):
        return sum(v.numel() * 4 for v in self.data.values()) + sum(v.numel() for v in self.truth.values()) + \
            sum(v.numel() * 4 for v in (self.mask or {}).values())


def random_list_generator(index_list):
    while True:
        np.random.seed()                                   # the reference re-seeds from the OS on every pass (generator.py:194-197)
        yield from random.sample(index_list, len(index_list))


def list_generator(index_list):
    while True:
        yield from index_list


COARSE_MIN_SIZE = 3


def _coarse_grid(shape2d, size_percent, rng):
    """imgaug parameters.FromLowerResolution: one size_percent per axis - a list is a choice among its values (the reference's default
    [0.10, 0.30]), a tuple a uniform range, a number itself; grid = int(extent * percent), at least MIN_SIZE = 3 per side (the `min_size`
    imgaug 0.4.0's CoarseDropout passes to FromLowerResolution: a patch under 30 voxels at 10 % still drops cells, not whole slices)"""
    out = []
    for extent in shape2d:
        if isinstance(size_percent, list):
            sp = size_percent[int(rng.randint(len(size_percent)))]
        elif isinstance(size_percent, tuple):
            sp = rng.uniform(size_percent[0], size_percent[1])
        else:
            sp = size_percent
        out.append(max(int(extent * sp), COARSE_MIN_SIZE))
    return tuple(out)


class _Sampler(object):
    def __init__(self, ddf, patch_shape, augment, truth_index, truth_size, prev_truth_index, prev_truth_size, strict, noise_seed):
        import torch
        from fmri_hip import ops
        self.torch, self.ops = torch, ops
        self.ddf = ddf
        self.patch_shape = tuple(int(v) for v in patch_shape)
        self.augment = augment
        self.truth_index, self.truth_size = truth_index, truth_size
        self.prev_truth_index, self.prev_truth_size = prev_truth_index, prev_truth_size
        self.n_chan = self.patch_shape[2] + (prev_truth_size if prev_truth_index is not None else 0)
        self.stats, self.ws = ops.aug_workspace(ddf.device)
        self.stats_b = self.ws_b = None                  # the batch path's [B] workspaces, made on first use
        self.seed, self.seq = int(noise_seed), 0
        self.batched = True
        self.gen = torch.Generator(device=ddf.device)
        self.gen.manual_seed(noise_seed)
        # imgaug keeps its own random state (the reference's numpy / python streams are not advanced by its draws): the grid sizes of the coarse
        # dropout come from a private host generator, everything per voxel from the device generator
        self.host_rng = np.random.RandomState(noise_seed)
        if augment is not None:
            bad = [k for k in _UNSUPPORTED if augment.get(k) is not None]
            if bad:
                msg = "augmenters not applied on the device path: %s" % ", ".join(bad)
                if strict:
                    raise NotImplementedError(msg)
                warnings.warn(msg)

    def _next_seq(self):
        # the kernels take the call number as uint32 and 0 means "this patch skips the step": wrap inside 1 .. 2^32 - 1
        self.seq = self.seq % 0xFFFFFFFF + 1
        return self.seq

    def plan(self, index):
        """every HOST draw of one patch, in the reference's order (numpy's global state, python's `random`, this sampler's private imgaug
        state and the call numbers of its in-kernel draws), and the matrices they give - nothing is enqueued.  -> dict for launch_one /
        launch_batch."""
        ddf, ps = self.ddf, self.patch_shape
        data, truth = ddf.data[index], ddf.truth[index]
        corner = [np.random.randint(low=0, high=h) for h in np.array(truth.shape) - np.array(ps)]
        q = {"index": index, "corner": corner, "corners": None, "elastic_seq": 0, "elastic_rng": True, "shot_seq": 0, "speckle_seq": 0,
             "gaussian_seq": 0, "dropout_seq": 0, "grid": (1, 1)}
        if self.augment is not None:
            p = draw_augment_parameters(self.augment, 3, ddf.min[index], ddf.max[index])
            geo = dict(flip_axis=p["flip_axis"], scale_factor=p["scale_factor"], rotate_factor=p["rotate_factor"], translate_factor=p["translate_factor"])
            _, A = distort_image(data, np.eye(4), **geo)
            _, At = distort_image(truth, np.eye(4), **geo)
            Am = distort_image(ddf.mask[index], np.eye(4), **geo)[1] if ddf.mask is not None else None
        else:
            p, A, At, Am = None, np.eye(4), np.eye(4), np.eye(4)
        q.update(p=p, A=A, At=At, Am=Am)
        if p is None:
            return q
        # imgaug's two geometric augmenters, each a resampling of its own as in the reference: piecewise affine (augment.py:344-347; a 2 x 2 grid:
        # the four corners move by Normal(0, scale) of the extent, clipped to the image; two triangles), then the elastic transform (:349-353).
        # ONE set of moved corners / ONE in-plane displacement field for every slice and for image (bilinear), truth, previous-slice truth and mask
        if p["piecewise_affine_scale"] > 0:
            h, w = float(ps[0]), float(ps[1])
            grid = np.array([[0, 0], [0, w], [h, 0], [h, w]], dtype=np.float64)
            corners = grid + self.host_rng.normal(0.0, p["piecewise_affine_scale"], size=(4, 2)) * np.array([h, w])
            corners[:, 0] = np.clip(corners[:, 0], 0, h - 1)
            corners[:, 1] = np.clip(corners[:, 1], 0, w - 1)
            q["corners"] = corners
        if p["elastic_transform_scale"] > 0:
            q["elastic_rng"] = self.ops.elastic_ksize(float(self.augment["elastic_transform"]["sigma"])) <= self.ops.ELASTIC_RNG_KMAX
            q["elastic_seq"] = self._next_seq() if q["elastic_rng"] else -1
        q["need_intensity"] = bool(p["contrast"] is not None or p["intensity_multiplication"] != 1 or p["apply_speckle_noise"] or p["apply_gaussian_noise"]
                                   or p["apply_gaussian_filter"] or p["apply_poisson_noise"] or p["coarse_dropout"])
        if p["apply_poisson_noise"]:
            q["shot_seq"] = self._next_seq()
        if p["apply_speckle_noise"]:
            q["speckle_seq"] = self._next_seq()
        if p["apply_gaussian_noise"]:
            q["gaussian_seq"] = self._next_seq()
        if p["coarse_dropout"]:
            # reference augment.py:373-375 (last step): imgaug CoarseDropout(p=rate, size_percent, per_channel) in a [0, 255] scaling
            q["grid"] = _coarse_grid((ps[0], ps[1]), self.augment["coarse_dropout"]["size_percent"], self.host_rng)
            q["dropout_seq"] = self._next_seq()
        return q

    def sample_into(self, index, x_slot, y_slot, m_slot=None):
        """one patch: x_slot: float32 view (X, Y, n_chan) of the batch tensor; y_slot: uint8 view (X, Y, truth_size); m_slot: float32 view
        (X, Y, truth_size) for the distance mask, when the data file has masks"""
        self.launch_one(self.plan(index), x_slot, y_slot, m_slot)

    def batchable(self, plans):
        """can launch_batch take these plans?  Not: previous-slice truth channels, piecewise affine, a Gaussian filter on some patch, an
        elastic kernel wider than the one-launch field kernel - those go patch by patch (launch_one)"""
        if self.prev_truth_index is not None or self.n_chan != self.patch_shape[2]:
            return False
        return all(q["corners"] is None and q["elastic_seq"] >= 0 and not (q["p"] is not None and q["p"]["apply_gaussian_filter"]) for q in plans)

    def launch(self, plans, x, y, m=None):
        """x (B, X, Y, n_chan) float32, y (B, X, Y, truth_size) uint8, m like y in float32 or None: the patches of `plans`, in order"""
        if self.batched and self.batchable(plans):
            self.launch_batch(plans, x, y, m)
        else:
            for b, q in enumerate(plans):
                self.launch_one(q, x[b], y[b], None if m is None else m[b])

    def launch_batch(self, plans, x, y, m=None):
        """the patches of a batch with ONE launch per step of the chain (fmri_*_batch): 2-3 gathers, the elastic fields and 2-3 warps, the
        min / max, then only the intensity steps some patch of the batch drew.  Same arithmetic, same draws as launch_one patch by patch
        (tests/test_gpu_augment.py)."""
        ops, ddf, torch, ps = self.ops, self.ddf, self.torch, self.patch_shape
        B = len(plans)
        idx = [q["index"] for q in plans]
        tshape = (ps[0], ps[1], self.truth_size)
        warped = any(q["elastic_seq"] for q in plans)

        def target(dst, shape, dtype):
            return dst if not warped else torch.empty((B,) + tuple(shape), device=dst.device, dtype=dtype)

        d = None
        if warped:
            d = ops.elastic_fields_rng_batch((ps[0], ps[1]), [q["p"]["elastic_transform_scale"] if q["elastic_seq"] else 0.0 for q in plans],
                                             self.augment["elastic_transform"]["sigma"], self.seed, [q["elastic_seq"] for q in plans], device=x.device)
        corners_t = [(q["corner"][0], q["corner"][1], q["corner"][2] + self.truth_index) for q in plans]
        # image: trilinear, outside = the volume's minimum; labels: nearest, outside = 0 (identity affine = the plain crop)
        xt = target(x, ps, torch.float32)
        ops.affine_sample_batch([ddf.data[i] for i in idx], [q["A"] for q in plans], [q["corner"] for q in plans], ps, xt, 1, [ddf.min[i] for i in idx])
        yt = target(y, tshape, torch.uint8)
        ops.affine_sample_batch([ddf.truth[i] for i in idx], [q["At"] for q in plans], corners_t, tshape, yt, 0, [0.0] * B)
        if warped:
            ops.elastic_warp_batch(xt, d, 1, x)
            ops.elastic_warp_batch(yt, d, 0, y)
        if m is not None:
            # augmented: outside the mask = 0 (interpolate_affine_range, cval 0); plain crop: edge values
            src = [ddf.mask[q["index"]] if q["p"] is not None else ddf.mask_for_crops(q["index"]) for q in plans]
            mt = target(m, tshape, torch.float32)
            ops.affine_sample_batch(src, [q["Am"] for q in plans], corners_t, tshape, mt, 0, [0.0] * B)
            if warped:
                ops.elastic_warp_batch(mt, d, 0, m)
        if not any(q["p"] is not None and q["need_intensity"] for q in plans):
            return
        stats, ws = self._workspace(B)
        ops.minmax_ws_batch(x, stats, ws)
        params = []
        for q in plans:
            p = q["p"]
            if p is None or (p["contrast"] is None and p["intensity_multiplication"] == 1):
                params.append((0, 0.0, 0.0, 1.0))
            else:
                lo, hi = p["contrast"] if p["contrast"] is not None else (0.0, 0.0)
                params.append((2 if p["contrast"] is not None else 1, lo, hi, p["intensity_multiplication"]))
        if any(r[0] for r in params):
            ops.rescale_intensity_ws_batch(x, stats, ws, params)
        # order of reference augment.py:354-367: (gaussian filter,) shot (poisson) noise, speckle, gaussian noise; coarse dropout last
        if any(q["shot_seq"] for q in plans):
            ops.shot_noise_rng_batch(x, stats, ws, self.seed, [q["shot_seq"] for q in plans])
        for key, aug_key, kind in (("speckle_seq", "speckle_noise", 1), ("gaussian_seq", "gaussian_noise", 0)):
            if any(q[key] for q in plans):
                ops.noise_rng_batch(x, stats, ws, kind, self.augment[aug_key]["sigma"], self.seed, [q[key] for q in plans])
        if any(q["dropout_seq"] for q in plans):
            cd = self.augment["coarse_dropout"]
            ops.coarse_dropout_rng_batch(x, [q["grid"] for q in plans], cd["rate"], stats, bool(cd.get("per_channel", True)), self.seed,
                                         [q["dropout_seq"] for q in plans])

    def _workspace(self, B):
        if self.stats_b is None or self.stats_b.shape[0] < B:
            self.stats_b, self.ws_b = self.ops.aug_workspace(self.ddf.device, max(B, 2))
        return self.stats_b[:B], self.ws_b[:B]

    def launch_one(self, q, x_slot, y_slot, m_slot=None):
        """one planned patch, launch by launch (every configuration; the batch path's reference in the tests)"""
        ops, ddf = self.ops, self.ddf
        index, corner, p, A, At, Am = q["index"], q["corner"], q["p"], q["A"], q["At"], q["Am"]
        data, truth = ddf.data[index], ddf.truth[index]
        ps = self.patch_shape
        zt = corner[2] + self.truth_index
        elastic, corners = None, q["corners"]
        if q["elastic_seq"]:
            sigma = self.augment["elastic_transform"]["sigma"]
            if q["elastic_rng"]:
                elastic = ops.elastic_fields_rng((ps[0], ps[1]), p["elastic_transform_scale"], sigma, self.seed, q["elastic_seq"], device=ddf.device)
            else:
                elastic = ops.elastic_fields((ps[0], ps[1]), p["elastic_transform_scale"], sigma, generator=self.gen)
        warped = elastic is not None or corners is not None

        def target(slot, shape, dtype):
            return slot if not warped else self.torch.empty(shape, device=slot.device, dtype=dtype)

        def settle(tmp, slot, order):
            if corners is not None:
                tmp = ops.piecewise_affine(tmp, corners, order, slot if elastic is None else self.torch.empty_like(tmp))
            if elastic is not None:
                ops.elastic_warp(tmp, elastic[0], elastic[1], order, slot)

        tshape = (ps[0], ps[1], self.truth_size)
        # image: trilinear, outside = the volume's minimum; labels: nearest, outside = 0 (identity affine = the plain crop)
        xt = target(x_slot[..., :ps[2]], ps, self.torch.float32)
        ops.affine_sample(data, A, corner, ps, xt, order=1, cval=ddf.min[index], out_ld=self.n_chan if not warped else ps[2])
        settle(xt, x_slot[..., :ps[2]], 1)
        yt = target(y_slot, tshape, self.torch.uint8)
        ops.affine_sample(truth, At, (corner[0], corner[1], zt), tshape, yt, order=0, cval=0.0, out_ld=self.truth_size)
        settle(yt, y_slot, 0)
        if m_slot is not None:
            # augmented: outside the mask = 0 (interpolate_affine_range, cval 0); plain crop: edge values
            src = ddf.mask[index] if p is not None else ddf.mask_for_crops(index)
            mt = target(m_slot, tshape, self.torch.float32)
            ops.affine_sample(src, Am, (corner[0], corner[1], zt), tshape, mt, order=0, cval=0.0, out_ld=self.truth_size)
            settle(mt, m_slot, 0)
        img = x_slot if self.n_chan == ps[2] else None
        if p is not None:
            if q["need_intensity"]:
                if img is None:                            # image channels interleaved with the previous-slice truth: work on a copy
                    img = x_slot[..., :ps[2]].contiguous()
                # Every step below wants the image's range (rescale_intensity's out_range, MinMaxScaler): taken ONCE, then each kernel that
                # rewrites the image leaves the new range in `stats` for the next one; the noise draws are made in the kernels (Philox keyed by
                # noise_seed, one counter value per call) - fmri_hip.h "in-kernel draws"
                stats, ws = self.stats, self.ws
                ops.minmax_ws(img, stats, ws)
                if p["contrast"] is not None or p["intensity_multiplication"] != 1:
                    lo, hi = p["contrast"] if p["contrast"] is not None else (0.0, 0.0)
                    ops.rescale_intensity_ws(img, stats, ws, p["contrast"] is not None, lo, hi, p["intensity_multiplication"])
                # order of reference augment.py:354-367: gaussian filter, shot (poisson) noise, speckle, gaussian noise
                if p["apply_gaussian_filter"]:
                    smooth = ops.gaussian_filter_f32(img, p["gaussian_sigma"])
                    if smooth is not img:
                        img.copy_(smooth)
                    ops.minmax_ws(img, stats, ws)
                if q["shot_seq"]:
                    ops.shot_noise_rng(img, stats, ws, self.seed, q["shot_seq"])
                for key, aug_key, kind in (("speckle_seq", "speckle_noise", 1), ("gaussian_seq", "gaussian_noise", 0)):
                    if q[key]:
                        ops.noise_rng(img, stats, ws, kind, self.augment[aug_key]["sigma"], self.seed, q[key])
                if q["dropout_seq"]:
                    cd = self.augment["coarse_dropout"]
                    ops.coarse_dropout_rng(img, q["grid"], cd["rate"], stats, bool(cd.get("per_channel", True)), self.seed, q["dropout_seq"])
                if img is not x_slot:
                    x_slot[..., :ps[2]] = img
        if self.prev_truth_index is not None:
            zp = corner[2] + self.prev_truth_index
            prev = self.torch.empty((ps[0], ps[1], self.prev_truth_size), device=x_slot.device, dtype=self.torch.float32)
            ops.affine_sample(truth, At, (corner[0], corner[1], zp), (ps[0], ps[1], self.prev_truth_size), prev, order=0, cval=0.0)
            if corners is not None:
                prev = ops.piecewise_affine(prev, corners, 0, self.torch.empty_like(prev))
            if elastic is not None:
                prev = ops.elastic_warp(prev, elastic[0], elastic[1], 0, self.torch.empty_like(prev))
            x_slot[..., ps[2]:] = prev


def device_data_generator(data_file, index_list, batch_size=1, n_labels=1, labels=None, augment=None, patch_shape=None,
                          shuffle_index_list=True, skip_blank=True, truth_index=-1, truth_size=1, truth_downsample=None, truth_crop=True,
                          categorical=True, prev_truth_index=None, prev_truth_size=None, drop_easy_patches=False, is3d=False,
                          samples_pad=3, strict=False, noise_seed=0, device="cuda", prefetch=0, batched=True):
    """Endless generator of (x, y) CUDA tensors.  `data_file`: a DeviceDataFile, or anything with .root.data / .root.truth
    (uploaded here).  3-D: x (N,1,X,Y,Z), y (N,1,X,Y,truth_size); 2-D: x (N,X,Y,C), y (N,X,Y,truth_size).  skip_blank and
    drop_easy_patches read one scalar back per patch (they decide on the host whether the patch is kept), everything else is
    enqueue-only.

    batched: launch the patches of a batch together, one launch per step of the sampling / augmentation chain (the default; configurations
    the batch kernels do not cover go patch by patch on their own); False: always patch by patch - same draws, same batches.

    prefetch (0 | n): n > 0 starts a producer thread with a HIP stream of its own that keeps up to n batches ready (the role of Keras'
    GeneratorEnqueuer behind the reference's fit_generator, training.py:110-124).  A batch is 11-13 short gather / element-wise launches
    (~0.34 ms of device time with the reference's default augmentation) which then run BESIDE the consumer's training step; a consumer
    that reads a scalar back every step (`train_on_batch`) leaves no other way to overlap them.  The consumer's current stream waits for the batch's
    event before the yield; the tensors are registered with that stream (record_stream) so the allocator does not recycle them under a
    step still in flight.  Draw order, and therefore every batch, is the same as with prefetch=0 (tests/test_gpu_augment.py); the draws
    of batch k+1 ... k+n come from numpy's global state BEFORE batch k is handed out, so a caller that re-seeds between batches wants
    prefetch=0 (the default).  With prefetch > 0 those draws happen on the producer THREAD against the process-global numpy / `random` states:
    anything else in the process that draws from them (a validation generator on the main thread, for one) interleaves with the
    producer by timing, and the batches of both stop being reproducible - seeded reproducibility holds for prefetch=0 only.
    `Model.fit_generator` has its own producer thread and copy stream: leave prefetch at 0 there."""
    import torch
    if truth_downsample is not None and truth_downsample > 1:
        raise NotImplementedError("truth_downsample is not part of the device generator")
    if patch_shape is None:
        raise ValueError("the device generator samples patches; patch_shape is required")
    ddf = data_file if isinstance(data_file, DeviceDataFile) else DeviceDataFile(data_file, patch_shape, samples_pad, truth_downsample,
                                                                                 indices=sorted(set(index_list)), device=device)
    sampler = _Sampler(ddf, patch_shape, augment, truth_index, truth_size, prev_truth_index, prev_truth_size, strict, noise_seed)
    sampler.batched = bool(batched)
    index_generator = random_list_generator(index_list) if shuffle_index_list else list_generator(index_list)
    ps = sampler.patch_shape

    last_done = [None]

    def produce():
        # the sampler's range / workspace buffers are reused batch after batch: a caller that pulls batches under changing streams must not
        # start batch k + 1's chain while batch k's still runs on another stream (same stream: ordered anyway, the wait is free)
        cur = torch.cuda.current_stream()
        if last_done[0] is not None:
            cur.wait_event(last_done[0])
        out = produce_batch()
        last_done[0] = torch.cuda.Event()
        last_done[0].record(cur)
        return out

    def produce_batch():
        x = torch.empty((batch_size, ps[0], ps[1], sampler.n_chan), device=ddf.device, dtype=torch.float32)
        y = torch.empty((batch_size, ps[0], ps[1], truth_size), device=ddf.device, dtype=torch.uint8)
        m = torch.empty((batch_size, ps[0], ps[1], truth_size), device=ddf.device, dtype=torch.float32) if ddf.mask is not None else None
        filled = 0
        while filled < batch_size and drop_easy_patches:
            # the keep / drop draw of a patch comes from numpy's stream BETWEEN its draws and the next patch's: patch by patch, as the reference
            index = next(index_generator)
            sampler.sample_into(index, x[filled], y[filled], None if m is None else m[filled])
            truth_mean = float(y[filled][16:-16, 16:-16, :].float().mean().item())
            if 1 - np.abs(truth_mean - 0.5) < np.random.random():
                continue
            if skip_blank and not bool(y[filled].any().item()):
                continue
            filled += 1
        while filled < batch_size:
            # the patches still missing, planned in the reference's draw order and launched together; blank ones (skip_blank) are dropped behind
            # ONE read-back, the kept ones move up in order and the next round plans the rest: the same patches in the same slots as one by one
            need = batch_size - filled
            plans = [sampler.plan(next(index_generator)) for _ in range(need)]
            sampler.launch(plans, x[filled:], y[filled:], None if m is None else m[filled:])
            keep = list(range(need))
            if skip_blank:
                keep = [i for i, f in enumerate(y[filled:].reshape(need, -1).any(dim=1).tolist()) if f]
            for dst, src in enumerate(keep):
                if dst != src:
                    for t in (x, y, m):
                        if t is not None:
                            t[filled + dst].copy_(t[filled + src])
            filled += len(keep)
        yy = y
        if categorical:
            # keras.utils.to_categorical(y, 2) (reference generator.py:390-391): a trailing axis of size 1 is dropped before the one-hot
            # axis is appended - (N,X,Y,1) -> (N,X,Y,2), (N,X,Y,T>1) -> (N,X,Y,T,2) - float32
            yc = y.squeeze(-1) if y.shape[-1] == 1 else y
            yy = torch.stack([1 - yc, yc], dim=-1).float()
        if is3d:
            xo, yo, mo = x.unsqueeze(1), yy.unsqueeze(1), (None if m is None else m.unsqueeze(1))
        else:
            xo, yo, mo = x, yy, m
        return xo, yo, mo

    if not prefetch:
        while True:
            xo, yo, mo = produce()
            yield (xo if mo is None else [xo, mo]), yo

    import queue
    import threading
    dev_index = torch.device(ddf.device).index if torch.device(ddf.device).index is not None else torch.cuda.current_device()
    side = torch.cuda.Stream(device=dev_index)
    side.wait_stream(torch.cuda.current_stream())      # the volumes were uploaded on the caller's stream: order the side stream behind it once
    ready_q = queue.Queue(maxsize=int(prefetch))
    stop = threading.Event()

    def put(item):
        while not stop.is_set():
            try:
                ready_q.put(item, timeout=0.05)
                return
            except queue.Full:
                pass

    def producer():
        try:
            torch.cuda.set_device(dev_index)
            with torch.cuda.stream(side):
                while not stop.is_set():
                    batch = produce()
                    ready = torch.cuda.Event()
                    ready.record(side)
                    put((batch, ready, None))
        except BaseException as e:                     # handed to the consumer: a generator that dies silently would hang the training loop
            put((None, None, e))

    worker = threading.Thread(target=producer, name="device_data_generator", daemon=True)
    worker.start()
    try:
        while True:
            batch, ready, err = ready_q.get()
            if err is not None:
                raise err
            xo, yo, mo = batch
            consumer = torch.cuda.current_stream()
            consumer.wait_event(ready)
            for t in (xo, yo, mo):
                if t is not None:
                    t.record_stream(consumer)
            yield (xo if mo is None else [xo, mo]), yo
    finally:                                           # generator closed or collected: stop the producer, let it leave its put()
        stop.set()
        worker.join(timeout=10.0)
