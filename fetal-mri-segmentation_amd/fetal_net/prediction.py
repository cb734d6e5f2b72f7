"""Sliding-window inference with the reference surface (reference fetal_net/prediction.py:88-210, :277-361).

`patch_wise_prediction` keeps the reference's geometry exactly (overlap interpolation, 1st-percentile padding, tile
index set, mean of overlapping tiles in float64).  With a model built by this package the whole tile loop stays on the
MI355X: the padded volume is uploaded once, tiles are gathered on the device (fmri_tile_gather), pushed through the
engine and overlap-added in float64 on the device (fmri_tile_scatter_accumulate / fmri_tile_finalize); one hipGraph per
tile-batch size replays gather -> network -> scatter.  Any other object with `.output_shape` / `.predict(ndarray)` (the
reference's duck-typing contract) is driven through the same geometry with host-side tiles.
"""
import gc
import itertools
import os

import numpy as np

from .augment import contrast_augment, generate_permutation_keys, permute_data, reverse_permute_data
from .engine_model import Model
from .utils.patches import get_patch_from_3d_data
from .utils.threaded_generator import ThreadedGenerator


def get_set_of_patch_indices_full(start, stop, step):
    axes = []
    for a, b, s in zip(start, stop, step):
        idx = list(range(int(a), int(b) + 1, int(s)))
        if int(b) % int(s) > 0:
            idx.append(int(b))
        axes.append(idx)
    return np.array(list(itertools.product(*axes)))


def batch_iterator(indices, batch_size, data_0, patch_shape, truth_0, prev_truth_index, truth_patch_shape):
    i = 0
    while i < len(indices):
        batch, curr = [], []
        while len(batch) < batch_size and i < len(indices):
            ix = indices[i]
            patch = get_patch_from_3d_data(data_0, patch_shape=patch_shape, patch_index=ix)
            if truth_0 is not None:
                t_ix = list(ix[:2]) + [ix[2] + prev_truth_index]
                patch = np.concatenate([patch, get_patch_from_3d_data(truth_0, patch_shape=truth_patch_shape, patch_index=t_ix)],
                                       axis=-1)
            batch.append(patch)
            curr.append(ix)
            i += 1
        yield [batch, curr]


def _half_pads(delta):
    return [(int(np.ceil(d / 2)), int(np.floor(d / 2))) for d in delta]


def _geometry(model, data, patch_shape, overlap_factor):
    out_shape = model.output_shape
    is3d = int(np.sum(np.array(out_shape[1:]) > 1)) > 2
    prediction_shape = tuple(out_shape[-3:]) if is3d else tuple(out_shape[-3:-1]) + (1,)
    min_overlap = np.subtract(patch_shape, prediction_shape)
    max_overlap = np.subtract(patch_shape, (1, 1, 1))
    overlap = min_overlap + (overlap_factor * (max_overlap - min_overlap)).astype(int)
    pad0 = _half_pads(np.subtract(patch_shape, prediction_shape))
    # the reference pads with the 1st percentile unconditionally (prediction.py:138-146); with zero pad widths np.pad returns the data
    # unchanged, so the percentile (a sort-sized pass over the whole volume, twice) is only computed when something is padded
    data_0 = data[0]
    if np.sum(pad0) > 0:
        data_0 = np.pad(data_0, pad0, mode='constant', constant_values=np.percentile(data_0, q=1))
    pad_for_fit = _half_pads(np.maximum(np.subtract(patch_shape, data_0.shape), 0))
    if np.sum(pad_for_fit) > 0:
        data_0 = np.pad(data_0, pad_for_fit, 'constant', constant_values=np.percentile(data_0, q=1))
    indices = get_set_of_patch_indices_full((0, 0, 0), np.subtract(data_0.shape, patch_shape), np.subtract(patch_shape, overlap))
    data_shape = list(np.asarray(data.shape[-3:]) + np.sum(pad_for_fit, -1))
    data_shape += [out_shape[1]] if is3d else [out_shape[-1]]
    return is3d, pad0, pad_for_fit, data_0, indices, data_shape


def _unpad(arr, pad_for_fit):
    if np.sum(pad_for_fit) > 0:
        sl = tuple(slice(p[0] if p[0] else None, -p[1] if p[1] else None) for p in pad_for_fit)
        arr = arr[sl]
    return arr


def patch_wise_prediction(model, data, patch_shape, overlap_factor=0, batch_size=5, permute=False, truth_data=None,
                          prev_truth_index=None, prev_truth_size=None):
    """data (1,X,Y,Z) -> (X,Y,Z,C) float64 mean of all tiles covering each voxel."""
    is3d, pad0, pad_for_fit, data_0, indices, data_shape = _geometry(model, data, patch_shape, overlap_factor)
    on_device = isinstance(model, Model) and truth_data is None and not permute and model._unsupported is None and \
        (is3d or getattr(model, "_input_layout", "") == "channels_last_2d")
    if on_device:
        out, count_ok = _device_overlap_add(model, data_0, indices, patch_shape, batch_size, data_shape, is3d)
        assert count_ok, 'Found zeros in count'
        out = _unpad(out, pad_for_fit)
        assert np.array_equal(out.shape[:-1], data[0].shape), 'prediction shape wrong'
        return out

    if truth_data is not None:
        truth_0 = np.pad(truth_data[0], pad0, mode='constant', constant_values=0)
        truth_0 = np.pad(truth_0, pad_for_fit, 'constant', constant_values=0)
        truth_patch_shape = list(patch_shape[:2]) + [prev_truth_size]
    else:
        truth_0, truth_patch_shape = None, None
    tb_iter = iter(ThreadedGenerator(batch_iterator(indices, batch_size, data_0, patch_shape, truth_0, prev_truth_index,
                                                    truth_patch_shape), queue_maxsize=50))
    predicted_output = np.zeros(data_shape)
    predicted_count = np.zeros(data_shape, dtype=np.int16)
    for curr_batch, batch_indices in tb_iter:
        curr_batch = np.asarray(curr_batch)
        if is3d:
            curr_batch = np.expand_dims(curr_batch, 1)
        prediction = predict(model, curr_batch, permute=permute)
        prediction = prediction.transpose([0, 2, 3, 4, 1]) if is3d else np.expand_dims(prediction, -2)
        for patch, (x, y, z) in zip(prediction, batch_indices):
            xl, yl, zl = patch.shape[:-1]
            predicted_output[x:x + xl, y:y + yl, z:z + zl, :] += patch
            predicted_count[x:x + xl, y:y + yl, z:z + zl] += 1
    assert np.all(predicted_count > 0), 'Found zeros in count'
    predicted_output, predicted_count = _unpad(predicted_output, pad_for_fit), _unpad(predicted_count, pad_for_fit)
    assert np.array_equal(predicted_count.shape[:-1], data[0].shape), 'prediction shape wrong'
    return predicted_output / predicted_count


def _device_overlap_add(model, data_0, indices, patch_shape, batch_size, data_shape, is3d=True):
    """Upload once, then per tile batch: gather -> network -> float64 overlap-add, all on the device.  The static buffers
    (volume, accumulators, tile batch, index list) and one captured hipGraph per distinct batch size are cached on the
    model and re-used for every following volume of the same padded shape.  2-D models: a tile (px, py, slices) IS the
    channels-last input of the network (the slice stack is the channel axis) and its output is one slice (px, py, 1)."""
    import torch
    from fmri_hip import ops
    patch = tuple(int(p) for p in patch_shape)
    vshape = tuple(int(s) for s in data_0.shape)
    ashape = tuple(int(s) for s in data_shape)
    use_graph = os.environ.get("FMRI_HIPGRAPH", "1") == "1"
    n = len(indices)
    if not is3d:
        # 2-D: a tile is a 5-slice stack and the caller's batch (reference default 5) is far too small to fill the device - the MFMA
        # kernels also want the slice count in multiples of 4.  The overlap-add does not depend on how tiles are grouped (float64 sums)
        batch_size = max(batch_size, 64)
    else:
        # 3-D: the caller's batch (reference default 5) leaves a 1-tile remainder on the usual 36-tile volume and every forward pass pays its
        # kernel tails; tiles are grouped by at least 12 and the groups are evened out (36 -> 3 x 12: device loop 52.4 -> 50.4 ms).  As above
        # the result does not depend on the grouping.
        groups = -(-n // max(batch_size, 12))
        batch_size = -(-n // groups)
    sizes = sorted({min(batch_size, n - i) for i in range(0, n, batch_size)}, reverse=True)
    key = (vshape, ashape, patch, tuple(sizes), use_graph)
    st = model.__dict__.get("_tile_state")
    if st is None or st["key"] != key:
        st = dict(key=key, vol=torch.empty(vshape, dtype=torch.float32, device="cuda"),
                  acc=torch.zeros(ashape, dtype=torch.float64, device="cuda"),
                  cnt=torch.zeros(ashape[:3], dtype=torch.int32, device="cuda"), per_b={})
        for B in sizes:
            eng = model.engine(B)
            # 3-D: (B, px, py, pz, 1) = NDHWC with one channel; 2-D: (1, B, px, py, slices) = the planar layout, slices as channels
            tshape = (B,) + patch + (1,) if is3d else (1, B) + patch
            opatch = patch if is3d else (patch[0], patch[1], 1)
            pb = dict(idx=torch.zeros((B, 3), dtype=torch.int32, device="cuda"),
                      tiles=torch.empty(tshape, dtype=eng.dtype, device="cuda"), graph=None)

            def body(pb=pb, B=B, opatch=opatch):
                e = model.engine(B)
                ops.tile_gather(st["vol"], pb["idx"], patch, pb["tiles"])
                e.predict(pb["tiles"])
                ops.tile_scatter_accumulate(e.probs, pb["idx"], opatch, st["acc"], st["cnt"])

            pb["body"] = body
            if use_graph:
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    body()                                   # warm-up outside capture (lazy allocations, module loads)
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                # No cyclic garbage collection while the stream is capturing: a collector run that happens to free an old CUDAGraph /
                # stream / event (the per-batch closures above sit in reference cycles, so earlier tile states die by the cycle
                # collector, at an arbitrary allocation) destroys HIP objects mid-capture, which HIP refuses - and a failing destructor
                # aborts the process.  Collect now, outside the capture, and keep the collector off until the capture has ended.
                gc_was_on = gc.isenabled()
                gc.collect()
                gc.disable()
                try:
                    with torch.cuda.graph(g):
                        body()
                finally:
                    if gc_was_on:
                        gc.enable()
                pb["graph"] = g
            st["per_b"][B] = pb
        model.__dict__["_tile_state"] = st
    # ---- the volume goes up and the result comes down in x-slabs, pipelined with the tile groups (23 % of a volume's
    # time was host work - a pageable 42 MB upload in front of the first tile and a pageable 84 MB float64 download behind the last).
    # Tiles are ordered x-major (itertools.product), so tile group g only needs the x-range up to its last tile's end, and every voxel
    # in front of the NEXT group's first tile is final once group g has run: slab k of the volume is copied into a pinned buffer
    # (converting copy for non-fp32 callers) and uploaded on a copy stream while group k - 1 computes; finished slabs are divided by
    # their counts and downloaded into a pinned result while the later groups compute.  The result array is backed by pinned memory
    # that is this call's own (torch's caching host allocator recycles it when the caller drops the array).
    if st.get("up_stream") is None:
        st["up_stream"], st["down_stream"] = torch.cuda.Stream(), torch.cuda.Stream()
        st["pin_in"] = torch.empty(vshape, dtype=torch.float32, pin_memory=True)
    up, down = st["up_stream"], st["down_stream"]
    main = torch.cuda.current_stream()
    st["acc"].zero_()
    st["cnt"].zero_()
    idx_all = torch.from_numpy(np.ascontiguousarray(indices, dtype=np.int32)).cuda()
    # A captured graph holds no dependency on the engine's weight-repack side stream (it was captured after a warm-up that had already
    # joined it), but an optimizer step since then may have left a repack of the deeper layers' filter images in flight there: make THIS
    # stream wait for it before any replay reads those images.
    for B in sizes:
        join = getattr(model.engine(B), "_join_packs", None)
        if join is not None:
            join()
    out_dev = torch.empty_like(st["acc"])
    out_host = torch.empty(ashape, dtype=torch.float64, pin_memory=True)
    bad = torch.zeros(1, dtype=torch.int32, device="cuda")
    pin_np, src = st["pin_in"].numpy(), np.asarray(data_0)
    starts_x = [int(ix[0]) for ix in indices]
    up_to = 0                                   # x-planes uploaded so far
    done_to = 0                                 # x-planes finalised and on their way down
    up.wait_stream(main)                        # (the previous volume's gathers have read st["vol"]; uploads never wait for this volume's tiles)
    for i in range(0, n, batch_size):
        hi = min(n, i + batch_size)
        need = min(vshape[0], max(starts_x[i:hi]) + patch[0])
        if need > up_to:
            np.copyto(pin_np[up_to:need], src[up_to:need], casting="unsafe")
            with torch.cuda.stream(up):
                st["vol"][up_to:need].copy_(st["pin_in"][up_to:need], non_blocking=True)
            up_to = need
            main.wait_stream(up)
        bidx = idx_all[i:hi]
        pb = st["per_b"][int(bidx.shape[0])]
        pb["idx"].copy_(bidx)
        if pb["graph"] is not None:
            pb["graph"].replay()
        else:
            pb["body"]()
        final = ashape[0] if hi >= n else min(ashape[0], min(starts_x[hi:]))
        if final > done_to:
            ops.tile_finalize(st["acc"][done_to:final], st["cnt"][done_to:final], out_dev[done_to:final], bad)
            down.wait_stream(main)
            with torch.cuda.stream(down):
                out_host[done_to:final].copy_(out_dev[done_to:final], non_blocking=True)
            done_to = final
    down.synchronize()
    main.synchronize()
    return out_host.numpy(), int(bad.item()) == 0


def predict(model, data, permute=False):
    """reference prediction.py:354-361"""
    if permute:
        return np.asarray([predict_with_permutations(model, data[b]) for b in range(data.shape[0])])
    return model.predict(data)


def predict_with_permutations(model, data):
    """mean over the 48 permutation keys of un-permuted predictions (reference prediction.py:362-367); data (C, X, Y, Z).
    Keys that act identically (only rotate_y and the three flips act) are predicted once and weighted by their multiplicity."""
    groups = {}
    for key in generate_permutation_keys():
        groups.setdefault((key[0][0], key[1], key[2], key[3]), []).append(key)
    total, n = None, 0
    for keys in groups.values():
        pred = reverse_permute_data(model.predict(permute_data(data, keys[0])[np.newaxis])[0], keys[0])
        total = pred * len(keys) if total is None else total + pred * len(keys)
        n += len(keys)
    return total / n


def flip_it(data_, axes):
    """mirror along every axis in `axes` (an involution: applying it twice restores the input)"""
    axes = tuple(int(a) for a in axes)
    return np.flip(data_, axis=axes) if axes else data_


def predict_flips(data, model, overlap_factor, config):
    """the 8 axis-flip variants, each predicted patch-wise and flipped back; returns the list (reference prediction.py:65-85)"""
    patch_shape = list(config["patch_shape"]) + [config["patch_depth"]]
    predictions = []
    for r in range(4):
        for axes in itertools.combinations([0, 1, 2], r):
            flipped = flip_it(data, axes)
            pred = patch_wise_prediction(model=model, data=np.expand_dims(flipped.squeeze(), 0), overlap_factor=overlap_factor,
                                         patch_shape=patch_shape).squeeze()
            predictions.append(flip_it(pred, axes).squeeze())
    return predictions


class _TTAVariant:
    """One random test-time variant of a volume: intensity window, mirror axes, optional x/y swap, in-plane rotation angle.
    `draw` consumes numpy's global RNG in the order the reference's loop does (two window draws, the angle, three mirror coins, the
    swap coin: reference prediction.py:34-42) - a seeded run therefore produces the reference's variants (tests/golden/augment_golden.*)."""
    __slots__ = ("lo", "hi", "angle", "mirror", "swap")

    @classmethod
    def draw(cls, vmin, vmax, jitter=0.10, max_angle=30):
        v = cls()
        span = vmax - vmin
        v.lo = vmin + jitter * np.random.uniform(-1, 1) * span
        v.hi = vmax + jitter * np.random.uniform(-1, 1) * span
        v.angle = np.random.uniform(-max_angle, max_angle)
        v.mirror = tuple(np.flatnonzero(np.random.choice([True, False], size=3)))
        v.swap = bool(np.random.choice([True, False]))
        return v

    @staticmethod
    def _rotate(vol, angle, order, reshape):
        """scipy.ndimage.rotate(vol, angle, order=order, reshape=reshape) - on the device through fetal_net.spline_rotate (scipy's own
        arithmetic in float64 on torch tensors, equal to 1e-12; a 160x256x256 volume takes scipy 1-2 s per rotation, 64 rotations per
        predict_augment call) when a GPU is there, FMRI_TTA_TORCH_ROTATE=1 forces the torch form on the host, =0 scipy"""
        mode = os.environ.get("FMRI_TTA_TORCH_ROTATE", "auto")
        import torch
        if mode == "0" or (mode == "auto" and not torch.cuda.is_available()) or vol.ndim != 3:
            from scipy import ndimage
            return ndimage.rotate(vol, angle, order=order, reshape=reshape)
        from .spline_rotate import rotate
        dev = "cuda" if torch.cuda.is_available() else "cpu"
        return rotate(torch.from_numpy(np.ascontiguousarray(vol, dtype=np.float64)).to(dev), angle, order=order, reshape=reshape).cpu().numpy()

    def forward(self, vol):
        """volume -> variant: window, mirror, swap, rotate (quadratic spline, shape kept)"""
        out = flip_it(contrast_augment(vol, self.lo, self.hi), self.mirror)
        if self.swap:
            out = np.swapaxes(out, 0, 1)
        return self._rotate(out, self.angle, 2, False)

    def inverse(self, pred):
        """prediction of the variant -> frame of the original volume.  The back-rotation keeps scipy's defaults (cubic spline,
        reshape=True), as the reference's does - the result can be larger than the input when angle != 0."""
        out = self._rotate(pred, -self.angle, 3, True)
        if self.swap:
            out = np.swapaxes(out, 0, 1)
        return flip_it(out, self.mirror)


def predict_augment(data, model, overlap_factor, patch_shape, num_augments=32):
    """Random intensity / mirror / swap / rotation variants of the volume (reference prediction.py:25-62), each predicted patch-wise on
    the device tile loop and mapped back; the volume-sized spline rotations run on the device too (fetal_net.spline_rotate).  Like the
    reference, stacking succeeds only when all back-rotated predictions end up with one shape (num_augments = 1, or equal angles)."""
    vmin, vmax = data.min(), data.max()
    vol = data.squeeze()
    out = []
    for _ in range(num_augments):
        variant = _TTAVariant.draw(vmin, vmax)
        pred = patch_wise_prediction(model=model, data=variant.forward(vol)[np.newaxis], overlap_factor=overlap_factor,
                                     patch_shape=patch_shape).squeeze()
        out.append(variant.inverse(pred).squeeze())
    return np.stack(out, axis=0)


def run_validation_case(data_index, output_dir, model, data_file, training_modalities, patch_shape, overlap_factor=0,
                        permute=False, prev_truth_index=None, prev_truth_size=None, use_augmentations=False):
    """Predict one case of an opened data file (any object with `.root.data[i]` / `.root.truth[i]`) and write
    data_<modality>.nii.gz, truth.nii.gz, prediction.nii.gz under output_dir (reference prediction.py:277-330)."""
    from .utils.nifti import save_nifti
    if not os.path.exists(output_dir):
        os.makedirs(output_dir)
    test_data = np.asarray([data_file.root.data[data_index]])
    test_truth_data = np.asarray([data_file.root.truth[data_index]]) if prev_truth_index is not None else None
    for i, modality in enumerate(training_modalities):
        save_nifti(test_data[i], os.path.join(output_dir, "data_{0}.nii.gz".format(modality)))
    save_nifti(np.asarray(data_file.root.truth[data_index]), os.path.join(output_dir, "truth.nii.gz"))
    if tuple(patch_shape) == tuple(test_data.shape[-3:]):
        prediction = predict(model, test_data[:, np.newaxis] if test_data.ndim == 4 else test_data, permute=permute)
    elif use_augmentations:
        prediction = predict_augment(data=test_data, model=model, overlap_factor=overlap_factor, patch_shape=patch_shape)
    else:
        prediction = patch_wise_prediction(model=model, data=test_data, overlap_factor=overlap_factor, patch_shape=patch_shape,
                                           truth_data=test_truth_data, prev_truth_index=prev_truth_index,
                                           prev_truth_size=prev_truth_size)[np.newaxis]
    prediction = prediction.squeeze()
    filename = os.path.join(output_dir, "prediction.nii.gz")
    save_nifti(prediction, filename)
    return filename


def run_validation_cases(validation_keys_file, model_file, training_modalities, hdf5_file, patch_shape, output_dir=".",
                         overlap_factor=0, permute=False, prev_truth_index=None, prev_truth_size=None, use_augmentations=False,
                         data_file=None, model=None):
    """reference prediction.py:333-351.  `hdf5_file` is opened with `fetal_net.data.open_data_file` (plain-layout files through libhdf5,
    PyTables files through PyTables when installed); callers may instead pass an already opened duck-typed `data_file` (and a `model`)."""
    import glob
    import pickle
    from .training import load_old_model
    with open(validation_keys_file, "rb") as f:
        validation_indices = pickle.load(f)
    if model is None:
        candidates = glob.glob(model_file + '*.h5')
        model = load_old_model(max(candidates, key=os.path.getmtime))      # newest checkpoint (reference fetal/utils.py:42-43)
    own = data_file is None
    if own:
        from .data import open_data_file
        data_file = open_data_file(hdf5_file, "r")
    has_ids = 'subject_ids' in data_file.root

    def case_dir(index):
        label = data_file.root.subject_ids[index].decode('utf-8') if has_ids else "validation_case_{}".format(index)
        return os.path.join(output_dir, label)

    try:
        return [run_validation_case(data_index=i, output_dir=case_dir(i), model=model, data_file=data_file,
                                    training_modalities=training_modalities, overlap_factor=overlap_factor, permute=permute,
                                    patch_shape=patch_shape, prev_truth_index=prev_truth_index, prev_truth_size=prev_truth_size,
                                    use_augmentations=use_augmentations) for i in validation_indices]
    finally:
        if own:
            data_file.close()
