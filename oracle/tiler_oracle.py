"""numpy restatement of the reference sliding-window path (test infrastructure).

Follows reference fetal_net/prediction.py:88-95 (get_set_of_patch_indices_full), :98-114 (batch_iterator),
:118-210 (patch_wise_prediction) and fetal_net/utils/patches.py:57-91 (get_patch_from_3d_data + edge-pad fix).
PINNED: tests/golden/tiler_golden.{json,npz} hold outputs of the reference's own functions run under stubs with a
deterministic fake model (tests/golden/make_fixtures.py).
"""
import itertools

import numpy as np


def patch_indices_full(start, stop, step):                   # prediction.py:88-95
    per_axis = []
    for a, b, s in zip(start, stop, step):
        idx = list(range(int(a), int(b) + 1, int(s)))
        if int(b) % int(s) > 0:
            idx.append(int(b))
        per_axis.append(idx)
    return np.array(list(itertools.product(*per_axis)))


def get_patch(data, patch_shape, patch_index):               # utils/patches.py:57-91
    idx = np.asarray(patch_index, dtype=np.int16).copy()
    ps = np.asarray(patch_shape)
    img = np.asarray(data.shape[-3:])
    if np.any(idx < 0) or np.any(idx + ps > img):
        before = np.abs((idx < 0) * idx)
        after = np.abs(((idx + ps) > img) * ((idx + ps) - img))
        pad = [[0, 0]] * (data.ndim - 3) + np.stack([before, after], axis=1).tolist()
        data = np.pad(data, pad, mode="edge")
        idx = idx + before
    return data[..., idx[0]:idx[0] + ps[0], idx[1]:idx[1] + ps[1], idx[2]:idx[2] + ps[2]]


def overlap_and_step(patch_shape, prediction_shape, overlap_factor):     # prediction.py:135-137
    min_overlap = np.subtract(patch_shape, prediction_shape)
    max_overlap = np.subtract(patch_shape, (1, 1, 1))
    overlap = min_overlap + (overlap_factor * (max_overlap - min_overlap)).astype(int)
    return overlap, np.subtract(patch_shape, overlap)


def patch_wise_prediction(model, data, patch_shape, overlap_factor=0, batch_size=5):
    """model: any object with .output_shape and .predict(ndarray).  data (1,X,Y,Z).  -> (X,Y,Z,C) float64."""
    out_shape = model.output_shape
    is3d = int(np.sum(np.array(out_shape[1:]) > 1)) > 2
    prediction_shape = tuple(out_shape[-3:]) if is3d else tuple(out_shape[-3:-1]) + (1,)
    overlap, step = overlap_and_step(patch_shape, prediction_shape, overlap_factor)
    halves = [(int(np.ceil(d / 2)), int(np.floor(d / 2))) for d in np.subtract(patch_shape, prediction_shape)]
    d0 = np.pad(data[0], halves, mode="constant", constant_values=np.percentile(data[0], q=1))
    pad_for_fit = [(int(np.ceil(d / 2)), int(np.floor(d / 2))) for d in np.maximum(np.subtract(patch_shape, d0.shape), 0)]
    d0 = np.pad(d0, pad_for_fit, "constant", constant_values=np.percentile(d0, q=1))
    indices = patch_indices_full((0, 0, 0), np.subtract(d0.shape, patch_shape), step)
    dshape = list(np.asarray(data.shape[-3:]) + np.sum(pad_for_fit, -1))
    dshape += [out_shape[1]] if is3d else [out_shape[-1]]
    acc = np.zeros(dshape)
    cnt = np.zeros(dshape, dtype=np.int16)
    for i in range(0, len(indices), batch_size):
        bidx = indices[i:i + batch_size]
        batch = np.asarray([get_patch(d0, patch_shape, ix) for ix in bidx])
        if is3d:
            batch = np.expand_dims(batch, 1)
        pred = np.asarray(model.predict(batch))
        pred = pred.transpose([0, 2, 3, 4, 1]) if is3d else np.expand_dims(pred, -2)
        for p, (x, y, z) in zip(pred, bidx):
            xl, yl, zl = p.shape[:-1]
            acc[x:x + xl, y:y + yl, z:z + zl, :] += p
            cnt[x:x + xl, y:y + yl, z:z + zl] += 1
    assert np.all(cnt > 0), "Found zeros in count"
    if np.sum(pad_for_fit) > 0:
        sl = tuple(slice(p[0] if p[0] else None, -p[1] if p[1] else None) for p in pad_for_fit)
        acc, cnt = acc[sl], cnt[sl]
    assert np.array_equal(cnt.shape[:-1], data[0].shape), "prediction shape wrong"
    return acc / cnt
