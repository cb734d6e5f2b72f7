#!/opt/conda/bin/python3.9
"""A checkpoint file written the way Keras 2.2.4 writes one - by h5py, through the call sequence of keras/engine/saving.py
(`_serialize_model`, `save_weights_to_hdf5_group`, `save_attributes_to_hdf5_group`, the H5Dict wrapper) restated below - for a small
`unet_model_3d`.  What this pins is the ENCODING an h5py-written file has (how h5py stores a bytes scalar, a list of bytes, a nested
dataset name, an int64 scalar): `fetal_net/keras_h5.py` must read it (tests/test_host_h5.py), and tests/keras_h5_read_like_keras.py - the
mirror image, Keras' LOADING sequence over h5py - must read what `keras_h5.save_model` writes.  Keras itself is not installable here; its
call sequence is restated from its published source, the h5py behaviour is the real thing (h5py 3.3.0 of the container's conda
interpreter; Keras 2.2.4 ran on h5py 2.x, whose storage rules for these calls are the same - only h5py 3 READS variable-length strings
back as str instead of bytes, which both readers tolerate).

    /opt/conda/bin/python3.9 tests/golden/make_keras_h5_fixture.py     ->  tests/golden/keras_like_golden.h5, keras_like_golden.npz
"""
import json
import os
import sys

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "fetal-mri-segmentation_amd"))
HDF5_OBJECT_HEADER_LIMIT = 64512


def save_attributes_to_hdf5_group(group, name, data):
    """keras/engine/saving.py:save_attributes_to_hdf5_group"""
    bad = [x for x in data if len(x) > HDF5_OBJECT_HEADER_LIMIT]
    if bad:
        raise RuntimeError("attribute item too large")
    data_npy = np.asarray(data)
    num_chunks = 1
    chunked = np.array_split(data_npy, num_chunks)
    while any(map(lambda x: x.nbytes > HDF5_OBJECT_HEADER_LIMIT, chunked)):
        num_chunks += 1
        chunked = np.array_split(data_npy, num_chunks)
    if num_chunks > 1:
        for chunk_id, chunk in enumerate(chunked):
            group.attrs['%s%d' % (name, chunk_id)] = chunk
    else:
        group.attrs[name] = data


def save_weights_to_hdf5_group(f, layers, weights):
    """keras/engine/saving.py:save_weights_to_hdf5_group; `layers` = [(name, [(weight name, array), ...])]"""
    save_attributes_to_hdf5_group(f, 'layer_names', [name.encode('utf8') for name, _ in layers])
    f.attrs['backend'] = 'tensorflow'.encode('utf8')
    f.attrs['keras_version'] = '2.2.4'.encode('utf8')
    for name, ws in layers:
        g = f.create_group(name)
        weight_names = [wn.encode('utf8') for wn, _ in ws]
        save_attributes_to_hdf5_group(g, 'weight_names', weight_names)
        for wn, val in zip(weight_names, [v for _, v in ws]):
            param_dset = g.create_dataset(wn, val.shape, dtype=val.dtype)
            if not val.shape:
                param_dset[()] = val
            else:
                param_dset[:] = val


def main():
    import fetal_net.model as fmodel
    from fetal_net import keras_h5
    kw = dict(input_shape=(1, 8, 8, 8), depth=2, n_base_filters=4, batch_normalization=True)
    model = fmodel.unet_model_3d(**kw)
    rs = np.random.RandomState(12)
    shapes = keras_h5.weight_shapes(model)
    W = {k: (rs.randn(*sh) * 0.1).astype(np.float32) for k, sh in shapes.items()}
    layers = []
    for l in model.layers:
        keys = dict((ll.name, ks) for ll, ks in keras_h5.weighted_layers(model)).get(l.name, ())
        layers.append((l.name, [("%s/%s:0" % (l.name, k), W["%s/%s" % (l.name, k)]) for k in keys]))
    trainable = keras_h5.trainable_keys(model)
    n = len(trainable)
    m = {k: (rs.randn(*shapes[k]) * 0.01).astype(np.float32) for k in trainable}
    v = {k: (rs.rand(*shapes[k]) * 0.001).astype(np.float32) for k in trainable}
    iterations = 37
    out = os.path.join(HERE, "keras_like_golden.h5")
    with h5py.File(out, "w") as f:
        # _serialize_model: the H5Dict wrapper stores bytes / str values as attributes, arrays as datasets
        f.attrs['keras_version'] = '2.2.4'.encode('utf8')
        f.attrs['backend'] = 'tensorflow'.encode('utf8')
        f.attrs['model_config'] = json.dumps(keras_h5.model_config(model)).encode('utf8')
        save_weights_to_hdf5_group(f.create_group('model_weights'), layers, W)
        f.attrs['training_config'] = json.dumps(keras_h5.training_config(model)).encode('utf8')
        og = f.create_group('optimizer_weights')
        var = lambda i: "training/Adam/Variable%s:0" % ("" if i == 0 else "_%d" % i)
        names = ["Adam/iterations:0"] + [var(i) for i in range(3 * n)]          # ms, then vs, then vhats: three list comprehensions in Adam.get_updates
        vals = [np.asarray(iterations, dtype=np.int64)] + [m[k] for k in trainable] + [v[k] for k in trainable] + [np.zeros((1,), np.float32)] * n
        og.attrs['weight_names'] = [nm.encode('utf8') for nm in names]
        for nm, val in zip(names, vals):
            d = og.create_dataset(nm.encode('utf8'), val.shape, dtype=val.dtype)          # H5Dict.__setitem__ for numpy values
            if not val.shape:
                d[()] = val
            else:
                d[:] = val
    arrays = {"w/" + k: a for k, a in W.items()}
    arrays.update({"m/" + k: a for k, a in m.items()})
    arrays.update({"v/" + k: a for k, a in v.items()})
    arrays["iterations"] = np.int64(iterations)
    arrays["builder_kwargs"] = np.frombuffer(json.dumps(kw).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "keras_like_golden.npz"), **arrays)
    print("keras_like_golden.h5:", os.path.getsize(out), "bytes;", len(layers), "layers,", n, "trainable weights; h5py", h5py.__version__)


if __name__ == "__main__":
    main()
