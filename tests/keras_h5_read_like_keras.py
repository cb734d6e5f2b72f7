#!/opt/conda/bin/python3.9
"""Read a checkpoint the way Keras 2.2.4 does - h5py plus the call sequence of keras/engine/saving.py (`_deserialize_model`,
`load_weights_from_hdf5_group`, `load_attributes_from_hdf5_group`, optimizer weights by `weight_names`) restated below - and dump what it
found as an .npz.  tests/test_host_h5.py runs this under the container's conda interpreter (the only one with h5py) on a file written by
fetal_net/keras_h5.py and compares the arrays: an independent HDF5 reader, following Keras' access pattern, sees exactly what was saved.

    python3.9 tests/keras_h5_read_like_keras.py file.h5 out.npz
"""
import json
import sys

import h5py
import numpy as np


def _dec(x):
    return x.decode('utf8') if isinstance(x, bytes) else str(x)      # h5py 3 hands variable-length strings back as str, h5py 2 as bytes


def load_attributes_from_hdf5_group(group, name):
    if name in group.attrs:
        data = [_dec(n) for n in group.attrs[name]]
    else:
        data, chunk_id = [], 0
        while ('%s%d' % (name, chunk_id)) in group.attrs:
            data.extend([_dec(n) for n in group.attrs['%s%d' % (name, chunk_id)]])
            chunk_id += 1
    return data


def main(path, out):
    res = {}
    with h5py.File(path, 'r') as f:
        model_config = f.attrs.get('model_config')
        if model_config is None:
            raise ValueError('No model found in config.')
        model_config = json.loads(_dec(model_config))
        res["layer_classes"] = np.array([l["class_name"] for l in model_config["config"]["layers"]])
        res["config_layer_names"] = np.array([l["name"] for l in model_config["config"]["layers"]])
        g = f['model_weights']
        res["keras_version"] = np.array(_dec(g.attrs['keras_version']))
        res["backend"] = np.array(_dec(g.attrs['backend']))
        layer_names = load_attributes_from_hdf5_group(g, 'layer_names')
        res["layer_names"] = np.array(layer_names)
        for name in layer_names:
            lg = g[name]
            for wn in load_attributes_from_hdf5_group(lg, 'weight_names'):
                res["w/" + wn] = np.asarray(lg[wn])
        training_config = f.attrs.get('training_config')
        if training_config is not None:
            tc = json.loads(_dec(training_config))
            res["optimizer_class"] = np.array(tc['optimizer_config']['class_name'])
            res["lr"] = np.array(tc['optimizer_config']['config']['lr'])
            res["loss"] = np.array(str(tc['loss']))
            if 'optimizer_weights' in f:
                og = f['optimizer_weights']
                names = [_dec(n) for n in og.attrs['weight_names']]
                res["opt_names"] = np.array(names)
                for i, n in enumerate(names):
                    res["opt/%04d" % i] = np.asarray(og[n])
    np.savez(out, **res)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
