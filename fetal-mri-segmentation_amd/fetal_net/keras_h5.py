"""Keras 2.2 HDF5 checkpoint interchange.

The reference's checkpoints are what Keras' ModelCheckpoint / `model.save()` write (reference fetal_net/training.py:31-32) and what
`keras.models.load_model` reads back (training.py:45-66).  Keras itself is not part of this stack; this module restates the FILE
LAYOUT of keras/engine/saving.py (Keras 2.2.x - the version the reference's API usage implies):

  /                      attrs: keras_version, backend, model_config (JSON: {"class_name": "Model", "config": {...}}),
                                training_config (JSON: optimizer_config, loss, metrics, sample_weight_mode, loss_weights)
  /model_weights         attrs: layer_names [S], backend, keras_version
     /<layer>            attrs: weight_names [S]  e.g. conv3d_1/kernel:0, conv3d_1/bias:0
        /<layer>/kernel:0   dataset (k1,k2,k3,Cin,Cout) float32      (Conv3DTranspose: (k1,k2,k3,Cout,Cin))
  /optimizer_weights     attrs: weight_names [S]; datasets: Adam iterations (int64 scalar), then m per trainable weight, then v
A weights-only file (`save_weights`) is the /model_weights group placed at the root.

Writing: the files are meant to open in Keras 2.2 (`load_model(..., custom_objects=...)`, `load_weights`) - the layer configs below
carry every key the Keras layer constructors serialise.  Pinned at the h5py level (Keras' own HDF5 layer): a file written by h5py through
Keras 2.2.4's saving call sequence is read here (tests/golden/keras_like_golden.h5, make_keras_h5_fixture.py), and files written here are
read by h5py through Keras' loading call sequence (tests/keras_h5_read_like_keras.py) - both run under the container's conda
interpreter, the only one with h5py.  Keras itself exists nowhere in this image: its call sequences and JSON configs are restated.
Reading: weights are matched to the model's weighted layers BY ORDER (what Keras' `load_weights` does when `by_name=False`), so files
whose auto-numbered names differ (conv3d_15 ...) load too; `infer_builder` recovers the builder call from `model_config`, so a
reference-trained `.h5` opens with `load_old_model(path)` alone.
"""
import json
from collections import OrderedDict

import numpy as np

from .utils import hdf5

KERAS_VERSION = "2.2.4"
BACKEND = "tensorflow"
_ATTR_LIMIT = 64512                       # keras/engine/saving.py HDF5_OBJECT_HEADER_LIMIT

_WEIGHT_KEYS = {"Conv3D": ("kernel", "bias"), "Conv2D": ("kernel", "bias"), "Conv3DTranspose": ("kernel", "bias"),
                "Conv2DTranspose": ("kernel", "bias"), "BatchNormalization": ("gamma", "beta", "moving_mean", "moving_variance"),
                "InstanceNormalization": ("gamma", "beta"), "Dense": ("kernel", "bias")}
_NON_TRAINABLE = ("moving_mean", "moving_variance")


def weighted_layers(model):
    """[(layer, (weight keys...))] in layer order - Keras' `model.layers` filtered to those that own weights"""
    return [(l, _WEIGHT_KEYS[l.class_name]) for l in model.layers if l.class_name in _WEIGHT_KEYS]


def trainable_keys(model):
    """'<layer>/<key>' of every trainable weight in Keras' `model.trainable_weights` order (= the order of the Adam moments)"""
    return ["%s/%s" % (l.name, k) for l, keys in weighted_layers(model) for k in keys if k not in _NON_TRAINABLE]


def weight_shapes(model):
    """{'<layer>/<key>': shape} in Keras layout, from the recorded layer graph"""
    by_name = dict((l.name, l) for l in model.layers)
    out = OrderedDict()
    for l, keys in weighted_layers(model):
        c_in = int(by_name[l.inbound[0]].output_shape[1])
        c = int(l.output_shape[1])
        for k in keys:
            if k != "kernel":
                out["%s/%s" % (l.name, k)] = (c,)
            elif l.class_name == "Dense":
                out["%s/kernel" % l.name] = (c_in, c)
            elif l.class_name.endswith("Transpose"):
                out["%s/kernel" % l.name] = tuple(int(v) for v in l.config["kernel_size"]) + (c, c_in)
            else:
                out["%s/kernel" % l.name] = tuple(int(v) for v in l.config["kernel_size"]) + (c_in, c)
    return out


# ------------------------------------------------------------------------------------------------------------ layer configs
def _init(class_name, **config):
    return {"class_name": class_name, "config": config}


_GLOROT = _init("VarianceScaling", scale=1.0, mode="fan_avg", distribution="uniform", seed=None)


def _data_format(model):
    return "channels_first"           # both builders compute channels-first (the 2-D one between two Permute layers)


def layer_config(layer, model):
    c, cls, name = layer.config, layer.class_name, layer.name
    base = {"name": name, "trainable": True}
    if cls == "InputLayer":
        return {"batch_input_shape": [None] + [int(v) for v in layer.output_shape[1:]], "dtype": "float32", "sparse": False, "name": name}
    if cls in ("Conv3D", "Conv2D", "Conv3DTranspose", "Conv2DTranspose"):
        nd = len(c["kernel_size"])
        out = dict(base, filters=int(c["filters"]), kernel_size=[int(k) for k in c["kernel_size"]],
                   strides=[int(s) for s in c.get("strides", (1,) * nd)], padding=c.get("padding", "valid"),
                   data_format=_data_format(model), dilation_rate=[1] * nd, activation="linear", use_bias=True,
                   kernel_initializer=_GLOROT, bias_initializer=_init("Zeros"), kernel_regularizer=None, bias_regularizer=None,
                   activity_regularizer=None, kernel_constraint=None, bias_constraint=None)
        if cls.endswith("Transpose"):
            out["output_padding"] = None
            del out["dilation_rate"]
        return out
    if cls == "BatchNormalization":
        return dict(base, axis=int(c.get("axis", 1)), momentum=0.99, epsilon=0.001, center=True, scale=True,
                    beta_initializer=_init("Zeros"), gamma_initializer=_init("Ones"), moving_mean_initializer=_init("Zeros"),
                    moving_variance_initializer=_init("Ones"), beta_regularizer=None, gamma_regularizer=None, beta_constraint=None,
                    gamma_constraint=None)
    if cls == "InstanceNormalization":
        return dict(base, axis=int(c.get("axis", 1)), epsilon=0.001, center=True, scale=True, beta_initializer=_init("Zeros"),
                    gamma_initializer=_init("Ones"), beta_regularizer=None, gamma_regularizer=None, beta_constraint=None,
                    gamma_constraint=None)
    if cls == "Activation":
        return dict(base, activation=c["activation"])
    if cls == "LeakyReLU":
        return dict(base, alpha=float(np.float32(c.get("alpha", 0.3))))
    if cls in ("MaxPooling3D", "MaxPooling2D"):
        p = [int(v) for v in c["pool_size"]]
        return dict(base, pool_size=p, padding="valid", strides=p, data_format=_data_format(model))
    if cls in ("UpSampling3D", "UpSampling2D"):
        out = dict(base, size=[int(v) for v in c["size"]], data_format=_data_format(model))
        if cls == "UpSampling2D":
            out["interpolation"] = "nearest"
        return out
    if cls == "Concatenate":
        return dict(base, axis=int(c.get("axis", 1)))
    if cls == "Add":
        return dict(base)
    if cls == "Permute":
        return dict(base, dims=[int(v) for v in c["dims"]])
    if cls.startswith("SpatialDropout"):
        return dict(base, rate=float(c["rate"]), noise_shape=None, seed=None)
    if cls in ("AveragePooling3D", "AveragePooling2D"):
        p = [int(v) for v in c["pool_size"]]
        return dict(base, pool_size=p, padding="valid", strides=p, data_format=_data_format(model))
    if cls in ("GlobalAveragePooling3D", "GlobalAveragePooling2D"):
        return dict(base, data_format=_data_format(model))
    if cls == "Dense":
        # Dense(128, activation=LeakyReLU()) (reference all_dis_3d.py:42): Keras serialises a layer instance used as an activation by its
        # class name, which its own deserialiser cannot resolve without custom_objects={'LeakyReLU': ...}
        act = {None: "linear", "leaky_relu": "LeakyReLU"}.get(c.get("activation"), c.get("activation"))
        return dict(base, units=int(c["units"]), activation=act, use_bias=True, kernel_initializer=_GLOROT, bias_initializer=_init("Zeros"),
                    kernel_regularizer=None, bias_regularizer=None, activity_regularizer=None, kernel_constraint=None, bias_constraint=None)
    raise TypeError("no Keras config for layer class %s" % cls)


def model_config(model):
    layers = []
    for l in model.layers:
        inbound = [[[n, 0, 0, {}] for n in l.inbound]] if l.inbound else []
        layers.append({"name": l.name, "class_name": l.class_name, "config": layer_config(l, model), "inbound_nodes": inbound})
    return {"class_name": "Model",
            "config": {"name": model.name, "layers": layers, "input_layers": [[model.layers[0].name, 0, 0]],
                       "output_layers": [[model.layers[-1].name, 0, 0]]}}


def _fn_name(f):
    return f if isinstance(f, str) else getattr(f, "__name__", str(f))


def training_config(model):
    opt = model.optimizer.get_config() if model.optimizer is not None else {}
    return {"optimizer_config": {"class_name": "Adam",
                                 "config": {"lr": float(np.float32(opt.get("lr", 0.001))), "beta_1": float(np.float32(opt.get("beta_1", 0.9))),
                                            "beta_2": float(np.float32(opt.get("beta_2", 0.999))), "decay": 0.0,
                                            "epsilon": float(opt.get("epsilon", 1e-7)), "amsgrad": False}},
            "loss": _fn_name(model.loss), "metrics": [_fn_name(m) for m in model.metrics], "sample_weight_mode": None, "loss_weights": None}


# ------------------------------------------------------------------------------------------------------------ writing
def _set_list_attr(group, name, values):
    """keras save_attributes_to_hdf5_group: one attribute, or name0, name1 ... pieces when it would exceed the header limit"""
    arr = np.asarray(values, dtype="S") if len(values) else np.zeros((0,), "S1")
    if arr.nbytes <= _ATTR_LIMIT:
        group.attrs[name] = arr
        return
    pieces = 2
    while any(c.nbytes > _ATTR_LIMIT for c in np.array_split(arr, pieces)):
        pieces += 1
    for i, chunk in enumerate(np.array_split(arr, pieces)):
        group.attrs["%s%d" % (name, i)] = chunk


def _get_list_attr(group, name):
    if name in group.attrs:
        return [bytes(v).decode("utf8") for v in np.atleast_1d(group.attrs[name])]
    out, i = [], 0
    while "%s%d" % (name, i) in group.attrs:
        out += [bytes(v).decode("utf8") for v in np.atleast_1d(group.attrs["%s%d" % (name, i)])]
        i += 1
    if i == 0:
        raise KeyError("attribute %r (or %r0...) not found in %s" % (name, name, group.name))
    return out


def write_weights_group(group, model, W):
    group.attrs["backend"] = BACKEND.encode()
    group.attrs["keras_version"] = KERAS_VERSION.encode()
    _set_list_attr(group, "layer_names", [l.name.encode() for l in model.layers])
    owners = dict((l.name, keys) for l, keys in weighted_layers(model))
    for l in model.layers:
        g = group.create_group(l.name)
        names = []
        for key in owners.get(l.name, ()):
            full = "%s/%s" % (l.name, key)
            if full not in W:
                if key in _NON_TRAINABLE:
                    continue
                raise KeyError("weight %s missing from the engine's export" % full)
            names.append((full, "%s/%s:0" % (l.name, key)))
        _set_list_attr(g, "weight_names", [n.encode() for _, n in names])
        for full, n in names:
            g.create_dataset(n, data=np.asarray(W[full], np.float32)).close()
        g.close()


def save_model(model, path, include_optimizer=True, weights_only=False, extra_meta=None):
    W = model.get_weights_dict()
    with hdf5.File(path, "w") as f:
        if weights_only:
            write_weights_group(f, model, W)
            return
        f.attrs["keras_version"] = KERAS_VERSION.encode()
        f.attrs["backend"] = BACKEND.encode()
        f.attrs["model_config"] = json.dumps(model_config(model)).encode("utf8")
        if extra_meta is not None:
            f.attrs["fmri_builder"] = json.dumps(extra_meta).encode("utf8")      # not a Keras key: exact builder call for this stack
        g = f.create_group("model_weights")
        write_weights_group(g, model, W)
        g.close()
        if model.optimizer is not None and model.loss is not None:
            f.attrs["training_config"] = json.dumps(training_config(model)).encode("utf8")
        state = model.get_optimizer_state() if include_optimizer else None
        if state is not None:
            m, v, t = state
            og = f.create_group("optimizer_weights")
            keys = trainable_keys(model)
            # Keras 2.2.x Adam.weights = [iterations] + ms + vs + vhats (keras/optimizers.py Adam.get_updates): the slots are anonymous
            # K.zeros variables "training/Adam/Variable[_k]:0".  get_updates builds ms, vs and vhats as three separate list comprehensions,
            # so the creation order - and with it the numbering - is ALL m (Variable .. Variable_{n-1}), then all v (_n .. _{2n-1}), then
            # all vhat (_2n .. _{3n-1}); with amsgrad=False every vhat is a K.zeros(1) placeholder that is still saved.  Keras restores
            # them BY POSITION (optimizer.set_weights), so the count - 1 + 3n - is what has to be right for a file written here to
            # resume there; the names only have to be unique.  (No Keras exists in this image: the golden file this layout is tested
            # against, tests/golden/keras_like_golden.h5, is generated by this repository's own restatement of Keras' saving sequence.)
            n = len(keys)
            var = lambda i: "training/Adam/Variable%s:0" % ("" if i == 0 else "_%d" % i)
            names = ["Adam/iterations:0"] + [var(i) for i in range(3 * n)]
            _set_list_attr(og, "weight_names", [n_.encode() for n_ in names])
            og.create_dataset(names[0], data=np.int64(t)).close()
            for i, k in enumerate(keys):
                og.create_dataset(names[1 + i], data=np.asarray(m[k], np.float32)).close()
                og.create_dataset(names[1 + n + i], data=np.asarray(v[k], np.float32)).close()
                og.create_dataset(names[1 + 2 * n + i], data=np.zeros((1,), np.float32)).close()
            og.close()


# ------------------------------------------------------------------------------------------------------------ reading
def _weights_root(f):
    if "model_weights" in f:
        return f["model_weights"]
    if "layer_names" in f.attrs or "layer_names0" in f.attrs:
        return f
    raise ValueError("%s holds neither /model_weights nor a root layer_names attribute: not a Keras weight file" % f.filename)


def read_weights(path):
    """-> [(layer name in the file, [(weight key, ndarray), ...])] for the layers that own weights, in file (= model) order"""
    out = []
    with hdf5.File(path) as f:
        root = _weights_root(f)
        for lname in _get_list_attr(root, "layer_names"):
            g = root[lname]
            wnames = _get_list_attr(g, "weight_names") if ("weight_names" in g.attrs or "weight_names0" in g.attrs) else []
            if wnames:
                out.append((lname, [(n.rsplit("/", 1)[-1].split(":")[0], np.asarray(g[n][()])) for n in wnames]))
            g.close()
    return out


def map_weights(model, file_layers):
    """order-based assignment of the file's weighted layers to the model's (Keras load_weights, by_name=False)"""
    mine = weighted_layers(model)
    if len(mine) != len(file_layers):
        raise ValueError("the file holds %d layers with weights, the model has %d" % (len(file_layers), len(mine)))
    W = OrderedDict()
    shapes = weight_shapes(model)
    for (layer, keys), (fname, arrays) in zip(mine, file_layers):
        have = OrderedDict(arrays)
        for k in keys:
            if k not in have:
                raise ValueError("layer %s of the file has no %r (model layer %s)" % (fname, k, layer.name))
            full = "%s/%s" % (layer.name, k)
            if tuple(have[k].shape) != shapes[full]:
                raise ValueError("%s of the file has shape %s, model layer %s expects %s" % (fname + "/" + k, tuple(have[k].shape),
                                                                                          layer.name, shapes[full]))
            W[full] = have[k]
    return W


def read_optimizer(path, model):
    """-> (m dict, v dict, iterations) keyed like trainable_keys(model), or None when the file has no optimizer state"""
    with hdf5.File(path) as f:
        if "optimizer_weights" not in f:
            return None
        og = f["optimizer_weights"]
        names = _get_list_attr(og, "weight_names")
        arrays = [np.asarray(og[n][()]) for n in names]
        og.close()
    keys = trainable_keys(model)
    n = len(keys)
    short = [nm.rsplit("/", 1)[-1].split(":")[0] for nm in names]
    if all(("m_" + k) in short and ("v_" + k) in short for k in keys):
        # files of this package's first format: slots named after their parameter
        at = dict(zip(short, arrays))
        it = [a for nm, a in zip(short, arrays) if nm == "iterations"]
        return (OrderedDict((k, at["m_" + k]) for k in keys), OrderedDict((k, at["v_" + k]) for k in keys), int(it[0]) if it else 0)
    # Keras layout, by position: [iterations] + ms + vs (+ one shape-(1,) vhat placeholder per parameter when amsgrad=False, or
    # parameter-shaped vhats when amsgrad=True - ignored either way: this engine's Adam has no amsgrad)
    if len(arrays) not in (1 + 2 * n, 1 + 3 * n):
        raise ValueError("optimizer_weights holds %d arrays, expected 1 + 2 x %d or 1 + 3 x %d (Keras Adam)" % (len(arrays), n, n))
    t = int(np.asarray(arrays[0]).reshape(-1)[0])
    m = OrderedDict(zip(keys, arrays[1:1 + n]))
    v = OrderedDict(zip(keys, arrays[1 + n:1 + 2 * n]))
    shapes = weight_shapes(model)
    for k in keys:
        for slot, a in (("m", m[k]), ("v", v[k])):
            if tuple(a.shape) != tuple(shapes[k]):
                raise ValueError("optimizer slot %s of %s has shape %s, the parameter has %s" % (slot, k, tuple(a.shape), tuple(shapes[k])))
    return m, v, t


def read_configs(path):
    """-> (model_config dict | None, training_config dict | None, fmri_builder dict | None)"""
    with hdf5.File(path) as f:
        def js(name):
            if name not in f.attrs:
                return None
            v = f.attrs[name]
            return json.loads(bytes(v).decode("utf8") if not isinstance(v, str) else v)
        return js("model_config"), js("training_config"), js("fmri_builder")


def infer_builder(mc, tc=None):
    """(builder name, kwargs) from a Keras `model_config` (+ `training_config` for lr and loss name) of a model one of this package's
    builders can produce: unet_model_3d, unet_model_2d, isensee2017_model_3d, isensee2017_model, discriminator_image_3d."""
    layers = mc["config"]["layers"]
    cls = [l["class_name"] for l in layers]
    inputs = [l for l in layers if l["class_name"] == "InputLayer"]
    input_shape = tuple(int(v) for v in inputs[0]["config"]["batch_input_shape"][1:])
    convs = [l for l in layers if l["class_name"] in ("Conv3D", "Conv2D")]
    if not convs:
        raise ValueError("model_config has no convolution layers")
    kw = dict(input_shape=input_shape)
    finals = [l for l in convs if all(int(k) == 1 for k in l["config"]["kernel_size"])]
    act = [l for l in layers if l["class_name"] == "Activation"]
    if act:
        kw["activation_name"] = act[-1]["config"]["activation"]
    if tc is not None:
        kw["initial_learning_rate"] = float(tc["optimizer_config"]["config"]["lr"])
        kw["loss_function"] = {"__callable__": tc["loss"]} if isinstance(tc["loss"], str) else None
    if "GlobalAveragePooling3D" in cls and "Dense" in cls:
        # PatchGAN discriminator (reference model/discriminator/all_dis_3d.py): one SpatialDropout3D per conv block built, every level
        # the early stop skipped became a Dense(128) in front of the Dense(1) output
        drops = [l for l in layers if l["class_name"] == "SpatialDropout3D"]
        dense = [l for l in layers if l["class_name"] == "Dense"]
        kw.pop("activation_name", None)
        kw.pop("loss_function", None)
        kw.update(n_base_filters=int(convs[0]["config"]["filters"]), depth=len(drops) + len(dense) - 1,
                  dropout_rate=float(drops[0]["config"]["rate"]) if drops else 0.0)
        return "discriminator_image_3d", kw
    if "Add" in cls or "LeakyReLU" in cls:
        drops = [l for l in layers if l["class_name"].startswith("SpatialDropout")]
        heads = [l for l in convs if all(int(k) == 1 for k in l["config"]["kernel_size"]) and
                 not _followed_by_norm(l["name"], layers)]
        kw.update(n_base_filters=int(convs[0]["config"]["filters"]), depth=len(drops),
                  dropout_rate=float(drops[0]["config"]["rate"]) if drops else 0.0, n_segmentation_levels=len(heads),
                  n_labels=int(heads[0]["config"]["filters"]))
        if "Conv2D" in cls:
            # 2-D twin: only the heads that reach the output are in the file; more than one means they were summed (summation=True).
            # With summation=False the file cannot tell how many (dead) heads the builder created - their count only shows in the layer
            # numbering: the head conv2d_K of level 0 is the (3*depth + 3*(depth-1) + created_heads)-th convolution
            last = int(heads[-1]["name"].rsplit("_", 1)[1])
            depth = len(drops)
            kw.update(summation=len(heads) > 1, n_segmentation_levels=(len(heads) if len(heads) > 1 else max(1, last - (6 * depth - 3))))
            return "isensee2017_model", kw
        if len(inputs) > 1:
            kw["mask_shape"] = tuple(int(v) for v in inputs[1]["config"]["batch_input_shape"][1:])
        return "isensee2017_model_3d", kw
    nd = 3 if "Conv3D" in cls else 2
    pools = [l for l in layers if l["class_name"] == "MaxPooling%dD" % nd]
    kw.update(depth=len(pools) + 1, n_base_filters=int(convs[0]["config"]["filters"]), n_labels=int(finals[-1]["config"]["filters"]),
              deconvolution=("Conv%dDTranspose" % nd) in cls, batch_normalization="BatchNormalization" in cls)
    if pools:
        kw["pool_size"] = tuple(int(v) for v in pools[0]["config"]["pool_size"])
    if nd == 2:
        drops = [l for l in layers if l["class_name"] == "SpatialDropout2D"]
        kw["dropout_rate"] = float(drops[0]["config"]["rate"]) if drops else 0
        return "unet_model_2d", kw
    if "InstanceNormalization" in cls:
        raise NotImplementedError("unet_model_3d has no instance_normalization argument in the reference builder; rebuild by hand")
    return "unet_model_3d", kw


def _followed_by_norm(name, layers):
    for l in layers:
        if l["class_name"] in ("InstanceNormalization", "BatchNormalization"):
            for node in l["inbound_nodes"]:
                if any(src[0] == name for src in node):
                    return True
    return False
