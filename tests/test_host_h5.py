"""Keras-HDF5 checkpoint interchange: the ctypes libhdf5 binding and the Keras 2.2 file layout restated in
fetal_net/keras_h5.py.  Host-only: weights are injected with set_weights_dict, no engine is built."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

from fetal_net.utils import hdf5

pytestmark = pytest.mark.skipif(not hdf5.available(), reason="no libhdf5 on this host")


def _random_weights(model, seed=0):
    from fetal_net import keras_h5
    rng = np.random.RandomState(seed)
    return dict((k, rng.standard_normal(shape).astype(np.float32)) for k, shape in keras_h5.weight_shapes(model).items())


def test_hdf5_binding_round_trip(tmp_path):
    p = str(tmp_path / "t.h5")
    with hdf5.File(p, "w") as f:
        f.attrs["backend"] = b"tensorflow"
        f.attrs["unicode"] = u"2.2.4"
        f.attrs["names"] = [b"conv3d_1", b"bn_22"]
        f.attrs["x"] = np.float32(3.5)
        f.attrs["v"] = np.arange(5, dtype=np.int64)
        g = f.create_group("a")
        g.create_dataset("a/kernel:0", data=np.arange(24, dtype=np.float32).reshape(2, 3, 4))
        g.create_dataset("it", data=np.int64(7))
        g.create_dataset("d", data=np.linspace(0, 1, 7))
        with pytest.raises(OSError):
            g.create_dataset("it", data=np.int64(8))             # name in use
    assert hdf5.is_hdf5(p)
    with hdf5.File(p) as f:
        assert f.attrs["backend"] == b"tensorflow" and f.attrs["unicode"] == b"2.2.4"
        assert list(f.attrs["names"]) == [b"conv3d_1", b"bn_22"]
        assert f.attrs["x"] == np.float32(3.5) and f.attrs["x"].dtype == np.float32
        assert np.array_equal(f.attrs["v"], np.arange(5))
        assert sorted(f.attrs.keys()) == ["backend", "names", "unicode", "v", "x"]
        assert f.keys() == ["a"] and sorted(f["a"].keys()) == ["a", "d", "it"]
        k = f["a"]["a/kernel:0"][()]
        assert k.dtype == np.float32 and np.array_equal(k, np.arange(24, dtype=np.float32).reshape(2, 3, 4))
        assert f["a/it"].shape == () and int(f["a/it"][()]) == 7
        assert f["a/d"][()].dtype == np.float64
        assert "a/nope" not in f and "nope/deeper" not in f
        with pytest.raises(KeyError):
            f["nope"]
        with pytest.raises(KeyError):
            f.attrs["nope"]
    with pytest.raises(OSError):
        hdf5.File(str(tmp_path / "absent.h5"))


def test_long_name_lists_are_split_like_keras(tmp_path):
    from fetal_net import keras_h5
    names = [("layer_with_a_long_name_%05d" % i).encode() for i in range(4000)]         # > 64 KiB as one attribute
    with hdf5.File(str(tmp_path / "s.h5"), "w") as f:
        keras_h5._set_list_attr(f, "layer_names", names)
        assert "layer_names" not in f.attrs and "layer_names0" in f.attrs
        assert keras_h5._get_list_attr(f, "layer_names") == [n.decode() for n in names]


@pytest.mark.parametrize("variant", ["unet3d_bn_deconv", "unet3d_plain", "unet2d", "isensee"])
def test_keras_file_layout_and_reload(tmp_path, variant):
    import fetal_net.model as models
    from fetal_net import keras_h5
    from fetal_net.metrics import dice_coefficient_loss
    from fetal_net.training import load_old_model
    if variant == "unet3d_bn_deconv":
        model = models.unet_model_3d(input_shape=(1, 8, 16, 16), depth=2, n_base_filters=4, deconvolution=True, batch_normalization=True,
                                     initial_learning_rate=3e-4)
    elif variant == "unet3d_plain":
        model = models.unet_model_3d(input_shape=(2, 8, 16, 16), depth=3, n_base_filters=4, n_labels=2)
    elif variant == "unet2d":
        model = models.unet_model_2d(input_shape=(16, 16, 5), depth=3, n_base_filters=4)
    else:
        model = models.isensee2017_model_3d(input_shape=(1, 16, 16, 16), depth=3, n_base_filters=4, n_segmentation_levels=2,
                                            dropout_rate=0.25)
    W = _random_weights(model)
    model.set_weights_dict(W)
    path = str(tmp_path / "model.h5")
    model.save(path)
    assert hdf5.is_hdf5(path)

    with hdf5.File(path) as f:
        assert f.attrs["keras_version"] == b"2.2.4" and f.attrs["backend"] == b"tensorflow"
        mc = json.loads(bytes(f.attrs["model_config"]).decode())
        tc = json.loads(bytes(f.attrs["training_config"]).decode())
        g = f["model_weights"]
        assert [bytes(n).decode() for n in g.attrs["layer_names"]] == [l.name for l in model.layers]
        for layer, keys in keras_h5.weighted_layers(model):
            lg = g[layer.name]
            assert [bytes(n).decode() for n in lg.attrs["weight_names"]] == ["%s/%s:0" % (layer.name, k) for k in keys]
            for k in keys:
                d = lg["%s/%s:0" % (layer.name, k)][()]
                assert d.dtype == np.float32 and np.array_equal(d, W["%s/%s" % (layer.name, k)])
        first_plain = [l for l in model.layers if l.class_name in ("Activation", "Concatenate", "Add", "MaxPooling3D")][0]
        assert len(g[first_plain.name].attrs["weight_names"]) == 0            # Keras writes an empty list for weightless layers
    assert mc["class_name"] == "Model" and [l["name"] for l in mc["config"]["layers"]] == [l.name for l in model.layers]
    assert tc["optimizer_config"]["class_name"] == "Adam" and tc["loss"] == "dice_coefficient_loss"
    assert tc["metrics"][:2] == ["binary_accuracy", "vod_coefficient"]

    # the builder call is recoverable from the Keras model_config alone (a file written by Keras has no fmri_builder attribute)
    name, kw = keras_h5.infer_builder(mc, tc)
    assert name == model._builder
    for key, val in kw.items():
        if key == "loss_function":
            assert val == {"__callable__": "dice_coefficient_loss"}
            continue
        ref = model._builder_kwargs[key]
        if isinstance(ref, float):
            assert abs(val - ref) <= 1e-6 * abs(ref), key
        else:
            assert (tuple(val) if isinstance(val, (list, tuple)) else val) == (tuple(ref) if isinstance(ref, (list, tuple)) else ref), key

    again = load_old_model(path, verbose=False)
    assert [l.name for l in again.layers] == [l.name for l in model.layers]
    assert again.loss is dice_coefficient_loss
    W2 = again.get_weights_dict()
    assert set(W2) == set(W) and all(np.array_equal(W2[k], W[k]) for k in W)
    assert abs(again.optimizer.lr - model.optimizer.lr) < 1e-12


def test_load_by_order_from_a_file_with_other_layer_numbers_and_no_builder_record(tmp_path):
    """what a reference-trained file looks like: Keras auto names continue across models of one session, and there is no
    fmri_builder attribute - the topology comes from model_config, the weights are matched by order"""
    import copy
    import fetal_net.model as models
    from fetal_net import keras_h5
    from fetal_net.training import load_old_model
    model = models.unet_model_3d(input_shape=(1, 8, 16, 16), depth=2, n_base_filters=4, batch_normalization=True)
    W = _random_weights(model, seed=3)
    shifted = copy.deepcopy(model)
    rename = {}
    for l in shifted.layers:
        base, num = l.name.rsplit("_", 1)
        rename[l.name] = "%s_%d" % (base, int(num) + 14)
    for l in shifted.layers:
        l.name = rename[l.name]
        l.inbound = [rename[n] for n in l.inbound]
    shifted.set_weights_dict(dict((rename[k.split("/")[0]] + "/" + k.split("/")[1], v) for k, v in W.items()))
    path = str(tmp_path / "ref_style.h5")
    keras_h5.save_model(shifted, path, extra_meta=None)
    with hdf5.File(path) as f:
        assert "fmri_builder" not in f.attrs and "conv3d_15" in f["model_weights"]
    got = load_old_model(path, verbose=False)
    assert got.layers[1].name == "conv3d_1"
    W2 = got.get_weights_dict()
    assert all(np.array_equal(W2[k], W[k]) for k in W)
    # weights-only file + by-hand model
    wpath = str(tmp_path / "weights.h5")
    shifted.save_weights(wpath)
    with hdf5.File(wpath) as f:
        assert "model_weights" not in f and "layer_names" in f.attrs
    fresh = models.unet_model_3d(input_shape=(1, 8, 16, 16), depth=2, n_base_filters=4, batch_normalization=True)
    fresh.load_weights(wpath)
    assert all(np.array_equal(fresh.get_weights_dict()[k], W[k]) for k in W)
    with pytest.raises(ValueError):
        load_old_model(wpath, verbose=False)                              # no model_config in a weights-only file
    other = models.unet_model_3d(input_shape=(1, 8, 16, 16), depth=2, n_base_filters=8, batch_normalization=True)
    with pytest.raises(ValueError):
        other.load_weights(wpath)                                         # shape mismatch is an error, never a partial load


def test_optimizer_state_round_trip(tmp_path):
    import fetal_net.model as models
    from fetal_net import keras_h5
    model = models.unet_model_3d(input_shape=(1, 8, 16, 16), depth=2, n_base_filters=4, batch_normalization=True)
    model.set_weights_dict(_random_weights(model))
    keys = keras_h5.trainable_keys(model)
    assert not any("moving" in k for k in keys) and keys[:4] == ["conv3d_1/kernel", "conv3d_1/bias", "batch_normalization_1/gamma",
                                                                  "batch_normalization_1/beta"]
    shapes = keras_h5.weight_shapes(model)
    rng = np.random.RandomState(5)
    m = dict((k, rng.standard_normal(shapes[k]).astype(np.float32)) for k in keys)
    v = dict((k, rng.random_sample(shapes[k]).astype(np.float32)) for k in keys)
    model._pending_opt = (m, v, 123)
    path = str(tmp_path / "opt.h5")
    model.save(path)
    with hdf5.File(path) as f:
        names = [bytes(n).decode() for n in f["optimizer_weights"].attrs["weight_names"]]
        # Keras 2.2.x Adam: [iterations] + ms + vs + one shape-(1,) vhat placeholder per parameter (amsgrad=False), anonymous slot names
        n = len(keys)
        assert len(names) == 1 + 3 * n and names[0] == "Adam/iterations:0"
        assert names[1] == "training/Adam/Variable:0" and names[1 + n] == "training/Adam/Variable_%d:0" % n and names[1 + 2 * n] == "training/Adam/Variable_%d:0" % (2 * n)
        assert f["optimizer_weights"][names[0]][()].dtype == np.int64
        assert all(f["optimizer_weights"][nm][()].shape == (1,) for nm in names[1 + 2 * n:])
    m2, v2, t2 = keras_h5.read_optimizer(path, model)
    assert t2 == 123 and all(np.array_equal(m2[k], m[k]) and np.array_equal(v2[k], v[k]) for k in keys)
    model.save(str(tmp_path / "noopt.h5"), include_optimizer=False)
    assert keras_h5.read_optimizer(str(tmp_path / "noopt.h5"), model) is None


def _write_keras224_style_file(path, model, W, m, v, t, vhats=True, bad_count=False):
    """a file laid out the way keras 2.2.4 `model.save()` lays it out (keras/engine/saving.py _serialize_model + optimizers.Adam):
    written with the raw binding, not with this package's writer."""
    from fetal_net import keras_h5
    keys = keras_h5.trainable_keys(model)
    with hdf5.File(path, "w") as f:
        f.attrs["keras_version"] = b"2.2.4"
        f.attrs["backend"] = b"tensorflow"
        f.attrs["model_config"] = json.dumps(keras_h5.model_config(model)).encode()
        f.attrs["training_config"] = json.dumps(keras_h5.training_config(model)).encode()
        g = f.create_group("model_weights")
        keras_h5.write_weights_group(g, model, W)
        g.close()
        og = f.create_group("optimizer_weights")
        n = len(keys)
        names = ["Adam/iterations:0"]
        slots = [np.int64(t)]
        var = lambda i: "training/Adam/Variable%s:0" % ("" if i == 0 else "_%d" % i)
        names += [var(3 * i) for i in range(n)] + [var(3 * i + 1) for i in range(n)]
        slots += [m[k] for k in keys] + [v[k] for k in keys]
        if vhats:
            names += [var(3 * i + 2) for i in range(n)]
            slots += [np.zeros((1,), np.float32) for _ in keys]
        if bad_count:
            names, slots = names[:-3], slots[:-3]
        og.attrs["weight_names"] = np.asarray([nm.encode() for nm in names], dtype="S")
        for nm, a in zip(names, slots):
            og.create_dataset(nm, data=a).close()
        og.close()


def test_keras_224_checkpoint_layout_opens_with_load_old_model(tmp_path):
    """a ModelCheckpoint file of the reference carries 1 + 3n optimizer arrays (vhat placeholders); it must open with
    load_old_model alone and restore m, v, iterations; an unreadable optimizer group is a warning, not a failed load"""
    import fetal_net.model as models
    from fetal_net import keras_h5
    from fetal_net.training import load_old_model
    model = models.unet_model_3d(input_shape=(1, 8, 16, 16), depth=2, n_base_filters=4)
    W = _random_weights(model)
    model.set_weights_dict(W)
    keys, shapes = keras_h5.trainable_keys(model), keras_h5.weight_shapes(model)
    rng = np.random.RandomState(9)
    m = dict((k, rng.standard_normal(shapes[k]).astype(np.float32)) for k in keys)
    v = dict((k, rng.random_sample(shapes[k]).astype(np.float32)) for k in keys)
    for vhats in (True, False):
        path = str(tmp_path / ("k224_%d.h5" % vhats))
        _write_keras224_style_file(path, model, W, m, v, 77, vhats=vhats)
        m2, v2, t2 = keras_h5.read_optimizer(path, model)
        assert t2 == 77 and all(np.array_equal(m2[k], m[k]) and np.array_equal(v2[k], v[k]) for k in keys)
        got = load_old_model(path, verbose=False)
        assert all(np.array_equal(got.get_weights_dict()[k], W[k]) for k in W)
        assert got._pending_opt is not None and got._pending_opt[2] == 77
    bad = str(tmp_path / "bad.h5")
    _write_keras224_style_file(bad, model, W, m, v, 5, bad_count=True)
    with pytest.warns(UserWarning, match="optimizer state"):
        got = load_old_model(bad, verbose=False)
    assert all(np.array_equal(got.get_weights_dict()[k], W[k]) for k in W)


def test_npz_container_still_loads(tmp_path, monkeypatch):
    import fetal_net.model as models
    from fetal_net.training import load_old_model
    model = models.unet_model_3d(input_shape=(1, 8, 16, 16), depth=2, n_base_filters=4)
    W = _random_weights(model)
    model.set_weights_dict(W)
    monkeypatch.setenv("FMRI_CHECKPOINT_FORMAT", "npz")
    path = str(tmp_path / "m.h5")
    model.save(path)
    assert not hdf5.is_hdf5(path)
    got = load_old_model(path, verbose=False)
    assert all(np.array_equal(got.get_weights_dict()[k], W[k]) for k in W)


def test_h5dump_reads_the_file(tmp_path):
    tool = shutil.which("h5dump") or ("/opt/conda/bin/h5dump" if os.path.exists("/opt/conda/bin/h5dump") else None)
    if tool is None:
        pytest.skip("no h5dump")
    import fetal_net.model as models
    model = models.unet_model_3d(input_shape=(1, 8, 16, 16), depth=2, n_base_filters=4)
    model.set_weights_dict(_random_weights(model))
    path = str(tmp_path / "m.h5")
    model.save(path)
    out = subprocess.run([tool, "-H", path], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0
    assert 'GROUP "model_weights"' in out.stdout and 'DATASET "kernel:0"' in out.stdout and "H5T_IEEE_F32LE" in out.stdout


# ---------------------------------------------------------------------------------------------- pinned against h5py (Keras' own HDF5 layer)
CONDA_PY = "/opt/conda/bin/python3.9"          # the container's stray interpreter: the only one with h5py (no Keras anywhere)


def test_h5py_written_keras_style_file_loads(golden_dir):
    """tests/golden/keras_like_golden.h5 was written by h5py 3.3.0 through Keras 2.2.4's saving call sequence (restated in
    make_keras_h5_fixture.py): variable-length ASCII scalars, fixed-length string lists, nested dataset names, an int64 scalar.  The
    reader recovers the builder, every weight and the Adam state; load_old_model opens it with no other information."""
    import fetal_net.model as models
    from fetal_net import keras_h5
    from fetal_net.training import load_old_model
    path = os.path.join(golden_dir, "keras_like_golden.h5")
    z = np.load(os.path.join(golden_dir, "keras_like_golden.npz"))
    kw = json.loads(bytes(z["builder_kwargs"]).decode())
    with hdf5.File(path) as f:
        assert bytes(f.attrs["keras_version"]) == b"2.2.4" and bytes(f.attrs["backend"]) == b"tensorflow"
        assert json.loads(bytes(f.attrs["model_config"]).decode())["class_name"] == "Model"
    meta = __import__("fetal_net.engine_model", fromlist=["x"]).read_checkpoint_meta(path)
    assert meta["builder"] == "unet_model_3d" and tuple(meta["builder_kwargs"]["input_shape"]) == tuple(kw["input_shape"])
    assert meta["builder_kwargs"]["depth"] == kw["depth"] and meta["builder_kwargs"]["n_base_filters"] == kw["n_base_filters"]
    model = models.unet_model_3d(**kw)
    W = keras_h5.map_weights(model, keras_h5.read_weights(path))
    want = {k[2:]: z[k] for k in z.files if k.startswith("w/")}
    assert set(W) == set(want) and all(np.array_equal(W[k], want[k]) for k in want)
    m, v, t = keras_h5.read_optimizer(path, model)
    assert t == int(z["iterations"])
    for k in keras_h5.trainable_keys(model):
        assert np.array_equal(m[k], z["m/" + k]) and np.array_equal(v[k], z["v/" + k])
    reopened = load_old_model(path)
    got = reopened.get_weights_dict()
    assert all(np.array_equal(got[k], want[k]) for k in want)
    assert reopened.get_optimizer_state()[2] == int(z["iterations"])


@pytest.mark.skipif(not os.path.exists(CONDA_PY), reason="needs the container's conda interpreter (h5py)")
def test_keras_style_h5py_reader_reads_our_checkpoints(tmp_path):
    """the mirror image: files written by keras_h5.save_model, read by h5py through Keras 2.2.4's LOADING call sequence
    (tests/keras_h5_read_like_keras.py, run under the conda interpreter): layer names, every weight, the optimizer slots by position"""
    import fetal_net.model as models
    from fetal_net import keras_h5
    for builder, kw in (("unet_model_3d", dict(input_shape=(1, 8, 16, 16), depth=2, n_base_filters=4, batch_normalization=True)),
                        ("isensee2017_model_3d", dict(input_shape=(1, 16, 16, 16), depth=3, n_base_filters=4, n_segmentation_levels=2))):
        model = getattr(models, builder)(**kw)
        W = _random_weights(model)
        model.set_weights_dict(W)
        keys = keras_h5.trainable_keys(model)
        shapes = keras_h5.weight_shapes(model)
        rng = np.random.RandomState(3)
        m = dict((k, rng.standard_normal(shapes[k]).astype(np.float32)) for k in keys)
        v = dict((k, rng.random_sample(shapes[k]).astype(np.float32)) for k in keys)
        model._pending_opt = (m, v, 41)
        path, out = str(tmp_path / (builder + ".h5")), str(tmp_path / (builder + ".npz"))
        model.save(path)
        script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "keras_h5_read_like_keras.py")
        env = {k: v_ for k, v_ in os.environ.items() if not k.startswith("PYTHON")}
        subprocess.run([CONDA_PY, "-W", "ignore", script, path, out], check=True, env=env, cwd=str(tmp_path))
        z = np.load(out)
        assert str(z["keras_version"]) == "2.2.4" and str(z["backend"]) == "tensorflow" and str(z["optimizer_class"]) == "Adam"
        assert list(z["layer_names"]) == [l.name for l in model.layers] == list(z["config_layer_names"])
        assert list(z["layer_classes"]) == [l.class_name for l in model.layers]
        for k, a in W.items():
            if k.rsplit("/", 1)[1] in ("moving_mean", "moving_variance") and ("w/%s:0" % k) not in z.files:
                continue
            assert np.array_equal(z["w/%s:0" % k], a), k
        n = len(keys)
        names = list(z["opt_names"])
        assert len(names) == 1 + 3 * n and names[0] == "Adam/iterations:0"
        assert int(z["opt/0000"]) == 41 and z["opt/0000"].dtype == np.int64
        for i, k in enumerate(keys):
            assert np.array_equal(z["opt/%04d" % (1 + i)], m[k]) and np.array_equal(z["opt/%04d" % (1 + n + i)], v[k])
            assert z["opt/%04d" % (1 + 2 * n + i)].shape == (1,)
