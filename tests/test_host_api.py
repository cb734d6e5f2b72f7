"""Host-side mirror of the reference API (no GPU): builders, naming, callbacks, metrics tokens, tiler geometry, checkpoints."""
import json
import os

import numpy as np
import pytest

import fetal_net.metrics as FM
import fetal_net.model as fmodel
from fetal_net import prediction as P
from fetal_net import training as T
from fetal_net.engine_model import (CSVLogger, EarlyStopping, LearningRateScheduler, ModelCheckpoint, ReduceLROnPlateau)


@pytest.fixture(scope="module")
def topo(golden_dir):
    with open(os.path.join(golden_dir, "topology_golden.json")) as f:
        return json.load(f)


def test_reference_test_model_naming():
    # reference test/test_model.py:7-15
    model = fmodel.unet_model_3d(input_shape=(1, 16, 16, 16), depth=2, deconvolution=True, metrics=[], n_labels=1,
                                 batch_normalization=True)
    names = [l.name for l in model.layers]
    for name in names[:-3]:
        if 'conv3d' in name and 'transpose' not in name:
            assert name.replace('conv3d', 'batch_normalization') in names


@pytest.mark.parametrize("case,fn,kw", [
    ("unet3d_cfg1", "unet_model_3d", dict(input_shape=(1, 16, 64, 64), depth=3, n_base_filters=8)),
    ("unet3d_default", "unet_model_3d", dict(input_shape=(1, 64, 128, 128))),
    ("unet3d_test_model", "unet_model_3d", dict(input_shape=(1, 16, 16, 16), depth=2, deconvolution=True, metrics=[], n_labels=1,
                                               batch_normalization=True)),
    ("unet2d_cfg4", "unet_model_2d", dict(input_shape=(256, 256, 5))),
    ("unet2d_dropout", "unet_model_2d", dict(input_shape=(64, 64, 5), depth=3, n_base_filters=16, dropout_rate=0.2)),
    ("isensee3d_d3", "isensee2017_model_3d", dict(input_shape=(1, 32, 32, 32), depth=3, n_base_filters=8, n_segmentation_levels=2)),
    ("isensee3d_default", "isensee2017_model_3d", dict()),
])
def test_builder_graph_matches_reference(topo, case, fn, kw):
    model = getattr(fmodel, fn)(**kw)
    gold = topo[case]
    assert [l.name for l in model.layers] == [l["name"] for l in gold["layers"]]
    for mine, ref in zip(model.layers, gold["layers"]):
        assert list(mine.output_shape) == ref["output_shape"], mine.name
        assert mine.inbound == ref["inputs"], mine.name
    assert list(model.output_shape) == gold["output_shape"]
    assert model.optimizer.lr == gold["compile"]["optimizer"]["lr"]
    assert model.metrics_names[1:] == gold["compile"]["metrics"]
    assert getattr(model.loss, "__name__") == gold["compile"]["loss"]


@pytest.mark.parametrize("case,kw", [
    ("isensee2d_d3", dict(input_shape=(32, 32, 3), depth=3, n_base_filters=8, n_segmentation_levels=2)),
    ("isensee2d_slices5", dict(input_shape=(128, 128, 5))),
    ("isensee2d_summation", dict(input_shape=(64, 64, 5), depth=4, n_base_filters=8, summation=True, loss_function=FM.dice_and_xent)),
])
def test_isensee2d_builder_matches_reference(topo, case, kw):
    """2-D Isensee (reference model/unet/isensee.py:14-105).  The fixture records every layer the reference builder CREATES; with
    summation=False the deeper segmentation heads are created but reach no output, so a Keras Model drops them: `model.layers` must be
    the recorded graph restricted to what the output depends on, under the recorded names (the dead heads still advance conv2d_N)."""
    model = fmodel.isensee2017_model(**kw)
    gold = topo[case]
    assert [l.name for l in model._created_layers] == [l["name"] for l in gold["layers"]]
    by = {l["name"]: l for l in gold["layers"]}
    keep, stack = set(), [gold["layers"][-1]["name"]]
    while stack:
        n = stack.pop()
        if n not in keep:
            keep.add(n)
            stack.extend(by[n]["inputs"])
    live = [l for l in gold["layers"] if l["name"] in keep]
    assert (len(live) < len(gold["layers"])) == (not kw.get("summation", False))
    assert [l.name for l in model.layers] == [l["name"] for l in live]
    for mine, ref in zip(model.layers, live):
        assert list(mine.output_shape) == ref["output_shape"], mine.name
        assert mine.inbound == ref["inputs"], mine.name
        assert mine.class_name == ref["class"], mine.name
    assert list(model.output_shape) == gold["output_shape"]
    assert model.optimizer.lr == gold["compile"]["optimizer"]["lr"]
    assert model.metrics_names[1:] == gold["compile"]["metrics"]
    assert getattr(model.loss, "__name__") == gold["compile"]["loss"]
    # the oracle twin owns exactly the model's weights
    from fetal_net import keras_h5
    from oracle import isensee_oracle as I
    okw = {k: v for k, v in kw.items() if k != "loss_function"}
    okw.setdefault("n_segmentation_levels", 3)
    W = I.IsenseeSpec(ndim=2, **okw).init_weights(1)
    shapes = keras_h5.weight_shapes(model)
    assert list(shapes) and set(shapes) == set(W) and all(tuple(W[k].shape) == tuple(shapes[k]) for k in W)
    assert model.count_params() == sum(int(v.size) for v in W.values())


def test_isensee2d_checkpoint_header_round_trip(tmp_path):
    """model_config of a saved 2-D Isensee file is enough to rebuild the builder call (load_old_model without a config)"""
    from fetal_net import keras_h5
    for kw in (dict(input_shape=(32, 32, 3), depth=3, n_base_filters=8, n_segmentation_levels=2),
               dict(input_shape=(32, 32, 3), depth=3, n_base_filters=8, n_segmentation_levels=2, summation=True)):
        model = fmodel.isensee2017_model(**kw)
        name, got = keras_h5.infer_builder(keras_h5.model_config(model), keras_h5.training_config(model))
        assert name == "isensee2017_model"
        for k in ("input_shape", "depth", "n_base_filters", "n_segmentation_levels"):
            assert tuple(np.atleast_1d(got[k])) == tuple(np.atleast_1d(kw[k])), k
        assert got["summation"] == kw.get("summation", False)
        again = fmodel.isensee2017_model(**{k: v for k, v in got.items() if k != "loss_function"})
        assert [l.name for l in again.layers] == [l.name for l in model.layers]


def test_train_fetal_call_signature(topo):
    # exactly how reference fetal/train_fetal.py:33-39 calls a builder (unknown kwargs are swallowed)
    m = fmodel.unet_model_3d(input_shape=[1, 32, 32, 16], initial_learning_rate=1e-4,
                             **{'dropout_rate': 0, 'loss_function': FM.dice_and_xent, 'mask_shape': None, 'old_model_path': None})
    assert m.metrics_names == ['loss', 'binary_accuracy', 'vod_coefficient', 'dice_coefficient']
    assert m.count_params() == 16315585
    assert getattr(fmodel, 'unet_model_3d') is fmodel.unet_model_3d


def test_param_counts():
    assert fmodel.unet_model_3d(input_shape=(1, 16, 64, 64), depth=3, n_base_filters=8).count_params() == 245873
    assert fmodel.unet_model_2d(input_shape=(256, 256, 5)).count_params() == 5441281


def test_callbacks_list():
    # reference test/test_training.py:8-15
    _, _, scheduler = T.get_callbacks(model_file='model.h5', learning_rate_patience=50, learning_rate_drop=0.5)
    assert isinstance(scheduler, ReduceLROnPlateau)
    _, _, _, stopper = T.get_callbacks(model_file='model.h5', early_stopping_patience=100)
    assert isinstance(stopper, EarlyStopping)
    ck, csvl, sched = T.get_callbacks(model_file='model.h5', learning_rate_epochs=10)
    assert isinstance(ck, ModelCheckpoint) and isinstance(csvl, CSVLogger) and isinstance(sched, LearningRateScheduler)
    assert T.step_decay(9, 1e-3, 0.5, 10) == pytest.approx(5e-4)


class _DummyModel:
    def __init__(self):
        from fetal_net.engine_model import Adam
        self.optimizer = Adam(lr=1.0)
        self.stop_training = False
        self.saved = []

    def save(self, path):
        self.saved.append(path)


def test_callback_semantics(tmp_path):
    m = _DummyModel()
    ck = ModelCheckpoint(str(tmp_path / 'm') + '-epoch{epoch:02d}-loss{val_loss:.3f}-acc{val_binary_accuracy:.3f}.h5',
                         save_best_only=True, monitor='val_loss')
    rl = ReduceLROnPlateau(factor=0.5, patience=2)
    es = EarlyStopping(patience=3)
    cl = CSVLogger(str(tmp_path / 'training'), append=True)
    for cb in (ck, rl, es, cl):
        cb.set_model(m)
        cb.on_train_begin({})
    vals = [-0.5, -0.6, -0.6, -0.6, -0.6, -0.6]
    for ep, v in enumerate(vals):
        logs = {'loss': v, 'val_loss': v, 'val_binary_accuracy': 0.9, 'binary_accuracy': 0.9}
        for cb in (ck, cl, rl, es):
            cb.on_epoch_end(ep, logs)
        if m.stop_training:
            break
    assert [os.path.basename(p) for p in m.saved] == ['m-epoch01-loss-0.500-acc0.900.h5', 'm-epoch02-loss-0.600-acc0.900.h5']
    assert m.optimizer.lr == 0.5            # one reduction after two epochs without a 1e-4 improvement (epochs 2,3)
    assert m.stop_training and es.stopped_epoch == 4   # three epochs (2,3,4) without improvement
    cl.on_train_end({})
    rows = open(str(tmp_path / 'training')).read().strip().split('\n')
    # CSVLogger sits BEFORE the lr callback in get_callbacks' order, so (as in Keras) its key set is fixed without 'lr'
    assert rows[0] == 'epoch,binary_accuracy,loss,val_binary_accuracy,val_loss' and len(rows) == 6


def test_metrics_tokens_match_reference(golden_dir):
    with open(os.path.join(golden_dir, "metrics_golden.json")) as f:
        gold = json.load(f)
    for c in gold["cases"]:
        rs = np.random.RandomState(c["seed"])
        y = (rs.rand(*c["shape"]) > c["thr"]).astype(np.float32)
        p = rs.rand(*c["shape"]).astype(np.float32)
        assert FM.dice_coefficient(y, p) == pytest.approx(c["dice"], rel=1e-12)
        assert FM.dice_coefficient_loss(y, p) == pytest.approx(c["dice_loss"], rel=1e-12)
        assert FM.vod_coefficient(y, p) == pytest.approx(c["vod"], rel=1e-6)
        assert FM.weighted_dice_coefficient(y, p) == pytest.approx(c["weighted_dice"], rel=1e-12)
        assert FM.dice_and_xent(y, p) == pytest.approx(c["dice_and_xent"], rel=1e-10)
        assert FM.focal_loss(y, p) == pytest.approx(c["focal_loss"], rel=1e-10)
        assert FM.double_dice_loss(y, p) == pytest.approx(c["double_dice_loss"], rel=1e-10)
    assert FM.dice_coef is FM.dice_coefficient and FM.dice_coef_loss is FM.dice_coefficient_loss


def test_patch_wise_prediction_host_geometry(golden_dir):
    """product tiler (foreign-model path) vs outputs of the reference's own patch_wise_prediction"""
    from test_oracle_tiler import FakeModel2D, FakeModel3D
    with open(os.path.join(golden_dir, "tiler_golden.json")) as f:
        meta = json.load(f)
    arrs = np.load(os.path.join(golden_dir, "tiler_golden.npz"))
    for c in meta["volume_cases"]:
        data = np.random.RandomState(c["seed"]).randn(1, *c["vol"])
        fm = FakeModel3D(c["patch"], c["n_out"]) if c["kind"] == "3d" else FakeModel2D(c["patch"][:2], c["patch"][2])
        out = P.patch_wise_prediction(model=fm, data=data, patch_shape=c["patch"], overlap_factor=c["overlap_factor"],
                                      batch_size=c["batch_size"])
        assert fm.calls == c["predict_calls"]
        np.testing.assert_allclose(out, arrs["out_" + c["name"]], rtol=0, atol=1e-13, err_msg=c["name"])
    for c in meta["index_cases"]:
        ov = np.subtract(c["patch"], c["pred"]) + (c["overlap_factor"] * (np.subtract(c["patch"], 1) - np.subtract(c["patch"], c["pred"]))).astype(int)
        idx = P.get_set_of_patch_indices_full((0, 0, 0), np.subtract(c["vol"], c["patch"]), np.subtract(c["patch"], ov))
        assert np.array_equal(idx, arrs["idx_" + c["name"]])


def test_checkpoint_roundtrip_without_gpu(tmp_path):
    from oracle.unet_oracle import Spec
    m = fmodel.unet_model_3d(input_shape=(1, 8, 16, 16), depth=2, n_base_filters=4, initial_learning_rate=3e-4)
    W = Spec((1, 8, 16, 16), depth=2, n_base_filters=4).init_weights(5)
    m.set_weights_dict(W)
    path = str(tmp_path / "fetal_net_model-epoch01-loss-0.500-acc0.900.h5")
    m.save(path)
    m2 = T.load_old_model(path)
    W2 = m2.get_weights_dict()
    assert m2.optimizer.lr == pytest.approx(3e-4) and m2.count_params() == m.count_params()
    for k in W:
        assert np.array_equal(W[k], W2[k])


def test_nifti_roundtrip(tmp_path):
    from fetal_net.utils.nifti import load_nifti, save_nifti
    a = np.random.RandomState(0).rand(5, 6, 7).astype(np.float32)
    p = save_nifti(a, str(tmp_path / "prediction.nii.gz"))
    assert np.array_equal(load_nifti(p), a)


def test_unsupported_topologies_fail_loudly():
    m = fmodel.unet_model_3d(input_shape=(1, 16, 16, 16), depth=2, activation_name="softmax")
    with pytest.raises(NotImplementedError):
        m.predict(np.zeros((1, 1, 16, 16, 16)))
    m2 = fmodel.isensee2017_model(input_shape=(32, 32, 5), activation_name="softmax")      # built, but not runnable on the engine
    with pytest.raises(NotImplementedError):
        m2.predict(np.zeros((1, 32, 32, 5)))


def test_nifti_reader_big_endian_scaled_with_affine(tmp_path):
    """a hand-built big-endian int16 NIfTI-1 with scl_slope / scl_inter and an sform: values, scaling and affine come back"""
    import struct
    from fetal_net.utils.nifti import load_nifti, save_nifti
    vol = (np.arange(2 * 3 * 4, dtype=np.int16).reshape(2, 3, 4) - 7)
    hdr = bytearray(348)
    struct.pack_into(">i", hdr, 0, 348)
    struct.pack_into(">8h", hdr, 40, 3, 2, 3, 4, 1, 1, 1, 1)
    struct.pack_into(">h", hdr, 70, 4)
    struct.pack_into(">h", hdr, 72, 16)
    struct.pack_into(">8f", hdr, 76, 1.0, 0.5, 0.5, 2.0, 1.0, 1.0, 1.0, 1.0)
    struct.pack_into(">f", hdr, 108, 352.0)
    struct.pack_into(">2f", hdr, 112, 0.25, 10.0)
    struct.pack_into(">h", hdr, 254, 1)
    A = np.array([[0.5, 0, 0, -3.0], [0, 0.5, 0, 4.0], [0, 0, 2.0, 1.5], [0, 0, 0, 1]])
    for r, o in enumerate((280, 296, 312)):
        struct.pack_into(">4f", hdr, o, *A[r])
    hdr[344:348] = b"n+1\x00"
    path = tmp_path / "be.nii"
    path.write_bytes(bytes(hdr) + b"\x00" * 4 + np.asfortranarray(vol.astype(">i2")).tobytes(order="F"))
    data, aff = load_nifti(str(path), return_affine=True)
    np.testing.assert_allclose(data, vol * 0.25 + 10.0)
    np.testing.assert_allclose(aff, A)
    assert np.array_equal(load_nifti(str(path), scaled=False), vol)
    # files written by save_nifti carry slope 1: the stored dtype comes back untouched, with the identity affine
    p2 = save_nifti(vol, str(tmp_path / "le.nii.gz"))
    d2, a2 = load_nifti(p2, return_affine=True)
    assert d2.dtype == np.int16 and np.array_equal(d2, vol) and np.array_equal(a2, np.eye(4))
    bad = tmp_path / "bad.nii"
    bad.write_bytes(b"\x00" * 400)
    with pytest.raises(ValueError):
        load_nifti(str(bad))
