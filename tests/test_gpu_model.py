"""Drop-in surface on the GPU: builder -> train_model (fit_generator + callbacks + checkpoints) -> load_old_model ->
patch_wise_prediction (device overlap-add, hipGraph) -> run_validation_case, checked against the oracle."""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _gen(shape, seed, n_batches=None):
    from oracle.unet_oracle import synthetic_batch
    k = 0
    while True:
        x, y = synthetic_batch(shape, seed_x=seed + k % 4, seed_y=seed + 100 + k % 4)
        yield x.astype(np.float64), y
        k += 1


def test_train_model_predict_and_resume(tmp_path, monkeypatch):
    monkeypatch.setenv("FMRI_DTYPE", "fp32")
    import fetal_net.metrics as FM
    import fetal_net.model as fmodel
    from fetal_net.training import load_old_model, train_model
    from oracle import unet_oracle as O
    shape = (2, 1, 8, 16, 16)
    model = fmodel.unet_model_3d(input_shape=shape[1:], depth=2, n_base_filters=8, initial_learning_rate=1e-2,
                                 loss_function=FM.dice_coefficient_loss)
    # identical weights in the oracle
    spec = O.Spec(shape[1:], depth=2, n_base_filters=8)
    W = spec.init_weights(42)
    model.set_weights_dict(W)
    x0, y0 = next(_gen(shape, 1))
    p_gpu = model.predict(x0)
    _, p_ref = O.forward(spec, O.to_torch(W, torch.float64), torch.tensor(x0))
    assert p_gpu.shape == (2, 1, 8, 16, 16)
    np.testing.assert_allclose(p_gpu, p_ref.numpy(), atol=2e-5)
    # Keras-semantics training loop with the reference's callbacks
    model_file = str(tmp_path / "fetal_net_model")
    hist = train_model(model, model_file, _gen(shape, 1), _gen(shape, 50), steps_per_epoch=6, validation_steps=2,
                       initial_learning_rate=1e-2, learning_rate_drop=0.5, learning_rate_patience=2, early_stopping_patience=5,
                       n_epochs=4, output_folder=str(tmp_path))
    h = hist.history
    assert set(h) >= {"loss", "binary_accuracy", "vod_coefficient", "val_loss", "val_binary_accuracy", "val_vod_coefficient", "lr"}
    assert h["loss"][-1] < h["loss"][0]
    # first epoch's mean training loss == oracle trajectory (same batches, Keras Adam)
    opt = O.KerasAdam(W, lr=1e-2)
    g = _gen(shape, 1)
    ref_losses = [O.train_step(spec, W, opt, *next(g))["loss"] for _ in range(6)]
    assert h["loss"][0] == pytest.approx(np.mean(ref_losses), abs=2e-4)
    ckpts = sorted(glob.glob(model_file + "*.h5"), key=os.path.getmtime)
    assert ckpts and "-epoch" in ckpts[-1] and "-loss" in ckpts[-1] and "-acc" in ckpts[-1]
    assert os.path.exists(str(tmp_path / "training"))
    # resume: newest checkpoint re-opens without any config
    m2 = load_old_model(ckpts[-1])
    assert m2.count_params() == model.count_params()
    W_saved = m2.get_weights_dict()
    p2 = m2.predict(x0)
    assert p2.shape == p_gpu.shape and np.isfinite(p2).all()
    for k, v in W_saved.items():
        assert v.shape == W[k].shape


@pytest.mark.parametrize("graph", ["1", "0"])
def test_patch_wise_prediction_device_vs_oracle(monkeypatch, graph):
    monkeypatch.setenv("FMRI_DTYPE", "fp32")
    monkeypatch.setenv("FMRI_HIPGRAPH", graph)
    import fetal_net.model as fmodel
    from fetal_net.prediction import patch_wise_prediction
    from oracle import tiler_oracle, unet_oracle as O
    patch = (8, 16, 16)
    model = fmodel.unet_model_3d(input_shape=(1,) + patch, depth=2, n_base_filters=8)
    spec = O.Spec((1,) + patch, depth=2, n_base_filters=8)
    W = spec.init_weights(3)
    model.set_weights_dict(W)

    class OracleModel:
        output_shape = (None, 1) + patch

        def predict(self, x):
            return O.forward(spec, O.to_torch(W, torch.float64), torch.tensor(np.asarray(x, np.float64)))[1].numpy()

    data = np.random.RandomState(9).randn(1, 20, 40, 27)
    for factor, bs in ((0.5, 5), (0.0, 4)):
        out = patch_wise_prediction(model=model, data=data, patch_shape=patch, overlap_factor=factor, batch_size=bs)
        ref = tiler_oracle.patch_wise_prediction(OracleModel(), data, patch, factor, bs)
        assert out.shape == ref.shape == (20, 40, 27, 1) and out.dtype == np.float64
        np.testing.assert_allclose(out, ref, atol=3e-5)


def test_run_validation_case_writes_reference_files(tmp_path, monkeypatch):
    monkeypatch.setenv("FMRI_DTYPE", "bf16")
    import fetal_net.model as fmodel
    from fetal_net.prediction import run_validation_case
    from fetal_net.utils.nifti import load_nifti
    from oracle import metrics_oracle as M

    class Root:
        pass

    class DataFile:
        root = Root()

    rs = np.random.RandomState(4)
    DataFile.root.data = [rs.randn(24, 48, 32)]
    DataFile.root.truth = [(rs.rand(24, 48, 32) > 0.7).astype(np.uint8)]
    patch = (16, 32, 32)
    model = fmodel.unet_model_3d(input_shape=(1,) + patch, depth=3, n_base_filters=32)
    fn = run_validation_case(0, str(tmp_path / "case0"), model, DataFile, ["volume"], patch_shape=patch, overlap_factor=0.5)
    assert os.path.basename(fn) == "prediction.nii.gz"
    for f in ("data_volume.nii.gz", "truth.nii.gz", "prediction.nii.gz"):
        assert os.path.exists(str(tmp_path / "case0" / f))
    pred = load_nifti(fn)
    assert pred.shape == (24, 48, 32) and np.isfinite(pred).all() and 0 <= pred.min() and pred.max() <= 1
    d = M.hard_dice(DataFile.root.truth[0], pred > 0.5)      # reference fetal/evaluate.py:13-17 (value itself is arbitrary here)
    assert 0.0 <= d <= 1.0


def test_unet_model_2d_train_and_slice_wise_prediction(monkeypatch):
    """BASELINE config-4 style 2-D model through the same surface: fit_generator + patch_wise_prediction (prediction_shape (X,Y,1))."""
    monkeypatch.setenv("FMRI_DTYPE", "fp32")
    import fetal_net.model as fmodel
    from fetal_net.prediction import patch_wise_prediction
    from oracle import tiler_oracle, unet_oracle as O
    X, Y, C = 16, 32, 5
    model = fmodel.unet_model_2d(input_shape=(X, Y, C), depth=2, n_base_filters=8, initial_learning_rate=1e-2)
    spec = O.Spec((X, Y, C), ndim=2, depth=2, n_base_filters=8)
    W = spec.init_weights(5)
    model.set_weights_dict(W)
    rs = np.random.RandomState(1)

    def gen():
        while True:
            x = rs.randn(4, X, Y, C)
            yield x, (x[..., 2:3] > 0.3).astype(np.uint8)

    x0 = rs.randn(3, X, Y, C)
    p = model.predict(x0)
    _, pr = O.forward(spec, O.to_torch(W, torch.float64), torch.tensor(x0))
    assert p.shape == (3, X, Y, 1)
    np.testing.assert_allclose(p, pr.numpy(), atol=2e-5)

    class OracleModel:
        output_shape = (None, X, Y, 1)

        def predict(self, x):
            return O.forward(spec, O.to_torch(W, torch.float64), torch.tensor(np.asarray(x, np.float64)))[1].numpy()

    data = rs.randn(1, 24, 40, 7)
    out = patch_wise_prediction(model=model, data=data, patch_shape=(X, Y, C), overlap_factor=0.3, batch_size=6)
    ref = tiler_oracle.patch_wise_prediction(OracleModel(), data, (X, Y, C), 0.3, 6)
    assert out.shape == ref.shape == (24, 40, 7, 1)
    np.testing.assert_allclose(out, ref, atol=3e-5)
    h = model.fit_generator(gen(), steps_per_epoch=8, epochs=3, verbose=0).history
    assert h["loss"][-1] < h["loss"][0]


def test_reference_test_model_config_trains(monkeypatch):
    """the configuration of reference test/test_model.py:8-9 (depth 2, Deconvolution3D, BatchNormalization) is not only named
    correctly but runs: training reduces the loss, predict uses the moving statistics, checkpoint round-trips them."""
    monkeypatch.setenv("FMRI_DTYPE", "fp32")
    import fetal_net.model as fmodel
    model = fmodel.unet_model_3d(input_shape=(1, 16, 16, 16), depth=2, deconvolution=True, metrics=[], n_labels=1,
                                 batch_normalization=True, initial_learning_rate=1e-2)
    h = model.fit_generator(_gen((2, 1, 16, 16, 16), 3), steps_per_epoch=10, epochs=3, verbose=0).history
    assert h["loss"][-1] < h["loss"][0]
    x0, _ = next(_gen((2, 1, 16, 16, 16), 9))
    p = model.predict(x0)
    assert p.shape == (2, 1, 16, 16, 16) and np.isfinite(p).all()
    W = model.get_weights_dict()
    assert "batch_normalization_1/moving_mean" in W and "conv3d_transpose_1/kernel" in W
    assert W["conv3d_transpose_1/kernel"].shape == (2, 2, 2, 128, 128)
    assert float(np.abs(W["batch_normalization_1/moving_mean"]).max()) > 0      # updated by training


def test_isensee_model_surface(tmp_path, monkeypatch):
    """isensee2017_model_3d through the same Keras-style surface: fit_generator, predict, save / load_old_model"""
    monkeypatch.setenv("FMRI_DTYPE", "fp32")
    import fetal_net.model as fmodel
    from fetal_net.training import load_old_model
    shape = (2, 1, 16, 16, 16)
    model = fmodel.isensee2017_model_3d(input_shape=shape[1:], depth=3, n_base_filters=4, n_segmentation_levels=2,
                                        initial_learning_rate=5e-3)
    h = model.fit_generator(_gen(shape, 3), steps_per_epoch=10, epochs=3, validation_data=_gen(shape, 40), validation_steps=2,
                            verbose=0).history
    assert h["loss"][-1] < h["loss"][0] and "val_loss" in h
    x0, _ = next(_gen(shape, 9))
    p = model.predict(x0)
    assert p.shape == (2, 1, 16, 16, 16) and np.isfinite(p).all() and 0 <= p.min() and p.max() <= 1
    path = str(tmp_path / "isensee-epoch01-loss-0.100-acc0.900.h5")
    model.save(path)
    m2 = load_old_model(path)
    assert [l.name for l in m2.layers] == [l.name for l in model.layers]
    np.testing.assert_allclose(m2.predict(x0), p, atol=1e-6)          # inference is deterministic (dropout off)


def test_isensee_mask_weighted_loss(tmp_path):
    """isensee2017_model_3d(mask_shape=..., loss_function=dice_and_xent_mask) takes [x, masks] batches (reference isensee2017.py:85-88,
    generator.py:397-401); its loss is Dice + mean(exp(-mask/3) * BCE) and differs from the unweighted dice_and_xent on the same
    weights; save / load_old_model keep the second input."""
    import fetal_net.model as fmodel
    from fetal_net import metrics as M
    from fetal_net.training import load_old_model
    shape = (1, 16, 16, 16)
    kw = dict(input_shape=shape, depth=3, n_base_filters=4, n_segmentation_levels=2, dropout_rate=0.0, compute_dtype="fp32")
    m1 = fmodel.isensee2017_model_3d(loss_function=M.dice_and_xent_mask, mask_shape=shape, **kw)
    m2 = fmodel.isensee2017_model_3d(loss_function=M.dice_and_xent, **kw)
    m2.set_weights_dict(m1.get_weights_dict())
    rs = np.random.RandomState(0)
    x = rs.randn(2, *shape)
    y = (rs.rand(2, *shape) > 0.6).astype(np.uint8)
    masks = rs.rand(2, *shape) * 12
    with pytest.raises(ValueError):
        m1.test_on_batch(x, y)
    l1 = m1.test_on_batch([x, masks], y)[0]
    l0 = m1.test_on_batch([x, np.zeros_like(masks)], y)[0]          # zero distance = weight 1 = plain dice_and_xent
    l2 = m2.test_on_batch(x, y)[0]
    assert abs(l0 - l2) <= 1e-5 and l1 < l2 - 1e-3
    # reference value from the host-side metric on the model's own probabilities
    p = m1.predict(x)
    want = M.dice_and_xent(y.astype(np.float64), p.astype(np.float64), weight_mask=np.exp(-masks / 3.0))
    assert abs(l1 - want) <= 1e-4
    losses = [m1.train_on_batch([x, masks], y)[0] for _ in range(15)]
    assert min(losses[-4:]) < losses[0]
    path = str(tmp_path / "m-epoch01-loss0.100-acc0.900.h5")
    m1.save(path)
    m3 = load_old_model(path)
    assert getattr(m3.loss, "mask_weighted", False) and abs(m3.test_on_batch([x, masks], y)[0] - m1.test_on_batch([x, masks], y)[0]) <= 1e-5


def test_fit_generator_staged_prefetch_equals_the_inline_path(monkeypatch):
    """fit_generator with the producer thread's pinned staging ring + deferred metric reads (round 4) against FMRI_STAGE_PREFETCH=0 (round 3:
    conversion, pageable upload and a synchronous read-back on the training thread): same batches, same weights -> the same epoch logs
    (fp32 engine; atomics order only), for a plain U-Net with varying batch sizes and validation, and for the mask-weighted Isensee model
    whose generator yields ([x, masks], y) (reference generator.py:397-401).  Batch logs read inside a callback equal train_on_batch's."""
    import fetal_net.model as fmodel
    from fetal_net import metrics as M
    from fetal_net.engine_model import Callback

    def unet():
        return fmodel.unet_model_3d(input_shape=(1, 8, 16, 16), depth=2, n_base_filters=8, initial_learning_rate=1e-2, compute_dtype="fp32")

    def gen(seed, sizes, with_masks=False, shape=(1, 8, 16, 16)):
        rs = np.random.RandomState(seed)
        k = 0
        while True:
            n = sizes[k % len(sizes)]
            k += 1
            x = rs.randn(n, *shape)
            y = (rs.rand(n, *shape) > 0.6).astype(np.uint8)
            yield ([x, rs.rand(n, *shape) * 12], y) if with_masks else (x, y)

    class BatchLosses(Callback):
        def __init__(self):
            self.v = []

        def on_batch_end(self, batch, logs=None):
            self.v.append((logs["size"], float(logs["loss"])))

    hists, batch_logs = [], []
    for staged in ("1", "0"):
        monkeypatch.setenv("FMRI_STAGE_PREFETCH", staged)
        m = unet()
        if hists:
            m.set_weights_dict(W0)
        else:
            W0 = m.get_weights_dict()
        cb = BatchLosses()
        h = m.fit_generator(gen(1, [2, 3, 1]), steps_per_epoch=7, epochs=2, validation_data=gen(2, [2]), validation_steps=3, verbose=0, callbacks=[cb]).history
        hists.append(h)
        batch_logs.append(cb.v)
    assert [s for s, _ in batch_logs[0]] == [s for s, _ in batch_logs[1]] == ([2, 3, 1] * 5)[:14]        # the generator runs on across epochs, as in Keras
    np.testing.assert_allclose([v for _, v in batch_logs[0]], [v for _, v in batch_logs[1]], atol=2e-5)
    for k in ("loss", "binary_accuracy", "vod_coefficient", "val_loss", "val_binary_accuracy"):
        np.testing.assert_allclose(hists[0][k], hists[1][k], atol=3e-5, err_msg=k)
    # mask-weighted loss: the masks are staged by the producer thread too
    shape = (1, 16, 16, 16)
    kw = dict(input_shape=shape, depth=3, n_base_filters=4, n_segmentation_levels=2, dropout_rate=0.0, compute_dtype="fp32")
    outs = []
    for staged in ("1", "0"):
        monkeypatch.setenv("FMRI_STAGE_PREFETCH", staged)
        m = fmodel.isensee2017_model_3d(loss_function=M.dice_and_xent_mask, mask_shape=shape, **kw)
        if outs:
            m.set_weights_dict(W1)
        else:
            W1 = m.get_weights_dict()
        outs.append(m.fit_generator(gen(3, [2], with_masks=True, shape=shape), steps_per_epoch=5, epochs=2, verbose=0).history["loss"])
    np.testing.assert_allclose(outs[0], outs[1], atol=3e-5)
    assert outs[0][1] < outs[0][0]


def test_2d_patch_wise_prediction_device_path_equals_host_tiling():
    """2-D models: the device overlap-add loop (tile = slice stack = channels-last input, one output slice per tile) against the
    reference-style host tiling around the same model (forced through a duck-typed proxy)"""
    import fetal_net.model as fmodel
    from fetal_net.prediction import patch_wise_prediction
    # fp32 engine: the device loop regroups the tiles into larger batches, which in bf16 would also switch some layers between the
    # VALU and MFMA kernels (1e-3 rounding differences); in fp32 both groupings run the same arithmetic
    model = fmodel.unet_model_2d(input_shape=(32, 32, 5), depth=3, n_base_filters=8, compute_dtype="fp32")

    class Proxy:                                   # not a fetal_net Model: patch_wise_prediction tiles on the host and calls .predict
        output_shape = model.output_shape

        @staticmethod
        def predict(x):
            return model.predict(x)

    vol = np.random.RandomState(3).randn(1, 48, 40, 12)
    for bs in (5, 7):
        dev = patch_wise_prediction(model, vol, (32, 32, 5), overlap_factor=0.5, batch_size=bs)
        host = patch_wise_prediction(Proxy(), vol, (32, 32, 5), overlap_factor=0.5, batch_size=bs)
        assert dev.shape == host.shape == (48, 40, 12, 1)
        np.testing.assert_allclose(dev, host, rtol=0, atol=2e-6)


@pytest.mark.parametrize("variant", ["unet3d_bn_deconv", "unet2d_deconv", "isensee"])
def test_hdf5_checkpoint_resumes_training_exactly(tmp_path, monkeypatch, variant):
    """save() after 3 steps -> load_old_model() -> both models take 2 more identical steps: same losses, same weights, i.e. the
    Keras-layout file carries weights, BatchNorm moving statistics, Adam moments and the step counter losslessly."""
    from fetal_net.utils import hdf5
    if not hdf5.available():
        pytest.skip("no libhdf5 on this host")
    monkeypatch.setenv("FMRI_DTYPE", "fp32")
    import fetal_net.model as fmodel
    from fetal_net.training import load_old_model
    rng = np.random.RandomState(11)
    if variant == "unet3d_bn_deconv":
        model = fmodel.unet_model_3d(input_shape=(1, 8, 16, 16), depth=2, n_base_filters=8, deconvolution=True, batch_normalization=True,
                                     initial_learning_rate=1e-2)
        xs = [rng.standard_normal((2, 1, 8, 16, 16)) for _ in range(5)]
        ys = [(rng.random_sample((2, 1, 8, 16, 16)) > 0.6).astype(np.uint8) for _ in range(5)]
    elif variant == "unet2d_deconv":
        model = fmodel.unet_model_2d(input_shape=(16, 16, 3), depth=2, n_base_filters=8, deconvolution=True, initial_learning_rate=1e-2)
        xs = [rng.standard_normal((4, 16, 16, 3)) for _ in range(5)]
        ys = [(rng.random_sample((4, 16, 16, 1)) > 0.6).astype(np.uint8) for _ in range(5)]
    else:
        model = fmodel.isensee2017_model_3d(input_shape=(1, 16, 16, 16), depth=3, n_base_filters=4, dropout_rate=0.0,
                                            initial_learning_rate=1e-2)
        xs = [rng.standard_normal((2, 1, 16, 16, 16)) for _ in range(5)]
        ys = [(rng.random_sample((2, 1, 16, 16, 16)) > 0.6).astype(np.uint8) for _ in range(5)]
    for i in range(3):
        model.train_on_batch(xs[i], ys[i])
    path = str(tmp_path / "ckpt.h5")
    model.save(path)
    assert hdf5.is_hdf5(path)
    twin = load_old_model(path, verbose=False)
    Wa, Wb = model.get_weights_dict(), twin.get_weights_dict()
    assert set(Wa) == set(Wb) and all(np.array_equal(Wa[k], Wb[k]) for k in Wa)
    for i in range(3, 5):
        la = model.train_on_batch(xs[i], ys[i])
        lb = twin.train_on_batch(xs[i], ys[i])
        assert la[0] == pytest.approx(lb[0], abs=1e-4)
    assert twin._engine.t == model._engine.t == 5
    Wa, Wb = model.get_weights_dict(), twin.get_weights_dict()
    normed = variant != "unet2d_deconv"
    for k in Wa:
        if normed and k.endswith("/bias") and not k.startswith("conv3d_transpose"):
            continue        # a bias in front of a normalisation has a zero gradient: Adam turns its rounding noise into +-lr steps
        # Lossless state = the two trajectories coincide.  The runs themselves are not bit-reproducible (fp32 atomics in the weight-gradient
        # flush and the normalisation sums), and Adam turns the rounding noise of a near-zero gradient element into a step of up to lr in
        # either direction: a handful of such elements may differ (2 of 3,456 by 1.2e-4 in one recorded run), a lost optimizer state or
        # step counter would move EVERY element by ~lr = 1e-2.
        d = np.abs(np.asarray(Wb[k], np.float64) - np.asarray(Wa[k], np.float64))
        assert float(np.mean(d > 1e-4)) <= 5e-3 and float(d.max()) <= 4e-2 and float(np.median(d)) <= 1e-5, (k, float(d.max()), float(np.mean(d > 1e-4)))


def test_two_stage_pipeline_on_device_models(monkeypatch):
    """fetal_net.pipeline.predict_volume with two engine-backed models (reference prod/predict_nifti2.py:98-160): both stages run through
    the device tile loop; the second stage equals a direct patch-wise prediction of the normalised ROI pasted into zeros."""
    monkeypatch.setenv("FMRI_DTYPE", "fp32")
    import fetal_net.model as fmodel
    from fetal_net.pipeline import predict_volume
    from fetal_net.prediction import patch_wise_prediction
    rs = np.random.RandomState(4)
    vol = rs.randn(40, 40, 20) * 50 + 100
    m1 = fmodel.unet_model_3d(input_shape=(1, 16, 16, 8), depth=2, n_base_filters=8)
    m2 = fmodel.unet_model_3d(input_shape=(1, 32, 32, 16), depth=2, n_base_filters=8)
    W = m1.get_weights_dict()
    final = [k for k in W if k.endswith("/bias")][-1]
    W[final] = W[final] + 3.0                                  # stage 1 says "foreground" everywhere: the ROI is the whole volume
    m1.set_weights_dict(W)
    cfg1 = {"patch_shape": [16, 16], "patch_depth": 8}
    cfg2 = {"patch_shape": [32, 32], "patch_depth": 16}
    n1, n2 = {"mean": 100.0, "std": 50.0}, {"mean": 90.0, "std": 40.0}
    out = predict_volume(vol, m1, cfg1, overlap_factor=0.5, norm_params=n1, model2=m2, config2=cfg2, norm_params2=n2)
    assert out["prediction"].squeeze().shape == vol.shape and out["mask"].all()
    direct = patch_wise_prediction(model=m2, data=((vol - 90.0) / 40.0)[None], overlap_factor=0.5, patch_shape=[32, 32, 16]).squeeze()
    assert out["prediction_roi"].shape == vol.shape
    np.testing.assert_allclose(out["prediction_roi"], direct, atol=1e-6)
    assert 0.0 < float(out["prediction_roi"].min()) and float(out["prediction_roi"].max()) < 1.0
