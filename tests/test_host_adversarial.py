"""Host side of the adversarial / semi-supervised loops: the discriminator builder's layer graph against the
fixture recorded from the reference builder, the oracle's own bookkeeping, and the batch-assembly helpers of
fetal/experiments/train_adv.py:36-124 (restated from the reference text; their numpy draws follow its order)."""
import json
import os

import numpy as np
import pytest

import fetal_net.model as fmodel
from fetal_net import adversarial as ADV
from oracle import discriminator_oracle as DO

CASES = [
    ("dis3d_train_adv", dict(input_shape=[2, 64, 64, 16], initial_learning_rate=1e-4, dropout_rate=0.3)),
    ("dis3d_full_depth", dict(input_shape=(2, 128, 128, 32), n_base_filters=8)),
    ("dis3d_shallow", dict(input_shape=(3, 32, 16, 8), depth=4, n_base_filters=16, dropout_rate=0.1)),
]


@pytest.fixture(scope="module")
def topo(golden_dir):
    with open(os.path.join(golden_dir, "topology_golden.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("case,kw", CASES)
def test_discriminator_graph_matches_reference(topo, case, kw):
    model = fmodel.discriminator_image_3d(**kw)
    gold = topo[case]
    assert [l.name for l in model.layers] == [l["name"] for l in gold["layers"]]
    for mine, ref in zip(model.layers, gold["layers"]):
        assert mine.class_name == ref["class"], mine.name
        assert list(mine.output_shape) == ref["output_shape"], mine.name
        assert mine.inbound == ref["inputs"], mine.name
    assert list(model.output_shape) == gold["output_shape"] == [None, 1]
    opt = gold["compile"]["optimizer"]
    assert model.optimizer.lr == opt["lr"] and model.optimizer.beta_1 == opt["beta_1"] == 0.5
    assert model.loss.__name__ == gold["compile"]["loss"] == "d_loss"
    assert model.metrics == gold["compile"]["metrics"] == ["mae"]
    assert model.metrics_names == ["loss", "mean_absolute_error"]
    # the first convolution is the only strided one: (2, 2, 1)
    convs = [l for l in gold["layers"] if l["class"] == "Conv3D"]
    assert convs[0]["kw"]["strides"] == [2, 2, 1] and all(c["kw"]["strides"] == 1 for c in convs[1:])
    mine = [l for l in model.layers if l.class_name == "Conv3D"]
    assert mine[0].config["strides"] == (2, 2, 1) and all(m.config["strides"] == (1, 1, 1) for m in mine[1:])
    # Dense(128, activation=LeakyReLU()) consumes a leaky_re_lu name although the instance is no node of the graph
    dense = [l for l in gold["layers"] if l["class"] == "Dense"]
    for d in dense[:-1]:
        assert d["kw"]["activation"].startswith("leaky_re_lu_")
    assert dense[-1]["kw"]["activation"] == "sigmoid"


@pytest.mark.parametrize("case,kw", CASES)
def test_oracle_spec_agrees_with_the_recorded_graph(topo, case, kw):
    gold = topo[case]
    spec = DO.DiscriminatorSpec(kw["input_shape"], kw.get("n_base_filters", 16), kw.get("depth", 5), kw.get("dropout_rate", 0.3))
    convs = [l for l in gold["layers"] if l["class"] == "Conv3D"]
    assert [c["name"] for c in convs] == [n for b in spec.blocks for n in (b["a"], b["b"])]
    assert [c["output_shape"][1] for c in convs] == [b["cout"] for b in spec.blocks for _ in (0, 1)]
    dense = [l["name"] for l in gold["layers"] if l["class"] == "Dense"]
    assert dense == spec.dense and len(dense) == spec.fc_layers + 1
    gap = [l for l in gold["layers"] if l["class"] == "GlobalAveragePooling3D"][0]
    assert gap["output_shape"] == [None, spec.gap_channels]
    assert gap["input_shapes"][0][2:] == list(spec.final_spatial)
    W = spec.init_weights(0)
    model = fmodel.discriminator_image_3d(**kw)
    assert sum(int(np.prod(v.shape)) for v in W.values()) == model.count_params()


def test_discriminator_builder_argument_errors():
    with pytest.raises(ValueError):                       # the reference's own default shape is not a Conv3D input either
        fmodel.discriminator_image_3d()
    with pytest.raises(ValueError, match="tuple of 2 integers"):      # reference all_dis_2d.py:31-32 hands Conv2D a 3-tuple of strides
        fmodel.discriminator_image_2d(input_shape=(2, 64, 64))


def test_scheduler_decays_on_plateau(capsys):
    s = ADV.Scheduler(2, 3, init_lr=1e-3, lr_decay=0.5, lr_patience=2)
    assert (s.get_dsteps(), s.get_gsteps(), s.get_lr()) == (2, 3, 1e-3)
    s.update_steps(0, 1.0)
    s.update_steps(1, 1.5)
    assert s.get_lr() == 1e-3
    s.update_steps(2, 1.2)                                 # second epoch without improvement
    assert s.get_lr() == 5e-4 and s.steps_stuck == 0
    s.update_steps(3, 0.9)
    assert s.best_loss == 0.9 and s.get_lr() == 5e-4
    assert "Reducing LR" in capsys.readouterr().out


def test_input2discriminator_layout_and_labels():
    rs = np.random.RandomState(3)
    x = rs.rand(3, 1, 4, 4, 2).astype(np.float32)
    segs = (rs.rand(3, 1, 4, 4, 2) > 0.5).astype(np.uint8)
    fake = rs.rand(3, 1, 4, 4, 2).astype(np.float32)
    np.random.seed(5)
    d_x, d_y = ADV.input2discriminator(x, segs, fake, (None, 1))
    assert d_x.shape == (6, 2, 4, 4, 2) and d_y.shape == (6, 1)
    assert np.all(d_y[:3] >= 0.9) and np.all(d_y[:3] <= 1.0) and np.all(d_y[3:] >= 0.0) and np.all(d_y[3:] <= 0.1 + 1e-12)
    # the generated half is exactly the mul-merge of (patch, fake map); the real half is the (possibly noised) truth
    np.testing.assert_array_equal(d_x[3:, :1], x * fake)
    np.testing.assert_array_equal(d_x[3:, 1:], x * (1 - fake))
    np.testing.assert_allclose(d_x[:3, :1] + d_x[:3, 1:], x, rtol=1e-6, atol=1e-7)
    # same seed -> same draws in the reference's order: choice (noise or not), [normal, normal], uniform
    np.random.seed(5)
    noisy = ADV.add_noise_to_segs(segs)
    lab = np.clip(np.random.uniform(0.9, 1.0, size=[6, 1]), 0, 1)
    np.testing.assert_array_equal(d_x[:3, :1], x * noisy)
    np.testing.assert_array_equal(d_y[:3], lab[:3])
    np.testing.assert_array_equal(d_y[3:], 1 - lab[3:])
    # concatenation form (mul_merge=False): patches first, maps second (train_adv.py:102-104)
    np.random.seed(6)
    c_x, _ = ADV.input2discriminator(x, segs, fake, (None, 1), mul_merge=False)
    np.testing.assert_array_equal(c_x[3:, :1], x)
    np.testing.assert_array_equal(c_x[3:, 1:], fake)


def test_input2gan_targets():
    rs = np.random.RandomState(0)
    x, segs, semi = rs.rand(2, 1, 4, 4, 2), rs.rand(2, 1, 4, 4, 2) > 0.5, rs.rand(2, 1, 4, 4, 2)
    np.random.seed(1)
    g_x, (valid, s) = ADV.input2gan(x, segs, (None, 1))
    assert g_x is x and s is segs and valid.shape == (2, 1) and np.all(valid >= 0.9)
    np.random.seed(1)
    (a, b), (s2, valid2) = ADV.input2gan(x, segs, (None, 1), semi_patches=semi)
    assert a is x and b is semi and s2 is segs
    np.testing.assert_array_equal(valid, valid2)


def test_add_noise_to_segs_is_clipped_and_optional():
    segs = (np.random.RandomState(2).rand(2, 1, 8, 8, 4) > 0.5).astype(np.uint8)
    seen = set()
    for seed in range(8):
        np.random.seed(seed)
        out = ADV.add_noise_to_segs(segs)
        if out is segs:
            seen.add("same")
        else:
            seen.add("noisy")
            assert out.dtype == np.float32 and out.min() >= 0.0 and out.max() <= 1.0
            assert np.abs(out - segs).max() < 0.25
    assert seen == {"same", "noisy"}


def test_build_dsc_format():
    assert ADV.build_dsc(["loss", "mae"], [0.12345, 2.0]) == "loss=0.123, mae=2.000|"


def test_oracle_discriminator_gradients_are_consistent():
    """the oracle against itself: its analytic pieces (TF 'same' strides, pooling, BCE clip) vs finite differences"""
    import torch
    spec = DO.DiscriminatorSpec((2, 16, 16, 4), n_base_filters=4, depth=3, dropout_rate=0.0)
    assert spec.fc_layers == 1 and spec.final_spatial == (2, 2, 1)
    W = spec.init_weights(1)
    rs = np.random.RandomState(0)
    x = rs.randn(2, 2, 16, 16, 4)
    t = np.array([[0.95], [0.03]])
    loss, mae, p, grads = DO.discriminator_step(spec, W, x, t)
    assert p.shape == (2, 1) and 0 < mae < 1 and np.isfinite(loss)
    name = "dense_2/kernel"
    eps = 1e-6
    W2 = dict(W)
    k = W[name].astype(np.float64).copy()
    k[3, 0] += eps
    W2[name] = k
    loss2 = DO.discriminator_step(spec, W2, x, t)[0]
    assert abs((loss2 - loss) / eps - float(grads[name][3, 0])) < 1e-4
    # the adversarial term's gradient lands on the probability channels only and has the generator's shape
    probs = rs.rand(2, 1, 16, 16, 4)
    val, g = DO.adversarial_term(spec, W, probs, x[:, :1], np.array([[1.0], [0.9]]))
    assert g.shape == (2, 1, 16, 16, 4) and np.isfinite(val) and float(g.abs().max()) > 0


def test_experiment_scripts_call_the_reference_generators_with_accepted_keywords(golden_dir):
    """fetal/experiments/_common.generator_kwargs hands get_training_and_validation_generators exactly keywords the reference function
    takes (fixture: its argument names, parsed from the reference source), and the semi-supervised stream swaps augment for val_augment
    as reference train_semi.py:215-240 does"""
    from fetal.experiments import _common as C
    with open(os.path.join(golden_dir, "signatures_golden.json")) as f:
        sig = json.load(f)["get_training_and_validation_generators"]
    keys = ["batch_size", "validation_split", "validation_file", "training_file", "test_file", "n_labels", "labels", "patch_shape", "patch_depth",
            "validation_batch_size", "augment", "skip_blank_train", "skip_blank_val", "truth_index", "truth_size", "prev_truth_index",
            "prev_truth_size", "truth_downsample", "truth_crop", "patches_per_epoch", "categorical", "3D", "drop_easy_patches_train",
            "drop_easy_patches_val"]
    cfg = {k: i for i, k in enumerate(keys)}
    cfg["patch_shape"] = [64, 64]
    kw = C.generator_kwargs(cfg, overwrite=False)
    assert set(kw) <= set(sig["args"]) - {"data_file"}, set(kw) - set(sig["args"])
    assert kw["patch_shape"] == (64, 64, cfg["patch_depth"]) and kw["is3d"] == cfg["3D"] and kw["data_split"] == cfg["validation_split"]
    semi = C.generator_kwargs(cfg, overwrite=True, val_augment=cfg["augment"])
    semi.pop("augment")
    assert set(semi) <= set(sig["args"]) and "val_augment" in semi and semi["overwrite"] is True
    d = C.config_with_defaults({"dis_model_name": "discriminator_image"})
    assert d["dis_model_name"] == "discriminator_image_3d" and d["gd_loss_ratio"] == 10 and d["dis_steps"] == d["gen_steps"] == 1
    import fetal_net.model as fmodel
    assert callable(getattr(fmodel, d["dis_model_name"]))


@pytest.mark.parametrize("case,kw", CASES)
def test_discriminator_builder_call_is_recovered_from_a_keras_model_config(case, kw):
    """keras_h5.infer_builder on the model_config / training_config of a discriminator: the same builder call comes back (what
    load_old_model uses for a checkpoint without this package's own record)"""
    from fetal_net import keras_h5
    model = fmodel.discriminator_image_3d(**kw)
    mc = json.loads(json.dumps(keras_h5.model_config(model)))
    tc = json.loads(json.dumps(keras_h5.training_config(model)))
    name, got = keras_h5.infer_builder(mc, tc)
    assert name == "discriminator_image_3d"
    again = fmodel.discriminator_image_3d(**{k: v for k, v in got.items() if k in ("input_shape", "n_base_filters", "depth", "dropout_rate",
                                                                                  "initial_learning_rate")})
    assert [(l.name, l.class_name, l.output_shape) for l in again.layers] == [(l.name, l.class_name, l.output_shape) for l in model.layers]
    assert abs(again.optimizer.lr - model.optimizer.lr) <= 1e-7 * model.optimizer.lr and tc["optimizer_config"]["config"]["beta_1"] == 0.5      # lr is stored as float32, like Keras
