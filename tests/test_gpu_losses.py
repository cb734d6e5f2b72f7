"""weighted_dice_coefficient_loss on the device (reference fetal_net/metrics.py:39-55): the three known-answer tests the reference's own
test suite holds for it (reference test/test_metrics.py:10-38), value and gradient against the metrics oracle / fp64 autograd, and a model
compiled with it trained through the C ABI."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _device_weighted_dice(y, p, nsamples, n_labels):
    """y, p: numpy arrays indexed [(n * vox + v) * L + l]; -> (coefficient, gsums, sums) through fmri_weighted_dice_fwd"""
    from fmri_hip import ops
    yd = torch.from_numpy(np.ascontiguousarray(y, dtype=np.uint8)).cuda().reshape(-1)
    pd = torch.from_numpy(np.ascontiguousarray(p, dtype=np.float32)).cuda().reshape(-1)
    gs = torch.empty(3 * nsamples * n_labels, dtype=torch.float64, device="cuda")
    sums = torch.zeros(16, dtype=torch.float64, device="cuda")
    ops.weighted_dice_fwd(pd, yd, gs, sums, nsamples, n_labels)
    torch.cuda.synchronize()
    s = sums.cpu().numpy()
    return -ops.loss_value_from_sums(s, ops.LOSS_WEIGHTED_DICE), gs, sums


def _kat_data():
    data = np.zeros((5 ** 3) * 3).reshape(3, 5, 5, 5)
    data[0, 0:1] = 1
    data[1, 0:2] = 1
    data[2, 1:4] = 1
    return data


def test_reference_kat_removing_one_label_costs_a_third():
    """reference test/test_metrics.py:11-23: with one of the three volumes predicted blank the coefficient is 2/3 of the perfect one"""
    data = _kat_data()
    max_dice, _, _ = _device_weighted_dice(data, data, 3, 1)
    assert abs(max_dice - 1.0) < 1e-6
    for index in range(3):
        pred = data.copy()
        pred[index] = 0
        dice, _, _ = _device_weighted_dice(data, pred, 3, 1)
        assert abs(dice - 2 * max_dice / 3) < 1e-5, (index, dice)


def test_reference_kat_blank_prediction_scores_zero():
    """reference test/test_metrics.py:25-32"""
    data = _kat_data()
    dice, _, _ = _device_weighted_dice(data, np.zeros_like(data), 3, 1)
    assert abs(dice) < 1e-5


def test_reference_kat_empty_label_still_scores_one():
    """reference test/test_metrics.py:34-39: a volume without foreground predicted as such counts as a perfect Dice (smooth / smooth)"""
    data = np.zeros((5 ** 3) * 3).reshape(3, 5, 5, 5)
    data[1, 0:2] = 1
    data[2, 1:4] = 1
    dice, _, _ = _device_weighted_dice(data, data, 3, 1)
    assert dice == 1.0


@pytest.mark.parametrize("n_labels", [1, 3])
def test_value_and_gradient_vs_oracle_and_autograd(n_labels):
    """value against oracle/metrics_oracle.weighted_dice_coefficient on the (N, labels, X, Y, Z) tensors the 3-D models produce; gradient
    w.r.t. the logits against fp64 autograd of the reference formula"""
    from fmri_hip import ops
    from oracle import metrics_oracle as MO
    N, sp = 3, (4, 6, 5)
    rs = np.random.RandomState(5)
    logits = rs.randn(N, *sp, n_labels).astype(np.float32)                 # device layout: channels last
    y = (rs.rand(N, *sp, n_labels) > 0.7).astype(np.uint8)
    y[1] = 0                                                               # a sample without foreground
    ld = torch.from_numpy(logits).cuda().reshape(-1, n_labels)
    yd = torch.from_numpy(y).cuda().reshape(-1)
    probs, sums = torch.empty_like(ld), torch.zeros(16, dtype=torch.float64, device="cuda")
    ops.sigmoid_dice_fwd(ld, yd, probs, sums)
    gs = torch.empty(3 * N * n_labels, dtype=torch.float64, device="cuda")
    ops.weighted_dice_fwd(probs, yd, gs, sums, N, n_labels)
    dl = torch.empty_like(ld)
    ops.weighted_dice_bwd(probs, yd, gs, sums, dl, N, n_labels, grad_scale=1.0)
    torch.cuda.synchronize()
    # reference layout (N, labels, X, Y, Z)
    lt = torch.tensor(logits, dtype=torch.float64).permute(0, 4, 1, 2, 3).contiguous().requires_grad_(True)
    yt = torch.tensor(y, dtype=torch.float64).permute(0, 4, 1, 2, 3)
    pt = torch.sigmoid(lt)
    s = 1e-5
    coef = (2.0 * ((yt * pt).sum((-3, -2, -1)) + s / 2) / (yt.sum((-3, -2, -1)) + pt.sum((-3, -2, -1)) + s)).mean()
    (-coef).backward()
    want = MO.weighted_dice_coefficient(yt.numpy(), pt.detach().numpy())
    assert abs(float(coef) - float(want)) < 1e-12
    got = -ops.loss_value_from_sums(sums.cpu().numpy(), ops.LOSS_WEIGHTED_DICE)
    assert abs(got - float(coef)) < 2e-6, (got, float(coef))
    g_ref = lt.grad.permute(0, 2, 3, 4, 1).reshape(-1, n_labels).numpy()
    g = dl.cpu().numpy().astype(np.float64)
    assert np.abs(g - g_ref).max() <= 2e-6 * np.abs(g_ref).max() + 1e-12, np.abs(g - g_ref).max() / np.abs(g_ref).max()


def test_model_compiled_with_weighted_dice_trains_on_the_device(monkeypatch):
    """builder(loss_function=weighted_dice_coefficient_loss) -> train_on_batch: the reported loss is the reference formula of the model's own
    prediction (batch of 3: per-sample Dice, not the whole-batch Dice of dice_coefficient_loss), and the loss goes down over a few steps"""
    monkeypatch.setenv("FMRI_DTYPE", "fp32")
    import fetal_net.metrics as FM
    import fetal_net.model as fmodel
    from oracle import metrics_oracle as MO
    from oracle.unet_oracle import synthetic_batch
    shape = (3, 1, 8, 16, 16)
    model = fmodel.unet_model_3d(input_shape=shape[1:], depth=2, n_base_filters=8, initial_learning_rate=1e-2,
                                 loss_function=FM.weighted_dice_coefficient_loss)
    x, y = synthetic_batch(shape)
    y[1] = 0
    p0 = model.predict(x)
    want = -MO.weighted_dice_coefficient(y.astype(np.float64), p0.astype(np.float64))
    logs = model.train_on_batch(x, y)
    loss0 = logs[0] if isinstance(logs, (list, tuple)) else float(logs)
    assert abs(loss0 - want) < 2e-6, (loss0, want)
    whole_batch = -MO.dice_coefficient(y.astype(np.float64), p0.astype(np.float64))
    assert abs(loss0 - whole_batch) > 1e-3                                # it really is the per-sample form
    losses = [loss0] + [model.train_on_batch(x, y)[0] for _ in range(8)]
    assert np.isfinite(losses).all() and losses[-1] < losses[0]
