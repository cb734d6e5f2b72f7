"""A LEARNABLE synthetic segmentation task for the val-Dice leg of the metric (SURVEY 8d "val Dice", VERDICT r3 row g).

bench.py's throughput batch draws image and label from independent noise fields: nothing to learn but the base rate.  Here both come
from one latent field, as tissue and intensity do in an MRI volume:

    f = gaussian_filter(N(0,1), sigma)            smooth latent field
    y = f > quantile(f, 1 - fg)                   label: blobs, 12 % foreground
    x = zscore(contrast * y + N(0,1))             image: the blobs are brighter by `contrast` noise sigmas, then z-scored like the
                                                  reference's volumes (normalize.py:66-83)

A patch is a pure function of its seed: training seeds 0, 1, 2, ..., held-out seeds HELD_OUT + k never meet.

Why 12 % foreground and not the 30 % of bench.py's throughput batch: with the reference's Dice loss (metrics.py:11-15, smooth 1) and Keras'
zero-bias glorot start every voxel sits at p = 0.5, where the loss gradient has a common-mode part that pushes ALL logits up; its size
relative to the discriminating part is the foreground fraction.  Measured on the CPU oracle (depth 4 / 32 filters, lr 1e-4): at 30 %
foreground the 1x32x64x128 run - and the recipe y = gaussian_filter(x) > quantile at any size - falls into the all-foreground solution,
Dice = 2 fg / (1 + fg) = 0.4615, within 25 steps and never leaves it; at 12 % (a fetal brain fills about that much of its patch) the
same run passes 0.85 training Dice after 30 steps, with contrast 1.5 as well as 2.5.

Host form (numpy / scipy): bit-identical inputs for the CPU oracle and both GPU engines in tests/test_gpu_val_dice.py.  Device form
(torch RNG + the library's own gaussian kernel): full-size batches for bench.py without a host round trip.
Definitions: soft Dice = reference fetal_net/metrics.py:11-15 (= -val_loss of the Keras log); hard Dice = reference fetal/evaluate.py:16-17
on p > 0.5 (threshold of metrics.py:22-24).
"""
import numpy as np

HELD_OUT = 1_000_000          # first held-out seed
SIGMA, FG, CONTRAST = 3.0, 0.12, 2.5


def host_patch(seed, spatial, sigma=SIGMA, fg=FG, contrast=CONTRAST):
    """-> x float32 [X,Y,Z] (z-scored), y uint8 [X,Y,Z]"""
    from scipy.ndimage import gaussian_filter
    rs = np.random.RandomState(seed % (2 ** 32))
    f = gaussian_filter(rs.randn(*spatial), [min(sigma, s / 8.0) for s in spatial])
    y = f > np.quantile(f, 1.0 - fg)
    x = contrast * y + rs.randn(*spatial)
    x = (x - x.mean()) / x.std()
    return x.astype(np.float32), y.astype(np.uint8)


def host_batch(first_seed, n, spatial, dtype=np.float64):
    """reference generator contract (generator.py:397-401): x (N,1,X,Y,Z) float64, y (N,1,X,Y,Z) uint8"""
    xs, ys = zip(*(host_patch(first_seed + i, spatial) for i in range(n)))
    return np.stack(xs)[:, None].astype(dtype), np.stack(ys)[:, None]


def host_generator(first_seed, n, spatial, steps=None, dtype=np.float64):
    """infinite (or `steps` long) generator of reference-style batches: batch k holds seeds first_seed + k*n ... + n - 1"""
    k = 0
    while steps is None or k < steps:
        yield host_batch(first_seed + k * n, n, spatial, dtype)
        k += 1


def _gauss_threshold(f, fg):
    """the (1 - fg) quantile of a smoothed N(0,1) field from its mean and standard deviation (the field is Gaussian): two reductions
    instead of a 1M-element selection (torch.kthvalue took 4 ms per patch - 0.85 ms per step of a profiled bench run went to the pool's
    creation).  The foreground fraction is fg up to the field's sampling error (a few tenths of a percent)."""
    from statistics import NormalDist
    return f.mean() + NormalDist().inv_cdf(1.0 - fg) * f.std(unbiased=False)


def device_patch(seed, spatial, device="cuda", sigma=SIGMA, fg=FG, contrast=CONTRAST):
    """the same recipe on the device: torch's device RNG, fmri_correlate1d_f32 (mode 'nearest') for the smoothing.  Not bit-identical to
    host_patch (other RNG, other border rule) - the same distribution.  -> x float32 [X,Y,Z], y uint8 [X,Y,Z]"""
    import torch
    from fmri_hip import ops
    g = torch.Generator(device=device).manual_seed(int(seed))
    f = torch.randn(spatial, generator=g, device=device, dtype=torch.float32)
    f = ops.gaussian_filter_f32(f, [min(sigma, s / 8.0) for s in spatial])
    y = f > _gauss_threshold(f, fg)
    x = contrast * y.to(torch.float32) + torch.randn(spatial, generator=g, device=device, dtype=torch.float32)
    x = (x - x.mean()) / x.std(unbiased=False)
    return x, y.to(torch.uint8)


def device_batch(first_seed, n, spatial, device="cuda"):
    """-> x float32 (N,1,X,Y,Z), y uint8 (N,1,X,Y,Z) CUDA tensors (Model.train_on_batch / fit_generator take them as they are)"""
    import torch
    xs, ys = zip(*(device_patch(first_seed + i, spatial, device) for i in range(n)))
    return torch.stack(xs)[:, None].contiguous(), torch.stack(ys)[:, None].contiguous()


def device_batch_2d(first_seed, n, plane, channels, device="cuda", sigma=SIGMA, fg=FG, contrast=CONTRAST):
    """2-D models (reference fetal_net/model/unet/unet.py: x (N,X,Y,C) slice stacks, y (N,X,Y,1) = the label of the middle slice): every slice of a
    stack carries its own in-plane blobs and is brighter on them, the label is the middle slice's.  -> x float32 (N,X,Y,C), y uint8 (N,X,Y,1)"""
    import torch
    from fmri_hip import ops
    xs, ys = [], []
    for i in range(n):
        g = torch.Generator(device=device).manual_seed(int(first_seed + i))
        shape = tuple(plane) + (channels,)
        f = ops.gaussian_filter_f32(torch.randn(shape, generator=g, device=device, dtype=torch.float32), [sigma, sigma, 0.0])
        yall = f > _gauss_threshold(f, fg)
        x = contrast * yall.to(torch.float32) + torch.randn(shape, generator=g, device=device, dtype=torch.float32)
        xs.append((x - x.mean()) / x.std(unbiased=False))
        ys.append(yall[..., channels // 2:channels // 2 + 1].to(torch.uint8))
    return torch.stack(xs).contiguous(), torch.stack(ys).contiguous()


def device_generator(first_seed, n, spatial, steps=None, device="cuda"):
    k = 0
    while steps is None or k < steps:
        yield device_batch(first_seed + k * n, n, spatial, device)
        k += 1


def soft_dice(y, p, smooth=1.0):
    """reference metrics.py:11-15 on numpy arrays (whole-tensor flatten)"""
    y, p = np.asarray(y, np.float64).reshape(-1), np.asarray(p, np.float64).reshape(-1)
    return float((2.0 * (y * p).sum() + smooth) / (y.sum() + p.sum() + smooth))


def hard_dice(truth, prediction):
    """reference fetal/evaluate.py:16-17 on binary masks"""
    truth, prediction = np.asarray(truth, np.float64), np.asarray(prediction, np.float64)
    return float(2.0 * np.sum(truth * prediction) / (np.sum(truth) + np.sum(prediction)))
