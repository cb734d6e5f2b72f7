import sys, os, random
import numpy as np
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/fetal-mri-segmentation_amd"); sys.path.insert(0, ROOT + "/tests")
import torch
from test_gpu_augment import synth_volumes, FakeDataFile
from fetal_net.device_generator import device_data_generator
default = {"flip": [0.5, 0.5, 0.5], "permute": False, "translate": (15, 15, 7), "scale": (0.1, 0.1, 0), "rotate": (0, 0, 90), "poisson_noise": 1,
           "gaussian_filter": {"prob": 0.0, "max_sigma": 1}, "contrast": {"prob": 0, "min_factor": 0.2, "max_factor": 0.1},
           "elastic_transform": {"alpha": 5, "sigma": 10},
           "coarse_dropout": {"rate": 0.2, "size_percent": [0.10, 0.30], "per_channel": True},
           "gaussian_noise": {"prob": 0.5, "sigma": 0.05}, "speckle_noise": {"prob": 0.5, "sigma": 0.05}}
vols, truths = synth_volumes(3, [(72, 72, 40), (64, 80, 36)])
df = FakeDataFile(vols, truths)
def run(aug, prefetch, skip_blank, streams):
    np.random.seed(11); random.seed(11)
    gen = device_data_generator(df, [0, 1], batch_size=2, augment=aug, patch_shape=(48, 48, 16), skip_blank=skip_blank, categorical=True, is3d=True,
                                truth_index=0, truth_size=16, noise_seed=3, prefetch=prefetch)
    got = []
    reader = torch.cuda.Stream()
    for k in range(4):
        with torch.cuda.stream(reader if (k % 2 and streams) else torch.cuda.current_stream()):
            x, y = next(gen)
            got.append((x.clone(), y.clone()))
    torch.cuda.synchronize()
    return [(x.cpu().numpy(), y.cpu().numpy()) for x, y in got]
cases = {"none": None, "geom": {k: default[k] for k in ("flip", "translate", "scale", "rotate")},
         "geom+elastic": {k: default[k] for k in ("flip", "translate", "scale", "rotate", "elastic_transform")},
         "geom+poisson": {k: default[k] for k in ("flip", "translate", "scale", "rotate", "poisson_noise")},
         "geom+noise": {k: default[k] for k in ("flip", "translate", "scale", "rotate", "gaussian_noise", "speckle_noise")},
         "geom+dropout": {k: default[k] for k in ("flip", "translate", "scale", "rotate", "coarse_dropout")},
         "full": default}
for name, aug in cases.items():
    for sb in (False, True):
        for streams in (False, True):
            a = run(aug, 0, sb, streams); b = run(aug, 1, sb, streams); c = run(aug, 0, sb, streams)
            d = [(float(np.abs(x0 - x1).max()), int((y0 != y1).sum())) for (x0, y0), (x1, y1) in zip(a, b)]
            e = [(float(np.abs(x0 - x1).max()), int((y0 != y1).sum())) for (x0, y0), (x1, y1) in zip(a, c)]
            print(name, "skip_blank", sb, "streams", streams, "0vs1", d, "0vs0", e, flush=True)
