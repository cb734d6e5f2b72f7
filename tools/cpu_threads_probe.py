import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import unet_oracle as O
spec = O.Spec((1, 32, 64, 64)); W = spec.init_weights(42); x, y = O.synthetic_batch((1, 1, 32, 64, 64))
for nt in (16, 32, 64, 128, 256):
    torch.set_num_threads(nt)
    opt = O.KerasAdam(W, lr=1e-4, dtype=np.float32)
    O.train_step(spec, W, opt, x, y, dtype=torch.float32)
    t0 = time.time(); O.train_step(spec, W, opt, x, y, dtype=torch.float32); dt = time.time() - t0
    print("threads", nt, "step %.2f s on 1/16 patch -> %.4f patches/s equiv" % (dt, 1 / dt / 16), flush=True)
