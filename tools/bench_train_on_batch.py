import sys, time, os
sys.path.insert(0, "/root/repo/fetal-mri-segmentation_amd")
import numpy as np, torch
from fetal_net.model import unet_model_3d
patch, B = (64, 128, 128), 4
model = unet_model_3d(input_shape=(1,) + patch, depth=4, n_base_filters=32, initial_learning_rate=1e-4)
g = torch.Generator().manual_seed(0)
x = torch.randn((B, 1) + patch, generator=g).cuda()
y = (torch.rand((B, 1) + patch, generator=g) > 0.7).to(torch.uint8).cuda()
for _ in range(5): model.train_on_batch(x, y)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(30): model.train_on_batch(x, y)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
print("train_on_batch (sync per step): %.2f ms  %.1f patches/s" % (dt * 1e3, B / dt))
eng = model._engine
xd, yd = model._to_device_x(x), model._to_device_y(y)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(30): eng.train_step(xd, yd, 1e-4)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
print("engine.train_step (no sync):   %.2f ms  %.1f patches/s" % (dt * 1e3, B / dt))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(30):
    s = eng.train_step(xd, yd, 1e-4); s.cpu()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
print("engine.train_step + sums.cpu(): %.2f ms  %.1f patches/s" % (dt * 1e3, B / dt))
t0 = time.perf_counter()
for _ in range(30): a = model._to_device_x(x); b = model._to_device_y(y)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
print("_to_device x,y: %.3f ms" % (dt * 1e3))
