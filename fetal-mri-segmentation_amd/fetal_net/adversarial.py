"""Adversarial and semi-supervised training of a segmentation model against a PatchGAN discriminator.

Mirrors reference fetal/experiments/train_adv.py and train_semi.py, whose loops are written against Keras objects:

    dis_model.train_on_batch(d_x, d_y)                      discriminator step, binary cross-entropy on soft labels
    combined_model.train_on_batch(g_x, g_y)                 generator step THROUGH the frozen discriminator:
        total = gd_loss_ratio * BCE(D(concat([G(x), x])), valid) + seg_loss(G(x), real_segs)         (train_adv.py:173-180)
        total = seg_loss(G(x_real), real_segs) + gd_loss_ratio * BCE(D(concat([G(x_semi), x_semi])), valid)   (train_semi.py:174-184)

Here both networks are engines on the device (the generator any `fetal_net.model` builder's engine, the discriminator a
`LayerGraphEngine` built with input_grad=True); the coupling is three kernels: fmri_discriminator_input assembles D's input from the
generator's probabilities, D's backward pass yields dL/d(input), fmri_sigmoid_chain folds its first n_labels channels into the
generator's logit gradient.  Nothing leaves HBM between the two networks.

The host-side helpers (`Scheduler`, `input2discriminator`, `input2gan`, `mul_merge_maps`, `add_noise_to_segs`, `build_dsc`) keep the
reference names, argument order and random-draw order, so a seeded run consumes numpy's global stream the same way.
"""
import os
from collections import OrderedDict

import numpy as np

from .engine_model import Adam, Model, _batch_len, _is_device_tensor


# ----------------------------------------------------------------------------------------------------------------- host helpers
class Scheduler(object):
    """step counts and learning-rate decay on a validation plateau (reference train_adv.py:36-71)"""

    def __init__(self, n_itrs_per_epoch_d, n_itrs_per_epoch_g, init_lr, lr_decay, lr_patience):
        self.init_dsteps, self.init_gsteps = n_itrs_per_epoch_d, n_itrs_per_epoch_g
        self.init_lr, self.lr_decay, self.lr_patience = init_lr, lr_decay, lr_patience
        self.dsteps, self.gsteps, self.lr = self.init_dsteps, self.init_gsteps, self.init_lr
        self.steps_stuck, self.best_loss = 0, np.inf

    def get_dsteps(self):
        return self.dsteps

    def get_gsteps(self):
        return self.gsteps

    def get_lr(self):
        return self.lr

    def update_steps(self, n_round, loss):
        if loss < self.best_loss:
            self.steps_stuck, self.best_loss = 0, loss
        else:
            self.steps_stuck += 1
        if self.steps_stuck >= self.lr_patience:
            self.lr *= self.lr_decay
            self.steps_stuck = 0
            print('Reducing LR to {}'.format(self.lr))


def build_dsc(out_labels, outs):
    return ', '.join('{}={:.3f}'.format(l, o) for l, o in zip(out_labels, outs)) + '|'


def add_noise_to_segs(segs):
    """with probability 1/2: additive N(0, .025) then multiplicative N(1, .025) noise, clipped to [0, 1] (reference train_adv.py:83-89)"""
    if np.random.choice([True, False]):
        segs = segs.astype(np.float32)
        segs += np.random.normal(0, 0.025, segs.shape)
        segs *= np.random.normal(1, 0.025, segs.shape)
        segs = np.clip(segs, a_min=0, a_max=1)
    return segs


def mul_merge_maps(r, s):
    return np.concatenate((r * s, r * (1 - s)), axis=1)


def _soft_labels(n, d_out_shape):
    return np.clip(np.random.uniform(0.9, 1.0, size=[n] + list(d_out_shape)[1:]), a_min=0, a_max=1)


class EngineLayout(object):
    """a discriminator batch already in the engine's layout ([2N][X][Y][Z][channels padded], device): passes through `_to_device_x`"""

    def __init__(self, tensor):
        self.tensor = tensor
        self.shape = tuple(tensor.shape)

    def __getitem__(self, key):                                  # batch slices (Model.evaluate)
        return EngineLayout(self.tensor[key].contiguous())


def _input2discriminator_device(real_patches, real_segs, fake_segs, d_out_shape, mul_merge, fake_patches, dis_model):
    """device tensors in (fetal_net.device_generator batches, Model.predict of a CUDA tensor): the batch is assembled in HBM by
    fmri_discriminator_input straight into the discriminator's engine layout.  The host generators are consumed in the reference's order
    (noise decision, labels); the noise FIELD itself comes from torch's device generator."""
    import torch
    from fmri_hip import ops
    if dis_model is None:
        raise TypeError("device batches need dis_model=... (the discriminator's engine layout is where they are assembled)")
    n = int(real_patches.shape[0])
    eng = dis_model.engine(2 * n)
    cp, dt_ = eng.shape[eng.input_name][0], eng.dtype
    sp = tuple(real_patches.shape[2:])
    nvox = n * int(np.prod(sp))

    def vox_major(t, dtype):                                    # (N, C, X, Y, Z) -> [nvox][C]
        t = t.to(dtype)
        return (t.reshape(n, *sp, 1) if t.shape[1] == 1 else t.permute(0, 2, 3, 4, 1)).reshape(nvox, -1).contiguous()

    segs = vox_major(real_segs, torch.float32)
    if np.random.choice([True, False]):                         # add_noise_to_segs
        gen = getattr(dis_model, "_noise_gen", None)
        if gen is None:
            gen = dis_model._noise_gen = torch.Generator(device=segs.device)
            gen.manual_seed(0)
        segs = segs + torch.randn(segs.shape, device=segs.device, generator=gen) * 0.025
        segs = (segs * (1.0 + torch.randn(segs.shape, device=segs.device, generator=gen) * 0.025)).clamp_(0.0, 1.0)
    out = torch.empty((2 * n,) + sp + (cp,), dtype=dt_, device=segs.device)
    xr = vox_major(real_patches, torch.float32)
    xf = xr if fake_patches is None else vox_major(fake_patches, torch.float32)
    ops.discriminator_input(segs, xr, out[:n], merge=bool(mul_merge))
    ops.discriminator_input(vox_major(fake_segs, torch.float32), xf, out[n:], merge=bool(mul_merge))
    if not mul_merge:
        # the reference's concatenation form puts the patches first ([x, s], train_adv.py:102-104); the kernel writes [s, x]
        L = segs.shape[1]
        C = xr.shape[1]
        out[..., :L + C] = torch.cat((out[..., L:L + C], out[..., :L]), dim=-1)
    d_y = _soft_labels(2 * n, d_out_shape)
    d_y[n:, ...] = 1 - d_y[n:, ...]
    return EngineLayout(out), d_y


def input2discriminator(real_patches, real_segs, fake_segs, d_out_shape, mul_merge=True, fake_patches=None, dis_model=None):
    """(d_x, d_y): the real pairs first, the generated ones after; labels ~U(0.9, 1) for real, 1 - U(0.9, 1) for fake
    (reference train_adv.py:97-115; train_semi.py:98-114 pairs the generated maps with their own unlabelled patches: `fake_patches`).
    CUDA tensors (with `dis_model`) are assembled on the device, see _input2discriminator_device."""
    if _is_device_tensor(real_patches):
        return _input2discriminator_device(real_patches, real_segs, fake_segs, d_out_shape, mul_merge, fake_patches, dis_model)
    fake_patches = real_patches if fake_patches is None else fake_patches
    if mul_merge:
        real = mul_merge_maps(real_patches, add_noise_to_segs(real_segs))
        fake = mul_merge_maps(fake_patches, fake_segs)
    else:
        real = np.concatenate((real_patches, add_noise_to_segs(real_segs)), axis=1)
        fake = np.concatenate((fake_patches, fake_segs), axis=1)
    d_x_batch = np.concatenate((real, fake), axis=0)
    d_y_batch = _soft_labels(d_x_batch.shape[0], d_out_shape)
    d_y_batch[real.shape[0]:, ...] = 1 - d_y_batch[real.shape[0]:, ...]
    return d_x_batch, d_y_batch


def input2discriminator_semi(real_patches, real_segs, semi_patches, semi_segs, d_out_shape, mul_merge=True):
    """the argument order of reference train_semi.py:98"""
    return input2discriminator(real_patches, real_segs, semi_segs, d_out_shape, mul_merge=mul_merge, fake_patches=semi_patches)


def input2gan(real_patches, real_segs, d_out_shape, semi_patches=None):
    """generator batch: every discriminator label says 'real'.  train_adv.py:118-124 returns (x, [valid, segs]); train_semi.py:116-123
    (`semi_patches` given) returns ([x_real, x_semi], [segs, valid])"""
    valid = _soft_labels(real_patches.shape[0], d_out_shape)
    if semi_patches is None:
        return real_patches, [valid, real_segs]
    return [real_patches, semi_patches], [real_segs, valid]


# ----------------------------------------------------------------------------------------------------------------- discriminator model
class DiscriminatorModel(Model):
    """Keras-Model duck type of `discriminator_image_3d`: train_on_batch / test_on_batch / evaluate / predict with float targets,
    loss = mean binary cross-entropy, metric 'mae' (Keras reports it as mean_absolute_error)."""

    _graph_engine = True

    @property
    def metrics_names(self):
        return ['loss'] + ['mean_absolute_error' if m in ('mae', 'MAE', 'mean_absolute_error') else (m if isinstance(m, str) else m.__name__)
                           for m in self.metrics]

    def engine(self, batch, training=True):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("no GPU visible: the fetal_net hot path has no CPU implementation")
        if self._engine is None:
            from fmri_hip.graph_engine import LayerGraphEngine
            dist_ctx = None
            try:
                import torch.distributed as dist
                if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                    from fmri_hip.dist import DataParallel
                    dist_ctx = DataParallel()
            except Exception:
                dist_ctx = None
            self._engine = LayerGraphEngine(self.layers, batch, dtype=self._compute_dtype(), training=True, dist_ctx=dist_ctx, input_grad=True)
            if self._pending_weights is not None:
                self._engine.load_keras_weights(self._pending_weights)
                self._pending_weights = None
            if getattr(self, "_pending_opt", None) is not None:
                self._apply_optimizer_state(self._pending_opt)
            if dist_ctx is not None:
                dist_ctx.broadcast_params(self._engine)
        self._engine.beta1 = float(self.optimizer.beta_1) if self.optimizer is not None else 0.9
        self._engine.set_batch(batch)
        return self._engine

    def _to_device_x(self, x):
        """(N, C, X, Y, Z) -> [N][X][Y][Z][Cp]: channels last, zero-extended to the engine's physical channel count"""
        import torch
        if isinstance(x, EngineLayout):
            return x.tensor
        t = x if _is_device_tensor(x) else torch.from_numpy(np.ascontiguousarray(np.asarray(x), dtype=np.float32)).cuda(non_blocking=True)
        t = t.permute(0, 2, 3, 4, 1)
        eng = self._engine
        cp = eng.shape[eng.input_name][0]
        if cp != t.shape[-1]:
            t = torch.nn.functional.pad(t, (0, cp - t.shape[-1]))
        return t.to(self._compute_dtype()).contiguous()

    @staticmethod
    def _to_device_target(y):
        import torch
        if _is_device_tensor(y):
            return y.float().reshape(-1).contiguous()
        return torch.from_numpy(np.ascontiguousarray(np.asarray(y), dtype=np.float32)).cuda(non_blocking=True).reshape(-1)

    @staticmethod
    def _logs(sums):
        s = [float(v) for v in sums[:3]]
        n = max(s[2], 1.0)
        return [s[0] / n, s[1] / n]

    def predict(self, x, batch_size=None, verbose=0):
        n = _batch_len(x)
        eng = self.engine(n)
        eng.predict(self._to_device_x(x))
        return eng.probs.reshape(n, -1).float().cpu().numpy()

    def train_on_batch(self, x, y, **kw):
        n = _batch_len(x)
        eng = self.engine(n)
        t = self._to_device_target(y)
        eng.forward(self._to_device_x(x))
        sums = eng.bce_forward(t).clone()
        eng.backward(t, grad_scale=(1.0 / eng.dist.world if eng.dist is not None else 1.0))
        eng.adam_step(self.optimizer.lr)
        return self._logs(sums.cpu().numpy())

    def test_on_batch(self, x, y, **kw):
        n = _batch_len(x)
        eng = self.engine(n)
        eng.forward(self._to_device_x(x), bn_training=False)
        return self._logs(eng.bce_forward(self._to_device_target(y)).cpu().numpy())

    def _check_loss(self):
        pass

    def _loss_kind(self):
        raise NotImplementedError


# ----------------------------------------------------------------------------------------------------------------- combined model
class CombinedModel(object):
    """The generator trained through the frozen discriminator.

    mode 'adv'  (train_adv.py:173-180):  train_on_batch(x, [valid, segs])                   -> [loss, dis_loss, seg_loss]
    mode 'semi' (train_semi.py:174-184): train_on_batch([x_real, x_semi], [segs, valid])    -> [loss, seg_real_loss, dis_loss]
    loss = seg_loss + gd_loss_ratio * dis_loss.  Only the generator's parameters move; the discriminator runs in training mode (its
    dropout is active, as under Keras' learning phase 1) but receives no update.  The Adam moments live in the generator's engine: in
    Keras they belong to the combined model's own optimizer, which is the only one the reference loops ever step for the generator."""

    def __init__(self, gen_model, dis_model, gd_loss_ratio=10, lr=None, mode="adv"):
        if mode not in ("adv", "semi"):
            raise ValueError(mode)
        if getattr(gen_model, "_mask_shape", None) is not None:
            raise NotImplementedError("a mask-weighted generator loss inside the combined model")
        if gen_model._input_layout != "channels_first_3d":
            raise NotImplementedError("the adversarial loop is 3-D (the reference's 2-D discriminator cannot be built)")
        self.gen, self.dis, self.ratio, self.mode = gen_model, dis_model, float(gd_loss_ratio), mode
        self.optimizer = Adam(lr if lr is not None else gen_model.optimizer.lr)
        self._d_in = {}

    @property
    def metrics_names(self):
        return ['loss', 'dis_loss', 'seg_loss'] if self.mode == "adv" else ['loss', 'seg_real_loss', 'dis_loss']

    def summary(self, print_fn=print):
        self.gen.summary(print_fn)
        self.dis.summary(print_fn)

    def _engines(self, n):
        return self.gen.engine(n), self.dis.engine(n)       # the two may compute in different dtypes (e.g. bf16 generator, fp32 discriminator)

    def _through_discriminator(self, eg, ed, xg, valid):
        """D(concat([probs, x])) forward, BCE against `valid`, backward to D's input only.  Returns (dis_loss tensor sums, dL/d(input))"""
        import torch
        from fmri_hip import ops
        N = xg.shape[0]
        key = (N, ed.shape[ed.input_name])
        buf = self._d_in.get(key)
        if buf is None:
            buf = self._d_in[key] = torch.empty(tuple(xg.shape[:-1]) + (ed.shape[ed.input_name][0],), dtype=ed.dtype, device=xg.device)
        ops.discriminator_input(eg.probs, xg, buf, merge=False)
        ed.forward(buf)
        sums = ed.bce_forward(valid).clone()
        world = ed.dist.world if ed.dist is not None else 1
        ed.backward(valid, grad_scale=self.ratio / world, params=False)
        return sums, ed.input_gradient()

    def train_on_batch(self, x, y, **kw):
        self.gen._check_loss()
        if self.mode == "adv":
            valid, segs = y
            x_seg = x_adv = x[0] if isinstance(x, (list, tuple)) else x
        else:
            (x_seg, x_adv), (segs, valid) = x, y
        n = _batch_len(x_seg)
        eg, ed = self._engines(n)
        valid_t = DiscriminatorModel._to_device_target(valid)
        yt = self.gen._to_device_y(segs)
        gs = eg.dist.grad_scale if getattr(eg, "dist", None) is not None else 1.0
        xg = self.gen._to_device_x(x_seg)
        eg.forward(xg)
        seg_sums = eg.loss_forward(yt).clone()
        if self.mode == "adv":
            d_sums, d_grad = self._through_discriminator(eg, ed, xg, valid_t)
            eg.backward(yt, grad_scale=gs, dprobs=d_grad, dprobs_scale=1.0)
        else:
            eg.backward(yt, grad_scale=gs)
            g_seg = eg.G.clone()
            xa = self.gen._to_device_x(x_adv)
            eg.forward(xa)
            eg.loss_forward(eg._dummy_y)                    # probabilities of the unlabelled batch (no loss is taken from them)
            d_sums, d_grad = self._through_discriminator(eg, ed, xa, valid_t)
            eg.backward(eg._dummy_y, grad_scale=gs, dprobs=d_grad, dprobs_scale=1.0, seg_loss=False)
            eg.G.add_(g_seg)
        eg.adam_step(self.optimizer.lr)
        seg_loss = self.gen._batch_logs(seg_sums.cpu().numpy())["loss"]
        ds = [float(v) for v in d_sums[:3].cpu().numpy()]
        dis_loss = ds[0] / max(ds[2], 1.0)
        total = seg_loss + self.ratio * dis_loss
        return [total, dis_loss, seg_loss] if self.mode == "adv" else [total, seg_loss, dis_loss]


# ----------------------------------------------------------------------------------------------------------------- the loops
def _save_generator(gen_model, base_dir, epoch, loss):
    """g_<epoch>_<loss>.json + .h5 (reference train_adv.py:266-271)"""
    stem = os.path.join(base_dir, "g_{}_{:.3f}".format(epoch, loss))
    with open(stem + ".json", 'w') as f:
        f.write(gen_model.to_json())
    gen_model.save_weights(stem + ".h5")
    return stem


def train_adversarial(config, gen_model, dis_model, train_generator, validation_generator, n_train_steps, n_validation_steps,
                      semi_generator=None, verbose=1):
    """The epoch loop of reference train_adv.py:213-287 (and, with `semi_generator`, train_semi.py:236-318): per round `dis_steps`
    discriminator steps then `gen_steps` generator steps; per epoch a validation pass, a checkpoint of the generator when its validation
    loss improved, and the Scheduler's learning-rate update.  Returns the per-epoch history (list of dicts)."""
    mode = "adv" if semi_generator is None else "semi"
    combined = CombinedModel(gen_model, dis_model, config.get("gd_loss_ratio", 10), lr=config["initial_learning_rate"], mode=mode)
    scheduler = Scheduler(config.get("dis_steps", 1), config.get("gen_steps", 1), init_lr=config["initial_learning_rate"],
                          lr_patience=config["patience"], lr_decay=config["learning_rate_drop"])
    d_out = dis_model.output_shape
    best_loss, history = np.inf, []
    for epoch in range(config["n_epochs"]):
        d_tot, g_tot, rounds = np.zeros(len(dis_model.metrics_names)), np.zeros(len(combined.metrics_names)), 0
        for n_round in range(n_train_steps // max(1, config.get("gen_steps", 1))):
            outputs = np.zeros(len(dis_model.metrics_names))
            for _ in range(scheduler.get_dsteps()):
                real_patches, real_segs = next(train_generator)[:2]
                if mode == "adv":
                    fake = gen_model.predict(real_patches, batch_size=config["batch_size"])
                    d_x, d_y = input2discriminator(real_patches, real_segs, fake, d_out, dis_model=dis_model)
                else:
                    semi_patches = next(semi_generator)[0]
                    fake = gen_model.predict(semi_patches, batch_size=config["batch_size"])
                    d_x, d_y = input2discriminator(real_patches, real_segs, fake, d_out, fake_patches=semi_patches, dis_model=dis_model)
                outputs += dis_model.train_on_batch(d_x, d_y)
            if scheduler.get_dsteps() > 0:
                d_tot += outputs / scheduler.get_dsteps()
            outputs = np.zeros(len(combined.metrics_names))
            for _ in range(scheduler.get_gsteps()):
                real_patches, real_segs = next(train_generator)[:2]
                semi_patches = next(semi_generator)[0] if mode == "semi" else None
                g_x, g_y = input2gan(real_patches, real_segs, d_out, semi_patches=semi_patches)
                outputs += combined.train_on_batch(g_x, g_y)
            if scheduler.get_gsteps() > 0:
                g_tot += outputs / scheduler.get_gsteps()
            rounds += 1
        dis_metrics = np.zeros(len(dis_model.metrics_names))
        gen_metrics = np.zeros(len(gen_model.metrics_names))
        for _ in range(n_validation_steps):
            val_patches, val_segs = next(validation_generator)[:2]
            if scheduler.get_dsteps() > 0:
                fake = gen_model.predict(val_patches, batch_size=config["validation_batch_size"])
                d_x, d_y = input2discriminator(val_patches, val_segs, fake, d_out, dis_model=dis_model)
                dis_metrics += dis_model.evaluate(d_x, d_y, batch_size=config["validation_batch_size"])
            gen_metrics += gen_model.evaluate(val_patches, val_segs, batch_size=config["validation_batch_size"])
        dis_metrics /= float(max(n_validation_steps, 1))
        gen_metrics /= float(max(n_validation_steps, 1))
        saved = None
        if gen_metrics[0] < best_loss:
            best_loss = gen_metrics[0]
            if config.get("base_dir"):
                saved = _save_generator(gen_model, config["base_dir"], epoch, gen_metrics[0])
        rec = OrderedDict(epoch=epoch, lr=scheduler.get_lr(), saved=saved)
        rec.update(("d_" + k, float(v)) for k, v in zip(dis_model.metrics_names, d_tot / max(rounds, 1)))
        rec.update(("g_" + k, float(v)) for k, v in zip(combined.metrics_names, g_tot / max(rounds, 1)))
        rec.update(("val_d_" + k, float(v)) for k, v in zip(dis_model.metrics_names, dis_metrics))
        rec.update(("val_g_" + k, float(v)) for k, v in zip(gen_model.metrics_names, gen_metrics))
        history.append(rec)
        if verbose:
            print('val_d: ' + build_dsc(dis_model.metrics_names, dis_metrics), end=' | ')
            print('val_g: ' + build_dsc(gen_model.metrics_names, gen_metrics))
        scheduler.update_steps(epoch, gen_metrics[0])
        dis_model.optimizer.lr = scheduler.get_lr()
        combined.optimizer.lr = scheduler.get_lr()
    return history
