"""Keras-`Model` duck type backed by the MI355X engine (fmri_hip.engine.UNetEngine).

Covers exactly the surface the reference's callers use: `.summary()`, `.load_weights()`, `.save()`,
`.fit_generator(...)`, `.predict()`, `.output_shape`, `.layers[i].name`, `.optimizer`, `.loss`, `.metrics`, `.compile()`,
plus `train_on_batch` / `test_on_batch` / `evaluate_generator` / `metrics_names` / `get_weights` / `set_weights`.
Loop semantics follow Keras 2.2's `fit_generator` (reference fetal_net/training.py:110-124): per-epoch logs are the
batch-size-weighted means of the per-batch values, `val_*` are means over `validation_steps` batches, callbacks see
`loss`, `binary_accuracy`, `vod_coefficient`, their `val_` twins and `lr`.
"""
import csv
import io
import json
import os
import threading
import time
from collections import OrderedDict
from collections import deque
from queue import Full, Queue

import numpy as np

from . import metrics as M


# ----------------------------------------------------------------------------------------------------------------- layers
class Layer(object):
    def __init__(self, name, class_name, output_shape, config=None, inbound=()):
        self.name = name
        self.class_name = class_name
        self.output_shape = tuple(output_shape)
        self.config = dict(config or {})
        self.inbound = list(inbound)

    def __repr__(self):
        return "<%s %s %s>" % (self.class_name, self.name, self.output_shape)


class Adam(object):
    """Optimizer token with the Keras attribute the callbacks touch (`lr`)."""

    def __init__(self, lr=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
        self.lr, self.beta_1, self.beta_2, self.epsilon = float(lr), beta_1, beta_2, epsilon

    def get_config(self):
        return dict(lr=self.lr, beta_1=self.beta_1, beta_2=self.beta_2, epsilon=self.epsilon)


# ----------------------------------------------------------------------------------------------------------------- callbacks
class Callback(object):
    def set_model(self, model):
        self.model = model

    def on_train_begin(self, logs=None): pass
    def on_train_end(self, logs=None): pass
    def on_epoch_begin(self, epoch, logs=None): pass
    def on_epoch_end(self, epoch, logs=None): pass
    def on_batch_begin(self, batch, logs=None): pass
    def on_batch_end(self, batch, logs=None): pass


class History(Callback):
    def on_train_begin(self, logs=None):
        self.epoch, self.history = [], {}

    def on_epoch_end(self, epoch, logs=None):
        self.epoch.append(epoch)
        for k, v in (logs or {}).items():
            self.history.setdefault(k, []).append(v)


class ModelCheckpoint(Callback):
    def __init__(self, filepath, monitor='val_loss', verbose=0, save_best_only=False, save_weights_only=False, mode='auto',
                 period=1):
        self.filepath, self.monitor, self.verbose = filepath, monitor, verbose
        self.save_best_only, self.save_weights_only, self.period = save_best_only, save_weights_only, period
        self.epochs_since_last_save = 0
        if mode == 'max' or (mode == 'auto' and ('acc' in monitor or monitor.startswith('fmeasure'))):
            self.monitor_op, self.best = np.greater, -np.inf
        else:
            self.monitor_op, self.best = np.less, np.inf

    def on_epoch_end(self, epoch, logs=None):
        logs = logs or {}
        self.epochs_since_last_save += 1
        if self.epochs_since_last_save < self.period:
            return
        self.epochs_since_last_save = 0
        filepath = self.filepath.format(epoch=epoch + 1, **logs)
        if self.save_best_only:
            current = logs.get(self.monitor)
            if current is None:
                return
            if self.monitor_op(current, self.best):
                if self.verbose:
                    print('\nEpoch %05d: %s improved from %0.5f to %0.5f, saving model to %s' % (epoch + 1, self.monitor, self.best, current, filepath))
                self.best = current
                self.model.save(filepath)
            elif self.verbose:
                print('\nEpoch %05d: %s did not improve from %0.5f' % (epoch + 1, self.monitor, self.best))
        else:
            self.model.save(filepath)


class CSVLogger(Callback):
    def __init__(self, filename, separator=',', append=False):
        self.filename, self.sep, self.append = filename, separator, append
        self.keys, self.append_header, self.csv_file = None, True, None

    def on_train_begin(self, logs=None):
        if self.append:
            if os.path.exists(self.filename):
                with open(self.filename, 'r') as f:
                    self.append_header = not bool(len(f.readline()))
            self.csv_file = open(self.filename, 'a')
        else:
            self.csv_file = open(self.filename, 'w')

    def on_epoch_end(self, epoch, logs=None):
        logs = logs or {}
        if self.keys is None:
            self.keys = sorted(logs.keys())
            if self.append_header:
                self.csv_file.write(self.sep.join(['epoch'] + self.keys) + '\n')
        row = [str(epoch)] + [repr(float(logs[k])) if k in logs else 'NA' for k in self.keys]
        self.csv_file.write(self.sep.join(row) + '\n')
        self.csv_file.flush()

    def on_train_end(self, logs=None):
        if self.csv_file:
            self.csv_file.close()
            self.csv_file = None


class LearningRateScheduler(Callback):
    def __init__(self, schedule, verbose=0):
        self.schedule, self.verbose = schedule, verbose

    def on_epoch_begin(self, epoch, logs=None):
        try:
            lr = self.schedule(epoch, self.model.optimizer.lr)
        except TypeError:
            lr = self.schedule(epoch)
        self.model.optimizer.lr = float(lr)

    def on_epoch_end(self, epoch, logs=None):
        if logs is not None:
            logs['lr'] = self.model.optimizer.lr


class ReduceLROnPlateau(Callback):
    def __init__(self, monitor='val_loss', factor=0.1, patience=10, verbose=0, mode='auto', min_delta=1e-4, cooldown=0, min_lr=0):
        if factor >= 1.0:
            raise ValueError('ReduceLROnPlateau does not support a factor >= 1.0.')
        self.monitor, self.factor, self.patience, self.verbose = monitor, factor, patience, verbose
        self.min_delta, self.cooldown, self.min_lr, self.mode = min_delta, cooldown, min_lr, mode
        self._reset()

    def _reset(self):
        if self.mode == 'max' or (self.mode == 'auto' and 'acc' in self.monitor):
            self.monitor_op, self.best = (lambda a, b: np.greater(a, b + self.min_delta)), -np.inf
        else:
            self.monitor_op, self.best = (lambda a, b: np.less(a, b - self.min_delta)), np.inf
        self.cooldown_counter, self.wait = 0, 0

    def on_train_begin(self, logs=None):
        self._reset()

    def on_epoch_end(self, epoch, logs=None):
        logs = logs if logs is not None else {}
        logs['lr'] = self.model.optimizer.lr
        current = logs.get(self.monitor)
        if current is None:
            return
        if self.cooldown_counter > 0:
            self.cooldown_counter -= 1
            self.wait = 0
        if self.monitor_op(current, self.best):
            self.best, self.wait = current, 0
        elif self.cooldown_counter <= 0:
            self.wait += 1
            if self.wait >= self.patience:
                old_lr = float(self.model.optimizer.lr)
                if old_lr > self.min_lr:
                    new_lr = max(old_lr * self.factor, self.min_lr)
                    self.model.optimizer.lr = new_lr
                    if self.verbose:
                        print('\nEpoch %05d: ReduceLROnPlateau reducing learning rate to %s.' % (epoch + 1, new_lr))
                    self.cooldown_counter, self.wait = self.cooldown, 0


class EarlyStopping(Callback):
    def __init__(self, monitor='val_loss', min_delta=0, patience=0, verbose=0, mode='auto'):
        self.monitor, self.patience, self.verbose, self.min_delta = monitor, patience, verbose, abs(min_delta)
        if mode == 'max' or (mode == 'auto' and 'acc' in monitor):
            self.monitor_op = np.greater
        else:
            self.monitor_op = np.less
            self.min_delta *= -1
        self.wait, self.stopped_epoch = 0, 0

    def on_train_begin(self, logs=None):
        self.wait, self.stopped_epoch = 0, 0
        self.best = np.inf if self.monitor_op == np.less else -np.inf

    def on_epoch_end(self, epoch, logs=None):
        current = (logs or {}).get(self.monitor)
        if current is None:
            return
        if self.monitor_op(current - self.min_delta, self.best):
            self.best, self.wait = current, 0
        else:
            self.wait += 1
            if self.wait >= self.patience:
                self.stopped_epoch = epoch
                self.model.stop_training = True

    def on_train_end(self, logs=None):
        if self.stopped_epoch > 0 and self.verbose:
            print('Epoch %05d: early stopping' % (self.stopped_epoch + 1))


class LambdaCallback(Callback):
    def __init__(self, **fns):
        for k, f in fns.items():
            if f is not None:
                setattr(self, k, f)


# ----------------------------------------------------------------------------------------------------------------- model
def _is_device_tensor(a):
    return type(a).__module__.startswith("torch") and getattr(a, "is_cuda", False)


def _batch_len(x):
    if isinstance(x, (list, tuple)):
        x = x[0]
    return int(x.shape[0]) if hasattr(x, "shape") else int(np.asarray(x).shape[0])


def _prefetch(generator, max_queue_size, stage=None):
    """one producer thread, bounded queue (Keras GeneratorEnqueuer with workers=1, reference training.py:115-117).  `stage` (a
    `_Stager.stage` bound method) turns a generator batch into a staged device batch ON THE PRODUCER THREAD: float64 -> fp32 / uint8
    conversion into pinned buffers and the H2D copy on a copy stream, off the thread that enqueues the training steps."""
    q = Queue(maxsize=max(1, max_queue_size))
    stop = threading.Event()

    def put(item):
        while not stop.is_set():
            try:
                q.put(item, timeout=0.1)
                return True
            except Full:
                continue
        return False

    def run():
        try:
            while not stop.is_set():
                if stage is not None:
                    b = stage(generator, stop)         # next(generator) + staging, under the copy stream
                    if b is None:                      # stopped while waiting for a free staging slot
                        return
                else:
                    b = next(generator)
                if not put(("ok", b)):
                    return
        except StopIteration:
            put(("stop", None))
        except BaseException as e:
            put(("err", e))

    th = threading.Thread(target=run, daemon=True)
    th.start()

    def get():
        kind, val = q.get()
        if kind == "err":
            raise val
        if kind == "stop":
            raise StopIteration
        return val

    class _Stop(object):
        """`set()` ends the producer and waits for it (bounded): the caller may hand the same generator - and the same staging ring - to
        the next fit_generator call right away"""
        def set(self, join=10.0):
            stop.set()
            if th is not threading.current_thread():
                th.join(timeout=join)

        def is_set(self):
            return stop.is_set()

    return get, _Stop()


def _overrides(cb, name):
    """does callback `cb` do anything in hook `name` (a subclass method, an instance attribute as LambdaCallback sets, or a foreign
    duck-typed callback)?"""
    f = getattr(cb, name, None)
    return f is not None and getattr(f, "__func__", f) is not getattr(Callback, name)


class _Staged(object):
    """one batch in HBM in the engine's layout: x (compute dtype, NDHWC / planar), y (flat uint8), optional loss weight; `ready` = HIP
    event on the copy stream behind the last copy / cast; `slot` = the staging slot to hand back once the step has been enqueued"""
    __slots__ = ("x", "y", "weight", "n", "ready", "slot", "raw")

    def __init__(self, x, y, weight, n, ready=None, slot=None, raw=None):
        self.x, self.y, self.weight, self.n, self.ready, self.slot, self.raw = x, y, weight, n, ready, slot, raw


class _Slot(object):
    __slots__ = ("pin", "dev", "free", "done")

    def __init__(self):
        self.pin, self.dev = {}, {}
        self.free = threading.Event()
        self.free.set()
        self.done = None                 # HIP event on the consumer's stream: the step that read this slot's device buffers has been enqueued up to here


class _Stager(object):
    """Producer-thread side of `fit_generator`: a ring of `depth` staging slots, each with pinned host buffers and
    device buffers that are allocated once per batch shape.  Host batches: numpy converts float64 -> fp32 (x, masks) / uint8 (y) straight
    into the pinned buffers, the H2D copies and the cast to the engine's dtype and layout run on a copy stream; the consumer only makes
    its stream wait for the slot's `ready` event.  Device batches (fetal_net.device_generator) are produced under the copy stream and
    pass through with the layout change only.  A slot is reused when the step that consumed it has finished on the device
    (`done.synchronize()` on the producer thread - never on the training thread)."""

    def __init__(self, model, depth=3):
        import torch
        self.model, self.torch = model, torch
        self.device = torch.cuda.current_device()
        self.main = torch.cuda.current_stream()
        self.copy = torch.cuda.Stream()
        self.slots = [_Slot() for _ in range(max(2, depth))]
        self.k = 0
        self._thread_ready = False
        self.owner = None              # the producer thread that fills the ring (set by its first stage() call)

    def busy(self):
        """is the ring still owned by a LIVE producer thread?  (an on_epoch_end callback that calls evaluate_generator while fit_generator's
        validation producer runs; a producer whose bounded join timed out inside a slow next(generator)) - such a ring must not be reset
        and handed to a second thread: two producers would refill slots whose H2D copy or consuming step is still in flight"""
        return self.owner is not None and self.owner.is_alive()

    def reset(self):
        """a new producer thread takes the ring over (the previous one has ended): all slots free, the caller's current stream is the consumer"""
        assert not self.busy(), "staging ring reset under a live producer thread"
        self.owner = None
        self.main = self.torch.cuda.current_stream()
        self.k = 0
        self._thread_ready = False
        for sl in self.slots:
            sl.free.set()

    def _buf(self, slot, key, shape, np_dtype, torch_dtype):
        torch = self.torch
        have = slot.pin.get(key)
        if have is None or tuple(have.shape) != tuple(shape) or have.dtype != torch_dtype:
            slot.pin[key] = torch.empty(tuple(shape), dtype=torch_dtype).pin_memory()
            slot.dev[key] = torch.empty(tuple(shape), dtype=torch_dtype, device="cuda")
        return slot.pin[key], slot.dev[key]

    def _upload(self, slot, key, arr, np_dtype, torch_dtype):
        """host array -> pinned buffer (conversion fused into the copy) -> device buffer of the slot, on the copy stream"""
        arr = np.asarray(arr)
        pin, dev = self._buf(slot, key, arr.shape, np_dtype, torch_dtype)
        np.copyto(pin.numpy(), arr, casting="unsafe")
        dev.copy_(pin, non_blocking=True)
        return dev

    def stage(self, generator, stop):
        """next(generator) -> _Staged.  The generator itself runs under the copy stream: a device generator's kernels and the casts
        behind them are ordered on it, the training stream joins through the `ready` event."""
        torch, m = self.torch, self.model
        if not self._thread_ready:
            torch.cuda.set_device(self.device)            # the producer thread issues device work: bind it to the trainer's GPU first
            self._thread_ready = True
            self.owner = threading.current_thread()
        slot = self.slots[self.k % len(self.slots)]
        self.k += 1
        while not slot.free.wait(timeout=0.1):
            if stop.is_set():
                return None
        slot.free.clear()
        if slot.done is not None:
            slot.done.synchronize()                       # the step that read this slot's device buffers is over
        with torch.cuda.stream(self.copy):
            try:
                batch = next(generator)
            except BaseException:
                slot.free.set()
                raise
            x, y = batch[0], batch[1]
            masks = None
            if isinstance(x, (list, tuple)):
                masks = x[1] if len(x) > 1 else None
                x = x[0]
            n = _batch_len(x)
            xd = x if _is_device_tensor(x) else self._upload(slot, "x", x, np.float32, torch.float32)
            yd = y if _is_device_tensor(y) else self._upload(slot, "y", y, np.uint8, torch.uint8)
            xe = m._to_device_x(xd)
            ye = m._to_device_y(yd)
            w = None
            if getattr(m.loss, "mask_weighted", False):
                if masks is None:
                    raise ValueError("this model was built with mask_shape: feed [x, masks] (reference generator.py:397-401)")
                md = masks if _is_device_tensor(masks) else self._upload(slot, "m", masks, np.float32, torch.float32)
                w = torch.exp(-md.float() / m.loss.dist_sigma).reshape(-1).contiguous()
            ready = torch.cuda.Event()
            ready.record(self.copy)
        for t in (xe, ye, w):
            if t is not None:
                t.record_stream(self.main)                # allocated under the copy stream, read on the training stream
        return _Staged(xe, ye, w, n, ready, slot, raw=(x, y))

    def consumed(self, staged):
        """training thread, after the step that reads `staged` has been enqueued: hand the slot back to the producer"""
        slot = staged.slot
        if slot is None:
            return
        if slot.done is None:
            slot.done = self.torch.cuda.Event()
        slot.done.record(self.torch.cuda.current_stream())
        slot.free.set()


class _PendingLogs(object):
    """metric sums of one step on their way to the host: a pinned fp64[16] buffer filled by an asynchronous D2H copy behind `event`.
    `values()` waits for that event only (not for the stream) and turns the sums into Keras' [loss, metric...] list."""
    __slots__ = ("model", "buf", "event", "_vals", "n")

    def __init__(self, model, buf, event, n):
        self.model, self.buf, self.event, self._vals, self.n = model, buf, event, None, n

    def done(self):
        return self._vals is not None or self.event.query()

    def values(self):
        if self._vals is None:
            self.event.synchronize()
            logs = self.model._batch_logs(self.buf.numpy().copy())
            self._vals = [logs[k] for k in self.model.metrics_names]
        return self._vals


class LazyBatchLogs(dict):
    """the `logs` dict of `on_batch_end`: `batch` and `size` are there at once, the metric values are read from the device the first
    time a callback looks at them (the reference's callbacks - checkpoint, CSV log, lr policy, early stopping - only act on epoch ends,
    so the training thread never waits for a step it has just enqueued)."""

    def __init__(self, pending, names, batch, size):
        dict.__init__(self, batch=batch, size=size)
        self._pending, self._names = pending, names

    def _force(self):
        p = self._pending
        if p is not None:
            self._pending = None
            for k, v in zip(self._names, p.values()):
                dict.__setitem__(self, k, v)

    def __getitem__(self, k):
        if k not in ("batch", "size"):
            self._force()
        return dict.__getitem__(self, k)

    def get(self, k, default=None):
        if k not in ("batch", "size"):
            self._force()
        return dict.get(self, k, default)

    def __contains__(self, k):
        if k in ("batch", "size"):
            return True
        self._force()
        return dict.__contains__(self, k)

    def __iter__(self):
        self._force()
        return dict.__iter__(self)

    def __len__(self):
        self._force()
        return dict.__len__(self)

    def keys(self):
        self._force()
        return dict.keys(self)

    def values(self):
        self._force()
        return dict.values(self)

    def items(self):
        self._force()
        return dict.items(self)

    def copy(self):
        self._force()
        return dict(self)

    def __repr__(self):
        self._force()
        return dict.__repr__(self)


class Model(object):
    """Keras-Model duck type.  `builder`/`builder_kwargs` are remembered so that `.save()` files can be re-opened by
    `load_old_model` without the original config (reference training.py:45-86)."""

    def __init__(self, layers, plan_args, builder, builder_kwargs, input_layout, name=None):
        self.layers = layers
        self.name = name or "model_1"
        self._plan_args = plan_args            # dict(in_channels, spatial, depth, n_base_filters, n_labels, ndim) or None if unsupported
        self._builder, self._builder_kwargs = builder, builder_kwargs
        self._input_layout = input_layout      # "channels_first_3d" | "channels_last_2d"
        self.inputs = [layers[0]]
        self.outputs = [layers[-1]]
        self.input_shape = layers[0].output_shape
        self.output_shape = layers[-1].output_shape
        self.optimizer, self.loss, self.metrics = None, None, []
        self.stop_training = False
        self._engine = None
        self._unsupported = None               # reason string when the engine cannot run this topology yet
        self._mask_shape = None                # shape of the second (distance-mask) input of a mask-weighted loss, when there is one
        self._pending_weights = None
        self.history = None

    # -- Keras surface -----------------------------------------------------------------------------------------------
    def compile(self, optimizer=None, loss=None, metrics=None, **kw):
        self.optimizer = optimizer if optimizer is not None else Adam()
        self.loss = loss
        self.metrics = list(metrics or [])

    @property
    def metrics_names(self):
        return ['loss'] + [m if isinstance(m, str) else getattr(m, '__name__', str(m)) for m in self.metrics]

    def count_params(self):
        return int(sum(l.config.get("params", 0) for l in self.layers))

    def summary(self, print_fn=print):
        line = "_" * 98
        print_fn(line)
        print_fn("%-34s %-30s %-12s %s" % ("Layer (type)", "Output Shape", "Param #", "Connected to"))
        print_fn("=" * 98)
        for l in self.layers:
            print_fn("%-34s %-30s %-12d %s" % ("%s (%s)" % (l.name, l.class_name), str(l.output_shape), l.config.get("params", 0),
                                               ", ".join(l.inbound)))
        print_fn("=" * 98)
        print_fn("Total params: {:,}".format(self.count_params()))
        print_fn(line)

    # -- engine ------------------------------------------------------------------------------------------------------
    def _compute_dtype(self):
        import torch
        want = (self._builder_kwargs or {}).get("compute_dtype") or os.environ.get("FMRI_DTYPE", "bf16")
        return {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "fp32": torch.float32, "float32": torch.float32}[str(want)]

    def engine(self, batch, training=True):
        if self._unsupported:
            raise NotImplementedError("the MI355X engine does not run this topology yet: " + self._unsupported)
        import torch
        from fmri_hip.engine import UNetEngine, UNetPlan
        if not torch.cuda.is_available():
            raise RuntimeError("no GPU visible: the fetal_net hot path has no CPU implementation")
        if self._engine is None:
            dist_ctx = None
            try:
                import torch.distributed as dist
                if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                    from fmri_hip.dist import DataParallel
                    dist_ctx = DataParallel()
            except Exception:
                dist_ctx = None
            if getattr(self, "_graph_engine", False):
                from fmri_hip.graph_engine import LayerGraphEngine
                self._engine = LayerGraphEngine(self.layers, batch, dtype=self._compute_dtype(), training=True, dist_ctx=dist_ctx)
            else:
                self._engine = UNetEngine(UNetPlan(**self._plan_args), batch, dtype=self._compute_dtype(), training=True, dist_ctx=dist_ctx)
            if self._pending_weights is not None:
                self._engine.load_keras_weights(self._pending_weights)
                self._pending_weights = None
            if getattr(self, "_pending_opt", None) is not None:
                self._apply_optimizer_state(self._pending_opt)
            if dist_ctx is not None:
                dist_ctx.broadcast_params(self._engine)
        self._engine.set_batch(batch)
        if self.loss is not None:
            try:
                self._engine.loss_kind, self._engine.loss_param = self._loss_kind()
            except NotImplementedError:
                pass                    # predict-only use of a model compiled with an unsupported loss
        return self._engine

    def _to_device_x(self, x):
        import torch
        eng_dtype = self._compute_dtype()
        if _is_device_tensor(x):                              # already in HBM (fetal_net.device_generator): no host round trip
            t = x
        else:
            t = torch.from_numpy(np.ascontiguousarray(np.asarray(x), dtype=np.float32)).cuda(non_blocking=True)
        if self._input_layout == "channels_first_3d":       # (N,C,X,Y,Z) -> (N,X,Y,Z,C)
            if t.shape[1] == 1:
                t = t.reshape(t.shape[0], *t.shape[2:], 1)
            else:
                t = t.permute(0, 2, 3, 4, 1)
        else:                                                # (N,X,Y,C) -> (1,N,X,Y,C): the slices of the batch are the planar D axis
            t = t.unsqueeze(0)
        return t.to(eng_dtype).contiguous()

    def _to_device_y(self, y):
        import torch
        if _is_device_tensor(y):
            t = y if y.dtype == torch.uint8 else y.to(torch.uint8)
        else:
            t = torch.from_numpy(np.ascontiguousarray(np.asarray(y)).astype(np.uint8)).cuda(non_blocking=True)
        if self._input_layout == "channels_first_3d" and t.shape[1] != 1:
            t = t.permute(0, 2, 3, 4, 1)
        return t.contiguous().reshape(-1)

    def _from_device_probs(self, eng, n):
        if self._input_layout == "channels_first_3d":
            p = eng.probs.reshape((n,) + tuple(eng.plan.level_dims(0)) + (eng.plan.n_labels,)).permute(0, 4, 1, 2, 3)
        else:                                                # (slices, X, Y, labels)
            p = eng.probs.reshape((n,) + tuple(eng.plan.spatial) + (eng.plan.n_labels,))
        return p.float().cpu().numpy()

    def predict(self, x, batch_size=None, verbose=0):
        """host array in -> host array out, as Keras; a CUDA tensor in (fetal_net.device_generator batches) -> a CUDA tensor out, in the
        model's output layout, valid until the next call on this model (it is a view of the engine's probability buffer)"""
        if isinstance(x, (list, tuple)):
            x = x[0]
        on_device = _is_device_tensor(x)
        if not on_device:
            x = np.asarray(x)
        n = x.shape[0]
        eng = self.engine(n)
        eng.predict(self._to_device_x(x))
        if on_device:
            if self._input_layout == "channels_first_3d":
                return eng.probs.reshape((n,) + tuple(eng.plan.level_dims(0)) + (eng.plan.n_labels,)).permute(0, 4, 1, 2, 3)
            return eng.probs.reshape((n,) + tuple(eng.plan.spatial) + (eng.plan.n_labels,))
        return self._from_device_probs(eng, n)

    def _loss_kind(self):
        """(kind, param) of fmri_sigmoid_loss_bwd for the compiled loss token"""
        table = {M.dice_coefficient_loss: (0, 1.0), M.binary_crossentropy_loss: (1, 1.0), M.dice_and_xent: (2, 1.0), M.focal_loss: (3, 1.0),
                 M.vod_coefficient_loss: (4, 1.0), M.double_dice_loss: (5, 10.0), M.weighted_dice_coefficient_loss: (6, 1.0)}
        if getattr(self.loss, "mask_weighted", False):       # dice_and_xent_mask(mask_input): Dice + w * mean(exp(-mask/sigma) * xent)
            return (2, self.loss.xent_weight)
        if self.loss not in table:
            raise NotImplementedError("loss %r is not differentiated on the device (available: %s)" % (
                getattr(self.loss, "__name__", self.loss), ", ".join(sorted(f.__name__ for f in table))))
        return table[self.loss]

    def _check_loss(self):
        self._loss_kind()

    def _batch_logs(self, sums):
        from fmri_hip.engine import UNetEngine
        kind, param = self._loss_kind()
        m = UNetEngine.metrics_from_sums(sums, 1.0, kind, param)
        out = OrderedDict(loss=m["loss"])
        for name in self.metrics_names[1:]:
            key = {"dice_coef": "dice_coefficient"}.get(name, name)
            out[name] = m[key]
        return out

    def _loss_weight(self, x):
        """per-voxel cross-entropy weight exp(-mask / sigma) from the model's second input (reference metrics.py:89-95), or None"""
        if not getattr(self.loss, "mask_weighted", False):
            return None
        import torch
        if not isinstance(x, (list, tuple)) or len(x) < 2:
            raise ValueError("this model was built with mask_shape: feed [x, masks] (reference generator.py:397-401)")
        m = x[1]
        t = m if _is_device_tensor(m) else torch.from_numpy(np.ascontiguousarray(np.asarray(m), dtype=np.float32)).cuda(non_blocking=True)
        return torch.exp(-t.float() / self.loss.dist_sigma).reshape(-1).contiguous()

    # -- steps: enqueue now, read the metric sums later --------------------------------------------------------------------------
    LOG_RING = 8          # pinned fp64[16] buffers for the metric sums in flight (the training thread runs at most this many steps ahead)

    def _stage_inline(self, x, y):
        """(x, y) of a direct train_on_batch / test_on_batch call -> _Staged on the calling thread's stream"""
        weight = self._loss_weight(x)
        if isinstance(x, (list, tuple)):
            x = x[0]
        return _Staged(self._to_device_x(x), self._to_device_y(y), weight, _batch_len(x))

    def _step_async(self, staged, train=True):
        """enqueue one training (or evaluation) step on the current stream and the D2H copy of its metric sums behind it; nothing here
        waits for the device.  -> _PendingLogs"""
        import torch
        eng = self.engine(staged.n)
        if staged.ready is not None:
            torch.cuda.current_stream().wait_event(staged.ready)
        if train:
            sums = eng.train_step(staged.x, staged.y, self.optimizer.lr, weight=staged.weight)
        else:
            eng.forward(staged.x, bn_training=False)                # Keras evaluates with learning_phase = 0
            sums = eng.loss_forward(staged.y, staged.weight)
        ring = self.__dict__.get("_log_ring")
        if ring is None:
            ring = self.__dict__["_log_ring"] = dict(k=0, slots=[[torch.empty(16, dtype=torch.float64).pin_memory(), torch.cuda.Event(), None]
                                                                 for _ in range(self.LOG_RING)])
        slot = ring["slots"][ring["k"] % self.LOG_RING]
        ring["k"] += 1
        if slot[2] is not None:
            slot[2].values()                                         # the buffer's previous owner reads it before it is overwritten
        slot[0].copy_(sums, non_blocking=True)
        slot[1].record(torch.cuda.current_stream())
        slot[2] = _PendingLogs(self, slot[0], slot[1], staged.n)
        return slot[2]

    def train_on_batch(self, x, y, **kw):
        self._check_loss()
        return list(self._step_async(self._stage_inline(x, y), train=True).values())

    def test_on_batch(self, x, y, **kw):
        return list(self._step_async(self._stage_inline(x, y), train=False).values())

    def evaluate(self, x, y, batch_size=None, verbose=0):
        """Keras `model.evaluate(x, y, batch_size)`: batch-size-weighted means of test_on_batch (reference fetal/experiments/train_adv.py:253-262)"""
        n, bs = _batch_len(x), int(batch_size or 32)
        outs, sizes = [], []
        for i in range(0, n, bs):
            xs = [a[i:i + bs] for a in x] if isinstance(x, (list, tuple)) else x[i:i + bs]
            outs.append(self.test_on_batch(xs, y[i:i + bs]))
            sizes.append(_batch_len(xs))
        return [float(np.average([o[k] for o in outs], weights=sizes)) for k in range(len(outs[0]))]

    def to_json(self, **kw):
        """the Keras model_config document of the recorded layer graph (reference fetal/experiments/train_adv.py:269-270)"""
        from . import keras_h5
        doc = keras_h5.model_config(self)
        doc.update(keras_version=keras_h5.KERAS_VERSION, backend=keras_h5.BACKEND)
        return json.dumps(doc, **kw)

    def _staging(self, max_queue_size, role="train"):
        """the staging ring for the producer thread of fit_generator / evaluate_generator (one per role, kept on the model: its pinned
        and device buffers outlive the call); FMRI_STAGE_PREFETCH=0 keeps the round-3 behaviour (conversion and pageable upload on the
        training thread) for A/B runs"""
        if os.environ.get("FMRI_STAGE_PREFETCH", "1") == "0" or self._unsupported:
            return None
        import torch
        if not torch.cuda.is_available():
            return None
        cache = self.__dict__.setdefault("_stagers", {})
        st = cache.get(role)
        if st is None or st.device != torch.cuda.current_device() or st.busy():
            # (busy: the cached ring still belongs to a live producer - leave it to that thread and give this call a ring of its own)
            st = cache[role] = _Stager(self, depth=int(os.environ.get("FMRI_STAGE_DEPTH", "3")))
        st.reset()
        return st

    def _as_staged(self, batch):
        if isinstance(batch, _Staged):
            return batch
        return self._stage_inline(batch[0], batch[1])

    def evaluate_generator(self, generator, steps, max_queue_size=10, workers=1, use_multiprocessing=False, verbose=0):
        stager = self._staging(max_queue_size, "val")
        get, stop = _prefetch(generator, max_queue_size, stager.stage if stager else None)
        pend = []
        try:
            for _ in range(steps):
                b = self._as_staged(get())
                pend.append(self._step_async(b, train=False))
                if stager:
                    stager.consumed(b)
        finally:
            stop.set()
        outs, sizes = [p.values() for p in pend], [p.n for p in pend]
        return [float(np.average([o[i] for o in outs], weights=sizes)) for i in range(len(outs[0]))]

    def fit_generator(self, generator, steps_per_epoch=None, epochs=1, verbose=1, callbacks=None, validation_data=None,
                      validation_steps=None, class_weight=None, max_queue_size=10, workers=1, use_multiprocessing=False,
                      shuffle=True, initial_epoch=0):
        """Keras 2.2 `fit_generator` loop (reference training.py:110-124).  The training thread only ENQUEUES: batches arrive staged in
        HBM from the producer thread (`_Stager`), the metric sums of step k are read back from a pinned buffer a few steps later
        (`_PendingLogs`), a callback that looks at a batch log forces that one read (`LazyBatchLogs`); epoch logs are complete before
        `on_epoch_end`, as in Keras."""
        self._check_loss()
        self.history = History()
        cbs = list(callbacks or []) + [self.history]
        for cb in cbs:
            cb.set_model(self)
        batch_cbs = [cb for cb in cbs if _overrides(cb, "on_batch_end") or _overrides(cb, "on_batch_begin")]
        self.stop_training = False
        names = self.metrics_names
        stager = self._staging(max_queue_size)
        stage = stager.stage if stager else None
        get, stop = _prefetch(generator, max_queue_size, stage)
        vget, vstop = (None, None)
        if validation_data is not None and not isinstance(validation_data, (tuple, list)):
            vstager = self._staging(max_queue_size, "val")
            vget, vstop = _prefetch(validation_data, max_queue_size, vstager.stage if vstager else None)
        lag = max(1, self.LOG_RING // 2)
        for cb in cbs:
            cb.on_train_begin({})
        try:
            for epoch in range(initial_epoch, epochs):
                for cb in cbs:
                    cb.on_epoch_begin(epoch, {})
                t0 = time.time()
                tot, seen = np.zeros(len(names)), 0
                pend = deque()

                def drain(keep):
                    nonlocal tot, seen
                    while len(pend) > keep:
                        p = pend.popleft()
                        tot += np.asarray(p.values()) * p.n
                        seen += p.n

                for step in range(steps_per_epoch):
                    b = self._as_staged(get())
                    for cb in batch_cbs:
                        cb.on_batch_begin(step, {"batch": step, "size": b.n})
                    p = self._step_async(b, train=True)
                    if stager:
                        stager.consumed(b)
                    pend.append(p)
                    if batch_cbs:
                        blog = LazyBatchLogs(p, names, step, b.n)
                        for cb in batch_cbs:
                            cb.on_batch_end(step, blog)
                    drain(lag)                                   # step k - lag has long finished: reading it does not stall the queue
                    if self.stop_training:
                        break
                drain(0)
                logs = OrderedDict((k, float(v)) for k, v in zip(names, tot / max(seen, 1)))
                if validation_data is not None:
                    if vget is not None:
                        vp = []
                        for _ in range(validation_steps):
                            vb = self._as_staged(vget())
                            vp.append(self._step_async(vb, train=False))
                            if vstager:
                                vstager.consumed(vb)
                        vouts, vsizes = [q.values() for q in vp], [q.n for q in vp]
                        vals = [float(np.average([o[i] for o in vouts], weights=vsizes)) for i in range(len(names))]
                    else:
                        vals = self.test_on_batch(validation_data[0], validation_data[1])
                    for k, v in zip(names, vals):
                        logs['val_' + k] = float(v)
                for cb in cbs:
                    cb.on_epoch_end(epoch, logs)
                if verbose:
                    print("Epoch %d/%d - %.1fs - %s" % (epoch + 1, epochs, time.time() - t0,
                                                        " - ".join("%s: %.4f" % kv for kv in logs.items())))
                if self.stop_training:
                    break
        finally:
            stop.set()
            if vstop is not None:
                vstop.set()
            for cb in cbs:
                cb.on_train_end({})
        return self.history

    # -- weights / checkpoints ---------------------------------------------------------------------------------------
    def get_weights_dict(self):
        if self._engine is not None:
            return self._engine.export_keras_weights()
        if self._pending_weights is not None:
            return self._pending_weights
        return self.engine(1).export_keras_weights()

    def set_weights_dict(self, W):
        if self._engine is not None:
            self._engine.load_keras_weights(W)
        else:
            self._pending_weights = OrderedDict((k, np.asarray(v)) for k, v in W.items())

    def get_optimizer_state(self):
        """(m, v, iterations) with m / v as {'<layer>/<key>': ndarray in Keras layout}, or None before the first training step"""
        eng = self._engine
        if eng is None or not eng.training or eng.t == 0:
            return getattr(self, "_pending_opt", None)
        m = eng.flat_to_keras(eng.M.detach().cpu().numpy(), moving=False)
        v = eng.flat_to_keras(eng.V.detach().cpu().numpy(), moving=False)
        return m, v, int(eng.t)

    def _apply_optimizer_state(self, state):
        eng = self._engine
        if eng is None or not eng.training:
            self._pending_opt = state
            return
        import torch
        m, v, t = state
        Wz = eng.flat_to_keras(np.zeros(eng.n_flat, np.float32), moving=False)       # zeros for anything the state does not name
        eng.M.copy_(torch.from_numpy(eng.keras_to_flat(dict(Wz, **m))))
        eng.V.copy_(torch.from_numpy(eng.keras_to_flat(dict(Wz, **v))))
        eng.t = int(t)
        self._pending_opt = None

    def save_weights(self, path):
        self.save(path, include_optimizer=False, weights_only=True)

    def _checkpoint_meta(self):
        return dict(format="fmri-npz-1", builder=self._builder, builder_kwargs=_jsonable(self._builder_kwargs),
                    optimizer=self.optimizer.get_config() if self.optimizer else None,
                    loss=getattr(self.loss, "__name__", None),
                    metrics=[m if isinstance(m, str) else getattr(m, "__name__", str(m)) for m in self.metrics])

    def save(self, path, include_optimizer=True, weights_only=False):
        """Full-model checkpoint in the reference's container: a Keras 2.2 HDF5 file (reference training.py:31-32 via ModelCheckpoint;
        layout restated in keras_h5.py) written through the system libhdf5.  Where no libhdf5 can be loaded the container is a numpy
        .npz under the SAME file name (naming and resume-by-mtime keep working: reference fetal/utils.py:42-43); `load_weights` and
        `load_old_model` tell the two apart by their magic bytes."""
        from .utils import hdf5
        tmp = path + ".tmp"
        if hdf5.available() and os.environ.get("FMRI_CHECKPOINT_FORMAT", "h5") != "npz":
            from . import keras_h5
            keras_h5.save_model(self, tmp, include_optimizer=include_optimizer, weights_only=weights_only,
                                extra_meta=self._checkpoint_meta())
            os.replace(tmp, path)
            return
        W = self.get_weights_dict()
        arrays = {"w/" + k: v for k, v in W.items()}
        meta = self._checkpoint_meta()
        state = self.get_optimizer_state() if include_optimizer else None
        if state is not None:
            for k, a in state[0].items():
                arrays["opt_m/" + k] = a
            for k, a in state[1].items():
                arrays["opt_v/" + k] = a
            meta["opt_t"] = state[2]
        arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
        with open(tmp, "wb") as f:
            np.savez(f, **arrays)
        os.replace(tmp, path)

    def load_weights(self, path, by_name=False):
        from .utils import hdf5
        if hdf5.is_hdf5(path):
            from . import keras_h5
            self.set_weights_dict(keras_h5.map_weights(self, keras_h5.read_weights(path)))
            try:
                state = keras_h5.read_optimizer(path, self)
            except ValueError as e:
                # Keras (saving.py load_model): "Error in loading the saved optimizer state. As a result, your model is starting with a
                # freshly initialized optimizer." - a warning, never a failed load_weights
                import warnings
                warnings.warn("optimizer state of %s not restored (%s): the optimizer starts fresh" % (path, e))
                state = None
        else:
            z = np.load(path, allow_pickle=False)
            self.set_weights_dict(OrderedDict((k[2:], z[k]) for k in z.files if k.startswith("w/")))
            state = None
            if any(k.startswith("opt_m/") for k in z.files):
                t = int(json.loads(bytes(z["meta"]).decode()).get("opt_t", 0))
                state = (OrderedDict((k[6:], z[k]) for k in z.files if k.startswith("opt_m/")),
                         OrderedDict((k[6:], z[k]) for k in z.files if k.startswith("opt_v/")), t)
        if state is not None:
            self._apply_optimizer_state(state)
        return self


def read_checkpoint_meta(path):
    """the builder call recorded in a checkpoint: this stack's own record when present, else inferred from the Keras model_config"""
    from .utils import hdf5
    if hdf5.is_hdf5(path):
        from . import keras_h5
        mc, tc, own = keras_h5.read_configs(path)
        if own is not None:
            return own
        if mc is None:
            raise ValueError("%s is a weights-only HDF5 file (no model_config): build the model and call load_weights" % path)
        name, kwargs = keras_h5.infer_builder(mc, tc)
        opt = tc["optimizer_config"]["config"] if tc else None
        return dict(format="keras-h5", builder=name, builder_kwargs=kwargs, optimizer=opt)
    z = np.load(path, allow_pickle=False)
    return json.loads(bytes(z["meta"]).decode())


def _jsonable(d):
    out = {}
    for k, v in (d or {}).items():
        if callable(v):
            out[k] = {"__callable__": getattr(v, "__name__", str(v))}
        elif isinstance(v, (np.integer,)):
            out[k] = int(v)
        elif isinstance(v, (np.floating,)):
            out[k] = float(v)
        elif isinstance(v, (tuple, list)):
            out[k] = [int(a) if isinstance(a, (int, np.integer)) else a for a in v]
        else:
            out[k] = v
    return out
