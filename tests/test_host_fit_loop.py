"""`fit_generator` / `evaluate_generator` loop semantics on the CPU with a stand-in for the device step (the engine itself needs a GPU):
Keras 2.2 epoch logs (batch-size-weighted means, reference fetal_net/training.py:110-124) out of DEFERRED metric reads - the training
thread enqueues step k and reads step k - lag; a callback that looks at a batch log forces exactly that read (engine_model.LazyBatchLogs)."""
import numpy as np
import pytest

import fetal_net.model as fmodel
from fetal_net.engine_model import (Callback, LambdaCallback, LazyBatchLogs, _Staged, _overrides, _prefetch)


class _FakePending:
    def __init__(self, vals, n, log):
        self._v, self.n, self._log, self.forced = vals, n, log, False

    def values(self):
        if not self.forced:
            self.forced = True
            self._log.append(self)
        return self._v


def _model(monkeypatch, per_step):
    """unet_model_3d whose device step is replaced: step k reports per_step[k] = ([loss, acc, vod], batch size)"""
    m = fmodel.unet_model_3d(input_shape=(1, 8, 8, 8), depth=2, n_base_filters=8)
    state = dict(k=0, forced=[], made=[])

    def step_async(staged, train=True):
        vals, n = per_step[state["k"] % len(per_step)]
        state["k"] += 1
        p = _FakePending(list(vals), n, state["forced"])
        state["made"].append(p)
        return p

    monkeypatch.setattr(m, "_staging", lambda mq, role="train": None)
    monkeypatch.setattr(m, "_as_staged", lambda b: _Staged(None, None, None, int(np.asarray(b[0]).shape[0])))
    monkeypatch.setattr(m, "_step_async", step_async)
    return m, state


def _gen(sizes):
    k = 0
    while True:
        n = sizes[k % len(sizes)]
        k += 1
        yield np.zeros((n, 1, 8, 8, 8)), np.zeros((n, 1, 8, 8, 8), np.uint8)


def test_epoch_logs_are_batch_size_weighted_means_of_deferred_reads(monkeypatch):
    per = [([-0.5, 0.9, 0.3], 4), ([-0.7, 0.8, 0.5], 2), ([-0.2, 0.7, 0.1], 4)]
    m, st = _model(monkeypatch, per)
    seen_at_batch_end = []
    cb = LambdaCallback(on_batch_end=lambda step, logs: seen_at_batch_end.append((step, logs["batch"], logs["size"], len(st["forced"]))))
    h = m.fit_generator(_gen([4, 2, 4]), steps_per_epoch=3, epochs=2, verbose=0, callbacks=[cb]).history
    w = np.array([4, 2, 4], float)
    for i, k in enumerate(m.metrics_names):
        want = float(np.average([p[0][i] for p in per], weights=w))
        assert h[k] == pytest.approx([want, want], abs=1e-12), k
    # 'batch' / 'size' are available without touching the device; nothing had been forced when the callbacks of the first steps ran
    assert [s[:3] for s in seen_at_batch_end[:3]] == [(0, 0, 4), (1, 1, 2), (2, 2, 4)]
    assert seen_at_batch_end[0][3] == 0 and seen_at_batch_end[1][3] == 0
    assert all(p.forced for p in st["made"])                      # ... and every step's sums were read by the epoch end


def test_a_callback_reading_a_batch_log_forces_that_step_only(monkeypatch):
    per = [([-0.1 * (k + 1), 0.5, 0.5], 2) for k in range(6)]
    m, st = _model(monkeypatch, per)
    got = []

    class Reader(Callback):
        def on_batch_end(self, batch, logs=None):
            if batch == 1:
                got.append((logs["loss"], dict(logs)))
                assert st["made"][1].forced and not st["made"][0].forced          # the read step, not its predecessor

    m.fit_generator(_gen([2]), steps_per_epoch=6, epochs=1, verbose=0, callbacks=[Reader()])
    assert got[0][0] == pytest.approx(-0.2)
    assert set(got[0][1]) == {"batch", "size", "loss", "binary_accuracy", "vod_coefficient"}


def test_stop_training_inside_an_epoch_and_validation_means(monkeypatch):
    per = [([-0.5, 0.9, 0.3], 2)]
    m, st = _model(monkeypatch, per)

    class Stop(Callback):
        def on_batch_end(self, batch, logs=None):
            if batch == 1:
                self.model.stop_training = True

    h = m.fit_generator(_gen([2]), steps_per_epoch=5, epochs=3, verbose=0, callbacks=[Stop()], validation_data=_gen([2, 4]),
                        validation_steps=2).history
    assert len(h["loss"]) == 1 and st["k"] == 2 + 2                # two training steps, then the epoch's validation, then out
    assert h["val_loss"] == [pytest.approx(-0.5)]
    out = m.evaluate_generator(_gen([2, 4]), steps=4)
    assert out == pytest.approx([-0.5, 0.9, 0.3])


def test_lazy_batch_logs_behaves_like_a_dict():
    forced = []
    p = _FakePending([1.0, 2.0], 3, forced)
    logs = LazyBatchLogs(p, ["loss", "acc"], batch=7, size=3)
    assert logs["batch"] == 7 and logs.get("size") == 3 and "batch" in logs and not forced
    assert "loss" in logs and forced == [p]
    assert dict(logs) == {"batch": 7, "size": 3, "loss": 1.0, "acc": 2.0} and len(logs) == 4
    assert sorted(logs.keys()) == ["acc", "batch", "loss", "size"] and logs.copy()["acc"] == 2.0


def test_override_detection_and_producer_shutdown():
    class Foreign:                                                  # duck-typed callback that is not a subclass
        def on_batch_end(self, batch, logs=None): pass
    assert _overrides(Foreign(), "on_batch_end") and not _overrides(Callback(), "on_batch_end")
    assert _overrides(LambdaCallback(on_batch_end=lambda b, l: None), "on_batch_end")
    assert not _overrides(LambdaCallback(on_epoch_end=lambda e, l: None), "on_batch_end")
    # a producer blocked on a full queue ends when the loop sets `stop`
    get, stop = _prefetch(iter(range(10 ** 9)), 2)
    assert get() == 0
    stop.set()

    def boom():
        yield 1
        raise RuntimeError("generator failed")
    get, stop = _prefetch(boom(), 2)
    assert get() == 1
    with pytest.raises(RuntimeError):
        get()


def test_a_staging_ring_with_a_live_producer_is_not_handed_to_a_second_thread():
    """`_staging()` reset the cached ring unconditionally - under a live producer (an evaluate_generator call from inside
    on_epoch_end, a producer whose bounded join timed out) two threads would share slot buffers.  The ring records its producer; a ring
    that is still owned is left alone (`busy()`), and resetting one asserts."""
    import threading
    from fetal_net.engine_model import _Stager
    st = object.__new__(_Stager)
    st.owner = None
    assert not st.busy()
    release = threading.Event()
    th = threading.Thread(target=release.wait, daemon=True)
    th.start()
    st.owner = th
    assert st.busy()
    with pytest.raises(AssertionError):
        st.reset()
    release.set()
    th.join()
    assert not st.busy()
