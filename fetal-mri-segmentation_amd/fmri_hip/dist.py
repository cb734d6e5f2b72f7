"""Data-parallel context: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI; "gloo" on CPU tests).

The reference has no multi-device code at all; the path shards as pure data parallel over independent
patches, so the only exchanges are
  (1) the three Dice sums (metrics.py:11-15 flattens the WHOLE batch, so the global-batch loss needs global sums), and
  (2) the gradient all-reduce, bucketed over the flat gradient buffer.  The buffer is laid out in backward-completion
      order, so a bucket is a contiguous range that closes as soon as the wgrad of its last layer has been enqueued;
      each bucket is reduced on a side stream behind an event, overlapping the rest of the backward pass.
xGMI is point-to-point (7 links x ~153 GB/s): a 65 MB fp32 gradient ring-reduces in < 1 ms against >= 9 ms of backward,
so a handful of ~16 MB buckets is enough to hide it; more, smaller buckets only add launch latency.
"""
import torch
import torch.distributed as dist


class DataParallel:
    def __init__(self, world=None, rank=None, bucket_bytes=16 << 20, global_dice=True, group=None, force_collectives=False):
        self.world = dist.get_world_size(group) if world is None else world
        self.rank = dist.get_rank(group) if rank is None else rank
        self.group = group
        self.bucket_elems = max(1, bucket_bytes // 4)
        self.global_dice = global_dice
        self.force = force_collectives      # issue the collectives even in a 1-rank group (exercises RCCL on a single GPU)
        self._start = 0
        self._comm = None
        self.launched = []          # (start, end) ranges reduced in the last backward (introspection / tests)

    # gradients are SUMMED when the loss used global sums (exact global-batch Dice), averaged otherwise
    @property
    def grad_scale(self):
        return 1.0 if self.global_dice else 1.0 / self.world

    def all_reduce_sums(self, sums):
        if self.world > 1 or self.force:
            dist.all_reduce(sums, group=self.group)

    def broadcast_params(self, eng):
        if self.world > 1:
            dist.broadcast(eng.P, src=0, group=self.group)
            eng.refresh_weight_copies()

    def _reduce_range(self, eng, start, end):
        if (self.world <= 1 and not self.force) or end <= start:
            return
        g = eng.G[start:end]
        if g.is_cuda:
            if self._comm is None:
                self._comm = torch.cuda.Stream(device=g.device)
            # the bucket's gradients may come from several streams (the engine runs the conv weight gradients on their own stream)
            streams = [torch.cuda.current_stream()] + [st for st in getattr(eng, "grad_streams", lambda: [])() if st != torch.cuda.current_stream()]
            events = []
            for st in streams:
                ev = torch.cuda.Event()
                ev.record(st)
                events.append(ev)
            with torch.cuda.stream(self._comm):
                for ev in events:
                    self._comm.wait_event(ev)
                dist.all_reduce(g, group=self.group)
        else:
            dist.all_reduce(g, group=self.group)
        self.launched.append((start, end))

    def begin(self):
        self._start = 0
        self.launched = []

    def grad_ready(self, eng, name):
        """called by the engine right after the wgrad of `name` was enqueued (layout order == call order)"""
        L = eng.layout[name]
        last = L.get("beta", L["b"])                    # the layer's last parameter range (norm beta sits behind the bias)
        end = last[0] + last[1]
        if name == next(iter(eng.layout)):
            self.begin()
        if end - self._start >= self.bucket_elems:
            self._reduce_range(eng, self._start, end)
            self._start = end

    def finish(self, eng):
        self._reduce_range(eng, self._start, eng.n_flat)
        self._start = eng.n_flat
        if self._comm is not None:
            torch.cuda.current_stream().wait_stream(self._comm)
