"""Run an iterator on a producer thread and hand its items to the consuming thread through a bounded queue.

Same constructor and iteration contract as the class the reference's sliding-window loop wraps its tile batches in
(reference fetal_net/utils/threaded_generator.py:16-53, used at prediction.py:167): `ThreadedGenerator(iterator, queue_maxsize=N)` is
iterable once; iteration starts the thread, yields the items in order and joins the thread at exhaustion.  Unlike the reference, an
exception raised by the producer is re-raised in the consumer instead of being lost with the thread.
"""
import queue
import threading

_DONE = object()


class ThreadedGenerator(object):
    def __init__(self, iterator, sentinel=_DONE, queue_maxsize=0, daemon=False, Thread=threading.Thread, Queue=queue.Queue):
        self.source = iterator
        self.end_marker = sentinel
        self.buffer = Queue(maxsize=queue_maxsize)
        self.failure = None
        self.worker = Thread(target=self._produce, name="ThreadedGenerator:%r" % (iterator,))
        self.worker.daemon = daemon

    def _produce(self):
        it = iter(self.source)
        while True:
            try:
                item = next(it)
            except StopIteration:
                break
            except BaseException as exc:           # keep it for the consumer
                self.failure = exc
                break
            self.buffer.put(item)
        self.buffer.put(self.end_marker)

    def __iter__(self):
        self.worker.start()
        while True:
            item = self.buffer.get()
            if item is self.end_marker:
                break
            yield item
        self.worker.join()
        if self.failure is not None:
            raise self.failure

    def __repr__(self):
        return "ThreadedGenerator(%r)" % (self.source,)
