"""Producer-thread wrapper around an iterator (same contract as reference fetal_net/utils/threaded_generator.py:16-53):
one daemon-less thread fills a bounded queue, the consumer iterates until the sentinel."""
from queue import Queue
from threading import Thread


class ThreadedGenerator(object):
    def __init__(self, iterator, sentinel=object(), queue_maxsize=0, daemon=False, Thread=Thread, Queue=Queue):
        self._iterator = iterator
        self._sentinel = sentinel
        self._queue = Queue(maxsize=queue_maxsize)
        self._thread = Thread(name=repr(iterator), target=self._run)
        self._thread.daemon = daemon
        self._error = None

    def __repr__(self):
        return 'ThreadedGenerator({!r})'.format(self._iterator)

    def _run(self):
        try:
            for value in self._iterator:
                self._queue.put(value, block=True)
        except BaseException as e:  # surfaced on the consumer side instead of dying silently
            self._error = e
        finally:
            self._queue.put(self._sentinel)

    def __iter__(self):
        self._thread.start()
        for value in iter(self._queue.get, self._sentinel):
            yield value
        self._thread.join()
        if self._error is not None:
            raise self._error
