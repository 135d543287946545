"""Message -- the handle a user `reduce_func` receives in Graph.recv.  Mirrors pgl/message.py:19-173."""
from . import autograd as ag
from . import math as _math


class Message(object):
    """Message(msg, segment_ids): msg is the (lazily re-ordered) message dict, segment_ids the dense
    destination rank of each CSR-ordered edge (pgl/message.py:30-32).  `num_segments` (known from the
    graph's cache) spares the device->host read of segment_ids[-1] on every reduce."""

    def __init__(self, msg, segment_ids, num_segments=None):
        self._segment_ids = segment_ids
        self._msg = msg
        self._num_segments = num_segments

    def reduce(self, msg, pool_type="sum"):
        """pgl/message.py:34-53."""
        return _math.segment_pool(msg, self._segment_ids, pool_type=pool_type, num_segments=self._num_segments)

    def reduce_sum(self, msg):
        return _math.segment_sum(msg, self._segment_ids, num_segments=self._num_segments)

    def reduce_mean(self, msg):
        return _math.segment_mean(msg, self._segment_ids, num_segments=self._num_segments)

    def reduce_max(self, msg):
        return _math.segment_max(msg, self._segment_ids, num_segments=self._num_segments)

    def reduce_min(self, msg):
        return _math.segment_min(msg, self._segment_ids, num_segments=self._num_segments)

    def edge_expand(self, msg):
        """pgl/message.py:107-157: the inverse of reduce (gather by segment id)."""
        return ag.gather_rows(msg, self._segment_ids)

    def reduce_softmax(self, msg):
        """pgl/message.py:159-170."""
        return _math.segment_softmax(msg, self._segment_ids, num_segments=self._num_segments)

    def __getitem__(self, key):
        return self._msg[key]
