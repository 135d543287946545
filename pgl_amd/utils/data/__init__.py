"""pgl.utils.data (pgl/utils/data/{dataset,dataloader}.py): Dataset / StreamDataset / Dataloader as the reference's
examples use them (examples/graphsage/cpu_sample_version/train.py:133-150: Dataloader(ds, batch_size, shuffle, num_workers,
collate_fn) iterated once per epoch).

Host-side glue in front of the hot path, kept deliberately small: batches of a map-style Dataset are index slices (shuffled
with numpy's global RNG, as the reference's Sampler does) passed through `collate_fn`; `num_workers > 1` prefetches batches on
a thread pool IN ORDER (the reference forks worker processes through paddle.reader and yields batches as they complete;
neighbour sampling here spends its time in numpy / the native library, which release the GIL).  The reference's Hadoop
reader and stream-shuffle buffer are out of scope."""
import warnings
from concurrent.futures import ThreadPoolExecutor

import numpy as np

__all__ = ["Dataset", "StreamDataset", "Dataloader"]


class Dataset(object):
    """pgl/utils/data/dataset.py:55-80: map-style dataset (subclass provides __len__ and __getitem__)."""

    def __len__(self):
        raise NotImplementedError

    def __getitem__(self, idx):
        raise NotImplementedError


class StreamDataset(object):
    """pgl/utils/data/dataset.py:83-110: iterable dataset of unknown length (subclass provides __iter__)."""

    def __iter__(self):
        raise NotImplementedError


class Dataloader(object):
    """pgl/utils/data/dataloader.py:30-146."""

    def __init__(self, dataset, batch_size=1, drop_last=False, shuffle=False, num_workers=1, collate_fn=None, buf_size=1000,
                 stream_shuffle_size=0):
        self.dataset, self.batch_size, self.drop_last, self.shuffle = dataset, int(batch_size), drop_last, shuffle
        self.num_workers, self.collate_fn, self.buf_size = int(num_workers), collate_fn, buf_size
        self.stream_shuffle_size = stream_shuffle_size
        if self.shuffle and isinstance(dataset, StreamDataset):
            warnings.warn("The argument [shuffle] should not be True with StreamDataset. It will be ignored.")
        if self.num_workers < 1:
            raise ValueError("num_workers(default: 1) should be larger than 0, but got [num_workers=%s] < 1." % self.num_workers)

    def __len__(self):
        if isinstance(self.dataset, StreamDataset):
            raise TypeError("StreamDataset has no length")
        n = len(self.dataset)
        return n // self.batch_size if self.drop_last else -(-n // self.batch_size)

    def _index_batches(self):
        n = len(self.dataset)
        perm = np.arange(n)
        if self.shuffle:
            np.random.shuffle(perm)
        for lo in range(0, n, self.batch_size):
            idx = perm[lo:lo + self.batch_size]
            if len(idx) < self.batch_size and self.drop_last:
                break
            yield idx

    def _make(self, idx):
        batch = [self.dataset[int(i)] for i in idx]
        return self.collate_fn(batch) if self.collate_fn is not None else batch

    def _stream_batches(self):
        batch = []
        for ex in self.dataset:
            batch.append(ex)
            if len(batch) == self.batch_size:
                yield self.collate_fn(batch) if self.collate_fn is not None else batch
                batch = []
        if batch and not self.drop_last:
            yield self.collate_fn(batch) if self.collate_fn is not None else batch

    def __iter__(self):
        if isinstance(self.dataset, StreamDataset):
            for b in self._stream_batches():
                yield b
            return
        if self.num_workers == 1:
            for idx in self._index_batches():
                yield self._make(idx)
            return
        with ThreadPoolExecutor(self.num_workers) as pool:
            pending = []
            for idx in self._index_batches():
                pending.append(pool.submit(self._make, idx))
                if len(pending) >= 2 * self.num_workers:
                    yield pending.pop(0).result()
            for f in pending:
                yield f.result()

    def __call__(self):
        return self.__iter__()


# the reference's submodule names (pgl/utils/data/dataloader.py, dataset.py: a few of its programs import from there) answer with this module
import sys as _sys                                                   # noqa: E402
dataloader = dataset = _sys.modules[__name__]
_sys.modules[__name__ + ".dataloader"] = _sys.modules[__name__ + ".dataset"] = dataloader

