"""Host input pipeline off the critical path.

The reference feeds every batch as lists of token *strings*: ``Dataset.batches`` buckets them
(dataset.py:467-579), ``pad_batch`` pads strings (vocabulary.py:331-354), the feed dict carries the
padded string matrix (model/sequence.py:202-222) and a TF hash table turns it into ids inside the
graph -- all of it on the host thread that also drives ``Session.run``, so the accelerator idles
while Python shuffles strings.  Here the same data take three steps away from the step loop:

1. ``preindex(dataset, feedables)``: every token series is mapped to int32 id arrays ONCE per
   dataset (epochs re-use them); batches are then assembled by slicing integers
   (``model.sequence.index_series`` takes id sequences as they are).
2. ``Prefetcher``: a worker thread walks the batch iterator ahead of the training loop (this is the
   reference's lazy bucketing buffer), builds the feed dicts of the next ``depth`` batches, copies
   them through pinned host memory to the device with asynchronous DMA on the session's copy stream
   (``Session.prefetch_scope``), and hands the batches over in order.  The compute stream waits on
   the copy's event only when the batch is actually used.
3. ``TensorFlowManager.execute`` finds the feed dict cached on the batch object and every array
   already resident in HBM (``Session.to_device`` is keyed by array identity): the step loop issues
   device-to-device stagings and kernels only.

``processors/bpe.py`` holds the BPE pre-/post-processors of the translation pipeline.
"""
import queue
import sys
import threading
from typing import Dict, Iterable, Iterator, List, Optional, Set

import numpy as np

from .dataset import Dataset
from .runtime import RunContext
from .vocabulary import UNK_TOKEN_INDEX, Vocabulary


def series_vocabularies(feedables: Iterable) -> Dict[str, Vocabulary]:
    """data series -> the vocabulary its readers index it with (series read through two different
    vocabularies stay strings)."""
    found: Dict[str, Vocabulary] = {}
    clash: Set[str] = set()
    for part in feedables:
        pairs = []
        if hasattr(part, "data_ids") and hasattr(part, "vocabularies"):          # EmbeddedFactorSequence
            pairs = list(zip(part.data_ids, part.vocabularies))
        elif hasattr(part, "data_id") and hasattr(part, "vocabulary"):           # autoregressive decoders
            pairs = [(part.data_id, part.vocabulary)]
        for sid, vocab in pairs:
            if sid in found and found[sid] is not vocab:
                clash.add(sid)
            found.setdefault(sid, vocab)
    return {sid: v for sid, v in found.items() if sid not in clash}


def preindex(dataset: Dataset, feedables: Iterable) -> Dataset:
    """A dataset whose token series are int32 id arrays (OOV -> <unk>, vocabulary.py:224-244)."""
    vocabs = series_vocabularies(feedables)
    series = {}
    for name in dataset.series:
        items = list(dataset.get_series(name))
        vocab = vocabs.get(name)
        if vocab is not None and items and isinstance(items[0], (list, tuple)) and \
                (not items[0] or isinstance(items[0][0], str)):
            w2i = vocab._word_to_index                                            # pylint: disable=protected-access
            items = [np.fromiter((w2i.get(tok, UNK_TOKEN_INDEX) for tok in sent), dtype=np.int32, count=len(sent))
                     for sent in items]
        series[name] = items
    return Dataset(dataset.name, series, dataset.batching, dataset.outputs, None, dataset.shuffled)


class Prefetcher:
    """Iterate ``batches`` with the next ``depth`` batches indexed and uploaded ahead of use.

        for batch in Prefetcher(tf_manager, feedables, train=True).iterate(dataset.batches()):
            tf_manager.execute(batch, feedables, [trainer], train=True)
    """

    def __init__(self, tf_manager, feedables: Set, train: bool = False, depth: int = 2) -> None:
        if depth < 1:
            raise ValueError("prefetch depth must be at least 1")
        self.tf_manager = tf_manager
        self.feedables = set(feedables)
        self.train = train
        self.depth = depth

    def upload(self, batch: Dataset) -> None:
        """Feed dict of ``batch`` (cached on the batch object) + asynchronous upload to every session."""
        from .tf_manager import _feed_dicts
        feed = _feed_dicts(batch, self.feedables, train=self.train)
        for sess in self.tf_manager.sessions:
            ctx = RunContext(sess, dict(feed))
            with sess.prefetch_scope():
                for part in self.feedables:
                    part.stage_inputs(ctx)

    def iterate(self, batches: Iterable[Dataset]) -> Iterator[Dataset]:
        ready: "queue.Queue" = queue.Queue(maxsize=self.depth)
        stop = threading.Event()
        done = object()

        def work() -> None:
            try:
                for batch in batches:
                    if stop.is_set():
                        return
                    self.upload(batch)
                    ready.put(batch)
                ready.put(done)
            except BaseException as exc:              # pylint: disable=broad-except
                ready.put(exc)                        # re-raised on the consumer's thread
        thread = threading.Thread(target=work, name="nm-prefetch", daemon=True)
        # The step thread launches kernels from Python; with CPython's default 5 ms switch interval a worker
        # in the middle of indexing would hold the interpreter that long and starve the launches.
        old_interval = sys.getswitchinterval()
        sys.setswitchinterval(min(old_interval, 2e-4))
        thread.start()
        try:
            while True:
                item = ready.get()
                if item is done:
                    return
                if isinstance(item, BaseException):
                    raise item
                yield item
        finally:
            sys.setswitchinterval(old_interval)
            stop.set()
            while thread.is_alive():                  # unblock a producer waiting on a full queue
                try:
                    ready.get_nowait()
                except queue.Empty:
                    thread.join(timeout=0.01)


def prefetched(tf_manager, batches: Iterable[Dataset], feedables: Set, train: bool = False,
               depth: int = 2) -> Iterator[Dataset]:
    return Prefetcher(tf_manager, feedables, train, depth).iterate(batches)
