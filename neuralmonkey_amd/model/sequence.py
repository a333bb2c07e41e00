"""Embedded input sequences (mirror of neuralmonkey/model/sequence.py).

``EmbeddedSequence``: ids [B,S] -> embedding gather * mask (sequence.py:170-194)
as one HIP kernel (nm_embedding_gather)."""
from typing import Dict, List, Optional

import numpy as np
import torch

from .. import ops
from ..runtime import Placeholder, tensor
from ..vocabulary import Vocabulary, pad_batch, sentence_mask
from .model_part import FeedDict, InitializerSpecs, ModelPart
from .stateful import TemporalStateful


class Sequence(ModelPart, TemporalStateful):
    def __init__(self, name: str, max_length: int = None, reuse: ModelPart = None,
                 save_checkpoint: str = None, load_checkpoint: str = None,
                 initializers: InitializerSpecs = None) -> None:
        ModelPart.__init__(self, name, reuse, save_checkpoint, load_checkpoint, initializers)
        self.max_length = max_length
        if self.max_length is not None and self.max_length <= 0:
            raise ValueError("Max sequence length must be a positive integer.")


class EmbeddedFactorSequence(Sequence):
    # pylint: disable=too-many-arguments
    def __init__(self, name: str, vocabularies: List[Vocabulary], data_ids: List[str],
                 embedding_sizes: List[int], max_length: int = None, add_start_symbol: bool = False,
                 add_end_symbol: bool = False, scale_embeddings_by_depth: bool = False,
                 trainable: bool = True, embeddings_source: "EmbeddedFactorSequence" = None,
                 reuse: ModelPart = None, save_checkpoint: str = None, load_checkpoint: str = None,
                 initializers: InitializerSpecs = None) -> None:
        Sequence.__init__(self, name, max_length, reuse, save_checkpoint, load_checkpoint, initializers)
        self.vocabularies = vocabularies
        self.vocabulary_sizes = [len(v) for v in vocabularies]
        self.data_ids = data_ids
        self.embedding_sizes = embedding_sizes
        self.add_start_symbol = add_start_symbol
        self.add_end_symbol = add_end_symbol
        self.scale_embeddings_by_depth = scale_embeddings_by_depth
        self.embeddings_source = embeddings_source
        self.trainable = trainable
        if not len(data_ids) == len(vocabularies) == len(embedding_sizes):
            raise ValueError("data_ids, vocabularies, and embedding_sizes lists need to have the "
                             "same length")
        if any(esize <= 0 for esize in embedding_sizes):
            raise ValueError("Embedding size must be a positive integer.")
        if embeddings_source is not None:
            if not all(v1 == v2 for v1, v2 in zip(vocabularies, embeddings_source.vocabularies)):
                raise ValueError("When reusing embeedings, vocabularies must be the same.")
            if not all(s1 == s2 for s1, s2 in zip(embedding_sizes, embeddings_source.embedding_sizes)):
                raise ValueError("When reusing embeedings, embeddings sizes must be equal.")
        self.input_factors = [Placeholder("{}/{}".format(name, d)) for d in data_ids]

    @property
    def input_types(self) -> Dict[str, type]:
        return {d: str for d in self.data_ids}

    @property
    def dimension(self) -> int:
        return sum(self.embedding_sizes)

    def declare_variables(self, store) -> None:
        if self.embeddings_source is not None:
            return
        for i, (vsz, esz) in enumerate(zip(self.vocabulary_sizes, self.embedding_sizes)):
            self.declare(store, "embedding_matrix_{}".format(i), (vsz, esz), trainable=self.trainable)

    def embedding_matrices(self, ctx) -> List[torch.Tensor]:
        if self.embeddings_source is not None:
            return self.embeddings_source.embedding_matrices(ctx)
        return [self.var(ctx, "embedding_matrix_{}".format(i)) for i in range(len(self.data_ids))]

    def embedding_matrix_names(self) -> List[str]:
        if self.embeddings_source is not None:
            return self.embeddings_source.embedding_matrix_names()
        return [self.var_name("embedding_matrix_{}".format(i)) for i in range(len(self.data_ids))]

    def stage_inputs(self, ctx) -> None:
        self.input_factor_indices(ctx)
        self.temporal_mask(ctx)
        self.lengths(ctx)

    @tensor
    def input_factor_indices(self, ctx) -> List[torch.Tensor]:
        return [ctx.session.staged((id(self), "ids", i), ctx.session.to_device(ctx.fed(p), torch.int32))
                for i, p in enumerate(self.input_factors)]

    @tensor
    def temporal_mask(self, ctx) -> torch.Tensor:
        return ctx.session.staged((id(self), "mask"), ctx.session.to_device(
            ctx.fed(self.input_factors[0]), torch.float32, "mask", sentence_mask))

    @tensor
    def lengths(self, ctx) -> torch.Tensor:
        return ctx.session.staged((id(self), "len"), ctx.session.to_device(
            ctx.fed(self.input_factors[0]), torch.int32, "len",
            lambda ids: sentence_mask(ids).sum(1).astype(np.int32)))

    @tensor
    def temporal_states(self, ctx) -> torch.Tensor:
        """[B,S,sum(E)]: gathered rows (optionally * sqrt(E)) * mask (sequence.py:170-194)."""
        ids = self.input_factor_indices(ctx)
        mats = self.embedding_matrices(ctx)
        bsz, slen = ids[0].shape
        total = self.dimension
        out = ctx.buffer((id(self), "emb"), (bsz, slen, total))
        col = 0
        for idx, mat, esz in zip(ids, mats, self.embedding_sizes):
            scale = float(esz) ** 0.5 if self.scale_embeddings_by_depth else 1.0
            view = out.view(bsz * slen, total)[:, col:col + esz]
            # masking uses the FIRST factor's pad positions (sequence.py:191,196-199);
            # single-factor case == mask by own ids
            if len(ids) == 1:
                ops.embedding_gather(mat, idx.reshape(-1), out=view, mask_pad=True, scale=scale)
            else:
                ops.embedding_gather(mat, idx.reshape(-1), out=view, mask_pad=False, scale=scale)
            col += esz
        if len(ids) > 1:
            flat = out.view(bsz * slen, total)
            ops.ew("rowscale", flat, self.temporal_mask(ctx).reshape(-1, 1), flat)
        return out

    def backward(self, ctx, d_states: torch.Tensor) -> None:
        """dL/d(temporal_states) [B,S,sum(E)] -> embedding matrix gradients
        (rows of padded positions carry no gradient: the mask multiply)."""
        if not self.trainable:
            return
        ids = self.input_factor_indices(ctx)
        names = self.embedding_matrix_names()
        bsz, slen, total = d_states.shape
        d2 = d_states.view(bsz * slen, total)
        col = 0
        for f, (idx, name, esz) in enumerate(zip(ids, names, self.embedding_sizes)):
            d_part = d2[:, col:col + esz]
            if self.scale_embeddings_by_depth:          # forward multiplied the rows by sqrt(E) (sequence.py:185-187)
                scaled = ctx.buffer((id(self), "d_scaled", col), (bsz * slen, esz))
                d_part = ops.ew("scale", d_part, None, scaled, alpha=float(esz) ** 0.5)
            if f == 0:                                  # own pad positions == the mask
                ops.embedding_scatter_add(ctx.store.g(name), idx.reshape(-1), d_part, skip_pad=True)
                self._exchange_sparse(ctx, name)
            else:                                       # every factor is masked by the FIRST factor's padding
                masked = ctx.buffer((id(self), "d_masked", col), (bsz * slen, esz))
                ops.ew("rowscale", d_part, self.temporal_mask(ctx).reshape(-1, 1), masked)
                ops.embedding_scatter_add(ctx.store.g(name), idx.reshape(-1), masked, skip_pad=False)
            col += esz

    def _exchange_sparse(self, ctx, name: str) -> None:
        """Data parallelism, NM_DP_SPARSE_EMB=1: this matrix's gradient is final here and has at most B*S non-zero
        rows -- exchange (ids, rows) instead of the dense slice (distributed.DataParallel.exchange_sparse_rows).
        Only for a single-factor sequence whose matrix nobody else reads (no ``embeddings_source`` / ``reuse``
        sharing: further contributions would arrive after the exchange), and only when the trainer applies one
        update per batch (the ``dp_overlap`` marker GenericTrainer.train_op sets)."""
        from .. import distributed
        from ..runtime import registered_parts
        dp = distributed.current()
        if dp is None or not dp.sparse_embeddings or not ctx.memo.get("dp_overlap", False):
            return
        if len(self.data_ids) != 1 or self.embeddings_source is not None or self.shares_variables:
            return
        if any(getattr(part, "embeddings_source", None) is self for part in registered_parts()):
            return
        negate = lambda src, dst: ops.ew("scale", src, None, dst, alpha=-1.0)
        scatter = lambda table, ids, rows: ops.embedding_scatter_add(table, ids, rows, skip_pad=False)
        dp.exchange_sparse_rows(ctx.store, name, ctx.fed(self.input_factors[0]), ops.gather_rows, scatter, negate)

    def feed_dict(self, dataset, train: bool = False) -> FeedDict:
        fd = ModelPart.feed_dict(self, dataset, train)
        for plc, name, vocab in zip(self.input_factors, self.data_ids, self.vocabularies):
            fd[plc] = cached_index(dataset, name, vocab, self.max_length, self.add_start_symbol,
                                   self.add_end_symbol)
        return fd


def cached_index(dataset, name: str, vocab: Vocabulary, max_length: Optional[int],
                 add_start_symbol: bool, add_end_symbol: bool) -> np.ndarray:
    """index_series memoised on the batch object: a batch that is executed
    again hands back the *same* id array, which keeps it resident on the device
    (runtime.Session.to_device caches by identity)."""
    cache = dataset.__dict__.setdefault("_index_cache", {})
    key = (name, id(vocab), max_length, add_start_symbol, add_end_symbol)
    if key not in cache:
        cache[key] = index_series(list(dataset.get_series(name)), vocab, max_length, add_start_symbol,
                                  add_end_symbol)
    return cache[key]


def index_series(sentences, vocab: Vocabulary, max_length: Optional[int], add_start_symbol: bool,
                 add_end_symbol: bool) -> np.ndarray:
    """pad_batch + strings_to_indices on the host.  Sentences may already be
    int id sequences (pre-indexed data), which skips the dictionary lookup."""
    from ..vocabulary import END_TOKEN_INDEX, PAD_TOKEN_INDEX, START_TOKEN_INDEX
    indexed = False
    for sent in sentences:                        # id arrays, or the first non-empty sentence holds no strings
        if isinstance(sent, np.ndarray) or len(sent):
            indexed = isinstance(sent, np.ndarray) or not isinstance(sent[0], str)
            break
    if indexed:
        longest = max(len(s) for s in sentences) + (1 if add_end_symbol else 0)
        if max_length is not None:
            longest = min(max_length, longest)
        off = 1 if add_start_symbol else 0
        n = len(sentences)
        out = np.full((n, longest + off), PAD_TOKEN_INDEX, dtype=np.int32)
        # one scatter for the whole batch: (row, column) of every kept token of the concatenated sentences
        lens = np.fromiter((len(s) for s in sentences), dtype=np.int64, count=n)
        kept = np.minimum(lens, longest)
        if (kept < lens).any():
            sentences = [s[:k] for s, k in zip(sentences, kept)]
        total = int(kept.sum())
        if total:
            flat = np.concatenate([np.asarray(s, dtype=np.int32) for s in sentences if len(s)])
            starts = np.cumsum(kept) - kept
            rows = np.repeat(np.arange(n), kept)
            out[rows, np.arange(total) - np.repeat(starts, kept) + off] = flat
        if add_end_symbol:                        # </s> right after the sentence unless max_length cut it away
            fits = np.nonzero(lens < longest)[0]
            out[fits, lens[fits] + off] = END_TOKEN_INDEX
        if add_start_symbol:
            out[:, 0] = START_TOKEN_INDEX
        return out
    return vocab.strings_to_indices(pad_batch(sentences, max_length, add_start_symbol, add_end_symbol))


class EmbeddedSequence(EmbeddedFactorSequence):
    # pylint: disable=too-many-arguments
    def __init__(self, name: str, vocabulary: Vocabulary, data_id: str, embedding_size: int,
                 max_length: int = None, add_start_symbol: bool = False, add_end_symbol: bool = False,
                 scale_embeddings_by_depth: bool = False, trainable: bool = True,
                 embeddings_source: "EmbeddedSequence" = None, reuse: ModelPart = None,
                 save_checkpoint: str = None, load_checkpoint: str = None,
                 initializers: InitializerSpecs = None) -> None:
        EmbeddedFactorSequence.__init__(
            self, name=name, vocabularies=[vocabulary], data_ids=[data_id],
            embedding_sizes=[embedding_size], max_length=max_length,
            add_start_symbol=add_start_symbol, add_end_symbol=add_end_symbol,
            scale_embeddings_by_depth=scale_embeddings_by_depth, trainable=trainable,
            embeddings_source=embeddings_source, reuse=reuse, save_checkpoint=save_checkpoint,
            load_checkpoint=load_checkpoint, initializers=initializers)

    @property
    def inputs(self):
        return self.input_factors[0]

    def embedding_matrix(self, ctx) -> torch.Tensor:
        return self.embedding_matrices(ctx)[0]

    @property
    def embedding_matrix_name(self) -> str:
        return self.embedding_matrix_names()[0]

    @property
    def vocabulary(self) -> Vocabulary:
        return self.vocabularies[0]

    @property
    def data_id(self) -> str:
        return self.data_ids[0]
