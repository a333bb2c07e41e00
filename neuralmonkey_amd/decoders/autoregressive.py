"""Autoregressive decoder base (mirror of neuralmonkey/decoders/autoregressive.py).

What the reference expresses as tf.while_loop over ``LoopState`` namedtuples
(autoregressive.py:425-562) runs here as an eager loop over pre-allocated
time-major history buffers; the fetchable surface (``train_loss``,
``runtime_loss``, ``runtime_logprobs``, ``decoded`` ...) is kept.
"""
from typing import List, Any, Dict, NamedTuple, Optional

import numpy as np
import torch

from .. import ops
from ..model.model_part import FeedDict, InitializerSpecs, ModelPart
from ..model.sequence import EmbeddedSequence, cached_index
from ..nn.dropout import dropout
from ..runtime import Placeholder, tensor
from ..variables import random_uniform_initializer, zeros_initializer
from ..vocabulary import (END_TOKEN_INDEX, PAD_TOKEN_INDEX, START_TOKEN_INDEX, UNK_TOKEN_INDEX,
                          Vocabulary, sentence_mask)


class LoopState(NamedTuple):
    """autoregressive.py:28-45."""
    histories: Any
    constants: Any
    feedables: Any


class DecoderHistories(NamedTuple):
    """autoregressive.py:48-79; time-major, rows < ``steps`` valid."""
    logits: Optional[torch.Tensor]        # [T,R,V] only when a runner asks for it
    output_states: torch.Tensor           # [T,R,E]
    output_symbols: torch.Tensor          # [T,R] int32
    output_mask: torch.Tensor             # [T,R] int32 (1 = not finished after the step)
    other: Any


class DecoderConstants(NamedTuple):
    train_inputs: Optional[torch.Tensor]


class DecoderFeedables(NamedTuple):
    """autoregressive.py:92-118."""
    step: int
    finished: torch.Tensor                # [R] int32
    embedded_input: torch.Tensor          # [R,E]
    other: Any


# pylint: disable=too-many-instance-attributes
class AutoregressiveDecoder(ModelPart):
    # pylint: disable=too-many-arguments
    def __init__(self, name: str, vocabulary: Vocabulary, data_id: str, max_output_len: int,
                 dropout_keep_prob: float = 1.0, embedding_size: int = None,
                 embeddings_source: EmbeddedSequence = None, tie_embeddings: bool = False,
                 label_smoothing: float = None, supress_unk: bool = False, reuse: ModelPart = None,
                 save_checkpoint: str = None, load_checkpoint: str = None,
                 initializers: InitializerSpecs = None) -> None:
        ModelPart.__init__(self, name, reuse, save_checkpoint, load_checkpoint, initializers)
        self.vocabulary = vocabulary
        self.data_id = data_id
        self.max_output_len = max_output_len
        self.dropout_keep_prob = dropout_keep_prob
        self._embedding_size = embedding_size
        self.embeddings_source = embeddings_source
        self.label_smoothing = label_smoothing
        self.tie_embeddings = tie_embeddings
        self.supress_unk = supress_unk
        self.encoder_states = lambda: []
        self.encoder_masks = lambda: []
        if self.max_output_len <= 0:
            raise ValueError("Maximum sequence length must be a positive integer.")
        if self._embedding_size is not None and self._embedding_size <= 0:
            raise ValueError("Embedding size must be a positive integer.")
        if self.dropout_keep_prob < 0.0 or self.dropout_keep_prob > 1.0:
            raise ValueError("Dropout keep probability must be a real number in the interval [0,1].")
        if label_smoothing is not None and not 0.0 <= label_smoothing < 1.0:
            raise ValueError("label_smoothing must be in [0, 1), was {}".format(label_smoothing))
        self.train_tokens = Placeholder("{}/{}".format(name, data_id))

    # -- static sizes ------------------------------------------------------------
    @property
    def embedding_size(self) -> int:
        if self.embeddings_source is None:
            if self._embedding_size is None:
                raise ValueError("You must specify either embedding size or the embedded sequence "
                                 "from which to reuse the embeddings (e.g. set 'embedding_size' or "
                                 "'embeddings_source' parameter)")
            return self._embedding_size
        return self.embeddings_source.embedding_sizes[0]

    @property
    def output_dimension(self) -> int:
        raise NotImplementedError("Abstract property")

    @property
    def input_types(self) -> Dict[str, type]:
        return {self.data_id: str}

    # -- variables -----------------------------------------------------------------
    def declare_variables(self, store) -> None:
        if self.embeddings_source is None:
            self.declare(store, "word_embeddings", (len(self.vocabulary), self.embedding_size))
        if self.tie_embeddings:
            if self.embedding_size != self.output_dimension:
                raise ValueError("`embedding_size must be equal to the output_projection size when "
                                 "using the `tie_embeddings` option")
        else:
            self.declare(store, "state_to_word_W", (self.output_dimension, len(self.vocabulary)),
                         random_uniform_initializer(-0.5, 0.5))
            self.declare(store, "state_to_word_b", (len(self.vocabulary),), zeros_initializer())

    def embedding_matrix(self, ctx) -> torch.Tensor:
        if self.embeddings_source is not None:
            return self.embeddings_source.embedding_matrix(ctx)
        return self.var(ctx, "word_embeddings")

    @property
    def embedding_matrix_name(self) -> str:
        if self.embeddings_source is not None:
            return self.embeddings_source.embedding_matrix_name
        return self.var_name("word_embeddings")

    def decoding_bias(self, ctx) -> Optional[torch.Tensor]:
        """state_to_word_b, with -1e9 on <unk> when ``supress_unk`` (autoregressive.py:450-459)."""
        if self.tie_embeddings:
            b = ctx.buffer((id(self), "zero_b"), (len(self.vocabulary),), zero=True)
        else:
            b = self.var(ctx, "state_to_word_b")
        if not self.supress_unk:
            return b
        key = (id(self), "unk_b")
        if key not in ctx.memo:
            beff = ctx.buffer(key, (len(self.vocabulary),))
            ops.copy(beff, b)
            beff[UNK_TOKEN_INDEX] -= 1e9
            ctx.memo[key] = beff
        return ctx.memo[key]

    def state_to_logits(self, ctx, state: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
        """logits = state . W + b as one MFMA GEMM."""
        bias = self.decoding_bias(ctx)
        # (the teacher-forced projection stays exact fp32 also under NM_PROJ_SPLIT: at 6400 rows the split kernel saves
        # 0.27 ms of 1.71 and re-splitting the updated weights costs 0.13 -- measured, 10.29 vs 10.32 ms per step)
        if self.tie_embeddings:
            return ops.gemm(state, self.embedding_matrix(ctx), out=out, bias=bias, trans_b=True)
        return ops.gemm(state, self.var(ctx, "state_to_word_W"), out=out, bias=bias)

    def logits_stats_ok(self, ctx, state: torch.Tensor) -> bool:
        """May the vocabulary projection run with its row statistics in the GEMM epilogue
        (``ops.logits_stats_gemm``: float4 operand loads, 16-byte aligned rows)?"""
        w = self.embedding_matrix(ctx) if self.tie_embeddings else self.var(ctx, "state_to_word_W")
        return (w.shape[0] % 4 == 0 and w.shape[1] % 4 == 0 and w.stride(0) % 4 == 0 and w.data_ptr() % 16 == 0
                and state.stride(0) % 4 == 0 and state.shape[1] % 4 == 0 and state.data_ptr() % 16 == 0)

    def state_to_logits_stats(self, ctx, state: torch.Tensor, stats: torch.Tensor,
                              out: Optional[torch.Tensor] = None) -> None:
        """``state_to_logits`` + per-tile {max, sum exp, argmax} of every row in ``stats``; the logits are
        written only when ``out`` is given (autoregressive.py:450-459 + :470 / beam_search_decoder.py:537-543)."""
        bias = self.decoding_bias(ctx)
        self.ensure_split_projection(ctx)
        if self.tie_embeddings:
            ops.logits_stats_gemm(state, self.embedding_matrix(ctx), bias, stats, out=out, trans_b=True)
        else:
            ops.logits_stats_gemm(state, self.var(ctx, "state_to_word_W"), bias, stats, out=out)

    def ensure_split_projection(self, ctx) -> bool:
        """NM_PROJ_SPLIT=1 (opt-in, inference): the projection's weights as three bf16 planes, re-split whenever the
        variables change (ops.proj_split_prepare); the statistics GEMM of the decoding steps then runs on the bf16
        matrix cores.  A no-op without the switch.  Returns whether the current weights are registered."""
        if not ops.PROJ_SPLIT or ctx.session.device.type != "cuda":
            return False
        w = self.embedding_matrix(ctx) if self.tie_embeddings else self.var(ctx, "state_to_word_W")
        sig = ctx.session.variables_signature()
        state = self.__dict__.setdefault("_split_proj", {})
        if state.get("sig") == sig:
            return True
        if torch.cuda.is_current_stream_capturing():
            # the planes on record belong to other weights and a split inside a capture would freeze today's: the
            # captured launches take the exact kernel
            ops.proj_split_forget(w)
            state["sig"] = None
            return False
        state["planes"] = ops.proj_split_prepare(w, trans_b=self.tie_embeddings, planes=state.get("planes"))
        state["sig"] = sig
        return state["planes"] is not None

    def embed_input_symbols(self, ctx, symbols: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
        emb = ops.embedding_gather(self.embedding_matrix(ctx), symbols, out=out)
        return dropout(ctx, emb, self.dropout_keep_prob, ctx.fed(self.train_mode))

    # -- fed data ---------------------------------------------------------------------
    def has_targets(self, ctx) -> bool:
        return ctx.is_fed(self.train_tokens)

    @tensor
    def train_inputs(self, ctx) -> torch.Tensor:
        """Target ids, time-major [T,B] (autoregressive.py:216-219)."""
        return ctx.session.staged((id(self), "tgt_tb"), ctx.session.to_device(
            ctx.fed(self.train_tokens), torch.int32, "tgt_tb", lambda ids: np.ascontiguousarray(ids.T)))

    @tensor
    def train_mask(self, ctx) -> torch.Tensor:
        return ctx.session.staged((id(self), "tgt_mask_tb"), ctx.session.to_device(
            ctx.fed(self.train_tokens), torch.float32, "tgt_mask_tb",
            lambda ids: np.ascontiguousarray(sentence_mask(ids).T)))

    def stage_inputs(self, ctx) -> None:
        """Evaluate every fetch that copies fed data to the device (see Feedable.stage_inputs)."""
        if self.has_targets(ctx):
            self.train_inputs(ctx)
            self.train_mask(ctx)

    def train_token_count(self, ctx) -> float:
        """Denominator of the training loss.  Plain: sum(train_mask) (autoregressive.py:312-316).
        With label smoothing the reference's loss function returns ONE scalar, the mean smoothed
        cross entropy over all B*T positions (tf.losses.softmax_cross_entropy reduces), which
        sequence_loss broadcasts against the mask (:292-310): sum(xents)/sum(mask) is then that mean
        itself -- over every position, padded ones included."""
        if self.label_smoothing:
            return float(np.asarray(ctx.fed(self.train_tokens)).size)
        return float(sentence_mask(ctx.fed(self.train_tokens)).sum())

    def xent_weights(self, mask):
        """Per-position weights of the training cross entropy (see ``train_token_count``)."""
        return None if self.label_smoothing else mask

    def feed_dict(self, dataset, train: bool = False) -> FeedDict:
        fd = ModelPart.feed_dict(self, dataset, train)
        sentences = dataset.maybe_get_series(self.data_id)
        if sentences is None and train:
            raise ValueError("When training, you must feed reference sentences")
        if sentences is not None:
            fd[self.train_tokens] = cached_index(dataset, self.data_id, self.vocabulary,
                                                 self.max_output_len, False, True)
        return fd

    # -- fetchable surface (the reference's @tensor names) --------------------------------
    @tensor
    def train_loop_result(self, ctx):
        return self.decoding_loop(ctx, train_mode=True)

    @tensor
    def runtime_loop_result(self, ctx):
        return self.decoding_loop(ctx, train_mode=False)

    @tensor
    def train_loss(self, ctx):
        raise NotImplementedError("Abstract")

    @tensor
    def runtime_loss(self, ctx):
        raise NotImplementedError("Abstract")

    @property
    def cost(self):
        return self.train_loss

    @tensor
    def train_xents(self, ctx) -> torch.Tensor:
        """[B,T] masked cross entropy of the teacher-forced pass (autoregressive.py:289-310; XentRunner)."""
        res = self.train_loop_result(ctx)
        rows, steps, bsz = res.saved["loss_rows"], res.saved["steps"], res.saved["bsz"]
        xents = rows.view(steps, bsz).t() if res.saved["loss_layout"] == "tb" else rows.view(bsz, steps)
        if self.label_smoothing:         # one scalar (mean over all positions) against the mask (:292-310)
            xents = (rows.sum() / rows.numel()) * self.train_mask(ctx).t()
        return xents

    @tensor
    def decoded(self, ctx) -> torch.Tensor:
        """[T,B] argmax(runtime_logits[:, :, 1:]) + 1: greedy symbols with <pad> excluded
        (autoregressive.py:341-349; PlainRunner)."""
        logits = self.runtime_logits(ctx)
        steps, bsz, vsz = logits.shape
        out = ctx.buffer((id(self), "decoded", steps, bsz), (steps * bsz,), torch.int32)
        ops.row_stats(logits.view(steps * bsz, vsz)[:, 1:], None, None, out)
        return (out + 1).view(steps, bsz)

    def decoding_loop(self, ctx, train_mode: bool, sample: bool = False, temperature: float = 1.0):
        raise NotImplementedError("Abstract method")

    def sampling_salts(self, ctx, steps: int) -> List[int]:
        """Salts of the ``steps`` draws of ONE sampling loop (autoregressive.py:470-473: tf.multinomial per step).
        TF seeds tf.multinomial from the graph / op seeds; here a draw is a counter-based hash of (salt, row,
        column) (``nm_gumbel_argmax``) and the salt folds in the session's seed, the decoder's name, how many
        sampling loops the session has run before this one and the step -- reproducible for a seeded session,
        different from loop to loop.  Kept in ``ctx.memo[(id(self), "sampling_salts")]`` for whoever checks."""
        sess = ctx.session
        draw = sess.__dict__.get("_sampling_loops", 0)
        sess.__dict__["_sampling_loops"] = draw + 1
        base = ctx.salt(self.name, "sample", getattr(sess, "seed", None))
        salts = [(base + draw * 0x9E3779B9 + t * 0x632BE5AB) & 0xFFFFFFFF for t in range(steps)]
        ctx.memo[(id(self), "sampling_salts")] = salts
        return salts

    @staticmethod
    def check_sampling_args(train_mode: bool, sample: bool, temperature: float) -> None:
        if temperature <= 0.0:
            raise ValueError("temperature must be positive, got {}".format(temperature))
        if train_mode and (sample or temperature != 1.0):
            # (autoregressive.py:466-473 lets `sample` override teacher forcing; the only caller in the reference,
            # trainers/rl_trainer.py:122-125 -- out of scope, SURVEY 8 -- passes train_mode=False)
            raise NotImplementedError("sampling / temperature in a teacher-forced loop are not implemented "
                                      "(the reference's only caller samples with train_mode=False)")
