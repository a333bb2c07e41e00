"""Pre-computed feature maps as encoders (mirror of neuralmonkey/encoders/numpy_stateful_filler.py).

``SpatialFiller`` (:155-245) feeds [B,H,W,D] convolutional maps (the captioning configuration:
8x8x2048 ResNet maps, BASELINE configs[3]) to the same Bahdanau attention kernels as a sentence
encoder; the optional 1x1 convolutions are MFMA GEMMs over the B*H*W positions.  ``output`` is
the mean over positions (:209-212), computed as a batched [1,S]x[S,D] GEMM.
"""
from typing import Dict, List

import numpy as np
import torch

from .. import autodiff as F
from .. import ops
from ..model.model_part import FeedDict, InitializerSpecs, ModelPart
from ..model.stateful import SpatialStatefulWithOutput, Stateful
from ..runtime import Placeholder, tensor
from ..variables import glorot_uniform_initializer, zeros_initializer


class StatefulFiller(ModelPart, Stateful):
    """One pre-computed vector per example as an encoder (numpy_stateful_filler.py:16-72): ``output`` is the fed
    [B, dimension] array, or -- when ``output_shape`` names another size -- its ``tf.layers.dense`` projection
    (variables ``<name>/dense/kernel``, ``<name>/dense/bias``; no activation).  A decoder takes it like any other
    encoder's final state (initial-state projection, ``StatefulContext``)."""

    # pylint: disable=too-many-arguments
    def __init__(self, name: str, dimension: int, data_id: str, output_shape: int = None, reuse: ModelPart = None,
                 save_checkpoint: str = None, load_checkpoint: str = None,
                 initializers: InitializerSpecs = None) -> None:
        ModelPart.__init__(self, name, reuse, save_checkpoint, load_checkpoint, initializers)
        self.data_id = data_id
        self.dimension = dimension
        self.output_shape = output_shape
        if self.dimension <= 0:
            raise ValueError("Input vector dimension must be positive.")
        if self.output_shape is not None and self.output_shape <= 0:
            raise ValueError("Output vector dimension must be positive.")
        self.vector_input = Placeholder("{}/vector".format(name))

    @property
    def input_types(self) -> Dict[str, type]:
        return {self.data_id: np.float32}

    @property
    def input_shapes(self) -> Dict[str, List]:
        return {self.data_id: [None, self.dimension]}

    @property
    def projected(self) -> bool:
        return self.output_shape is not None and self.output_shape != self.dimension

    @property
    def output_size(self) -> int:
        return self.output_shape if self.projected else self.dimension

    def declare_variables(self, store) -> None:
        if self.projected:
            self.declare(store, "dense/kernel", (self.dimension, self.output_shape), glorot_uniform_initializer())
            self.declare(store, "dense/bias", (self.output_shape,), zeros_initializer())

    def feed_dict(self, dataset, train: bool = False) -> FeedDict:
        fd = ModelPart.feed_dict(self, dataset, train)
        fd[self.vector_input] = np.stack([np.asarray(x, np.float32) for x in dataset.get_series(self.data_id)])
        return fd

    @tensor
    def vector(self, ctx) -> torch.Tensor:
        """The fed vectors in a persistent device buffer."""
        fed = ctx.fed(self.vector_input)
        if tuple(fed.shape[1:]) != (self.dimension,):
            raise ValueError("StatefulFiller '{}': fed vectors of shape {}, expected {}"
                             .format(self.name, tuple(fed.shape[1:]), (self.dimension,)))
        return ctx.session.staged((id(self), "vector"), ctx.session.to_device(fed, torch.float32, "vector_input"))

    def stage_inputs(self, ctx) -> None:
        self.vector(ctx)

    def graph_safe_training(self, train_mode: bool) -> bool:
        return True

    @tensor
    def _activations(self, ctx):
        x = self.vector(ctx)
        train = bool(ctx.fed(self.train_mode))
        tape = F.Tape(ctx, (id(self), "filler"), recording=ctx.wants_backward(train) and self.projected)
        cur = tape.leaf(x)
        if self.projected:
            cur = F.linear(tape, cur, tape.param(self, "dense/kernel"), tape.param(self, "dense/bias"))
        return {"tape": tape, "out_var": cur}

    @tensor
    def output(self, ctx) -> torch.Tensor:
        return self._activations(ctx)["out_var"].data

    def backward(self, ctx, d_states, d_final) -> None:
        """dL/d(output) [B, output_size]; there are no temporal states."""
        act = self._activations(ctx)
        tape, var = act["tape"], act["out_var"]
        if not tape.recording or d_final is None:
            return                                   # the raw vectors: nothing trainable upstream
        ops.ew("copy", d_final, None, tape.grad(var), accumulate=True)
        tape.backward()


class SpatialFiller(ModelPart, SpatialStatefulWithOutput):
    # pylint: disable=too-many-arguments
    def __init__(self, name: str, input_shape: List[int], data_id: str, projection_dim: int = None,
                 ff_hidden_dim: int = None, reuse: ModelPart = None, save_checkpoint: str = None,
                 load_checkpoint: str = None, initializers: InitializerSpecs = None) -> None:
        ModelPart.__init__(self, name, reuse, save_checkpoint, load_checkpoint, initializers)
        self.data_id = data_id
        self.input_shape = list(input_shape)
        self.projection_dim = projection_dim
        self.ff_hidden_dim = ff_hidden_dim
        if self.ff_hidden_dim is not None and self.projection_dim is None:
            raise ValueError("projection_dim must be provided when using ff_hidden_dim")
        if len(self.input_shape) != 3:
            raise ValueError("The input shape should have 3 dimensions.")
        self.spatial_input = Placeholder("{}/spatial_input".format(name))

    @property
    def input_types(self) -> Dict[str, type]:
        return {self.data_id: np.float32}

    @property
    def input_shapes(self) -> Dict[str, List]:
        return {self.data_id: [None] + self.input_shape}

    @property
    def dimension(self) -> int:
        return self.projection_dim if self.projection_dim else self.input_shape[2]

    @property
    def output_size(self) -> int:
        return self.dimension

    def _layers(self):
        """[(variable scope, in, out, relu)] of the 1x1 convolutions (tf.layers.conv2d default names)."""
        layers, d = [], self.input_shape[2]
        if self.ff_hidden_dim:
            layers.append(("conv2d", d, self.ff_hidden_dim, True))
            d = self.ff_hidden_dim
        if self.projection_dim:
            layers.append(("conv2d_1" if layers else "conv2d", d, self.projection_dim, False))
        return layers

    def declare_variables(self, store) -> None:
        for scope, d_in, d_out, _ in self._layers():
            # tf.layers.conv2d kernel [1,1,in,out] stored as the [in,out] matrix it is
            self.declare(store, scope + "/kernel", (d_in, d_out), glorot_uniform_initializer())
            self.declare(store, scope + "/bias", (d_out,), zeros_initializer())

    def feed_dict(self, dataset, train: bool = False) -> FeedDict:
        fd = ModelPart.feed_dict(self, dataset, train)
        fd[self.spatial_input] = np.stack([np.asarray(x, np.float32) for x in dataset.get_series(self.data_id)])
        return fd

    @tensor
    def _maps(self, ctx) -> torch.Tensor:
        """The fed maps in a persistent device buffer."""
        maps = ctx.fed(self.spatial_input)
        if tuple(maps.shape[1:]) != tuple(self.input_shape):
            raise ValueError("SpatialFiller '{}': fed maps of shape {}, expected {}"
                             .format(self.name, tuple(maps.shape[1:]), tuple(self.input_shape)))
        return ctx.session.staged((id(self), "maps"), ctx.session.to_device(maps, torch.float32, "spatial_input"))

    def stage_inputs(self, ctx) -> None:
        self._maps(ctx)

    def graph_safe_training(self, train_mode: bool) -> bool:
        return True

    @tensor
    def _activations(self, ctx):
        x = self._maps(ctx)
        bsz, h, w, d = x.shape
        s = h * w
        train = bool(ctx.fed(self.train_mode))
        tape = F.Tape(ctx, (id(self), "filler"), recording=ctx.wants_backward(train) and bool(self._layers()))
        cur = tape.leaf(x.reshape(bsz * s, d))
        for scope, _, _, use_relu in self._layers():
            cur = F.linear(tape, cur, tape.param(self, scope + "/kernel"), tape.param(self, scope + "/bias"))
            if use_relu:
                cur = F.relu(tape, cur)
        dim = cur.shape[1]
        states = cur.data.view(bsz, h, w, dim)
        # average_image: mean over the S positions as ones[1,S]/S . states[b]
        avg_w = ctx.buffer((id(self), "avg_w", s), (1, s))
        ops.fill(avg_w, 1.0 / s)
        out = ctx.buffer((id(self), "output", bsz), (bsz, 1, dim))
        ops.gemm(avg_w.expand(bsz, 1, s), cur.data.view(bsz, s, dim), out=out)
        return {"tape": tape, "states_var": cur, "states": states, "output": out.view(bsz, dim), "avg_w": avg_w,
                "shape": (bsz, s, dim)}

    @tensor
    def spatial_states(self, ctx) -> torch.Tensor:
        return self._activations(ctx)["states"]

    @tensor
    def spatial_mask(self, ctx) -> torch.Tensor:
        st = self._activations(ctx)["states"]
        mask = ctx.buffer((id(self), "mask", tuple(st.shape[:3])), tuple(st.shape[:3]))
        ops.fill(mask, 1.0)
        return mask

    @tensor
    def output(self, ctx) -> torch.Tensor:
        return self._activations(ctx)["output"]

    def backward(self, ctx, d_states, d_final) -> None:
        """dL/d(spatial_states) [B,S,D] (from the attention) and dL/d(output) [B,D]."""
        act = self._activations(ctx)
        tape, var = act["tape"], act["states_var"]
        if not tape.recording:
            return                                   # raw maps: nothing trainable upstream
        bsz, s, dim = act["shape"]
        g = tape.grad(var).view(bsz, s, dim)
        if d_states is not None:
            ops.ew("copy", d_states.reshape(bsz * s, dim), None, g.view(bsz * s, dim), accumulate=True)
        if d_final is not None:                      # d mean: every position receives d_final / S
            ops.gemm(act["avg_w"].expand(bsz, 1, s), d_final.reshape(bsz, 1, dim), out=g, trans_a=True,
                     accumulate=True)
        tape.backward()
