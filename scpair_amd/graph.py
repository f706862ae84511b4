"""One training step captured in a HIP graph.

The reference's step (utils.py:118-148) is ~100 kernel launches at C2 shapes; issued one by one through autograd and
ctypes they cost the host as long as the GPU needs to run them.  `CapturedStep` records the step once
(torch.cuda.graph: stream capture of everything the step launches, the C-ABI kernels included -- they are launched on
torch's current stream) and replays it with three host calls per step: the copy of the minibatch's row ids into the
graph's input buffer, one tiny launch that rewrites the per-step device words (dropout seed offset, AdamW bias
corrections: ops.StepState), and hipGraphLaunch.

What makes the step capturable:
  * the minibatch is a device vector of row ids; tile lists are built on the GPU with fixed-capacity entry buffers
    (BatchTiles(..., capacity=csr.max_batch_nnz(B))), so no buffer size depends on the batch's contents;
  * nothing in the step reads device memory on the host;
  * by-value kernel arguments that change per step are replaced by device words (seed_dev / bc_dev arguments).
The arithmetic and the kernel order are those of the eager step, so with dropout off a replayed step is bitwise the
eager step.  Single-process training only (a gradient exchange between backward and the optimiser step stays eager).

EXPERIMENTAL, and not reachable from scPair_object or bench.py unless SCPAIR_EXPERIMENTAL_GRAPH=1 is set: bitwise-tested
at small shapes only.  The replay of the full-size (C2) step ended with an unresponsive GPU box twice in round 2 and
the cause was never isolated, so nothing here is offered as a supported path.
"""
from __future__ import annotations

import torch

from . import ops


class CapturedStep:
    """fn(rows) -> loss tensor or tuple of tensors; `rows` is the int64 device vector of the minibatch's row ids (always
    the same buffer, refilled before each replay).  `optimizers`: the FusedAdamW instances fn steps."""

    RUN_AHEAD = 8

    def __init__(self, fn, optimizers, batch, device, seed_stride=1_000_003):
        self._fence = None
        self.fn, self.opts = fn, list(optimizers)
        self.device = torch.device(device)
        self.rows = torch.zeros(int(batch), dtype=torch.int64, device=self.device)
        self.state = ops.StepState(self.device)
        self.graph = None
        self.out = None
        self.replays = 0
        self.seed_stride = int(seed_stride)
        betas = {tuple(g["betas"]) for o in self.opts for g in o.param_groups}
        if len(betas) > 1:
            raise RuntimeError("CapturedStep: the optimisers of one captured step must share their betas")

    def capture(self, rows):
        """Records fn on the rows given (they must be a valid minibatch: the recording pass does not execute, but the
        buffer keeps them for the first replay).  Run fn eagerly at least once before (lazy initialisation: optimiser
        state, bf16 weight shadows, kernel attributes must exist before a capture)."""
        if self.graph is not None:
            raise RuntimeError("CapturedStep: already captured")
        self.rows.copy_(rows)
        steps = {o.state[p]["step"] for o in self.opts for g in o.param_groups for p in g["params"] if len(o.state[p])}
        if len(steps) > 1:
            raise RuntimeError("CapturedStep: parameters at different step counts cannot share one captured step")
        torch.cuda.synchronize(self.device)
        g = torch.cuda.CUDAGraph()
        for o in self.opts:
            o.begin_capture(self.state)
        ops.set_step_state(self.state)
        try:
            with torch.cuda.graph(g):
                self.out = self.fn(self.rows)
            for o in self.opts:
                o.end_capture()
        except BaseException:
            for o in self.opts:
                o.release_capture()
            raise
        finally:
            ops.set_step_state(None)
        self.graph = g
        return self

    def __call__(self, rows):
        """One training step on the minibatch `rows` (int64 device tensor of the captured batch size).  Returns fn's
        outputs: static tensors that the next replay overwrites (clone what must be kept)."""
        if self.graph is None:
            raise RuntimeError("CapturedStep: capture() first")
        if rows.shape != self.rows.shape:
            raise ValueError(f"CapturedStep: captured for {tuple(self.rows.shape)} row ids, got {tuple(rows.shape)}")
        self.rows.copy_(rows, non_blocking=True)
        # every optimiser advances; the bias corrections come from those that own captured parameters (one that updated
        # nothing returns (1, 1)) and must agree: the replay's kernels all read the same two device words
        pairs = [o.dyn_advance() for o in self.opts]
        live = {(float(a), float(b)) for a, b in pairs if (a, b) != (1.0, 1.0)}
        if len(live) > 1:
            raise RuntimeError(f"CapturedStep: the optimisers of one captured step disagree on AdamW's bias corrections {sorted(live)}")
        bc1, bc2 = live.pop() if live else (1.0, 1.0)
        self.replays += 1
        self.state.set(self.replays * self.seed_stride, bc1, bc2)
        self.graph.replay()
        # bounded run-ahead: at most 2 * RUN_AHEAD replays are ever queued behind the GPU
        if self.replays % self.RUN_AHEAD == 0:
            if self._fence is not None:
                self._fence.synchronize()
            self._fence = torch.cuda.Event()
            self._fence.record()
        return self.out

    def release(self):
        """Drops the graph (its private memory pool is freed with it)."""
        for o in self.opts:
            o.release_capture()
        self.graph = None
        self.out = None
