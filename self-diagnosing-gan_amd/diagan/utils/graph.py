"""hipGraph capture of a launch-bound training step.

A global step of the small networks (MNIST-DCGAN: ~530 kernel launches of a few microseconds each) is bound by
the host's launch rate, not by the GPU (tools/host_time.py: 11.3 ms of launch loop per step, GPU idle in between).
`GraphedStep` records the step ONCE as a hipGraph (torch.cuda.CUDAGraph: stream capture of every launch this
library makes on torch's current stream -- the kernels do not allocate, synchronise or keep global state, so they
are capturable as they are) and replays it with one host call per step.

What is not a kernel argument baked into the graph is handled around the replay:
  * real batches: copied into the static input tensors the captured launches read;
  * Adam's step-dependent scalars and the scheduled learning rate: `FusedAdam` launches the device-hyper variant of
    its kernel during capture and writes the rows for the coming updates before every replay;
  * host-side counters the step advances (BatchNorm's `num_batches_tracked`, the networks' `param_version`): advanced
    by the amount one captured step advances them;
  * random numbers: torch's device generator is graph-safe (philox offsets are patched per replay), so `torch.randn`
    inside the step keeps drawing fresh values.
Single process only: under data parallelism the step contains RCCL collectives and runs eagerly."""
import torch

from diagan.models.layers import BatchNorm
from diagan.ops import conv as C
from diagan.trainer import distributed as dist


class GraphedStep:
    def __init__(self, fn, nets, optimizers, static_inputs=(), warmup=3, after=None):
        """fn(): one step reading its real batches from `static_inputs` (device tensors); nets / optimizers: every
        FlatNet and FusedAdam the step touches; after(): the host-only tail of a step (step counter, LR schedule),
        run after every warm-up step and every replay"""
        if dist.get_world_size() > 1:
            raise RuntimeError("GraphedStep is for single-process runs (collectives are not captured)")
        self.fn, self.nets, self.optimizers = fn, [n for n in nets if n is not None], [o for o in optimizers if o is not None]
        self.static_inputs = list(static_inputs)
        self.warmup, self.after = warmup, after
        self.graph = None
        self._bn_delta, self._ver_delta = {}, {}

    def _bn_modules(self):
        return [m for net in self.nets for m in net.modules() if isinstance(m, BatchNorm)]

    def capture(self):
        if C.TIMER is not None:
            raise RuntimeError("per-launch HIP-event timing cannot be captured; clear diagan.ops.conv.TIMER first")
        for _ in range(self.warmup):            # populates every lazily built table / scratch buffer / cache
            self.fn()
            if self.after is not None:
                self.after()
        torch.cuda.synchronize()
        bns = self._bn_modules()
        bn0 = [m._pending_batches for m in bns]
        ver0 = [n.param_version for n in self.nets]
        for o in self.optimizers:
            o.capture_begin()
        self.graph = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(self.graph):
                self.fn()
        finally:
            for o in self.optimizers:
                o.capture_end()
        # capture ran the host code of one step without any device work: take the counters it advanced as the
        # per-replay deltas and roll BatchNorm's back (param_version only ever needs to move forward)
        self._bn_delta = {m: m._pending_batches - b for m, b in zip(bns, bn0)}
        for m, b in zip(bns, bn0):
            m._pending_batches = b
        self._ver_delta = {n: max(n.param_version - v, 1) for n, v in zip(self.nets, ver0)}
        return self

    def set_inputs(self, tensors):
        for dst, src in zip(self.static_inputs, tensors):
            dst.copy_(src, non_blocking=True)

    def __call__(self, inputs=None):
        if self.graph is None:
            self.capture()
        if inputs is not None:
            self.set_inputs(inputs)
        for o in self.optimizers:
            o.before_replay()
        self.graph.replay()
        for m, d in self._bn_delta.items():
            m._pending_batches += d
        for n, d in self._ver_delta.items():
            n.param_version += d
        if self.after is not None:
            self.after()
