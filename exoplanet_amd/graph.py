"""hipGraph replay of a leapfrog step.

A value+gradient step is a dozen short launches (record packing, scan, heavy, reduce,
packing VJP, a few tensor shuffles): issued eagerly it is bound by launch latency
(~0.3-0.4 ms per step), replayed as one hipGraph it costs ~50 us.  ``GraphedStep``
captures ``fn(*inputs)`` once -- inputs and outputs become static buffers -- and replays
it; new parameter values are copied into the static inputs before each replay.

Requirements (those of torch.cuda.graphs): every tensor ``fn`` touches lives on the GPU
already (Python scalars are fine: they become fill kernels), shapes never change, and
``fn`` does no host synchronisation (no ``.item()``, no data-dependent Python branches).
The exoplanet_amd ops satisfy this: they launch on the current stream and take their
scratch from torch's allocator.  One more, when ``fn`` differentiates with respect to the inputs:
no autograd graph built from the same leaves OUTSIDE the capture may still be alive (e.g. a
non-detached result of an eager call of ``fn``) -- the engine would synchronise with the stream that
graph was built on, which a capturing stream cannot do (on ROCm 7.2 that ends the capture with a crash,
not an error): detach what you keep.
"""
import torch

__all__ = ["GraphedStep"]


class GraphedStep:
    def __init__(self, fn, *inputs, warmup=3):
        """``fn(*inputs) -> tensor or tuple of tensors``; ``inputs`` are device tensors
        (leaves may require grad: ``fn`` can call torch.autograd.grad inside)."""
        if not inputs or not all(isinstance(x, torch.Tensor) and x.is_cuda for x in inputs):
            raise ValueError("GraphedStep needs device tensors as inputs")
        self.inputs = inputs
        dev = inputs[0].device
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                fn(*inputs)
        torch.cuda.current_stream(dev).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.outputs = fn(*inputs)

    def __call__(self, *new_values):
        """copy ``new_values`` (same shapes; omit to reuse the current contents) into the
        static inputs, replay, return the static outputs (valid until the next replay)"""
        if new_values:
            if len(new_values) != len(self.inputs):
                raise ValueError("expected one value per captured input")
            with torch.no_grad():
                for dst, src in zip(self.inputs, new_values):
                    if src is not dst:
                        dst.copy_(src)
        self.graph.replay()
        return self.outputs
