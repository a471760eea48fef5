"""hipGraph capture of a whole training step.

At the reference's batch sizes a step is a chain of 10-20 short kernels and is launch/host-bound
(profiles/r1/triplet_kernel_stats.csv).  Every libesr_hip.so entry point only enqueues work on the
stream it is given and never allocates or synchronises, so the whole step -- gathers, fused loss kernels,
sort, segment-reduce, Adagrad -- can be captured once into a hipGraph and replayed with one launch.
``torch.cuda.CUDAGraph`` is the capture vehicle (it owns the capture stream and the private memory pool
the transient buffers come from); nothing is traced or compiled.

Only optimizers whose update does not depend on a host-side step counter can be replayed
(sparse Adagrad / SGD; dense Adam's bias correction changes every step).
"""
import torch


class GraphedStep:
    """``fn(*tensors) -> tensor or tuple of tensors`` captured once; call with new inputs of the same
    shapes to replay.  Inputs are copied into the captured static buffers; outputs are the captured
    output tensors (overwritten by the next replay)."""

    def __init__(self, fn, example_inputs, warmup=2):
        self.static_inputs = [t.clone() for t in example_inputs]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):          # sizes the workspace caches and the allocator pools
                fn(*self.static_inputs)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.outputs = fn(*self.static_inputs)

    def __call__(self, *inputs):
        for dst, src in zip(self.static_inputs, inputs):
            dst.copy_(src, non_blocking=True)
        self.graph.replay()
        return self.outputs
