"""HIP-graph replay of a fixed sequence of hot-path calls (forward AND backward).

The kernels of this package are launched through ctypes on PyTorch's current stream, so they can be captured
like any PyTorch op.  At the small stages of the training configuration (BASELINE config 5: SpaMatFunction
forward + backward at 60 x 108 and 180 x 324) a step is HOST-bound when issued eagerly -- ~0.12 ms of Python /
autograd / allocator per stage around 0.05 - 0.14 ms of kernels -- so the step is captured once and replayed:

    step = decnet_amd.graphs.GraphedStep(lambda: [spamat(L, R, rm, tm, D).backward(g) for ...], grads_of=[L, R, ...])
    step()            # one hipGraphLaunch; the .grad tensors of ``grads_of`` hold the result

The surface of SpaMatFunction (functions/SpaMat.py:8-50) is unchanged: the capture runs the very same
autograd.Function; only who issues the launches changes.  Inputs are read from the tensors that existed at capture
time (update them in place); the ``.grad`` tensors are allocated from the graph's private pool during capture and
rewritten by every replay.
"""
import torch


class GraphedStep:
    def __init__(self, fn, grads_of=(), warmup=3):
        """fn: a callable without arguments that runs forward + backward; grads_of: the leaf tensors whose .grad
        the step produces (reset to None before the capture, as whole-step capture requires)."""
        self.grads_of = list(grads_of)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                    # warm-up off the capture stream (allocator, lazy init)
            for _ in range(warmup):
                for t in self.grads_of:
                    t.grad = None
                fn()
        torch.cuda.current_stream().wait_stream(side)
        for t in self.grads_of:
            t.grad = None
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.result = fn()

    def __call__(self):
        self.graph.replay()
        return self.result
