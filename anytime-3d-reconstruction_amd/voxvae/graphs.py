"""One evaluation step as a HIP graph.

Capturing the step once and replaying the graph removes the per-launch host work (ctypes call + argument marshalling): the
kernels, their arguments and the buffers they use are frozen in the graph, the caller writes the next batch into the graph's
input tensors.  MEASURED (MI355X): it buys nothing on this path -- at batch 4 a step takes 0.241 ms launched
call by call and 0.245 ms replayed, at batch 256 0.633 vs 0.647 ms: the launches are queued ahead of the GPU, and a step costs the
dependency latency of its launches, not host time.  bench.py times the eager path; this class exists for callers whose host thread
is busy elsewhere and as the capturability test of the step (no allocation, no host sync, no stream switch inside it).

    g = GraphedEvalStep(model, x, y, eps)      # captures model.eval_forward_device(x, y, eps) after a few warm-up calls
    pred, stats, metrics, kl = g(x_next, y_next, eps_next)     # copies the inputs in, replays, returns the graph's outputs

The outputs are the graph's own tensors: they are overwritten by the next replay (clone what must outlive it).  Everything the
step allocates during capture comes from the graph's private pool (torch's allocator), the engines' workspaces included, so
the model must not run eagerly between capture and the last replay on the same workspace -- use a model replica per graph.
"""
import torch


class GraphedEvalStep:
    def __init__(self, model, x, y, eps=None, warmup=3):
        if not (x.is_cuda and y.is_cuda):
            raise ValueError('GraphedEvalStep needs device tensors')
        self.model = model
        self.same = y is x or y.data_ptr() == x.data_ptr()
        self.x = x.clone()
        self.y = self.x if self.same else y.clone()
        self.eps = None if eps is None else eps.clone()
        side = torch.cuda.Stream(device=x.device)
        side.wait_stream(torch.cuda.current_stream(x.device))
        with torch.cuda.stream(side):
            for _ in range(warmup):                          # packs weights, sets kernel attributes, sizes the workspaces
                model.eval_forward_device(self.x, self.y, self.eps)
        torch.cuda.current_stream(x.device).wait_stream(side)
        torch.cuda.synchronize(x.device)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = model.eval_forward_device(self.x, self.y, self.eps)

    def __call__(self, x=None, y=None, eps=None):
        """Copy the given inputs into the graph's input tensors (None = keep what is there), replay, return the outputs."""
        if x is not None:
            self.x.copy_(x, non_blocking=True)
        if y is not None and self.same and not (y is x or (x is not None and y.data_ptr() == x.data_ptr())):
            raise ValueError('this graph was captured with the input as its own target (y is x): a different target needs a graph '
                             'captured with a separate y tensor')
        if y is not None and not self.same:
            self.y.copy_(y, non_blocking=True)
        if eps is not None and self.eps is not None:
            self.eps.copy_(eps, non_blocking=True)
        self.graph.replay()
        return self.out
