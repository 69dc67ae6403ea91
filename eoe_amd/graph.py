"""HIP-graph replay of the step's device work for the launch-bound configurations (CNN32 at 32x32: ~60 short kernels
per step, 0.84 ms of GPU time behind 1.1-1.3 ms of host launch time).  Forward + loss + backward (+ anomaly scores) of
one FIXED-SHAPE step batch are captured once (`torch.cuda.CUDAGraph` = hipGraph on ROCm; every kernel of libeoe_hip.so is
launched on the capturing stream) and replayed per step; the optimiser step stays outside the graph because its bias
corrections are host scalars that change every step.  New relative to the reference (eager PyTorch, ad_trainer.py:406-455);
results are those of the eager step."""
import torch


class GraphedStep:
    """`loss, scores = graphed(imgs, lbls)` == eager `feats = model(imgs); loss = loss_fn(feats, lbls); loss.backward();
    scores = score_fn(feats)`; parameter gradients land in `p.grad` (static tensors owned by the graph's memory pool)."""

    def __init__(self, model, loss_fn, score_fn, imgs, lbls, warmup: int = 2):
        if not imgs.is_cuda:
            raise RuntimeError("GraphedStep needs GPU tensors")
        self.model, self.imgs, self.lbls = model, imgs.clone(), lbls.clone()
        params = [p for p in model.parameters() if p.requires_grad]
        # the warm-up passes must not count as training steps: keep the BatchNorm running statistics
        saved = [b.detach().clone() for b in model.buffers()]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                for p in params:
                    p.grad = None
                loss_fn(model(self.imgs), self.lbls).backward()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        with torch.no_grad():
            for b, s in zip(model.buffers(), saved):
                b.copy_(s)
        for p in params:
            p.grad = None
        self.graph = torch.cuda.CUDAGraph()
        # captured on the stream the warm-up ran on: ops.scratch buffers (keyed by stream) exist before the capture and are not
        # allocated inside the graph's private pool
        with torch.cuda.graph(self.graph, stream=side):
            feats = model(self.imgs)
            self.loss = loss_fn(feats, self.lbls)
            self.loss.backward()
            self.scores = score_fn(feats) if score_fn is not None else None
        self.grads = [p.grad for p in params]
        self.params = params

    def __call__(self, imgs, lbls):
        if imgs.shape != self.imgs.shape or lbls.shape != self.lbls.shape:
            raise ValueError("GraphedStep replays one fixed batch shape; run ragged batches eagerly")
        if imgs.data_ptr() != self.imgs.data_ptr():
            self.imgs.copy_(imgs)
        if lbls.data_ptr() != self.lbls.data_ptr():
            self.lbls.copy_(lbls)
        for p, g in zip(self.params, self.grads):          # an optimiser's zero_grad(set_to_none=True) drops them
            if p.grad is not g:
                p.grad = g
        self.graph.replay()
        return self.loss, self.scores
