"""Training-loop helpers of the MI355X build (not in the reference).

``GraphedTrainStep`` captures ONE iteration of the reference's diffusion training loop (R/main.py:243-252:
``abdiff.train_iter(x)['loss']`` -> ``optim.zero_grad()`` -> ``loss.backward()`` -> ``optim.step()`` ->
``functional.reset_net``) as a hipGraph and replays it per batch.  At the reference's batch of 32 token maps the iteration is
bound by launch overhead (about 150 launches, 4.3 ms of kernels in 5.5 ms of wall time); the replay removes it.  The random
draws (``sample_time``, ``q_sample``) are device-side and graph-safe: every replay draws fresh noise from torch's CUDA
generator.  The optimizer must be constructed with ``capturable=True``.

``GraphedVQVAETrainStep`` does the same for the VQ-VAE training loop (R/main.py:118-146: ``model(spike_input, images)`` ->
``loss_eq + loss_rec`` -> backward -> AdamW step -> ``reset_net``): about 100 launches and 1.4 ms of kernels per iteration at the
reference's batch of 32, which the host needs 3.5 ms to issue one by one."""
from __future__ import annotations

import torch

from .fused import invalidate_derived


class GraphedTrainStep:
    def __init__(self, abdiff, optimizer, example_batch: torch.Tensor, warmup: int = 3):
        from spikingjelly.activation_based import functional
        den = abdiff._denoise_fn
        if not den.training:
            raise RuntimeError('spkdiff: put the denoiser in train() mode before capturing a training step')
        if example_batch.device.type != 'cuda':
            raise RuntimeError('spkdiff: the training step runs on a ROCm device')
        for g in optimizer.param_groups:
            if not g.get('capturable', False):
                raise RuntimeError('spkdiff: construct the optimizer with capturable=True to capture its step')
        self.abdiff, self.optimizer, self.den = abdiff, optimizer, den
        self.static_x = example_batch.detach().clone()
        dev = example_batch.device

        def step():
            loss = abdiff.train_iter(self.static_x)['loss']
            loss.backward()
            optimizer.step()
            functional.reset_net(net=den)
            return loss

        # warm-up on a side stream (library algorithm selection, allocator pools, optimizer state), then capture
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                optimizer.zero_grad(set_to_none=True)
                step()
        torch.cuda.current_stream(dev).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        optimizer.zero_grad(set_to_none=True)          # gradients are allocated inside the capture (the graph's pool)
        with torch.cuda.graph(self.graph):
            self.loss = step()

    def __call__(self, batch: torch.Tensor) -> torch.Tensor:
        """One training iteration on ``batch`` (same shape as the example).  Returns the loss tensor (overwritten by the
        next call)."""
        self.static_x.copy_(batch)
        self.graph.replay()
        # the replayed optimizer step rewrote the weights without touching any tensor's _version: whatever was derived from
        # them (packed digit planes, folded BN terms, captured sampler graphs of this denoiser) is stale from here on
        invalidate_derived(self.den)
        return self.loss


class GraphedVQVAETrainStep:
    """One captured iteration of the VQ-VAE training loop (R/main.py:118-146).  ``model``: an ``SNN_VQVAE`` in train() mode with
    step mode 'm'; ``example_images`` [B, C, H, W] (the spike input is the image repeated over T, as the reference builds it,
    R/main.py:127-128).  ``__call__(images)`` returns (loss_eq, loss_rec, real_loss_rec) tensors (overwritten by the next call)."""

    def __init__(self, model, optimizer, example_images: torch.Tensor, T: int = 16, warmup: int = 3):
        from spikingjelly.activation_based import functional
        if not model.training:
            raise RuntimeError('spkdiff: put the model in train() mode before capturing a training step')
        if example_images.device.type != 'cuda':
            raise RuntimeError('spkdiff: the training step runs on a ROCm device')
        for g in optimizer.param_groups:
            if not g.get('capturable', False):
                raise RuntimeError('spkdiff: construct the optimizer with capturable=True to capture its step')
        self.model, self.optimizer = model, optimizer
        self.static_x = example_images.detach().clone()
        dev = example_images.device

        def step():
            spike = self.static_x.unsqueeze(0).repeat(T, 1, 1, 1, 1)
            loss_eq, loss_rec, real = model(spike, self.static_x)
            (loss_eq + loss_rec).backward()
            optimizer.step()
            functional.reset_net(model)
            return loss_eq, loss_rec, real

        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                optimizer.zero_grad(set_to_none=True)
                step()
        torch.cuda.current_stream(dev).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        optimizer.zero_grad(set_to_none=True)
        with torch.cuda.graph(self.graph):
            self.losses = step()

    def __call__(self, images: torch.Tensor):
        self.static_x.copy_(images)
        self.graph.replay()
        invalidate_derived(self.model)
        return self.losses
