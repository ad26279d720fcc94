"""``snn_model.vq_diffusion`` of the MI355X build: spiking denoiser + absorbing-state reverse diffusion sampler.

Counterpart of R/snn_model/vq_diffusion.py -- ``DummyModel`` (:150-208), ``AbsorbingDiffusion`` (:43-147, only
``sample`` is on the hot path) and ``get_data_for_diff`` (:23-36) -- with the same names, constructor arguments,
attributes (``num_embeddings``, ``n_samples``, ``mask_id``, ``shape``...) and ``state_dict`` keys.

The denoiser runs as fused Conv+BN+LIF kernels (conv1 sees a time-invariant input, conv6 + the time mean are one
kernel), the per-step token update is one ``spk_psample_step`` launch, and ``sample()`` enqueues the whole
reverse process without any host synchronisation.  Latent size and T are parameters (the reference hard-codes
7x7 and 16: vq_diffusion.py:47-48,106,198,206).
"""
import math

import torch
import torch.nn as nn

from spikingjelly.activation_based import neuron, functional, layer, surrogate, monitor  # noqa: F401
from spikingjelly import visualizing  # noqa: F401

from spkdiff import ops
from spkdiff.fused import FusedSequential, invalidate_derived, has_hooks, derived_epoch, derived_refs
from spkdiff.ops import IN_PTC, IN_TINV

from .vae_model import *  # noqa: F401,F403  (R/snn_model/vq_diffusion.py:21)


def get_data_for_diff(train_loader, model, T: int = 16, carry_state: bool = True):
    """Encode a data set to code indices (R/snn_model/vq_diffusion.py:23-36).

    The reference calls ``model(images_spike, images)`` batch after batch with no ``reset_net`` inside the loop, so every
    LIF layer starts a batch from the membrane potentials the previous batch left (its loaders drop the ragged last
    batch, R/load_dataset_snn.py:65-66).  ``carry_state=True`` (default) is that call sequence on the fused module path
    -- same indices as the reference (fixture F12), same module state afterwards; a batch of another size raises, as it
    does there.  ``carry_state=False`` encodes every batch from the reset state with the encoder alone (time-invariant
    input folded into the first kernel; nothing of the module state is read or written)."""
    print('prepare data for train diffusion...')
    model.eval()
    train_indices = []
    dev = next(model.parameters()).device
    for images, labels in train_loader:
        images = (images - 0.5).to(dev).float().contiguous()  # normalize to [-0.5, 0.5]
        with torch.inference_mode():
            if carry_state:
                images_spike = images.unsqueeze(0).repeat(T, 1, 1, 1, 1)
                _, _, encoding_indices = model(images_spike, images)
                L = images.shape[-1] // 4
                idx = encoding_indices.reshape(images.shape[0], L, L)
            else:
                idx = model.encode_images(images, T)
            train_indices.append(idx.cpu())
    return train_indices


class Sampler(nn.Module):
    def __init__(self):
        super().__init__()


class AbsorbingDiffusion(Sampler):
    def __init__(self, denoise_fn, mask_id, latent_shape=(7, 7)):
        super().__init__()
        self.num_classes = denoise_fn.num_embeddings
        self.shape = list(latent_shape)
        self.num_timesteps = latent_shape[0] * latent_shape[1]
        self.mask_id = mask_id
        self._denoise_fn = denoise_fn
        self.n_samples = 16
        self.mask_schedule = 'random'
        self.loss_type = 'reweighted_elbo'
        # 'philox': on-device counter-based noise (throughput); 'host': u and q drawn per step from torch's global
        # CPU generator in the reference's order (rand_like, then multinomial's exponential_), which reproduces the
        # reference CPU path token for token under the same torch.manual_seed.
        self.noise_source = 'philox'
        # Philox contract ('philox' mode): every sample() call takes ONE 62-bit draw from torch's global CPU generator as
        # its key, so ``torch.manual_seed(s); sample(); sample()`` gives two different batches and re-seeding repeats
        # them -- the reference's behaviour -- and two samplers in one process never share a stream.
        # ``noise_layout`` says how the counters are laid out (csrc/psample_common.h; include/spkdiff.h, spk_psample_step):
        #   'global' (default): counter = step * 2^40 + (GLOBAL image index * h*w + position) * K + class.  The draws of
        #       image i at step s depend on (key, s, i, position, class) only -- not on the batch size, not on how the batch
        #       is split over processes: an 8-GPU job, a 1-GPU job and the oracle on the dumped noise give the same tokens for
        #       the same images (SURVEY.md §8e "parity mode => result independent of G"; the reference draws one batch from
        #       one stream, R/snn_model/vq_diffusion.py:103-142).  A shard sets ``global_first`` (index of its first image,
        #       see ``set_shard``); ranks must use the SAME key: seed them alike, or let ``sync_key`` broadcast rank 0's draw.
        #   'rank': the rounds 1-3 form -- local image index, step stride b*h*w*K, the RANK folded into the key
        #       (``philox_stream``): ranks seeded alike draw distinct noise, but the sample depends on the split.
        import os
        self.noise_layout = 'global'
        self.global_first = 0
        # The key broadcast is a COLLECTIVE, so it is opt-in: it happens only in a sampler that ``set_shard`` declared a shard of a
        # multi-rank job (every rank of the job then calls sample()).  A sampler that never called set_shard inside an initialised
        # process group (a preview on rank 0 during DDP training, or old-style per-rank sampling) takes no collective and folds the
        # rank into its key, as rounds 1-3 did: no deadlock, and ranks seeded alike still draw distinct images (one warning).
        self.sync_key = True                 # set_shard + 'global' layout + world_size > 1: broadcast the key from rank 0
        self._shard_set = False
        self._warned_unsharded = False
        self.last_key = None
        self.philox_stream = int(os.environ.get('RANK', '0'))
        # Replay the whole reverse process as ONE hipGraph (philox mode, no hooks): the ~800 kernel launches of a
        # 100-step sample are captured once per (batch, steps, temp) and replayed; fresh noise per replay comes from a
        # 2-word device buffer {seed, counter base} the kernels read (spk_psample_step philox_state).
        self.use_graph = True
        self._graphs = {}
        # Reverse step t only writes the positions in `changes` (computed before the denoiser call, :113-124,140): an
        # image without a change at step t never has its denoiser output read.  True = evaluate the denoiser only for the
        # images spk_select_active lists for the step (61 % of (image, step) pairs drop out at 100 steps x 49 positions);
        # the sampled tokens are those of the dense loop, draw for draw.
        self.skip_untouched = True
        # ... and, of a touched image, the logits are read only at the positions that change: with 3x3 layers below them a
        # layer r levels down is needed within distance r of a change.  True = the MFMA layers of the denoiser compute the
        # positions spk_select_needed lists for the step (7x7 latents; again the same tokens, draw for draw).
        self.list_positions = True
        # ... from this batch size on: below it every launch of a reverse step is latency bound and the list bookkeeping (one more
        # launch per step, per-class item division) costs more than the skipped positions save -- R/main.py's own n_samples = 16:
        # 7.04 ms per 49-step sample without lists, 7.75 with; B = 32: 8.9 / 8.25 (tools/small_batch_time.py, profiles/r6_small_batch.txt)
        self.list_min_batch = 24
        # elimination forms: conv6 on the spike counts + the token update of the ACTIVE images as one launch per slot (spk_den_step_tail with
        # the active list) instead of two (spk_den_conv3x3_counts_mfma, spk_psample_step).  Same tokens -- and measured SLOWER (round 6, one
        # box: B = 256 x 100 steps 35.30 against 34.88 ms, B = 64 18.70 / 17.58, B = 16 x 49 steps 7.93 / 7.13: a workgroup of the step tail
        # is one image's 90-iteration weight stream, ~30 us whatever the number of slots, where the counts kernel spreads the active images'
        # rows over the chip).  Off; kept as an opt-in with its test (profiles/r6_ab_kernel_variants.txt (6)).
        self.step_tail_in_elimination = False
        self.list_radii = 3                 # layers below the logits that take lists (1: conv5 only ... 4: conv2..conv5;
                                            # conv2 needs nearly every position anyway: 3 measured fastest)
        # Derived weight forms (digit planes, folded BN terms, captured graphs) are keyed on (data_ptr, _version), which
        # writes through ``.data`` and graph-replayed optimizer steps do not change.  True = every sample() call compares
        # a content checksum of the denoiser's parameters and buffers (one launch + one 8-byte read-back, ~30 us) with the
        # one the derived forms were built from and rebuilds them when it differs.
        self.verify_weights = True
        self._wsum = None

    # ---- training step (R/snn_model/vq_diffusion.py:56-101,144-147) -------------------------
    def sample_time(self, b, device):
        t = torch.randint(1, self.num_timesteps + 1, (b,), device=device).long()
        pt = torch.ones_like(t).float() / self.num_timesteps
        return t, pt

    def q_sample(self, x_0, t):
        """Mask each token of x_0 [B,1,h,w] with probability t/T.  Returns (x_t, x_0_ignore, mask): masked positions
        hold ``mask_id`` in x_t, unmasked positions hold -1 (the loss's ignore index) in x_0_ignore."""
        b = x_0.shape[0]
        if x_0.is_cuda and x_0.dtype == torch.float32 and t.is_cuda and t.dtype == torch.int64 and x_0.dim() == 4:
            # one native launch after the framework's draw (same RNG call, same order as the reference's rand_like)
            return ops.q_sample(x_0, t, torch.rand_like(x_0), self.num_timesteps, self.mask_id)
        t_mask = t.reshape(b, 1, 1, 1).expand(b, 1, x_0.shape[2], x_0.shape[3])
        mask = torch.rand_like(x_0.float()) < (t_mask.float() / self.num_timesteps)
        x_t = torch.where(mask, torch.full_like(x_0, self.mask_id), x_0)
        x_0_ignore = torch.where(mask, x_0, torch.full_like(x_0, -1))
        return x_t, x_0_ignore, mask

    def _loss_from_logits(self, x_0_hat_logits, x_0_ignore, t):
        """Loss tail of _train_loss (:85-101): masked cross-entropy summed over positions, weighted per sample, in bits
        per latent dimension, mean over the batch.  Cross-entropy and its gradient: one spk_masked_ce launch."""
        b = x_0_hat_logits.shape[0]
        denom = math.log(2) * x_0_ignore.shape[1:].numel()
        if self.loss_type == 'elbo':
            pt = torch.ones_like(t).float() / self.num_timesteps
            coef = 1.0 / t.float() / pt / denom
        elif self.loss_type == 'reweighted_elbo':
            coef = (1 - (t / self.num_timesteps)).float() / denom
        else:
            raise ValueError
        return ops.MaskedCEFunction.apply(x_0_hat_logits, x_0_ignore.float(), coef / b)

    def _train_loss(self, x_0):
        b, device = x_0.size(0), x_0.device
        t, pt = self.sample_time(b, device)
        x_t, x_0_ignore, mask = self.q_sample(x_0=x_0, t=t)
        x_0_hat_logits = self._denoise_fn(x_t, t=t)
        return self._loss_from_logits(x_0_hat_logits, x_0_ignore, t)

    def train_iter(self, x):
        loss = self._train_loss(x)
        stats = {'loss': loss}
        return stats

    @torch.no_grad()
    def sample(self, temp=1.0, sample_steps=None, noise=None, record=None):
        """Reverse absorbing diffusion (R/snn_model/vq_diffusion.py:103-142).  Returns x_t int64 [B,1,h,w].

        ``noise``: optional callable t -> (u [B,1,h,w], q [B*h*w, K]) of device tensors (tests inject fixtures).
        ``record``: optional list receiving (t, x_t.clone(), unmasked.clone(), logits.clone()) per step."""
        dn = self._denoise_fn
        dev = next(dn.parameters()).device
        if dev.type != 'cuda':
            raise RuntimeError('spkdiff: the sampler runs on a ROCm device; move the denoiser with .cuda()')
        b = int(self.n_samples)
        h, w = self.shape
        K = self.num_classes
        if sample_steps is None:
            sample_steps = self.num_timesteps
        seed, base = 0, 0
        if noise is None and self.noise_source == 'philox':
            seed = self._philox_key()
            self.last_key = seed               # (read-only record: bench.py compares it across ranks after a timed region)
        self._check_weights(dn)
        if self.use_graph and noise is None and record is None and self.noise_source == 'philox':
            self._capturing = False
            try:
                return self._sample_graphed(dev, b, h, w, K, float(temp), int(sample_steps), seed, base)
            except (NotImplementedError, ValueError, TypeError):
                raise                          # an argument / support error of a kernel, not a capture problem
            except RuntimeError as e:
                # Only a failure raised while the capture block was open is a capture problem (the runtime refused an
                # operation on a capturing stream, another thread touched the device, ...): same kernels, launched one by
                # one.  The kernels' own return codes surface as ValueError / NotImplementedError / SpkdiffError with the
                # entry point's name and propagate.
                from spkdiff._lib import SpkdiffError
                if not self._capturing or isinstance(e, SpkdiffError):
                    raise
                import warnings
                warnings.warn(f'spkdiff: hipGraph capture of the sampler failed ({e}); launching eagerly')
                self.use_graph = False
                self._graphs.clear()
                torch.cuda.synchronize(dev)
            finally:
                self._capturing = False
        x_t = torch.full((b, 1, h, w), int(self.mask_id), dtype=torch.int64, device=dev)
        unmasked = torch.zeros((b, 1, h, w), dtype=torch.bool, device=dev)
        skip = self._skip_ok(h, w) and record is None
        act = None
        need = ops.NeedLists(b, int(self.list_radii), dev) if skip and self._list_ok(h, w, b) else None
        tail = (not skip) and dn.tail_fusable(h, w)                      # dense loop: the fused step tail (same tokens)
        tail_act = skip and self.step_tail_in_elimination and dn.tail_fusable(h, w)
        pre1 = None
        for t in reversed(range(1, sample_steps + 1)):
            u = q = None
            if noise is not None:
                u, q = noise(t)
            elif self.noise_source == 'host':
                u = torch.rand(b, 1, h, w).to(dev)                       # rand_like(x_t.float()), drawn first (:116)
            off = base + self._step_offset(sample_steps - t, b, h, w, K)
            if tail:
                if noise is None and self.noise_source == 'host':
                    q = torch.empty(b * h * w, K).exponential_(1).to(dev)    # (the denoiser call draws nothing: same order)
                pre1, logits = dn.sample_step(x_t, unmasked, t, temp, u, q, seed, off, pre1=pre1, want_next=t > 1,
                                              want_logits=record is not None)
                if record is not None:
                    record.append((t, x_t.clone(), unmasked.clone(), logits))
                continue
            if skip:
                act = ops.select_active(unmasked, t, u, seed, off, out=act, K=K)
                if need is not None:
                    ops.select_needed(unmasked, t, act, need, u, seed, off, K=K)
            with ops.active_set(*(act if skip else (None, None)), need=need):
                if skip and tail_act:
                    # elimination forms, round 6: conv6 on the counts + the token update as ONE launch per active slot (the step tail
                    # without a next-step first layer: that belongs to the next step's active set)
                    if noise is None and self.noise_source == 'host':
                        q = torch.empty(b * h * w, K).exponential_(1).to(dev)
                    _, logits = dn.sample_step(x_t, unmasked, t, temp, u, q, seed, off, want_next=False,
                                               want_logits=record is not None)
                else:
                    logits = dn.logits_from_tokens(x_t, t)                   # denoiser + reset_net (:128-129)
                    if noise is None and self.noise_source == 'host':
                        q = torch.empty(b * h * w, K).exponential_(1).to(dev)    # multinomial's one-draw fast path (:138)
                    ops.psample_step(logits, x_t, unmasked, t, temp, u, q, seed, off)
            if record is not None:
                record.append((t, x_t.clone(), unmasked.clone(), logits.clone()))
        return x_t

    def _skip_ok(self, h, w):
        return bool(self.skip_untouched)            # every kernel family takes the device-side image count

    def _list_ok(self, h, w, b=None):
        return bool(self.list_positions) and (h, w) == (7, 7) and (b is None or b >= int(self.list_min_batch))

    def form_for(self, b, h, w, sample_steps=None):
        """Name of the launch form ``sample()`` takes for a batch of ``b`` on an h x w latent (same tokens in every form)."""
        if not self._skip_ok(h, w):
            return 'dense_step_tail' if self._denoise_fn.tail_fusable(h, w) else 'dense'
        return 'elimination_lists' if self._list_ok(h, w, b) else 'elimination'

    STEP_STRIDE = 1 << 40        # 'global' layout: counters of one reverse step (images * h*w * K of them must fit)

    def set_shard(self, first: int, count: int = None):
        """This sampler generates images [first, first + count) of a larger job ('global' noise layout): the draws of an
        image are those the whole job would make for it.  ``count`` (optional) also sets ``n_samples``."""
        self.global_first = int(first)
        self._shard_set = True
        if count is not None:
            self.n_samples = int(count)
        return self

    def _step_offset(self, step_index, b, h, w, K):
        """Philox counter offset of reverse step number ``step_index`` (0 = the first step taken) for this sampler's shard."""
        if self.noise_layout == 'global':
            if (int(self.global_first) + b) * h * w * K > self.STEP_STRIDE:
                raise ValueError('spkdiff: global noise layout holds 2^40 counters per reverse step')
            return step_index * self.STEP_STRIDE + int(self.global_first) * h * w * K
        if self.noise_layout != 'rank':
            raise ValueError("noise_layout must be 'global' or 'rank'")
        return step_index * (b * h * w * K)

    def _philox_key(self):
        draw = int(torch.randint(0, 1 << 62, (1,), dtype=torch.int64))
        if self.noise_layout == 'global':
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                if not self._shard_set:
                    if not self._warned_unsharded:
                        import warnings
                        warnings.warn("spkdiff: sample() inside a process group without set_shard(): no key broadcast, the rank "
                                      "is folded into the key (per-rank images).  Call set_shard(first, count) on every rank "
                                      "(spkdiff.dist.sample_images_sharded(..., sampler=ab) does) for one split-independent job.")
                        self._warned_unsharded = True
                    return (draw ^ ((int(dist.get_rank()) * 0x9E3779B97F4A7C15) & 0x7FFFFFFFFFFFFFFF)) & 0x7FFFFFFFFFFFFFFF
                if self.sync_key:
                    dev = next(self._denoise_fn.parameters()).device if dist.get_backend() == 'nccl' else 'cpu'
                    k = torch.tensor([draw], dtype=torch.int64, device=dev)
                    dist.broadcast(k, 0)
                    draw = int(k.item())
            return draw & 0x7FFFFFFFFFFFFFFF
        return (draw ^ ((int(self.philox_stream) * 0x9E3779B97F4A7C15) & 0x7FFFFFFFFFFFFFFF)) & 0x7FFFFFFFFFFFFFFF

    def invalidate(self):
        """Drop captured graphs and every derived weight form of the denoiser (see spkdiff.fused.invalidate_derived)."""
        self._graphs.clear()
        invalidate_derived(self._denoise_fn)

    def _check_weights(self, dn):
        """Content checksum of the denoiser's floating-point tensors against the one seen by the previous call: a change
        that left every (data_ptr, _version) pair alone -- ``p.data.copy_(...)``, an optimizer step replayed from a graph
        -- drops the derived forms and the captured graphs here."""
        if not self.verify_weights:
            return
        ts = [t for t in list(dn.parameters()) + list(dn.buffers()) if t.is_floating_point() and t.is_cuda]
        ws = self._wsum
        if ws is None or ws[0].key != ops.TensorChecksum.key_of(ts):
            # (another set of tensors -- e.g. the training path re-laid the weights out channels-last: a new address -- is a
            #  change by itself: whatever was derived from the old ones is dropped)
            if ws is not None:
                self.invalidate()
            ws = self._wsum = [ops.TensorChecksum(ts), None]
        v = ws[0].value()
        if ws[1] is not None and ws[1] != v:
            self.invalidate()
        ws[1] = v


def _weights_key(module):
    return tuple((p.data_ptr(), p._version) for p in list(module.parameters()) + list(module.buffers())) + derived_epoch(module)


def _sample_graphed(self, dev, b, h, w, K, temp, sample_steps, seed, base):
    """Capture-once / replay-many form of the loop in ``sample``; same kernels, same results as the eager loop for the
    same (seed, counter base)."""
    dn = self._denoise_fn
    skip = self._skip_ok(h, w)
    lists = skip and self._list_ok(h, w, b)
    key = (str(dev), b, h, w, K, temp, sample_steps, int(self.mask_id), skip, lists, int(self.list_radii),
           bool(dn.use_step_tail), bool(self.step_tail_in_elimination), self.noise_layout, int(self.global_first), _weights_key(dn))
    entry = self._graphs.get(key)
    if entry is None:
        if len(self._graphs) >= 2:                              # at most two live graphs per sampler (e.g. dense and
            self._graphs.clear()                                #  elimination forms): their buffers are not small
        state = torch.zeros(2, dtype=torch.int64, device=dev)
        x_t = torch.empty((b, 1, h, w), dtype=torch.int64, device=dev)
        unmasked = torch.empty((b, 1, h, w), dtype=torch.bool, device=dev)

        act = (torch.zeros(b, dtype=torch.int32, device=dev), torch.zeros(2, dtype=torch.int32, device=dev)) if skip else None
        need = ops.NeedLists(b, int(self.list_radii), dev) if lists else None

        # dense form: the fused step tail where the architecture fits (conv6 + token update + the next step's first layer in
        # one launch), else every spk_psample_step also writes the next step's denoiser input (one launch less per step)
        tail = (not skip) and dn.tail_fusable(h, w)
        tail_act = skip and self.step_tail_in_elimination and dn.tail_fusable(h, w)
        inp = None if (skip or tail) else torch.empty((b, 2, h, w), dtype=torch.float32, device=dev)

        def body():
            x_t.fill_(int(self.mask_id))
            unmasked.zero_()
            pre1 = None
            for t in reversed(range(1, sample_steps + 1)):
                off = self._step_offset(sample_steps - t, b, h, w, K)
                if tail:
                    pre1, _ = dn.sample_step(x_t, unmasked, t, temp, None, None, 0, off, philox_state=state, pre1=pre1,
                                             want_next=t > 1)
                    continue
                if skip:
                    ops.select_active(unmasked, t, None, 0, off, philox_state=state, out=act, K=K)
                    if lists:
                        ops.select_needed(unmasked, t, act, need, None, 0, off, philox_state=state, K=K)
                elif t == sample_steps:
                    ops.den_build_input(x_t, t, out=inp)
                with ops.active_set(*(act if skip else (None, None)), need=need):
                    if tail_act:
                        dn.sample_step(x_t, unmasked, t, temp, None, None, 0, off, philox_state=state, want_next=False)
                    else:
                        logits = dn.logits_from_tokens(x_t, t, inp=inp)
                        ops.psample_step(logits, x_t, unmasked, t, temp, None, None, 0, off, philox_state=state,
                                         next_input=inp if t > 1 else None)

        # warm-up on a side stream (weight packing, BN terms, allocator pools, this graph's own flag workspaces), then capture
        flag_ws = {}
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side), ops.flag_scope(flag_ws):
            dn.logits_from_tokens(torch.full((b, 1, h, w), int(self.mask_id), dtype=torch.int64, device=dev), 1)
        torch.cuda.current_stream(dev).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        self._capturing = True
        with torch.cuda.graph(graph, capture_error_mode="thread_local"), ops.flag_scope(flag_ws):
            body()
        self._capturing = False
        # every buffer the captured launches address by raw pointer lives as long as the graph: a tensor freed here would
        # hand its block to the next allocation (another sampler's state, say) while replays keep writing to it.  That
        # includes the denoiser's derived tensors (packed weights, BN terms: an invalidation re-keys the graph, and until the
        # stale entry is evicted its memory must not be recycled) and the flag workspaces of the certified kernels.
        entry = (graph, state, x_t, (need, inp, unmasked, act, flag_ws, derived_refs(dn)))
        self._graphs[key] = entry
    graph, state, x_t = entry[:3]
    state.copy_(torch.tensor([seed, base], dtype=torch.int64), non_blocking=False)
    graph.replay()
    return x_t.clone()


AbsorbingDiffusion._sample_graphed = _sample_graphed


class DummyModel(nn.Module):
    """Spiking convolutional denoiser ("SDID"); R/snn_model/vq_diffusion.py:150-208."""

    def __init__(self, n_channel: int, num_embeddings, n_steps: int = 16) -> None:
        super(DummyModel, self).__init__()
        self.num_embeddings = num_embeddings
        self.n_steps = n_steps

        def block(cin, cout):
            return FusedSequential(
                layer.Conv2d(in_channels=cin, out_channels=cout, kernel_size=3, stride=1, padding=1),
                layer.BatchNorm2d(cout),
                neuron.LIFNode(surrogate_function=surrogate.ATan()))

        self.conv1 = block(n_channel * 2, 64)
        self.conv2 = block(64, 128)
        self.conv3 = block(128, 256)
        self.conv4 = block(256, 512)
        self.conv5 = block(512, 256)
        self.conv6 = FusedSequential(
            layer.Conv2d(256 + 64, num_embeddings, 3, 1, 1),
        )

    def _fused_ok(self):
        return all(c._fusable(c._blocks()) for c in (self.conv1, self.conv2, self.conv3, self.conv4, self.conv5,
                                                     self.conv6)) and self.n_steps <= ops.MAX_T

    # 'mfma-fp6x6': conv2..conv5 on the block-scaled fp6/fp4 MFMA (six exact radix-32 digit planes), conv6 time-collapsed
    # on int8 spike counts; 'mfma-i8x4': conv2..conv6 on the int8 MFMA (four exact base-256 digit planes); 'direct-f64':
    # fp64-accumulating direct kernels.  All return correctly rounded pre-activations of (fixed-point) exact dot products.
    # Request: 'auto' (fastest supported), 'fp6', 'i8', 'direct'.
    conv_impl_request = 'auto'
    # conv6 + mean over T evaluated as ONE convolution of the per-neuron spike counts (exact linearity; the sum over T
    # is rounded once instead of T times: logits agree with the per-step form to ~1 ulp).  False = per-step form.
    collapse_conv6 = True
    # training: conv6 + mean over T as ONE autograd operator (ops.SpikeConvMeanTrainFunction): same forward operations, the
    # backward convolves the output gradient once with the spike counts / once for all steps.  False: two operators.
    collapse_conv6_backward = True

    _latent_hw = (7, 7)        # latent size of the last call (7x7 MNIST-shaped, 8x8 CIFAR-shaped)

    def _impl_base(self, h, w):
        """Kernel family used for conv2..conv6 on an h x w latent.  conv2..conv5 never see ``num_embeddings``; the fp6 families'
        logits layer runs on the spike counts with its output channels zero-padded to a multiple of 16 inside the packed weights
        (round 6), so EVERY --codebook_size the reference accepts (R/main.py:58) stays on the matrix cores.  Only the int8 family's
        per-step logits kernel (``collapse_conv6 = False`` or request 'i8') still needs a multiple of 16 and otherwise leaves the
        call to the fp64 direct kernels."""
        req = self.conv_impl_request
        if req == 'direct':
            return 'direct-f64'
        if (req in ('auto', 'fp6') and self.collapse_conv6 and
                ops.den_fp6_supported(128, 64, 3, 1, 1, self.n_steps, h, w)):
            return 'mfma-fp6x6'
        if self.conv6[0].out_channels % 16 != 0:
            return 'direct-f64'
        return 'mfma-i8x4' if ops.den_mfma_supported(128, 64, 3, 1, 1, self.n_steps, h, w) else 'direct-f64'

    # the sampler's calls (fresh LIF state, nothing written back) take the second-generation fp6 kernel where it applies
    # (7x7 latents): the same spikes from the four leading digits + certified decisions + exact recomputation of the few neurons
    # near the threshold (csrc/den_mfma_fp6v2.hip).  False = always the first-generation kernel.
    use_fp6v2 = True
    _last_stateful = False

    def impl_for(self, h, w, stateful=False):
        """Kernel family used for conv2..conv5 on an h x w latent (see _impl_base); stateless calls on 7x7 latents refine
        'mfma-fp6x6' to 'mfma-fp6v2'."""
        base = self._impl_base(h, w)
        if (base == 'mfma-fp6x6' and self.use_fp6v2 and not stateful and
                ops.den_fp6v2_supported(128, 64, 3, 1, 1, self.n_steps, h, w)):
            return 'mfma-fp6v2'
        return base

    @property
    def conv_impl(self):
        return self.impl_for(*self._latent_hw, stateful=self._last_stateful)

    def _trunk(self, inp_b2hw, stateful, record=None, pre1=None):
        """conv1 .. conv5 on the input map [B,2,h,w] (``pre1 = (spikes, counts)``: the first layer's output is already there --
        the previous reverse step's tail launch produced it).  Returns (x5, cnt5, x1, cnt1, which, impl, collapse)."""
        T = self.n_steps
        # spikes travel channel-chunked: CPTC (32 u8 channels per chunk) for the int8 kernel, C4 (64 fp4 nibbles per
        # chunk) for the fp6 kernel -- the layout each stages per K chunk
        hw = (int(inp_b2hw.shape[-2]), int(inp_b2hw.shape[-1])) if pre1 is None else (int(pre1[0].shape[2]), int(pre1[0].shape[3]))
        self._latent_hw = hw
        self._last_stateful = bool(stateful)
        which = self.conv_impl
        impl = 'direct' if which == 'direct-f64' else 'auto'
        collapse = which != 'direct-f64' and self.collapse_conv6
        chunk = ops.CHUNK_S32 if which == 'mfma-fp6v2' else (ops.CHUNK_C4 if which == 'mfma-fp6x6' else 32)
        if pre1 is None:
            with ops.timed('den.conv1'):
                r1 = self.conv1.run(inp_b2hw, IN_TINV, final='ptc', T=T, stateful=stateful, chunk_out=chunk,
                                    want_counts=collapse)
            x1, cnt1 = r1['ptc'], r1['cnt']
        else:
            x1, cnt1 = pre1
        x = x1
        outs = [x1]
        cnt5 = None
        # (inside a position-list scope of the sampler: conv5 feeds the 3x3 logits convolution -> radius 1, conv4 radius 2, ...)
        for name, blk, radius in (('den.conv2', self.conv2, 4), ('den.conv3', self.conv3, 3), ('den.conv4', self.conv4, 2),
                                  ('den.conv5', self.conv5, 1)):
            with ops.timed(name):
                r = blk.run(x, IN_PTC, final='ptc', stateful=stateful, chunk_out=chunk, impl=impl,
                            want_counts=collapse and blk is self.conv5,
                            need_radius=radius if which == 'mfma-fp6v2' and collapse else None)
            x, cnt5 = r['ptc'], r['cnt']
            outs.append(x)
        if record is not None:
            record.extend(outs)
        return x, cnt5, x1, cnt1, which, impl, collapse

    def _conv6_params(self):
        conv = self.conv6[0]
        if not hasattr(conv, '_spk_params'):
            from spkdiff.fused import ConvParams
            object.__setattr__(conv, '_spk_params', ConvParams())
        return conv, conv._spk_params.get_i8(conv, pad_cout=True)

    def _run(self, inp_b2hw, stateful, record=None):
        T = self.n_steps
        x, cnt5, x1, cnt1, which, impl, collapse = self._trunk(inp_b2hw, stateful, record)
        with ops.timed('den.conv6'):
            if collapse and cnt5 is not None and cnt1 is not None:
                # conv6 is linear and followed by the mean over T: convolve the spike COUNTS once instead of T frames
                conv, packed = self._conv6_params()
                return ops.den_conv3x3_counts(cnt5, packed, conv.out_channels, T, cnt1=cnt1)
            return self.conv6.run(x, IN_PTC, final='mean', in1=x1, impl=impl)['f32']

    # The dense sampler's step: the tail of a reverse step -- conv6 on the spike counts, the token update and the NEXT step's
    # first layer -- is ONE launch per image (csrc/step_tail.hip) instead of three launches and a logits round trip through
    # memory.  False = conv6, spk_psample_step and conv1 as separate launches (same tokens).
    use_step_tail = True

    def tail_fusable(self, h, w):
        """Can ``sample_step`` take the fused tail launch on an h x w latent?  (the reference's architecture on the certified
        fp6 kernel family: up to 512 classes (any --codebook_size, R/main.py:58), 256 + 64 channels into conv6, T = 16, 7x7 or 8x8)"""
        return (self.use_step_tail and self._fused_ok() and not self.training and not has_hooks(self) and self.collapse_conv6
                and self.n_steps == 16 and (h, w) in ((7, 7), (8, 8)) and 1 <= self.conv6[0].out_channels <= ops.STEP_TAIL_MAX_K
                and self.conv6[0].in_channels == 320 and self.conv5[0].out_channels == 256 and self.conv1[0].out_channels == 64
                and self.conv1[0].in_channels == 2 and self.impl_for(h, w, stateful=False) == 'mfma-fp6v2')

    @torch.no_grad()
    def sample_step(self, x_t, unmasked, t, temp, u=None, q=None, seed=0, offset=0, philox_state=None, pre1=None,
                    want_next=True, want_logits=False):
        """One DENSE reverse step of the sampler on this denoiser (R/snn_model/vq_diffusion.py:113-140 with the call of
        :128-129 inside): x_t / unmasked are updated in place from the logits of ``self(x_t, t)`` (fresh LIF state, nothing
        written back).  ``pre1``: the first layer's (spikes, counts) for this step as returned by the previous call;
        ``want_next``: also evaluate it for step t - 1.  Returns (pre1 for the next step or None, logits or None)."""
        functional.reset_net(self)
        inp = None if pre1 is not None else ops.den_build_input(x_t, int(t))
        x, cnt5, x1, cnt1, which, impl, collapse = self._trunk(inp, False, pre1=pre1)
        conv6, packed6 = self._conv6_params()
        conv1, bn1 = self.conv1[0], self.conv1[1]
        nxt = None
        if want_next:
            a1, b1 = bn1.affine_terms()
            nxt = (conv1._spk_params.get(conv1), None if conv1.bias is None else conv1.bias.detach(), a1, b1)
        with ops.timed('den.tail'):
            return ops.den_step_tail(cnt5, cnt1, packed6, x_t, unmasked, int(t), float(temp), T=self.n_steps,
                                     K=conv6.out_channels, u=u, q=q, seed=seed, offset=offset, philox_state=philox_state,
                                     conv1=nxt, want_logits=want_logits)

    def _run_train(self, x, t):
        """train() mode (R/snn_model/vq_diffusion.py:189-208 with batch-statistics BN and surrogate-gradient LIF): the
        spike-input convolutions run the exact MFMA forward and the native backward (``ops.SpikeConvTrainFunction``; conv6 +
        the time mean: ``ops.SpikeConvMeanTrainFunction``), conv1 the library operator, each block tail is the native fused
        BN+LIF operator (``FusedSequential.train_forward``).  Membrane state follows the module semantics (kept until
        reset_net)."""
        T = self.n_steps
        inp = ops.den_build_input(x.detach(), t)                      # [B,2,h,w]: token ids and step as floats
        h = inp.unsqueeze(0).repeat(T, 1, 1, 1, 1)
        nbt = []                                                      # the five BatchNorm step counters: ONE launch at the end
        for blk in (self.conv1, self.conv2, self.conv3, self.conv4, self.conv5):
            object.__setattr__(blk, '_nbt_sink', nbt)
        preps = self._train_weight_prep(x)
        try:
            # (want_c4: the block tail also leaves its spikes in the packed format the next layer's exact forward reads)
            x1 = (self.conv1.train_forward(h, want_c4=preps is not None)
                  if preps is not None and self.conv1._trainable_fused(self.conv1._blocks(), h) else self.conv1(h))
            x5 = x1
            for i, blk in enumerate((self.conv2, self.conv3, self.conv4, self.conv5)):
                # conv2..conv5 see spikes: exact MFMA forward and native backward where the shape fits
                x5 = (blk.train_forward(x5, binary_input=True, prep=None if preps is None else preps[i],
                                        want_c4=preps is not None and i < 3)
                      if blk._trainable_fused(blk._blocks(), x5) else blk(x5))
        finally:
            for blk in (self.conv1, self.conv2, self.conv3, self.conv4, self.conv5):
                object.__setattr__(blk, '_nbt_sink', None)
        if nbt:
            torch._foreach_add_(nbt, 1)
        # (the fused block tails hand over channels-last spikes: the operator concatenates the 4-D views so that the layout
        #  survives, and its backward hands the two halves of the gradient on as views)
        cat = ops.CatChannelsFunction.apply(x5, x1)
        c6 = self.conv6
        blocks6 = c6._blocks()
        if self.collapse_conv6_backward and c6._trainable_fused(blocks6, cat) and c6.exact_conv_fits(blocks6, cat):
            # conv6 + the time mean as one operator whose backward runs on the spike counts (1/T of the per-step backward)
            conv = blocks6[0][0]
            return ops.SpikeConvMeanTrainFunction.apply(cat, conv.weight, conv.bias, None if preps is None else preps[4])
        x6 = c6.train_forward(cat, binary_input=True) if c6._trainable_fused(blocks6, cat) else c6(cat)
        return torch.sum(x6, dim=0) / T

    # False: every layer packs its own weights inside its forward / backward (20 small launches per iteration more)
    train_weight_prep = True

    def _train_weight_prep(self, x):
        """The weights of conv2..conv6 in the formats this iteration's forward (fp6 digit planes) and backward (two fp16 terms)
        read them in, made by two launches for all five layers (ops.train_weight_prep) -- or None when the model is not in the
        shape that path takes (single Conv-BN-LIF blocks on 7x7 / 8x8 maps, T = 16, channels-last parameters)."""
        if not self.train_weight_prep or self.n_steps != 16 or not x.is_cuda:
            return None
        T, B, H, W = self.n_steps, int(x.shape[0]), int(x.shape[2]), int(x.shape[3])
        layers = []
        for i, blk in enumerate((self.conv2, self.conv3, self.conv4, self.conv5, self.conv6)):
            blocks = blk._blocks()
            if blocks is None or len(blocks) != 1 or not blk.exact_train_forward:
                return None
            conv = blocks[0][0]
            w = conv.weight
            if w.dim() == 4 and not w.is_contiguous(memory_format=torch.channels_last):
                w.data = w.data.contiguous(memory_format=torch.channels_last)
            if not (conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1) and conv.groups == 1
                    and tuple(conv.dilation) == (1, 1) and conv.padding_mode == 'zeros' and w.requires_grad and conv.training
                    and ops.den_fp6_supported(conv.out_channels, conv.in_channels, 3, 1, 1, T, H, W)):
                return None
            last = i == 4
            if last and not self.collapse_conv6_backward:
                return None
            n_dg = B if last else (T * B if ops.conv3x3_dgrad_supported(conv.out_channels, conv.in_channels, H, W, T * B) else 0)
            if last and not (conv.out_channels % 16 == 0 and conv.in_channels % 32 == 0 and (H, W) in ((7, 7), (8, 8))):
                n_dg = 0
            layers.append((w, conv.bias, n_dg, (H, W)))
        return ops.train_weight_prep(layers)

    def invalidate(self):
        """Rebuild packed weights / BN terms on the next call (needed after writes through ``.data``)."""
        invalidate_derived(self)

    def train(self, mode: bool = True):
        if mode != self.training:
            invalidate_derived(self)
        return super().train(mode)

    def forward(self, x, t) -> torch.Tensor:
        # x: b,c,h,w (token ids as floats); t: b
        if self.training and torch.is_grad_enabled():
            return self._run_train(x, t)
        if self.training:
            raise NotImplementedError('spkdiff: DummyModel in train() mode runs the differentiable training graph and '
                                      'needs autograd enabled; call .eval() for inference')
        if not self._fused_ok():
            raise RuntimeError('spkdiff: DummyModel needs functional.set_step_mode(net, "m") and .eval()')
        if has_hooks(self):
            return self._run_modules(x, t)
        return self._run(ops.den_build_input(x, t), stateful=True)

    def _run_modules(self, x, t):
        """The reference's forward (R/snn_model/vq_diffusion.py:189-208) module by module -- each child a HIP kernel, fp32
        [T,B,C,H,W] tensors in between -- so that forward hooks registered on the children fire."""
        T = self.n_steps
        h = ops.den_build_input(x, t).unsqueeze(0).repeat(T, 1, 1, 1, 1)
        x1 = self.conv1(h)
        x5 = self.conv5(self.conv4(self.conv3(self.conv2(x1))))
        x6 = self.conv6(torch.cat((x5, x1), dim=2))
        return torch.sum(x6, dim=0) / T

    @torch.no_grad()
    def logits_from_tokens(self, x_t, t: int, record=None, inp=None):
        """Sampler fast path: ``self(x_t.float(), full((b,), t))`` followed by ``functional.reset_net(self)``
        (R/snn_model/vq_diffusion.py:128-129) -- every LIF starts from and returns to the reset state, so no
        membrane tensors are read or written."""
        if not self._fused_ok() or self.training:
            raise RuntimeError('spkdiff: DummyModel needs functional.set_step_mode(net, "m") and .eval()')
        functional.reset_net(self)
        # inp: the input map [B,2,h,w] already on the device (the previous spk_psample_step wrote it): no build launch
        return self._run(inp if inp is not None else ops.den_build_input(x_t, int(t)), stateful=False, record=record)
