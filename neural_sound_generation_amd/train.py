"""The reference's per-batch training step (src/train.py:104-148, ljspeech branch) on the HIP path.

Two ways to run it:

  * `train_vqvae(args, model, optimizer, train_loader, device, epoch)` -- same signature and
    behaviour as the reference's function: autograd through the model's fused stacks, three
    F.mse_loss terms, optimizer.step().  Any torch optimizer works (FlatAdam is the fused one).

  * `FusedTrainStep` -- the same arithmetic with no autograd in the loop: forward, the three loss
    terms and their gradients from two fused kernels, explicit backward writing straight into the
    flat gradient bucket, one all-reduce when data-parallel, one Adam kernel.  This is what bench.py
    times.  The zero-pad-to-input-width of train.py:118-120 is folded into the loss kernel (the
    reference bounces through host memory there).
"""
from __future__ import annotations

import os

import torch
import torch.nn.functional as F

from . import distributed as nsg_dist
from . import engine, functional as Fn, ops
from .optim import FlatAdam


# The bf16 step without an fp32 z_q (see _forward_backward); NSG_LEAN_VQ=0 keeps the materialised form.
LEAN_VQ = os.environ.get("NSG_LEAN_VQ", "1") == "1"

class FusedTrainStep:
    def __init__(self, model, lr: float = 1e-3, beta: float = 1.0, betas=(0.9, 0.999), eps: float = 1e-8,
                 process_group=None, optimizer: FlatAdam | None = None):
        self.model = model
        self.beta = float(beta)
        self.group = process_group
        self.opt = optimizer if optimizer is not None else FlatAdam(model.parameters(), lr=lr, betas=betas, eps=eps)
        self.world = nsg_dist.world_size(process_group)
        self.dtype = getattr(model, "compute_dtype", torch.float32)
        self.ema = getattr(model.codebook, "ema_decay", None) is not None
        if self.world > 1:
            # every replica starts from rank 0's state: the parameter bucket, and -- packed into one flat tensor, one more
            # broadcast -- everything outside it: BatchNorm running statistics and counters, an EMA-trained codebook
            # (requires_grad=False keeps it out of the bucket) with its ema_count / ema_sum buffers
            nsg_dist.broadcast_flat(self.opt.flat_param, 0, process_group)
            nsg_dist.broadcast_tensors_packed(nsg_dist.state_outside(model, self.opt), 0, process_group)
        if self.ema:
            # [n (K, padded to 256 bytes) | sum z (K x D)] lives BEHIND the gradients: ONE all-reduce carries both
            K, D = model.codebook.embedding.weight.shape
            self._n_pad = (K + 63) // 64 * 64
            tail = self.opt.reserve_tail(self._n_pad + K * D)
            self.ema_n = tail[:K]
            self.ema_s = tail[self._n_pad:].view(K, D)
        # the codebook scatter-add as a sorted segment sum (fp32, deterministic; NSG_SCATTER_IMPL=onehot restores the one-hot
        # GEMMs: bf16x2 on the bf16 pipe in the bf16 mode, exact fp32 products in the fp32 mode)
        if os.environ.get("NSG_SCATTER_IMPL", "sorted") == "sorted":
            self.scatter_impl = "sorted"
        else:
            self.scatter_impl = "bf16x2" if self.dtype == torch.bfloat16 else "f32"
        self.encP = engine.encoder_params(model.encoder)
        self.decP = engine.decoder_params(model.decoder)
        self.codebook = model.codebook.embedding.weight
        self.g_enc = self.opt.grads_for(engine.encoder_param_list(self.encP))
        self.g_dec = self.opt.grads_for(engine.decoder_param_list(self.decP))
        self.g_code = None if self.ema else self.opt.grads_for([self.codebook])[0]
        self.spk = getattr(model, "speaker_embedding", None)
        self.g_spk = self.opt.grads_for([self.spk.weight])[0] if self.spk is not None else None
        # Data parallel: the communication buffer is [encoder | codebook | decoder | speaker table | EMA statistics]; everything
        # from the decoder's first gradient on is final when the decoder's backward has been enqueued, BEFORE the encoder's
        # backward starts, so that part is all-reduced on the collective's own stream beside the encoder backward and only the
        # front part waits for the end of the step (two collectives, the first one hidden).
        lo = self.opt.flat_grad.data_ptr()
        self._comm_split = (self.g_dec[0].data_ptr() - lo) // 4
        if not (0 < self._comm_split < self.opt.flat_grad.numel()) or any(t.data_ptr() < self.g_dec[0].data_ptr() for t in self.g_dec):
            self._comm_split = 0          # (unexpected parameter order: one collective at the end, as before)
        self._reduce = None               # nsg_dist.TwoPartAllReduce over the communication buffer (made on first use: reserve_tail may still grow it)
        self._stepping = False            # collectives only from step(): forward_backward() alone never communicates
        # test hook: (N,) int64 code indices to use INSTEAD of the search's (the search still runs and is ignored).  Lets a test
        # compare this mode's backward with another evaluation's on the SAME codes (tests/test_gpu_model.py, bf16 fidelity).
        self.force_indices = None

    @torch.no_grad()
    def forward_backward(self, c: torch.Tensor, g: torch.Tensor | None = None):
        """c: (B,1,80,T) float32 on the GPU; g: optional (B,) int64 speaker ids (speaker-conditioned
        decoder extension).  Fills the flat gradient bucket; returns the three loss tensors (device
        scalars; nothing here synchronises with the host)."""
        model = self.model
        if not model.training:
            raise RuntimeError("FusedTrainStep needs model.train()")
        x = Fn.to_nhwc(c)
        B, H, T, _ = x.shape
        with engine.deferred_batch_counters():
            return self._forward_backward(x, g, B, H, T)

    def _forward_backward(self, x, g, B, H, T):
        enc_packs, dec_packs = engine.pack_all(self.encP, self.decP, B, H, T, self.dtype)
        ze, es = engine.encoder_forward(x, self.encP, True, dtype=self.dtype, packs=enc_packs)
        D = ze.shape[-1]
        K = self.codebook.shape[0]
        search = getattr(self.model.codebook, "search_impl", "mfma")
        # bf16 mode: z_q is never materialised in fp32 -- the search writes the decoder's (ReLU'd, bf16) input itself, with the
        # clip's speaker row added when the decoder is speaker-conditioned; the losses read codebook[idx], the codebook gradient
        # comes from per-code sums of z_e
        lean = LEAN_VQ and search == "bf16x3" and self.dtype == torch.bfloat16 and D % 8 == 0
        spk_rows = None
        if self.spk is not None and g is not None:
            g = g.view(-1).to(torch.int64).contiguous()
            spk_rows = ops.gather_rows(self.spk.weight.detach(), g)          # (B, D) fp32
        if self.force_indices is not None:
            idx = self.force_indices.view(-1).to(device=ze.device, dtype=torch.int64).contiguous()
            if idx.numel() != ze.numel() // D:
                raise ValueError("force_indices must hold one code index per latent row")
            zq = ops.gather_rows(self.codebook.detach(), idx).view_as(ze)
            zdec = zq
            if lean:
                zdec = ops.add_per_clip(zq, spk_rows) if spk_rows is not None else zq
                zdec, zq = ops.convert(zdec, self.dtype, relu=True), None
        elif lean:
            idx, _, _, zdec = ops.vq_forward(ze.view(-1, D), self.codebook.detach(), want_codes=False, impl=search, codes_bf16="relu",
                                             clip_rows=spk_rows)
            zq = None
            zdec = zdec.view(ze.shape)
        else:
            idx, zq, _ = ops.vq_forward(ze.view(-1, D), self.codebook.detach(), want_codes=True, impl=search)
            zq = zq.view_as(ze)
            zdec = zq
        if spk_rows is not None and not lean:
            zdec = ops.add_per_clip(zq, spk_rows, out_dtype=self.dtype)
        # loss_recons = mse(zero-pad(x_tilde), c) and d/dx_tilde             (train.py:118-129)
        xt, ds = engine.decoder_forward(zdec, self.decP, True, dtype=self.dtype, packs=dec_packs, zq_is_relu=lean, mse_target=x,
                                        mse_dbias=self.g_dec[21])
        if isinstance(xt, tuple):       # the fused output layer formed the loss and the gradient at the Tanh's input with the image
            loss_recons, dpre = xt
            dzq, _ = engine.decoder_backward(dpre, ds, self.decP, need_dz=True, dxt_is_pre_tanh=True, gout=self.g_dec)
        else:
            loss_recons, dxt = ops.mse_padded(xt, x, B * H, xt.shape[2], T)
            dzq, _ = engine.decoder_backward(dxt, ds, self.decP, need_dz=True, gout=self.g_dec)
        if self.spk is not None:
            if g is not None:   # d loss / d speaker rows = per-clip pixel sums of dzq, scattered to the speakers
                gs = ops.index_add_rows(g, ops.clip_colsum(dzq, B), self.spk.weight.shape[0])
                ops.add(gs, None, out=self.g_spk)
            else:
                self.g_spk.zero_()
        if self.ema:    # per-code counts and sums of the assigned encoder rows straight into the communication buffer's tail
            ops.index_add_rows(idx, ze.view(-1, D), K, impl=self.scatter_impl, out=self.ema_s, counts=self.ema_n)
        self._reduce_back_part()        # decoder, speaker table, EMA statistics: final from here on (no-op on one rank / in a graph)
        # loss_vq = mse(z_q, sg(z_e)) -> codebook; loss_commit = mse(z_e, sg(z_q)) -> encoder,
        # plus the straight-through gradient from the decoder               (train.py:131-134)
        bn2_sums = None
        if lean:
            # dz is the incoming gradient of the encoder's closing BatchNorm: its backward sums are formed while dz is written
            bn2 = engine.encoder_closing_bn(es)
            if bn2 is not None and bn2[0].dtype == self.dtype and ops.vq_losses_indexed_bn_supported(D):
                h2, m2, i2 = bn2
                loss_vq, dz, dg, db = ops.vq_losses_indexed(ze.view(-1, D), self.codebook.detach(), idx, dz_scale=self.beta, dz_add=dzq.view(-1, D),
                                                            grad_dtype=self.dtype, bn=(h2.view(-1, D), m2, i2), dgamma=self.g_enc[20],
                                                            dbeta=self.g_enc[21])
                bn2_sums = (dg, db)
            else:
                loss_vq, dz = ops.vq_losses_indexed(ze.view(-1, D), self.codebook.detach(), idx, dz_scale=self.beta, dz_add=dzq.view(-1, D),
                                                    grad_dtype=self.dtype)
            dz = dz.view(ze.shape)
            if not self.ema:    # d loss_vq / d e_k = 2/numel * sum over the rows assigned to k of (e_k - z) = 2/numel * (n_k e_k - s_k)
                s, n = ops.index_add_rows(idx, ze.view(-1, D), K, want_counts=True, impl=self.scatter_impl)
                ops.codebook_grad_from_sums(self.codebook.detach(), n, s, 2.0 / ze.numel(), out=self.g_code)
        elif self.ema:
            # EMA codebook (extension): no codebook gradient; per-code counts and sums of the assigned
            # encoder rows are the statistics every rank contributes (summed over ranks in step())
            loss_vq, dz, _ = ops.vq_losses(ze, zq, dz_scale=self.beta, dq_scale=1.0, dz_add=dzq, want_dq=False, grad_dtype=self.dtype)
        else:
            loss_vq, dz, dq = ops.vq_losses(ze, zq, dz_scale=self.beta, dq_scale=1.0, dz_add=dzq, grad_dtype=self.dtype)
            self._codebook_grad(idx, dq.view(-1, D), K)
        engine.encoder_backward(dz, es, self.encP, gout=self.g_enc, bn2_sums=bn2_sums)
        self.last_indices = idx
        return loss_recons, loss_vq, loss_vq

    def _reduce_back_part(self):
        """Start the all-reduce of [decoder | speaker | EMA statistics] (called when they are final: the collective waits for the
        kernels enqueued so far, then runs on its own stream beside the encoder backward).  step() joins it before Adam."""
        if self.world > 1 and self._comm_split > 0 and self._stepping and not torch.cuda.is_current_stream_capturing():
            self._two_part().start_back()

    def _two_part(self):
        if self._reduce is None or self._reduce.flat.data_ptr() != self.opt.flat_comm.data_ptr() or self._reduce.flat.numel() != self.opt.flat_comm.numel():
            self._reduce = nsg_dist.TwoPartAllReduce(self.opt.flat_comm, self._comm_split, self.group)
        return self._reduce

    def _codebook_grad(self, idx, dq, K):
        g = ops.index_add_rows(idx, dq, K, impl=self.scatter_impl)
        ops.add(g, None, out=self.g_code)

    # ---- HIP graph of forward + backward (optional) ------------------------------------------------------------
    # The ~140 kernel launches that fill the gradient bucket are captured once and replayed (hipGraph through
    # torch.cuda.CUDAGraph: every ctypes launch goes to torch's current stream, which is the capturing stream).  The
    # optimiser step stays outside (its bias-correction scalars change every step), as do the collectives.
    @torch.no_grad()
    def capture(self, c: torch.Tensor, g: torch.Tensor | None = None, warmup: int = 2):
        """Capture forward_backward for inputs of c's shape.  The warm-up steps are REAL training steps."""
        # The captured launches carry the scratch buffer's ADDRESS, and ops.WS keeps one buffer per (device, stream).  So the
        # warm-up steps (first-use work: LDS attributes, workspace growth -- and they are REAL training steps) run on the very
        # stream the capture then uses: the capture finds the warmed-up buffer under its own key and must not outgrow it.  A
        # later call on that stream that needs more scratch makes the workspace allocate a new buffer and drop the old one; this
        # graph holds a reference to the old one (_graph_ws), so its replays never touch memory someone else owns.
        self._graph_stream = torch.cuda.Stream(device=c.device)
        self._graph_stream.wait_stream(torch.cuda.current_stream(c.device))
        with torch.cuda.stream(self._graph_stream):
            for _ in range(warmup):
                self.step(c, g)
            self._static_c = c.clone()
            self._static_g = g.clone() if g is not None else None
            ws = ops.WS.current(c.device)              # (this stream's buffer)
        torch.cuda.current_stream(c.device).wait_stream(self._graph_stream)
        torch.cuda.synchronize()
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph, stream=self._graph_stream):
            self._graph_losses = self.forward_backward(self._static_c, self._static_g)
            ws_after = ops.WS.current(c.device)
        if ws is None or ws_after is not ws:
            self._graph = None
            raise RuntimeError("FusedTrainStep.capture: the workspace grew during capture (warm-up steps must see the captured shapes)")
        self._graph_ws = ws
        return self

    def _replay(self, c, g):
        self._static_c.copy_(c)
        if g is not None:
            self._static_g.copy_(g)
        self._graph.replay()
        return self._graph_losses

    @torch.no_grad()
    def step(self, c: torch.Tensor, g: torch.Tensor | None = None):
        if getattr(self, "_graph", None) is not None and c.shape == self._static_c.shape and (g is None) == (self._static_g is None):
            losses = self._replay(c, g)
        else:
            self._stepping = True
            try:
                losses = self.forward_backward(c, g)
            finally:
                self._stepping = False
        if self.world > 1:
            # the gradient bucket and behind it (EMA mode) the per-code counts and sums: the back part is already in flight
            # (_reduce_back_part), the front part (encoder, codebook) goes now; a replayed graph holds no collective: one for all
            self._two_part().finish()
        self.opt.step(grad_scale=1.0 / self.world)
        if self.ema:
            self.apply_ema()
        return losses

    @torch.no_grad()
    def apply_ema(self, n: torch.Tensor | None = None, s: torch.Tensor | None = None):
        """n (K,), s (K, D): per-code counts and sums of the encoder rows, summed over ranks (default: this step's, as
        left in the communication buffer by the all-reduce): the identical update on every rank."""
        cb = self.model.codebook
        ops.vq_ema_update(self.codebook.data, cb.ema_count, cb.ema_sum, self.ema_n if n is None else n.contiguous(),
                          self.ema_s if s is None else s.contiguous(), decay=cb.ema_decay, eps=cb.ema_eps)


def vqvae_loss_terms(c, x_tilde, z_e_x, z_q_x):
    """train.py:118-133 on device tensors (autograd path): zero-pad without the host round trip."""
    if x_tilde.size(3) != c.size(3):
        target = F.pad(x_tilde, (0, c.size(3) - x_tilde.size(3)))
    else:
        target = x_tilde
    loss_recons = F.mse_loss(target, c)
    loss_vq = F.mse_loss(z_q_x, z_e_x.detach())
    loss_commit = F.mse_loss(z_e_x, z_q_x.detach())
    return loss_recons, loss_vq, loss_commit


def train_vqvae(args, model, optimizer, train_loader, device, epoch):
    """Drop-in for the reference's train_vqvae (src/train.py:104-148, ljspeech branch).
    `train_loader` yields (x, y, c, g, input_lengths) with c: (B, 80, T) mel frames."""
    model.train()
    train_loss = 0
    n_batches = 0
    for batch_idx, (x, y, c, g, input_lengths) in enumerate(train_loader):
        optimizer.zero_grad()
        c = c.to(device).unsqueeze(1)
        x_tilde, z_e_x, z_q_x = model(c)
        loss_recons, loss_vq, loss_commit = vqvae_loss_terms(c, x_tilde, z_e_x, z_q_x)
        loss = loss_recons + loss_vq + args.beta * loss_commit
        loss.backward()
        optimizer.step()
        train_loss = loss_recons.item() + loss_vq.item()
        n_batches += 1
        if batch_idx % args.log_interval == 0:
            print('Train Epoch: {} [{}/{}]\tLoss: {:.6f}'.format(epoch, batch_idx * len(c), len(train_loader.dataset), train_loss))
    return train_loss
