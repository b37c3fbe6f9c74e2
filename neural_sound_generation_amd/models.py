"""nn.Module surface of the reference's src/models.py (VQ-VAE part), backed by libnsg.so.

    VQVAE(input_dim, dim, z_dim=512)   .encoder .codebook .decoder
        .forward(x) -> (x_tilde, z_e_x, z_q_x)      models.py:198-216
        .encode(x)  -> int64 latents                 models.py:188-191
        .decode(latents) -> x_tilde                  models.py:193-196
    VQEmbedding(z_dim, dim)            .embedding (nn.Embedding), .forward, .straight_through   :121-142
    ResBlock(dim)                      .block (nn.Sequential)                                    :145-158
    weights_init(m)                                                                              :25-32

Constructor signatures, attribute names, state_dict keys/shapes and the RNG consumption order of
construction are the reference's, so checkpoints interchange both ways and `torch.manual_seed(s);
VQVAE(...)` gives the same initial weights.  The child modules are the stock parameter containers
(nn.Conv2d, nn.BatchNorm2d, ...); the stacks that own them override forward() to run the fused HIP
engine (engine.py), so no vendor convolution library is ever called.  There is no CPU fallback:
calling a module on a CPU tensor raises.

Input is the reference's (B, 1, 80, T) float32 mel batch.  Outputs have the reference's logical
NCHW shapes; z_e_x / z_q_x are channels_last in memory.
"""
from __future__ import annotations

import torch
from torch import nn

from . import engine, functional as Fn, ops
from .vector_quantization import vq, vq_st, codebook_lookup, add_per_clip


def to_scalar(arr):
    if type(arr) == list:
        return [x.item() for x in arr]
    return arr.item()


def weights_init(m):
    """Xavier-uniform weights and zero bias on every module whose class name contains 'Conv'
    (models.py:25-32).  Works on the fused stacks too: apply() visits their child containers."""
    classname = m.__class__.__name__
    if classname.find('Conv') != -1:
        try:
            nn.init.xavier_uniform_(m.weight.data)
            m.bias.data.fill_(0)
        except AttributeError:
            print("Skipping initialization of ", classname)


class VQEmbedding(nn.Module):
    """Reference signature VQEmbedding(z_dim, dim) (models.py:121-125).  Extension (not in the
    reference, opt-in): ema_decay=<float> switches the codebook from gradient training to the
    exponential-moving-average update of the VQ-VAE paper (appendix A.1); the codebook then gets no
    gradient and `FusedTrainStep` updates it from all-reduced per-code counts and sums."""

    def __init__(self, z_dim, dim, ema_decay=None, ema_eps=1e-5):
        super().__init__()
        self.embedding = nn.Embedding(z_dim, dim)
        self.embedding.weight.data.uniform_(-1. / z_dim, 1. / z_dim)
        self.ema_decay = ema_decay
        self.ema_eps = ema_eps
        self.search_impl = "mfma"     # "bf16x3" in the bf16 compute mode (set by VQVAE)
        self.scatter_impl = "f32"     # "bf16x2" in the bf16 compute mode: the codebook-gradient scatter-add on the bf16 pipe
        if ema_decay is not None:
            self.embedding.weight.requires_grad_(False)
            self.register_buffer("ema_count", torch.zeros(z_dim))
            self.register_buffer("ema_sum", self.embedding.weight.data.clone())

    def forward(self, z_e_x):
        z_e_x_ = Fn.to_nhwc(z_e_x)
        return vq(z_e_x_, self.embedding.weight, self.search_impl)

    def straight_through(self, z_e_x):
        z_e_x_ = Fn.to_nhwc(z_e_x)
        z_q_x_, indices = vq_st(z_e_x_, self.embedding.weight.detach(), self.search_impl)
        z_q_x = Fn.to_nchw_view(z_q_x_)
        z_q_x_bar_ = codebook_lookup(self.embedding.weight, indices, self.scatter_impl).view_as(z_e_x_)
        z_q_x_bar = Fn.to_nchw_view(z_q_x_bar_)
        return z_q_x, z_q_x_bar


class ResBlock(nn.Module):
    compute_dtype = torch.float32   # storage type of the activations inside the block (fp32 or bfloat16)

    def __init__(self, dim):
        super().__init__()
        self.block = nn.Sequential(
            nn.ReLU(True),
            nn.Conv2d(dim, dim, 3, 1, 1),
            nn.BatchNorm2d(dim),
            nn.ReLU(True),
            nn.Conv2d(dim, dim, 1),
            nn.BatchNorm2d(dim)
        )

    def forward(self, x):
        # y = relu(x) + block(relu(x)), AND the caller's tensor is overwritten with relu(x): the reference's block opens with
        # nn.ReLU(True) (models.py:149) and its residual add reads the mutated tensor (models.py:158; SURVEY.md 8a note 1).
        # Inside VQVAE the fused encoder / decoder stacks never come through here (nothing reads that tensor again there);
        # a stand-alone ResBlock keeps the reference's visible side effect.
        y = Fn.resblock_apply(x, engine.resblock_params(self), self.training, self.compute_dtype)
        with torch.no_grad():
            xn = x.permute(0, 2, 3, 1)
            if xn.is_contiguous():                   # channels_last storage: ReLU in place by the library's own kernel
                ops.convert(xn, torch.float32, out=xn, relu=True)
                # a raw kernel wrote through the pointer: tell autograd, so that an upstream op that saved x raises "modified by
                # an inplace operation" exactly as the reference's nn.ReLU(True) makes it (the NCHW branch below goes through copy_)
                torch.autograd.graph.increment_version(x)
            else:
                x.copy_(Fn.to_nchw_view(ops.convert(xn.contiguous(), torch.float32, relu=True)))
        return y


class _Encoder(nn.Sequential):
    compute_dtype = torch.float32

    def forward(self, x):
        return Fn.encoder_apply(x, engine.encoder_params(self), self.training, self.compute_dtype)


class _Decoder(nn.Sequential):
    compute_dtype = torch.float32

    def forward(self, z):
        return Fn.decoder_apply(z, engine.decoder_params(self), self.training, self.compute_dtype)


class VQVAE(nn.Module):
    def __init__(self, input_dim, dim, z_dim=512, ema_decay=None, n_speakers=None, compute_dtype=torch.float32):
        """Reference signature VQVAE(input_dim, dim, z_dim=512) (models.py:162).  Opt-in extensions,
        neither present in the reference: ema_decay (EMA codebook), n_speakers
        (speaker embedding added to the decoder input, BASELINE configs[2]) and compute_dtype:
        torch.float32 (default; the parity mode) or torch.bfloat16 (activations and conv operands in
        bf16, fp32 accumulation / BatchNorm statistics / quantiser / parameters / optimiser)."""
        super().__init__()
        if compute_dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("compute_dtype must be torch.float32 or torch.bfloat16")
        if compute_dtype == torch.bfloat16 and dim % 8 != 0:
            raise ValueError("compute_dtype=bfloat16 needs dim to be a multiple of 8")
        self.compute_dtype = compute_dtype
        if input_dim != 1:
            raise NotImplementedError("the HIP path implements the speech configuration (input_dim == 1: "
                                      "one-channel 80-bin mel images, src/train.py:115)")
        self.encoder = _Encoder(
            nn.Conv2d(input_dim, dim, 4, 2, 1),
            nn.BatchNorm2d(dim),
            nn.ReLU(True),
            nn.Conv2d(dim, dim, 4, 2, 1),
            ResBlock(dim),
            ResBlock(dim),
        )
        self.codebook = VQEmbedding(z_dim, dim, ema_decay=ema_decay)
        self.decoder = _Decoder(
            ResBlock(dim),
            ResBlock(dim),
            nn.ReLU(True),
            nn.ConvTranspose2d(dim, dim, 4, 2, 1),
            nn.BatchNorm2d(dim),
            nn.ReLU(True),
            nn.ConvTranspose2d(dim, input_dim, 4, 2, 1),
            nn.Tanh()
        )
        self.encoder.compute_dtype = compute_dtype
        self.decoder.compute_dtype = compute_dtype
        if compute_dtype == torch.bfloat16:
            # the quantiser's search on the bf16 matrix pipe with split fp32 operands (distances to ~2^-16; the
            # fp32 parity mode keeps the bit-exact search)
            self.codebook.search_impl = "bf16x3"
            self.codebook.scatter_impl = "bf16x2"
        self.n_speakers = n_speakers
        if n_speakers is not None:
            self.speaker_embedding = nn.Embedding(n_speakers, dim)
            self.speaker_embedding.weight.data.normal_(0.0, 0.1)
        self.apply(weights_init)

    def _condition(self, z_q_x, g):
        """Add the speaker embedding of each clip to every latent pixel of that clip (extension)."""
        if self.n_speakers is None or g is None:
            return z_q_x
        rows = codebook_lookup(self.speaker_embedding.weight, g.view(-1).to(torch.int64))
        return Fn.to_nchw_view(add_per_clip(Fn.to_nhwc(z_q_x), rows))

    def encode(self, x):
        z_e_x = self.encoder(x)
        return self.codebook(z_e_x)

    def decode(self, latents, g=None):
        z_q_x = Fn.to_nchw_view(ops.gather_rows(self.codebook.embedding.weight.detach().contiguous(), latents.contiguous()))
        return self.decoder(self._condition(z_q_x, g))

    def forward(self, x, g=None):
        z_e_x = self.encoder(x)
        z_q_x_st, z_q_x = self.codebook.straight_through(z_e_x)
        x_tilde = self.decoder(self._condition(z_q_x_st, g))
        return x_tilde, z_e_x, z_q_x
