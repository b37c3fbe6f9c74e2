"""CPU: the host-side mirror of the reference interface (no kernel runs): module tree, state_dict
compatibility, initialisation, flat parameter bucket, geometry helpers, loud failure without a GPU."""
import os

import numpy as np
import pytest
import torch

from neural_sound_generation_amd import models as M, ops
from neural_sound_generation_amd._lib import NsgError
from neural_sound_generation_amd.distributed import shard_batch
from neural_sound_generation_amd.optim import FlatAdam
from neural_sound_generation_amd.vector_quantization import vq, vq_st
from oracle import vqvae_oracle as O


def test_state_dict_matches_reference_layout():
    for dim, z_dim in ((16, 32), (64, 128)):
        m = M.VQVAE(1, dim, z_dim)
        assert [(k, tuple(v.shape)) for k, v in m.state_dict().items()] == O.state_keys(dim, z_dim)
    assert sum(p.numel() for p in M.VQVAE(1, 64, 128).parameters()) == 307137
    assert sum(p.numel() for p in M.VQVAE(1, 128, 512).parameters()) == 1253249


def test_initialisation_is_the_references(golden_dir):
    """torch.manual_seed(1); VQVAE(1,16,32) reproduces the reference's initial weights bit for bit
    (same constructors in the same order, xavier-uniform on Conv*, codebook U(-1/K,1/K))."""
    g = np.load(os.path.join(golden_dir, "model_tiny.npz"))
    torch.manual_seed(1)
    m = M.VQVAE(1, 16, 32)
    for k, v in m.state_dict().items():
        assert np.array_equal(v.numpy(), g["sd0." + k]), k
    w = m.codebook.embedding.weight
    assert float(w.abs().max()) <= 1.0 / 32
    assert float(m.encoder[0].bias.abs().max()) == 0.0


def test_module_tree_names_and_load_state_dict(golden_dir):
    g = np.load(os.path.join(golden_dir, "model_cfg1.npz"))
    m = M.VQVAE(1, 64, 128)
    m.load_state_dict({k[4:]: torch.from_numpy(np.array(g[k])) for k in g.files if k.startswith("sd0.")})
    assert isinstance(m.encoder[4], M.ResBlock) and isinstance(m.decoder[3], torch.nn.ConvTranspose2d)
    assert isinstance(m.codebook, M.VQEmbedding) and isinstance(m.codebook.embedding, torch.nn.Embedding)
    assert "VQEmbedding" in repr(m) and "ResBlock" in repr(m)
    assert m.training
    m.eval()
    assert not m.decoder.training
    with pytest.raises(NotImplementedError):
        M.VQVAE(3, 16, 32)


def test_no_cpu_fallback():
    m = M.VQVAE(1, 16, 32)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.rand(1, 1, 80, 16))
    with pytest.raises(NsgError):
        vq(torch.rand(4, 16), torch.rand(8, 16))
    with pytest.raises(NsgError):
        vq_st(torch.rand(4, 16), torch.rand(8, 16))
    opt = FlatAdam(m.parameters())
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        opt.step()


def test_flat_adam_bucket_layout():
    m = M.VQVAE(1, 16, 32)
    before = {k: v.clone() for k, v in m.state_dict().items()}
    opt = FlatAdam(m.parameters(), lr=1e-3)
    base = opt.flat_param.data_ptr()
    for p, off in zip(m.parameters(), opt.offsets):
        assert p.data_ptr() == base + 4 * off and off % 64 == 0       # views into one bucket, 256-byte aligned
        assert p.grad is not None and p.grad.shape == p.shape
    for k, v in m.state_dict().items():
        assert torch.equal(v, before[k])                             # values preserved
    for p in m.parameters():
        p.grad.add_(1.0)
    assert float(opt.flat_grad.sum()) == sum(p.numel() for p in m.parameters())
    opt.zero_grad()
    assert float(opt.flat_grad.abs().sum()) == 0.0 and all(p.grad is not None for p in m.parameters())
    views = opt.grads_for([m.codebook.embedding.weight])
    assert views[0].data_ptr() == m.codebook.embedding.weight.grad.data_ptr()


def test_conv_geometry_helpers():
    d = ops.conv_desc(2, 80, 1024, 1, 128, 4, 2, 1)
    assert (d.OH, d.OW) == (40, 512)
    d = ops.conv_desc(2, 80, 31, 1, 16, 4, 2, 1)
    assert (d.OH, d.OW) == (40, 15)
    d = ops.conv_desc(2, 20, 7, 16, 16, 4, 2, 1, transposed=True)
    assert (d.OH, d.OW) == (40, 14)
    d = ops.conv_desc(2, 20, 256, 128, 128, 3, 1, 1)
    assert (d.OH, d.OW) == (20, 256)
    assert ops._gemm_flops(d) == 2.0 * 2 * 20 * 256 * 9 * 128 * 128


def test_shard_batch():
    b = torch.arange(8).view(8, 1)
    assert shard_batch(b, 1, 4).flatten().tolist() == [2, 3]
    with pytest.raises(ValueError):
        shard_batch(b, 0, 3)


# ---------------------------------------------------------------------------------------------
# on-disk input format, sampler, collate and optimiser-state interchange (8f-3)
# ---------------------------------------------------------------------------------------------
def test_data_root_format_split_and_collate(tmp_path):
    from neural_sound_generation_amd import data as Dm
    root = str(tmp_path / "ljs")
    Dm.write_synthetic_data_root(root, n_utts=40, min_frames=30, max_frames=120, n_speakers=3, seed=3)
    tr = Dm.MelSpecDataSource(root, train=True)
    te = Dm.MelSpecDataSource(root, train=False)
    assert tr.multi_speaker and len(tr) + len(te) == 40 and len(te) == 2          # test_size 0.05, sklearn's split
    assert not set(tr.paths) & set(te.paths)
    assert all(l % Dm.HOP_SIZE == 0 for l in tr.lengths) and len(tr.speaker_ids) == len(tr)
    assert [os.path.basename(p) for p in Dm.RawAudioDataSource(root, train=True).paths] == \
        [os.path.basename(p).replace("mel", "audio") for p in tr.paths]          # the same split for both columns
    one = Dm.MelSpecDataSource(root, train=True, speaker_id=1)                     # a single speaker of the set
    assert not one.multi_speaker and 0 < len(one) < len(tr)

    ds = Dm.MelDataset(tr, Dm.RawAudioDataSource(root, train=True))
    batch = [ds[i] for i in range(6)]
    col = Dm.Collate(max_time_steps=64 * Dm.HOP_SIZE, frame_multiple=4, rng=np.random.RandomState(0))
    x, y, c, g, lens = col(batch)
    B, T = 6, max(min(len(b[1]), 64) // 4 * 4 for b in batch)
    assert tuple(c.shape) == (B, 80, T) and c.dtype == torch.float32 and c.is_contiguous()
    assert tuple(x.shape) == (B, 1, T * Dm.HOP_SIZE) and tuple(y.shape) == (B, T * Dm.HOP_SIZE, 1)
    assert g.dtype == torch.int64 and tuple(g.shape) == (B,) and lens.tolist() == [min(len(b[1]), 64) // 4 * 4 * Dm.HOP_SIZE for b in batch]
    for i, (xa, ca, _) in enumerate(batch):                                        # crop is frame-aligned on both streams
        n = lens[i].item() // Dm.HOP_SIZE
        win = c[i, :, :n].t().numpy()
        starts = [s for s in range(len(ca) - n + 1) if np.array_equal(ca[s:s + n], win)]
        assert starts, "the mel crop must be a contiguous window of the utterance"
        assert any(np.array_equal(xa[s * Dm.HOP_SIZE:(s + n) * Dm.HOP_SIZE], x[i, 0, :n * Dm.HOP_SIZE].numpy()) for s in starts)
        assert float(c[i, :, n:].abs().sum()) == 0.0                                # zero padding
    with pytest.raises(ValueError):
        open(os.path.join(root, "train.txt"), "w").write("a|b|c\n")
        Dm.MelSpecDataSource(root)


def test_similar_length_sampler_groups_by_length():
    from neural_sound_generation_amd.data import PartialyRandomizedSimilarTimeLengthSampler
    import random
    random.seed(0)
    n = 4099                                                                         # 8 groups of 32 batches + a tail
    lengths = list(np.random.RandomState(1).randint(100, 10000, size=n))
    s = PartialyRandomizedSimilarTimeLengthSampler(lengths, batch_size=16)
    order = list(iter(s))
    assert sorted(order) == list(range(n)) and len(s) == n                          # a permutation
    spreads = [np.ptp([lengths[i] for i in order[b:b + 16]]) for b in range(0, 4096, 16)]
    assert np.mean(spreads) < 0.2 * np.ptp(lengths)                                  # batches hold similar lengths (one group's spread)
    assert order != list(iter(s))                                                    # and it is randomised


def test_flat_adam_state_dict_interchanges_with_torch_adam():
    torch.manual_seed(0)
    ref = [torch.nn.Parameter(torch.randn(5, 3)), torch.nn.Parameter(torch.randn(7))]
    topt = torch.optim.Adam(ref, lr=2e-3)
    for _ in range(3):
        for p in ref:
            p.grad = torch.randn_like(p)
        topt.step()
    mine = [torch.nn.Parameter(p.detach().clone()) for p in ref]
    fopt = FlatAdam(mine, lr=1e-3)
    fopt.load_state_dict(topt.state_dict())                                          # torch -> flat bucket
    assert fopt.step_count == 3 and fopt.param_groups[0]["lr"] == 2e-3
    for i, (p, off) in enumerate(zip(mine, fopt.offsets)):
        assert torch.equal(fopt.exp_avg[off:off + p.numel()].view_as(p), topt.state[ref[i]]["exp_avg"])
        assert torch.equal(fopt.exp_avg_sq[off:off + p.numel()].view_as(p), topt.state[ref[i]]["exp_avg_sq"])
    sd = fopt.state_dict()                                                           # flat bucket -> torch
    topt2 = torch.optim.Adam([torch.nn.Parameter(p.detach().clone()) for p in ref], lr=1e-3)
    topt2.load_state_dict(sd)
    for p in topt2.param_groups[0]["params"]:
        p.grad = torch.ones_like(p)
    for p in ref:
        p.grad = torch.ones_like(p)
    topt2.step(); topt.step()                                                        # identical continuation
    for a, b in zip(topt2.param_groups[0]["params"], ref):
        assert torch.equal(a, b)


def test_prior_module_tree_and_initialisation_are_the_references(golden_dir):
    """GatedPixelCNN: same state_dict keys / shapes and, under the same seed, the reference's initial weights bit for bit."""
    from neural_sound_generation_amd.prior import GatedPixelCNN
    from oracle import pixelcnn_oracle as P
    g = np.load(os.path.join(golden_dir, "prior_tiny.npz"))
    input_dim, dim, n_layers, n_classes = (int(v) for v in g["cfg"])
    torch.manual_seed(1)
    m = GatedPixelCNN(input_dim, dim, n_layers, n_classes)
    assert [(k, tuple(v.shape)) for k, v in m.state_dict().items()] == P.state_keys(input_dim, dim, n_layers, n_classes)
    for k, v in m.state_dict().items():
        assert np.array_equal(v.numpy(), g["sd0." + k]), k
    assert m.layers[0].mask_type == 'A' and not m.layers[0].residual and m.layers[1].mask_type == 'B' and m.layers[1].residual
    with pytest.raises(NsgError):
        m(torch.zeros(1, 4, 4, dtype=torch.int64), torch.zeros(1, dtype=torch.int64))     # CPU tensors: no fallback
