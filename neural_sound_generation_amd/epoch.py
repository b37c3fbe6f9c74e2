"""One epoch of the reference's driver loop (src/main.py:128-220, vqvae / ljspeech branch) with every stage on this
package: train_vqvae -> test_vqvae -> reconstruct the first test batch -> np.save it -> invert the last clip's mel to a
waveform (Griffin-Lim) -> save_wav -> checkpoint.  File names follow the reference's.  The CLI / argument parsing of
main.py stays out of scope; `args` is any object with the fields used below
(model, dataset, dim, z_dim, beta, log_interval, sampledir).
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import audio as nsg_audio
from .evaluate import checkpoint_state, save_checkpoint, test_vqvae
from .train import train_vqvae

SAMPLING_RATE, FFT_SIZE, HOP_SIZE, N_MELS = 22050, 1024, 256, 80      # src/main.py:167-170


def run_epoch(args, model, optimizer, train_loader, test_loader, device, epoch, checkpoint_path=None, export_audio=True):
    """Returns a dict with the numbers and the files written."""
    train_loss = train_vqvae(args, model, optimizer, train_loader, device, epoch)
    loss_recons, loss_vq = test_vqvae(args, model, test_loader, device, epoch)
    out = {"train_loss": train_loss, "test_loss_recons": loss_recons, "test_loss_vq": loss_vq}
    sample_dir = os.path.join(args.sampledir, format(args.dataset))
    os.makedirs(sample_dir, exist_ok=True)
    stem = '_' + str(args.model) + '_data_' + str(args.dataset) + '_dim_' + str(args.dim) + '_z_dim_' + str(args.z_dim) + '_epoch_' + str(epoch)
    with torch.no_grad():
        x, y, c, g, input_lengths = next(iter(test_loader))
        c = c.to(device).unsqueeze(1)
        print("Evaluating samples")
        model.eval()
        reconstruction, _, _ = model(c)                                   # main.py:150-153
        reconstruction = reconstruction.squeeze(1)
        rec_np = reconstruction.float().cpu().numpy()
        out["reconstruction"] = os.path.join(sample_dir, 'reconstruction' + stem + '.npy')
        np.save(out["reconstruction"], rec_np, allow_pickle=False)
        if export_audio:
            print("Trying audio reconstruction on test set..")
            # (the reference concatenates the batch's mels but then inverts `mel`, the last clip: main.py:166-187)
            mel = reconstruction[-1:].contiguous()
            assert mel.shape[1] == N_MELS
            signal = nsg_audio.inv_mel_spectrogram(mel, SAMPLING_RATE, FFT_SIZE, HOP_SIZE, N_MELS)[0].cpu().numpy()
            out["wav"] = os.path.join(sample_dir, 'audio_recon' + stem + '_fftsize_' + str(FFT_SIZE) + '_hopsize_' + str(HOP_SIZE) + '.wav')
            nsg_audio.save_wav(signal, out["wav"], SAMPLING_RATE)
    out["checkpoint"] = save_checkpoint(args, checkpoint_state(epoch, args.model, model, optimizer), filename=checkpoint_path)
    return out
