"""Timing of the mel -> waveform inversion (60 Griffin-Lim iterations) for a batch of reconstructions (B x 80 x 1024)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_sound_generation_amd import audio as Au

dev = "cuda:0"
for B in (1, 16, 64):
    mel = torch.rand(B, 80, 1024, device=dev)
    Au.inv_mel_spectrogram(mel, iters=2)
    torch.cuda.synchronize()
    t = time.perf_counter()
    y = Au.inv_mel_spectrogram(mel)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print(f"inv_mel_spectrogram: B={B} x 1024 frames, 60 Griffin-Lim iterations: {dt * 1e3:.1f} ms = {B * y.shape[1] / 22050 / dt:.0f} x real time ({y.shape[1] / 22050:.1f} s of audio per clip)")
