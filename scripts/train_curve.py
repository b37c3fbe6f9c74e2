"""Does the bf16 throughput mode TRAIN like the fp32 parity mode?  (The gradient cosines of tests/test_gpu_model.py pin one step;
this pins what they are for.)  Two models from the same seed, the same stream of synthetic batches, N optimiser steps each through
FusedTrainStep -- fp32 activations with the exact search, and bf16 activations with the bf16x3 search -- and the three losses
and the codebook perplexity, step by step.

The batches are mel-like rather than white noise (data.synthetic_mel_batch: a VQ-VAE cannot reconstruct noise, so its loss
would say nothing).

    python scripts/train_curve.py [--steps 200] [--clips 32] [--dim 128] [--z-dim 512] [--frames 1024] [--out gpurun_out/train_curve.json]
"""
import argparse
import json
import math
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neural_sound_generation_amd.data import synthetic_mel_batch   # noqa: E402
from neural_sound_generation_amd.models import VQVAE          # noqa: E402
from neural_sound_generation_amd.train import FusedTrainStep  # noqa: E402


def perplexity(idx, K):
    p = torch.bincount(idx.view(-1), minlength=K).double()
    p = p / p.sum()
    return float(torch.exp(-(p[p > 0] * p[p > 0].log()).sum()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--clips", type=int, default=32)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--z-dim", type=int, default=512)
    ap.add_argument("--frames", type=int, default=1024)
    ap.add_argument("--every", type=int, default=10)
    ap.add_argument("--out", default="gpurun_out/train_curve.json")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    runs = {}
    for name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        torch.manual_seed(1)                                    # src/main.py:43
        model = VQVAE(1, a.dim, a.z_dim, compute_dtype=dt).to(dev).train()
        step = FusedTrainStep(model, lr=1e-3, beta=1.0)
        gen = torch.Generator(device=dev).manual_seed(1234)     # the same batches for both runs
        rec = []
        for s in range(a.steps):
            c = synthetic_mel_batch(a.clips, a.frames, gen, dev)
            l = step.step(c)
            if s % a.every == 0 or s == a.steps - 1:
                row = dict(step=s, recons=l[0].item(), vq=l[1].item(), commit=l[2].item(),
                           perplexity=perplexity(step.last_indices, a.z_dim))
                if not all(math.isfinite(row[k]) for k in ("recons", "vq", "commit")):
                    raise SystemExit(f"{name}: non-finite loss at step {s}: {row}")
                rec.append(row)
                print(f"[{name}] step {s:4d}  recons {row['recons']:.5f}  vq {row['vq']:.3e}  perplexity {row['perplexity']:.1f}", flush=True)
        runs[name] = rec
    rows = []
    for r32, r16 in zip(runs["f32"], runs["bf16"]):
        rows.append(dict(step=r32["step"], recons_f32=r32["recons"], recons_bf16=r16["recons"],
                         rel=abs(r16["recons"] - r32["recons"]) / max(r32["recons"], 1e-12),
                         vq_f32=r32["vq"], vq_bf16=r16["vq"], perplexity_f32=r32["perplexity"], perplexity_bf16=r16["perplexity"]))
    out = dict(config=dict(steps=a.steps, clips=a.clips, dim=a.dim, z_dim=a.z_dim, frames=a.frames, data="data.synthetic_mel_batch, seed 1234", lr=1e-3),
               curve=rows)
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)
    last = rows[-1]
    print(f"final: recons f32 {last['recons_f32']:.5f}  bf16 {last['recons_bf16']:.5f}  (rel {last['rel']:.3f}); "
          f"first: {rows[0]['recons_f32']:.5f}")


if __name__ == "__main__":
    main()
