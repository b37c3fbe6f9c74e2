#!/usr/bin/env python
"""Benchmark of the VQ-VAE training hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one full training step (forward, three losses, backward, [all-reduce], Adam) of
VQVAE(1, D=128, K=512) (BASELINE.json configs[1], the reference's `--dim 128 --z-dim 512`, bf16) on a
synthetic batch of B = 128 clips of 80-mel x 1024 frames per GPU, inputs resident in HBM before the timed
region.  Weak scaling: B per GPU is fixed, rank r draws its own clips.  Prints ONE JSON line: the bf16
mode is `value` (configs[1] is quoted in bf16), the fp32 parity mode is timed in the same run (other_mode).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0  # same guide: ~2.5 PFLOP/s dense bf16
PEAK_HBM_GBPS = 8000.0          # same guide: HBM3E 8 TB/s (spec; ~6.3 TB/s achievable)


def flops_per_frame(D: int, K: int) -> float:
    """fwd+bwd FLOP per mel frame."""
    return 2160.0 * D * D + 3200.0 * D + 10.0 * K * D


def host_threads() -> int:
    """CPU threads this process may really use: affinity mask, cgroup quota, and the GPU box's
    documented per-GPU share (16) -- os.cpu_count() reports every core of the host."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, int(os.environ.get("NSG_CPU_THREADS", "16"))))


def cpu_baseline(D, K, T, threads, batch=4, warmup=1, steps=3):
    """The oracle (CPU restatement of the reference path, oracle/vqvae_oracle.py) timed on the host
    cores: a bounded sample (batch clips x `steps` steps) of the same workload."""
    from oracle import vqvae_oracle as O
    from neural_sound_generation_amd import models as M
    torch.set_num_threads(threads)
    torch.manual_seed(1)
    st = O.clone_state(M.VQVAE(1, D, K).state_dict())
    opt = O.adam_init(st)
    c = torch.rand(batch, 1, 80, T, generator=torch.Generator().manual_seed(1234))
    for _ in range(warmup):
        O.train_step(st, opt, c)
    t0 = time.perf_counter()
    for _ in range(steps):
        O.train_step(st, opt, c)
    dt = (time.perf_counter() - t0) / steps
    return batch * T / dt, dt


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without torchrun: start the N ranks ourselves, one fresh process per GPU with torchrun's
    environment contract (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).  This parent makes no GPU call before or after (a
    process that has initialised the GPU must not exec, and device_count() does not initialise it on this image); rank 0
    prints the JSON line on the inherited stdout."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n and "NSG_DEVICE_INDEX" not in os.environ:
        print(f"[bench] --gpus {n} but only {have} GPU(s) visible", file=sys.stderr)
        return 2
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "1"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rcs = [p.wait() for p in procs]
    return max(abs(rc) for rc in rcs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=128, help="clips per GPU (SURVEY.md 8d: best of {32, 64, 128} for configs[1])")
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--z-dim", type=int, default=512)
    ap.add_argument("--frames", type=int, default=1024)
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="bf16",
                    help="bf16 (default: BASELINE.json configs[1] names bf16): bf16 activations / conv operands with fp32 accumulation, "
                         "statistics, quantiser and optimiser; f32: the parity mode (bit-exact code indices, losses within 1e-5 of "
                         "the reference).  The mode not chosen is timed too and reported under other_mode.")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-second-mode", action="store_true", help="skip the short run of the other compute mode")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short runs of BASELINE configs[3] (K=8192, D=256)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    from neural_sound_generation_amd import _lib, distributed as nsg_dist, models as M, ops
    from neural_sound_generation_amd.train import FusedTrainStep

    rank, world, local = nsg_dist.init_from_env()
    if world != args.gpus:
        if rank == 0:
            print(f"[bench] note: WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE", file=sys.stderr)
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs an AMD GPU: the HIP path has no CPU fallback")
    if "NSG_DEVICE_INDEX" in os.environ:       # rehearsal knob: several ranks on one GPU (with NSG_DIST_BACKEND=gloo)
        local = int(os.environ["NSG_DEVICE_INDEX"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    D, K, T, B = args.dim, args.z_dim, args.frames, args.batch

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def run_mode(dtype_name, steps, warmup, use_timer, D=D, K=K, B=B, census_steps=0):
        """Time `steps` training steps of the given compute mode.  Returns dict(value, ms, losses, gather (the dominant
        kernel's HIP-event summary over the timed region), census (per-kernel summary of `census_steps` extra steps taken
        AFTER the timed region with every entry-point call bracketed by events))."""
        torch.manual_seed(1)                       # src/main.py:43,71 -- identical init on every rank
        cdtype = torch.bfloat16 if dtype_name == "bf16" else torch.float32
        model = M.VQVAE(1, D, K, compute_dtype=cdtype).to(dev).train()
        step = FusedTrainStep(model, lr=1e-3, beta=1.0)
        c = torch.rand(B, 1, 80, T, generator=torch.Generator().manual_seed(1234 + rank)).to(dev)
        for _ in range(warmup):
            step.step(c)
        timer = None
        if use_timer:
            timer = ops.KernelTimer()
            ops.KERNEL_TIMER = timer
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            losses = step.step(c)
        sync()
        elapsed = time.perf_counter() - t0
        ops.KERNEL_TIMER = None
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            elapsed = float(t.item())
        out = {"value": world * B * T * steps / elapsed, "ms": elapsed / steps * 1e3, "losses": [float(x.item()) for x in losses],
               "gather": timer.summary().get("gather_gemm_f32") if timer is not None else None, "census": None}
        if census_steps > 0 and rank == 0:
            cen = ops.KernelTimer()
            _lib.CENSUS = cen
            for _ in range(census_steps):
                step.forward_backward(c)           # (no collective inside: the other ranks need not take part)
                step.opt.step()
            torch.cuda.synchronize()
            _lib.CENSUS = None
            out["census"] = {k: dict(v, per_step=v["launches"] / census_steps) for k, v in cen.summary().items()}
        del step, model
        torch.cuda.empty_cache()
        return out

    def _conv_kernel_name(dtype):
        """The kernel behind the conv forward / data-gradient launches that the in-region timer brackets (profiles/*_kernel_stats.csv rows)."""
        if dtype == "bf16":
            return "patch_gemm_kernel<...> (gemm_patch.hip: every 3x3 / 4x4-s2 / transposed conv forward and data gradient, bf16 operands)"
        return "gather_gemm_kernel<float, float, 128x128> (gemm_gather.hip: conv forward and data gradients, fp32 operands)"

    def kernel_table(census, dtype_name, ms_per_step):
        """One row per kernel group of the step: launches per step, HIP-event time per launch, algorithmic work per launch
        (FLOP for the MFMA-bound contractions, bytes moved once per pass the algorithm needs for the HBM-bound ones;
        ops.py states each call's figure where it is launched), achieved rate, fraction of the bounding peak."""
        rows = []
        for label, v in sorted(census.items(), key=lambda kv: -kv[1]["total_ms"]):
            hbm = v["bytes_per_launch"] > 0
            if not hbm and v["flops_per_launch"] <= 0:
                bound, ach, peak, unit = None, None, None, None
            elif hbm:
                bound, ach, peak, unit = "hbm", v["tbps"] * 1e3, PEAK_HBM_GBPS, "GB/s"
            else:
                f32_pipe = dtype_name == "f32" or "fp32" in label or "f32" in label
                bound, ach, peak, unit = "mfma", v["tflops"], (PEAK_F32_MFMA_TFLOPS if f32_pipe else PEAK_BF16_MFMA_TFLOPS), "TFLOP/s"
            rows.append({"kernel": label, "per_step": round(v["per_step"], 2), "us_per_launch": round(v["avg_ms"] * 1e3, 1),
                         "share_of_step": round(v["avg_ms"] * v["per_step"] / ms_per_step, 4), "bound": bound,
                         "gflop_per_launch": round(v["flops_per_launch"] / 1e9, 3) if not hbm and bound else None,
                         "mb_per_launch": round(v["bytes_per_launch"] / 1e6, 2) if hbm else None,
                         "achieved": round(ach, 1) if ach is not None else None, "unit": unit,
                         "frac": round(ach / peak, 4) if ach is not None else None})
        return rows

    main_run = run_mode(args.dtype, args.steps, args.warmup, not args.no_kernel_timer, census_steps=0 if args.no_kernel_timer else 3)
    value, ms_per_step, loss_triple, s = main_run["value"], main_run["ms"], main_run["losses"], main_run["gather"]
    other = None
    if not args.no_second_mode:
        # the other compute mode, same shapes, a short run: fp32 = parity mode, bf16 = throughput mode
        od = "bf16" if args.dtype == "f32" else "f32"
        o = run_mode(od, max(3, args.steps // 4), 2, not args.no_kernel_timer, census_steps=0 if args.no_kernel_timer else 2)
        other = {"dtype": od, "value": round(o["value"], 1), "unit": "mel-frames/s", "ms_per_step": round(o["ms"], 3), "losses": o["losses"]}
        if o["gather"]:
            opeak = PEAK_BF16_MFMA_TFLOPS if od == "bf16" else PEAK_F32_MFMA_TFLOPS
            other["roofline"] = {"bound": "mfma", "kernel": _conv_kernel_name(od), "achieved": round(o["gather"]["tflops"], 2),
                                 "peak": opeak, "unit": "TFLOP/s", "frac": round(o["gather"]["tflops"] / opeak, 4),
                                 "step_frac": round(o["value"] * flops_per_frame(D, K) / 1e12 / opeak, 4)}
            if o["census"]:
                other["roofline"]["kernels"] = kernel_table(o["census"], od, o["ms"])
    other_configs = None
    if not args.no_other_configs and world == 1 and (D, K) == (128, 512):
        # BASELINE configs[3]: the large codebook (K = 8192, D = 256), 16 clips -- the config that stresses the distance
        # contraction + argmin; both searches timed: the bit-exact fp32 one (parity mode) and the bf16x3 one (bf16 mode)
        other_configs = []
        for od in ("bf16", "f32"):
            o = run_mode(od, 5, 2, False, D=256, K=8192, B=16, census_steps=2)
            peak = PEAK_BF16_MFMA_TFLOPS if od == "bf16" else PEAK_F32_MFMA_TFLOPS
            ent = {"workload": "BASELINE configs[3]: VQVAE(1, dim=256, z_dim=8192), 16 clips of 80-mel x %d frames, train step" % T,
                   "dtype": od, "value": round(o["value"], 1), "unit": "mel-frames/s", "ms_per_step": round(o["ms"], 3), "losses": o["losses"],
                   "algorithmic_tflops": round(o["value"] * flops_per_frame(256, 8192) / 1e12, 2),
                   "step_frac": round(o["value"] * flops_per_frame(256, 8192) / 1e12 / peak, 4)}
            if o["census"]:
                rows = kernel_table(o["census"], od, o["ms"])
                vq = [r for r in rows if r["kernel"].startswith("vq_forward")]
                if vq:
                    ent["vq_forward"] = vq[0]
                ent["kernels"] = rows[:8]
            other_configs.append(ent)

    if rank == 0:
        roof = None
        if s is not None:
            traffic = None
            pmc = os.path.join(ROOT, "profiles", "pmc_traffic_latest.json")
            if os.path.exists(pmc) and (D, K, B, T) == (128, 512, 128, 1024):
                # HBM bytes per launch of the same kernel from the committed rocprofv3 --pmc passes of this
                # command (scripts/profile_round.sh + scripts/pmc_summary.py; PMC cannot be sampled inside the timed run)
                ent = json.load(open(pmc)).get(args.dtype)
                if ent:
                    traffic = round(ent["hbm_bytes_per_launch"] / 1e9, 3)
            if s:
                peak = PEAK_BF16_MFMA_TFLOPS if args.dtype == "bf16" else PEAK_F32_MFMA_TFLOPS
                roof = {"bound": "mfma", "kernel": _conv_kernel_name(args.dtype), "achieved": round(s["tflops"], 2),
                        "peak": peak, "unit": "TFLOP/s", "frac": round(s["tflops"] / peak, 4),
                        "traffic": traffic, "traffic_unit": "GB/launch (rocprofv3 PMC, profiles/)", "launches": s["launches"], "avg_launch_ms": round(s["avg_ms"], 4),
                        "gflop_per_launch": round(s["flops_per_launch"] / 1e9, 3),
                        "share_of_step": round(s["total_ms"] / (ms_per_step * args.steps), 3),
                        # the whole step against the same peak: algorithmic FLOP per second / MFMA peak
                        "step_frac": round(value / world * flops_per_frame(D, K) / 1e12 / peak, 4)}
                if main_run["census"]:
                    roof["kernels"] = kernel_table(main_run["census"], args.dtype, ms_per_step)
                    roof["kernels_note"] = ("per-kernel rows: HIP events around every C-ABI call over 3 extra steps taken right after the timed "
                                            "region (same process, same tensors); the headline row above is timed INSIDE the timed region")
        cpu = None
        if not args.no_cpu_baseline and world == 1:     # the CPU baseline is taken at N = 1 only
            threads = host_threads()
            v, dt = cpu_baseline(D, K, T, threads)
            cpu = {"value": round(v, 1), "unit": "mel-frames/s", "cores": threads, "kind": "port",
                   "sample": f"oracle/vqvae_oracle.py train_step, 4 clips x 80x{T}, 1 warm-up + 3 timed steps ({dt * 1e3:.0f} ms/step)"}
        line = {
            "metric": "mel-frames/sec VQ-VAE fwd+bwd+Adam (80-mel x 1024)", "value": round(value, 1), "unit": "mel-frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"BASELINE configs[1]: VQVAE(1, dim={D}, z_dim={K}), {B} clips/GPU of 80-mel x {T} frames, "
                                   f"train step (fwd + 3 losses + bwd + Adam{' + grad all-reduce' if world > 1 else ''})",
                       "clips_per_gpu": B, "global_batch": B * world, "frames": T, "parallelism": f"dp{world}",
                       "per_gpu_value": round(value / world, 1),
                       "algorithmic_tflops": round(value * flops_per_frame(D, K) / 1e12, 2),
                       "losses": loss_triple,
                       "parity": ("bf16 storage: same computation at bf16 accuracy (tests/test_gpu_model.py::test_bf16_mode_*); the north_star "
                                  "parity bar (bit-exact code indices, losses within 1e-5 of the reference CPU path) is met by the fp32 mode "
                                  "timed in this same run (other_mode)") if args.dtype == "bf16" else
                                 "fp32 parity mode: bit-exact code indices, losses within 1e-5 of the reference CPU path (tests/)"},
            "roofline": roof, "cpu_baseline": cpu, "other_mode": other, "other_configs": other_configs,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
