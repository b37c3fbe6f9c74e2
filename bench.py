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


def cpu_baseline(D, K, T, threads, batch=4, warmup=1, steps=10):
    """The oracle (CPU restatement of the reference path, oracle/vqvae_oracle.py) timed on the host
    cores: a bounded sample (batch clips x `steps` steps, about 10 s of CPU work) of the same workload."""
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


LINE_LIMIT = 3072   # the driver keeps only the tail of stdout: the record must be ONE short line


def build_line(*, value, ms_per_step, world, steps, warmup, dtype, D, K, B, T, roof=None, cpu=None, other=None, other_configs=None) -> str:
    """The ONE JSON line of the bench contract, compact by construction (< LINE_LIMIT bytes; tests/test_bench_line.py).
    Per-kernel tables never go in here: `write_kernel_tables` puts them in a side file and on stderr."""
    line = {
        "metric": "mel-frames/sec VQ-VAE fwd+bwd+Adam (80-mel x 1024)", "value": round(value, 1), "unit": "mel-frames/s",
        "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
        "config": {"workload": f"BASELINE configs[1]: VQVAE(1,dim={D},z_dim={K}) train step, {B} clips/GPU x 80-mel x {T} frames",
                   "clips_per_gpu": B, "global_batch": B * world, "frames": T, "parallelism": f"dp{world}"},
        "roofline": roof, "cpu_baseline": cpu, "other_mode": other, "other_configs": other_configs,
    }
    text = json.dumps(line, separators=(",", ":"))
    if len(text) >= LINE_LIMIT:      # never let an addition push the head of the record out of the driver's window again
        line["other_configs"] = None
        text = json.dumps(line, separators=(",", ":"))
    assert len(text) < LINE_LIMIT, len(text)
    return text


def write_kernel_tables(tables: dict) -> str:
    """Per-kernel tables (launches per step, HIP-event time, algorithmic work, fraction of the bounding peak) go to
    `bench_kernels.json` (under gpurun_out/ when that exists, else next to this script) and to stderr -- never on stdout."""
    out_dir = os.path.join(ROOT, "gpurun_out") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else ROOT
    path = os.path.join(out_dir, "bench_kernels.json")
    try:
        with open(path, "w") as f:
            json.dump(tables, f, indent=1)
    except OSError as e:
        print(f"[bench] could not write {path}: {e}", file=sys.stderr)
    for name, rows in tables.items():
        print(f"[bench] kernels {name}:", file=sys.stderr)
        for r in rows:
            print("[bench]   " + json.dumps(r, separators=(",", ":")), file=sys.stderr)
    sys.stderr.flush()
    return path


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

    def run_mode(dtype_name, steps, warmup, use_timer, D=D, K=K, B=B, census_steps=0, n_speakers=None):
        """Time `steps` training steps of the given compute mode.  Returns dict(value, ms, losses, gather (the dominant
        kernel's HIP-event summary over the timed region), census (per-kernel summary of `census_steps` extra steps taken
        AFTER the timed region with every entry-point call bracketed by events))."""
        torch.manual_seed(1)                       # src/main.py:43,71 -- identical init on every rank
        cdtype = torch.bfloat16 if dtype_name == "bf16" else torch.float32
        model = M.VQVAE(1, D, K, compute_dtype=cdtype, n_speakers=n_speakers).to(dev).train()
        step = FusedTrainStep(model, lr=1e-3, beta=1.0)
        gen = torch.Generator().manual_seed(1234 + rank)
        c = torch.rand(B, 1, 80, T, generator=gen).to(dev)
        g = torch.randint(0, n_speakers, (B,), generator=gen).to(dev) if n_speakers else None
        for _ in range(warmup):
            step.step(c, g)
        timer = None
        if use_timer:
            timer = ops.KernelTimer()
            ops.KERNEL_TIMER = timer
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            losses = step.step(c, g)
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
                step.forward_backward(c, g)        # (no collective inside: the other ranks need not take part)
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
            return "patch_gemm_kernel (gemm_patch.hip: conv forward + data gradient, bf16)"
        return "gather_gemm_kernel (gemm_gather.hip: conv forward + data gradient, fp32)"

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
    value, ms_per_step, s = main_run["value"], main_run["ms"], main_run["gather"]
    tables = {}
    if main_run["census"]:
        tables[f"configs[1] {args.dtype} B={B}"] = kernel_table(main_run["census"], args.dtype, ms_per_step)
    other = None
    if not args.no_second_mode:
        # the other compute mode, same shapes, a short run: fp32 = parity mode, bf16 = throughput mode
        od = "bf16" if args.dtype == "f32" else "f32"
        o = run_mode(od, max(3, args.steps // 4), 2, not args.no_kernel_timer, census_steps=0 if args.no_kernel_timer else 2)
        opeak = PEAK_BF16_MFMA_TFLOPS if od == "bf16" else PEAK_F32_MFMA_TFLOPS
        other = {"dtype": od, "value": round(o["value"], 1), "ms_per_step": round(o["ms"], 3),
                 "frac": round(o["gather"]["tflops"] / opeak, 4) if o["gather"] else None,
                 "step_frac": round(o["value"] / world * flops_per_frame(D, K) / 1e12 / opeak, 4)}
        if o["census"]:
            tables[f"configs[1] {od} B={B}"] = kernel_table(o["census"], od, o["ms"])
    other_configs = None
    if not args.no_other_configs and world == 1 and (D, K) == (128, 512):
        # the other single-GPU configurations of BASELINE.json, short runs:
        #   configs[2]  speaker-conditioned decoder (7 speakers, hparams.py:84), K = 512, D = 128, same batch as the headline
        #   configs[3]  the large codebook (K = 8192, D = 256), 32 clips (the best of SURVEY.md 8d's {8, 16, 32}: scripts/batch_sweep_c3.sh):
        #               stresses the distance contraction + argmin; both searches: the bit-exact fp32 one (parity mode) and the
        #               bf16x3 one (bf16 mode)
        #   configs[4]  its per-GPU share on ONE GPU: 256 clips (the 8-GPU curve itself is the driver's to measure)
        other_configs = []
        plan = [("configs[2] 7 speakers", "bf16", 128, 512, B, 7, 5), ("configs[3] D=256 K=8192", "bf16", 256, 8192, 32, None, 5),
                ("configs[3] D=256 K=8192", "f32", 256, 8192, 32, None, 4), ("configs[4] share: 256 clips", "bf16", 128, 512, 256, None, 4)]
        for tag, od, d_, k_, b_, spk, st in plan:
            o = run_mode(od, st, 2, False, D=d_, K=k_, B=b_, census_steps=2, n_speakers=spk)
            peak = PEAK_BF16_MFMA_TFLOPS if od == "bf16" else PEAK_F32_MFMA_TFLOPS
            ent = {"workload": tag, "dtype": od, "clips": b_, "value": round(o["value"], 1), "ms_per_step": round(o["ms"], 3),
                   "step_frac": round(o["value"] * flops_per_frame(d_, k_) / 1e12 / peak, 4)}
            if o["census"]:
                rows = kernel_table(o["census"], od, o["ms"])
                tables[f"{tag} {od} B={b_}"] = rows
                vq = [r for r in rows if r["kernel"].startswith("vq_forward")]
                if vq and k_ >= 4096:
                    ent["vq_frac"] = vq[0]["frac"]
                    ent["vq_share"] = vq[0]["share_of_step"]
            other_configs.append(ent)

    if rank == 0:
        roof = None
        if s:
            traffic = None
            pmc = os.path.join(ROOT, "profiles", "pmc_traffic_latest.json")
            if os.path.exists(pmc) and (D, K, B, T) == (128, 512, 128, 1024):
                # HBM bytes per launch of the same kernel from the committed rocprofv3 --pmc passes of this
                # command (scripts/profile_round.sh + scripts/pmc_summary.py; PMC cannot be sampled inside the timed run)
                ent = json.load(open(pmc)).get(args.dtype)
                if ent:
                    traffic = round(ent["hbm_bytes_per_launch"] / 1e9, 3)
            peak = PEAK_BF16_MFMA_TFLOPS if args.dtype == "bf16" else PEAK_F32_MFMA_TFLOPS
            roof = {"bound": "mfma", "kernel": _conv_kernel_name(args.dtype), "achieved": round(s["tflops"], 2),
                    "peak": peak, "unit": "TFLOP/s", "frac": round(s["tflops"] / peak, 4),
                    "traffic": traffic, "traffic_unit": "GB/launch (rocprofv3 PMC, profiles/)", "launches": s["launches"],
                    "avg_launch_ms": round(s["avg_ms"], 4), "gflop_per_launch": round(s["flops_per_launch"] / 1e9, 3),
                    "share_of_step": round(s["total_ms"] / (ms_per_step * args.steps), 3),
                    # the whole step against the same peak: algorithmic FLOP per second / MFMA peak
                    "step_frac": round(value / world * flops_per_frame(D, K) / 1e12 / peak, 4)}
        cpu = None
        if not args.no_cpu_baseline and world == 1:     # the CPU baseline is taken at N = 1 only
            threads = host_threads()
            v, dt = cpu_baseline(D, K, T, threads)
            cpu = {"value": round(v, 1), "unit": "mel-frames/s", "cores": threads, "kind": "port",
                   "sample": f"oracle train_step, 4 clips x 80x{T}, 1 warm-up + 10 timed steps ({dt * 1e3:.0f} ms/step)"}
        if tables:
            write_kernel_tables(tables)
        print(build_line(value=value, ms_per_step=ms_per_step, world=world, steps=args.steps, warmup=args.warmup, dtype=args.dtype,
                         D=D, K=K, B=B, T=T, roof=roof, cpu=cpu, other=other, other_configs=other_configs), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
