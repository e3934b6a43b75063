#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE.json's config.

Workload (configs[1]): a 4096-channel biquad IIR bank (one resonator per
channel, 50 Hz .. 20 kHz log-spaced at 48 kHz), float64, 1 Mi-sample blocks,
time-major [N, C] rows (the reference's vector-valued-sample layout), one
MI355X per rank.  A "step" is one pass of the bank over one block of synthetic
uniform(-1, 1) noise that is already resident in HBM; filter state carries over
from step to step, so K steps are one continuous K*N-sample stream per channel.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU; channels are
   sharded with no data-path collective -> "scaling": "weak")

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
ALG_BYTES_PER_SAMPLE = 16.0    # 8 B read + 8 B written per channel-sample


def resonator_coefs(C, fs=48000.0):
  """resonator.z_exp(fc, fc/10) per channel (reference lazy_filters.py:1245-1276),
  designed by audiolazy_amd's own host-side mirror when present."""
  fc = np.geomspace(50.0, 20000.0, C)
  try:
    from audiolazy_amd.filters import resonator
    filts = [resonator.z_exp(2 * np.pi * f / fs, 2 * np.pi * f / 10 / fs) for f in fc]
    b = np.array([fl.numlist for fl in filts], dtype=np.float64)
    a = np.array([fl.denlist for fl in filts], dtype=np.float64)
    return b, a
  except ImportError:
    w, bw = 2 * np.pi * fc / fs, 2 * np.pi * fc / 10 / fs
    r = np.exp(-bw / 2)
    a = np.stack([np.ones(C), -2 * r * np.cos(w), r * r], axis=1)
    g = (1 - r * r) / 2
    return np.stack([g, np.zeros(C), -g], axis=1), a


def fir_taps(ntaps=256, fc=0.1):
  """256-point Hamming-windowed sinc lowpass at 0.1 fs in plain float64."""
  import math
  m = (ntaps - 1) / 2.0
  out = []
  for i in range(ntaps):
    t = i - m
    ideal = 2 * fc if t == 0 else math.sin(2 * math.pi * fc * t) / (math.pi * t)
    out.append(ideal * (0.54 - 0.46 * math.cos(2 * math.pi * i / (ntaps - 1))))
  return np.array(out)


def cpu_baseline(b, a, n_samples, budget_s=12.0):
  """The oracle (a C port of the reference's generated loop, 1 thread) on a bounded
  sample of the same workload: 64 channels x n_samples, repeated until ~budget_s."""
  from oracle import oracle
  C = 64
  n = min(n_samples, 1 << 20)
  rng = np.random.default_rng(1)
  x = rng.uniform(-1, 1, (C, n))
  sel = np.linspace(0, b.shape[0] - 1, C).astype(int)
  bs, as_ = np.ascontiguousarray(b[sel]), np.ascontiguousarray(a[sel])
  oracle.bank([3], [3], bs, as_, x[:, :1024], layout="chan")  # warm
  done, t0 = 0, time.perf_counter()
  while True:
    oracle.bank([3], [3], bs, as_, x, layout="chan")
    done += C * n
    el = time.perf_counter() - t0
    if el >= budget_s:
      break
  return {"value": done / el / 1e9, "unit": "Gsamples/s", "cores": 1, "kind": "port",
          "sample": "oracle/alz_oracle.c DF-I loop, %d of the %d channels x %d samples, channel-major, "
                    "%d passes, 1 thread (host has %d logical cores)" % (C, b.shape[0], n, done // (C * n),
                                                                         os.cpu_count() or 0)}


def side_workload(args, alz, torch, dev, rank, world, local, red_dev):
  """configs[3] (gammatone bank) and configs[4] (LPC frames): same timing protocol, own metric."""
  import torch.distributed as dist
  ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
  g = torch.Generator(device=dev).manual_seed(20260924 + rank)
  if args.workload == "gammatone":
    B, S, N = 256, 64, 1 << 16            # 512 streams over 8 GPUs -> 64 streams per GPU, all bands local
    s_, Hz = alz.sHz(48000)
    fcs = [f * Hz for f in alz.erb_space(50., 20000., B)]
    bank = alz.gammatone_bank(fcs, S, strategy="slaney", Hz=Hz, device=local)
    bank.reset()
    x = torch.empty((S, N), dtype=torch.float64, device=dev).uniform_(-1.0, 1.0, generator=g)
    y = torch.empty((B * S, N), dtype=torch.float64, device=dev)
    step = lambda: bank.process(x, layout="chan", out=y)
    units, unit_name = float(B) * S * N, "Gsamples/s"
    alg_bytes = (8.0 + 8.0 / B) * B * S * N
    metric = "Gsamples/s (band x stream x sample outputs) through the ERB gammatone bank"
    workload = ("configs[3]: ERB gammatone filterbank (gammatone.slaney, 4-section cascades), %d bands x %d "
                "input streams per GPU (512 streams sharded over 8), %d-sample blocks, float64, "
                "x [S, N] -> y [B, S, N]" % (B, S, N))
  else:
    F, L, order = 65536, 480, 16
    sig = torch.empty((F * L,), dtype=torch.float64, device=dev).uniform_(-1.0, 1.0, generator=g)
    from audiolazy_amd.lpc import kautocor_frames
    step = lambda: kautocor_frames(sig, L, order)
    units, unit_name = float(F), "Gframes/s"
    alg_bytes = 3984.0 * F
    metric = "Gframes/s through lpc.kautocor (autocorrelation + Levinson-Durbin, order 16, 10 ms frames)"
    workload = "configs[4]: lazy_lpc order-16 on %d concurrent 480-sample frames, float64" % F

  def sync_all():
    torch.cuda.synchronize(dev)
    if world > 1:
      dist.barrier()
    torch.cuda.synchronize(dev)
  for _ in range(args.warmup):
    step()
  sync_all()
  t0 = time.perf_counter()
  for s in range(args.steps):
    ev[s][0].record()
    step()
    ev[s][1].record()
  sync_all()
  elapsed = time.perf_counter() - t0
  if world > 1:
    t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
  parity = None
  if rank == 0 and not args.no_parity_check:
    try:   # the oracle is a checker, never a dependency of the timed path
      from oracle import oracle
      if args.workload == "gammatone":
        nchk = 256
        bank.reset()
        xs = x[:, :nchk].contiguous()
        got = bank.process(xs, layout="chan").cpu().numpy()
        k = alz.gammatone_erb_constants(4)[0]
        bands = [alz.gammatone.slaney(fc, k * alz.erb(fc, Hz)) for fc in fcs]
        nbs, nas = [len(f.numlist) for f in bands[0]], [len(f.denlist) for f in bands[0]]
        bcat = np.repeat(np.array([sum((f.numlist for f in band), []) for band in bands]), S, axis=0)
        acat = np.repeat(np.array([sum((f.denlist for f in band), []) for band in bands]), S, axis=0)
        ref = oracle.bank(nbs, nas, bcat, acat, np.tile(xs.cpu().numpy(), (B, 1)), layout="chan")
        parity = "bit-exact" if np.array_equal(got.view(np.uint64), ref.view(np.uint64)) else "MISMATCH"
      else:
        nf = 256
        coefs, err, status = kautocor_frames(sig[:nf * L].contiguous(), L, order)
        rc, re, rs = oracle.kautocor_frames(sig[:nf * L].cpu().numpy(), nf, L, L, order)
        worst = float(np.max(np.abs(coefs.cpu().numpy() - rc) / np.maximum(1.0, np.abs(rc))))
        parity = ("coefficients within %.1e of the oracle (Levinson is not bit-pinned; contract 1e-6)" % worst
                  if worst <= 1e-9 and np.array_equal(status.cpu().numpy(), rs) else "MISMATCH (%.3g)" % worst)
    except Exception as exc:
      parity = "unchecked (%s)" % exc
  if rank == 0:
    k_ms = sum(e0.elapsed_time(e1) for e0, e1 in ev) / len(ev)
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9
    print(json.dumps({
      "metric": metric, "value": world * units * args.steps / elapsed / 1e9, "unit": unit_name,
      "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
      "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
      "config": {"workload": workload, "kernel": bank.last_kernel if args.workload == "gammatone" else "k_acorr_stage<17,lev> (autocorrelation + Levinson-Durbin in one launch)",
                 "parity_spot_check": parity},
      "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": achieved / HBM_PEAK_GBS, "traffic": None, "kernel_ms_avg": k_ms,
                   "algorithmic_bytes_per_launch": alg_bytes}}))
  if world > 1:
    dist.barrier()
    dist.destroy_process_group()


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=10)
  ap.add_argument("--warmup", type=int, default=2)
  ap.add_argument("--channels", type=int, default=4096, help="channels per GPU")
  ap.add_argument("--log2-samples", type=int, default=20, help="block length per channel = 2**this")
  ap.add_argument("--layout", choices=["time", "chan"], default="time")
  ap.add_argument("--no-cpu-baseline", action="store_true")
  ap.add_argument("--no-parity-check", action="store_true",
                  help="skip the post-run 4096-sample parity launch (keeps a rocprofv3 --stats average clean)")
  ap.add_argument("--fused", action="store_true",
                  help="opt-in FMA mode of the streaming kernel: NOT bit-exact (reported as such); default off")
  ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                  help="process-group backend for N > 1 (nccl = RCCL; gloo lets tests run several ranks on one GPU)")
  ap.add_argument("--workload", choices=["biquad", "fir", "gammatone", "lpc"], default="biquad",
                  help="biquad = configs[1] (the contract line); fir = configs[2]; gammatone = configs[3] "
                       "(256 bands x 64 streams per GPU); lpc = configs[4] (65536 frames x 480, order 16)")
  args = ap.parse_args()

  import torch
  import audiolazy_amd as alz
  alz.load_library()  # raises loudly when libalzhip.so is missing: there is no CPU path

  rank = int(os.environ.get("RANK", "0"))
  world = int(os.environ.get("WORLD_SIZE", "1"))
  local = int(os.environ.get("LOCAL_RANK", "0"))
  if args.backend == "gloo":
    local %= max(torch.cuda.device_count(), 1)   # test mode: ranks may share a GPU
  if world > 1:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if args.backend == "nccl":
      dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
      dist.init_process_group("gloo")
  torch.cuda.set_device(local)
  dev = torch.device("cuda", local)
  red_dev = dev if args.backend == "nccl" else torch.device("cpu")   # where the max-time all-reduce lives

  if args.workload in ("gammatone", "lpc"):
    return side_workload(args, alz, torch, dev, rank, world, local, red_dev)
  C, N = args.channels, 1 << args.log2_samples
  if args.workload == "fir":
    if args.channels == 4096 and args.log2_samples == 20:   # configs[2] defaults
      C, N = 8192, 1 << 18
    b, a = fir_taps(), np.array([1.0])
    bank = alz.FilterBank([(b, a)], n_inputs=C, device=local)
    nsec = ([256], [1])
  else:
    # the global bank has world*C channels; this rank owns the contiguous shard [rank*C, (rank+1)*C)
    b_all, a_all = resonator_coefs(world * C)
    b, a = b_all[rank * C:(rank + 1) * C], a_all[rank * C:(rank + 1) * C]
    bank = alz.FilterBank([(b, a)], n_inputs=C, device=local)
    nsec = ([3], [3])
  if args.fused:
    bank.set_fused(True)
  bank.reset()

  shape = (N, C) if args.layout == "time" else (C, N)
  g = torch.Generator(device=dev).manual_seed(20260924 + rank)
  x = torch.empty(shape, dtype=torch.float64, device=dev)
  rows = shape[0]
  step_rows = max(1, rows // 16)
  for r0 in range(0, rows, step_rows):   # chunked fill keeps the generator's temporaries small
    x[r0:r0 + step_rows].uniform_(-1.0, 1.0, generator=g)
  y = torch.empty(shape, dtype=torch.float64, device=dev)

  def sync_all():
    torch.cuda.synchronize(dev)
    if world > 1:
      dist.barrier()
    torch.cuda.synchronize(dev)

  for _ in range(args.warmup):
    bank.process(x, layout=args.layout, out=y)
  ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        for _ in range(args.steps)]
  sync_all()
  t0 = time.perf_counter()
  for s in range(args.steps):
    ev[s][0].record()                  # same stream the kernel is launched on (torch's current)
    bank.process(x, layout=args.layout, out=y)
    ev[s][1].record()
  sync_all()
  elapsed = time.perf_counter() - t0
  kernel_ms = [e0.elapsed_time(e1) for e0, e1 in ev]
  kernel_name = bank.last_kernel

  if world > 1:
    t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

  # yardstick, outside the timed region: a plain device-to-device copy of the same block
  d2d_gbps = None
  if rank == 0:
    c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    y.copy_(x)
    c0.record()
    for _ in range(3):
      y.copy_(x)
    c1.record()
    torch.cuda.synchronize(dev)
    d2d_gbps = 3 * 2.0 * x.numel() * 8 / (c0.elapsed_time(c1) * 1e-3) / 1e9

  # spot parity: 4 channels of the last block against the oracle would need the whole
  # stream history; instead re-run a fresh 4096-sample block and compare bit for bit
  parity = None
  if rank == 0 and args.no_parity_check:
    parity = "skipped (--no-parity-check)"
  elif rank == 0:
    try:
      from oracle import oracle
      nchk = 4096
      bank.reset()
      xs = x[:nchk].contiguous() if args.layout == "time" else x[:, :nchk].contiguous()
      ys = bank.process(xs, layout=args.layout).cpu().numpy()
      ref = oracle.bank(nsec[0], nsec[1], b, a, xs.cpu().numpy(), layout=args.layout)
      if np.array_equal(ys.view(np.uint64), ref.view(np.uint64)):
        parity = "bit-exact"
      elif args.fused:
        ax = 0 if args.layout == "time" else 1
        nerr = float((np.abs(ys - ref).max(axis=ax) / np.abs(ref).max(axis=ax)).max())
        parity = "fused mode, not bit-exact: max normalised error %.3g (contract 1e-6)" % nerr
      else:
        parity = "MISMATCH"
    except Exception as exc:  # the oracle is a checker, never a dependency of the timed path
      parity = "unchecked (%s)" % exc

  if rank == 0:
    samples = float(world) * C * N * args.steps
    value = samples / elapsed / 1e9
    k_avg_ms = sum(kernel_ms) / len(kernel_ms)
    achieved = ALG_BYTES_PER_SAMPLE * C * N / (k_avg_ms * 1e-3) / 1e9
    roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": None,
            "kernel_ms_avg": k_avg_ms, "algorithmic_bytes_per_launch": ALG_BYTES_PER_SAMPLE * C * N,
            "d2d_copy_same_block_GBps": d2d_gbps}
    # HBM bytes per launch from the committed PMC passes (FETCH_SIZE / WRITE_SIZE cannot be read
    # from inside this process): measured traffic / algorithmic ratio of the same kernel at
    # 2**18-sample blocks, scaled to this launch's algorithmic bytes
    try:
      pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
      if args.workload == "biquad" and pmc["kernel"].split("<")[0] in kernel_name:
        roof["traffic"] = pmc["traffic_over_algorithmic"] * ALG_BYTES_PER_SAMPLE * C * N
        roof["traffic_source"] = "profiles/r01_pmc_traffic.json (rocprofv3 --pmc passes, x%.5f of algorithmic)" \
                                 % pmc["traffic_over_algorithmic"]
    except (OSError, KeyError, ValueError):
      pass
    workload = ("configs[1]: %d-channel biquad IIR bank (resonator.z_exp per channel), 48 kHz float64, "
                "%d-sample blocks, 1 MI355X per rank" % (C, N))
    metric = "Gsamples/s through ZFilter IIR biquad bank"
    if args.workload == "fir":
      # 256 mul + 255 add per output sample, unfused (bit-exact mode): FP64 vector issue is the
      # roof (78.6 TFLOP/s spec, MI355X_MICROARCH.md / SURVEY.md 8d), not HBM
      flops = 511.0 * C * N
      tf = flops / (k_avg_ms * 1e-3) / 1e12
      roof = {"bound": "valu_f64", "achieved": tf, "peak": 78.6, "unit": "TFLOP/s", "frac": tf / 78.6,
              "traffic": None, "kernel_ms_avg": k_avg_ms, "algorithmic_flops_per_launch": flops,
              "hbm_GBps": achieved,
              "note": "unfused mul + add halves the FMA spec (39.3); under this FP64 load the engine clock "
                      "sits at 2.0 GHz, not 2.4 (GRBM_GUI_ACTIVE), which puts the unfused roof at 32.8 "
                      "TFLOP/s with the VALUs 87 % busy: profiles/r01_pmc_fir.txt"}
      workload = ("configs[2]: 256-tap FIR lowpass (Hamming-windowed sinc, shared taps) x %d channels, "
                  "float64, %d-sample blocks, 1 MI355X per rank" % (C, N))
      metric = "Gsamples/s through ZFilter FIR-256 bank"
    out = {
      "metric": metric,
      "value": value, "unit": "Gsamples/s", "n_gpus": world, "steps": args.steps,
      "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
      "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
      "dtype": "f64", "data": "synthetic",
      "config": {"workload": workload,
                 "channels_per_gpu": C, "block_samples": N,
                 "layout": "time-major [N, C]" if args.layout == "time" else "channel-major [C, N]",
                 "kernel": kernel_name, "parity_spot_check": parity},
      "roofline": roof,
    }
    if not args.no_cpu_baseline and world == 1 and args.workload == "biquad":
      out["cpu_baseline"] = cpu_baseline(b, a, N)
    print(json.dumps(out))
  if world > 1:
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
  main()
