#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE.json's config.

Primary workload (configs[1]): a 4096-channel biquad IIR bank (one resonator per channel,
50 Hz .. 20 kHz log-spaced at 48 kHz), float64, 1 Mi-sample blocks, time-major [N, C] rows (the
reference's vector-valued-sample layout), one MI355X per rank.  A "step" is one pass of the bank
over one block of synthetic uniform(-1, 1) noise that is already resident in HBM; filter state
carries over from step to step, so K steps are one continuous K*N-sample stream per channel.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU; channels are sharded with no
   data-path collective.  Default "scaling": "weak" -- every rank owns a configs[1]-sized shard of
   an N x 4096-channel bank; --scaling strong divides configs[1]'s 4096 channels over the ranks,
   and the default run reports that figure too, as secondary.strong_scaling.)

Prints ONE JSON line on rank 0.  At N = 1 the line also carries
  "secondary":    configs[2] (FIR-256 x 8192, bit-exact and FMA mode), configs[3] (gammatone bank),
                  configs[4] (LPC frames) and the narrow-bank (512-channel) figure, each timed with
                  the same protocol and checked against the oracle,
  "cpu_baseline": the reference's own CPython path (generated DF-I generator fed by random.uniform
                  noise through a blocks(4096) consumer) on the host cores, next to the C port.
Exit status is non-zero when any parity check says MISMATCH.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
FP64_PEAK_TFLOPS = 78.6        # MI355X FP64 vector peak (FMA = 2 flop), same guide / SURVEY.md 8d
ALG_BYTES_PER_SAMPLE = 16.0    # 8 B read + 8 B written per channel-sample


# ---------------------------------------------------------------------------------------------
# workload definitions
# ---------------------------------------------------------------------------------------------
def resonator_coefs(C, fs=48000.0):
  """resonator.z_exp(fc, fc/10) per channel (reference lazy_filters.py:1245-1276), designed by
  audiolazy_amd's own host-side mirror of the reference's closed form."""
  from audiolazy_amd.filters import resonator
  fc = np.geomspace(50.0, 20000.0, C)
  filts = [resonator.z_exp(2 * np.pi * f / fs, 2 * np.pi * f / 10 / fs) for f in fc]
  b = np.array([fl.numlist for fl in filts], dtype=np.float64)
  a = np.array([fl.denlist for fl in filts], dtype=np.float64)
  return b, a


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


# ---------------------------------------------------------------------------------------------
# CPU baseline: the reference's path on the host cores (rank 0, N = 1, bounded sample)
# ---------------------------------------------------------------------------------------------
def reference_package():
  """The unmodified reference, when this machine has it: AUDIOLAZY_REF (a directory that holds the ``audiolazy``
  package) or /root/reference.  The GPU boxes do not have it; the CPU legs then run the restatement
  (oracle/pyref.py), and ``cpu_baseline.kind`` says which one was timed."""
  global _REF
  if _REF is None:
    _REF = False
    root = os.environ.get("AUDIOLAZY_REF") or "/root/reference"
    if os.path.isdir(os.path.join(root, "audiolazy")):
      sys.dont_write_bytecode = True            # (/root/reference is read-only)
      sys.path.insert(0, root)
      try:
        import audiolazy
        _REF = audiolazy
      except Exception:
        _REF = False
      finally:
        sys.path.remove(root)
  return _REF or None


_REF = None


def _py_channel(job):
  """One process's share of the reference path: white_noise -> filt -> .blocks(4096)."""
  b, a, n, seed, passes = job
  ref = reference_package()
  t0 = time.perf_counter()
  done = 0
  if ref is not None:                            # the reference itself: ZFilter(b, a)(white_noise(n)).blocks(4096)
    import random
    filt = ref.ZFilter(list(b), list(a))
    for p in range(passes):
      random.seed(seed + p)
      for blk in filt(ref.white_noise(n)).blocks(4096):
        done += len(blk)
  else:
    from oracle import pyref
    for p in range(passes):
      done += pyref.consume_blocks(pyref.df1(b, a, pyref.noise(n, seed + p)), 4096)
  return done, time.perf_counter() - t0


def usable_cores():
  """Cores this process may actually run on: affinity mask and cgroup CPU quota (a container can
  show 256 logical CPUs in os.cpu_count() and be allowed a handful)."""
  n = float(len(os.sched_getaffinity(0))) if hasattr(os, "sched_getaffinity") else float(os.cpu_count() or 1)
  try:
    quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
    if quota != "max":
      n = min(n, float(quota) / float(period))
  except (OSError, ValueError):
    try:
      quota = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
      period = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
      if quota > 0:
        n = min(n, quota / period)
    except (OSError, ValueError):
      pass
  return round(n, 2)


def cpu_baseline(b, a, budget_s=3.5):
  """Four figures, each on a bounded sample of configs[1] (about `budget_s` seconds apiece):
    py_1proc   the reference's per-sample path, one process, one channel at a time;
    py_pool    the same in multiprocessing.Pool(os.cpu_count()), independent seeded channels;
    py_rows    the reference's vector-valued-sample idiom: items are ndarray rows of all C channels,
               coefficients are repeat(ndarray) series (tests/test_filters_extdep.py:49-89) -- the
               strongest number the unmodified reference can produce;
    c_port     oracle/alz_oracle.c (the same DF-I statement compiled with gcc), 1 thread.
  Must run BEFORE torch / HIP are initialised in this process (the pool forks)."""
  import itertools
  import multiprocessing
  from oracle import oracle, pyref
  C = b.shape[0]
  cores = os.cpu_count() or 1
  usable = usable_cores()
  sel = lambda i: ([float(v) for v in b[i]], [float(v) for v in a[i]])
  legs = {}

  # (i) one process
  n1 = 1 << 19
  bb, aa = sel(C // 2)
  _py_channel((bb, aa, 1 << 14, 1, 1))                      # warm (imports, exec)
  done, el = 0, 0.0
  while el < budget_s:
    d, e = _py_channel((bb, aa, n1, 1234 + done, 1))
    done, el = done + d, el + e
  legs["py_1proc"] = {"value": done / el / 1e9, "unit": "Gsamples/s", "cores": 1,
                      "sample": "1 channel at a time x %d samples, %d passes" % (n1, done // n1)}

  # (ii) every core, one channel per process per pass
  # one process per core this container may actually use (os.cpu_count() can be far above the
  # cgroup allowance: 256 processes on 16 allowed cores measured 40 Msamples/s, less than half of what
  # 16 processes reach)
  procs = max(1, min(cores, int(usable + 0.5)))
  per = max(1 << 16, int(legs["py_1proc"]["value"] * 1e9 * budget_s) // (1 << 16) * (1 << 16))
  jobs = [sel(int(i * (C - 1) / max(procs - 1, 1))) + (per, 99 + i, 1) for i in range(procs)]
  ctx = multiprocessing.get_context("fork")
  with ctx.Pool(procs) as pool:
    pool.map(_py_channel, [(bb, aa, 1 << 10, 7, 1)] * procs)   # start every worker
    t0 = time.perf_counter()
    res = pool.map(_py_channel, jobs, chunksize=1)
    el = time.perf_counter() - t0
  tot = sum(r[0] for r in res)
  legs["py_pool"] = {"value": tot / el / 1e9, "unit": "Gsamples/s", "cores": procs,
                     "per_process_Msamples_s": float(np.mean([r[0] / r[1] for r in res]) / 1e6),
                     "sample": "multiprocessing.Pool(%d) x 1 channel x %d samples per process (os.cpu_count() = %d, "
                               "CPU allowance of this container by cgroup quota / affinity = %s cores)"
                               % (procs, per, cores, usable)}

  # (iii) rows of all channels as samples, per-channel coefficients as repeat(ndarray) series
  rows = 2048
  rng = np.random.default_rng(20260924)
  x = rng.uniform(-1, 1, (rows, C))
  rep = itertools.repeat
  nz = lambda col: not np.all(col == 0)
  bs = [rep(np.ascontiguousarray(b[:, k])) if nz(b[:, k]) else 0.0 for k in range(b.shape[1])]
  as_ = [1.0] + [rep(np.ascontiguousarray(a[:, k])) if nz(a[:, k]) else 0.0 for k in range(1, a.shape[1])]
  zero = np.zeros(C)
  ref = reference_package()
  if ref is not None:
    zf = ref.z
    num = sum((bk if isinstance(bk, float) else ref.Stream(bk)) * zf ** -k for k, bk in enumerate(bs) if not isinstance(bk, float) or bk != 0.0)
    den = 1 + sum((ak if isinstance(ak, float) else ref.Stream(ak)) * zf ** -k for k, ak in enumerate(as_) if k > 0 and (not isinstance(ak, float) or ak != 0.0))
    ref_filt = num / den
  done, t0 = 0, time.perf_counter()
  while True:
    if ref is not None:                           # tests/test_filters_extdep.py:49-89 idiom on the reference itself
      for blk in ref_filt(iter(x), zero=zero).blocks(4096):
        done += len(blk) * C
    else:
      done += pyref.consume_blocks(pyref.df1(bs, as_, iter(x), zero=zero), 4096) * C
    el = time.perf_counter() - t0
    if el >= budget_s:
      break
  legs["py_rows"] = {"value": done / el / 1e9, "unit": "Gsamples/s", "cores": 1,
                     "sample": "%d-channel ndarray rows x %d rows, %d passes (NumPy elementwise per sample)"
                               % (C, rows, done // (rows * C))}

  # (iv) the C port
  Cc, n = 64, 1 << 18
  xs = rng.uniform(-1, 1, (Cc, n))
  pick = np.linspace(0, C - 1, Cc).astype(int)
  bsel, asel = np.ascontiguousarray(b[pick]), np.ascontiguousarray(a[pick])
  oracle.bank([3], [3], bsel, asel, xs[:, :1024], layout="chan")
  done, t0 = 0, time.perf_counter()
  while True:
    oracle.bank([3], [3], bsel, asel, xs, layout="chan")
    done += Cc * n
    el = time.perf_counter() - t0
    if el >= budget_s:
      break
  legs["c_port"] = {"value": done / el / 1e9, "unit": "Gsamples/s", "cores": 1,
                    "sample": "oracle/alz_oracle.c, %d of the %d channels x %d samples, %d passes"
                              % (Cc, C, n, done // (Cc * n))}
  # (v) the reference's comb: D memory variables shifted per sample (lazy_filters.py:254-255), one process
  D = 441
  cb, ca = [1.0], [1.0] + [0.0] * (D - 1) + [-math.e ** (-D / 20000.0)]
  _py_channel((cb, ca, 1 << 8, 1, 1))
  done, el = 0, 0.0
  while el < budget_s:
    d, e = _py_channel((cb, ca, 1 << 13, 4321 + done, 1))
    done, el = done + d, el + e
  legs["py_comb_1proc"] = {"value": done / el / 1e9, "unit": "Gsamples/s", "cores": 1,
                           "sample": "comb.tau(441, 20000): 1 channel x %d samples, %d passes (the generated loop shifts %d memory "
                                     "variables per sample)" % (1 << 13, done // (1 << 13), D)}
  head = legs["py_pool"]
  return {"value": head["value"], "unit": "Gsamples/s", "cores": head["cores"], "host_logical_cpus": cores,
          "usable_cores": usable, "kind": "reference" if reference_package() is not None else "port",
          "sample": ("the unmodified reference (audiolazy %s: ZFilter(b, a)(white_noise(n)).blocks(4096)) "
                     % getattr(reference_package(), "__version__", "?") if reference_package() is not None else
                     "the reference's CPython path restated (oracle/pyref.py: the generated DF-I generator of "
                     "lazy_filters.py:197-260 executed by this interpreter, fed by random.uniform noise, consumed "
                     "through blocks(4096)) ") + "on resonators of configs[1]: " + head["sample"]
                    + "; legs = one process / all cores / vector-valued rows / C port (the C port is always oracle/alz_oracle.c)",
          "legs": legs}


# ---------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` starts its own N ranks
# ---------------------------------------------------------------------------------------------
def free_port():
  import socket
  s = socket.socket()
  s.bind(("127.0.0.1", 0))
  port = s.getsockname()[1]
  s.close()
  return port


def relaunch(n_ranks):
  """Re-run this very command line as ``n_ranks`` processes, one per GPU, under torch.distributed.run
  (what the driver's own N > 1 command does); the ranks' stdout / stderr / exit status pass through."""
  import subprocess
  env = dict(os.environ, MASTER_ADDR="127.0.0.1")
  env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL needs it on these hosts
  env.setdefault("OMP_NUM_THREADS", "1")
  cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks),
         "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
  return subprocess.call(cmd, env=env)


def launch_check(args, rank, world):
  """--launch-check: the rank start-up and the three collectives the timing protocol uses (barrier, MAX
  all-reduce, all_gather of the per-rank figures) and nothing else -- no device, no kernels.  Lets the
  N > 1 launcher be tested on a machine without GPUs (gloo)."""
  import torch
  import torch.distributed as dist
  os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
  if "MASTER_PORT" not in os.environ:
    os.environ["MASTER_PORT"] = str(free_port())
  if args.backend == "nccl":
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    dev = torch.device("cuda", local)
  else:
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
  dist.barrier()
  t = torch.tensor([float(rank + 1)], dtype=torch.float64, device=dev)
  dist.all_reduce(t, op=dist.ReduceOp.MAX)
  parts = [torch.empty_like(t) for _ in range(world)]
  dist.all_gather(parts, torch.tensor([float(rank)], dtype=torch.float64, device=dev))
  ok = float(t.item()) == float(world) and [float(p.item()) for p in parts] == [float(r) for r in range(world)]
  dist.barrier()
  dist.destroy_process_group()
  if rank == 0:
    print(json.dumps({"launch_check": "ok" if ok else "MISMATCH", "n_gpus": world, "requested_gpus": args.gpus,
                      "backend": args.backend, "launcher": "torch.distributed.run" if os.environ.get("TORCHELASTIC_RUN_ID")
                      else "none"}))
  return 0 if ok else 3


# ---------------------------------------------------------------------------------------------
# GPU side
# ---------------------------------------------------------------------------------------------
class Ctx(object):
  """Process-group / device context shared by the workloads."""

  def __init__(self, args):
    import torch
    self.torch = torch
    self.rank = int(os.environ.get("RANK", "0"))
    self.world = int(os.environ.get("WORLD_SIZE", "1"))
    self.local = int(os.environ.get("LOCAL_RANK", "0"))
    self.dist = None
    self.local_elapsed = []                    # per ctx.timed() call; [0] is the headline workload
    self.robust = False                        # secondary workloads: timed_batches() instead of the contract's K steps
    self.last_stats = None
    if args.backend == "gloo":
      self.local %= max(torch.cuda.device_count(), 1)   # test mode: ranks may share a GPU
    elif self.world > torch.cuda.device_count():
      # RCCL refuses two ranks on one device: an N > device count run cannot produce a valid line
      sys.exit("bench.py: %d ranks on the nccl (RCCL) backend need %d GPUs, this node shows %d -- refusing "
               "(use --backend gloo for the shared-GPU test mode)" % (self.world, self.world, torch.cuda.device_count()))
    if self.world > 1 or args.init_dist:
      import torch.distributed as dist
      os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
      if "MASTER_PORT" not in os.environ:      # --init-dist without a launcher: a one-rank group of our own
        os.environ["MASTER_PORT"] = str(free_port())
      if args.backend == "nccl":               # "nccl" IS RCCL on ROCm
        dist.init_process_group("nccl", rank=self.rank, world_size=self.world,
                                device_id=torch.device("cuda", self.local))
      else:
        dist.init_process_group("gloo", rank=self.rank, world_size=self.world)
      self.dist = dist
    torch.cuda.set_device(self.local)
    self.dev = torch.device("cuda", self.local)
    self.red_dev = self.dev if args.backend == "nccl" else torch.device("cpu")

  def sync_all(self):
    self.torch.cuda.synchronize(self.dev)
    if self.dist is not None:
      self.dist.barrier()
    self.torch.cuda.synchronize(self.dev)

  def all_ranks(self, values):
    """[values of rank 0, values of rank 1, ...] on every rank (one all_gather of a small float64 tensor)."""
    if self.dist is None:
      return [list(values)]
    t = self.torch.tensor(list(values), dtype=self.torch.float64, device=self.red_dev)
    parts = [self.torch.empty_like(t) for _ in range(self.world)]
    self.dist.all_gather(parts, t)
    return [[float(v) for v in p.tolist()] for p in parts]

  def timed_batches(self, step, steps, warmup):
    """The secondary workloads' timer (round 5).  A 20-step region of an 80 us kernel is 1.6 ms: on a fresh box the
    driver's record of `lpc_bit_identical` came out at 0.13 ms per step two rounds running where every run behind a
    test suite measured 0.08 (profiles/NOTES_r05.md has what was found).  So: W warm-up steps, a
    calibration run of K steps, then 3 - 5 batches of max(K, 20 ms worth of) steps, each bracketed like the contract's
    region (barrier + synchronize, one pair of HIP events); the MEDIAN batch is reported, min and max beside it.
    Returns (median seconds per step x K, the median batch's per-step device time in ms)."""
    torch = self.torch
    for _ in range(warmup):
      step()
    self.sync_all()
    t0 = time.perf_counter()
    for _ in range(steps):
      step()
    self.sync_all()
    est = max((time.perf_counter() - t0) / steps, 1e-6)
    per_batch = min(max(steps, int(math.ceil(0.020 / est))), 5000)
    batches = 5 if per_batch * est < 0.1 else 3
    rows = []
    for _ in range(batches):
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      self.sync_all()
      t0 = time.perf_counter()
      e0.record()
      for _ in range(per_batch):
        step()
      e1.record()
      self.sync_all()
      rows.append(((time.perf_counter() - t0) / per_batch, e0.elapsed_time(e1) / per_batch))
    rows.sort()
    med = rows[len(rows) // 2]
    self.last_stats = {"batches": batches, "steps_per_batch": per_batch, "calibration_ms_per_step": est * 1e3,
                       "ms_per_step_min": rows[0][0] * 1e3, "ms_per_step_median": med[0] * 1e3,
                       "ms_per_step_max": rows[-1][0] * 1e3,
                       "kernel_ms_min": min(r[1] for r in rows), "kernel_ms_max": max(r[1] for r in rows)}
    elapsed = med[0] * steps
    self.local_elapsed.append(elapsed)
    return elapsed, med[1]

  def timed(self, step, steps, warmup):
    """W untimed steps, then exactly K steps bracketed by barrier + synchronize; MAX over ranks.
    Returns (elapsed seconds, mean per-step device time in ms from HIP events on the launch stream)."""
    if self.robust and self.dist is None:
      return self.timed_batches(step, steps, warmup)
    self.last_stats = None
    torch = self.torch
    for _ in range(warmup):
      step()
    # ONE pair of HIP events brackets the K steps on the launch stream (torch's current stream): device time per
    # step = their distance / K.  (A pair per step, as in rounds 1 - 2, puts two marker packets between
    # consecutive launches: +12 us on a 60 us LPC step, tools/lpc_time.py.)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    self.sync_all()
    t0 = time.perf_counter()
    e0.record()
    for s in range(steps):
      step()
    e1.record()
    self.sync_all()
    elapsed = time.perf_counter() - t0
    self.local_elapsed.append(elapsed)         # this rank's own clock (reported per rank next to the MAX)
    if self.dist is not None:
      t = torch.tensor([elapsed], dtype=torch.float64, device=self.red_dev)
      self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
      elapsed = float(t.item())
    return elapsed, e0.elapsed_time(e1) / steps

  def noise(self, shape, seed_off=0):
    torch = self.torch
    g = torch.Generator(device=self.dev).manual_seed(20260924 + self.rank + 1000 * seed_off)
    x = torch.empty(shape, dtype=torch.float64, device=self.dev)
    rows = shape[0]
    step_rows = max(1, rows // 16)
    for r0 in range(0, rows, step_rows):   # chunked fill keeps the generator's temporaries small
      x[r0:r0 + step_rows].uniform_(-1.0, 1.0, generator=g)
    return x


def bits_equal(a, b):
  return a.shape == b.shape and np.array_equal(a.view(np.uint64), b.view(np.uint64))


def norm_err(got, ref, axis):
  """max over channels of max_n |got - ref| / max_n |ref| (SURVEY.md 8d's error definition)."""
  den = np.abs(ref).max(axis=axis)
  den[den == 0] = 1.0
  return float((np.abs(got - ref).max(axis=axis) / den).max())


def hbm_roof(alg_bytes, k_ms):
  ach = alg_bytes / (k_ms * 1e-3) / 1e9
  return {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
          "traffic": None, "kernel_ms_avg": k_ms, "algorithmic_bytes_per_launch": alg_bytes}


def wl_biquad(ctx, args, alz, C, N, c_lo, c_total, steps, warmup, check=True, time_parallel=None, layout=None):
  """Channels [c_lo, c_lo + C) of a c_total-channel resonator bank on this rank."""
  torch = ctx.torch
  lay = layout or args.layout
  b_all, a_all = resonator_coefs(c_total)
  b, a = b_all[c_lo:c_lo + C], a_all[c_lo:c_lo + C]
  bank = alz.FilterBank([(b, a)], n_inputs=C, device=ctx.local)
  if args.fused:
    bank.set_fused(True)
  if time_parallel is not None:
    bank.set_time_parallel(True if time_parallel == 1 else "one-pass" if time_parallel == -2 else time_parallel)
  bank.reset()
  shape = (N, C) if lay == "time" else (C, N)
  x = ctx.noise(shape)
  y = x if getattr(args, "in_place", False) else torch.empty(shape, dtype=torch.float64, device=ctx.dev)    # (--in-place: y = x, measurement runs only)
  elapsed, k_ms = ctx.timed(lambda: bank.process(x, layout=lay, out=y), steps, warmup)
  kernel = bank.last_kernel
  exact = not args.fused and not (time_parallel and ("k_scan" in kernel or "k_look" in kernel))
  parity = "skipped (--no-parity-check)"
  if check and ctx.rank == 0 and not args.no_parity_check:
    # the WHOLE bank width on a fresh stream, against the oracle
    from oracle import oracle
    nchk = min(N, args.parity_samples)
    bank.reset()
    xs = x[:nchk].contiguous() if lay == "time" else x[:, :nchk].contiguous()
    got = bank.process(xs, layout=lay).cpu().numpy()
    ref = oracle.bank([3], [3], b, a, xs.cpu().numpy(), layout=lay)
    if bits_equal(got, ref):
      parity = "bit-exact vs oracle, %d channels x %d samples" % (C, nchk)
    elif not exact:
      err = norm_err(got, ref, 0 if lay == "time" else 1)
      # the reference's own per-sample test (almost_eq, lazy_misc.py:264-267: |a - b| <= 2**-23 |a + b|)
      close = float(np.mean(np.abs(got - ref) <= 2.0 ** -23 * np.abs(got + ref)))
      parity = ("not bit-exact by design (%s): max normalised error %.3g vs oracle, %d channels x %d samples "
                "(contract 1e-6); %.6f of the samples pass the reference's almost_eq"
                % ("FMA mode" if args.fused else "time-parallel mode", err, C, nchk, close))
      if not err <= 1e-6:
        parity = "MISMATCH: " + parity
    else:
      parity = "MISMATCH"
    # ... and the WHOLE block length on 64 strided channels (every tile and chunk boundary a 2^20-sample block
    # walks; the C oracle does 64 channels x 2^20 samples in ~0.4 s)
    if N > nchk and not parity.startswith("MISMATCH"):
      bank.reset()
      bank.process(x, layout=lay, out=y)
      pick = np.unique(np.linspace(0, C - 1, min(C, 64)).astype(int))
      idx = torch.from_numpy(pick).to(ctx.dev)
      tm = lay == "time"
      got = (y.index_select(1, idx) if tm else y.index_select(0, idx)).cpu().numpy()
      xs = (x.index_select(1, idx) if tm else x.index_select(0, idx)).cpu().numpy()
      ref = oracle.bank([3], [3], np.ascontiguousarray(b[pick]), np.ascontiguousarray(a[pick]), xs, layout=lay)
      if bits_equal(got, ref):
        parity += "; full block length: %d strided channels x %d samples bit-exact" % (len(pick), N)
      elif not exact:
        err = norm_err(got, ref, 0 if tm else 1)
        parity += "; full block length (%d strided channels x %d samples): max normalised error %.3g" % (len(pick), N, err)
        if not err <= 1e-6:
          parity = "MISMATCH: " + parity
      else:
        parity = "MISMATCH on the full block length: " + parity
      del got, xs, ref
  # yardstick, outside the timed region: a plain device-to-device copy of the same block
  d2d = None
  if ctx.rank == 0 and check:
    c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    y.copy_(x)
    c0.record()
    for _ in range(3):
      y.copy_(x)
    c1.record()
    torch.cuda.synchronize(ctx.dev)
    d2d = 3 * 2.0 * x.numel() * 8 / (c0.elapsed_time(c1) * 1e-3) / 1e9
  del x, y, bank
  torch.cuda.empty_cache()
  roof = hbm_roof(ALG_BYTES_PER_SAMPLE * C * N, k_ms)
  if d2d is not None:
    roof["d2d_copy_same_block_GBps"] = d2d
  return {"units": float(C) * N, "elapsed": elapsed, "timing": ctx.last_stats, "kernel": kernel, "parity": parity, "roofline": roof,
          "C": C, "N": N}


_REF_CACHE = {}


def wl_fir(ctx, args, alz, C, N, steps, warmup, fused):
  torch = ctx.torch
  taps = fir_taps()
  bank = alz.FilterBank([(taps, np.array([1.0]))], n_inputs=C, device=ctx.local)
  if fused:
    bank.set_fused(True)
  bank.reset()
  x = ctx.noise((N, C), 1)
  y = torch.empty((N, C), dtype=torch.float64, device=ctx.dev)
  elapsed, k_ms = ctx.timed(lambda: bank.process(x, layout="time", out=y), steps, warmup)
  kernel = bank.last_kernel
  parity = "skipped (--no-parity-check)"
  if ctx.rank == 0 and not args.no_parity_check:
    from oracle import oracle
    nchk = min(N, 512)
    bank.reset()
    xs = x[:nchk].contiguous()
    got = bank.process(xs, layout="time").cpu().numpy()
    ref = oracle.bank([256], [1], taps, np.ones(1), xs.cpu().numpy(), layout="time")
    if bits_equal(got, ref):
      parity = "bit-exact vs oracle, %d channels x %d samples" % (C, nchk)
    elif fused:
      err = norm_err(got, ref, 0)
      parity = "FMA mode, not bit-exact by design: max normalised error %.3g vs oracle (contract 1e-6)" % err
      if not err <= 1e-6:
        parity = "MISMATCH: " + parity
    else:
      parity = "MISMATCH"
    # ... and the benchmarked EXTENT: the whole block (every run and grid block of k_fir_ring) on 64 strided channels
    if N > nchk and not parity.startswith("MISMATCH"):
      bank.reset()
      bank.process(x, layout="time", out=y)
      pick = np.unique(np.linspace(0, C - 1, min(C, 64)).astype(int))
      idx = torch.from_numpy(pick).to(ctx.dev)
      got = y.index_select(1, idx).cpu().numpy()
      key = ("fir", C, N)
      if key not in _REF_CACHE:        # (both modes filter the same seeded block: one oracle run serves them)
        _REF_CACHE.clear()
        _REF_CACHE[key] = oracle.bank([256], [1], taps, np.ones(1), x.index_select(1, idx).cpu().numpy(), layout="time")
      ref = _REF_CACHE[key]
      if bits_equal(got, ref):
        parity += "; full extent: %d strided channels x %d samples bit-exact" % (len(pick), N)
      elif fused:
        err = norm_err(got, ref, 0)
        parity += "; full extent (%d strided channels x %d samples): max normalised error %.3g" % (len(pick), N, err)
        if not err <= 1e-6:
          parity = "MISMATCH: " + parity
      else:
        parity = "MISMATCH on the full extent: " + parity
      del got
  del x, y, bank
  torch.cuda.empty_cache()
  flops = 511.0 * C * N   # 256 mul + 255 add per output sample (an FMA counts as its two operations)
  tf = flops / (k_ms * 1e-3) / 1e12
  roof = {"bound": "valu_f64", "achieved": tf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
          "frac": tf / FP64_PEAK_TFLOPS, "traffic": None, "kernel_ms_avg": k_ms,
          "algorithmic_flops_per_launch": flops, "algorithmic_bytes_per_launch": ALG_BYTES_PER_SAMPLE * C * N,
          "hbm_GBps": ALG_BYTES_PER_SAMPLE * C * N / (k_ms * 1e-3) / 1e9}
  if not fused:
    roof["note"] = ("bit-exact mode: separately rounded v_mul_f64 + v_add_f64, two instructions per tap, so at "
                    "most half of the FMA peak by construction (39.3 TFLOP/s)")
  return {"units": float(C) * N, "elapsed": elapsed, "timing": ctx.last_stats, "kernel": kernel, "parity": parity, "roofline": roof,
          "C": C, "N": N}


def wl_gammatone(ctx, args, alz, steps, warmup, fused=False, streams=64, log2n=16, time_parallel=False, layout="chan"):
  torch = ctx.torch
  B, S, N = 256, streams, 1 << log2n    # 512 streams over 8 GPUs -> 64 streams per GPU, all bands local
  s_, Hz = alz.sHz(48000)
  fcs = [f * Hz for f in alz.erb_space(50., 20000., B)]
  bank = alz.gammatone_bank(fcs, S, strategy="slaney", Hz=Hz, device=ctx.local)
  if fused:
    bank.set_fused(True)
  if time_parallel:
    bank.set_time_parallel(True)
  bank.reset()
  tm = layout == "time"
  x = ctx.noise((N, S) if tm else (S, N), 2)
  y = torch.empty((N, B * S) if tm else (B * S, N), dtype=torch.float64, device=ctx.dev)
  elapsed, k_ms = ctx.timed(lambda: bank.process(x, layout=layout, out=y), steps, warmup)
  kernel = bank.last_kernel
  parity = "skipped (--no-parity-check)"
  if ctx.rank == 0 and not args.no_parity_check:
    from oracle import oracle
    nchk = min(N, 512 if S > 1 else 16384)
    bank.reset()
    xs = x[:nchk].t().contiguous() if tm else x[:, :nchk].contiguous()          # [S, nchk]
    got = bank.process(xs.t().contiguous() if tm else xs, layout=layout).cpu().numpy()
    if tm:
      got = np.ascontiguousarray(got.T)
    k = alz.gammatone_erb_constants(4)[0]
    bands = [alz.gammatone.slaney(fc, k * alz.erb(fc, Hz)) for fc in fcs]
    nbs, nas = [len(f.numlist) for f in bands[0]], [len(f.denlist) for f in bands[0]]
    bcat = np.repeat(np.array([sum((f.numlist for f in band), []) for band in bands]), S, axis=0)
    acat = np.repeat(np.array([sum((f.denlist for f in band), []) for band in bands]), S, axis=0)
    ref = oracle.bank(nbs, nas, bcat, acat, np.tile(xs.cpu().numpy(), (B, 1)), layout="chan")
    if bits_equal(got, ref):
      parity = "bit-exact vs oracle, %d bands x %d streams x %d samples" % (B, S, nchk)
    elif fused or time_parallel:
      err = norm_err(got, ref, 1)
      parity = "%s mode, not bit-exact by design: max normalised error %.3g vs oracle (contract 1e-6)" % (
          "FMA" if fused else "time-parallel", err)
      if not err <= 1e-6:
        parity = "MISMATCH: " + parity
    else:
      parity = "MISMATCH"
  del x, y, bank
  torch.cuda.empty_cache()
  return {"units": float(B) * S * N, "elapsed": elapsed, "timing": ctx.last_stats, "kernel": kernel, "parity": parity,
          "roofline": hbm_roof((8.0 + 8.0 / B) * B * S * N, k_ms), "B": B, "S": S, "N": N, "layout": layout}


def wl_lpc(ctx, args, alz, steps, warmup, fused=False, exact=False, frames=65536):
  from audiolazy_amd.lpc import kautocor_frames
  F, L, order = frames, 480, 16
  sig = ctx.noise((F * L,), 3)
  torch = ctx.torch
  out = (torch.empty((F, order + 1), dtype=torch.float64, device=ctx.dev), torch.empty((F,), dtype=torch.float64, device=ctx.dev),
         torch.empty((F,), dtype=torch.int32, device=ctx.dev))          # results land in the same tensors every step
  elapsed, k_ms = ctx.timed(lambda: kautocor_frames(sig, L, order, fused=fused, exact=exact, out=out), steps, warmup)
  kernel = alz.last_kernel()                       # read from the library (alz_last_kernel), not assumed
  parity = "skipped (--no-parity-check)"
  if ctx.rank == 0 and not args.no_parity_check:
    from oracle import oracle
    # frames compared with the oracle: all of a 65 536-frame batch (4096 of them for the default recursion); of a
    # larger batch every 16th frame ACROSS THE WHOLE BATCH plus the last 4096 (every workgroup range, the chunk
    # ring's later rounds), taken from the results of the timed launches themselves
    if F > 65536:
      frames = np.unique(np.r_[0:F:16, F - 4096:F])
    else:
      frames = np.arange(F if (fused or exact) else min(F, 4096))
    idx = torch.from_numpy(frames).to(ctx.dev)
    coefs, err, status = (t.index_select(0, idx).cpu().numpy() for t in out)
    blk = sig.reshape(F, L).index_select(0, idx).cpu().numpy()
    rc, re, rs = oracle.kautocor_frames(blk.reshape(-1), len(frames), L, L, order)
    worst = float(np.max(np.abs(coefs - rc) / np.maximum(1.0, np.abs(rc))))
    what = "%d frames%s" % (len(frames), " (every 16th of %d + the last 4096)" % F if F > 65536 else "")
    if exact:
      ok = bits_equal(coefs, rc) and bits_equal(err, re) and np.array_equal(status, rs)
      parity = "bit-exact vs oracle: coefficients and error of %s" % what if ok else "MISMATCH (%.3g)" % worst
    else:
      ok = worst <= 1e-9 and np.array_equal(status, rs)
      parity = ("%s: coefficients within %.1e of the oracle (this recursion is not bit-pinned; contract 1e-6)"
                % (what, worst)) if ok else "MISMATCH (%.3g)" % worst
  del sig
  ctx.torch.cuda.empty_cache()
  return {"units": float(F), "elapsed": elapsed, "timing": ctx.last_stats, "parity": parity, "kernel": kernel,
          "roofline": hbm_roof(3984.0 * F, k_ms), "F": F}


def wl_envelope(ctx, args, alz, C, N, steps, warmup):
  """envelope.abs for a whole bank: lowpass.pole(cutoff)(abs(x)) per channel with |x|
  fused into the kernel's input reads -- the elementwise stage must not cost a pass over the block."""
  torch = ctx.torch
  cut = np.geomspace(2 * np.pi * 5 / 48000., 2 * np.pi * 200 / 48000., C)
  filts = [alz.lowpass(float(c)) for c in cut]
  b, a = np.array([f.numlist for f in filts]), np.array([f.denlist for f in filts])
  bank = alz.FilterBank([(b, a)], n_inputs=C, device=ctx.local).set_input_map("abs")
  bank.reset()
  x = ctx.noise((N, C), 4)
  y = x if getattr(args, "in_place", False) else torch.empty((N, C), dtype=torch.float64, device=ctx.dev)
  elapsed, k_ms = ctx.timed(lambda: bank.process(x, layout="time", out=y), steps, warmup)
  kernel = bank.last_kernel + " (|x| fused into the input reads)"
  parity = "skipped (--no-parity-check)"
  if ctx.rank == 0 and not args.no_parity_check:
    from oracle import oracle
    nchk = min(N, args.parity_samples)
    bank.reset()
    xs = x[:nchk].contiguous()
    got = bank.process(xs, layout="time").cpu().numpy()
    ref = oracle.bank([1], [2], b, a, np.abs(xs.cpu().numpy()), layout="time")
    parity = "bit-exact vs oracle, %d channels x %d samples" % (C, nchk) if bits_equal(got, ref) else "MISMATCH"
  del x, y, bank
  torch.cuda.empty_cache()
  return {"units": float(C) * N, "elapsed": elapsed, "timing": ctx.last_stats, "kernel": kernel, "parity": parity,
          "roofline": hbm_roof(ALG_BYTES_PER_SAMPLE * C * N, k_ms)}


def comb_coefs(C, D, linearized=False, fs=48000.0):
  """comb.tau-style feedback combs (lazy_filters.py:1118-1147): y[n] = x[n] + alpha y[n - D], alpha = e ** (-D / tau) with
  the decay time tau spread over 0.05 - 2 s per channel; `linearized`: the delay D + frac as the reference's linearize()
  leaves it (lazy_filters.py:339-373) -- two adjacent taps alpha (1 - frac) z^-D and alpha frac z^-(D+1)."""
  tau = np.geomspace(0.05 * fs, 2.0 * fs, C)
  alpha = np.array([math.e ** (-D / t) for t in tau])
  a = np.zeros((C, D + 2 if linearized else D + 1))
  a[:, 0] = 1.0
  if linearized:
    frac = np.linspace(0.1, 0.9, C)
    a[:, D] = -(alpha * (1.0 - frac))
    a[:, D + 1] = -(alpha * frac)
  else:
    a[:, D] = -alpha
  return np.ones((C, 1)), a


def wl_comb(ctx, args, alz, C, N, D, steps, warmup, layout="time", linearized=False, inplace=False):
  """A bank of feedback combs / Karplus-Strong strings (north_star's "resonator/comb"): the step kernels
  of csrc/alz_comb.hip.  16 algorithmic bytes per channel-sample, HBM-bound for a wide bank; ONE string is bound by the
  serial hand-over from one period of the delay line to the next."""
  torch = ctx.torch
  b, a = comb_coefs(C, D, linearized)
  na = a.shape[1]
  bank = alz.FilterBank([(b, a)], n_inputs=C, device=ctx.local)
  bank.reset()
  tm = layout == "time"
  x = ctx.noise((N, C) if tm else (C, N), 7)
  y = x if inplace else torch.empty_like(x)
  elapsed, k_ms = ctx.timed(lambda: bank.process(x, layout=layout, out=y), steps, warmup)
  kernel = bank.last_kernel
  parity = "skipped (--no-parity-check)"
  if ctx.rank == 0 and not args.no_parity_check:
    from oracle import oracle
    # full width on a short block (several periods of the delay line), then the whole length on a few strided channels
    nchk = min(N, 4 * D + 100)
    xs = ctx.noise((nchk, C) if tm else (C, nchk), 8)
    bank.reset()
    got = bank.process(xs, layout=layout).cpu().numpy()
    ref = oracle.bank([1], [na], b, a, xs.cpu().numpy(), layout=layout)
    parity = "bit-exact vs oracle, %d channels x %d samples" % (C, nchk) if bits_equal(got, ref) else "MISMATCH"
    if N > nchk and not parity.startswith("MISMATCH"):
      xl = ctx.noise((N, C) if tm else (C, N), 7)         # (the timed block again: an in-place run has overwritten it)
      bank.reset()
      yl = bank.process(xl, layout=layout)
      pick = np.unique(np.linspace(0, C - 1, min(C, 8)).astype(int))
      idx = torch.from_numpy(pick).to(ctx.dev)
      got = yl.index_select(1 if tm else 0, idx).cpu().numpy()
      ref = oracle.bank([1], [na], b[pick], a[pick], xl.index_select(1 if tm else 0, idx).cpu().numpy(), layout=layout)
      parity += ("; full block length: %d strided channels x %d samples bit-exact" % (len(pick), N) if bits_equal(got, ref)
                 else "; MISMATCH on the full block length")
      if "MISMATCH" in parity:
        parity = "MISMATCH: " + parity
      del xl, yl
  del x, y, bank
  torch.cuda.empty_cache()
  roof = hbm_roof(ALG_BYTES_PER_SAMPLE * C * N, k_ms)
  if C == 1:
    roof["note"] = ("one string: the %d samples of a period are independent, the periods are serial -- bound by the hand-over "
                    "from period to period (an LDS round trip + the DF-I sum per step), not by HBM" % D)
  return {"units": float(C) * N, "elapsed": elapsed, "timing": ctx.last_stats, "kernel": kernel, "parity": parity, "roofline": roof,
          "C": C, "N": N}


def wl_mid(ctx, args, alz, C, N, steps, warmup, kind="butter6"):
  """One-section filters between "biquad" and "long FIR" (csrc/alz_mid.hip): `butter6` = ZFilter(butter(6, cutoff)) per channel
  as the reference's examples/butterworth_with_noise.py:52-67 builds it (nb = na = 7, one section: the recurrence is one multiply
  and six dependent additions per sample -- chain-bound); `maverage256` = maverage.recursive(256) (lazy_analysis.py:569-591:
  b0 = 1/256, b256 = -1/256, a1 = -1: HBM-bound)."""
  torch = ctx.torch
  if kind == "butter6":
    from scipy import signal
    ba = [signal.butter(6, wn) for wn in np.linspace(0.08, 0.6, C)]
    b, a = np.array([v[0] for v in ba]), np.array([v[1] for v in ba])
  else:
    b = np.zeros(257)
    b[0], b[256] = 1.0 / 256, -1.0 / 256
    a = np.array([1.0, -1.0])
  nb, na = b.shape[-1], a.shape[-1]
  bank = alz.FilterBank([(b, a)], n_inputs=C, device=ctx.local)
  bank.reset()
  x = ctx.noise((N, C), 9)
  y = torch.empty_like(x)
  elapsed, k_ms = ctx.timed(lambda: bank.process(x, layout="time", out=y), steps, warmup)
  kernel = bank.last_kernel
  parity = "skipped (--no-parity-check)"
  if ctx.rank == 0 and not args.no_parity_check:
    from oracle import oracle
    nchk = min(N, 2048)
    bank.reset()
    got = bank.process(x[:nchk].contiguous(), layout="time").cpu().numpy()
    ref = oracle.bank([nb], [na], b, a, x[:nchk].cpu().numpy(), layout="time")
    parity = "bit-exact vs oracle, %d channels x %d samples" % (C, nchk) if bits_equal(got, ref) else "MISMATCH"
    if N > nchk and not parity.startswith("MISMATCH"):
      bank.reset()
      bank.process(x, layout="time", out=y)
      pick = np.unique(np.linspace(0, C - 1, min(C, 16)).astype(int))
      idx = torch.from_numpy(pick).to(ctx.dev)
      got = y.index_select(1, idx).cpu().numpy()
      ref = oracle.bank([nb], [na], b[pick] if b.ndim == 2 else b, a[pick] if a.ndim == 2 else a,
                        x.index_select(1, idx).cpu().numpy(), layout="time")
      parity += ("; full block length: %d strided channels x %d samples bit-exact" % (len(pick), N) if bits_equal(got, ref)
                 else "; MISMATCH on the full block length")
      if "MISMATCH" in parity:
        parity = "MISMATCH: " + parity
  del x, y, bank
  torch.cuda.empty_cache()
  roof = hbm_roof(ALG_BYTES_PER_SAMPLE * C * N, k_ms)
  if kind == "butter6":
    roof["note"] = ("chain-bound, not HBM-bound: the reference's left-to-right sum puts one multiply and six additions of every "
                    "sample behind y[n-1] (7 dependent FP64 operations x ~7.5 cycles), so %d channels cannot pass ~%.0f Gsamples/s "
                    "bit-exactly whatever the kernel" % (C, C * 2.1e9 / 52.5 / 1e9))
  return {"units": float(C) * N, "elapsed": elapsed, "timing": ctx.last_stats, "kernel": kernel, "parity": parity, "roofline": roof,
          "C": C, "N": N}


def timevar_bank(torch, dev, C, N, per_channel):
  """Coefficients of the time-varying resonator bank: b0[n] x[n] + b2 x[n-2] - a1[n] y[n-1] - a2[n] y[n-2] whose series
  sweep the centre frequency from 200 Hz to 4 kHz at 48 kHz (``resonator.z_exp(Stream(freqs), bw)`` with vector-valued
  samples).  Shared form: one value per tap and sample for the whole bank ([N] tensors); per-channel form: [N, C] rows
  (the reference's ``repeat(ndarray)``-style coefficient Streams), channel c detuned by a factor 1 + c / 4 C.
  Returns (b, a, host copies of the shared series)."""
  n = np.arange(N)
  w = 2 * np.pi * np.geomspace(200., 4000., N) / 48000.
  r = np.exp(-(2 * np.pi * 100. / 48000.) / 2)
  g = (1 - r * r) / 2
  series = {"b0": np.full(N, g) * (1 + 1e-3 * np.sin(n / 997.)), "a1": -2 * r * np.cos(w), "a2": np.full(N, r * r)}
  d = {k: torch.from_numpy(v).to(dev) for k, v in series.items()}
  if per_channel:
    det = 1.0 + torch.arange(C, dtype=torch.float64, device=dev) / (4.0 * C)
    wd = torch.from_numpy(w).to(dev)[:, None] * det[None, :]
    d = {"b0": d["b0"][:, None].expand(N, C).contiguous(), "a1": (-2 * r) * torch.cos(wd),
         "a2": d["a2"][:, None].expand(N, C).contiguous()}
    del wd
  return [d["b0"], 0., -g], [1., d["a1"], d["a2"]], series


def wl_timevar(ctx, args, alz, C, N, steps, warmup, per_channel=False):
  """A bank steered by control streams: see timevar_bank."""
  torch = ctx.torch
  from audiolazy_amd import timevar
  b, a, _series = timevar_bank(torch, ctx.dev, C, N, per_channel)
  x = ctx.noise((N, C), 5)
  xh = torch.zeros((2, C), dtype=torch.float64, device=ctx.dev)
  yh = torch.zeros((2, C), dtype=torch.float64, device=ctx.dev)
  elapsed, k_ms = ctx.timed(lambda: timevar.process_block(b, a, x, xh=xh, yh=yh), steps, warmup)
  kernel = alz.last_kernel()                       # read from the library (alz_last_kernel), not assumed
  parity = "skipped (--no-parity-check)"
  if ctx.rank == 0 and not args.no_parity_check:
    # the WHOLE benchmarked block from a zero state, 64 strided channels against the C restatement alzo_tv_df1
    # (pinned on reference-generated vectors, tests/test_timevar_cpu.py)
    from oracle import oracle
    y = timevar.process_block(b, a, x)
    pick = np.unique(np.linspace(0, C - 1, min(C, 64)).astype(int))
    idx = torch.from_numpy(pick).to(ctx.dev)
    got, xs = y.index_select(1, idx).cpu().numpy(), x.index_select(1, idx).cpu().numpy()
    del y
    host = lambda v: ((v.index_select(1, idx) if v.dim() == 2 else v).cpu().numpy() if hasattr(v, "dim") else v)
    ref = oracle.tv_bank([host(v) for v in b], [host(v) for v in a], xs, layout="time")
    parity = ("bit-exact vs oracle (alzo_tv_df1): %d strided channels x %d samples (the whole block)" % (len(pick), N)
              if bits_equal(got, ref) else "MISMATCH")
  del x
  torch.cuda.empty_cache()
  # per-channel series: x + three series read, y written: 40 algorithmic bytes per channel-sample
  return {"units": float(C) * N, "elapsed": elapsed, "timing": ctx.last_stats, "kernel": kernel, "parity": parity,
          "roofline": hbm_roof((40.0 if per_channel else ALG_BYTES_PER_SAMPLE) * C * N, k_ms)}


def wl_collective(ctx, args, C, n):
  """The optional downstream step of SURVEY.md 8(e) / north_star: the filtering itself has no exchange, but
  a consumer that mixes every channel needs them on one device.  Two forms, each ONE collective of
  audiolazy_amd.sharding on the process group's backend (nccl = RCCL over xGMI): (a) reduce first -- every
  rank sums its own channels, then one all_reduce of the [n] mix; (b) the raw gather of the [n, C] shards
  to rank 0.  Timed outside the headline figure; the data stands for one block of this rank's output."""
  torch = ctx.torch
  from audiolazy_amd import sharding
  y = ctx.noise((n, C), 9)
  part = y.sum(dim=1)
  reps, out = 3, {}
  gathered = None
  for key, fn, nbytes in (("mixdown_all_reduce", lambda: sharding.mixdown(part, force=True), n * 8),
                          ("gather_to_rank0", lambda: sharding.gather_channels(y, ctx.world * C, channel_dim=1, dst=0, force=True),
                           n * C * 8),
                          ("mix_exact_to_rank0", lambda: sharding.mix_exact(y, ctx.world * C, channel_dim=1, dst=0, force=True),
                           n * C * 8)):
    got = fn()                                   # warm-up (communicator set-up) and the checked result
    ctx.sync_all()
    t0 = time.perf_counter()
    for _ in range(reps):
      fn()
    ctx.sync_all()
    el = (time.perf_counter() - t0) / reps
    el = max(r[0] for r in ctx.all_ranks([el]))
    out[key] = {"ms": el * 1e3, "bytes_per_rank": nbytes, "GBps_per_rank": nbytes / el / 1e9}
    if key == "mixdown_all_reduce":
      tot = sum(r[0] for r in ctx.all_ranks([float(part.sum())]))
      ok = abs(float(got.sum()) - tot) <= 1e-9 * max(1.0, abs(tot)) and (ctx.world > 1 or bool(torch.equal(got, part)))
    elif key == "mix_exact_to_rank0":
      # the reference's order over the GLOBAL channel index: ((c0 + c1) + c2) ..., bit for bit (checked on 512 rows)
      if ctx.rank == 0:
        rows = gathered[:512]
        acc = rows[:, 0].clone()
        for ch in range(1, rows.shape[1]):
          acc = acc + rows[:, ch]
        ok = tuple(got.shape) == (n,) and bool(torch.equal(got[:512], acc))
      else:
        ok = got is None
    elif ctx.rank == 0:
      ok = tuple(got.shape) == (n, ctx.world * C) and bool(torch.equal(got[:, :C], y))
      gathered = got
    else:
      ok = got is None
    out[key]["check"] = "ok" if all(r[0] == 1.0 for r in ctx.all_ranks([1.0 if ok else 0.0])) else "MISMATCH"
  if args.backend == "nccl" and ctx.world == 1:
    # the same collective through the C ABI's own RCCL binding (include/alz.h alz_comm_*), one-rank communicator:
    # the path of a caller that does not use torch.distributed at all
    try:
      comm = sharding.DirectComm(0, 1, sharding.DirectComm.unique_id(), device=ctx.local)
      g = comm.gather(y, dst=0)
      m = comm.sum(part)
      torch.cuda.synchronize(ctx.dev)
      ok = bool(torch.equal(g[0], y)) and bool(torch.equal(m, part))
      comm.close()
      out["c_abi_direct_rccl"] = {"check": "ok" if ok else "MISMATCH", "calls": "alz_comm_create / alz_comm_gather / alz_comm_sum"}
    except Exception as exc:        # (reported, not fatal: the torch.distributed path above is the one bench.py relies on)
      out["c_abi_direct_rccl"] = {"check": "skipped", "why": "%s: %s" % (type(exc).__name__, str(exc)[:160])}
  del y, gathered
  torch.cuda.empty_cache()
  bad = [k for k, v in out.items() if v["check"] not in ("ok", "skipped")]
  return {"workload": "downstream step, outside the filter path: channel mix over ranks as one all_reduce of the per-rank "
                      "sums, and the raw gather of every rank's [%d, %d] float64 shard to rank 0" % (n, C),
          "backend": "%s (%s)" % (args.backend, "RCCL" if args.backend == "nccl" else "CPU tensors, test mode"),
          "ranks": ctx.world, "parity": ("MISMATCH: " + ", ".join(bad)) if bad else "collective results checked on every rank",
          "collectives": out}


_PMC = None


PMC_TABLES = ("r06_pmc_traffic_table.json", "r05_pmc_traffic_table.json", "r04_pmc_traffic_table.json", "r03_pmc_traffic_table.json")     # newest first


def fill_traffic(roof, key):
  """roofline.traffic from the committed per-workload PMC table (profiles/r0N_pmc_traffic_table.json), scaled to this
  launch's algorithmic bytes; labelled as what it is: measured by rocprofv3 in separate runs, not inside this one."""
  global _PMC
  if _PMC is None:
    _PMC = {}
    for name in PMC_TABLES:
      try:
        table = json.load(open(os.path.join(ROOT, "profiles", name)))["workloads"]
      except (OSError, KeyError, ValueError):
        continue
      for k, row in table.items():
        _PMC.setdefault(k, dict(row, table=name))
  row = _PMC.get(key)
  if not row:
    return
  alg = roof.get("algorithmic_bytes_per_launch")
  if not alg:
    return
  roof["traffic"] = row["traffic_over_algorithmic"] * alg
  roof["traffic_ratio"] = row["traffic_over_algorithmic"]
  roof["traffic_source"] = ("NOT measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload "
                            "(profiles/%s, key %r): x%.4f of algorithmic%s"
                            % (row["table"], key, row["traffic_over_algorithmic"], "" if row.get("fetch_correction") == 2.0 else
                               " (FETCH_SIZE taken raw: a table of rounds 3 - 4)"))


def entry(res, world, steps, unit, workload, key=None):
  """A secondary-workload record: same fields as the main line's core."""
  roof = dict(res["roofline"])
  if key:
    fill_traffic(roof, key)
  # the same fraction priced on the wall time of a step (launch gaps included), next to the kernel-time one
  roof["frac_from_ms_per_step"] = roof["frac"] * roof["kernel_ms_avg"] / (res["elapsed"] / steps * 1e3)
  out = {"workload": workload, "value": world * res["units"] * steps / res["elapsed"] / 1e9, "unit": unit,
         "steps": steps, "ms_per_step": res["elapsed"] / steps * 1e3, "kernel": res["kernel"],
         "parity": res["parity"], "roofline": roof}
  if res.get("timing"):
    out["timing"] = res["timing"]      # timed_batches(): the figures above are the MEDIAN batch's
    # ... and the contract's own protocol beside it (W warm-up steps, then exactly K steps between two synchronisations:
    # the calibration run of timed_batches), so that records stay comparable with rounds 1 - 4 and with multi-rank runs
    out["ms_per_step_contract"] = res["timing"]["calibration_ms_per_step"]
  return out


def short_parity(p):
  """The parity verdict in a few dozen characters: status + the extents compared (the prose stays in the full record)."""
  import re
  p = str(p)
  if p.startswith("MISMATCH") or p.startswith("skipped"):
    return p[:100]
  dims = re.findall(r"\d+(?: strided)? (?:channels|bands|frames)(?: x \d+ (?:input )?streams)?(?: x \d+ samples)?"
                    r"(?: \(every 16th of \d+ \+ the last 4096\))?", p)
  dims = [d.replace(" strided", "").replace(" channels", "ch").replace(" bands", "b").replace(" input streams", "s")
           .replace(" streams", "s").replace(" samples", "").replace(" frames", "fr").replace("every 16th of ", "1/16 of ")
           .replace(" + the last 4096", "+last 4096").replace(" x ", "x") for d in dims]
  err = re.findall(r"(?:max normalised error|within) ([0-9.e+-]+)", p)
  if p.startswith("bit-exact") and not err:
    status = "bit-exact"
  elif err:
    status = "err<=%s (%s)" % (max(err, key=float), "mixed: see full record" if "bit-exact" in p.split(";")[0] and "not bit-exact" not in p
                               else "opt-in mode" if "by design" in p else "contract 1e-6")
  else:
    status = p[:60]
  return status + (" [" + "; ".join(dims) + "]" if dims else "")


# The order of the compact line's secondary entries: the BASELINE configs come LAST, so that a record that keeps only
# the tail of the line still holds configs[2..4].
SECONDARY_ORDER = ("downstream_collective", "strong_scaling", "narrow512_bit_exact", "narrow512_time_parallel",
                   "narrow512_time_parallel_three_launch", "narrow512_time_parallel_chan", "biquad_chan", "biquad_chan_fma", "biquad_8192", "envelope_abs", "comb_fb", "comb_fb_chan", "karplus_one_string", "iir_order6", "maverage_recursive_256",
                   "timevar_shared", "timevar_per_channel",
                   "gammatone_one_stream", "gammatone_one_stream_time_parallel", "gammatone_one_stream_time_parallel_tm", "lpc_1m", "lpc_1m_bit_identical",
                   "lpc_fma", "gammatone_fma", "biquad_fma", "fir256_fma", "fir256_bit_exact", "gammatone", "lpc", "lpc_bit_identical")


def compact_line(full):
  """The ONE line bench.py prints: the contract's fields, `roofline`, `cpu_baseline` and a slim `secondary`
  ({value, unit, ms_per_step, frac, traffic_ratio, parity} per workload) -- under ~6 KB.  Workload prose, kernel names,
  traffic provenance and the CPU legs' sampling notes are in the full record (--full-json, default
  gpurun_out/bench_full.json) and in profiles/bench_legend.md."""
  line = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                               "scaling", "vs_baseline", "dtype", "data")}
  cfg = dict(full["config"])
  cfg["workload"] = cfg["workload"][:160]
  if "kernel" in cfg:
    cfg["kernel"] = cfg["kernel"][:120]
  line["config"] = cfg
  roof = full["roofline"]
  line["roofline"] = {k: roof[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms_avg",
                                           "algorithmic_bytes_per_launch", "algorithmic_flops_per_launch", "traffic_ratio")
                      if k in roof}
  if roof.get("traffic") is not None:
    line["roofline"]["traffic_source"] = "rocprofv3 --pmc table under profiles/ (separate runs of this workload)"
  if "cpu_baseline" in full:
    cpu = full["cpu_baseline"]
    line["cpu_baseline"] = {"value": cpu["value"], "unit": cpu["unit"], "cores": cpu["cores"], "kind": cpu["kind"],
                            "host_logical_cpus": cpu.get("host_logical_cpus"),
                            "sample": cpu["sample"][:150],
                            "legs": {k: round(v["value"], 6) for k, v in cpu.get("legs", {}).items()}}
  for k in ("per_rank", "process_group"):
    if k in full:
      line[k] = full[k]
  sec = full.get("secondary") or {}
  if sec:
    line["legend"] = "profiles/bench_legend.md"
    keys = [k for k in SECONDARY_ORDER if k in sec] + [k for k in sec if k not in SECONDARY_ORDER]
    keys.sort(key=lambda k: SECONDARY_ORDER.index(k) if k in SECONDARY_ORDER else -1)
    slim = {}
    for k in keys:
      e = sec[k]
      if "roofline" not in e:          # the downstream collective: its own small record
        slim[k] = {"parity": e.get("parity", "")[:60], "ranks": e.get("ranks"), "backend": e.get("backend", "")[:40],
                   "collectives": {n: {f: (round(v, 4) if isinstance(v, float) else v) for f, v in c.items() if f in ("ms", "GBps_per_rank", "check")}
                                   for n, c in e.get("collectives", {}).items()}}
        continue
      r = e["roofline"]
      slim[k] = {"value": round(e["value"], 4), "unit": e["unit"], "ms_per_step": round(e["ms_per_step"], 5),
                 "frac": round(r["frac"], 4), "traffic_ratio": None if r.get("traffic_ratio") is None else round(r["traffic_ratio"], 4),
                 "parity": short_parity(e["parity"])}
      if "per_rank_frac" in e:
        slim[k]["per_rank_frac"] = e["per_rank_frac"]
      if "timing" in e:                # median of `batches` batches; the spread beside it
        t = e["timing"]
        slim[k]["ms_min_max"] = [round(t["ms_per_step_min"], 4), round(t["ms_per_step_max"], 4)]
        slim[k]["n"] = "%dx%d" % (t["batches"], t["steps_per_batch"])
    line["secondary"] = slim
  return line


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=10)
  ap.add_argument("--warmup", type=int, default=2)
  ap.add_argument("--channels", type=int, default=4096, help="channels per GPU (weak) / in total (strong)")
  ap.add_argument("--log2-samples", type=int, default=20, help="block length per channel = 2**this")
  ap.add_argument("--layout", choices=["time", "chan"], default="time")
  ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
  ap.add_argument("--streams", type=int, default=64, help="--workload gammatone: input streams per GPU")
  ap.add_argument("--bank-layout", choices=["time", "chan"], default="chan",
                  help="--workload gammatone: x [S, N] -> y [B * S, N] (chan) or x [N, S] -> y [N, B * S] (time)")
  ap.add_argument("--no-cpu-baseline", action="store_true")
  ap.add_argument("--no-secondary", action="store_true", help="skip configs[2..4] and the narrow-bank run")
  ap.add_argument("--no-parity-check", action="store_true",
                  help="skip the post-run parity launches (keeps a rocprofv3 --stats average clean)")
  ap.add_argument("--parity-samples", type=int, default=16384, help="samples per channel of the parity block")
  ap.add_argument("--fused", action="store_true",
                  help="opt-in FMA mode of the streaming kernels: NOT bit-exact (reported as such); default off")
  ap.add_argument("--time-parallel", type=int, default=None,
                  help="force the time-parallel (chunked state propagation) kernel on (1) / off (0) for the "
                       "biquad bank; default: the engine's own choice (bit-exact kernels)")
  ap.add_argument("--lpc-exact", action="store_true", help="--workload lpc: the bit-identical path (ALZ_LPC_DENSE)")
  ap.add_argument("--lpc-frames", type=int, default=65536, help="--workload lpc: frames per launch")
  ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                  help="process-group backend for N > 1 (nccl = RCCL; gloo lets tests run several ranks on one GPU)")
  ap.add_argument("--init-dist", action="store_true",
                  help="initialise the process group even for ONE rank, so that the barrier, the MAX all-reduce and the "
                       "downstream gather / mixdown run on the chosen backend (RCCL on a one-GPU box)")
  ap.add_argument("--launch-check", action="store_true",
                  help="start the ranks, run the protocol's collectives, print one line and exit (no device work)")
  ap.add_argument("--full-json", default=None,
                  help="where rank 0 writes the full (verbose) record; default gpurun_out/bench_full.json when that "
                       "directory can be made, '-' to skip")
  ap.add_argument("--comb-delay", type=int, default=441, help="--workload comb: the feedback delay in samples")
  ap.add_argument("--comb-linearized", action="store_true", help="--workload comb: two adjacent feedback taps (linearize()d fractional delay)")
  ap.add_argument("--in-place", action="store_true", help="--workload comb / biquad / envelope: y = x (with --no-parity-check: the block is overwritten by every step)")
  ap.add_argument("--workload", choices=["biquad", "fir", "gammatone", "lpc", "envelope", "timevar", "comb", "butter6", "maverage256"], default="biquad",
                  help="biquad = configs[1] (the contract line); fir = configs[2]; gammatone = configs[3] "
                       "(256 bands x 64 streams per GPU); lpc = configs[4] (65536 frames x 480, order 16)")
  args = ap.parse_args()

  if "WORLD_SIZE" not in os.environ and args.gpus > 1:
    sys.exit(relaunch(args.gpus))      # plain `python bench.py --gpus N`: start the N ranks ourselves
  rank = int(os.environ.get("RANK", "0"))
  world = int(os.environ.get("WORLD_SIZE", "1"))
  if world != args.gpus and rank == 0:
    print("bench.py: --gpus %d but the launcher started %d rank(s); reporting n_gpus = %d"
          % (args.gpus, world, world), file=sys.stderr)
  if args.launch_check:
    sys.exit(launch_check(args, rank, world))
  C, N = args.channels, 1 << args.log2_samples

  # the CPU legs fork a process pool: run them before torch / HIP exist in this process
  cpu = None
  if rank == 0 and world == 1 and not args.no_cpu_baseline and args.workload == "biquad":
    cpu = cpu_baseline(*resonator_coefs(C))

  import audiolazy_amd as alz
  alz.load_library()  # raises loudly when libalzhip.so is missing: there is no CPU path
  ctx = Ctx(args)

  secondary = {}
  if args.workload == "biquad":
    if args.scaling == "strong":
      from audiolazy_amd.sharding import shard_range
      lo, hi = shard_range(C, world, rank)
      res = wl_biquad(ctx, args, alz, hi - lo, N, lo, C, args.steps, args.warmup, time_parallel=args.time_parallel)
      total_units = float(C) * N
      workload = ("configs[1], strong scaling: the %d-channel biquad IIR bank (resonator.z_exp per channel) divided "
                  "over %d rank(s), 48 kHz float64, %d-sample blocks" % (C, world, N))
    else:
      res = wl_biquad(ctx, args, alz, C, N, rank * C, world * C, args.steps, args.warmup,
                      time_parallel=args.time_parallel)
      total_units = float(world) * C * N
      workload = ("configs[1]: %d-channel biquad IIR bank (resonator.z_exp per channel), 48 kHz float64, "
                  "%d-sample blocks, 1 MI355X per rank" % (C, N))
    metric, unit = "Gsamples/s through ZFilter IIR biquad bank", "Gsamples/s"
    config = {"workload": workload, "channels_per_gpu": res["C"], "block_samples": N,
              "layout": "time-major [N, C]" if args.layout == "time" else "channel-major [C, N]",
              "kernel": res["kernel"], "parity_spot_check": res["parity"]}
    roof = res["roofline"]
    # HBM bytes per launch: FETCH_SIZE / WRITE_SIZE cannot be read from inside this process; the figures are the
    # committed rocprofv3 --pmc measurements of the same workloads (tools/pmc_workloads.sh -> pmc_table.py)
    if (C, N) == (4096, 1 << 20) and args.layout == "time" and not args.fused and not args.time_parallel:
      fill_traffic(roof, "headline")
    if not args.no_secondary and (C, N) == (4096, 1 << 20):
      if world == 1:
        ctx.robust = True      # (timed_batches: median of several >= 20 ms batches per workload)
        r = wl_fir(ctx, args, alz, 8192, 1 << 18, 5, 1, fused=False)
        secondary["fir256_bit_exact"] = entry(r, 1, 5, "Gsamples/s", "configs[2]: 256-tap FIR lowpass (Hamming-"
                                              "windowed sinc, shared taps) x 8192 channels x 2^18 samples, float64", key="fir256_bit_exact")
        r = wl_fir(ctx, args, alz, 8192, 1 << 18, 5, 1, fused=True)
        secondary["fir256_fma"] = entry(r, 1, 5, "Gsamples/s", "configs[2] in the opt-in FMA mode (alz_bank_set_fused)", key="fir256_fma")
        r = wl_gammatone(ctx, args, alz, 10, 2)
        secondary["gammatone"] = entry(r, 1, 10, "Gsamples/s", "configs[3]: ERB gammatone filterbank (gammatone.slaney), "
                                       "256 bands x 64 input streams per GPU (512 over 8), 2^16-sample blocks", key="gammatone")
        r = wl_gammatone(ctx, args, alz, 10, 2, fused=True)
        secondary["gammatone_fma"] = entry(r, 1, 10, "Gsamples/s", "configs[3] in the opt-in FMA mode (alz_bank_set_fused)", key="gammatone_fma")
        r = wl_gammatone(ctx, args, alz, 10, 2, streams=1, log2n=20)
        secondary["gammatone_one_stream"] = entry(r, 1, 10, "Gsamples/s", "the reference's own shape of configs[3]: 256 bands "
                                                  "on ONE stream x 2^20 samples (input expanded to a column per band, then the sections as a pipeline over chunks of the time axis)", key="gammatone_one_stream")
        r = wl_gammatone(ctx, args, alz, 10, 2, streams=1, log2n=20, time_parallel=True)
        secondary["gammatone_one_stream_time_parallel"] = entry(r, 1, 10, "Gsamples/s", "same, opt-in time-parallel mode", key="gammatone_one_stream_time_parallel")
        r = wl_gammatone(ctx, args, alz, 10, 2, streams=1, log2n=20, time_parallel=True, layout="time")
        secondary["gammatone_one_stream_time_parallel_tm"] = entry(r, 1, 10, "Gsamples/s", "same, time-major: x [N, 1] -> y [N, 256], the "
                                                                   "reference's vector-valued samples", key="gammatone_one_stream_time_parallel_tm")
        r = wl_lpc(ctx, args, alz, 20, 3)
        secondary["lpc"] = entry(r, 1, 20, "Gframes/s", "configs[4]: lpc.kautocor order 16 on 65536 concurrent "
                                 "480-sample frames", key="lpc")
        r = wl_lpc(ctx, args, alz, 20, 3, exact=True)
        secondary["lpc_bit_identical"] = entry(r, 1, 20, "Gframes/s", "configs[4] with the reference's dense Levinson-Durbin "
                                               "(ALZ_LPC_DENSE), one launch: coefficients and error bit-identical on every frame", key="lpc_bit_identical")
        r = wl_lpc(ctx, args, alz, 20, 3, fused=True)
        secondary["lpc_fma"] = entry(r, 1, 20, "Gframes/s", "configs[4] with fused multiply-adds in the autocorrelation "
                                     "sums (opt-in ALZ_LPC_FUSED; not pinned to the last bit)", key="lpc_fma")
        r = wl_lpc(ctx, args, alz, 5, 1, frames=1 << 20)
        secondary["lpc_1m"] = entry(r, 1, 5, "Gframes/s", "configs[4] as a 2^20-frame batch (SURVEY.md 8d: a bandwidth fraction "
                                    "that is not dominated by the 65536-frame launch): 4 GB of signal per launch", key="lpc_1m")
        r = wl_lpc(ctx, args, alz, 5, 1, exact=True, frames=1 << 20)
        secondary["lpc_1m_bit_identical"] = entry(r, 1, 5, "Gframes/s", "the same batch through the bit-identical path", key="lpc_1m_bit_identical")
        r = wl_envelope(ctx, args, alz, 4096, N, 5, 1)
        secondary["envelope_abs"] = entry(r, 1, 5, "Gsamples/s", "envelope.abs (lowpass.pole of |x|) on 4096 channels x 2^20 "
                                          "samples: the elementwise stage of SURVEY.md 8 (f1) fused into the filter kernel", key="envelope_abs")
        r = wl_comb(ctx, args, alz, 4096, 1 << 18, 441, 5, 1)
        secondary["comb_fb"] = entry(r, 1, 5, "Gsamples/s", "comb.tau (y[n] = x[n] + alpha y[n - 441], lazy_filters.py:1118-1147) x 4096 "
                                     "channels x 2^18 samples, time-major: lanes over the delay line, the ring in LDS", key="comb_fb")
        r = wl_comb(ctx, args, alz, 4096, 1 << 18, 441, 5, 1, layout="chan")
        secondary["comb_fb_chan"] = entry(r, 1, 5, "Gsamples/s", "the same bank on channel-major blocks [C, N]: a wave per channel", key="comb_fb_chan")
        r = wl_comb(ctx, args, alz, 1, 1 << 22, 109, 5, 1, layout="chan", linearized=True)
        secondary["karplus_one_string"] = entry(r, 1, 5, "Gsamples/s", "ONE Karplus-Strong string (lazy_synth.py:624-657: comb.tau(...).linearize(), "
                                                "two adjacent feedback taps at 109 / 110 samples) x 2^22 samples: the reference's own use "
                                                "(examples/ode_to_joy.py:84)", key="karplus_one_string")
        r = wl_mid(ctx, args, alz, 4096, 1 << 18, 5, 1, kind="butter6")
        secondary["iir_order6"] = entry(r, 1, 5, "Gsamples/s", "ZFilter(butter(6, cutoff)) as ONE section (examples/butterworth_with_noise.py:52-67) "
                                        "x 4096 channels x 2^18 samples", key="iir_order6")
        r = wl_mid(ctx, args, alz, 4096, 1 << 18, 5, 1, kind="maverage256")
        secondary["maverage_recursive_256"] = entry(r, 1, 5, "Gsamples/s", "maverage.recursive(256) (lazy_analysis.py:569-591) x 4096 channels x 2^18 "
                                                    "samples", key="maverage_recursive_256")
        r = wl_timevar(ctx, args, alz, 4096, 1 << 18, 5, 1)
        secondary["timevar_shared"] = entry(r, 1, 5, "Gsamples/s", "time-varying resonator bank: 4096 channels x 2^18 samples "
                                            "steered by three coefficient series shared by the channels (Stream coefficients, "
                                            "lazy_filters.py:197-224)", key="timevar_shared")
        r = wl_timevar(ctx, args, alz, 4096, 1 << 18, 5, 1, per_channel=True)
        secondary["timevar_per_channel"] = entry(r, 1, 5, "Gsamples/s", "time-varying resonator bank: 4096 channels x 2^18 samples, "
                                                 "three coefficient series PER CHANNEL (rows of coefficients per sample: "
                                                 "40 B per channel-sample)", key="timevar_per_channel")
        # configs[1] as channel-major rows [C, N] (one Stream per row), bit-exact and in the opt-in FMA mode (round 6: the fused
        # channel-major kernel with the storing wave and non-temporal tiles)
        if not args.fused:
          r = wl_biquad(ctx, args, alz, 4096, N, 0, 4096, 5, 1, check=True, layout="chan")
          secondary["biquad_chan"] = entry(r, 1, 5, "Gsamples/s", "configs[1] with the block as channel-major rows [4096, 2^20]", key="biquad_chan")
          args.fused = True
          try:
            r = wl_biquad(ctx, args, alz, 4096, N, 0, 4096, 5, 1, check=True, layout="chan")
          finally:
            args.fused = False
          secondary["biquad_chan_fma"] = entry(r, 1, 5, "Gsamples/s", "the same in the opt-in FMA mode (alz_bank_set_fused): k_duo's fused "
                                               "instantiation with the storing wave and non-temporal tiles", key="biquad_chan_fma")
          # configs[1] itself (time-major) in the opt-in FMA mode: the three-wave fused kernel on a common tile clock
          args.fused = True
          try:
            r = wl_biquad(ctx, args, alz, 4096, N, 0, 4096, 5, 1, check=True, layout="time")
          finally:
            args.fused = False
          secondary["biquad_fma"] = entry(r, 1, 5, "Gsamples/s", "configs[1] (time-major) in the opt-in FMA mode (alz_bank_set_fused): k_duo's fused "
                                          "instantiation with the storing wave and non-temporal tiles, the workgroups' tile requests "
                                          "on one clock (alz_wave.hip, tile_pace)", key="biquad_fma")
        # twice configs[1]'s channels on one GPU (the same bytes per launch: 8192 channels x 2^19 samples): two workgroups per CU, so the
        # recurrence waves' issue rate no longer bounds the bit-exact kernel -- the memory system does, on the common tile clock
        if not args.fused:
          r = wl_biquad(ctx, args, alz, 8192, 1 << 19, 0, 8192, 5, 1, check=True, layout="time")
          secondary["biquad_8192"] = entry(r, 1, 5, "Gsamples/s", "configs[1]'s bank at twice the width: 8192 channels x 2^19 samples, bit-exact "
                                           "(k_duo, two workgroups per CU on the common tile clock)", key="biquad_8192")
        if hasattr(alz.FilterBank, "set_time_parallel"):
          for mode, key in ((0, "narrow512_bit_exact"), (1, "narrow512_time_parallel"), (8192, "narrow512_time_parallel_three_launch"),
                            (1, "narrow512_time_parallel_chan")):
            r = wl_biquad(ctx, args, alz, 512, N, 0, 4096, 10, 2, check=True, time_parallel=mode,
                          layout="chan" if key.endswith("_chan") else None)
            secondary[key] = entry(r, 1, 10, "Gsamples/s", "one GPU's share of configs[1] sharded over 8: 512 channels "
                                   "x 2^20 samples" + (", time-parallel mode (opt-in, not bit-exact)" if mode else "")
                                   + (": the engine's choice, the ONE-pass form (chunks resident in LDS, 16 B of traffic per sample)" if mode == 1 else "")
                                   + (": the three-launch form (explicit chunk length; 24 B of traffic per sample)" if mode > 1 else ""),
                                   key=key)
      elif args.scaling == "weak" and C % world == 0:
        from audiolazy_amd.sharding import shard_range
        lo, hi = shard_range(C, world, rank)
        tp = 1 if hasattr(alz.FilterBank, "set_time_parallel") else None
        r = wl_biquad(ctx, args, alz, hi - lo, N, lo, C, 5, 1, check=False, time_parallel=tp)
        e = entry(r, 1, 5, "Gsamples/s", "configs[1]'s 4096 channels divided over %d ranks (strong scaling), "
                  "%d channels per GPU" % (world, hi - lo))
        e["value"] = float(C) * N * 5 / r["elapsed"] / 1e9
        # every rank's own roofline fraction on its shard, next to the whole-job figure
        e["per_rank_frac"] = [round(row[0], 4) for row in ctx.all_ranks([r["roofline"]["frac"]])]
        secondary["strong_scaling"] = e
  elif args.workload == "fir":
    if (C, N) == (4096, 1 << 20):   # configs[2] defaults
      C, N = 8192, 1 << 18
    res = wl_fir(ctx, args, alz, C, N, args.steps, args.warmup, fused=args.fused)
    total_units = float(world) * C * N
    metric, unit = "Gsamples/s through ZFilter FIR-256 bank", "Gsamples/s"
    config = {"workload": "configs[2]: 256-tap FIR lowpass (Hamming-windowed sinc, shared taps) x %d channels, "
                          "float64, %d-sample blocks, 1 MI355X per rank" % (C, N),
              "channels_per_gpu": C, "block_samples": N, "layout": "time-major [N, C]",
              "kernel": res["kernel"], "parity_spot_check": res["parity"]}
    roof = res["roofline"]
  elif args.workload == "envelope":
    res = wl_envelope(ctx, args, alz, C, N, args.steps, args.warmup)
    total_units = float(world) * C * N
    metric, unit = "Gsamples/s through a bank of envelope.abs followers (lowpass.pole on |x|)", "Gsamples/s"
    config = {"workload": "envelope.abs x %d channels, float64, %d-sample blocks, |x| fused into the filter kernel" % (C, N),
              "channels_per_gpu": C, "block_samples": N, "layout": "time-major [N, C]",
              "kernel": res["kernel"], "parity_spot_check": res["parity"]}
    roof = res["roofline"]
  elif args.workload == "gammatone":
    res = wl_gammatone(ctx, args, alz, args.steps, args.warmup, fused=args.fused, streams=args.streams,
                       log2n=args.log2_samples if args.streams < 64 else 16, time_parallel=bool(args.time_parallel),
                       layout=args.bank_layout)
    total_units = float(world) * res["units"]
    metric, unit = "Gsamples/s (band x stream x sample outputs) through the ERB gammatone bank", "Gsamples/s"
    config = {"workload": "configs[3]: ERB gammatone filterbank (gammatone.slaney, 4-section cascades), %d bands x %d "
                          "input streams per GPU (512 streams sharded over 8), %d-sample blocks, float64, "
                          "%s" % (res["B"], res["S"], res["N"],
                                  "x [N, S] -> y [N, B, S]" if res["layout"] == "time" else "x [S, N] -> y [B, S, N]"),
              "kernel": res["kernel"], "parity_spot_check": res["parity"]}
    roof = res["roofline"]
  elif args.workload == "comb":
    if (C, N) == (4096, 1 << 20):
      N = 1 << 18
    res = wl_comb(ctx, args, alz, C, N, args.comb_delay, args.steps, args.warmup, layout=args.layout,
                  linearized=args.comb_linearized, inplace=args.in_place)
    total_units = float(world) * C * N
    metric, unit = "Gsamples/s through a bank of feedback combs (comb.tau)", "Gsamples/s"
    config = {"workload": "comb.tau x %d channels, delay %d%s, float64, %d-sample blocks%s" % (
                  C, args.comb_delay, " (linearized: two taps)" if args.comb_linearized else "", N, ", in place" if args.in_place else ""),
              "channels_per_gpu": C, "block_samples": N, "layout": "time-major [N, C]" if args.layout == "time" else "channel-major [C, N]",
              "kernel": res["kernel"], "parity_spot_check": res["parity"]}
    roof = res["roofline"]
  elif args.workload in ("butter6", "maverage256"):
    if (C, N) == (4096, 1 << 20):
      N = 1 << 18
    res = wl_mid(ctx, args, alz, C, N, args.steps, args.warmup, kind=args.workload)
    total_units = float(world) * C * N
    metric, unit = "Gsamples/s through a bank of one-section filters (%s)" % args.workload, "Gsamples/s"
    config = {"workload": "%s x %d channels, float64, %d-sample blocks" % (args.workload, C, N),
              "channels_per_gpu": C, "block_samples": N, "layout": "time-major [N, C]",
              "kernel": res["kernel"], "parity_spot_check": res["parity"]}
    roof = res["roofline"]
  elif args.workload == "timevar":
    per_ch = args.streams != 64          # (--streams 0: per-channel series; default: series shared by the bank)
    if (C, N) == (4096, 1 << 20):
      N = 1 << 18
    res = wl_timevar(ctx, args, alz, C, N, args.steps, args.warmup, per_channel=per_ch)
    total_units = float(world) * C * N
    metric, unit = "Gsamples/s through a time-varying resonator bank (Stream coefficients)", "Gsamples/s"
    config = {"workload": "time-varying resonator bank, %d channels x %d samples, coefficient series %s" % (
                  C, N, "per channel" if per_ch else "shared by the bank"),
              "channels_per_gpu": C, "block_samples": N, "layout": "time-major [N, C]",
              "kernel": res["kernel"], "parity_spot_check": res["parity"]}
    roof = res["roofline"]
  else:
    res = wl_lpc(ctx, args, alz, args.steps, args.warmup, fused=args.fused, exact=args.lpc_exact, frames=args.lpc_frames)
    total_units = float(world) * res["units"]
    metric, unit = "Gframes/s through lpc.kautocor (autocorrelation + Levinson-Durbin, order 16, 10 ms frames)", "Gframes/s"
    config = {"workload": "configs[4]: lazy_lpc order-16 on %d concurrent 480-sample frames, float64" % res["F"],
              "kernel": res["kernel"], "parity_spot_check": res["parity"]}
    roof = res["roofline"]

  # every rank's own figures next to the whole-job one (MAX over ranks prices the line's value)
  per_rank = None
  if ctx.dist is not None:
    rows = ctx.all_ranks([ctx.local_elapsed[0], res["units"], roof["kernel_ms_avg"], roof["frac"]])
    per_rank = [{"rank": r, "value": u * args.steps / el / 1e9, "unit": unit, "ms_per_step": el / args.steps * 1e3,
                 "kernel_ms_avg": k, "roofline_frac": f} for r, (el, u, k, f) in enumerate(rows)]
    if args.workload == "biquad":
      secondary["downstream_collective"] = wl_collective(ctx, args, res["C"], min(N, 1 << 16))

  bad = False
  if rank == 0:
    out = {
      "metric": metric, "value": total_units * args.steps / res["elapsed"] / 1e9, "unit": unit,
      "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
      "ms_per_step": res["elapsed"] / args.steps * 1e3,
      "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
      "dtype": "f64", "data": "synthetic", "config": config, "roofline": roof,
    }
    if per_rank is not None:
      out["per_rank"] = per_rank
      out["process_group"] = {"backend": args.backend, "ranks": world,
                              "launcher": os.environ.get("TORCHELASTIC_RUN_ID") and "torch.distributed.run" or "none (--init-dist)"}
    if cpu is not None:
      out["cpu_baseline"] = cpu
    if secondary:
      out["secondary"] = secondary
    full_path = args.full_json
    if full_path is None:
      full_path = os.path.join(ROOT, "gpurun_out", "bench_full.json")
    if full_path != "-":
      try:
        os.makedirs(os.path.dirname(os.path.abspath(full_path)), exist_ok=True)
        with open(full_path, "w") as f:
          json.dump(out, f, indent=1)
      except OSError:
        pass
    print(json.dumps(compact_line(out)))
    checks = [config.get("parity_spot_check") or ""] + [e["parity"] for e in secondary.values()]
    bad = any(str(c).startswith("MISMATCH") for c in checks)
    if bad:
      print("bench.py: PARITY MISMATCH -- the number above is not a valid result", file=sys.stderr)
  if ctx.dist is not None:
    ctx.dist.barrier()
    ctx.dist.destroy_process_group()
  if bad:
    sys.exit(3)


if __name__ == "__main__":
  main()
