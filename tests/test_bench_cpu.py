"""bench.py's CPU side, which needs no GPU: the reference-path legs of ``cpu_baseline`` (the generated
DF-I generator executed by CPython: one process, a process pool, vector-valued rows; the C port) and the
workload definitions they share with the GPU side."""
import os

import numpy as np


def test_cpu_baseline_legs_run_and_rank_as_expected():
  import bench
  b, a = bench.resonator_coefs(64)
  assert b.shape == (64, 3) and a.shape == (64, 3) and np.all(a[:, 0] == 1) and np.all(b[:, 1] == 0)
  cpu = bench.cpu_baseline(b, a, budget_s=0.3)
  assert {"value", "unit", "cores", "kind", "sample", "legs", "usable_cores", "host_logical_cpus"} <= set(cpu)
  # "reference" where the unmodified package is importable (this container), "port" elsewhere (the GPU boxes)
  assert cpu["kind"] == ("reference" if bench.reference_package() is not None else "port")
  assert cpu["unit"] == "Gsamples/s" and cpu["host_logical_cpus"] == os.cpu_count()
  legs = cpu["legs"]
  assert set(legs) == {"py_1proc", "py_pool", "py_rows", "c_port", "py_comb_1proc"}
  assert legs["py_comb_1proc"]["value"] < legs["py_1proc"]["value"]     # (the O(D) shift per sample, lazy_filters.py:254-255)
  assert all(leg["value"] > 0 and leg["unit"] == "Gsamples/s" for leg in legs.values())
  assert cpu["value"] == legs["py_pool"]["value"] and cpu["cores"] == legs["py_pool"]["cores"] >= 1
  # the interpreter pays per sample: one Python generator is far below the same statement compiled by gcc
  assert legs["py_1proc"]["value"] < legs["c_port"]["value"]
  assert legs["py_1proc"]["value"] < 0.05           # (tens of Msamples/s at the very most)


def test_fir_taps_are_the_windowed_sinc_of_the_survey():
  import bench
  taps = bench.fir_taps()
  assert taps.shape == (256,) and np.allclose(taps, taps[::-1]) and abs(taps.sum() - 1.0) < 0.02
  assert 1 <= bench.usable_cores() <= (os.cpu_count() or 1)


def test_bench_cli_names_every_workload_and_layout_knob():
  """The command line the driver and the notes use: parsed without a GPU (``--help`` exits before any device call)."""
  import subprocess, sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
  assert out.returncode == 0
  for flag in ("--gpus", "--steps", "--warmup", "--scaling", "--workload", "--streams", "--bank-layout", "--time-parallel",
               "--fused", "--no-cpu-baseline", "--no-secondary", "--no-parity-check"):
    assert flag in out.stdout, flag
  for workload in ("biquad", "fir", "gammatone", "lpc", "envelope"):
    assert workload in out.stdout


def test_lpc_flag_word_matches_the_header():
  """kautocor_frames(fused=, exact=) -> the flags of alz_lpc_kautocor_dev_ex as include/alz.h defines them."""
  import importlib, re
  lpc = importlib.import_module("audiolazy_amd.lpc")
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  header = open(os.path.join(root, "include", "alz.h")).read()
  fused = int(re.search(r"#define ALZ_LPC_FUSED (\d+)", header).group(1))
  dense = int(re.search(r"#define ALZ_LPC_DENSE (\d+)", header).group(1))
  assert lpc._lpc_flags(False, False) == 0
  assert lpc._lpc_flags(True, False) == fused and lpc._lpc_flags(False, True) == dense
  assert lpc._lpc_flags(True, True) == fused | dense and fused & dense == 0


def test_plain_gpus_n_starts_n_ranks_itself():
  """``python bench.py --gpus 2`` with no launcher around it must come back as TWO ranks (round-2 verdict: it
  used to run one rank and print n_gpus 1).  --launch-check stops before any device work, so this runs on
  gloo without a GPU: rank start-up, barrier, MAX all-reduce and the per-rank all_gather."""
  import json, subprocess, sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
  for n in (1, 2, 3):
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--backend", "gloo",
                          "--launch-check"], capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-1500:]
    lines = [json.loads(ln) for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    assert lines[0]["launch_check"] == "ok" and lines[0]["n_gpus"] == n
    assert lines[0]["launcher"] == ("torch.distributed.run" if n > 1 else "none")


def test_launcher_env_wins_over_the_flag():
  """Started by torchrun with a world size that differs from --gpus: the line reports the ranks that exist."""
  import json, subprocess, sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
         "--master-port", "29641", os.path.join(root, "bench.py"), "--gpus", "4", "--backend", "gloo", "--launch-check"]
  out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=root)
  assert out.returncode == 0, out.stderr[-1500:]
  line = [json.loads(ln) for ln in out.stdout.splitlines() if ln.startswith("{")][0]
  assert line["n_gpus"] == 2 and line["requested_gpus"] == 4
  assert "--gpus 4 but the launcher started 2" in out.stderr


def test_cpu_baseline_falls_back_to_the_port_without_the_reference(monkeypatch):
  import bench
  monkeypatch.setenv("AUDIOLAZY_REF", "/nonexistent")
  monkeypatch.setattr(bench, "_REF", None)
  b, a = bench.resonator_coefs(16)
  cpu = bench.cpu_baseline(b, a, budget_s=0.2)
  assert cpu["kind"] == "port" and "restated" in cpu["sample"]
  monkeypatch.setattr(bench, "_REF", None)


def test_compact_line_fits_a_truncating_record_and_ends_with_the_baseline_configs():
  """The line bench.py prints must survive a driver that keeps ~6 KB of it (round-3 review: the 14 KB line lost
  configs[2..4]).  Input: the committed full record of the round-3 driver command; checked: size, the contract's
  keys, the slim secondary fields, and that the BASELINE configs are the LAST entries."""
  import json
  import bench
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  full = json.load(open(os.path.join(root, "profiles", "r03_bench_full.json")))
  line = bench.compact_line(full)
  text = json.dumps(line)
  assert len(text) < 6144, len(text)
  for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "secondary"):
    assert key in line, key
  assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(line["roofline"])
  assert {"value", "unit", "cores", "kind", "sample"} <= set(line["cpu_baseline"])
  sec = line["secondary"]
  assert set(sec) == set(full["secondary"])
  for k, e in sec.items():
    assert set(e) == {"value", "unit", "ms_per_step", "frac", "traffic_ratio", "parity"}, k
    assert len(e["parity"]) <= 100 and e["parity"].split()[0] in ("bit-exact", "err<=1e-06", "MISMATCH") or e["parity"].startswith("err<="), e
  assert list(sec)[-5:] == ["fir256_fma", "fir256_bit_exact", "gammatone", "lpc", "lpc_bit_identical"]
  assert text.index('"secondary"') > text.index('"cpu_baseline"')      # the secondaries close the line
  # round 5's record: two more workloads, and the batch timer's spread (`ms_min_max`, `n`) per entry -- still under the cut
  full5 = json.load(open(os.path.join(root, "profiles", "r05_bench_first_process.json")))
  line5 = bench.compact_line(full5)
  assert len(json.dumps(line5)) < 6144, len(json.dumps(line5))
  for k, e in line5["secondary"].items():
    assert set(e) == {"value", "unit", "ms_per_step", "frac", "traffic_ratio", "parity", "ms_min_max", "n"}, k
    assert e["ms_min_max"][0] <= e["ms_per_step"] <= e["ms_min_max"][1]
  assert list(line5["secondary"])[-5:] == ["fir256_fma", "fir256_bit_exact", "gammatone", "lpc", "lpc_bit_identical"]
  # statuses survive the shortening
  assert bench.short_parity("MISMATCH on the full extent: x").startswith("MISMATCH")
  assert bench.short_parity("bit-exact vs oracle, 8192 channels x 512 samples; full extent: 64 strided channels x 262144 "
                            "samples bit-exact") == "bit-exact [8192chx512; 64chx262144]"
