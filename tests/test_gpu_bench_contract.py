"""bench.py's contract (one JSON line with the agreed keys) and __graft_entry__.smoke(), run as
the driver runs them; the N > 1 path is exercised with two ranks sharing the one GPU of the
test box (gloo for the barrier / max-time all-reduce, which is all the collectives it uses)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config", "roofline"}


def run(cmd, timeout=600):
  env = dict(os.environ, MASTER_ADDR="127.0.0.1")
  out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
  assert out.returncode == 0, out.stderr[-2000:]
  lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
  assert len(lines) == 1, out.stdout
  return json.loads(lines[0])


def check(line, n_gpus, steps, warmup):
  assert KEYS <= set(line), sorted(KEYS - set(line))
  assert (line["n_gpus"], line["steps"], line["warmup"]) == (n_gpus, steps, warmup)
  assert line["higher_is_better"] is True and line["scaling"] in ("weak", "strong") and line["vs_baseline"] is None
  assert line["dtype"] == "f64" and line["data"] == "synthetic" and "workload" in line["config"]
  roof = line["roofline"]
  assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(roof)
  assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-12
  assert line["value"] > 0 and line["ms_per_step"] > 0


def test_single_gpu_line_with_cpu_baseline():
  line = run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--channels", "512", "--log2-samples", "14"])
  check(line, 1, 3, 1)
  assert line["unit"] == "Gsamples/s" and line["config"]["parity_spot_check"].startswith("bit-exact vs oracle, 512 channels")
  cpu = line["cpu_baseline"]
  assert {"value", "unit", "cores", "kind", "sample", "legs"} <= set(cpu) and cpu["kind"] in ("port", "reference")
  assert 1 <= cpu["cores"] <= os.cpu_count() and cpu["host_logical_cpus"] == os.cpu_count()
  assert set(cpu["legs"]) == {"py_1proc", "py_pool", "py_rows", "c_port", "py_comb_1proc"}
  assert all(leg > 0 for leg in cpu["legs"].values())          # (the compact line keeps the legs' values only)
  # the interpreter path is orders of magnitude below the C port of the same statement
  assert cpu["legs"]["py_1proc"] < cpu["legs"]["c_port"]
  # the verbose record of the same run is on disk
  full = json.load(open(os.path.join(ROOT, "gpurun_out", "bench_full.json")))
  assert full["value"] == line["value"] and "sample" in full["cpu_baseline"]["legs"]["py_pool"]


def test_mismatch_is_a_failure(tmp_path):
  # a parity MISMATCH must end the run with a non-zero status (the value is not a result)
  code = ("import sys, numpy as np; sys.argv = ['bench.py', '--steps', '1', '--warmup', '0', '--channels', '64', "
          "'--log2-samples', '12', '--no-cpu-baseline']; import bench; "
          "bench.bits_equal = lambda a, b: False; bench.main()")
  out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=600)
  assert out.returncode == 3, (out.returncode, out.stderr[-500:])
  assert "MISMATCH" in out.stdout


@pytest.mark.parametrize("workload", ["fir", "gammatone", "lpc"])
def test_side_workloads(workload):
  extra = ["--channels", "256", "--log2-samples", "12"] if workload == "fir" else []
  line = run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--workload", workload] + extra)
  check(line, 1, 2, 1)


def test_two_ranks_weak_scaling_path():
  cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
         "--master-addr", "127.0.0.1", "--master-port", "29611", "bench.py", "--gpus", "2", "--steps", "2",
         "--warmup", "1", "--channels", "256", "--log2-samples", "13", "--backend", "gloo"]
  line = run(cmd)
  check(line, 2, 2, 1)
  assert "cpu_baseline" not in line          # rank 0 at N = 1 only
  assert line["config"]["channels_per_gpu"] == 256


def test_two_ranks_strong_scaling_path():
  cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
         "--master-addr", "127.0.0.1", "--master-port", "29613", "bench.py", "--gpus", "2", "--steps", "2",
         "--warmup", "1", "--channels", "512", "--log2-samples", "13", "--backend", "gloo", "--scaling", "strong"]
  line = run(cmd)
  check(line, 2, 2, 1)
  assert line["scaling"] == "strong" and line["config"]["channels_per_gpu"] == 256


def test_plain_gpus_2_launches_two_ranks_without_torchrun():
  # no launcher around it: bench.py starts its own ranks (they share this box's one GPU, hence gloo)
  env_clean = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
  out = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--channels", "256",
                        "--log2-samples", "13", "--backend", "gloo"], cwd=ROOT, env=env_clean, capture_output=True,
                       text=True, timeout=600)
  assert out.returncode == 0, out.stderr[-2000:]
  line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
  check(line, 2, 2, 1)
  assert [r["rank"] for r in line["per_rank"]] == [0, 1] and all(r["value"] > 0 for r in line["per_rank"])
  assert abs(sum(r["value"] for r in line["per_rank"]) - line["value"]) / line["value"] < 0.5
  col = line["secondary"]["downstream_collective"]
  assert col["ranks"] == 2 and col["parity"].startswith("collective results checked")


def test_one_rank_group_on_rccl():
  # init_process_group("nccl"), the barrier, the MAX all-reduce, the per-rank all_gather and the downstream
  # gather / mixdown of audiolazy_amd.sharding on CUDA tensors: all on RCCL, on the one GPU this box has
  line = run([sys.executable, "bench.py", "--gpus", "1", "--backend", "nccl", "--init-dist", "--steps", "2", "--warmup", "1",
              "--channels", "512", "--log2-samples", "14", "--no-cpu-baseline"])
  check(line, 1, 2, 1)
  assert line["process_group"] == {"backend": "nccl", "ranks": 1, "launcher": "none (--init-dist)"}
  assert len(line["per_rank"]) == 1 and abs(line["per_rank"][0]["value"] - line["value"]) / line["value"] < 1e-6
  col = line["secondary"]["downstream_collective"]
  assert col["backend"].startswith("nccl") and col["parity"].startswith("collective results checked")
  assert set(col["collectives"]) == {"mixdown_all_reduce", "gather_to_rank0", "mix_exact_to_rank0", "c_abi_direct_rccl"}
  # ("skipped" is what a direct-RCCL leg that threw reports: it must have RUN here)
  assert all(c["check"] == "ok" for c in col["collectives"].values()), col["collectives"]


def test_smoke_entry():
  out = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke()"], cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
  assert out.returncode == 0, out.stderr[-2000:]
