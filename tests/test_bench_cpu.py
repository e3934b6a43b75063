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
  assert cpu["kind"] == "port" and cpu["unit"] == "Gsamples/s" and cpu["host_logical_cpus"] == os.cpu_count()
  legs = cpu["legs"]
  assert set(legs) == {"py_1proc", "py_pool", "py_rows", "c_port"}
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
