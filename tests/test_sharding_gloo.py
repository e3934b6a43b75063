"""The N > 1 path on CPU: world_size-2 (and 3, ragged) gloo process groups.  Every rank
"filters" its channel shard with the oracle (test infrastructure standing in for the GPU),
then the sharding module's collectives must reproduce the single-process result exactly."""
import os
import socket

import numpy as np
import pytest

from audiolazy_amd import sharding


def test_shard_range_partitions_exactly():
  for n in (0, 1, 7, 64, 4096, 4099):
    for world in (1, 2, 3, 8):
      spans = [sharding.shard_range(n, world, r) for r in range(world)]
      assert spans[0][0] == 0 and spans[-1][1] == n
      assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
      sizes = [b - a for a, b in spans]
      assert max(sizes) - min(sizes) <= 1 and sizes == sharding.shard_sizes(n, world)
  with pytest.raises(ValueError):
    sharding.shard_range(10, 2, 2)


def _free_port():
  s = socket.socket()
  s.bind(("127.0.0.1", 0))
  port = s.getsockname()[1]
  s.close()
  return port


def _worker(rank, world, port, C, N, ret):
  import torch
  import torch.distributed as dist
  from oracle import oracle
  os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
  dist.init_process_group("gloo", rank=rank, world_size=world)
  try:
    rng = np.random.default_rng(42)                 # same data on every rank
    x = rng.uniform(-1, 1, (N, C))
    w = 2 * np.pi * np.geomspace(50., 20000., C) / 48000.
    r = np.exp(-w / 20.)
    a = np.stack([np.ones(C), -2 * r * np.cos(w), r * r], axis=1)
    b = np.stack([(1 - r * r) / 2, np.zeros(C), -(1 - r * r) / 2], axis=1)
    start, stop = sharding.shard_range(C, world, rank)
    y_local = oracle.bank([3], [3], b[start:stop], a[start:stop], np.ascontiguousarray(x[:, start:stop]))
    full = sharding.gather_channels(torch.from_numpy(y_local), C, channel_dim=1)
    ref = oracle.bank([3], [3], b, a, x)
    ok = np.array_equal(full.numpy().view(np.uint64), ref.view(np.uint64))
    # channel-major shards gather along dim 0
    full_cm = sharding.gather_channels(torch.from_numpy(np.ascontiguousarray(y_local.T)), C, channel_dim=0)
    ok = ok and np.array_equal(full_cm.numpy(), ref.T)
    # gather to one rank only
    only0 = sharding.gather_channels(torch.from_numpy(y_local), C, channel_dim=1, dst=0)
    ok = ok and ((only0 is None) == (rank != 0))
    if rank == 0:
      ok = ok and np.array_equal(only0.numpy(), ref)
    # mixdown over shards == sum over all channels' shard sums (floating point: allclose)
    part = torch.from_numpy(y_local.sum(axis=1))
    mix = sharding.mixdown(part)
    ok = ok and np.allclose(mix.numpy(), ref.sum(axis=1), rtol=1e-12, atol=1e-12)
    # ... and the ordered mix: ((c0 + c1) + c2) ... over the GLOBAL channel index (ParallelFilter's order,
    # lazy_filters.py:1048-1054), bit for bit whatever the number of ranks
    acc = ref[:, 0].copy()
    for ch in range(1, C):
      acc = acc + ref[:, ch]
    exact = sharding.mix_exact(torch.from_numpy(y_local), C, channel_dim=1)
    ok = ok and np.array_equal(exact.numpy().view(np.uint64), acc.view(np.uint64))
    exact0 = sharding.mix_exact(torch.from_numpy(np.ascontiguousarray(y_local.T)), C, channel_dim=0, dst=0)
    ok = ok and ((exact0 is None) == (rank != 0))
    if rank == 0:
      ok = ok and np.array_equal(exact0.numpy().view(np.uint64), acc.view(np.uint64))
    ret[rank] = bool(ok)
  finally:
    dist.destroy_process_group()


@pytest.mark.parametrize("world,C", [(2, 64), (3, 50)])
def test_gloo_gather_matches_single_process(world, C):
  import torch.multiprocessing as mp
  port = _free_port()
  ctx = mp.get_context("spawn")
  with ctx.Manager() as mgr:
    ret = mgr.dict()
    procs = [ctx.Process(target=_worker, args=(r, world, port, C, 200, ret)) for r in range(world)]
    for p in procs:
      p.start()
    for p in procs:
      p.join(timeout=180)
      assert p.exitcode == 0
    assert dict(ret) == {r: True for r in range(world)}


def test_single_process_is_identity():
  import torch
  t = torch.arange(6.0).reshape(2, 3)
  assert sharding.gather_channels(t, 3) is t and sharding.mixdown(t) is t
  assert sharding.world_info()[1] >= 1
