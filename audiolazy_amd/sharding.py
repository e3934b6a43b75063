"""Channel sharding across the GPUs of a node (one process per GPU).

The filter path has no exchange step: channels (cfg2/3), input streams of a filterbank
(cfg4) and frames (cfg5) are independent units, so every rank filters its own contiguous
shard with no data-path collective.  The only collective offered is the one
a *downstream* consumer may need -- all channels of a block on every rank / on one rank, or
the ParallelFilter-style sum over bands -- as a single torch.distributed call (backend
"nccl" is RCCL over xGMI on ROCm; "gloo" on CPU for tests).

Nothing here touches the reference's algorithms: the reference is single-process
(lazy_stream.py:114 "not thread-safe"), this module is the MI355X-side scale-out.
"""
import os


def world_info():
  """(rank, world_size, local_rank) from the torchrun environment (defaults: single process)."""
  return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
          int(os.environ.get("LOCAL_RANK", "0")))


def shard_range(n_units, world, rank):
  """Contiguous balanced shard [start, stop) of ``n_units`` for ``rank``: the first
  ``n_units % world`` ranks get one extra unit; every unit belongs to exactly one rank."""
  if world < 1 or not 0 <= rank < world:
    raise ValueError("bad rank/world")
  base, extra = divmod(n_units, world)
  start = rank * base + min(rank, extra)
  return start, start + base + (1 if rank < extra else 0)


def shard_sizes(n_units, world):
  return [shard_range(n_units, world, r)[1] - shard_range(n_units, world, r)[0] for r in range(world)]


def local_bank(sections, n_channels, rank=None, world=None, device=None, **kwargs):
  """The FilterBank for this rank's shard of a DIAGONAL bank.

  sections: [(b, a), ...] with b/a either shared ([nb]) or per channel ([n_channels, nb]);
  per-channel rows are sliced to the shard.  Returns (bank, (start, stop)).
  """
  import numpy as np
  from .bank import FilterBank
  r, w, local = world_info()
  rank = r if rank is None else rank
  world = w if world is None else world
  device = local if device is None else device
  start, stop = shard_range(n_channels, world, rank)
  secs = []
  for b, a in sections:
    b, a = np.asarray(b, dtype=np.float64), np.asarray(a, dtype=np.float64)
    secs.append((b[start:stop] if b.ndim == 2 else b, a[start:stop] if a.ndim == 2 else a))
  return FilterBank(secs, n_inputs=stop - start, device=device, **kwargs), (start, stop)


def gather_channels(y_local, n_channels, channel_dim=-1, group=None, dst=None, force=False):
  """Assemble the full-width block from the per-rank shards along ``channel_dim``.

  y_local : torch tensor holding this rank's channels [.., stop - start, ..].
  dst None -> every rank gets the full block (all_gather); dst = r -> only rank r does
  (gather; others get None).  Shards may be ragged: they are padded to the largest shard
  for the collective and trimmed afterwards.  A one-rank group returns ``y_local`` itself
  unless ``force`` asks for the collective call anyway (bench.py --init-dist: the RCCL path on
  a one-GPU box).
  """
  import torch
  import torch.distributed as dist
  if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
    return y_local
  world, rank = dist.get_world_size(group), dist.get_rank(group)
  sizes = shard_sizes(n_channels, world)
  biggest = max(sizes)
  moved = y_local.movedim(channel_dim, 0).contiguous()
  if moved.shape[0] != sizes[rank]:
    raise ValueError("rank %d holds %d channels, its shard has %d" % (rank, moved.shape[0], sizes[rank]))
  if moved.shape[0] < biggest:
    pad = torch.zeros((biggest - moved.shape[0],) + tuple(moved.shape[1:]), dtype=moved.dtype, device=moved.device)
    moved = torch.cat([moved, pad])
  if dst is None:
    parts = [torch.empty_like(moved) for _ in range(world)]
    dist.all_gather(parts, moved, group=group)
  else:
    parts = [torch.empty_like(moved) for _ in range(world)] if rank == dst else None
    dist.gather(moved, parts, dst=dst, group=group)
    if rank != dst:
      return None
  full = torch.cat([p[:n] for p, n in zip(parts, sizes)])
  return full.movedim(0, channel_dim)


def mixdown(y_local, group=None, dst=None, force=False):
  """Sum of every rank's block (ParallelFilter / Streamix-style mix over shards): all_reduce,
  or reduce to ``dst``.  Note the summation order across ranks is the collective's, so this is
  floating-point (not bit-exact) with respect to a single-process left-to-right sum."""
  import torch.distributed as dist
  if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
    return y_local
  out = y_local.clone()
  if dst is None:
    dist.all_reduce(out, op=dist.ReduceOp.SUM, group=group)
    return out
  dist.reduce(out, dst=dst, op=dist.ReduceOp.SUM, group=group)
  return out if dist.get_rank(group) == dst else None


def mix_exact(y_local, n_channels, channel_dim=-1, group=None, dst=None, force=False):
  """The mix of EVERY channel of a sharded block in the reference's own order -- ``((c0 + c1) + c2) + ...`` over the
  global channel index, what ``ParallelFilter.__call__`` computes (reference lazy_filters.py:1048-1054:
  ``reduce(operator.add, ...)``) -- bit-identical to the single-process sum whatever the number of ranks.  One
  gather of the shards (to ``dst``, or to every rank) followed by the ordered sum on the device (alz_mix_dev); the
  all-reduce form, :func:`mixdown`, moves 1 / C of the bytes but sums in the collective's order.  Returns the mixed
  block with ``channel_dim`` removed (None on the ranks that are not ``dst``)."""
  full = gather_channels(y_local, n_channels, channel_dim=channel_dim, group=group, dst=dst, force=force)
  if full is None:
    return None
  moved = full.movedim(channel_dim, 0)
  if moved.shape[0] != n_channels:            # (no process group: the local block is the whole block)
    n_channels = moved.shape[0]
  rest = tuple(moved.shape[1:])
  if moved.is_cuda:
    from .bank import mix_sets
    flat = moved.reshape(n_channels, -1).contiguous()
    return mix_sets(flat, n_channels, 1, layout="chan").reshape(rest)
  acc = moved[0].clone()
  for ch in range(1, n_channels):             # CPU tensors (the gloo test mode): the same order, one add per channel
    acc = acc + moved[ch]
  return acc


class DirectComm(object):
  """The same single collective through the C ABI's own RCCL binding (include/alz.h: alz_comm_*) -- for callers
  that drive the engine without torch.distributed.  One process per GPU; rank 0 makes the 128-byte id
  (``DirectComm.unique_id()``) and hands it to the other ranks by whatever started them (an environment variable,
  a file, MPI, a torch.distributed broadcast); every rank then constructs the communicator (collective call).

  gather(x, dst=None): every rank's contiguous float64 CUDA tensor -> [world, *x.shape] on ``dst`` (on every rank
  when None); rank-major order, so channel-major shards [C / G, N] concatenate into the [C, N] block.
  sum(x, dst=None): element-wise sum over the ranks (the mix of per-rank partial mixes)."""

  def __init__(self, rank, world, unique_id, device=None):
    import ctypes
    from . import _ffi
    if len(unique_id) != 128:
      raise ValueError("the RCCL unique id is 128 bytes")
    self._L, self.rank, self.world = _ffi.load(), int(rank), int(world)
    self.device = world_info()[2] if device is None else int(device)
    self._h = ctypes.c_void_p()
    buf = ctypes.create_string_buffer(bytes(unique_id), 128)
    _ffi.check(self._L.alz_comm_create(self.device, self.world, self.rank, buf, ctypes.byref(self._h)))

  @staticmethod
  def unique_id():
    import ctypes
    from . import _ffi
    buf = ctypes.create_string_buffer(128)
    _ffi.check(_ffi.load().alz_comm_unique_id(buf))
    return bytes(buf.raw)

  def _check(self, x):
    import torch
    if not (x.is_cuda and x.dtype == torch.float64 and x.is_contiguous()):
      raise ValueError("contiguous float64 CUDA tensors only")

  def gather(self, x, dst=None):
    import ctypes
    import torch
    from . import _ffi
    self._check(x)
    here = dst is None or dst == self.rank
    out = torch.empty((self.world,) + tuple(x.shape), dtype=torch.float64, device=x.device) if here else None
    stream = torch.cuda.current_stream(x.device).cuda_stream
    _ffi.check(self._L.alz_comm_gather(self._h, x.data_ptr(), out.data_ptr() if here else None, x.numel(),
                                       -1 if dst is None else int(dst), ctypes.c_void_p(stream)))
    return out

  def sum(self, x, dst=None):
    import ctypes
    import torch
    from . import _ffi
    self._check(x)
    here = dst is None or dst == self.rank
    out = torch.empty_like(x) if here else None
    stream = torch.cuda.current_stream(x.device).cuda_stream
    _ffi.check(self._L.alz_comm_sum(self._h, x.data_ptr(), out.data_ptr() if here else None, x.numel(),
                                    -1 if dst is None else int(dst), ctypes.c_void_p(stream)))
    return out

  def close(self):
    h, self._h = self._h, None
    if h:
      self._L.alz_comm_destroy(h)

  def __del__(self):
    try:
      self.close()
    except Exception:
      pass
