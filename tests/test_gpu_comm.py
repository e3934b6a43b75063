"""The C ABI's own RCCL binding (include/alz.h alz_comm_*, audiolazy_amd.sharding.DirectComm) on the one GPU of the
test box: a one-rank communicator, where gather / all-gather return the shard itself and the sums are exact.  (More
ranks need more GPUs: RCCL refuses two ranks on one device; the N > 1 launch of bench.py exercises them there.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_one_rank_communicator_gathers_and_sums():
  import torch
  from audiolazy_amd import sharding
  uid = sharding.DirectComm.unique_id()
  assert isinstance(uid, bytes) and len(uid) == 128
  comm = sharding.DirectComm(0, 1, uid, device=0)
  x = torch.from_numpy(np.random.default_rng(5).uniform(-1, 1, (48, 1000))).cuda()
  full = comm.gather(x)                 # all-gather
  assert tuple(full.shape) == (1, 48, 1000) and torch.equal(full[0], x)
  only = comm.gather(x, dst=0)          # gather to rank 0
  assert torch.equal(only[0], x)
  assert torch.equal(comm.sum(x), x) and torch.equal(comm.sum(x, dst=0), x)
  torch.cuda.synchronize()
  with pytest.raises(ValueError):
    comm.gather(x.float())
  comm.close()
  with pytest.raises(ValueError):
    sharding.DirectComm(3, 2, uid)
