"""CPU, world_size 2 over gloo: the collectives of the data-parallel path (gradient mean over one
flat slab, logit shard all-gather, sharded samplers).  127.0.0.1 rendezvous."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import PKG, ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    import sys
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from diagan.trainer import distributed as dist
    r, lr, w = dist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and dist.get_rank() == rank and dist.get_world_size() == world
    res = {}
    # 1. gradient slab: mean over ranks, in place
    flat = torch.arange(10, dtype=torch.float32) * (rank + 1)
    dist.all_reduce_mean_(flat)
    res['mean'] = flat.clone()
    # 2. logit record shard gather: N = 11 not divisible by 2
    n = 11
    lo, hi, per = dist.shard_range(n)
    row = torch.zeros(n, dtype=torch.float64)
    truth = torch.arange(n, dtype=torch.float64) * 0.5 - 2.0      # includes -0.0-free negatives
    row[lo:hi] = truth[lo:hi]
    res['row'] = dist.gather_row_shards(row, n)
    res['range'] = (lo, hi, per)
    # 3. (idx, logit) all-gather variant (stylegan2/train_ffhq.py:139-141)
    idx = torch.tensor([rank, rank + 2, rank + 4])
    res['idx_all'] = dist.all_gather_cat(idx)
    # 4. data-parallel gradient == mean of per-rank micro-batch gradients (oracle D on CPU)
    from oracle import nets as O
    torch.manual_seed(0)
    netD = O.MNIST_DCGAN_Discriminator(loss_type='hinge')
    netD.eval()
    g = torch.Generator().manual_seed(100 + rank)
    x = torch.rand(2, 3, 32, 32, generator=g) * 2 - 1
    netD.zero_grad()
    netD(x).mean().backward()
    slab = torch.cat([p.grad.reshape(-1) for p in netD.parameters()])
    res['local_grad'] = slab.clone()
    dist.all_reduce_mean_(slab)
    res['dp_grad'] = slab
    # 5. samplers: same seed on every rank -> disjoint strided shards of one draw
    from diagan.datasets.sampler import ShardedSampler, make_weighted_sampler
    torch.manual_seed(7)
    res['shard'] = list(iter(ShardedSampler(make_weighted_sampler(np.linspace(0.1, 1, 20)), rank, world)))
    # 6. scalar reduction helpers
    res['sum'] = dist.reduce_sum(torch.tensor([float(rank + 1)]))
    res['loss'] = dist.reduce_loss_dict({'d': torch.tensor(float(rank)), 'g': torch.tensor(2.0 * rank)})
    # 7. broadcast of a module
    lin = torch.nn.Linear(3, 2)
    with torch.no_grad():
        lin.weight.fill_(float(rank + 1))
    dist.broadcast_module_(lin)
    res['bcast'] = lin.weight.clone()
    # 8. the reference's helper names: per-parameter gradient mean and the pickled-object gather
    lin2 = torch.nn.Linear(2, 1)
    lin2.weight.grad = torch.full((1, 2), float(rank + 1))
    dist.gather_grad(lin2.parameters())                 # bias has no gradient: skipped
    res['gather_grad'] = lin2.weight.grad.clone()
    res['objects'] = dist.all_gather({'rank': rank, 'payload': 'x' * (rank + 1)})
    # 9. BatchNorm running statistics reconciled at checkpoint time (mean over ranks; other buffers untouched)
    from diagan.models.layers import BatchNorm
    holder = torch.nn.Module()
    holder.b1, holder.b2 = BatchNorm(4), BatchNorm(2)
    holder.register_buffer('sn_u', torch.full((1, 3), float(rank)))
    with torch.no_grad():
        holder.b1.running_mean.fill_(float(rank)), holder.b1.running_var.fill_(1.0 + 2 * rank)
        holder.b2.running_mean.copy_(torch.tensor([1.0, -1.0]) * (rank + 1))
    holder.b1._pending_batches = 3
    res['reconciled'] = dist.reconcile_running_stats_(holder)
    res['bn'] = (holder.b1.running_mean.clone(), holder.b1.running_var.clone(), holder.b2.running_mean.clone(),
                 holder.sn_u.clone(), holder.state_dict()['b1.num_batches_tracked'].item())
    dist.synchronize()
    torch.save(res, os.path.join(out_dir, f"r{rank}.pt"))
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(300)
def test_world_size_2_gloo(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(tmp_path / f"r{k}.pt", weights_only=False) for k in range(world)]
    exp_mean = torch.arange(10, dtype=torch.float32) * 1.5
    truth = torch.arange(11, dtype=torch.float64) * 0.5 - 2.0
    for k in range(world):
        assert torch.equal(r[k]['mean'], exp_mean)
        assert torch.equal(r[k]['row'], truth)                       # bit-exact index -> slot placement
        assert torch.equal(r[k]['idx_all'], torch.tensor([0, 2, 4, 1, 3, 5]))   # rank-major concat
        assert torch.allclose(r[k]['dp_grad'], (r[0]['local_grad'] + r[1]['local_grad']) / 2, rtol=0, atol=1e-7)
        assert r[k]['sum'].item() == 3.0
        assert torch.equal(r[k]['bcast'], torch.ones(2, 3))
        assert torch.equal(r[k]['gather_grad'], torch.full((1, 2), 1.5))
        assert r[k]['objects'] == [{'rank': 0, 'payload': 'x'}, {'rank': 1, 'payload': 'xx'}]
        assert r[k]['reconciled'] == 4
        m1, v1, m2, u, nb = r[k]['bn']
        assert torch.equal(m1, torch.full((4,), 0.5)) and torch.equal(v1, torch.full((4,), 2.0))
        assert torch.equal(m2, torch.tensor([1.5, -1.5])) and torch.equal(u, torch.full((1, 3), float(k))) and nb == 3
    assert r[0]['range'] == (0, 6, 6) and r[1]['range'] == (6, 11, 6)
    assert r[0]['loss']['d'].item() == 0.5 and r[0]['loss']['g'].item() == 1.0
    # strided shards of the same multinomial draw
    from diagan.datasets.sampler import make_weighted_sampler
    torch.manual_seed(7)
    full = list(iter(make_weighted_sampler(np.linspace(0.1, 1, 20))))
    assert r[0]['shard'] == full[0::2] and r[1]['shard'] == full[1::2]


def test_shard_range_edges():
    from diagan.trainer import distributed as dist
    assert [dist.shard_range(10, r, 4) for r in range(4)] == [(0, 3, 3), (3, 6, 3), (6, 9, 3), (9, 10, 3)]
    assert [dist.shard_range(2, r, 4) for r in range(4)] == [(0, 1, 1), (1, 2, 1), (2, 2, 1), (2, 2, 1)]
    assert dist.shard_range(7, 0, 1) == (0, 7, 7)
