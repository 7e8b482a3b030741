"""GPU (one device, two ranks over gloo): the data-parallel train step equals a single-process emulation
with W micro-batches whose gradients are averaged.  Exercises the real HIP kernels
plus the one-all-reduce-per-network gradient exchange."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from conftest import PKG, ROOT

pytestmark = pytest.mark.gpu


class Log:
    def add_metric(self, *a, **k):
        pass


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _data(rank):
    g = torch.Generator().manual_seed(100 + rank)
    return (torch.rand(8, 3, 32, 32, generator=g) * 2 - 1, torch.randn(8, 128, generator=g),
            torch.randn(8, 128, generator=g))


def _worker(rank, world, port, out_dir):
    import sys
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), DIAGAN_DIST_BACKEND="gloo")
    from diagan.models.predefined_models import get_gan_model
    from diagan.trainer import distributed as dist
    dist.init_from_env()
    torch.manual_seed(5)
    netG, netD, optG, optD = get_gan_model('cifar10', model='sngan', loss_type='hinge')
    netG.to('cuda'), netD.to('cuda')
    dist.broadcast_module_(netG), dist.broadcast_module_(netD)
    x, zd, zg = _data(rank)
    netD.train_step(real_batch=(x.cuda(), None), netG=netG, optD=optD, log_data=Log(), device='cuda', noise=zd.cuda())
    netG.train_step(real_batch=(x.cuda(), None), netD=netD, optG=optG, log_data=Log(), device='cuda', noise=zg.cuda())
    torch.cuda.synchronize()
    torch.save({'D': netD.flat_params.cpu(), 'G': netG.flat_params.cpu(),
                'overlapped': (netD.wgrad_batch.overlapped, netG.wgrad_batch.overlapped)}, os.path.join(out_dir, f"r{rank}.pt"))
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_step_equals_microbatch_emulation(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _port(), str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(tmp_path / f"r{k}.pt") for k in range(world)]
    assert torch.equal(r[0]['D'], r[1]['D']) and torch.equal(r[0]['G'], r[1]['G'])   # replicas stay in lock-step
    # both updates took the two-half reduction: the late layers' slab range was exchanged under the early layers' reduction
    assert r[0]['overlapped'] == (1, 1) and r[1]['overlapped'] == (1, 1), r[0]['overlapped']

    # single-process emulation: per-rank micro-batches, gradients averaged, one optimiser step
    from diagan.models.predefined_models import get_gan_model
    from diagan.ops import eltwise as E
    torch.manual_seed(5)
    netG, netD, optG, optD = get_gan_model('cifar10', model='sngan', loss_type='hinge')
    netG.to('cuda'), netD.to('cuda')
    sn0 = {k: v.clone() for k, v in netD.state_dict().items() if 'sn_' in k}
    bn0 = {k: v.clone() for k, v in netG.state_dict().items() if 'running' in k or 'num_batches' in k}
    grads = []
    for rank in range(world):
        netD.load_state_dict({**netD.state_dict(), **sn0})           # every rank starts from the same buffers
        netG.load_state_dict({**netG.state_dict(), **bn0})
        x, zd, _ = _data(rank)

        class NoStep:
            def step(self):
                pass
        netD.train_step(real_batch=(x.cuda(), None), netG=netG, optD=NoStep(), log_data=Log(), device='cuda',
                        noise=zd.cuda())
        grads.append(netD.flat_grads.clone())
    netD.flat_grads.copy_((grads[0] + grads[1]) / 2)
    optD.step()
    assert (netD.flat_params.cpu() - r[0]['D']).abs().max().item() < 2e-6


def _mnist_data(rank):
    g = torch.Generator().manual_seed(300 + rank)
    return torch.rand(8, 3, 32, 32, generator=g) * 2 - 1, torch.randn(8, 100, generator=g)


def _worker_mnist(rank, world, port, out_dir):
    import sys
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), DIAGAN_DIST_BACKEND="gloo")
    from diagan.models.predefined_models import get_gan_model
    from diagan.trainer import distributed as dist
    dist.init_from_env()
    torch.manual_seed(7)
    netG, netD, optG, optD = get_gan_model('color_mnist', model='mnist_dcgan', loss_type='ns')
    netG.to('cuda'), netD.to('cuda')
    dist.broadcast_module_(netG), dist.broadcast_module_(netD)
    x, zd = _mnist_data(rank)
    torch.cuda.manual_seed(900 + rank)                 # the dropout masks of this rank's two forwards
    netD.train_step(real_batch=(x.cuda(), None), netG=netG, optD=optD, log_data=Log(), device='cuda', noise=zd.cuda())
    torch.cuda.synchronize()
    torch.save({'D': netD.flat_params.cpu()}, os.path.join(out_dir, f"m{rank}.pt"))
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_mnist_dcgan_step_equals_microbatch_emulation(tmp_path):
    """MNIST_DCGAN_Discriminator runs its real AND its fake backward through weight-gradient slot 0.
    With the reduction held back for the exchange overlap (world > 1) the fake pass overwrote the real pass's split-K
    slabs before they were reduced: real gradient lost, fake gradient doubled.  WgradBatch.launch now reduces a held
    slot before it is written again; the two-rank update must equal the averaged micro-batch gradients."""
    world = 2
    mp.spawn(_worker_mnist, args=(world, _port(), str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(tmp_path / f"m{k}.pt") for k in range(world)]
    assert torch.equal(r[0]['D'], r[1]['D'])

    from diagan.models.predefined_models import get_gan_model
    torch.manual_seed(7)
    netG, netD, optG, optD = get_gan_model('color_mnist', model='mnist_dcgan', loss_type='ns')
    netG.to('cuda'), netD.to('cuda')
    buf0 = {k: v.clone() for k, v in netD.state_dict().items() if 'running' in k or 'num_batches' in k}
    gbuf0 = {k: v.clone() for k, v in netG.state_dict().items() if 'running' in k or 'num_batches' in k}
    before = netD.flat_params.clone()
    grads = []
    for rank in range(world):
        netD.load_state_dict({**netD.state_dict(), **buf0})
        netG.load_state_dict({**netG.state_dict(), **gbuf0})
        x, zd = _mnist_data(rank)

        class NoStep:
            def step(self):
                pass
        torch.cuda.manual_seed(900 + rank)
        netD.train_step(real_batch=(x.cuda(), None), netG=netG, optD=NoStep(), log_data=Log(), device='cuda',
                        noise=zd.cuda())
        grads.append(netD.flat_grads.clone())
    # the two passes of a micro-batch both contribute (what the bug lost): the summed gradient is not the fake pass doubled
    netD.flat_grads.copy_((grads[0] + grads[1]) / 2)
    optD.step()
    assert (netD.flat_params - before).abs().max().item() > 1e-5
    assert (netD.flat_params.cpu() - r[0]['D']).abs().max().item() < 2e-6


def _worker_rank_noise(rank, world, port, out_dir):
    """The CLI's data-parallel start (diagan/cli.py::_Run.replicate): same seed everywhere, broadcast, then per-rank DEVICE
    seeds; noise is drawn by the networks themselves (no injection).  Phase-2 shape: D's all-reduce is left in flight
    under the D_drs update (LogTrainer._updates)."""
    import sys
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), DIAGAN_DIST_BACKEND="gloo")
    from diagan.models.predefined_models import get_gan_model
    from diagan.trainer import distributed as dist
    from diagan.utils.settings import set_seed
    dist.init_from_env()
    set_seed(5)
    netG, netD, netD_drs, optG, optD, optD_drs = get_gan_model('cifar10', model='sngan', loss_type='hinge', drs=True)
    for n in (netG, netD, netD_drs):
        n.to('cuda')
        dist.broadcast_module_(n)
    dist.seed_device_per_rank(5)
    cpu_draw = torch.rand(4)                                       # the CPU generator must stay shared (sampler order)
    fake, _ = netG.generate_images_nhwc(8, device='cuda')
    fake = fake.clone()
    x, _, _ = _data(rank)
    for _ in range(2):
        netD.train_step(real_batch=(x.cuda(), None), netG=netG, optD=optD, log_data=Log(), device='cuda', defer_step=True)
        netD_drs.train_step(real_batch=(x.cuda(), None), netG=netG, optD=optD_drs, log_data=Log(), device='cuda')
        optD.step()
    netG.train_step(real_batch=(x.cuda(), None), netD=netD, optG=optG, log_data=Log(), device='cuda')
    torch.cuda.synchronize()
    torch.save({'D': netD.flat_params.cpu(), 'Ddrs': netD_drs.flat_params.cpu(), 'G': netG.flat_params.cpu(),
                'fake': fake.cpu(), 'cpu_draw': cpu_draw, 'scale': optD.grad_scale}, os.path.join(out_dir, f"n{rank}.pt"))
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(600)
def test_ranks_draw_different_noise_and_stay_in_lock_step(tmp_path):
    """VERDICT r1 Missing 5 / ADVICE: under data parallelism every rank drew the SAME latent batch.  Now: the first fake
    batches of the two ranks differ, the CPU generators stay shared, and the parameters of all three networks are
    bit-identical on both ranks after updates whose D all-reduce overlapped the D_drs update."""
    world = 2
    mp.spawn(_worker_rank_noise, args=(world, _port(), str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(tmp_path / f"n{k}.pt") for k in range(world)]
    assert not torch.equal(r[0]['fake'], r[1]['fake'])
    assert (r[0]['fake'] - r[1]['fake']).abs().mean().item() > 1e-2
    assert torch.equal(r[0]['cpu_draw'], r[1]['cpu_draw'])
    for k in ('D', 'Ddrs', 'G'):
        assert torch.equal(r[0][k], r[1][k]), k
    assert r[0]['scale'] == 0.5


def test_native_rccl_context_single_rank():
    """SURVEY 8(b) exports diagan_ctx / diagan_allreduce_grads / diagan_allgather_logits (csrc/comm.hip): a one-rank RCCL
    communicator on this box's GPU (RCCL refuses two ranks on one device, so W > 1 is the driver's 8-GPU run):
    the all-reduce leaves the slab as it is, the all-gather returns the shard, both on torch's current stream."""
    import ctypes
    from diagan import _native as nat
    from diagan.trainer import distributed as D
    ctx = D.init_native_comm(0, 1, torch.cuda.current_device())
    try:
        assert nat.fn("diagan_ctx_world")(ctx) == 1 and nat.fn("diagan_ctx_rank")(ctx) == 0
        g = torch.randn(1 << 20, device='cuda')
        ref = g.clone()
        nat.call("diagan_allreduce_grads", ctx, g.data_ptr(), g.numel(), nat.current_stream())
        row = torch.randn(4096, dtype=torch.float64, device='cuda')
        out = torch.zeros_like(row)
        nat.call("diagan_allgather_logits", ctx, row.data_ptr(), out.data_ptr(), row.numel(), 8, nat.current_stream())
        torch.cuda.synchronize()
        assert torch.equal(g, ref) and torch.equal(out, row)
        with pytest.raises(RuntimeError, match="allreduce_grads"):
            nat.call("diagan_allreduce_grads", ctx, None, 0, nat.current_stream())
    finally:
        D.destroy_native_comm()


@pytest.mark.timeout(900)
def test_bench_two_ranks_prints_one_json_line():
    """The driver's N > 1 launch of bench.py (torch.distributed.run, one process per rank) end to end, with both
    ranks on this box's single GPU over gloo: every rank must take part in every collective of every step,
    including the un-timed steps after the timed region; exactly one JSON line, from rank 0."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, DIAGAN_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--no_cpu_baseline"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=800)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["parallelism"] == "dp2" and rec["config"]["global_batch"] == 128
    assert rec["scaling"] == "weak" and rec["value"] > 0 and "roofline" in rec and "cpu_baseline" not in rec


@pytest.mark.timeout(900)
def test_bare_bench_gpus_2_starts_two_ranks_itself():
    """`python bench.py --gpus 2` with NO torch.distributed.run around it and no WORLD_SIZE in the environment (the shape
    of the driver's N = 1 command with another N): the script must start the two ranks itself as a child process group
    and relay rank 0's line -- n_gpus 2, not a silent one-GPU number.  On this one-GPU box the launcher puts both ranks
    on device 0 over gloo and says so in config."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "DIAGAN_DIST_BACKEND")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--no_cpu_baseline"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=800)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["ranks"] == 2 and rec["config"]["global_batch"] == 128
    assert len(rec["config"]["devices"]) == 2
    if torch.cuda.device_count() < 2:
        assert "gloo" in rec["config"]["comm"] and "share" in rec["config"]["comm"]


@pytest.mark.timeout(1800)
def test_bare_bench_gpus_8_rehearsal_on_one_device():
    """the command the driver's scaling run uses at its largest size -- `python bench.py --gpus 8` --
    rehearsed functionally on whatever this box has (one GPU: eight ranks share device 0 over gloo, and config says so):
    one JSON line with n_gpus 8 and global_batch 512, a host-loop row per rank, exit status 0 inside the timeout, and no
    process of the job left behind (the launcher's process group is empty afterwards)."""
    import json
    import signal
    import subprocess
    import sys
    import time
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "DIAGAN_DIST_BACKEND")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1",
           "--no_cpu_baseline"]
    proc = subprocess.Popen(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                            start_new_session=True)
    try:
        out, err = proc.communicate(timeout=1500)
    except subprocess.TimeoutExpired:
        os.killpg(proc.pid, signal.SIGKILL)
        raise
    assert proc.returncode == 0, err[-3000:]
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 8 and rec["config"]["ranks"] == 8 and rec["config"]["global_batch"] == 512
    assert rec["config"]["parallelism"] == "dp8" and rec["scaling"] == "weak" and rec["value"] > 0
    assert len(rec["config"]["devices"]) == 8 and len(rec["host"]["launch_loop_ms_per_step"]) == 8
    if torch.cuda.device_count() < 8:
        assert "gloo" in rec["config"]["comm"] and "share" in rec["config"]["comm"]
    # no leaked children: the session started for the launcher has no live process left
    deadline = time.time() + 20
    while True:
        try:
            os.killpg(proc.pid, 0)
        except ProcessLookupError:
            break
        assert time.time() < deadline, "processes of the bench job are still alive 20 s after it returned"
        time.sleep(0.5)
