"""Process-group helpers: one process per GPU, torch.distributed over RCCL/xGMI ('nccl' backend on
ROCm) or gloo on CPU.

Same helper names as the reference (diagan-pkg/diagan/trainer/distributed.py:8-126 ==
stylegan2/distributed.py) plus the two collectives the SNGAN data-parallel path needs:
`all_reduce_mean_` (gradient slab, pattern of DDP at stylegan2/train_ffhq.py:572-585) and
`all_gather_cat` (per-sample logits, pattern of concat_all_gather at train_ffhq.py:150-161).
"""
import os

import torch
import torch.distributed as dist


# ---- native exchange: RCCL behind the C ABI (csrc/comm.hip) ---------------------------------------------------------
# The gradient all-reduce and the logit all-gather go through diagan_allreduce_grads / diagan_allgather_logits on an
# explicit diagan_ctx (stream-ordered launches: no host wait, capturable in a hipGraph) instead of torch.distributed's
# process group, which then only carries the 128-byte unique id, the barriers and the start-up broadcast.
# Policy (DIAGAN_COMM): "rccl" = native or fail; "torch" = process group only; unset / "auto" = native whenever every
# rank has a device of its own (backend nccl and device_count() >= ranks on this node) AND the context passes a
# start-up self-test against the process group's all-reduce on every rank -- otherwise the process group, with the
# reason on stderr (RCCL refuses two ranks on one device, so single-GPU test boxes always take the process group).
_NATIVE = {'ctx': None, 'side': None, 'why': 'single process'}


def native_ctx():
    return _NATIVE['ctx']


def comm_description():
    """what carries the gradient exchange of this process (bench.py's config.comm)"""
    if get_world_size() == 1:
        return "none (single process)"
    if _NATIVE['ctx'] is not None:
        return "rccl-native"
    return f"torch.distributed/{dist.get_backend()} ({_NATIVE['why']})"


def _register_native():
    from diagan import _native as nat
    P, I, I64 = nat.c_void_p, nat.c_int, nat.c_i64
    for name, sig in (("diagan_comm_unique_id", [P]), ("diagan_ctx_create", [P, P, I, I, I]), ("diagan_ctx_destroy", [P]),
                      ("diagan_ctx_rank", [P]), ("diagan_ctx_world", [P]), ("diagan_allreduce_grads", [P, P, I64, P]),
                      ("diagan_allgather_logits", [P, P, P, I64, I, P])):
        nat.register(name, sig)
    return nat


def init_native_comm(rank, world, device_index):
    """Create the process's diagan_ctx: rank 0 draws the unique id, every rank receives it (torch.distributed object
    broadcast when a process group exists, else world must be 1), all ranks enter ncclCommInitRank together.
    The context is destroyed at interpreter exit (`destroy_native_comm`, atexit)."""
    import atexit
    import ctypes
    nat = _register_native()
    ident = ctypes.create_string_buffer(128)
    id_err = None
    if rank == 0:
        try:
            nat.call("diagan_comm_unique_id", ctypes.cast(ident, ctypes.c_void_p))
        except Exception as e:      # noqa: BLE001 -- the other ranks sit in the broadcast below: tell them
            id_err = repr(e)
    if world > 1:
        box = [None if id_err else ident.raw]
        dist.broadcast_object_list(box, src=0)
        if box[0] is None:
            raise RuntimeError("rank 0 could not draw the RCCL unique id" + (f": {id_err}" if id_err else ""))
        ident = ctypes.create_string_buffer(box[0], 128)
    elif id_err:
        raise RuntimeError(id_err)
    handle = ctypes.c_void_p()
    nat.call("diagan_ctx_create", ctypes.cast(ctypes.pointer(handle), ctypes.c_void_p), ctypes.cast(ident, ctypes.c_void_p),
             rank, world, device_index)
    if _NATIVE['ctx'] is None:
        atexit.register(destroy_native_comm)
    _NATIVE['ctx'] = handle
    _NATIVE['why'] = 'native'
    return handle


def destroy_native_comm():
    if _NATIVE['ctx'] is not None:
        nat = _register_native()
        if torch.cuda.is_available():
            torch.cuda.synchronize()        # nothing of ours may still be queued on the communicator
        nat.call("diagan_ctx_destroy", _NATIVE['ctx'])
        _NATIVE['ctx'] = None
        _NATIVE['side'] = None


def _all_ranks_agree(ok):
    """True iff `ok` on every rank (one tiny MIN all-reduce on the process group)"""
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device='cuda' if dist.get_backend() == 'nccl' else 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item())


def _native_selftest(rank, world):
    """The native context against the process group on the same data: a 1 Mi-element SUM all-reduce (values that sum
    exactly in any order) on the current stream, the same all-reduce issued on the SIDE stream the way FlatNet.sync_grads
    issues it under data parallelism (two slab halves in flight beside compute on the current stream), and a float64
    all-gather.  Collectives of the two communicators are never in flight together: a device synchronisation separates
    every native collective from the process group's (two RCCL communicators whose kernels the ranks may order
    differently are the known deadlock).  Returns an error string or None."""
    try:
        from diagan import _native as nat
        n = 1 << 20
        a = (torch.arange(n, device='cuda', dtype=torch.float32) % 1024) + rank
        b, c = a.clone(), a.clone()
        torch.cuda.synchronize()
        dist.all_reduce(b, op=dist.ReduceOp.SUM)
        torch.cuda.synchronize()
        nat.call("diagan_allreduce_grads", _NATIVE['ctx'], a.data_ptr(), a.numel(), nat.current_stream())
        # the overlapped form: late half, compute on the current stream, early half, then both waits
        cut = n // 2
        works = [_native_allreduce(c[cut:], True)]
        busy = torch.ones(1 << 22, device='cuda').mul_(2.0).sum()
        works.append(_native_allreduce(c[:cut], True))
        for w in works:
            w.wait()
        row = torch.full((4096,), float(rank), dtype=torch.float64, device='cuda')
        out = torch.empty(world * 4096, dtype=torch.float64, device='cuda')
        nat.call("diagan_allgather_logits", _NATIVE['ctx'], row.data_ptr(), out.data_ptr(), row.numel(), 8,
                 nat.current_stream())
        torch.cuda.synchronize()
        want = torch.arange(world, device='cuda', dtype=torch.float64).repeat_interleave(4096)
        if not torch.equal(a, b):
            return "native all-reduce disagrees with the process group's"
        if not torch.equal(c, b) or float(busy) != float(2 << 22):
            return "native side-stream all-reduce (the overlapped gradient exchange) disagrees with the process group's"
        if not torch.equal(out, want):
            return "native all-gather returned the wrong rows"
        return None
    except Exception as e:      # noqa: BLE001 -- any failure here means: use the process group
        return repr(e)


# exit status of a rank whose native-context start-up did not finish in time (bench.py's launcher starts the job again
# on the process group when it sees it: a process that has touched the GPU is never re-exec'ed)
NATIVE_INIT_TIMEOUT_EXIT = 75


class _StartupWatchdog:
    """ncclCommInitRank and the first collectives block inside native code until EVERY rank has arrived; a rank that died
    before it got there leaves the others waiting for good.  While this is armed, a process that is still inside after
    DIAGAN_COMM_INIT_TIMEOUT seconds (default 180) writes the reason to stderr and leaves with NATIVE_INIT_TIMEOUT_EXIT."""

    def __init__(self, rank, what):
        import threading
        self.secs = float(os.environ.get("DIAGAN_COMM_INIT_TIMEOUT", "180"))
        self.timer = threading.Timer(self.secs, self._expire, args=(rank, what))
        self.timer.daemon = True

    def _expire(self, rank, what):
        import sys
        try:
            flag = os.environ.get("DIAGAN_COMM_TIMEOUT_FLAG")
            if flag:
                open(flag, "w").close()
            print(f"FATAL: rank {rank}: {what} did not finish within {self.secs:.0f} s (another rank never arrived?); "
                  f"exiting with status {NATIVE_INIT_TIMEOUT_EXIT}.  DIAGAN_COMM=torch keeps the exchange on "
                  "torch.distributed's process group.", file=sys.stderr, flush=True)
        finally:
            os._exit(NATIVE_INIT_TIMEOUT_EXIT)

    def __enter__(self):
        if self.secs > 0:
            self.timer.start()
        return self

    def __exit__(self, *exc):
        self.timer.cancel()
        return False


def _maybe_init_native(rank, world, local_rank, backend):
    import sys
    policy = (os.environ.get("DIAGAN_COMM") or "auto").lower()
    if policy == "torch" or not torch.cuda.is_available():
        _NATIVE['why'] = "DIAGAN_COMM=torch" if policy == "torch" else "no GPU"
        return
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
    own_device = torch.cuda.device_count() >= local_world
    if policy == "auto" and (backend != "nccl" or not own_device):
        _NATIVE['why'] = ("ranks share a device" if not own_device else f"backend {backend}")
        return
    err = None
    with _StartupWatchdog(rank, "creating the native RCCL context and its self-test"):
        try:
            init_native_comm(rank, world, local_rank % max(torch.cuda.device_count(), 1))
        except Exception as e:      # noqa: BLE001
            err = repr(e)
        # the self-test's first collective would wait for good on a communicator that not every rank joined
        created = _all_ranks_agree(err is None)
        if created:
            err = _native_selftest(rank, world)
            torch.cuda.synchronize()
        elif err is None:
            err = "context creation failed on another rank"
        ok = _all_ranks_agree(err is None)
    if policy == "rccl":
        if not ok:
            raise RuntimeError(f"DIAGAN_COMM=rccl: {err or 'failed on another rank'}")
        if rank == 0:
            print(f"native RCCL exchange: {world} ranks, self-test passed (DIAGAN_COMM=rccl)", file=sys.stderr)
        return
    if ok and rank == 0:
        print(f"native RCCL exchange: {world} ranks, self-test passed (all-reduce on the compute stream and on the side "
              "stream, all-gather); DIAGAN_COMM=torch selects torch.distributed's process group instead", file=sys.stderr)
    if not ok:
        if _NATIVE['ctx'] is not None:
            try:
                destroy_native_comm()
            except Exception:   # noqa: BLE001
                _NATIVE['ctx'] = None
        _NATIVE['why'] = "native context failed its self-test: " + (err or "on another rank")
        if rank == 0:
            print(f"WARNING: native RCCL exchange disabled ({_NATIVE['why']}); using torch.distributed", file=sys.stderr)


class _NativeWork:
    """handle of a native collective issued on the side stream: wait() orders torch's current stream behind it"""

    def __init__(self, event):
        self.event = event

    def wait(self):
        torch.cuda.current_stream().wait_event(self.event)
        return True


def _native_allreduce(flat, async_op):
    from diagan import _native as nat
    if not async_op:
        nat.call("diagan_allreduce_grads", _NATIVE['ctx'], flat.data_ptr(), flat.numel(), nat.current_stream())
        return None                              # ordered on the current stream: nothing to wait for
    # in flight beside the compute stream: the side stream picks up behind everything queued so far
    # (fork / join by events, so the pattern is also legal inside a stream capture)
    if _NATIVE['side'] is None:
        _NATIVE['side'] = torch.cuda.Stream()
    side = _NATIVE['side']
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        nat.call("diagan_allreduce_grads", _NATIVE['ctx'], flat.data_ptr(), flat.numel(), side.cuda_stream)
        done = torch.cuda.Event()
        done.record(side)
    return _NativeWork(done)


def is_dist():
    return dist.is_available() and dist.is_initialized()


def get_rank():
    return dist.get_rank() if is_dist() else 0


def get_world_size():
    return dist.get_world_size() if is_dist() else 1


def init_from_env(backend=None):
    """env:// rendezvous (RANK / WORLD_SIZE / MASTER_* set by torch.distributed.run), as the
    reference does at stylegan2/train_ffhq.py:503-506.  Returns (rank, local_rank, world_size)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not is_dist():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:      # DIAGAN_DIST_BACKEND=gloo: functional test of the N>1 path on a 1-GPU box
            backend = os.environ.get("DIAGAN_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))
        dist.init_process_group(backend=backend, init_method="env://", rank=rank, world_size=world)
        synchronize()
        _maybe_init_native(rank, world, local_rank, backend)
    return rank, local_rank, world


def synchronize():
    if get_world_size() > 1:
        dist.barrier()


def all_reduce_mean_(flat):
    """In-place mean over ranks of one contiguous slab (no-op for a single process)."""
    world = get_world_size()
    if world == 1:
        return flat
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat.mul_(1.0 / world)
    return flat


def all_reduce_sum_(flat, async_op=False):
    """In-place SUM over ranks of one contiguous slab; the 1/W of the mean is folded into the optimiser's gradient
    read (FusedAdam.grad_scale), which saves a pass over the slab.  async_op: returns the work handle (the exchange
    runs on the backend's own stream; `wait()` orders the current stream behind it)."""
    if get_world_size() == 1:
        return None
    if _NATIVE['ctx'] is not None and flat.is_cuda:
        return _native_allreduce(flat, async_op)
    return dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=async_op)


def seed_device_per_rank(seed, rank=None, enable=True):
    """Give every rank's DEVICE generator its own stream (seed + rank) once the replicas have been made identical.

    The CLIs seed every generator with the same value on every rank (diagan-pkg/diagan/utils/settings.py:8-18 is
    called before the process group exists, stylegan2/train_ffhq.py:497).  The CPU generators MUST stay shared: the
    strided ShardedSampler and StyleGAN2's `mixing_noise` (random.random()) rely on lock-step draws.  The device
    generator feeds only the latent noise (`generate_images`) and dropout masks; left shared, every rank would draw the
    same z and the all-reduced fake-side gradient would be the mean of W copies of ONE fake batch.
    enable=False keeps the reference's behaviour (StyleGAN2 path default, see stylegan2_cli.py)."""
    if not enable or seed is None or get_world_size() == 1 or not torch.cuda.is_available():
        return
    rank = get_rank() if rank is None else rank
    torch.cuda.manual_seed(int(seed) + int(rank))


def reduce_sum(tensor):
    if get_world_size() == 1:
        return tensor
    tensor = tensor.clone()
    dist.all_reduce(tensor, op=dist.ReduceOp.SUM)
    return tensor


def gather_grad(params):
    """mean of every parameter's gradient over the ranks (reference distributed.py:55-64); the engine's own networks
    use ONE all-reduce of their flat gradient slab instead (`all_reduce_mean_`)"""
    world = get_world_size()
    if world == 1:
        return
    for p in params:
        if p.grad is not None:
            dist.all_reduce(p.grad.data, op=dist.ReduceOp.SUM)
            p.grad.data.div_(world)


def all_gather(data):
    """list of one picklable object per rank, rank order (reference distributed.py:67-101 pads pickled byte tensors by
    hand; torch.distributed's object collective does the same exchange)"""
    world = get_world_size()
    if world == 1:
        return [data]
    out = [None] * world
    dist.all_gather_object(out, data)
    return out


def all_gather_cat(tensor):
    """Rank-major concatenation of equally shaped per-rank tensors (the shapes are checked: one MAX / MIN all-reduce of
    the element count -- this runs once per logit snapshot, not per step)."""
    world = get_world_size()
    if world == 1:
        return tensor
    if _NATIVE['ctx'] is not None and tensor.is_cuda and tensor.element_size() in (1, 4, 8):
        from diagan import _native as nat
        send = tensor.contiguous()
        n = torch.tensor([send.numel(), -send.numel()], dtype=torch.int64, device=send.device)
        dist.all_reduce(n, op=dist.ReduceOp.MAX)
        # two communicators (the process group's and the native context's) must never have collectives in flight at the
        # same time: the process-group all-reduce above has to be COMPLETE on the device before the native all-gather below
        # is enqueued.  Reading n on the host would do it implicitly; keep the dependence explicit.
        torch.cuda.synchronize(send.device)
        if int(n[0]) != -int(n[1]):
            raise RuntimeError(f"all_gather_cat: per-rank element counts differ ({-int(n[1])} .. {int(n[0])})")
        recv = torch.empty((world * send.shape[0],) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
        nat.call("diagan_allgather_logits", _NATIVE['ctx'], send.data_ptr(), recv.data_ptr(), send.numel(),
                 send.element_size(), nat.current_stream())
        return recv
    out = [torch.empty_like(tensor) for _ in range(world)]
    dist.all_gather(out, tensor.contiguous())
    return torch.cat(out, dim=0)


def shard_range(n, rank=None, world=None):
    """Contiguous index range [lo, hi) of rank `rank` when n items are split over `world` ranks in
    equal blocks of `per` = ceil(n / world) (the last blocks may be short or empty)."""
    rank = get_rank() if rank is None else rank
    world = get_world_size() if world is None else world
    per = (n + world - 1) // world
    return min(rank * per, n), min((rank + 1) * per, n), per


def gather_row_shards(row, n):
    """Every rank filled row[lo:hi] for its own shard_range; returns the full row on every rank with
    ONE all-gather of `per` elements per rank (bit-exact: values are copied, never summed)."""
    if get_world_size() == 1:
        return row
    lo, hi, per = shard_range(n)
    mine = torch.zeros(per, dtype=row.dtype, device=row.device)
    mine[: hi - lo] = row[lo:hi]
    return all_gather_cat(mine)[:n]


def reduce_loss_dict(loss_dict):
    """Mean of each scalar over ranks, valid on rank 0 (reference: distributed.py:104-126)."""
    world = get_world_size()
    if world < 2:
        return loss_dict
    with torch.no_grad():
        keys = sorted(loss_dict.keys())
        stacked = torch.stack([loss_dict[k].detach().float().reshape(()) for k in keys])
        dist.reduce(stacked, dst=0)
        if dist.get_rank() == 0:
            stacked /= world
        return {k: v for k, v in zip(keys, stacked)}


def reconcile_running_stats_(module):
    """Make the BatchNorm running statistics of `module` the mean over the ranks, in place, on every rank (collective).

    Under data parallelism every rank normalises with the statistics of its OWN micro-batch and folds those into its own
    running_mean / running_var -- buffers are deliberately not synchronised per step, like DistributedDataParallel with
    `broadcast_buffers=False` in stylegan2/train_ffhq.py:577.  Parameters stay in lock-step, the running statistics drift
    apart by sampling noise, and a checkpoint written by rank 0 alone would carry the evaluation-mode statistics of an
    arbitrary eighth of the data.  LogTrainer calls this right before it saves (all ranks take part, rank 0 writes): the
    checkpoint then holds the average over all ranks' batches, and the ranks continue from identical buffers.
    Spectral-norm buffers (sn_u, sn_sigma) need nothing: same weights and same start give the same power iterations.
    Returns the number of buffers reconciled."""
    world = get_world_size()
    if world == 1:
        return 0
    bufs = [b for n, b in module.named_buffers()
            if b.is_floating_point() and (n.endswith('running_mean') or n.endswith('running_var'))]
    if not bufs:
        return 0
    flat = torch.cat([b.detach().reshape(-1).to(torch.float32) for b in bufs])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat.mul_(1.0 / world)
    off = 0
    for b in bufs:
        n = b.numel()
        b.copy_(flat[off: off + n].view(b.shape))
        off += n
    return len(bufs)


def broadcast_module_(module, src=0):
    """Make parameters and buffers identical on every rank (done once at start; afterwards the
    replicas stay in lock-step because they apply identical averaged gradients)."""
    if get_world_size() == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src)
