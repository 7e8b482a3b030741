"""LogTrainer: the phase-1 / phase-2 training loop with the per-index logit record.

Host-side mirror of diagan-pkg/diagan/trainer/trainer.py (class LogTrainer, :15-361), which itself
extends torch_mimicry.training.Trainer.  Constructor keywords, loop order, snapshot condition, file
names and checkpoint cadence are the reference's; what differs is what runs underneath:

  * netD/netG.train_step launch HIP kernels (no autograd, no per-step .item() syncs);
  * the logit record lives in HBM (utils.plot.LogitRecord): D(x) rows are scattered on device
    (diagan_logit_scatter) instead of `.cpu().numpy()` per batch (trainer.py:154);
  * under torch.distributed (one process per GPU) gradients are averaged by one all-reduce per
    network inside train_step, and the logit pass is sharded by contiguous index range with one
    all-gather per snapshot (pattern: stylegan2/train_ffhq.py:128-143).
"""
import os
import pickle
import time

import torch

from diagan.trainer import distributed as dist
from diagan.trainer.logger import Logger, MetricLog
from diagan.trainer.scheduler import DRS_LRScheduler
from diagan.utils.plot import LogitRecord
from diagan.utils.settings import quiesce_gc


class _Clock:
    """Wall time per printed step (the reference prints the mean over `print_steps`)."""

    def __init__(self):
        self.t = time.time()

    def lap(self, steps):
        now = time.time()
        dt, self.t = (now - self.t) / steps, now
        return dt


class LogTrainer:
    """Keyword-compatible with the reference's LogTrainer (trainer.py:16-49); extra: `compat_fetch_quirk`."""

    _AT_LEAST_ONE = ('num_steps', 'n_dis', 'print_steps', 'vis_steps', 'log_steps', 'save_steps', 'flush_secs')

    def __init__(self, output_path, netD, netG, optD, optG, dataloader, num_steps, netD_drs=None, optD_drs=None,
                 dataloader_drs=None, netD_drs_ckpt_file=None, log_dir='./log', n_dis=1, lr_decay=None, device=None,
                 netG_ckpt_file=None, netD_ckpt_file=None, print_steps=1, vis_steps=500, log_steps=50,
                 save_steps=5000, flush_secs=30, logit_save_steps=500, amp=False, save_logits=True, topk=False,
                 gold=False, gold_step=None, save_logit_after=0, stop_save_logit_after=100000,
                 save_eval_logits=True, compat_fetch_quirk=False, logit_eval_batch=1024):
        given = dict(locals())
        given.pop('self')
        for name in self._AT_LEAST_ONE:
            if given[name] < 1:
                raise ValueError('{} must be at least 1 but got {}.'.format(name, given[name]))
        if amp:
            raise NotImplementedError("amp is not part of the fp32 MI355X path")
        if gold and gold_step is None:
            raise AssertionError("gold re-weighting needs gold_step")
        self.train_drs = netD_drs is not None
        if self.train_drs and (dataloader_drs is None or optD_drs is None):
            raise AssertionError("netD_drs needs dataloader_drs and optD_drs")
        given.pop('flush_secs')
        for name, value in given.items():            # every constructor keyword is an attribute of the same name
            setattr(self, name, value)

        os.makedirs(self.log_dir, exist_ok=True)
        self.device = self._resolve_device(device)
        self.logger = Logger(log_dir=self.log_dir, num_steps=self.num_steps, dataset_size=len(self.dataloader),
                             flush_secs=flush_secs, device=self.device)
        # the scheduler caches the base learning rates NOW, i.e. before any checkpoint restore (scheduler.py:38)
        live = [opt for opt in (self.optD, self.optD_drs, self.optG) if opt is not None]
        self.scheduler = DRS_LRScheduler(lr_decay=self.lr_decay, optimizers=live, num_steps=self.num_steps)
        ckpt_root = os.path.join(self.log_dir, 'checkpoints')
        self.netG_ckpt_dir, self.netD_ckpt_dir = os.path.join(ckpt_root, 'netG'), os.path.join(ckpt_root, 'netD')
        self.netD_drs_ckpt_dir = os.path.join(ckpt_root, 'netD_drs') if self.train_drs else None
        for net in (self.netD, self.netG, self.netD_drs):
            if net is not None and net.device != self.device:
                net.to(self.device)
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.logit_records = {}
        self.events = []          # (global_step, event) trace used by the control-flow tests

    @staticmethod
    def _resolve_device(device):
        dev = torch.device(device if device else ('cuda:0' if torch.cuda.is_available() else 'cpu'))
        if dev.type == 'cuda' and dev.index is None:
            dev = torch.device('cuda', torch.cuda.current_device())
        return dev

    # ---- logit record -------------------------------------------------------------------------
    @property
    def logit_results(self):
        """dict{name -> dict{step -> float64 ndarray[N]}}: what the reference pickles (trainer.py:138-140)."""
        return {name: rec.to_dict() for name, rec in self.logit_records.items()}

    def _save_logit(self, logits_dict=None):
        if self.rank != 0:
            return
        logits_dict = self.logit_results if logits_dict is None else logits_dict
        for name, logits in logits_dict.items():
            with open(os.path.join(str(self.output_path), f'logits_{name}.pkl'), 'wb') as f:
                pickle.dump(logits, f)

    def _eval_loader(self):
        """Loader for the logit pass.  Single process: the training loader itself (trainer.py:143).
        Data parallel: this rank's contiguous index range [lo, hi) of the dataset."""
        if self.world == 1:
            return self.dataloader, None
        ds = self.dataloader.dataset
        lo, hi, per = dist.shard_range(len(ds), self.rank, self.world)
        sub = torch.utils.data.Subset(ds, range(lo, hi))
        loader = torch.utils.data.DataLoader(sub, batch_size=self.dataloader.batch_size, shuffle=False,
                                             num_workers=getattr(self.dataloader, 'num_workers', 0))
        return loader, (lo, hi, per)

    def _get_logit(self, netD, eval_mode=False, record=None, step=None):
        """Full pass of D over the dataset; rec[row, idx] = D(x).  Returns the float64 row (device)."""
        n_data = len(self.dataloader.dataset)
        if record is None:
            record = LogitRecord(n_data, capacity=1, device=self.device)
        row = record.new_snapshot(step if step is not None else -1)
        loader, shard = self._eval_loader()
        if eval_mode:
            netD.eval()
        # The pass walks the TRAINING loader (trainer.py:143), whose batch of 64 leaves the D forward launch-bound
        # (84 000 images/s); the loader batches are therefore gathered into evaluation batches of >= `logit_eval_batch`
        # images before the forward.  Placement is by index (`scatter`), so the record is unchanged.  Eval mode only: in
        # train mode every forward advances spectral norm's power iteration (and DCGAN's BatchNorm / dropout depend on
        # the batch), so the train-mode record of the colour-MNIST scripts keeps the loader's batches.
        group = self.logit_eval_batch if eval_mode else 0
        pend_x, pend_i, pending = [], [], 0

        def flush():
            nonlocal pend_x, pend_i, pending
            if not pend_x:
                return
            x = pend_x[0] if len(pend_x) == 1 else torch.cat(pend_x)
            logit = netD(x)
            if type(logit) is tuple:
                logit = logit[0]
            record.scatter(row, pend_i[0] if len(pend_i) == 1 else torch.cat(pend_i), logit.view(-1))
            pend_x, pend_i, pending = [], [], 0
        # Tensor-backed datasets (`fetch_range`, datasets/predefined.py) hand out whole evaluation batches as slices: the
        # walk over 782 loader batches of 64 items -- 50 000 `__getitem__` calls and collates -- was what bounded the pass
        # (145 000 images/s) once D itself was batched.
        ds = self.dataloader.dataset
        lo, hi = (shard[0], shard[1]) if shard is not None else (0, n_data)
        # ... only where the loader visits every index exactly once (uniform shuffling or sequential): the reference walks
        # `self.dataloader`, and with phase 2's WeightedRandomSampler (replacement=True) the indices it never draws keep the
        # row's initial zeros (trainer.py:142-156: np.zeros + assignment by index) -- that record is reproduced by the walk
        # (a RandomSampler with replacement / a num_samples of its own does NOT visit every index once: that walk is reproduced
        #  like the weighted one.  Assumption of the sliced path, stated: fetching items consumes no global RNG -- tensor-backed
        #  datasets have no random transforms -- so `iter(loader)` + one `next` leaves the generator where the full walk would.)
        smp = getattr(loader, 'sampler', None)
        once = (isinstance(smp, torch.utils.data.SequentialSampler)
                or (isinstance(smp, torch.utils.data.RandomSampler) and not getattr(smp, 'replacement', False)
                    and getattr(smp, '_num_samples', None) is None))
        complete = shard is not None or once
        ranged = (group > 0 and complete and hasattr(ds, 'fetch_range') and hi > lo
                  and ds.fetch_range(lo, lo + 1) is not None)
        with torch.no_grad():
            if ranged:
                if shard is None:
                    # The reference WALKS the training loader here (trainer.py:143).  Starting that walk is what touches the
                    # CPU generator -- the loader draws its base seed, a shuffling / weighted sampler its seed or its whole
                    # multinomial sample -- and everything after the snapshot (the next epoch's order, phase-2 sampling)
                    # sees the stream behind those draws.  Start the walk and drop it: same draws, one batch fetched.
                    it = iter(loader)
                    next(it, None)
                    del it
                for a0 in range(lo, hi, group):
                    data, idx = ds.fetch_range(a0, min(a0 + group, hi))
                    pend_x.append(data.to(self.device, non_blocking=True))
                    pend_i.append(idx)
                    flush()
            else:
                for data, targets, _, idx in loader:
                    pend_x.append(data.to(self.device, non_blocking=True))
                    pend_i.append(idx)
                    pending += data.shape[0]
                    if pending >= group:
                        flush()
                flush()
        netD.train()
        record.check_bounds()
        if shard is not None:                     # one all-gather of the contiguous shards
            record.buf[row].copy_(dist.gather_row_shards(record.buf[row], n_data))
        return record.buf[row]

    # ---- checkpoints --------------------------------------------------------------------------
    def _restore_models_and_step(self):
        global_step_D = global_step_G = 0
        if self.netD_ckpt_file:
            assert os.path.exists(self.netD_ckpt_file)
            print("INFO: Restoring checkpoint for D...")
            global_step_D = self.netD.restore_checkpoint(ckpt_file=self.netD_ckpt_file, optimizer=self.optD)
        if self.netG_ckpt_file:
            assert os.path.exists(self.netG_ckpt_file)
            print("INFO: Restoring checkpoint for G...")
            global_step_G = self.netG.restore_checkpoint(ckpt_file=self.netG_ckpt_file, optimizer=self.optG)
        if self.train_drs and self.netD_drs_ckpt_file:
            assert os.path.exists(self.netD_drs_ckpt_file)
            print("INFO: Restoring checkpoint for D_drs...")
            global_step_D = self.netD_drs.restore_checkpoint(ckpt_file=self.netD_drs_ckpt_file,
                                                             optimizer=self.optD_drs)
        if global_step_D != global_step_G:
            print(f'WARN: global_step_D {global_step_D} != global_step_G {global_step_G}, use global_step_G')
        return global_step_G

    def _save_model_checkpoints(self, global_step, collective=True):
        # data parallel: every rank's BatchNorm running statistics come from its own batches -- average them over the
        # ranks (collective: before the rank-0 gate) so the file does not carry one arbitrary rank's evaluation statistics.
        # collective=True is a promise that EVERY rank makes this call at the same step (the periodic and the final save);
        # a rank-asymmetric caller (the KeyboardInterrupt handler: the ranks are interrupted at different points, one may
        # already sit in a gradient all-reduce the others never enter) passes False and rank 0 writes its own statistics.
        if collective:
            for net in (self.netG, self.netD, self.netD_drs if self.train_drs else None):
                if net is not None:
                    dist.reconcile_running_stats_(net)
        if self.rank != 0:
            return
        self.netG.save_checkpoint(directory=self.netG_ckpt_dir, global_step=global_step, optimizer=self.optG)
        if self.netD is not None:
            self.netD.save_checkpoint(directory=self.netD_ckpt_dir, global_step=global_step, optimizer=self.optD)
        if self.train_drs:
            self.netD_drs.save_checkpoint(directory=self.netD_drs_ckpt_dir, global_step=global_step,
                                          optimizer=self.optD_drs)

    # ---- data ---------------------------------------------------------------------------------
    def _fetch_data(self, iter_dataloader, dataloader=None):
        """next(batch) with re-iteration at the end of an epoch, moved to the device.

        mimicry's Trainer._fetch_data always re-iterates `self.dataloader`; LogTrainer does not
        override it, so in the reference an exhausted D_drs iterator silently restarts on the
        *weighted* loader.  Default here is the intended behaviour (each iterator restarts on its
        own loader); compat_fetch_quirk=True reproduces the reference's."""
        if dataloader is None or self.compat_fetch_quirk:
            dataloader = self.dataloader
        try:
            real_batch = next(iter_dataloader)
        except StopIteration:
            iter_dataloader = iter(dataloader)
            real_batch = next(iter_dataloader)
        real_batch = (real_batch[0].to(self.device, non_blocking=True), real_batch[1].to(self.device, non_blocking=True))
        return iter_dataloader, real_batch

    # ---- the loop -----------------------------------------------------------------------------
    # Order of one global step (trainer.py:238-299): [top-k decay] [GOLD switch] n_dis x (D update [, D_drs update]),
    # the G update on the LAST D batch, step += 1, LR schedule, then the periodic duties in the order
    # summaries, console line, sample grid, logit snapshot, checkpoint (+ pickle of the record).
    def _updates(self, step, streams, log):
        batches, batches_drs = self._fetch_step(streams)
        full = all(b[0].shape[0] == self.dataloader.batch_size for b in batches + batches_drs)
        if full and self._graph_wanted():
            return self._graphed_updates(step, batches, batches_drs)
        if getattr(self, '_graph', None) is not None or getattr(self, '_graph_seen', 0):
            # an ordinary step with another batch size re-allocates the per-layer weight-gradient slabs and descriptor
            # tables the captured launches point to: drop the graph; one ordinary full step rebuilds them, then capture again
            self._graph, self._graph_seen = None, self._GRAPH_EAGER_STEPS - 1
        return self._device_updates(step, batches, batches_drs, log, full)

    # ---- launch-bound networks: the device work of a global step replayed as ONE hipGraph ------------------------------
    # MNIST-DCGAN's step is ~1000 launches of ~10 us: bound by the per-launch floor, not by the GPU's arithmetic
    # (bench.py --workload dcgan: 8.75 -> 8.47 ms per step with --graph, profiles/r04_raw/bench_dcgan{,_graph}.json).  Networks that declare `launch_bound = True` get the
    # step captured once (diagan/utils/graph.py) after a few ordinary steps and replayed from then on: same kernels, same
    # order, bit-identical parameters (tests/test_graph_gpu.py).  Not under data parallelism (collectives), top-k or GOLD
    # (host values that change from step to step are baked into the captured launches); a step with a ragged last batch
    # runs eagerly.  DIAGAN_GRAPH=0 / 1 forces it off / on.
    _GRAPH_EAGER_STEPS = 3

    def _graph_wanted(self):
        env = os.environ.get("DIAGAN_GRAPH")
        if env == "0" or self.world > 1 or self.topk or self.gold or self.device.type != 'cuda':
            return False
        nets = [n for n in (self.netG, self.netD, self.netD_drs) if n is not None]
        return env == "1" or all(getattr(n, 'launch_bound', False) for n in nets)

    def _graphed_updates(self, step, batches, batches_drs):
        from diagan.utils.graph import GraphedStep
        g = getattr(self, '_graph', None)
        if g is None:
            self._graph_seen = getattr(self, '_graph_seen', 0) + 1
            if self._graph_seen <= self._GRAPH_EAGER_STEPS:      # ordinary steps first: every lazily built table exists
                return self._device_updates(step, batches, batches_drs, MetricLog(), True)
            # static input tensors the captured launches read; the capture itself runs the host code of ONE step without
            # device work, the replay right below is this step's device work
            self._graph_static = [tuple(t.clone() for t in b) for b in batches]
            self._graph_static_drs = [tuple(t.clone() for t in b) for b in batches_drs]
            cap_log, n0 = MetricLog(), len(self.events)
            nets = (self.netG, self.netD, self.netD_drs)
            g = GraphedStep(lambda: self._device_updates(step, self._graph_static, self._graph_static_drs, cap_log, True),
                            nets, (self.optG, self.optD, self.optD_drs), warmup=0).capture()
            self._graph_kinds = [k for _, k in self.events[n0:]]
            del self.events[n0:]
            self._graph_metrics = [(name, m._value, m.group, m.precision) for name, m in cap_log.items()]
            self._graph = g
        for dst, src in zip(self._graph_static + self._graph_static_drs, batches + batches_drs):
            for d, t in zip(dst, src):
                d.copy_(t, non_blocking=True)
        g()
        self.events.extend((step, k) for k in self._graph_kinds)
        log = MetricLog()
        for name, value, group, precision in self._graph_metrics:       # device scalars the replay has just rewritten
            log.add_metric(name, value, group=group, precision=precision)
        return log

    def _fetch_step(self, streams):
        # The real batches of the step are fetched first, in the order the updates consume them (main, drs, main, ...: the
        # iterators advance -- and restart at an epoch's end -- in the reference's order).  Knowing their sizes keeps the
        # stacked generator forward honest: it draws the noise of all the D (and D_drs) updates up front, which equals the
        # reference's draw order only when every update sees a full batch; a step that contains an epoch's ragged last
        # batch therefore draws its noise update by update, like the reference.
        batches, batches_drs = [], []
        for i in range(self.n_dis):
            streams['main'], b = self._fetch_data(iter_dataloader=streams['main'])
            batches.append(b)
            if self.train_drs:
                streams['drs'], b = self._fetch_data(iter_dataloader=streams['drs'], dataloader=self.dataloader_drs)
                batches_drs.append(b)
        return batches, batches_drs

    def _device_updates(self, step, batches, batches_drs, log, full):
        prefetch = getattr(self.netG, 'prefetch_fakes', None)       # optional part of the generator protocol
        if prefetch is not None:
            prefetch(self.n_dis * (2 if self.train_drs else 1) if full else 0, self.dataloader.batch_size,
                     device=self.device, g_step=True)
        # data parallel phase 2: D and D_drs are independent (reference trainer.py:250-277), so D's gradient all-reduce
        # stays in flight under D_drs's forward / backward and D's Adam step follows it
        # (only with a discriminator that declares the protocol: one that swallowed the extra keyword and stepped anyway
        #  would get two Adam steps per update)
        overlap = self.train_drs and self.world > 1 and getattr(self.netD, 'supports_defer_step', False)
        for i in range(self.n_dis):
            batch = batches[i]
            extra = dict(defer_step=True) if overlap else {}
            log = self.netD.train_step(real_batch=batch, netG=self.netG, optD=self.optD, log_data=log,
                                       global_step=step, device=self.device, scaler=None, **extra)
            self.events.append((step, 'D'))
            if self.train_drs:
                log = self.netD_drs.train_step(real_batch=batches_drs[i], netG=self.netG, optD=self.optD_drs,
                                               log_data=log, global_step=step, device=self.device, scaler=None)
                self.events.append((step, 'D_drs'))
            if overlap:
                self.optD.step()
        log = self.netG.train_step(real_batch=batch, netD=self.netD, optG=self.optG, global_step=step, log_data=log,
                                   device=self.device, scaler=None)
        self.events.append((step, 'G'))
        return log

    def _report(self, step, log, clock):
        if step % self.log_steps == 0 and self.rank == 0:
            self.logger.write_summaries(log_data=log, global_step=step)
        if step % self.print_steps == 0:
            log.add_metric('topk_rate', getattr(self.netG, 'topk_rate', 1), group='topk_rate', precision=6)
            per_step = clock.lap(self.print_steps)
            if self.rank == 0:
                self.logger.print_log(global_step=step, log_data=log, time_taken=per_step)
        if step % self.vis_steps == 0 and self.rank == 0:
            self.logger.vis_images(netG=self.netG, global_step=step)

    def _snapshot_due(self, step):
        """Bounds are inclusive on both sides (trainer.py:328)."""
        return (self.save_logits and step % self.logit_save_steps == 0
                and self.save_logit_after <= step <= self.stop_save_logit_after)

    def _snapshot(self, step):
        net, name = (self.netD_drs, 'netD_drs') if self.train_drs else (self.netD, 'netD')
        mode = 'eval' if self.save_eval_logits else 'train'
        print(f"INFO: logit saving {mode} netD: {name}...")
        key = f'{name}_{mode}'
        if key not in self.logit_records:
            self.logit_records[key] = LogitRecord(len(self.dataloader.dataset), capacity=64, device=self.device)
        self._get_logit(netD=net, eval_mode=(mode == 'eval'), record=self.logit_records[key], step=step)
        self.events.append((step, 'logit'))

    def _persist(self, step, banner, collective=True):
        print(banner)
        self._save_model_checkpoints(step, collective=collective)
        if self.save_logits and step >= self.save_logit_after:
            self._save_logit()

    def train(self):
        step = self._restore_models_and_step()
        if self.gold and step >= self.gold_step:
            self.netD.use_gold = True
        print("INFO: Starting training from global step {}...".format(step))
        streams = {'main': iter(self.dataloader)}
        if self.train_drs:
            streams['drs'] = iter(self.dataloader_drs)
        clock = _Clock()
        quiesced = False
        try:
            while step < self.num_steps:
                if self.topk:
                    self.netG.decay_topk_rate(step, epoch_steps=len(self.dataloader))
                if self.gold and step == self.gold_step:
                    self.netD.use_gold = True
                log = self._updates(step, streams, MetricLog())
                if not quiesced:                           # (after the first step: every launch table and workspace exists now)
                    quiesce_gc()
                    quiesced = True
                step += 1                                  # the schedule and all periodic duties see the NEW step
                log = self.scheduler.step(log_data=log, global_step=step)
                self._report(step, log, clock)
                if self._snapshot_due(step):
                    self._snapshot(step)
                if step % self.save_steps == 0:
                    self._persist(step, "INFO: Saving checkpoints...")
                    self.events.append((step, 'ckpt'))
            self._persist(step, "INFO: Saving final checkpoints...")
        except KeyboardInterrupt:
            self._persist(step, "INFO: Saving checkpoints from keyboard interrupt...", collective=False)
        finally:
            self.logger.close_writers()
        print("INFO: Training Ended.")
