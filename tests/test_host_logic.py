"""CPU: host-side logic of the drop-in surface -- scheduler (pinned to the reference's golden),
LogTrainer control flow (event trace written from diagan-pkg/diagan/trainer/trainer.py:238-346),
samplers, CLI flag sets."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT


class Opt:
    def __init__(self, lr):
        self.param_groups = [{'lr': lr}]

    def state_dict(self):
        return {}


class Log:
    def __init__(self):
        self.m = {}

    def add_metric(self, name, value, group=None, precision=4):
        self.m[name] = value


def test_scheduler_matches_reference_golden(golden_dir):
    from diagan.trainer.scheduler import DRS_LRScheduler
    g = np.load(os.path.join(golden_dir, "scheduler.npz"))
    for tag, decay in (("linear", "linear"), ("none", None)):
        opts = [Opt(2e-4), Opt(1e-4), Opt(2e-4)]
        sch = DRS_LRScheduler(lr_decay=decay, optimizers=opts, num_steps=50000)
        for s, row in zip(g[f"{tag}_steps"], g[f"{tag}_lrs"]):
            log = sch.step(Log(), int(s))
            got = [o.param_groups[0]['lr'] for o in opts] + [log.m['lr_0'], log.m['lr_1'], log.m['lr_2']]
            assert got == list(row), (tag, s)
    with pytest.raises(NotImplementedError):
        DRS_LRScheduler(lr_decay='cosine', optimizers=[], num_steps=10)


class FakeNet:
    def __init__(self, name, trace):
        self.name, self.trace = name, trace
        self.device = torch.device('cpu')
        self.use_gold = False
        self.saved = []

    def to(self, d):
        return self

    def train_step(self, real_batch, log_data, global_step=None, **kw):
        self.trace.append((global_step, self.name, int(real_batch[0][0, 0])))
        return log_data

    def save_checkpoint(self, directory, global_step, optimizer=None):
        self.saved.append((os.path.basename(directory), global_step))

    def eval(self):
        pass

    def train(self):
        pass


class TinyDS(torch.utils.data.Dataset):
    def __init__(self, n, tag):
        self.n, self.tag = n, tag

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return torch.tensor([float(self.tag * 1000 + i)]), 0, 1.0, i


def make_trainer(tmp_path, drs=False, **kw):
    from diagan.trainer.trainer import LogTrainer
    trace = []
    netD, netG = FakeNet('D', trace), FakeNet('G', trace)
    netD_drs = FakeNet('D_drs', trace) if drs else None
    dl = torch.utils.data.DataLoader(TinyDS(6, 1), batch_size=2, shuffle=False)
    dl_drs = torch.utils.data.DataLoader(TinyDS(4, 2), batch_size=2, shuffle=False) if drs else None
    t = LogTrainer(output_path=tmp_path, netD=netD, netG=netG, optD=Opt(2e-4), optG=Opt(2e-4), dataloader=dl,
                   num_steps=kw.pop('num_steps', 4), netD_drs=netD_drs, optD_drs=Opt(2e-4) if drs else None,
                   dataloader_drs=dl_drs, log_dir=str(tmp_path), n_dis=kw.pop('n_dis', 2), lr_decay='linear',
                   device='cpu', print_steps=1000, vis_steps=1000, log_steps=1000, **kw)
    return t, trace, (netD, netG, netD_drs)


def test_trainer_loop_order_phase1(tmp_path):
    t, trace, (netD, netG, _) = make_trainer(tmp_path, save_logits=False, save_steps=3)
    t.train()
    # per global step: n_dis D updates on successive batches, G once on the LAST D batch (trainer.py:250-291)
    assert [(s, n) for s, n, _ in trace] == [(s, n) for s in range(4) for n in ('D', 'D', 'G')]
    for s in range(4):
        d2, g = trace[3 * s + 1], trace[3 * s + 2]
        assert d2[2] == g[2]
    # loader of 3 batches re-iterated when exhausted (mimicry _fetch_data)
    assert [b for _, n, b in trace if n == 'D'] == [1000, 1002, 1004, 1000, 1002, 1004, 1000, 1002]
    # checkpoints every save_steps and at the end; scheduler ran after the increment (lr at step 4 of 4 = 0)
    assert netG.saved == [('netG', 3), ('netG', 4)] and netD.saved == [('netD', 3), ('netD', 4)]
    assert t.optD.param_groups[0]['lr'] == 0.0


def test_trainer_loop_order_phase2_drs(tmp_path):
    t, trace, (netD, netG, netD_drs) = make_trainer(tmp_path, drs=True, save_logits=False, save_steps=100)
    t.train()
    assert [(s, n) for s, n, _ in trace] == [(s, n) for s in range(4) for n in ('D', 'D_drs', 'D', 'D_drs', 'G')]
    # D_drs draws from ITS OWN loader (tag 2) even after exhaustion: the intended behaviour
    assert all(b // 1000 == 2 for _, n, b in trace if n == 'D_drs')
    assert [b for _, n, b in trace if n == 'D_drs'][:4] == [2000, 2002, 2000, 2002]
    assert netD_drs.saved == [('netD_drs', 4)]


def test_trainer_fetch_quirk_switch(tmp_path):
    """compat_fetch_quirk=True reproduces mimicry's _fetch_data: an exhausted D_drs iterator restarts
    on self.dataloader (the weighted loader)."""
    t, trace, _ = make_trainer(tmp_path, drs=True, save_logits=False, save_steps=100, compat_fetch_quirk=True)
    t.train()
    drs_batches = [b for _, n, b in trace if n == 'D_drs']
    assert drs_batches[:2] == [2000, 2002] and drs_batches[2] // 1000 == 1


def test_logit_snapshot_schedule(tmp_path, monkeypatch):
    t, trace, _ = make_trainer(tmp_path, num_steps=12, n_dis=1, save_logits=True, logit_save_steps=2,
                               save_logit_after=4, stop_save_logit_after=8, save_steps=100)
    calls = []
    monkeypatch.setattr(t, "_get_logit", lambda netD, eval_mode, record, step: calls.append((netD.name, eval_mode, step)))
    monkeypatch.setattr(t, "_save_logit", lambda logits_dict=None: None)
    t.train()
    assert calls == [('D', True, s) for s in (4, 6, 8)]           # bounds inclusive (trainer.py:328)
    assert list(t.logit_records.keys()) == ['netD_eval']


def test_gold_switch(tmp_path):
    t, trace, (netD, _, _) = make_trainer(tmp_path, save_logits=False, gold=True, gold_step=2, save_steps=100)
    seen = []
    orig = netD.train_step
    netD.train_step = lambda **kw: (seen.append((kw['global_step'], netD.use_gold)), orig(**kw))[1]
    t.train()
    assert [g for s, g in seen if s < 2] == [False] * 4 and all(g for s, g in seen if s >= 2)


def test_weighted_sampler_floor_and_sharding():
    from diagan.datasets.sampler import ShardedSampler, floor_weights, make_weighted_sampler
    w = np.array([0.0, 1e-9, 0.5, 2.0])
    assert floor_weights(w) == [1e-6, 1e-6, 0.5, 2.0]
    torch.manual_seed(0)
    full = list(iter(make_weighted_sampler(np.ones(10))))
    shards = []
    for r in range(3):
        torch.manual_seed(0)
        shards.append(list(iter(ShardedSampler(make_weighted_sampler(np.ones(10)), r, 3))))
    assert all(len(s) == 3 for s in shards)
    assert [x for trio in zip(*shards) for x in trio] == full[:9]


def test_weighted_dataset_item_contract():
    from diagan.datasets.predefined import WeightedDataset, get_predefined_dataset
    ds = get_predefined_dataset('cifar10', num_data=5)
    x, y, w, i = ds[3]
    assert x.shape == (3, 32, 32) and x.dtype == torch.float32 and -1 <= x.min() and x.max() <= 1
    assert (y, w, i) == (0, 1.0, 3) and len(ds) == 5
    ds2 = WeightedDataset(ds.dataset, weights=np.arange(5.0))
    assert ds2[4][2] == 4.0


REF_P1_FLAGS = ["--dataset", "--root", "--work_dir", "--exp_name", "--model", "--loss_type", "--gpu", "--num_pack",
                "--batch_size", "--seed", "--download_dataset", "--topk", "--num_steps", "--logit_save_steps",
                "--decay", "--n_dis", "--imb_factor", "--celeba_class_attr", "--ckpt_step", "--no_save_logits",
                "--save_logit_after", "--stop_save_logit_after"]
REF_P2_FLAGS = ["--dataset", "--root", "--work_dir", "--exp_name", "--baseline_exp_name", "--p1_step", "--model",
                "--loss_type", "--gpu", "--num_steps", "--batch_size", "--seed", "--decay", "--n_dis",
                "--resample_score", "--gold", "--topk"]


def test_cli_flag_surface():
    """Flag names and defaults of train_mimicry_phase1.py:29-51 / train_mimicry_phase2.py:39-56."""
    sys.path.insert(0, ROOT)
    import train_mimicry_phase1 as p1
    import train_mimicry_phase2 as p2
    o1 = {a.option_strings[0] if a.option_strings[0].startswith('--') else a.option_strings[-1]: a
          for a in p1.build_parser()._actions if a.option_strings}
    o2 = {a.option_strings[0] if a.option_strings[0].startswith('--') else a.option_strings[-1]: a
          for a in p2.build_parser()._actions if a.option_strings}
    assert all(f in o1 for f in REF_P1_FLAGS) and all(f in o2 for f in REF_P2_FLAGS)
    d1 = vars(p1.build_parser().parse_args([]))
    assert (d1['batch_size'], d1['n_dis'], d1['loss_type'], d1['seed'], d1['num_steps'], d1['decay']) == \
        (64, 5, 'hinge', 1, 100000, 'linear')
    d2 = vars(p2.build_parser().parse_args([]))
    assert (d2['p1_step'], d2['num_steps'], d2['batch_size'], d2['n_dis']) == (40000, 80000, 64, 5)


def test_conf_keys_match_reference_naming():
    from diagan.utils.plot import conf_key, conf_t_values
    keys = [conf_key(t) for t in conf_t_values()]
    assert len(keys) == 99 and keys[0] == 'ldr_conf_0.1_ratio_50' and keys[2] == 'ldr_conf_0.3_ratio_50' \
        and keys[49] == 'ldr_conf_5.0_ratio_50' and keys[-1] == 'ldr_conf_9.9_ratio_50'


def test_topk_rate_decay():
    from diagan.models.topk_models import TopKGenerator
    t = TopKGenerator(use_topk=True)
    t.decay_topk_rate(0, epoch_steps=782)
    assert t.topk_rate == 1
    t.decay_topk_rate(782 * 10 + 5, epoch_steps=782)
    assert t.topk_rate == 0.99 ** 10
    t.decay_topk_rate(782 * 1000, epoch_steps=782)
    assert t.topk_rate == 0.5


def test_drs_acceptance_matches_reference(golden_dir):
    """DRS eval-time rejection sampling: same logits + same NumPy RNG state -> same accepted samples as the
    reference class (golden from diagan-pkg/diagan/models/drs.py run on scripted nets)."""
    from diagan.models.drs import DRS
    g = np.load(os.path.join(golden_dir, "drs.npz"))

    class FakeG:
        def generate_images(self, n, device=None):
            return torch.arange(n, dtype=torch.float32).view(n, 1)

    class FakeD:
        def __init__(self):
            self.g = torch.Generator().manual_seed(23)

        def __call__(self, imgs):
            return torch.randn(imgs.shape[0], 1, generator=self.g) * 1.5

    np.random.seed(7)
    drs = DRS(FakeG(), FakeD(), device='cpu')
    assert float(drs.maximum) == float(g["maximum_after_init"])
    for i in range(3):
        kept = drs.sub_rejection_sampler(torch.arange(256, dtype=torch.float32).view(256, 1), g[f"ldr{i}"])
        assert np.array_equal(kept.numpy().reshape(-1), g[f"kept{i}"])
    assert float(drs.maximum) == float(g["maximum_final"])


def test_batchnorm_counter_is_folded_into_state_dict():
    """num_batches_tracked is counted on the host and materialised by state_dict() (nn.BatchNorm2d semantics)."""
    from diagan.models.layers import BatchNorm
    bn = BatchNorm(8)
    bn._pending_batches += 3            # what three training-mode forwards do
    sd = bn.state_dict()
    assert int(sd['num_batches_tracked']) == 3 and bn._pending_batches == 0
    bn._pending_batches = 5
    bn.load_state_dict(sd)
    assert int(bn.state_dict()['num_batches_tracked']) == 3


def test_deepcopy_of_a_flat_net_drops_learned_launch_state():
    """the Winograd call sites of a FlatNet close over the ORIGINAL's layers; a deep copy must start without them,
    with parameters that are views of its OWN slab and a new slab generation (baked descriptor tables are rebuilt)."""
    import copy
    from diagan.models.predefined_models import get_gan_model
    netG, _ = get_gan_model(dataset_name='cifar10', gan_type='sngan', loss_type='ns')[:2]
    conv = next(m for m in netG.modules() if m.__class__.__name__ == 'ConvLayer')
    netG.flat_params                                     # slabs built (as after .to(device) / a first optimizer step)
    netG._wino_batches[('f', None)] = object()           # what a first pass leaves behind
    conv.__dict__['_wsites'] = {('f', None, 0): object()}
    gen = netG.slab_generation
    twin = copy.deepcopy(netG)
    assert twin._wino_batches == {} and netG._wino_batches != {}
    tconv = next(m for m in twin.modules() if m.__class__.__name__ == 'ConvLayer')
    assert '_wsites' not in tconv.__dict__ and '_wsites' in conv.__dict__
    assert tconv._net is twin and twin.slab_generation == gen + 1
    flat = twin.flat_params
    for p in twin.parameters():
        assert p.data.untyped_storage().data_ptr() == flat.untyped_storage().data_ptr()
    assert flat.untyped_storage().data_ptr() != netG.flat_params.untyped_storage().data_ptr()
    twin.flat_params.add_(1.0)                           # the copy diverges; the original does not move
    assert not torch.equal(next(twin.parameters()), next(netG.parameters()))


def test_color_mnist_front_ends_keep_the_reference_flags():
    """train_mimicry_color_mnist_phase{1,2}.py: flag names and defaults (reference :48-67 / :41-61)"""
    from diagan import cli
    p1, p2 = cli.color_mnist_phase1_parser().parse_args([]), cli.color_mnist_phase2_parser().parse_args([])
    for a in (p1, p2):
        assert (a.dataset, a.root, a.work_dir, a.exp_name, a.model) == \
            ("color_mnist", "./dataset/colour_mnist", "./exp_results", "colour_mnist", "mnistgan")
        assert (a.gpu, a.num_pack, a.batch_size, a.seed, a.num_steps, a.logit_save_steps) == ('0', 1, 64, 1, 20000, 100)
        assert (a.decay, a.n_dis, a.major_ratio, a.num_data, a.resample_score) == ('None', 1, 0.99, 10000, None)
    assert (p1.loss_type, p1.use_clipping, p1.topk) == ("ns", False, 0)
    assert (p2.loss_type, p2.baseline_exp_name, p2.p1_step, p2.use_eval_logits) == ("hinge", "colour_mnist", 10000, None)
    # the weight conditioning of the phase-1 script (:22-32): floor at 0.1, or clip to mean -/+ 2 var
    w = np.array([0.0, 0.05, 0.2, 0.5, 3.0])
    np.testing.assert_array_equal(cli.floor_or_clip_weights(w), [0.1, 0.1, 0.2, 0.5, 3.0])
    mean, var = w.mean(), w.var()
    expect = np.clip(w, max(mean - 2 * var, 0.1), mean + 2 * var)
    np.testing.assert_allclose(cli.floor_or_clip_weights(w, clip=True), expect)


def test_bench_refuses_a_world_size_that_differs_from_gpus():
    """bench.py --gpus N inside a job of another size: non-zero exit and NO JSON line (a line that silently reports a
    different job size than the one asked for is worse than none); decided before any rendezvous or GPU access."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="1")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "1"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert "WORLD_SIZE=2" in out.stderr
    env = dict(env, WORLD_SIZE="2")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "1"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
