"""GPU: a global step replayed as ONE hipGraph (diagan/utils/graph.py) is the same computation as the eager step --
same parameters bit for bit after several steps, including Adam's step-dependent bias corrections, the scheduled
learning rate, BatchNorm's batch counter and fresh random numbers per replay."""
import pytest
import torch

import bench

pytestmark = pytest.mark.gpu


def _run(dataset, graph, steps=4, batch=16, n_dis=2):
    dev = torch.device("cuda", 0)
    netG, netD, netD_drs, optG, optD, optD_drs = bench.build_models(dataset, 'ns', 1, dev)
    gen = torch.Generator().manual_seed(3)
    res = 32
    batches = [(torch.rand(batch, 3, res, res, generator=gen) * 2 - 1).to(dev) for _ in range(n_dis)]
    step = bench.make_global_step(netG, netD, netD_drs, optG, optD, optD_drs, batches, n_dis, num_steps=40, device=dev)
    torch.manual_seed(11)
    if graph:
        from diagan.utils.graph import GraphedStep
        g = GraphedStep(step.device_part, (netG, netD), (optG, optD), warmup=2, after=step.host_part).capture()
        for _ in range(steps - 2):
            g()
    else:
        for _ in range(steps):
            step()
    torch.cuda.synchronize()
    return netG, netD, optG, optD


@pytest.mark.parametrize("dataset", ["color_mnist", "cifar10"])
def test_graph_replay_equals_eager(dataset):
    eg, ed, eog, eod = _run(dataset, graph=False)
    gg, gd, gog, god = _run(dataset, graph=True)
    assert torch.equal(eg.flat_params, gg.flat_params) and torch.equal(ed.flat_params, gd.flat_params)
    assert eog._step == gog._step == 4 and eod._step == god._step == 8
    assert eog.param_groups[0]['lr'] == gog.param_groups[0]['lr'] < 2e-4        # the schedule ran after every replay
    sd_e, sd_g = eg.state_dict(), gg.state_dict()
    for k in sd_e:
        if 'num_batches_tracked' in k or 'running_' in k:
            assert torch.equal(sd_e[k], sd_g[k]), k
    # the optimiser state written after replays is the eager one
    se, sg = eod.state_dict(), god.state_dict()
    assert all(torch.equal(se['state'][i]['exp_avg'], sg['state'][i]['exp_avg']) for i in se['state'])


def test_graph_refuses_event_timing_and_multi_rank():
    from diagan.ops import conv as C
    from diagan.utils.graph import GraphedStep
    C.TIMER = C.KernelTimer()
    try:
        with pytest.raises(RuntimeError, match="HIP-event"):
            GraphedStep(lambda: None, (), ()).capture()
    finally:
        C.TIMER = None


def _train_color_mnist(tmp_path, graph, monkeypatch, steps=10, n_dis=2):
    import os
    from diagan.datasets.predefined import get_predefined_dataset
    from diagan.models.predefined_models import get_gan_model
    from diagan.trainer.trainer import LogTrainer
    from diagan.utils.settings import set_seed
    monkeypatch.setenv("DIAGAN_QUIET", "1")
    monkeypatch.setenv("DIAGAN_GRAPH", "1" if graph else "0")
    set_seed(3)
    netG, netD, optG, optD = get_gan_model('color_mnist', model='mnist_dcgan', loss_type='ns')
    ds = get_predefined_dataset('color_mnist', num_data=32 * 7 + 8)          # a ragged last batch every 8th fetch
    dl = torch.utils.data.DataLoader(ds, batch_size=32, shuffle=True)
    out = os.path.join(str(tmp_path), "g" if graph else "e")
    t = LogTrainer(output_path=out, netD=netD, netG=netG, optD=optD, optG=optG, dataloader=dl, num_steps=steps,
                   log_dir=out, n_dis=n_dis, lr_decay='linear', device='cuda', print_steps=2, save_steps=100,
                   logit_save_steps=100, save_logits=False)
    t.train()
    torch.cuda.synchronize()
    return t


def test_log_trainer_replays_launch_bound_steps_as_a_graph(tmp_path, monkeypatch):
    """LogTrainer replays the device work of a launch-bound network's global step (MNIST-DCGAN declares
    `launch_bound`) as one hipGraph after three ordinary steps -- same parameters bit for bit as the eager loop after ten
    steps, including a step with a ragged last batch (which runs eagerly between replays), same event trace, metrics readable."""
    e = _train_color_mnist(tmp_path, False, monkeypatch)
    g = _train_color_mnist(tmp_path, True, monkeypatch)
    assert getattr(e, '_graph', None) is None and g._graph is not None
    assert torch.equal(e.netG.flat_params, g.netG.flat_params) and torch.equal(e.netD.flat_params, g.netD.flat_params)
    assert e.events == g.events
    assert e.optD._step == g.optD._step == 20 and e.optG._step == g.optG._step == 10
    sd_e, sd_g = e.netD.state_dict(), g.netD.state_dict()
    for k in sd_e:
        assert torch.equal(sd_e[k], sd_g[k]), k
    monkeypatch.delenv("DIAGAN_GRAPH")
    assert g._graph_wanted()                       # the default for this family; SNGAN stays eager (GPU-bound)
    from diagan.models.predefined_models import get_gan_model
    assert not getattr(get_gan_model('cifar10', model='sngan', loss_type='ns')[0], 'launch_bound', False)
