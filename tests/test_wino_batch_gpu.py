"""GPU: Winograd weights transformed ahead of their launches, many layers per launch (include/diagan_hip.h:
diagan_wino_weights_batched, diagan_conv_gemm_weights_hint, diagan_conv_gemm_last_weight_format; host side
diagan/ops/conv.py WinoWeightBatch).  The transform arithmetic is the per-launch kernels' own, so everything must stay
bit-identical; what changes is the number of launches."""
import pytest
import torch

import bench
from test_wino_gpu import make
from test_conv_gpu import nhwc

pytestmark = pytest.mark.gpu


def test_hint_protocol_right_and_wrong_format():
    """a launch uses the caller's transformed weights iff the hinted format is the one it needs; a hint holds for one call"""
    from diagan import _native as nat
    from diagan.ops import conv as C
    geom, x, w, wp = make(8, 16, 16, 64, 64, seed=51)
    xg = nhwc(x).cuda()
    ref = C.conv_fwd(geom, xg, wp, tile_cfg=13)
    fmt, n0 = C.last_weight_format()
    assert fmt[0] == 40 and fmt[1] == 0 and fmt[2] == 1.0 and fmt[3] == 36 * 64 * 64
    # the caller transforms the weights itself (one job) and hands them over
    batch = C.WinoWeightBatch()
    site = batch.site(lambda: wp, 64, 64, geom.Kp)
    site.fmt, site.used = fmt, True
    batch.prepare(version=1)
    assert site.ready == 1 and site.u.numel() == fmt[3]
    y = C.conv_fwd(geom, xg, wp, tile_cfg=13, wsite=site, wversion=1)
    assert torch.equal(y, ref) and C.last_weight_format()[1] == n0          # no per-launch transform was issued
    y = C.conv_fwd(geom, xg, wp, tile_cfg=13)                               # the hint is gone: this call transforms again
    assert torch.equal(y, ref) and C.last_weight_format()[1] == n0 + 1
    # F(2x2) needs another format: the F(4x4) buffer is ignored, the result is right, and the site re-learns
    r9 = C.conv_fwd(geom, xg, wp, tile_cfg=9)
    n1 = C.last_weight_format()[1]
    y = C.conv_fwd(geom, xg, wp, tile_cfg=9, wsite=site, wversion=1)
    assert torch.equal(y, r9) and C.last_weight_format()[1] == n1 + 1 and site.fmt[0] == 2 and site.ready is None
    batch.prepare(version=2)
    y = C.conv_fwd(geom, xg, wp, tile_cfg=9, wsite=site, wversion=2)
    assert torch.equal(y, r9) and C.last_weight_format()[1] == n1 + 1
    # a data gradient (taps reversed) and the up-sampled-input mode (scale 1/16) are formats of their own
    wd = torch.zeros(64, geom.Kd, device="cuda")
    C.pack_weights(wp, 64, 64, 9, geom.Kp, geom.Kd, Wd=wd)
    C.conv_dgrad(geom, xg, wd, (16, 16), tile_cfg=13)
    assert C.last_weight_format()[0][:3] == (40, 1, 1.0)
    C.set_winograd4('force-pool')
    try:
        C.conv_fwd(geom, xg[:, ::2, ::2].contiguous(), wp, up_in=True)
        assert C.last_weight_format()[0][:3] == (40, 0, 0.0625)
    finally:
        C.set_winograd4(None)
    C.conv_fwd(geom, xg, wp, tile_cfg=7)
    assert C.last_weight_format()[0][0] == 0                                 # implicit GEMM: no Winograd weights at all


@pytest.mark.parametrize("workload", ["sngan32", "sngan64"])
def test_training_steps_bit_identical_with_fewer_launches(workload, monkeypatch):
    """three global steps with and without the batched transforms: same parameters bit for bit; after the first (learning)
    step no launch transforms its own weights any more"""
    from diagan.ops import conv as C
    dataset, res, _ = bench.WORKLOADS[workload]
    dev = torch.device("cuda", 0)

    def run(on):
        monkeypatch.setattr(C, "WINO_BATCH", on)
        nets = bench.build_models(dataset, 'ns', 1, dev)
        g = torch.Generator().manual_seed(3)
        batches = [(torch.rand(64, 3, res, res, generator=g) * 2 - 1).to(dev) for _ in range(10)]
        step = bench.make_global_step(*nets, batches, 5, 50000, dev)
        torch.cuda.manual_seed(5)
        counts = []
        for _ in range(3):
            c0 = C.last_weight_format()[1]
            step()
            counts.append(C.last_weight_format()[1] - c0)
        torch.cuda.synchronize()
        return counts, nets[0].flat_params.clone(), nets[1].flat_params.clone()

    c_off, g_off, d_off = run(False)
    c_on, g_on, d_on = run(True)
    assert torch.equal(g_off, g_on) and torch.equal(d_off, d_on)
    assert c_off[0] == c_off[1] == c_off[2] >= 40
    assert c_on[1] == 0 and c_on[2] == 0 and c_on[0] < c_off[0]
