"""diagan_wgrad_finish_batched (csrc/conv_wgrad.hip) against float64 NumPy: the deferred epilogue of a backward pass -- sum of
the split-K partial slabs of every layer, and for spectral-norm layers the gradient through W / sigma
(reference: torch.nn.utils.spectral_norm's backward as mimicry's SNConv2d runs it, torch_mimicry/modules/spectral_norm.py:80-103;
restated in oracle/nets.py SNConv).  Layers of every block size of the kernels (1, 2, 3..4, 5..8, 9+ splits), one or two
contexts, element counts that are no multiple of a block, a bias behind the weights."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DESC = np.dtype([('p', np.uint64, 12), ('stride', np.int64), ('i', np.int32, 6)])


def _layer(rng, Co, Kp, splits, nctx, sn, bias):
    n_w = Co * Kp
    n_elem = n_w + (Co if bias else 0)
    stride = n_elem + 32                                   # slack between the splits, as the engine's slabs may have
    d = dict(Co=Co, Kp=Kp, splits=splits, nctx=nctx, sn=sn, n_w=n_w, n_elem=n_elem, stride=stride)
    d['slab'] = rng.standard_normal((nctx, splits, stride)).astype(np.float32)
    d['grad'] = rng.standard_normal(n_elem).astype(np.float32)
    d['W'] = rng.standard_normal(n_w).astype(np.float32)
    d['u'] = rng.standard_normal((nctx, Co)).astype(np.float32)
    d['v'] = rng.standard_normal((nctx, Kp)).astype(np.float32)
    d['state'] = np.stack([np.array([s, 1.0 / s], np.float32) for s in (1.7, 0.6)][:nctx])      # (sigma, 1 / sigma)
    return d


def _expected(d):
    g = d['grad'].astype(np.float64).copy()
    for c in range(d['nctx']):
        G = d['slab'][c, :, :d['n_elem']].astype(np.float64).sum(0)
        if d['sn']:
            inv = float(d['state'][c, 1])
            Gw = G[:d['n_w']]
            dot = float(Gw @ d['W'].astype(np.float64))
            uv = np.outer(d['u'][c].astype(np.float64), d['v'][c].astype(np.float64)).ravel()
            g[:d['n_w']] += (Gw - dot * inv * uv) * inv
            g[d['n_w']:] += G[d['n_w']:]
        else:
            g += G
    return g


CASES = [  # Co, Kp, splits per context, contexts, spectral norm, bias
    (64, 64, 1, 1, True, True), (128, 1152, 2, 2, True, True), (72, 96, 3, 2, True, False), (256, 2304, 4, 1, True, True),
    (64, 576, 5, 2, True, True), (32, 288, 8, 2, True, False), (16, 160, 9, 1, True, True), (64, 32, 17, 2, True, True),
    (8, 64, 40, 2, True, False), (128, 128, 2, 1, False, True), (36, 96, 6, 1, False, False), (64, 64, 33, 1, False, True),
]


def test_finish_batched_against_float64():
    from diagan import _native as nat
    from diagan.ops import conv  # noqa: F401  (registers the entry points)
    rng = np.random.default_rng(3)
    layers = [_layer(rng, *c) for c in CASES]
    dev = torch.device('cuda')
    keep, tab, total_blocks = [], np.zeros(len(layers), dtype=DESC), 0
    fbe = nat.fn("diagan_wgrad_finish_block_elems")
    for li, d in enumerate(layers):
        t = {k: torch.from_numpy(d[k]).to(dev) for k in ('slab', 'grad', 'W', 'u', 'v', 'state')}
        t['partials'] = torch.full((d['nctx'], (d['stride'] + 1023) // 1024 + 1), float('nan'), dtype=torch.float64, device=dev)
        keep.append(t)
        pp = [0] * 12
        for c in range(d['nctx']):
            pp[c] = t['slab'][c].data_ptr()
            if d['sn']:
                pp[2 + c], pp[4 + c], pp[6 + c] = t['u'][c].data_ptr(), t['v'][c].data_ptr(), t['state'][c].data_ptr()
            pp[8 + c] = t['partials'][c].data_ptr()
        pp[10] = t['grad'].data_ptr()
        pp[11] = t['W'].data_ptr() if d['sn'] else 0
        tab[li]['p'], tab[li]['stride'] = pp, d['stride']
        tab[li]['i'] = [d['splits'], d['n_elem'], d['n_w'], d['Kp'], d['nctx'], total_blocks]
        be = fbe(d['splits'])
        assert be in (1024, 2048, 4096, 8192)
        total_blocks += (d['n_elem'] + be - 1) // be
    tdev = torch.from_numpy(tab.view(np.uint8).copy()).to(dev)
    nat.call("diagan_wgrad_finish_batched", tdev.data_ptr(), len(layers), total_blocks, 1, nat.current_stream())
    torch.cuda.synchronize()
    for d, t in zip(layers, keep):
        want = _expected(d)
        got = t['grad'].cpu().numpy().astype(np.float64)
        scale = np.abs(want).max()
        assert np.abs(got - want).max() < 2e-5 * scale, (d['Co'], d['Kp'], d['splits'], d['nctx'], d['sn'],
                                                        np.abs(got - want).max() / scale)
        # nothing behind the layer's end is touched: the slack of split 0 keeps its values
        for c in range(d['nctx']):
            assert np.array_equal(t['slab'][c, 0, d['n_elem']:].cpu().numpy(), d['slab'][c, 0, d['n_elem']:])


def test_finish_batched_is_deterministic():
    from diagan import _native as nat
    from diagan.ops import conv  # noqa: F401
    rng = np.random.default_rng(4)
    d = _layer(rng, 64, 576, 12, 2, True, True)
    dev = torch.device('cuda')
    outs = []
    for _ in range(2):
        t = {k: torch.from_numpy(d[k]).to(dev) for k in ('slab', 'grad', 'W', 'u', 'v', 'state')}
        part = torch.zeros((2, (d['stride'] + 1023) // 1024 + 1), dtype=torch.float64, device=dev)
        tab = np.zeros(1, dtype=DESC)
        tab[0]['p'] = [t['slab'][0].data_ptr(), t['slab'][1].data_ptr(), t['u'][0].data_ptr(), t['u'][1].data_ptr(),
                       t['v'][0].data_ptr(), t['v'][1].data_ptr(), t['state'][0].data_ptr(), t['state'][1].data_ptr(),
                       part[0].data_ptr(), part[1].data_ptr(), t['grad'].data_ptr(), t['W'].data_ptr()]
        tab[0]['stride'] = d['stride']
        tab[0]['i'] = [d['splits'], d['n_elem'], d['n_w'], d['Kp'], 2, 0]
        be = nat.fn("diagan_wgrad_finish_block_elems")(d['splits'])
        tdev = torch.from_numpy(tab.view(np.uint8).copy()).to(dev)
        nat.call("diagan_wgrad_finish_batched", tdev.data_ptr(), 1, (d['n_elem'] + be - 1) // be, 1, nat.current_stream())
        torch.cuda.synchronize()
        outs.append(t['grad'].cpu().numpy())
    assert np.array_equal(outs[0], outs[1])
