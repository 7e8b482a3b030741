"""Convolution ops over the C ABI: implicit-GEMM forward / data-gradient / weight-gradient,
spectral-norm power iteration and GEMM operand packing.

All tensors are torch CUDA fp32.  Activations are NHWC ([B,H,W,C], C % 4 == 0); weights live in
the packed layout Wp[Co][Kp], k = (r*S+s)*Ci + c, Kp = roundup(R*S*Ci, 32)."""
import torch

from diagan import _native as nat

P, I, F, I64 = nat.c_void_p, nat.c_int, nat.c_f32, nat.c_i64
nat.register("diagan_conv_gemm", [P, P, P, P, P, I, P, F, P, P, I, F, P, P, I] + [I] * 15 + [P, I64, P, I, P])
nat.register("diagan_conv_gemm_pick_ksplit", [I, I, I, I])
nat.register("diagan_conv_gemm_pick_cfg", [I, I, I, I])
nat.register("diagan_conv_wino_supported", [I] * 12)
nat.register("diagan_conv_gemm_set_wino", [I])
nat.register("diagan_conv_gemm_get_wino", [])
nat.register("diagan_conv_gemm_set_wino4", [I])
nat.register("diagan_conv_gemm_set_splitk_fused", [I])
nat.register("diagan_conv_gemm_set_splitk_tickets", [P, I64])
nat.register("diagan_conv_gemm_next_opts", [P])
nat.register("diagan_conv_gemm_last_cfg", [])
nat.register("diagan_conv_gemm_set_x3", [I])
nat.register("diagan_conv_gemm_get_x3", [])
nat.register("diagan_conv_gemm_set_x3b", [I])
nat.register("diagan_conv_gemm_get_x3b", [])
nat.register("diagan_conv_gemm_x3b_force_form", [I])
nat.register("diagan_conv_gemm_out_map", [I] * 9)
nat.register("diagan_conv_gemm_final_cfg", [I] * 15 + [I64] + [I] * 4)
nat.register("diagan_conv_wgrad_batched", [P, I, P])
nat.register("diagan_conv_wgrad_batch_max", [])
nat.register("diagan_conv_wgrad_batch_class", [I] * 14)
nat.register("diagan_conv_gemm_set_wino4x", [I])
nat.register("diagan_conv_gemm_get_wino4x", [])
nat.register("diagan_conv_wino4_pool_used", [I] * 5 + [I64])
nat.register("diagan_conv_wino4_upin_supported", [I] * 13 + [I64, I])
nat.register("diagan_conv_gemm_weights_hint", [P, I, I, F])
nat.register("diagan_conv_gemm_last_weight_format", [P, P, P, P, P])
nat.register("diagan_wino_weight_blocks", [I, I])
nat.register("diagan_wino_weights_batched", [P, I, I, P])
nat.register("diagan_conv_wgrad_uses_wino", [I] * 13)
nat.register("diagan_conv_wgrad_splits_geom", [I] * 14)
nat.register("diagan_conv_gemm_set_x3_pieces", [I])
nat.register("diagan_conv_gemm_get_x3_pieces", [])
nat.register("diagan_conv_wgrad_uses_x3", [I] * 15 + [I64])
nat.register("diagan_conv_wgrad_set_x3", [I])
nat.register("diagan_conv_gemm_pick_cfg_geom", [I] * 15 + [I64])
nat.register("diagan_conv_gemm_pick_cfg_grouped", [I] * 15 + [I64, I])
nat.register("diagan_conv_gemm_tile_rows", [I])
nat.register("diagan_conv_wino_pool_supported", [I] * 14 + [I64])
nat.register("diagan_conv_wino_unpool_supported", [I] * 13 + [I64])
nat.register("diagan_conv_gemm_tile_cols", [I])
nat.register("diagan_conv_gemm_set_stamp_buffer", [P, I64])
nat.register("diagan_conv_gemm_tune", [I, I, I])
nat.register("diagan_conv3x3_co4_supported", [I] * 8)
nat.register("diagan_conv3x3_co4", [P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, P])
nat.register("diagan_conv3x3_ci4_supported", [I] * 8)
nat.register("diagan_conv3x3_ci4", [P, P, P, P, F, P, P, I, I, I, I, I, I, P])
nat.register("diagan_conv3x3_co4_wgrad_supported", [I] * 8)
nat.register("diagan_conv3x3_co4_wgrad_splits", [I, I])
nat.register("diagan_conv3x3_co4_wgrad", [P, P, P, I64, I64, P, P, I, I, I, I, I, I, P])
nat.register("diagan_conv_wgrad", [P, P, P, I, I, I64, I64, P, P, I] + [I] * 14 + [P])
nat.register("diagan_pack_batched", [P, I, I, I, I, I, P])
nat.register("diagan_wgrad_finish_batched", [P, I, I64, I, P])
nat.register("diagan_wgrad_finish_block_elems", [I])
nat.register("diagan_conv_wgrad_splits", [I, I, I])
nat.register("diagan_wgrad_reduce", [P, I, I64, P, I, P, P, P])
nat.register("diagan_sn_power_iter", [P, P, P, P, P, P, P, I, I, F, I, P])
nat.register("diagan_sn_prepare_batched", [P, I, I, I, I, I, F, I, I, P])
nat.register("diagan_pack_weights", [P, P, P, P, I, I, I, I, I, P])
nat.register("diagan_sn_grad_fix", [P, P, I, P, P, P, P, I, I, I, P])

PRO_NONE, PRO_RELU, PRO_AFFINE_RELU, PRO_LRELU, PRO_AFFINE = 0, 1, 2, 3, 4
# weight gradients only: the gathered image is the (H+1) x (W+1) grid of 0.25 * 2x2 box sums of x / relu(x) (eltwise.boxsum2), summed by
# the kernel's loader from the H x W tensor it is handed (csrc/conv_common.h)
PRO_BOX, PRO_BOX_RELU = 5, 6
# kernel names as rocprofv3 prints them (template arguments: BM, BN, WM, WN, BK, PRO; PRO = -1: run-time mode)
# kernel names as rocprofv3 prints them (template arguments: BM, BN, WM, WN, BK, PRO, STAMP, FP; PRO = -1: run-time mode)
TILE_SHAPES = {1: (128, 128, 2, 2, 32, False), 2: (128, 64, 2, 2, 32, False), 3: (64, 64, 2, 2, 32, False),
               4: (128, 64, 4, 1, 32, False), 5: (256, 64, 4, 1, 32, False), 6: (64, 64, 2, 2, 64, False),
               7: (64, 64, 2, 2, 32, True), 8: (128, 64, 2, 2, 32, True), 14: (64, 64, 2, 2, 32, True)}


def gemm_kernel_name(cfg, mode, Co=128, w4pool=False, Ci=0, RS=9, mapped=False):
    """the kernel's name as rocprofv3 prints it (the F(4x4) kernel's third template argument: the bf16-split build, which a
    launch takes only with the switch on and a K loop of a multiple of four steps)"""
    if cfg in (11, 12) and w4pool:            # the pooled launches on the F(4x4) kernel (MODE 1 / 2)
        return f"conv_wino4_kernel<{mode if cfg == 11 else 0},{cfg - 10},false>"
    if cfg == 9:
        return f"conv_wino_kernel<{mode}>"
    if cfg in (13, 15):
        x3 = "true" if (nat.fn("diagan_conv_gemm_get_wino4x")() > 0 and Ci % 32 == 0) else "false"
        return f"conv_wino4_kernel<{mode},{0 if cfg == 13 else 3},{x3}>"
    if cfg == 11:
        return f"conv_wino_pool_kernel<{mode},false,{2 if Co % 128 == 0 else 1}>"
    if cfg == 12:
        return f"conv_wino_pool_kernel<0,true,{2 if Co % 128 == 0 else 1}>"
    if cfg == 17:
        return f"conv_gemm_x3b_kernel<{mode},{'true' if mapped else 'false'}>"
    if cfg == 16:
        return f"conv_gemm_x3_kernel<{mode}>"      # (3x3 layers: the only lone-tile launches of the networks)
    bm, bn, wm, wn, bk, fp = TILE_SHAPES[cfg]
    kg = ",2" if cfg == 14 else ",1"          # K-groups per workgroup (rocprofv3 prints the defaulted template argument too)
    return f"conv_gemm_kernel<{bm},{bn},{wm},{wn},{bk},{mode},false,{'true' if fp else 'false'}{kg}>"


class KernelTimer:
    """Optional per-launch HIP-event timing of the GEMM kernels (bench.py's roofline leg).
    Events are recorded on torch's current stream, which is the stream the kernels are launched on."""

    def __init__(self, only=None):
        self.records = []          # (kernel name, flop, start event, stop event)
        self.only = only           # optional set of kernel names: launches of other kernels are not bracketed

    def wants_any(self):
        """False for a timer that brackets nothing (bench.py's timed region on launch-bound workloads)"""
        return self.only is None or len(self.only) > 0

    def begin(self, name=None):
        if self.only is not None and name not in self.only:
            return None
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        return ev

    def end(self, name, flop, start, shape=None):
        stop = torch.cuda.Event(enable_timing=True)
        stop.record()
        self.records.append((name, flop, start, stop, shape))

    def summary(self):
        """name -> dict(launches, flop, seconds); call after a device synchronize."""
        out = {}
        for name, flop, s, e, _ in self.records:
            d = out.setdefault(name, dict(launches=0, flop=0.0, seconds=0.0))
            d['launches'] += 1
            d['flop'] += flop
            d['seconds'] += s.elapsed_time(e) * 1e-3
        return out

    def by_shape(self):
        """(kernel, M, N, K, tag) -> dict(launches, flop, seconds)"""
        out = {}
        for name, flop, s, e, shape in self.records:
            d = out.setdefault((name,) + tuple(shape or ()), dict(launches=0, flop=0.0, seconds=0.0))
            d['launches'] += 1
            d['flop'] += flop
            d['seconds'] += s.elapsed_time(e) * 1e-3
        return out


TIMER = None      # set to a KernelTimer by bench.py
_NAME_CACHE = {}


# ---- Winograd weights transformed ahead of their launches, many layers per launch (include/diagan_hip.h, round 4) -------------
# Every Winograd launch otherwise starts with a 5-7 us weight-transform kernel of its own (48 / 124 per SNGAN-32 / -64 step).
# A `WinoWeights` object stands for ONE call site of one layer (its forward, its data gradient, ...): the first pass through
# the site runs as before and notes which format the launch needed; from then on the network's `WinoWeightBatch` transforms
# all its sites' weights in ONE launch per pass and the launches take the ready buffer.
# A launch that turns out to need another format than last time (another batch size -> another kernel) ignores the hint
# and transforms its own weights, and the site re-learns.  DIAGAN_WINO_BATCH=0: every launch transforms its own, as before.
import ctypes as _ct
import os as _os

WINO_BATCH = _os.environ.get("DIAGAN_WINO_BATCH", "1") != "0"


class WinoWeights:
    __slots__ = ("w_of", "Co", "Ci", "Kp", "fmt", "u", "ready", "used")

    def __init__(self, w_of, Co, Ci, Kp):
        self.w_of, self.Co, self.Ci, self.Kp = w_of, Co, Ci, Kp       # w_of(): the packed operand this site's launches read
        self.fmt, self.u, self.ready, self.used = None, None, None, False


_LWF = (_ct.c_int(), _ct.c_int(), _ct.c_float(), _ct.c_int64(), _ct.c_int64())
_LWF_REFS = tuple(_ct.byref(v) for v in _LWF)


def last_weight_format():
    # (the five out-parameters are made once: this runs behind every convolution launch of a network pass)
    nat.fn("diagan_conv_gemm_last_weight_format")(*_LWF_REFS)
    k, f, sc, n, cnt = _LWF
    return (k.value, f.value, sc.value, n.value), cnt.value


class WinoWeightBatch:
    """The call sites of one network pass (e.g. "D forward, pair mode") whose transformed weights are made together."""

    def __init__(self):
        self.sites = []
        self._table_key, self._tables = None, None

    def site(self, w_of, Co, Ci, Kp):
        s = WinoWeights(w_of, Co, Ci, Kp)
        self.sites.append(s)
        return s

    def prepare(self, version):
        """Transform the weights of every site with a known format whose buffer is not at `version` yet, in one launch.
        Called at the start of the pass, after the operands the sites read have been written."""
        if not WINO_BATCH:
            return
        import numpy as np
        todo = [s for s in self.sites if s.fmt is not None and s.fmt[0] and s.used and s.ready != version]
        if not todo:
            return
        key = tuple((id(s), s.fmt, s.w_of().data_ptr()) for s in todo)
        if key != self._table_key:
            desc = np.dtype([('w', np.uint64), ('u', np.uint64), ('i', np.int32, 6), ('scale', np.float32), ('pad', np.int32)])
            tab, blk = np.zeros(len(todo), dtype=desc), 0
            for j, s in enumerate(todo):
                w = s.w_of()
                if s.u is None or s.u.numel() != s.fmt[3] or s.u.device != w.device:
                    s.u = torch.empty(s.fmt[3], dtype=torch.float32, device=w.device)
                tab[j]['w'], tab[j]['u'] = w.data_ptr(), s.u.data_ptr()
                tab[j]['i'] = [s.Co, s.Ci, s.Kp, s.fmt[0], s.fmt[1], blk]
                tab[j]['scale'] = s.fmt[2]
                blk += nat.fn("diagan_wino_weight_blocks")(s.Co, s.Ci)
            dev_tab = torch.from_numpy(tab.view(np.uint8).copy()).to(todo[0].w_of().device)
            self._table_key, self._tables = key, (dev_tab, len(todo), blk)
        tab, n, blocks = self._tables
        nat.call("diagan_wino_weights_batched", nat.ptr(tab), n, blocks, nat.current_stream())
        for s in todo:
            s.ready = version


def _hint(wsite, version):
    """before a diagan_conv_gemm call of this site: hand over the ready buffer, if there is one for this version"""
    if wsite is not None and WINO_BATCH and wsite.u is not None and wsite.ready == version and wsite.fmt is not None:
        nat.call("diagan_conv_gemm_weights_hint", wsite.u.data_ptr(), wsite.fmt[0], wsite.fmt[1], wsite.fmt[2])


def _learn(wsite):
    """after the call: the format it needed (kind 0: no Winograd kernel ran)"""
    if wsite is not None and WINO_BATCH:
        fmt, _ = last_weight_format()
        wsite.used = True
        if fmt != wsite.fmt:
            wsite.fmt, wsite.ready = fmt, None


def set_winograd(mode):
    """True / False: allow / forbid the Winograd kernel for auto-selected tile configurations; None: the default
    (on, or what DIAGAN_WINO says)"""
    nat.call("diagan_conv_gemm_set_wino", -1 if mode is None else (1 if mode else 0))


def set_winograd4x(mode):
    """True / False: the F(4x4,3x3) launches with K loops of a multiple of four steps run on the bf16 matrix pipe with exactly
    split operands (conv_wino4.hip, X3) / on the fp32 one; None: what DIAGAN_WINO4_X3 says"""
    nat.call("diagan_conv_gemm_set_wino4x", -1 if mode is None else (1 if mode else 0))


def set_winograd4(mode):
    """True / False: allow / forbid the F(4x4,3x3) kernel (tile_cfg 13, and the pooled launches 11 / 12 on it) for auto-selected
    launches; None: the default (on, or what DIAGAN_WINO4 says); 'force-pool': the pooled launches take it at any size (tests)"""
    nat.call("diagan_conv_gemm_set_wino4", -1 if mode is None else (2 if mode == 'force-pool' else (1 if mode else 0)))


def round_up(x, m):
    return (x + m - 1) // m * m


class Geom:
    """Gather geometry of one convolution (see include/diagan_hip.h).

    kind 'conv'  : y = conv2d(x, stride, pad)            fwd params (s,+1,-p,1)
    kind 'convT' : y = conv_transpose2d(x, stride, pad)  fwd params (1,-1,+p,s)
    The data-gradient of either kind is the other kind's gather with Ci/Co swapped."""

    def __init__(self, kind, Ci, Co, R, S, stride=1, pad=0):
        assert kind in ('conv', 'convT')
        self.kind, self.Ci, self.Co, self.R, self.S, self.stride, self.pad = kind, Ci, Co, R, S, stride, pad
        self.Kp = round_up(R * S * Ci, 32)      # packed forward/wgrad operand row
        self.Kd = round_up(R * S * Co, 32)      # packed data-gradient operand row

    def out_hw(self, Hi, Wi):
        if self.kind == 'conv':
            return ((Hi + 2 * self.pad - self.R) // self.stride + 1,
                    (Wi + 2 * self.pad - self.S) // self.stride + 1)
        return ((Hi - 1) * self.stride - 2 * self.pad + self.R, (Wi - 1) * self.stride - 2 * self.pad + self.S)

    def fwd_params(self):
        return (self.stride, 1, -self.pad, 1) if self.kind == 'conv' else (1, -1, self.pad, self.stride)

    def dgrad_params(self):
        return (1, -1, self.pad, self.stride) if self.kind == 'conv' else (self.stride, 1, -self.pad, 1)


def _pro3(pro):
    """(mode, scale, shift) of a prologue tuple for the kernels that take ONE batch (weight gradients)."""
    if pro is None:
        return PRO_NONE, None, None
    if len(pro) > 3 and pro[3]:
        raise RuntimeError("grouped BatchNorm prologues exist for the forward kernels only")
    return tuple(pro)[:3]


def set_gemm_x3(on):
    """True / False: the lone-tile implicit-GEMM launches (tile_cfg 14) run on the bf16 matrix pipe with exactly split operands where
    their geometry qualifies (csrc/conv_gemm_x3.hip) / on the fp32 pipe; None: what DIAGAN_GEMM_X3 says"""
    nat.call("diagan_conv_gemm_set_x3", -1 if on is None else (1 if on else 0))
    # ... and with it the large-launch form (csrc/conv_gemm_x3b.hip, round 6): the callers of this switch want the fp32 fmaf chain or not
    nat.call("diagan_conv_gemm_set_x3b", -1 if on is None else (1 if on else 0))


def set_gemm_x3b(on):
    """the same for the 128 x 128 / 256 x 128 split-operand kernel alone (tile_cfg 17); None: what DIAGAN_GEMM_X3B says"""
    nat.call("diagan_conv_gemm_set_x3b", -1 if on is None else (1 if on else 0))


def set_x3_pieces(n):
    """pieces per operand of the large split-operand kernels: 3 = fp32-grade (default), 2 = opt-in (operands at ~2^-16, 1.5e-5-2e-5 of the
    output scale per layer, 1.4-1.5x shorter launches); None: what DIAGAN_X3_PIECES says"""
    nat.call("diagan_conv_gemm_set_x3_pieces", 0 if n is None else int(n))


def set_wgrad_x3(on):
    """the split-operand weight-gradient kernel (csrc/conv_wgrad_x3.hip): True / False, 2 = on for every geometry it can run
    (no work floor: tests); None: what DIAGAN_WGRAD_X3 says.  (set_gemm_x3(False), the exact-fp32 mode, turns it off too.)"""
    nat.call("diagan_conv_wgrad_set_x3", -1 if on is None else (int(on) if on else 0))


def wgrad_uses_x3(geom, B, Hi, Wi, Ho, Wo, mode=0, bias_off=-1):
    sy, dr, off, up = geom.fwd_params()
    return bool(nat.fn("diagan_conv_wgrad_uses_x3")(B, Hi, Wi, geom.Ci, Ho, Wo, geom.Co, geom.R, geom.S, sy, dr, off, up, geom.Kp,
                                                    mode, bias_off))


_TICKETS = {}


def set_splitk_fused(on):
    """True / False: the split-K launches' last-arriving workgroup runs the epilogue / a second launch does; None: the default.
    (Process-wide, diagnostics / tests.)  The ticket buffer of the in-kernel combine is the caller's: one is allocated here and
    registered with the library while the switch is on."""
    nat.call("diagan_conv_gemm_set_splitk_fused", -1 if on is None else (1 if on else 0))
    if on:
        dev = torch.cuda.current_device()
        if dev not in _TICKETS:
            _TICKETS[dev] = torch.zeros(1 << 16, dtype=torch.int32, device=torch.device('cuda', dev))
        nat.call("diagan_conv_gemm_set_splitk_tickets", _TICKETS[dev].data_ptr(), _TICKETS[dev].numel())
    else:
        nat.call("diagan_conv_gemm_set_splitk_tickets", None, 0)


class ConvOpts(_ct.Structure):
    """diagan_conv_opts (include/diagan_hip.h): the selection options of ONE call -- `with conv_opts(wino=0): conv_fwd(...)` hands them
    to the next diagan_conv_gemm call of this thread; nothing process-wide is touched"""
    _fields_ = [(n, _ct.c_int32) for n in ("wino", "wino4", "wino4x", "gemm_x3", "gemm_x3b", "splitk_fused", "force_ksplit", "tune")] + \
               [("tickets", _ct.c_void_p), ("ticket_slots", _ct.c_int64)]


def next_opts(tickets=None, **fields):
    """options of the next convolution launch of this thread: wino / wino4 / wino4x / gemm_x3 / gemm_x3b / splitk_fused (-1 default, 0, 1),
    force_ksplit (0 = policy), tune (-1 default); tickets: a zeroed int32 CUDA tensor for the in-kernel split-K combine"""
    o = ConvOpts(-1, -1, -1, -1, -1, -1, 0, -1, None, 0)
    for k, v in fields.items():
        setattr(o, k, int(v))
    if tickets is not None:
        o.tickets, o.ticket_slots = tickets.data_ptr(), tickets.numel()
    nat.call("diagan_conv_gemm_next_opts", _ct.byref(o))


def last_cfg():
    """tile configuration the last convolution launch of this thread resolved to"""
    return nat.fn("diagan_conv_gemm_last_cfg")()


def wg_x_shape(x, pro):
    """shape of the tensor a weight gradient GATHERS from: x's own, or the box-sum grid over x for the PRO_BOX modes"""
    B, H, W, Ci = x.shape
    if pro is not None and int(pro[0]) >= PRO_BOX:
        return B, H + 1, W + 1, Ci
    return B, H, W, Ci


def _chk(t, name):
    if t is None:
        return
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise RuntimeError(f"{name}: expected a contiguous CUDA float32 tensor, got "
                           f"{t.dtype} {t.device} contiguous={t.is_contiguous()}")


def _gemm(x, w, out, geo_params, R, S, Kp, bias, residual, mask_src, mask_slope, pro, out_scale, tile_cfg,
          res_relu=False, row_scale=None, want_stats=False, wino=True, res_up=False, pool=False, unpool=False, up_in=False,
          wsite=None, wversion=None, res_unpool=False, out_map=None):
    """want_stats: also return (partials, tiles) -- per-tile column sums of y, y^2 from the epilogue
    (None when the problem takes the split-K / small-Co path; the caller then reduces y itself)."""
    B, Hi, Wi, Ci = x.shape
    _, Ho, Wo, Co = out.shape
    if out_map is not None:               # `out` is the larger tensor the launch's own Ho x Wo grid is mapped into
        Ho, Wo = out_map[9], out_map[10]
    if up_in:                             # `x` is the half-resolution input whose bilinear x2 is convolved: tile_cfg 15
        Hi, Wi, tile_cfg = 2 * Hi, 2 * Wi, 15
        if pool or unpool or mask_src is not None:
            raise RuntimeError("conv_gemm: the up-sampled-input launch takes no pooling and no backward mask")
    if unpool:                            # `x` is the half-resolution (pooled) gradient: tile_cfg 12
        Hi, Wi, tile_cfg = 2 * Hi, 2 * Wi, 12
    if pool:                              # `out` (and `residual`) are the 2x2-average-pooled tensors: tile_cfg 11
        Ho, Wo, tile_cfg = 2 * Ho, 2 * Wo, 11
        if want_stats or res_up or mask_src is not None:
            raise RuntimeError("conv_gemm: the pooled launch takes no statistics, mask or half-resolution residual")
    # pro = (mode, scale, shift[, group_imgs]): group_imgs > 0 -> scale / shift are [G, Ci], one row per group of images
    mode, scale, shift, group_imgs = (tuple(pro) + (0,))[:4] if pro is not None else (PRO_NONE, None, None, 0)
    for t, n in ((x, 'x'), (w, 'w'), (out, 'out'), (bias, 'bias'), (residual, 'residual'),
                 (mask_src, 'mask_src'), (scale, 'pro_scale'), (shift, 'pro_shift')):
        _chk(t, n)
    if w.shape != (Co, Kp):
        raise RuntimeError(f"conv_gemm: packed weight shape {tuple(w.shape)} != ({Co}, {Kp})")
    for t, n in ((residual, 'residual'), (mask_src, 'mask_src')):
        want = (B, Ho // 2, Wo // 2, Co) if ((res_up or res_unpool) and n == 'residual') else tuple(out.shape)
        if t is not None and tuple(t.shape) != want:
            raise RuntimeError(f"conv_gemm: {n} shape {tuple(t.shape)} != {want}")
    if (res_up or res_unpool) and (residual is None or res_relu or not wino or (res_up and res_unpool)):
        raise RuntimeError("conv_gemm: res_up / res_unpool need a half-resolution residual, no ReLU on it, and the Winograd kernel")
    sy, dr, off, up = geo_params
    stats = None
    if (mask_src is None and row_scale is None and out_scale == 1.0 and not res_relu and tile_cfg == 0 and not res_unpool
            and not want_stats and nat.fn("diagan_conv3x3_co4_supported")(Ci, Co, R, S, sy, dr, off, up)):
        t0 = TIMER.begin("conv3x3_co4_kernel") if TIMER is not None else None
        nat.call("diagan_conv3x3_co4", nat.ptr(x), nat.ptr(w), nat.ptr(out), nat.ptr(bias), nat.ptr(residual),
                 nat.ptr(scale), nat.ptr(shift), mode, B, Hi, Wi, Ci, dr, off, Kp, group_imgs, nat.current_stream())
        if t0 is not None:
            TIMER.end("conv3x3_co4_kernel", 2.0 * B * Ho * Wo * Co * R * S * Ci, t0,
                      (B * Ho * Wo, Co, R * S * Ci, f"pro{mode}"))
        return out
    if (mode == PRO_NONE and residual is None and mask_src is None and not res_relu and tile_cfg == 0 and not want_stats
            and nat.fn("diagan_conv3x3_ci4_supported")(Ci, Co, R, S, sy, dr, off, up)):
        t0 = TIMER.begin("conv3x3_ci4_kernel") if TIMER is not None else None
        nat.call("diagan_conv3x3_ci4", nat.ptr(x), nat.ptr(w), nat.ptr(out), nat.ptr(bias), out_scale,
                 nat.ptr(row_scale[0]) if row_scale else None, nat.ptr(row_scale[1]) if row_scale else None,
                 (B // 2) * Ho * Wo if row_scale else 0, B, Hi, Wi, Co, Kp, nat.current_stream())
        if t0 is not None:
            TIMER.end("conv3x3_ci4_kernel", 2.0 * B * Ho * Wo * Co * R * S * Ci, t0, (B * Ho * Wo, Co, R * S * Ci, "pro0"))
        return out
    ws = _splitk_ws(x.device)
    if tile_cfg == 0 and not wino:        # caller keeps to the implicit GEMM (the StyleGAN2 autograd ops by default)
        tile_cfg = nat.fn("diagan_conv_gemm_pick_cfg")(B * Ho * Wo, Co, Kp, 0 if want_stats else 1)
    if want_stats:
        # statistics from the epilogue always win over split-K + a separate reduction pass over y (G-32 block2,
        # M=4096: 60 us unsplit with statistics vs 54 + 6 (second stage) + 20 (column reduction) us)
        M = B * Ho * Wo
        cfg = tile_cfg or nat.fn("diagan_conv_gemm_pick_cfg_grouped")(B, Hi, Wi, Ci, Ho, Wo, Co, R, S, sy, dr, off, up, Kp, 0,
                                                                       ws.numel(), group_imgs * Ho * Wo)
        bm = nat.fn("diagan_conv_gemm_tile_rows")(cfg)
        tiles = (M + bm - 1) // bm
        stats = (torch.empty((tiles, 2, Co), dtype=torch.float32, device=x.device), tiles)
        tile_cfg = cfg
    kname = None
    if TIMER is not None and TIMER.wants_any():
        # (cached per call signature: on launch-bound workloads the name lookup itself was 6 ms of host time per step)
        key = (tile_cfg, B, Hi, Wi, Ci, Ho, Wo, Co, R, S, sy, dr, off, up, Kp, mode, want_stats, group_imgs, out_map is not None,
               mask_src is None, row_scale is None, res_relu, res_up, res_unpool)
        kname = _NAME_CACHE.get(key)
        modes = (nat.fn("diagan_conv_gemm_get_wino")(), nat.fn("diagan_conv_gemm_get_wino4x")(), nat.fn("diagan_conv_gemm_get_x3")(),
                 nat.fn("diagan_conv_gemm_get_x3b")())
        if kname is None or _NAME_CACHE.get('modes') != modes:
            if _NAME_CACHE.get('modes') != modes:
                _NAME_CACHE.clear()
                _NAME_CACHE['modes'] = modes
            allow = 0 if want_stats else 1
            # (the launch's own answer, upgrades to the split-operand kernels included: diagan_conv_gemm_final_cfg)
            plain = mask_src is None and row_scale is None and not res_relu and not res_up and not res_unpool
            kname = gemm_kernel_name(tile_cfg or nat.fn("diagan_conv_gemm_final_cfg")(
                B, Hi, Wi, Ci, Ho, Wo, Co, R, S, sy, dr, off, up, Kp, allow, ws.numel(), group_imgs * Ho * Wo, mode,
                1 if plain else 0, 1 if want_stats else 0), mode, Co,
                w4pool=tile_cfg in (11, 12) and bool(nat.fn("diagan_conv_wino4_pool_used")(B, Ho, Wo, Ci, Co, ws.numel())), Ci=Ci, RS=R * S,
                mapped=out_map is not None)
            _NAME_CACHE[key] = kname
    t0 = TIMER.begin(kname) if TIMER is not None else None
    _hint(wsite, wversion)
    if out_map is not None:
        nat.call("diagan_conv_gemm_out_map", *out_map[:9])
    nat.call("diagan_conv_gemm", nat.ptr(x), nat.ptr(w), nat.ptr(out), nat.ptr(bias), nat.ptr(residual),
             (1 if res_relu else 0) | (2 if res_up else 0) | (4 if res_unpool else 0), nat.ptr(mask_src), mask_slope, nat.ptr(scale), nat.ptr(shift), mode, out_scale,
             nat.ptr(row_scale[0]) if row_scale else None, nat.ptr(row_scale[1]) if row_scale else None,
             (B // 2) * Ho * Wo if row_scale else 0,
             B, Hi, Wi, Ci, Ho, Wo, Co, R, S, sy, dr, off, up, Kp, tile_cfg, nat.ptr(ws), ws.numel(),
             nat.ptr(stats[0]) if stats else None, group_imgs * Ho * Wo, nat.current_stream())
    _learn(wsite)
    if t0 is not None:
        TIMER.end(kname, 2.0 * B * Ho * Wo * Co * R * S * Ci, t0,
                  (B * Ho * Wo, Co, R * S * Ci, f"pro{mode}{'+res' if residual is not None else ''}"
                                               f"{'+mask' if mask_src is not None else ''}{'+up' if up > 1 else ''}"))
    return (out, stats) if want_stats else out


_skws = {}


def _splitk_ws(dev):
    """Scratch for the split-K path of small problems (consumed by the second-stage kernel on the same
    stream right after it is written)."""
    w = _skws.get(dev.index)
    if w is None:
        # 256 MiB: the transformed weights of the largest Winograd layer (1024 -> 1024 channels: 64 MiB) plus the split-K
        # slab of the same launch (with 64 MiB the 4x4 / 1024-channel layers of D-64 could not split: 330 vs 177 us)
        w = torch.empty(64 << 20, dtype=torch.float32, device=dev)
        _skws[dev.index] = w
    return w


def res_up_fused(geom, B, Hi, Wi, want_stats=False, group_imgs=0):
    """Will conv_fwd(geom, x[B,Hi,Wi,Ci], ..., res_up=True) run?  The bilinear x2 of a half-resolution residual is blended in
    by the Winograd kernel's epilogue (and its split-K second stage) only: True iff the automatic choice for this launch
    is that kernel (tile_cfg 9); otherwise the caller up-samples the residual itself (diagan_upsample2x).
    group_imgs: images per prologue group of the launch (0: ungrouped) -- part of the choice, see
    diagan_conv_gemm_pick_cfg_grouped."""
    Ho, Wo = geom.out_hw(Hi, Wi)
    sy, dr, off, up = geom.fwd_params()
    ws = _splitk_ws(torch.device('cuda', torch.cuda.current_device()))
    return nat.fn("diagan_conv_gemm_pick_cfg_grouped")(B, Hi, Wi, geom.Ci, Ho, Wo, geom.Co, geom.R, geom.S, sy, dr, off,
                                                       up, geom.Kp, 0 if want_stats else 1, ws.numel(),
                                                       group_imgs * Ho * Wo) in (9, 13)


def upin_fused(geom, B, Hl, Wl, group_imgs=0):
    """Will conv_fwd(geom, x[B,Hl,Wl,Ci], ..., up_in=True) run?  True iff conv3x3(bilinear_x2(pro(x))) of this layer qualifies
    for the one-launch F(4x4) kernel on the half-resolution input (tile_cfg 15); otherwise the caller up-samples first
    (diagan_upsample2x).  group_imgs: images per prologue group (stacked generator forward), 0: ungrouped."""
    Hi, Wi = 2 * Hl, 2 * Wl
    Ho, Wo = geom.out_hw(Hi, Wi)
    sy, dr, off, up = geom.fwd_params()
    ws = _splitk_ws(torch.device('cuda', torch.cuda.current_device()))
    return bool(nat.fn("diagan_conv_wino4_upin_supported")(B, Hi, Wi, geom.Ci, Ho, Wo, geom.Co, geom.R, geom.S, sy, dr, off, up,
                                                           ws.numel(), group_imgs * Ho * Wo))


def pool_fused(geom, B, Hi, Wi, pro=None):
    """Will conv_fwd(geom, x[B,Hi,Wi,Ci], ..., pool=True) run?  True iff avg_pool2d(conv(pro(x)), 2) of this layer qualifies
    for the one-launch Winograd + pooling kernel (tile_cfg 11: 9 of the 16 transform-domain products) and the launch is
    large enough; otherwise the caller pools the convolution's output itself (diagan_avgpool2)."""
    Ho, Wo = geom.out_hw(Hi, Wi)
    sy, dr, off, up = geom.fwd_params()
    mode = pro[0] if pro is not None else PRO_NONE
    ws = _splitk_ws(torch.device('cuda', torch.cuda.current_device()))
    return bool(nat.fn("diagan_conv_wino_pool_supported")(B, Hi, Wi, geom.Ci, Ho, Wo, geom.Co, geom.R, geom.S, sy, dr, off, up,
                                                          mode, ws.numel()))


def conv_fwd(geom, x, wf, bias=None, residual=None, pro=None, out=None, tile_cfg=0, res_relu=False, row_scale=None,
             want_stats=False, out_scale=1.0, wino=True, res_up=False, pool=False, up_in=False, wsite=None, wversion=None,
             out_map=None):
    """y = conv(pro(x)) + bias + residual.   x [B,Hi,Wi,Ci] -> y [B,Ho,Wo,Co].
    res_up: `residual` is [B,Ho/2,Wo/2,Co] and its bilinear x2 up-sampling is added (see res_up_fused).
    pool: y = avg_pool2d(conv(pro(x)) + bias, 2) + residual, y and residual [B,Ho/2,Wo/2,Co] (see pool_fused).
    up_in: y = conv(bilinear_x2(pro(x))) + ..., x [B,Hi,Wi,Ci] -> y [B,2Hi,2Wi,Co] (see upin_fused).
    out_map = (mul, offy, offx, y0, y1, x0, x1): `out` [B,OH,OW,Co] must be given; pixel (oy, ox) of the convolution's own output
    grid is written to (mul*(oy-y0)+offy, mul*(ox-x0)+offx) of it, pixels outside [y0,y1) x [x0,x1) are dropped (tile_cfg 17 only:
    ask out_map_ok first)."""
    B, Hi, Wi, Ci = x.shape
    if Ci != geom.Ci:
        raise RuntimeError(f"conv_fwd: input has {Ci} channels, layer expects {geom.Ci}")
    Ho, Wo = geom.out_hw(2 * Hi, 2 * Wi) if up_in else geom.out_hw(Hi, Wi)
    if pool:
        Ho, Wo = Ho // 2, Wo // 2
    if out_map is not None:
        if out is None:
            raise RuntimeError("conv_fwd: an output map needs the target tensor")
        out_map = tuple(out_map) + (out.shape[1], out.shape[2], Ho, Wo)
    if out is None:
        out = torch.empty((B, Ho, Wo, geom.Co), dtype=torch.float32, device=x.device)
    return _gemm(x, wf, out, geom.fwd_params(), geom.R, geom.S, geom.Kp, bias, residual, None, 0.0, pro, out_scale,
                 tile_cfg, res_relu=res_relu, row_scale=row_scale, want_stats=want_stats, wino=wino, res_up=res_up, pool=pool,
                 up_in=up_in, wsite=wsite, wversion=wversion, out_map=out_map)


def out_map_ok(geom, B, Hi, Wi, pro=None):
    """Will conv_fwd(geom, x[B,Hi,Wi,Ci], w, out=..., out_map=...) run?  True iff the automatic choice for this plain launch is
    the split-operand kernel (tile_cfg 17), the only one that writes through an output map."""
    Ho, Wo = geom.out_hw(Hi, Wi)
    sy, dr, off, up = geom.fwd_params()
    ws = _splitk_ws(torch.device('cuda', torch.cuda.current_device()))
    mode = pro[0] if pro is not None else PRO_NONE
    return nat.fn("diagan_conv_gemm_final_cfg")(B, Hi, Wi, geom.Ci, Ho, Wo, geom.Co, geom.R, geom.S, sy, dr, off, up, geom.Kp, 1,
                                               ws.numel(), 0, mode, 1, 0) == 17


def conv_dgrad(geom, dy, wd, in_hw, residual=None, mask_src=None, mask_slope=0.0, out=None, tile_cfg=0,
               row_scale=None, wino=True, unpool=False, wsite=None, wversion=None, res_unpool=False):
    """dx = conv^T(dy) (+ residual) (* relu'(mask_src)).  dy [B,Ho,Wo,Co] -> dx [B,Hi,Wi,Ci].
    res_unpool: `residual` is [B,Hi/2,Wi/2,Ci] and avg_pool2d_backward of it (a quarter of its value at (y / 2, x / 2)) is added
    (see res_unpool_fused).
    unpool: dy is the gradient of the 2x2-average-POOLED output, [B,Ho/2,Wo/2,Co]: dx = conv^T(avg_pool2d_backward(dy))
    in one launch (see unpool_fused)."""
    B, Ho, Wo, Co = dy.shape
    if Co != geom.Co:
        raise RuntimeError(f"conv_dgrad: dy has {Co} channels, layer has {geom.Co}")
    Hi, Wi = in_hw
    if unpool and (2 * Ho, 2 * Wo) != tuple(geom.out_hw(Hi, Wi)):
        raise RuntimeError(f"conv_dgrad: pooled gradient {Ho}x{Wo} does not match the layer's output {geom.out_hw(Hi, Wi)}")
    if out is None:
        out = torch.empty((B, Hi, Wi, geom.Ci), dtype=torch.float32, device=dy.device)
    return _gemm(dy, wd, out, geom.dgrad_params(), geom.R, geom.S, geom.Kd, None, residual, mask_src, mask_slope,
                 None, 1.0, tile_cfg, row_scale=row_scale, wino=wino, unpool=unpool, wsite=wsite, wversion=wversion,
                 res_unpool=res_unpool)


def res_unpool_fused(geom, B, Hi, Wi):
    """Will conv_dgrad(geom, dy, ..., residual=r_half, res_unpool=True) run for a layer with input [B,Hi,Wi,Ci]?  The quarter of
    a half-resolution residual (the gradient arriving through an average pool) is added by the Winograd kernels' epilogues
    and their split-K second stage only: True iff the automatic choice for this data gradient is one of them (tile_cfg 9 / 13);
    otherwise the caller un-pools the residual itself (diagan_avgpool2_bwd)."""
    if Hi % 2 or Wi % 2 or _os.environ.get("DIAGAN_RES_UNPOOL", "1") == "0":
        return False
    Ho, Wo = geom.out_hw(Hi, Wi)
    sy, dr, off, up = geom.dgrad_params()
    ws = _splitk_ws(torch.device('cuda', torch.cuda.current_device()))
    return nat.fn("diagan_conv_gemm_pick_cfg_geom")(B, Ho, Wo, geom.Co, Hi, Wi, geom.Ci, geom.R, geom.S, sy, dr, off, up, geom.Kd, 1,
                                                    ws.numel()) in (9, 13)


def unpool_fused(geom, B, Hi, Wi):
    """Will conv_dgrad(geom, dy_pooled, ..., unpool=True) run for a layer with input [B,Hi,Wi,Ci]?  (tile_cfg 12: the
    data-gradient through the average pool from nine Winograd products; else the caller up-samples the gradient with
    diagan_avgpool2_bwd and takes the ordinary data-gradient.)"""
    Ho, Wo = geom.out_hw(Hi, Wi)
    sy, dr, off, up = geom.dgrad_params()
    ws = _splitk_ws(torch.device('cuda', torch.cuda.current_device()))
    # the data-gradient gathers from the layer's OUTPUT tensor: its channels are the GEMM's K, the layer's inputs its columns
    return bool(nat.fn("diagan_conv_wino_unpool_supported")(B, Ho, Wo, geom.Co, Hi, Wi, geom.Ci, geom.R, geom.S, sy, dr, off, up,
                                                            ws.numel()))


_slabs = {}
_slabs_retired = []   # outgrown scratch stays allocated: a captured hipGraph may still point at it


def _slab(dev, nfloat):
    key = (dev.index, )
    s = _slabs.get(key)
    if s is None or s.numel() < nfloat:
        if s is not None:
            _slabs_retired.append(s)
        s = torch.empty(max(nfloat, 1 << 22), dtype=torch.float32, device=dev)
        _slabs[key] = s
    return s


def conv_wgrad(geom, dy, x, grad, accumulate, pro=None, sn=None):
    """grad[Co][Kp] (+)= d(loss)/d(Wp) given dy and the layer input x (prologue recomputed).

    sn = (W_master, u, v, state): backward through W/sigma (diagan_sn_grad_fix)."""
    B, Ho, Wo, Co = dy.shape
    _, Hi, Wi, Ci = wg_x_shape(x, pro)
    mode, scale, shift = _pro3(pro)
    for t, n in ((dy, 'dy'), (x, 'x'), (grad, 'grad'), (scale, 'pro_scale'), (shift, 'pro_shift')):
        _chk(t, n)
    M = B * Ho * Wo
    splits = wgrad_splits_geom(geom, B, Hi, Wi, Ho, Wo)
    n_elem = Co * geom.Kp
    extra = n_elem if sn is not None else 0
    slab = _slab(dy.device, splits * n_elem + extra)
    sy, dr, off, up = geom.fwd_params()
    st = nat.current_stream()
    timed = TIMER is not None and TIMER.wants_any()
    kn = _wgrad_kernel_name(Co, geom.Kp, mode, Ho, Wo, wino=wgrad_uses_wino(geom, Hi, Wi, Ho, Wo),
                            x3=wgrad_uses_x3(geom, B, Hi, Wi, Ho, Wo, mode, -1)) if timed else None
    t0 = TIMER.begin(kn) if timed else None
    nat.call("diagan_conv_wgrad", nat.ptr(dy), nat.ptr(x), nat.ptr(slab), splits, 1, n_elem, -1, nat.ptr(scale), nat.ptr(shift),
             mode, B, Hi, Wi, Ci, Ho, Wo, Co, geom.R, geom.S, sy, dr, off, up, geom.Kp, st)
    if t0 is not None:
        TIMER.end(kn, 2.0 * M * Co * geom.R * geom.S * Ci, t0,
                  (M, Co, geom.R * geom.S * Ci, f"pro{mode} x{splits}"))
    if sn is None:
        nat.call("diagan_wgrad_reduce", nat.ptr(slab), splits, n_elem, nat.ptr(grad), 1 if accumulate else 0,
                 None, None, st)
    else:
        W, u, v, state = sn
        G = slab[splits * n_elem: splits * n_elem + n_elem]
        nparts = (n_elem // 4 + 255) // 256
        parts = torch.empty(nparts, dtype=torch.float64, device=dy.device)
        nat.call("diagan_wgrad_reduce", nat.ptr(slab), splits, n_elem, G.data_ptr(), 0, nat.ptr(W),
                 nat.ptr(parts), st)
        nat.call("diagan_sn_grad_fix", G.data_ptr(), nat.ptr(parts), nparts, nat.ptr(u), nat.ptr(v),
                 nat.ptr(state), nat.ptr(grad), Co, geom.Kp, 1 if accumulate else 0, st)
    return grad


def wgrad_splits(M, Co, Kp):
    return nat.fn("diagan_conv_wgrad_splits")(M, Co, Kp)


def wgrad_splits_geom(geom, B, Hi, Wi, Ho, Wo):
    """split count for this layer geometry (the Winograd weight gradient has its own policy)"""
    sy, dr, off, up = geom.fwd_params()
    return nat.fn("diagan_conv_wgrad_splits_geom")(B, Hi, Wi, geom.Ci, Ho, Wo, geom.Co, geom.R, geom.S, sy, dr, off, up,
                                                   geom.Kp)


def wgrad_uses_wino(geom, Hi, Wi, Ho, Wo):
    sy, dr, off, up = geom.fwd_params()
    return bool(nat.fn("diagan_conv_wgrad_uses_wino")(Hi, Wi, geom.Ci, Ho, Wo, geom.Co, geom.R, geom.S, sy, dr, off, up,
                                                      geom.Kp))


def small_co_wgrad(geom):
    """True if the layer's weight gradient has the dedicated 4-output-channel kernel."""
    sy, dr, off, up = geom.fwd_params()
    return bool(nat.fn("diagan_conv3x3_co4_wgrad_supported")(geom.Ci, geom.Co, geom.R, geom.S, sy, dr, off, up)
                and geom.Kp == 9 * geom.Ci)


def small_co_wgrad_splits(B, H):
    return nat.fn("diagan_conv3x3_co4_wgrad_splits")(B, H)


def conv_wgrad_into(geom, dy, x, slab, splits, stride, bias_off, pro=None, segments=1, pooled=False):
    """Split-K weight (+bias) gradient partials into a caller-owned slab [splits][stride]; the sum over
    splits is done later for all layers at once (diagan_wgrad_finish_batched).
    pooled: this launch is ConvLayer.wgrad_pooled's strided form of a stride-1 layer's weight gradient (the kernel timer
    then books the reference convolution's FLOP, 4x what the launch executes, under a marked kernel name)."""
    B, Ho, Wo, Co = dy.shape
    _, Hi, Wi, Ci = wg_x_shape(x, pro)
    mode, scale, shift = _pro3(pro)
    sy, dr, off, up = geom.fwd_params()
    if segments == 1 and small_co_wgrad(geom):
        if splits != small_co_wgrad_splits(B, Ho):
            raise RuntimeError(f"conv_wgrad_into: slab has {splits} splits, the 4-channel kernel writes "
                               f"{small_co_wgrad_splits(B, Ho)}")
        t0 = TIMER.begin("conv3x3_co4_wgrad_kernel") if TIMER is not None else None
        nat.call("diagan_conv3x3_co4_wgrad", nat.ptr(dy), nat.ptr(x), nat.ptr(slab), stride, bias_off, nat.ptr(scale),
                 nat.ptr(shift), mode, B, Hi, Wi, Ci, geom.Kp, nat.current_stream())
        if t0 is not None:
            TIMER.end("conv3x3_co4_wgrad_kernel", 2.0 * B * Ho * Wo * Co * 9 * Ci, t0, (B * Ho * Wo, Co, 9 * Ci, f"pro{mode}"))
        return
    timed = TIMER is not None and TIMER.wants_any()
    kn = _wgrad_kernel_name(Co, geom.Kp, mode, Ho, Wo, wino=wgrad_uses_wino(geom, Hi, Wi, Ho, Wo),
                            x3=wgrad_uses_x3(geom, B, Hi, Wi, Ho, Wo, mode, bias_off)) if timed else None
    if kn is not None and pooled:
        kn += POOLED_TAG
    t0 = TIMER.begin(kn) if timed else None
    nat.call("diagan_conv_wgrad", nat.ptr(dy), nat.ptr(x), nat.ptr(slab), splits, segments, stride, bias_off, nat.ptr(scale),
             nat.ptr(shift), mode, B, Hi, Wi, Ci, Ho, Wo, Co, geom.R, geom.S, sy, dr, off, up, geom.Kp,
             nat.current_stream())
    if t0 is not None:
        px = B * Ho * Wo * (4 if pooled else 1)
        TIMER.end(kn, 2.0 * px * Co * geom.R * geom.S * Ci, t0, (px, Co, geom.R * geom.S * Ci, f"pro{mode}"))


# ---- the Winograd weight gradients of several layers in ONE launch (include/diagan_hip.h: diagan_conv_wgrad_batched, round 5) ----
# A layer's own launch fills the chip by cutting its pixel range 256 / tiles ways: for the small maps that is 4-16 K-steps per
# workgroup behind ~15 us of fixed cost, and 38 MB of slab per layer whatever its size.  Deferred to the end of the backward
# pass the layers of one prologue mode share the chip's workgroups in proportion to their work (`batched_wgrad_splits`).
# DIAGAN_WGRAD_BATCH=0: every layer launches on its own, as before.
WGRAD_BATCH = _os.environ.get("DIAGAN_WGRAD_BATCH", "1") != "0"
# a layer whose OWN launch would run at least this many K-steps per workgroup launches on its own; default: never -- sweep on
# MI355X (tools/probe/wgrad_minsteps.sh; SNGAN-32 / SNGAN-64 images/s): 12 -> 4994 / 2876, 24 -> 5053 / 2916, 40 -> 5061 / 2920,
# everything batched 5074 / 2949: the long layers' batched launches are a little slower than their own, the slabs and launches
# saved are worth more
WGRAD_BATCH_MIN_STEPS = int(_os.environ.get("DIAGAN_WGRAD_BATCH_MIN_STEPS", str(1 << 30)))
_WG_JOB = None


def batched_wgrad_splits(jobs, slots=256, fixed=8.0):
    """jobs: [(tiles, steps, segments)] per layer (64x64 output tiles, K-steps of 8 Winograd tiles over the whole pixel range,
    segments the splits may not straddle) -> split count per layer.  Model: the launch runs ceil(workgroups / slots) rounds
    of (longest K loop of a workgroup + `fixed` K-steps of prologue / epilogue / slab write-out, ~15 us at 1.8 us per step);
    every layer is cut so that no workgroup runs more than `per` steps (at least 4 steps per split, one split per segment,
    workgroup ranges padded to multiples of 8), and `per` is the candidate with the cheapest launch."""
    def plan(per):
        out, total, longest = [], 0, 0
        for tiles, steps, seg in jobs:
            seg_steps = -(-steps // seg)
            k = max(1, min(-(-seg_steps // per), max(1, seg_steps // 4)))
            out.append(k * seg)
            total += (tiles * k * seg + 7) // 8 * 8
            longest = max(longest, -(-seg_steps // k))
        return (-(-total // slots)) * (longest + fixed), out
    best, per = None, 4.0
    top = max(-(-steps // seg) for _, steps, seg in jobs)
    while True:
        cost, out = plan(int(per))
        if best is None or cost < best[0]:
            best = (cost, out)
        if per > top:
            return best[1]
        per = max(per + 1, per * 1.08)


_WG_TABS = {}


def conv_wgrad_batched(jobs, key=None, kernel_name=None, flop_scale=1.0, cache=None):
    """jobs: [(geom, dy, x, slab, splits, stride, bias_off, pro, segments)] -- the arguments of conv_wgrad_into, every job a
    layer of the Winograd weight gradient, all with the same prologue mode; at most wgrad_batch_max() of them.
    key: a hashable that identifies everything but the tensors' addresses (layers, shapes, splits): the job table's constant
    columns are then built once and only the five pointers per job are refreshed (host time matters on launch-bound nets)."""
    global _WG_JOB
    import numpy as np
    if _WG_JOB is None:
        _WG_JOB = np.dtype([('p', np.uint64, 5), ('l', np.int64, 2), ('i', np.int32, 18)])
        assert _WG_JOB.itemsize == 128
    # `cache`: the CALLER's table cache (WgradBatch keeps one per network: it dies with the net, so the layer ids in `key` cannot be
    # recycled under it -- ADVICE r5).  Without one the module-level cache is used and everything the cached constant columns hold
    # becomes part of the key (a new layer with an old id and another geometry then builds its own table).
    tabs = cache if cache is not None else _WG_TABS
    if key is not None and cache is None:
        key = (key,) + tuple((g.kind, g.Ci, g.Co, g.R, g.S, g.stride, g.pad, g.Kp, tuple(dy.shape), tuple(x.shape), sp, st, bo,
                              _pro3(pro)[0], seg) for g, dy, x, _, sp, st, bo, pro, seg in jobs)
    cached = tabs.get(key) if key is not None else None
    if cached is None:
        tab = np.zeros(len(jobs), dtype=_WG_JOB)
        flop, px, mode0 = 0.0, 0, None
        for j, (geom, dy, x, slab, splits, stride, bias_off, pro, segments) in enumerate(jobs):
            B, Ho, Wo, Co = dy.shape
            _, Hi, Wi, Ci = wg_x_shape(x, pro)
            mode = _pro3(pro)[0]
            mode0 = mode if mode0 is None else mode0
            sy, dr, off, up = geom.fwd_params()
            tab[j]['l'] = [stride, bias_off]
            tab[j]['i'] = [splits, segments, mode, B, Hi, Wi, Ci, Ho, Wo, Co, geom.R, geom.S, sy, dr, off, up, geom.Kp, 0]
            flop += 2.0 * B * Ho * Wo * Co * geom.R * geom.S * Ci
            px += B * Ho * Wo
        cached = (tab, flop, px, mode0)
        if key is not None:
            if len(tabs) > 256:
                tabs.clear()
            tabs[key] = cached
    tab, flop, px, mode0 = cached
    ptrs = []
    for geom, dy, x, slab, splits, stride, bias_off, pro, segments in jobs:
        scale, shift = (pro[1], pro[2]) if pro is not None and len(pro) > 2 else (None, None)
        ptrs.append((dy.data_ptr(), x.data_ptr(), slab.data_ptr(), 0 if scale is None else scale.data_ptr(),
                     0 if shift is None else shift.data_ptr()))
    tab['p'] = ptrs
    timed = TIMER is not None and TIMER.wants_any()
    kn = (kernel_name or f"conv_wgrad_wino_batched_kernel<{mode0}>") if timed else None
    t0 = TIMER.begin(kn) if timed else None
    nat.call("diagan_conv_wgrad_batched", tab.ctypes.data, len(jobs), nat.current_stream())
    if t0 is not None:
        g0 = jobs[0][0]
        TIMER.end(kn, flop * flop_scale, t0, (int(px * flop_scale), g0.Co, g0.R * g0.S * g0.Ci, f"pro{mode0} {len(jobs)} layers"))


def wgrad_batch_max():
    return nat.fn("diagan_conv_wgrad_batch_max")()


def wgrad_batch_class(geom, Hi, Wi, Ho, Wo, mode):
    """identity of the kernel template a layer's weight gradient runs on (0: not batchable); equal classes share a launch"""
    sy, dr, off, up = geom.fwd_params()
    return nat.fn("diagan_conv_wgrad_batch_class")(Hi, Wi, geom.Ci, Ho, Wo, geom.Co, geom.R, geom.S, sy, dr, off, up, geom.Kp, mode)


def wgrad_batch_shape(geom, B, Ho, Wo, cls):
    """(output tiles, K-steps over the whole pixel range, workgroup slots of the chip, fixed cost in K-steps) of a layer in a
    batched launch of class `cls` -- the inputs of batched_wgrad_splits"""
    if cls < 1000:                          # Winograd: 64 x 64 tiles, 8 tiles of 2x2 outputs per K-step, one workgroup per CU
        return ((geom.Co + 63) // 64) * ((geom.Ci + 63) // 64), (B * (Ho // 2) * (Wo // 2) + 7) // 8, 256, 8.0
    bn, bk = (64 if geom.Co <= 64 else 128), (64 if geom.Kp <= 64 else 128)
    if bn == 128 and bk == 64:
        bn = 64
    slots = _WG_SLOTS64 if (bn == 64 and bk == 64) else 512
    return ((geom.Co + bn - 1) // bn) * ((geom.Kp + bk - 1) // bk), (B * Ho * Wo + 31) // 32, slots, 6.0


# workgroup slots the split policy assumes for the 64 x 64 weight-gradient tile.  The chip holds four of them per CU (84
# registers, 32 KB of LDS), but planning for 1024 / 1280 is not faster: SNGAN-32 5277-5283 / 5277-5280 / 5259-5266 images/s,
# SNGAN-64 3033-3048 / 3030-3035 / 3010-3013 for 512 / 1024 / 1280 on one box
_WG_SLOTS64 = int(_os.environ.get("DIAGAN_WGRAD_SLOTS64", "512"))
POOLED_TAG = " [pooled gradient]"     # kernel-timer name suffix of ConvLayer.wgrad_pooled's launches


def _wgrad_kernel_name(Co, Kp, mode=0, Ho=0, Wo=0, wino=False, x3=False):
    """Kernel name as rocprofv3 prints it (template arguments BNn, BNk, PRO, P2)."""
    if wino:
        return f"conv_wgrad_wino_kernel<{mode}>"
    if x3:
        return "conv_wgrad_x3_kernel"
    bn, bk = (64 if Co <= 64 else 128), (64 if Kp <= 64 else 128)
    if bn == 128 and bk == 64:
        bn = 64
    if bk != 128:
        return f"conv_wgrad_kernel<{bn},{bk},-1,false>"
    p2 = Ho > 0 and Wo > 0 and (Ho & (Ho - 1)) == 0 and (Wo & (Wo - 1)) == 0
    return f"conv_wgrad_kernel<{bn},128,{mode},{'true' if p2 else 'false'}>"


def sn_power_iter(W, u_buffer, sigma_buffer, training=True, eps=1e-12):
    """One torch_mimicry SpectralNorm step on packed W[Co][Kp]; returns (u, v, state[sigma, 1/sigma])."""
    Co, Kp = W.shape
    dev = W.device
    u = torch.empty(Co, dtype=torch.float32, device=dev)
    v = torch.empty(Kp, dtype=torch.float32, device=dev)
    state = torch.empty(2, dtype=torch.float32, device=dev)
    work = torch.empty(Kp + Co, dtype=torch.float32, device=dev)
    nat.call("diagan_sn_power_iter", nat.ptr(W), nat.ptr(u_buffer), nat.ptr(sigma_buffer), nat.ptr(u), nat.ptr(v),
             nat.ptr(state), nat.ptr(work), Co, Kp, eps, 1 if training else 0, nat.current_stream())
    return u, v, state


def pack_weights(W, Co, Ci, RS, Kp, Kd, inv_sigma=None, Wf=None, Wd=None):
    nat.call("diagan_pack_weights", nat.ptr(W), nat.ptr(inv_sigma), nat.ptr(Wf), nat.ptr(Wd), Co, Ci, RS, Kp, Kd,
             nat.current_stream())


# ---- layout helpers (host/torch side, run at init / checkpoint time only) --------------------

def pack_oihw(w_oihw, Kp, ci_pad=None):
    """torch Conv2d weight [Co,Ci,R,S] -> packed [Co][Kp] (k = (r*S+s)*Ci' + c), zero padded."""
    Co, Ci, R, S = w_oihw.shape
    Cp = ci_pad or Ci
    w = w_oihw.permute(0, 2, 3, 1)                      # [Co,R,S,Ci]
    if Cp != Ci:
        w = torch.nn.functional.pad(w, (0, Cp - Ci))
    w = w.reshape(Co, R * S * Cp)
    out = torch.zeros((Co, Kp), dtype=w.dtype, device=w.device)
    out[:, : R * S * Cp] = w
    return out


def unpack_oihw(wp, Co, Ci, R, S, ci_pad=None):
    Cp = ci_pad or Ci
    w = wp[:, : R * S * Cp].reshape(Co, R, S, Cp)[..., :Ci]
    return w.permute(0, 3, 1, 2).contiguous()


def pack_iohw(w_iohw, Kp):
    """torch ConvTranspose2d weight [Ci,Co,R,S] -> packed forward operand [Co][(r,s,ci)]."""
    return pack_oihw(w_iohw.permute(1, 0, 2, 3), Kp)


def unpack_iohw(wp, Ci, Co, R, S):
    return unpack_oihw(wp, Co, Ci, R, S).permute(1, 0, 2, 3).contiguous()
