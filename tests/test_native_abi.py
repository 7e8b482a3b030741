"""CPU: libdiagan_hip.so loads and exports every symbol include/diagan_hip.h declares.
No compute calls (no GPU here)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def _declared():
    txt = open(os.path.join(ROOT, "include", "diagan_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(diagan_[a-z0-9_]+)\s*\(", txt)))


def test_header_declares_something():
    names = _declared()
    assert "diagan_ldr_scores_f64" in names and "diagan_last_error" in names


def test_library_loads_and_exports_all_symbols():
    from diagan import _native as nat
    assert os.path.exists(nat.LIB_PATH), "run __graft_entry__.build() first"
    L = ctypes.CDLL(nat.LIB_PATH)
    missing = [n for n in _declared() if not hasattr(L, n)]
    assert not missing, missing
    assert nat.lib().diagan_target_arch() == b"gfx950"
    assert nat.fn("diagan_abi_version")() >= 1


def test_binding_table_matches_header():
    from diagan import _native as nat
    import diagan.ops  # noqa: F401  (registers the op signatures)
    import diagan.trainer.compute_pr  # noqa: F401  (registers the precision/recall entry points)
    from diagan.trainer import distributed as D
    D._register_native()                 # the RCCL context entry points (csrc/comm.hip)
    declared = set(_declared()) - {"diagan_last_error", "diagan_target_arch"}
    assert declared == set(nat._SIGS), declared ^ set(nat._SIGS)


def test_missing_library_fails_loudly(monkeypatch):
    from diagan import _native as nat
    monkeypatch.setattr(nat, "_lib", None)
    monkeypatch.setattr(nat, "LIB_PATH", "/nonexistent/libdiagan_hip.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        nat.lib()


def test_launch_selection_queries_are_host_logic():
    """Which kernel a convolution launch takes is decided on the host (no device call): the Winograd kernel for a
    qualifying 3x3 layer with enough workgroups, the nine-product convolution + average-pool launch and its data gradient
    for the down-sampling DBlocks' c2 (mimicry DBlock, predefined_models.py:38-40,76-78) -- checked here without a GPU."""
    from diagan import _native as nat
    import diagan.ops  # noqa: F401
    pick = nat.fn("diagan_conv_gemm_pick_cfg_geom")
    pool, unpool = nat.fn("diagan_conv_wino_pool_supported"), nat.fn("diagan_conv_wino_unpool_supported")
    ws = 64 << 20
    assert pick(128, 32, 32, 128, 32, 32, 128, 3, 3, 1, 1, -1, 1, 1152, 1, ws) == 13     # D-32 block1.c2, pair pass: 512 workgroups of F(4x4,3x3)
    nat.call("diagan_conv_gemm_set_wino4", 0)
    try:
        assert pick(128, 32, 32, 128, 32, 32, 128, 3, 3, 1, 1, -1, 1, 1152, 1, ws) == 9  # ... F(2x2,3x3) when that kernel is off
    finally:
        nat.call("diagan_conv_gemm_set_wino4", -1)
    assert pick(64, 16, 16, 256, 16, 16, 256, 3, 3, 1, 1, -1, 1, 2304, 1, ws) == 13      # 128 workgroups of F(4x4): split two ways
    assert pick(64, 16, 16, 256, 16, 16, 256, 3, 3, 1, 1, -1, 1, 2304, 0, ws) == 9       # ... not where the epilogue must write statistics
    assert pick(64, 16, 16, 128, 16, 16, 128, 3, 3, 1, 1, -1, 1, 1152, 1, ws) == 9       # 64 workgroups: F(2x2) with its own split
    assert pick(384, 16, 16, 128, 16, 16, 128, 3, 3, 1, 1, -1, 1, 1152, 1, ws) == 9      # 1.5 rounds against F(2x2)'s exact 3
    assert pick(64, 6, 6, 256, 6, 6, 256, 3, 3, 1, 1, -1, 1, 2304, 1, ws) not in (13,)   # H, W not multiples of 4
    assert pick(128, 8, 8, 128, 8, 8, 128, 3, 3, 1, 1, -1, 1, 1152, 1, ws) != 9          # 8x8 / 128 channels: implicit GEMM
    assert pick(64, 16, 16, 128, 8, 8, 128, 3, 3, 2, 1, -1, 1, 1152, 1, ws) != 9         # stride 2
    assert pool(128, 32, 32, 128, 32, 32, 128, 3, 3, 1, 1, -1, 1, 1, ws) == 1            # ... + average pool: tile_cfg 11
    assert pool(128, 64, 64, 64, 64, 64, 64, 3, 3, 1, 1, -1, 1, 1, ws) == 1              # D-64 block1.c2: 128-tile workgroups
    assert pool(128, 32, 32, 128, 32, 32, 128, 3, 3, 1, 1, -1, 1, 2, ws) == 0            # BatchNorm prologue: not this kernel
    assert pool(8, 32, 32, 128, 32, 32, 128, 3, 3, 1, 1, -1, 1, 1, ws) == 0              # batch 8: too few workgroups
    assert unpool(128, 32, 32, 128, 32, 32, 128, 3, 3, 1, -1, 1, 1, ws) == 1             # its data gradient: tile_cfg 12
    assert unpool(128, 32, 32, 128, 32, 32, 128, 3, 3, 1, 1, -1, 1, ws) == 0             # (a forward geometry)
    assert nat.fn("diagan_conv_gemm_tile_rows")(11) == 256 and nat.fn("diagan_conv_gemm_tile_cols")(12) == 128


def test_grouped_prologue_never_gets_a_tile_that_straddles_two_groups():
    """The stacked generator forward (BaseGenerator.prefetch_fakes) gives the conv kernels one BatchNorm row per group of
    batch*Ho*Wo GEMM rows.  A batch size that is not a multiple of 4 makes a group at 8x8 a non-multiple of the Winograd
    kernel's 256-row tile (--batch_size 50, n_dis 5: 6 groups of 3200 rows): the automatic choice must then be a kernel
    whose tile divides the group -- what diagan_conv_gemm itself launches -- instead of an error at launch time."""
    from diagan import _native as nat
    import diagan.ops  # noqa: F401
    pick, grouped = nat.fn("diagan_conv_gemm_pick_cfg_geom"), nat.fn("diagan_conv_gemm_pick_cfg_grouped")
    rows = nat.fn("diagan_conv_gemm_tile_rows")
    ws = 64 << 20
    for batch in (64, 50, 30, 25, 7):
        for (H, C) in ((8, 256), (16, 256), (32, 256), (8, 1024), (64, 64)):
            B = 6 * batch
            geo = (B, H, H, C, H, H, C, 3, 3, 1, 1, -1, 1, 9 * C)
            group = batch * H * H
            for allow in (0, 1):
                cfg = grouped(*geo, allow, ws, group)
                assert rows(cfg) > 0 and group % rows(cfg) == 0, (batch, H, C, cfg)
                if group % rows(pick(*geo, allow, ws)) == 0:
                    assert cfg == pick(*geo, allow, ws)              # untouched where the first choice already fits
                assert grouped(*geo, allow, ws, 0) == pick(*geo, allow, ws)
    # the ADVICE r2 example: batch 50 at 8x8, six stacked batches -> Winograd would be picked, 3200 % 256 != 0
    geo = (300, 8, 8, 256, 8, 8, 256, 3, 3, 1, 1, -1, 1, 2304)
    assert pick(*geo, 1, ws) in (9, 13) and grouped(*geo, 1, ws, 3200) not in (9, 13)
    # a group that the F(4x4) kernel's 512-row tile does not divide but the F(2x2) kernel's 256 does: batch 12 at 8x8 = 768 rows
    geo = (24 * 12, 8, 8, 256, 8, 8, 256, 3, 3, 1, 1, -1, 1, 2304)
    assert pick(*geo, 1, ws) == 13 and grouped(*geo, 1, ws, 768) == 9


def test_one_hip_runtime_whatever_the_import_order():
    """build() (which loads the library) followed by torch's device initialisation in ONE process used to map two HIP runtimes --
    /opt/rocm's through the library's RUNPATH and the torch wheel's own copy -- and the library's then saw no device on a GPU
    box.  The loader imports torch first: exactly one libamdhip64 may be mapped (torch's own where the wheel ships one)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (f"import sys; sys.path.insert(0, {os.path.join(root, 'self-diagnosing-gan_amd')!r})\n"
            "from diagan import _native as nat\n"
            "assert 'torch' not in sys.modules\n"
            "nat.lib()\n"
            "import torch\n"
            "libs = sorted({l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l})\n"
            "assert len(libs) == 1, libs\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-800:]


@pytest.mark.gpu
def test_selection_options_travel_with_the_call_not_through_globals():
    """"no global mutable state".  The kernel-selection switches are per CALL (diagan_conv_opts handed over through
    diagan_conv_gemm_next_opts, thread-local, consumed by one call): the same launch with two option words back to back takes two
    different kernels, the process-wide diagnostic setters are never touched, the option does not leak into the third call, and
    two THREADS with different options do not see each other's."""
    import threading
    import torch
    from diagan import _native as nat
    from diagan.ops import conv as C
    geom = C.Geom("conv", 128, 128, 3, 3, 1, 1)
    x = torch.randn(64, 32, 32, 128, device="cuda")
    wp = torch.randn(128, geom.Kp, device="cuda") * (9 * 128) ** -0.5
    before = (nat.fn("diagan_conv_gemm_get_wino")(), nat.fn("diagan_conv_gemm_get_wino4x")(), nat.fn("diagan_conv_gemm_get_x3")(),
              nat.fn("diagan_conv_gemm_get_x3b")())
    C.next_opts(wino4=0)
    y9 = C.conv_fwd(geom, x, wp)
    cfg9 = C.last_cfg()
    C.next_opts(wino=0)
    y1 = C.conv_fwd(geom, x, wp)
    cfg1 = C.last_cfg()
    y13 = C.conv_fwd(geom, x, wp)
    cfg13 = C.last_cfg()
    assert (cfg9, cfg13) == (9, 13) and cfg1 not in (9, 13), (cfg9, cfg1, cfg13)
    ref = y1.double()
    for y in (y9, y13):
        assert float((y.double() - ref).abs().max() / ref.abs().max()) < 1e-4
    after = (nat.fn("diagan_conv_gemm_get_wino")(), nat.fn("diagan_conv_gemm_get_wino4x")(), nat.fn("diagan_conv_gemm_get_x3")(),
             nat.fn("diagan_conv_gemm_get_x3b")())
    assert before == after
    # a caller-owned ticket buffer for the in-kernel split-K combine, with the call (the library allocates nothing)
    g2 = C.Geom("conv", 256, 256, 3, 3, 1, 1)
    x2 = torch.randn(16, 8, 8, 256, device="cuda")
    w2 = torch.randn(256, g2.Kp, device="cuda") * (9 * 256) ** -0.5
    tickets = torch.zeros(4096, dtype=torch.int32, device="cuda")
    ya = C.conv_fwd(g2, x2, w2, tile_cfg=9)
    C.next_opts(splitk_fused=1, tickets=tickets)
    yb = C.conv_fwd(g2, x2, w2, tile_cfg=9)
    assert float((ya - yb).abs().max()) < 1e-4 * float(ya.abs().max()) and int(tickets.abs().sum()) == 0
    # two threads, two option words
    got = {}

    def worker(name, **opts):
        torch.cuda.set_device(0)
        for _ in range(20):
            C.next_opts(**opts)
            C.conv_fwd(geom, x, wp)
            got.setdefault(name, set()).add(C.last_cfg())
    ts = [threading.Thread(target=worker, args=("a",), kwargs=dict(wino4=0)), threading.Thread(target=worker, args=("b",), kwargs=dict(wino=0))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    torch.cuda.synchronize()
    assert got["a"] == {9} and len(got["b"]) == 1 and not (got["b"] & {9, 13}), got
