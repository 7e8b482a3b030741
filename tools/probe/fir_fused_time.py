#!/usr/bin/env python
"""Round 6: the fused forms of fir_cl4_kernel against the plain blur and against the launches they replace, same tensors.
    python tools/probe/fir_fused_time.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "self-diagnosing-gan_amd"))
import torch  # noqa: E402


def timed(f, reps=10):
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    from diagan.models.op import fused_act as FA, fused_tail as FT
    from diagan.models.op.upfirdn2d import upfirdn2d_nhwc
    dev = torch.device("cuda", 0)
    k1 = torch.tensor([1.0, 3.0, 3.0, 1.0])
    kern = (torch.outer(k1, k1) / 64).to(dev)
    for (B, H, C) in [(32, 256, 128), (32, 128, 256), (32, 64, 512)]:
        x = torch.randn(B, H, H, C, device=dev)
        b = torch.randn(C, device=dev)
        gb = 4.0 * x.numel() / 1e9
        with torch.no_grad():
            t_fir = timed(lambda: upfirdn2d_nhwc(x, kern, pad=(2, 2)))
            t_act = timed(lambda: FA.fused_leaky_relu(x, b, 0.2, 1.4, bias_dim=-1))
            t_f1 = timed(lambda: FT.bias_act_blur(x, b, kern, (2, 2), 0.2, 1.4))
            d = torch.rand(B, C, device=dev) + 0.5
            nz = torch.randn(B, H + 1, H + 1, 1, device=dev)
            s = torch.randn(1, device=dev)
            t_tail = timed(lambda: FA.styled_bias_act(upfirdn2d_nhwc(x, kern, pad=(2, 2)), d, nz, s, b))
            t_f2 = timed(lambda: FT.blur_styled_act(x, kern, (2, 2), d, nz, s, b))
        g = torch.randn(B, H + 1, H + 1, C, device=dev)
        xr, br = x.clone().requires_grad_(True), b.clone().requires_grad_(True)
        y = FT.bias_act_blur(xr, br, kern, (2, 2), 0.2, 1.4)
        t_f3 = timed(lambda: torch.autograd.grad(y, [xr, br], g, retain_graph=True))
        FT.FUSED_GATE = False
        t_two = timed(lambda: torch.autograd.grad(y, [xr, br], g, retain_graph=True))
        FT.FUSED_GATE = True
        print(f"{B}x{H}x{H}x{C} ({gb:.2f} GB per tensor): blur {t_fir:6.1f} us ({2 * gb / t_fir * 1e3:4.1f} TB/s), bias+lrelu {t_act:6.1f}; "
              f"FUSE 1 {t_f1:6.1f} ({2 * gb / t_f1 * 1e3:4.1f} TB/s) vs {t_fir + t_act:6.1f};  blur + tail {t_tail:6.1f}, FUSE 2 {t_f2:6.1f};  "
              f"backward FUSE 3 {t_f3:6.1f} ({3 * gb / t_f3 * 1e3:4.1f} TB/s) vs adjoint + gate {t_two:6.1f}")


if __name__ == "__main__":
    main()
