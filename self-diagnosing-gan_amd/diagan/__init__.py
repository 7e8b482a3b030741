"""diagan -- host-side mirror of the reference's `diagan` package (diagan-pkg/diagan) for the
Dia-GAN hot path, running on hand-written gfx950 HIP kernels behind libdiagan_hip.so."""
__version__ = "0.1.0"
