"""ctypes binding of libdiagan_hip.so (the C ABI declared in include/diagan_hip.h).

This is the only way device work is issued by the package: there is NO CPU or eager-PyTorch
fallback.  If the shared library is missing, or a call returns non-zero, a RuntimeError is raised
(mirroring TORCH_CHECK -> RuntimeError of the reference's pybind ops,
diagan-pkg/diagan/models/op/fused_bias_act.cpp:7,13-14).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DIAGAN_LIB_PATH: another build of the same library (kernel A/B runs of the tuning tools); default: the in-tree build
LIB_PATH = os.environ.get("DIAGAN_LIB_PATH") or os.path.join(_HERE, "libdiagan_hip.so")

_lib = None

c_void_p = ctypes.c_void_p
c_int = ctypes.c_int
c_i64 = ctypes.c_int64
c_f32 = ctypes.c_float
c_f64 = ctypes.c_double

# name -> argtypes (restype is int unless listed in _RESTYPE)
_SIGS = {
    "diagan_abi_version": [],
    "diagan_ldr_scores_f64": [c_void_p, c_int, c_i64, c_i64, c_void_p, c_void_p, c_void_p, c_void_p,
                              c_void_p, c_int, c_void_p, c_f64, c_f64, c_void_p, c_void_p],
    "diagan_ldr_scores_f32": [c_void_p, c_int, c_i64, c_i64, c_void_p, c_void_p, c_void_p, c_void_p,
                              c_void_p, c_int, c_void_p, c_f32, c_f32, c_void_p, c_void_p],
    "diagan_logit_scatter": [c_void_p, c_void_p, c_i64, c_void_p, c_i64, c_int, c_void_p, c_void_p],
}
_RESTYPE = {
    "diagan_last_error": ctypes.c_char_p,
    "diagan_target_arch": ctypes.c_char_p,
}


def register(name, argtypes):
    """Used by the op modules to declare further entry points before first use."""
    _SIGS[name] = argtypes


def lib():
    """Load (once) and return the ctypes handle; raise loudly when the HIP extension is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"libdiagan_hip.so not found at {LIB_PATH}: build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950). "
                "There is no CPU fallback for the device path.")
        # torch FIRST: its wheel ships its own libamdhip64.so.7 / libhsa-runtime64 and dlopens them by path.  Loaded after this
        # library (whose RUNPATH finds /opt/rocm's copy of the same SONAME) the process ends up with TWO HIP runtimes, and the
        # one this library is bound to then reports "no ROCm-capable device" (seen with build() followed by smoke() in one
        # process).  With torch's runtime already mapped, the DT_NEEDED entry resolves to it and there is exactly one.
        import torch  # noqa: F401
        L = ctypes.CDLL(LIB_PATH)
        for name, rt in _RESTYPE.items():
            getattr(L, name).restype = rt
            getattr(L, name).argtypes = []
        _lib = L
    return _lib


_bound = {}


def fn(name):
    f = _bound.get(name)
    if f is None:
        L = lib()
        f = getattr(L, name)
        f.argtypes = _SIGS[name]
        f.restype = ctypes.c_int
        _bound[name] = f
    return f


def last_error():
    return lib().diagan_last_error().decode()


def call(name, *args):
    """Invoke an entry point; non-zero return -> RuntimeError with the library's message."""
    rc = fn(name)(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed ({rc}): {last_error()}")


def ptr(t):
    """Device pointer of a torch tensor (or None -> NULL)."""
    return None if t is None else t.data_ptr()


_raw_stream = None


def current_stream():
    """hipStream_t of torch's current stream on the current device.  `torch.cuda.current_stream().cuda_stream` builds a
    Stream object through four Python layers (8 us per call, ~30 % of the host time of a training step); the raw
    getter underneath is one C call."""
    global _raw_stream
    if _raw_stream is None:
        import torch
        get, cur = getattr(torch._C, "_cuda_getCurrentRawStream", None), getattr(torch._C, "_cuda_getDevice", torch.cuda.current_device)
        if get is None:
            _raw_stream = lambda: torch.cuda.current_stream().cuda_stream
        else:
            torch.cuda.init()
            _raw_stream = lambda: get(cur())
    return _raw_stream()
