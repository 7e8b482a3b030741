"""Python-side op wrappers over the C ABI (one module per kernel family)."""
from diagan.ops import conv, diffconv, eltwise  # noqa: F401  (register the entry-point signatures)
from diagan.models import op as _stylegan_ops  # noqa: F401
