"""Python-side op wrappers over the C ABI (one module per kernel family)."""
from diagan.ops import conv  # noqa: F401  (registers the entry-point signatures)
