"""Python-side op wrappers over the C ABI (one module per kernel family)."""
