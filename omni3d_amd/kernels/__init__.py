"""Python launchers for the C-ABI kernels (one module per kernel family)."""
