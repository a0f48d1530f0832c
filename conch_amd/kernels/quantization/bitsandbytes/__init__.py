"""Launchers of the bitsandbytes-style blockwise kernels (C-ABI seam)."""
