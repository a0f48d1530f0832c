"""bitsandbytes-compatible blockwise quantisation (public API of conch/ops/quantization/bitsandbytes)."""
