"""theatergen_amd — MI355X-native implementation of TheaterGen's per-character denoising hot path.

Host side is Python on PyTorch-ROCm (device memory, streams, torch.distributed only); every device
computation goes through the hand-written HIP library ``theatergen_amd/lib/libtheatergen_hip.so``
(C ABI declared in ``include/theatergen_hip.h``).  There is NO CPU / PyTorch fallback for compute:
ops raise ``RuntimeError`` when the library is missing.
"""
__version__ = "0.1.0"
