"""czech_contriever_amd — MI355X (gfx950) native MoCo contrastive-training hot path.

Python is only the host glue: device memory, streams and torch.distributed come from PyTorch-ROCm;
all arithmetic on the path is in libczc.so (hand-written HIP, C ABI in include/czc.h)."""
from . import lib  # noqa: F401

__all__ = ["lib"]
