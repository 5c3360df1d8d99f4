"""Training path: PyTorch-ROCm autograd over the SAME parameter tree the HIP inference
engine packs.  See graph.py (differentiable graph) and loss.py (focal + smooth-L1)."""
from .graph import train_forward  # noqa: F401
from .loss import detection_loss  # noqa: F401
