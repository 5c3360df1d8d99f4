"""`from coperception.models.det import *` equivalent (README.md:101 benchmarks)."""
from .base import (ClassificationHead, DetModelBase, IntermediateModelBase, LidarDecoder,  # noqa: F401
                   LidarEncoder, NonIntermediateModelBase, SingleRegressionHead)
from .FaFNet import FaFNet  # noqa: F401
from .V2VNet import V2VNet  # noqa: F401
from .When2com import When2com  # noqa: F401
from .fusion import CatFusion, DiscoNet, MaxFusion, MeanFusion, SumFusion  # noqa: F401

__all__ = ["FaFNet", "V2VNet", "When2com", "SumFusion", "MeanFusion", "MaxFusion", "CatFusion", "DiscoNet", "LidarEncoder",
           "LidarDecoder", "DetModelBase"]
