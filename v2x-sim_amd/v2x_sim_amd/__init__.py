"""v2x_sim_amd -- MI355X-native hot path of the V2X-Sim / coperception collaborative-perception
baselines (lowerbound/upperbound FaFNet, V2VNet, when2com/who2com; det + seg heads).

Python here is the host-side mirror of the coperception model/Dataset API
(/root/reference/README.md:101 points at coperception/tools/{det,seg}); all arithmetic runs in
the hand-written gfx950 kernels of lib/libv2x_amd.so through the C ABI of include/v2x_amd.h.
"""
from . import _lib  # noqa: F401

__version__ = "0.1.0"
