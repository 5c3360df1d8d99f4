from .V2XSimDet import V2XSimDet, collate_dense, collate_to_device, write_sample  # noqa: F401
from .prefetch import DevicePrefetcher  # noqa: F401
