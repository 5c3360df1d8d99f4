from .V2XSimDet import V2XSimDet, collate_dense, collate_to_device, write_sample  # noqa: F401
from .V2XSimSeg import V2XSimSeg, seg_batch_on_device, write_seg_sample  # noqa: F401
from .prefetch import DevicePrefetcher  # noqa: F401
