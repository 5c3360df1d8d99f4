"""Parsed SEGMENTATION dataset reader / writer -- mirror of upstream
coperception/datasets/V2XSimSeg.py, which is not in /root/reference.

On-disk layout as the reference documents it (/root/reference/README.md:66-79; `V2X-Sim-seg` sits next to `V2X-Sim-det`):

    V2X-Sim-seg/{train,val,test}/agent{0..5}/{scene}_{frame}/0.npy          agent0 = RSU (README.md:70)

Content of 0.npy (build-owned, like the det format -- oracle/ASSUMPTIONS.md row 7): the det sample's sweep and poses plus the
per-cell class map upstream's create_data_seg.py rasterises from the annotated boxes:

    {"voxel_indices_0": int32 (M, 3), "trans_matrices": float32 (A, 4, 4), "target_agent_id": int, "num_sensor": int,
     "bev_seg": uint8 (X, Y)   class id per BEV cell (0 = background, 1 = vehicle, ... < n_classes = 8)}

__getitem__ -> one tuple per agent: (padded_voxel_points, bev_seg, filename, target_agent_id, num_sensor, trans_matrices).
As for detection, `densify="cpu"` reproduces upstream's DataLoader-side scatter and `densify="none"` ships the sparse indices
so that `seg_batch_on_device` densifies on the GPU (v2x_indices_to_bits), the same kernel the det reader uses.
"""
import os

import numpy as np
import torch

from .V2XSimDet import V2XSimDet, write_sample


def write_seg_sample(root, split, agent, scene, frame, voxel_indices, trans_matrices, num_sensor, bev_seg):
    """One (agent, scene, frame) sample of V2X-Sim-seg.  bev_seg: (X, Y) integer class map."""
    d = write_sample(root, split, agent, scene, frame, voxel_indices, trans_matrices, num_sensor)
    path = os.path.join(d, "0.npy")
    sample = np.load(path, allow_pickle=True).item()
    seg = np.asarray(bev_seg)
    if seg.ndim != 2 or seg.min() < 0 or seg.max() > 255:
        raise ValueError("bev_seg must be an (X, Y) map of class ids in [0, 255]")
    sample["bev_seg"] = seg.astype(np.uint8)
    np.save(path, sample, allow_pickle=True)
    return d


class V2XSimSeg(V2XSimDet):
    """Same directory scan / agent alignment as V2XSimDet; samples additionally carry the class map."""

    def __getitem__(self, idx):
        name = self.seq_names[idx]
        res = []
        for root in self.dataset_roots:
            gt = np.load(os.path.join(root, name, "0.npy"), allow_pickle=True).item()
            if "bev_seg" not in gt:
                raise KeyError("%s holds no 'bev_seg': not a V2X-Sim-seg sample" % os.path.join(root, name))
            seg = np.asarray(gt["bev_seg"], dtype=np.uint8)
            if seg.shape != self.dims[:2]:
                raise ValueError("bev_seg %s does not match the BEV grid %s" % (seg.shape, self.dims[:2]))
            indices = np.asarray(gt["voxel_indices_0"], dtype=np.int32).reshape(-1, 3)
            if self.densify == "cpu":
                vox = np.zeros(self.dims, dtype=bool)
                vox[indices[:, 0], indices[:, 1], indices[:, 2]] = 1
                pvp = vox[None].astype(np.float32)
            else:
                pvp = indices
            res.append((pvp, seg, os.path.join(root, name), int(gt["target_agent_id"]), int(gt["num_sensor"]),
                        np.asarray(gt["trans_matrices"], dtype=np.float32)))
        return res


def seg_batch_on_device(samples, grid, device):
    """V2XSimSeg samples (densify='none') -> the data dict SegModule.step / predict consume: sweeps densified on the GPU,
    labels (A*B, X, Y) uint8 on the device, agent-major like the models batch."""
    from .. import ops
    B, A = len(samples), len(samples[0])
    order = [(a, b) for a in range(A) for b in range(B)]
    cap = max(1, max(samples[b][a][0].shape[0] for a, b in order))
    idx = np.zeros((len(order), cap, 3), np.int32)
    cnt = np.zeros((len(order),), np.int32)
    for m, (a, b) in enumerate(order):
        it = samples[b][a][0]
        idx[m, :it.shape[0]] = it
        cnt[m] = it.shape[0]
    bits = ops.indices_to_bits(torch.from_numpy(idx).to(device), torch.from_numpy(cnt).to(device), grid)
    return {"bev_seq": ops.bits_to_dense(bits, grid.dims[2])[:, None],
            "labels": torch.from_numpy(np.stack([samples[b][a][1] for a, b in order])).to(device),
            "trans_matrices": torch.from_numpy(np.stack([np.stack([samples[b][a][5] for a in range(A)]) for b in range(B)])).to(device),
            "num_agent": torch.tensor([[samples[b][a][4] for a in range(A)] for b in range(B)])}
