"""Multi-object inference sharded over GPUs: one process per GPU, objects
(encoder weights + codebook) distributed round-robin, a mixed batch of detector
crops routed to the owners, and ONE collective -- an all_gather of the
fixed-capacity (index, score) buffers over RCCL/xGMI -- to reassemble the
answer in the original order.

The reference holds N independent AAEs in one TF session and loops over the
detections one at a time (/root/reference/auto_pose/m3_interface/ae_pose_estimator.py:61-78,
143-170); crops of different objects never interact, so the path shards with
no data-path collective other than the final gather.

The per-object compute is injected (``local_infer``) so the routing/gather
logic is testable on CPU with the gloo backend (tests/test_dist_gloo.py); the
product default is the HIP path of ``Codebook``.
"""
from __future__ import annotations

import numpy as np


def owner_of(obj_id, world_size):
    """object o lives on rank o mod G."""
    return int(obj_id) % int(world_size)


def route(class_ids, world_size, rank):
    """{obj_id: positions in the batch} for the objects this rank owns (stable order)."""
    class_ids = np.asarray(class_ids)
    out = {}
    for pos, o in enumerate(class_ids.tolist()):
        if owner_of(o, world_size) == rank:
            out.setdefault(int(o), []).append(pos)
    return {o: np.asarray(p, dtype=np.int64) for o, p in out.items()}


class ShardedPoseEngine(object):
    """local_infer(obj_id, crops_subset) -> (idx int64 [n], score float32 [n]); crops_subset is
    ``crops[positions]`` (numpy array or torch tensor, whatever the caller passed in)."""

    def __init__(self, local_infer, world_size=None, rank=None, group=None, device=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.distributed = dist.is_available() and dist.is_initialized()
        self.world_size = int(world_size) if world_size is not None else (dist.get_world_size(group) if self.distributed else 1)
        self.rank = int(rank) if rank is not None else (dist.get_rank(group) if self.distributed else 0)
        self.local_infer = local_infer
        self.device = device

    def infer(self, crops, class_ids):
        """Every rank passes the same (crops, class_ids); every rank returns the full
        (idx int64 [B], score float32 [B]) in batch order."""
        import torch
        B = len(class_ids)
        dev = self.device if self.device is not None else torch.device('cpu')
        # fixed-capacity buffers, -1 sentinel for "not mine" (equal sizes for all_gather)
        packed = torch.full((B, 2), -1, dtype=torch.int64, device=dev)
        for obj, pos in route(class_ids, self.world_size, self.rank).items():
            sel = crops[torch.as_tensor(pos, device=crops.device)] if torch.is_tensor(crops) else crops[pos]
            idx, score = self.local_infer(obj, sel)
            idx = torch.as_tensor(idx, dtype=torch.int64, device=dev).reshape(-1)
            score = torch.as_tensor(score, dtype=torch.float32, device=dev).reshape(-1)
            p = torch.as_tensor(pos, device=dev)
            packed[p, 0] = idx
            packed[p, 1] = score.view(torch.int32).to(torch.int64)
        if self.world_size > 1:
            gathered = torch.empty((self.world_size * B, 2), dtype=torch.int64, device=dev)
            self.dist.all_gather_into_tensor(gathered, packed, group=self.group)
            gathered = gathered.view(self.world_size, B, 2)
            owners = torch.as_tensor([owner_of(o, self.world_size) for o in np.asarray(class_ids).tolist()],
                                     dtype=torch.int64, device=dev)
            packed = gathered[owners, torch.arange(B, device=dev)]
        idx = packed[:, 0]
        score = packed[:, 1].to(torch.int32).view(torch.float32)
        return idx, score
