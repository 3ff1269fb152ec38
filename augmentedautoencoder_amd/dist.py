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


# ---- one huge codebook, rows sharded over the ranks ("one exchange step") ----
def row_shard_bounds(n_rows, world_size, align=1):
    """Contiguous row ranges [lo, hi) per rank, every boundary a multiple of ``align`` (= num_cyclo when the
    upright search is used, so that 'every num_cyclo-th row' means the same thing globally and locally)."""
    units = -(-int(n_rows) // int(align))
    per = -(-units // int(world_size))
    return [(min(r * per * align, n_rows), min((r + 1) * per * align, n_rows)) for r in range(world_size)]


def merge_topk(scores, indices, k):
    """scores/indices [B, M] candidate lists (global row ids; -inf/int64-max padding allowed) -> the k best per
    query in the canonical order of the single-GPU scan: descending score, lowest row first among equals."""
    import torch
    by_row = torch.argsort(indices, dim=1, stable=True)
    s = scores.gather(1, by_row)
    i = indices.gather(1, by_row)
    by_score = torch.argsort(s, dim=1, descending=True, stable=True)[:, :k]
    return i.gather(1, by_score), s.gather(1, by_score)


class RowShardedCodebook(object):
    """A codebook too large to want on one GPU: rank r keeps rows [lo_r, hi_r); every rank scans its rows for
    the same (replicated) latent batch, then ONE all_gather of the k local (score, global row) pairs per query
    and a k-way merge reproduce exactly what one scan over all rows returns (same tie rule).

    local_nn(z, k, col_stride) -> (idx int64 [B,k'], score float32 [B,k']) over the LOCAL rows (k' <= k when the
    shard holds fewer candidates); the product default is ``CodebookEngine.nn`` of the local slice."""

    def __init__(self, local_nn, n_rows, world_size=None, rank=None, group=None, device=None, align=1):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.distributed = dist.is_available() and dist.is_initialized()
        self.world_size = int(world_size) if world_size is not None else (dist.get_world_size(group) if self.distributed else 1)
        self.rank = int(rank) if rank is not None else (dist.get_rank(group) if self.distributed else 0)
        self.bounds = row_shard_bounds(n_rows, self.world_size, align)
        self.lo, self.hi = self.bounds[self.rank]
        self.local_nn = local_nn
        self.device = device

    @classmethod
    def from_array(cls, E, dtype='f32', device=None, align=1, **kw):
        """Product path: this rank's slice of the host array ``E`` [N,J] goes into a HIP ``CodebookEngine``."""
        from .engine import CodebookEngine
        self = cls(None, len(E), device=device, align=align, **kw)
        if self.hi > self.lo:
            self.engine = CodebookEngine(E[self.lo:self.hi], dtype=dtype, device=device)
            self.local_nn = lambda z, k, stride: self.engine.nn(z, min(k, -(-(self.hi - self.lo) // stride)), stride)
        return self

    def local_candidates(self, z, topk=1, col_stride=1):
        """This rank's k best (global row int64 [B,k], score float32 [B,k]); short shards pad with (int64 max, -inf)."""
        import torch
        dev = self.device if self.device is not None else torch.device('cpu')
        B = len(z)
        sc = torch.full((B, topk), float('-inf'), dtype=torch.float32, device=dev)
        ix = torch.full((B, topk), torch.iinfo(torch.int64).max, dtype=torch.int64, device=dev)
        if self.hi > self.lo:
            li, ls = self.local_nn(z, topk, col_stride)
            li = torch.as_tensor(li, dtype=torch.int64, device=dev).reshape(B, -1)
            ls = torch.as_tensor(ls, dtype=torch.float32, device=dev).reshape(B, -1)
            ix[:, :li.shape[1]] = li + self.lo
            sc[:, :ls.shape[1]] = ls
        return ix, sc

    def nn(self, z, topk=1, col_stride=1):
        import torch
        ix, sc = self.local_candidates(z, topk, col_stride)
        B = len(z)
        if self.world_size > 1:
            packed = torch.stack([ix, sc.view(torch.int32).to(torch.int64)], dim=2).contiguous()        # [B,k,2]
            gathered = torch.empty((self.world_size,) + tuple(packed.shape), dtype=torch.int64, device=ix.device)
            self.dist.all_gather_into_tensor(gathered.view(-1, 2), packed.view(-1, 2), group=self.group)
            ix = gathered[..., 0].permute(1, 0, 2).reshape(B, -1)
            sc = gathered[..., 1].to(torch.int32).view(torch.float32).permute(1, 0, 2).reshape(B, -1)
        return merge_topk(sc, ix, topk)
