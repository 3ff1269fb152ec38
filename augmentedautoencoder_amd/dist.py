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


def _group_shape(dist, group, world_size, rank, what):
    """(world_size, rank, gather?) of an engine.  Sizes left to None come from the process group (a single-rank group
    still runs its collective); explicit sizes must agree with an initialised group, except the explicit single-rank
    engine (world_size = 1, rank 0), which never gathers -- a local engine inside a larger job."""
    initialised = dist.is_available() and dist.is_initialized()
    if world_size is None:
        ws = dist.get_world_size(group) if initialised else 1
        rk = int(rank) if rank is not None else (dist.get_rank(group) if initialised else 0)
        return ws, rk, initialised
    ws = int(world_size)
    rk = int(rank) if rank is not None else (dist.get_rank(group) if initialised and ws > 1 else 0)
    if ws > 1 and initialised and ws != dist.get_world_size(group):
        raise ValueError('%s: world_size=%d but the process group has %d ranks' % (what, ws, dist.get_world_size(group)))
    # (ws > 1 without a group: shard engines built by hand, e.g. to merge their local candidates in one process --
    #  only the collective itself needs the group)
    if not 0 <= rk < ws:
        raise ValueError('%s: rank %d outside [0, %d)' % (what, rk, ws))
    return ws, rk, ws > 1


class ShardedPoseEngine(object):
    """local_infer(obj_id, crops_subset) -> (idx int64 [n] or [n,1], score float32 likewise); crops_subset is
    ``crops[positions]`` (numpy array or torch tensor, whatever the caller passed in).  local_infer_many (optional)
    answers all of the rank's objects in one call instead.

    pack_pairs(idx, score, pos int32 tensor, packed) / unpack_pairs(gathered, owner int32 tensor, n, rows_per_rank,
    idx_out, score_out): optional one-launch writers of the gather payload and of its way back (the HIP path passes
    ``engine.pack_pairs`` / ``engine.unpack_pairs``); without them framework tensor ops do it (CPU / gloo tests).

    Nothing is allocated per call once a batch layout has been seen: the payload buffer (sentinel -1 in the rows of
    other ranks, written once), the gather buffer and the outputs belong to the cached plan -- the (idx, score) tensors
    a call returns are overwritten by the next call with the same layout."""

    def __init__(self, local_infer, world_size=None, rank=None, group=None, device=None, pack_pairs=None, unpack_pairs=None, local_infer_many=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.distributed = dist.is_available() and dist.is_initialized()
        self.world_size, self.rank, self._gather = _group_shape(dist, group, world_size, rank, 'ShardedPoseEngine')
        self.local_infer = local_infer
        # local_infer_many({obj_id: crops_subset}) -> {obj_id: (idx, score)} or (idx_all, score_all, [obj ids in concatenation order]): ALL of this rank's objects in one call (the HIP path answers a
        # rank's buckets with one launch per conv layer across its objects: engine.MultiObjectQuery); local_infer may then be None
        self.local_infer_many = local_infer_many
        self.device = device
        self.pack_pairs = pack_pairs
        self.unpack_pairs = unpack_pairs
        self._plan_key, self._plan = None, None

    def plan(self, class_ids):
        """Routing of one batch layout, reusable while the class ids stay the same (a detector's boxes change per frame,
        a benchmark's do not): this rank's buckets {obj: positions}, their position tensors on the device, the owner of
        every batch row, and the buffers of the exchange."""
        import torch
        key = tuple(np.asarray(class_ids).tolist())
        if key != self._plan_key:
            dev = self.device if self.device is not None else torch.device('cpu')
            B = len(key)
            buckets = route(class_ids, self.world_size, self.rank)
            pos_dev = {o: torch.as_tensor(p, dtype=torch.int32, device=dev) for o, p in buckets.items()}
            owners = torch.as_tensor([owner_of(o, self.world_size) for o in key], dtype=torch.int32, device=dev)
            bufs = {
                # fixed-capacity payload, -1 sentinel for "not mine" (equal sizes for all_gather); this rank's rows are
                # overwritten by every call, the others keep the sentinel for the life of the plan
                'packed': torch.full((B, 2), -1, dtype=torch.int64, device=dev),
                'gathered': torch.empty((self.world_size * B, 2), dtype=torch.int64, device=dev) if self._gather else None,
                'idx': torch.empty((B,), dtype=torch.int64, device=dev),
                'score': torch.empty((B,), dtype=torch.float32, device=dev),
                'owners64': owners.long(), 'rows': torch.arange(B, device=dev),
            }
            self._plan_key, self._plan = key, (buckets, pos_dev, owners, bufs)
        return self._plan

    def infer(self, crops, class_ids):
        """Every rank passes the same class_ids and either the whole batch ``crops`` (rows of other ranks are never
        touched) or just its own share as a dict {obj_id: crops of that object, in batch order} -- what a deployment
        that routes on the host sends to each GPU.  Every rank returns the full (idx int64 [B], score float32 [B]) in
        batch order."""
        import torch
        B = len(class_ids)
        dev = self.device if self.device is not None else torch.device('cpu')
        buckets, pos_dev, owners, bufs = self.plan(class_ids)
        packed = bufs['packed']

        def select(obj, pos):
            if isinstance(crops, dict):
                return crops[obj]
            return crops[pos_dev[obj].to(crops.device).long()] if torch.is_tensor(crops) else crops[pos]
        answers = self.local_infer_many({obj: select(obj, pos) for obj, pos in buckets.items()}) if self.local_infer_many is not None and buckets else None
        if isinstance(answers, tuple):
            # (idx_all, score_all, [obj ids in concatenation order]): the rank's answers as ONE array -- one pack launch instead of one per object
            idx_all, score_all, order = answers
            key = tuple(order)
            if bufs.get('pos_cat_key') != key:
                bufs['pos_cat'] = torch.cat([pos_dev[o] for o in order]) if order else torch.empty(0, dtype=torch.int32, device=dev)
                bufs['pos_cat_key'] = key
            idx_all = torch.as_tensor(idx_all, dtype=torch.int64, device=dev).reshape(len(bufs['pos_cat']), -1).contiguous()
            score_all = torch.as_tensor(score_all, dtype=torch.float32, device=dev).reshape(len(bufs['pos_cat']), -1).contiguous()
            if self.pack_pairs is not None:
                self.pack_pairs(idx_all, score_all, bufs['pos_cat'], packed)
            else:
                p = bufs['pos_cat'].long()
                packed[p, 0] = idx_all[:, 0]
                packed[p, 1] = score_all[:, 0].contiguous().view(torch.int32).to(torch.int64)
            buckets = {}
        for obj, pos in buckets.items():
            idx, score = answers[obj] if answers is not None else self.local_infer(obj, select(obj, pos))
            idx = torch.as_tensor(idx, dtype=torch.int64, device=dev).reshape(len(pos), -1).contiguous()
            score = torch.as_tensor(score, dtype=torch.float32, device=dev).reshape(len(pos), -1).contiguous()
            if self.pack_pairs is not None:
                self.pack_pairs(idx, score, pos_dev[obj], packed)
            else:
                p = pos_dev[obj].long()
                packed[p, 0] = idx[:, 0]
                packed[p, 1] = score[:, 0].contiguous().view(torch.int32).to(torch.int64)
        src, own = packed, None
        if self._gather:
            # the one collective of the path (RCCL over xGMI on the GPUs); a single-rank group runs it too
            self.dist.all_gather_into_tensor(bufs['gathered'], packed, group=self.group)
            src, own = bufs['gathered'], owners
        if self.unpack_pairs is not None:
            self.unpack_pairs(src, own, B, B, bufs['idx'], bufs['score'])
            return bufs['idx'], bufs['score']
        if own is not None:
            src = src.view(self.world_size, B, 2)[bufs['owners64'], bufs['rows']]
        return src[:, 0], src[:, 1].to(torch.int32).view(torch.float32)


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
        self.world_size, self.rank, self._gather = _group_shape(dist, group, world_size, rank, 'RowShardedCodebook')
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
        if self._gather:
            packed = torch.stack([ix, sc.view(torch.int32).to(torch.int64)], dim=2).contiguous()        # [B,k,2]
            gathered = torch.empty((self.world_size,) + tuple(packed.shape), dtype=torch.int64, device=ix.device)
            self.dist.all_gather_into_tensor(gathered.view(-1, 2), packed.view(-1, 2), group=self.group)
            ix = gathered[..., 0].permute(1, 0, 2).reshape(B, -1)
            sc = gathered[..., 1].to(torch.int32).view(torch.float32).permute(1, 0, 2).reshape(B, -1)
        return merge_topk(sc, ix, topk)
