"""N > 1 path on CPU: world_size-2 gloo run of the object-sharded routing + all_gather
(augmentedautoencoder_amd/dist.py).  The per-object compute is the oracle here --
only the distribution logic is under test."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _problem():
    from oracle import synth
    n_obj, B, N, J = 5, 37, 36 * 20, 32
    Es = [synth.make_codebook(N, J, seed=100 + o, planted_duplicates=4) for o in range(n_obj)]
    rng = np.random.default_rng(0)
    class_ids = rng.integers(0, n_obj, B)
    z = rng.standard_normal((B, J)).astype(np.float32)       # "crops" are latents in this logic test
    return n_obj, Es, class_ids, z


def _oracle_infer(Es):
    from oracle import reference_cpu as ref

    def fn(obj, zsub):
        cs = ref.cos_similarity(np.asarray(zsub), Es[obj])
        return np.argmax(cs, axis=1), cs.max(axis=1).astype(np.float32)
    return fn


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from augmentedautoencoder_amd.dist import ShardedPoseEngine
    n_obj, Es, class_ids, z = _problem()
    calls = []
    base = _oracle_infer(Es)

    def infer(obj, zsub):
        calls.append(obj)
        return base(obj, zsub)

    eng = ShardedPoseEngine(infer)
    idx, score = eng.infer(z, class_ids)
    # pre-routed form: this rank is handed only ITS buckets (what a host that routes by class id sends to each GPU),
    # with a caller-supplied pair writer standing in for engine.pack_pairs
    from augmentedautoencoder_amd.dist import route
    mine = {o: z[pos] for o, pos in route(class_ids, world, rank).items()}
    writes = []

    def pack(ix, sc, pos, packed):
        writes.append(len(pos))
        packed[pos.long(), 0] = ix.reshape(-1)
        packed[pos.long(), 1] = sc.reshape(-1).view(torch.int32).to(torch.int64)

    eng2 = ShardedPoseEngine(infer, pack_pairs=pack)
    idx2, score2 = eng2.infer(mine, class_ids)
    assert torch.equal(idx, idx2) and torch.equal(score, score2) and sum(writes) == sum(len(v) for v in mine.values())
    idx3, _ = eng2.infer(mine, class_ids)                     # the cached routing plan of an unchanged batch layout
    assert torch.equal(idx3, idx)
    # the one-launch way back (engine.unpack_pairs on the GPU) as a CPU double: same answers, outputs owned by the plan and
    # re-used by the next call with the same layout (no allocation per step)
    def unpack(gathered, owner, n, rows_per_rank, idx_out, score_out):
        g = gathered.view(-1, rows_per_rank, 2)
        own = owner.long() if owner is not None else torch.zeros(n, dtype=torch.int64)
        idx_out[:n] = g[own, torch.arange(n), 0]
        score_out[:n] = g[own, torch.arange(n), 1].to(torch.int32).view(torch.float32)

    eng3 = ShardedPoseEngine(infer, pack_pairs=pack, unpack_pairs=unpack)
    idx4, score4 = eng3.infer(mine, class_ids)
    assert torch.equal(idx4, idx) and torch.equal(score4, score)
    ptrs = (idx4.data_ptr(), score4.data_ptr(), eng3._plan[3]['packed'].data_ptr(), eng3._plan[3]['gathered'].data_ptr())
    idx5, score5 = eng3.infer(mine, class_ids)
    assert (idx5.data_ptr(), score5.data_ptr(), eng3._plan[3]['packed'].data_ptr(), eng3._plan[3]['gathered'].data_ptr()) == ptrs
    assert torch.equal(idx5, idx) and torch.equal(score5, score)
    # all of the rank's objects in ONE call (the HIP path: engine.MultiObjectQuery, one launch per conv layer across the objects), answers per object ...
    many_calls = []

    def infer_many(buckets):
        many_calls.append(sorted(buckets))
        return {o: base(o, zs) for o, zs in buckets.items()}
    eng4 = ShardedPoseEngine(None, pack_pairs=pack, unpack_pairs=unpack, local_infer_many=infer_many)
    idx6, score6 = eng4.infer(mine, class_ids)
    assert torch.equal(idx6, idx) and torch.equal(score6, score) and many_calls == [sorted(mine)]
    # ... or as one concatenated array with its object order: one pack per rank
    del writes[:]

    def infer_concat(buckets):
        order = sorted(buckets, reverse=True)
        parts = [base(o, buckets[o]) for o in order]
        return np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts]), order
    eng5 = ShardedPoseEngine(None, pack_pairs=pack, unpack_pairs=unpack, local_infer_many=infer_concat)
    idx7, score7 = eng5.infer(mine, class_ids)
    assert torch.equal(idx7, idx) and torch.equal(score7, score) and len(writes) == (1 if mine else 0)
    idx8, _ = eng5.infer(z, class_ids)                        # (whole batch handed over: the engine selects the rank's rows itself)
    assert torch.equal(idx8, idx)
    # an explicit single-rank engine inside this 2-rank job is a LOCAL engine: it never enters the collective (the
    # other rank does not call it here -- a gather would hang or fail on the size mismatch)
    if rank == 0:
        solo = ShardedPoseEngine(base, world_size=1, rank=0)
        si, ss = solo.infer(z, class_ids)
        assert torch.equal(si, idx) and torch.equal(ss, score)
    with pytest.raises(ValueError):
        ShardedPoseEngine(base, world_size=3)                 # disagrees with the initialised group
    np.savez(os.path.join(out_dir, 'rank%d.npz' % rank), idx=idx.numpy(), score=score.numpy(), calls=np.array(sorted(set(calls))))
    dist.destroy_process_group()


def test_sharded_inference_two_ranks_matches_single(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    n_obj, Es, class_ids, z = _problem()
    base = _oracle_infer(Es)
    want_idx = np.empty(len(class_ids), dtype=np.int64)
    want_score = np.empty(len(class_ids), dtype=np.float32)
    for o in range(n_obj):
        pos = np.nonzero(class_ids == o)[0]
        if len(pos):
            want_idx[pos], want_score[pos] = base(o, z[pos])
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), 'rank%d.npz' % r))
        assert np.array_equal(got['idx'], want_idx)              # every rank ends with the full answer, in batch order
        assert np.array_equal(got['score'], want_score)          # scores travel bit-exactly through the int64 packing
        assert all(o % world == r for o in got['calls'])          # a rank only ever computes its own objects


def test_single_process_no_collective():
    sys.path.insert(0, ROOT)
    from augmentedautoencoder_amd.dist import ShardedPoseEngine, owner_of, route
    n_obj, Es, class_ids, z = _problem()
    eng = ShardedPoseEngine(_oracle_infer(Es), world_size=1, rank=0)
    idx, score = eng.infer(z, class_ids)
    base = _oracle_infer(Es)
    for o in range(n_obj):
        pos = np.nonzero(class_ids == o)[0]
        wi, ws = base(o, z[pos])
        assert np.array_equal(idx.numpy()[pos], wi) and np.array_equal(score.numpy()[pos], ws)
    assert owner_of(11, 8) == 3
    r = route([3, 0, 3, 1, 8], 8, 3)
    assert list(r.keys()) == [3] and r[3].tolist() == [0, 2]
    assert route([3, 0, 3, 1, 8], 8, 0)[8].tolist() == [4]


# ---- row-sharded codebook: local top-k per rank, one all_gather, k-way merge ----
def _rows_problem():
    from oracle import synth
    N, J, B = 36 * 53 + 7, 32, 19                      # ragged: the last shard is shorter and not a multiple of 36
    E = synth.make_codebook(N, J, seed=5, planted_duplicates=12)
    rng = np.random.default_rng(3)
    z = rng.standard_normal((B, J)).astype(np.float32)
    z[:6] = E[[0, 35, 36, 71, 36 * 30, N - 1]]         # queries that hit planted duplicates on both sides of the cut
    return E, z


def _local_nn(E, lo, hi):
    from oracle import reference_cpu as ref

    def fn(z, k, stride):
        cs = ref.cos_similarity(np.asarray(z), E[lo:hi])
        cols = np.arange(0, hi - lo, stride)
        idx = cols[ref.topk_canonical(cs[:, cols], min(k, len(cols)))]
        return idx, np.take_along_axis(cs, idx, axis=1).astype(np.float32)
    return fn


def _rows_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from augmentedautoencoder_amd.dist import RowShardedCodebook, row_shard_bounds
    E, z = _rows_problem()
    lo, hi = row_shard_bounds(len(E), world, align=36)[rank]
    cb = RowShardedCodebook(_local_nn(E, lo, hi), len(E), align=36)
    out = {}
    for name, k, stride in (('top1', 1, 1), ('top5', 5, 1), ('up1', 1, 36), ('up3', 3, 36)):
        i, s = cb.nn(z, k, stride)
        out[name + '_i'], out[name + '_s'] = i.numpy(), s.numpy()
    np.savez(os.path.join(out_dir, 'rows%d.npz' % rank), **out)
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_row_sharded_codebook_matches_single_scan(tmp_path, world):
    port = _free_port()
    mp.spawn(_rows_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    E, z = _rows_problem()
    whole = _local_nn(E, 0, len(E))
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), 'rows%d.npz' % r))
        for name, k, stride in (('top1', 1, 1), ('top5', 5, 1), ('up1', 1, 36), ('up3', 3, 36)):
            wi, ws = whole(z, k, stride)
            assert np.array_equal(got[name + '_i'], wi), (name, r)      # same rows, same tie order as one scan
            assert np.array_equal(got[name + '_s'], ws), (name, r)


def test_row_shard_bounds_and_merge():
    sys.path.insert(0, ROOT)
    from augmentedautoencoder_amd.dist import merge_topk, row_shard_bounds
    assert row_shard_bounds(92232, 8, 36) == [(i * 11556, min((i + 1) * 11556, 92232)) for i in range(8)]
    b = row_shard_bounds(100, 3, 36)
    assert b == [(0, 36), (36, 72), (72, 100)] and row_shard_bounds(10, 4, 36)[1:] == [(10, 10)] * 3
    s = torch.tensor([[0.5, 0.9, 0.9, float('-inf'), 0.9]])
    i = torch.tensor([[7, 40, 3, torch.iinfo(torch.int64).max, 12]])
    mi, ms = merge_topk(s, i, 3)
    assert mi.tolist() == [[3, 12, 40]] and ms.tolist() == [[pytest.approx(0.9)] * 3]
