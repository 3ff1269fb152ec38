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
