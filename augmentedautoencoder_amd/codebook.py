"""``Codebook`` with the method signatures of
/root/reference/auto_pose/ae/codebook.py (nearest_rotation, auto_pose6d,
nearest_rotation_batch, test_embedding, update_embedding), backed by the HIP
encoder + codebook-scan engines.  ``session`` arguments are accepted and
ignored (may be None)."""
from __future__ import annotations

import ast

import numpy as np

from . import session as S
from . import utils as u


def _parse_K(text):
    """[Dataset] K is written as a Python list with arithmetic (``720/2``, cfg/train_template.cfg:11); the reference
    eval()s the text.  Here the expression is parsed and only lists / tuples of numbers combined with + - * / (and
    unary minus) are evaluated -- nothing else a cfg file could smuggle in is executed."""
    def ev(node):
        if isinstance(node, (ast.List, ast.Tuple)):
            return [ev(e) for e in node.elts]
        if isinstance(node, ast.Constant) and isinstance(node.value, (int, float)) and not isinstance(node.value, bool):
            return node.value
        if isinstance(node, ast.UnaryOp) and isinstance(node.op, (ast.USub, ast.UAdd)):
            v = ev(node.operand)
            return -v if isinstance(node.op, ast.USub) else v
        if isinstance(node, ast.BinOp) and isinstance(node.op, (ast.Add, ast.Sub, ast.Mult, ast.Div)):
            a, b = ev(node.left), ev(node.right)
            if isinstance(a, list) or isinstance(b, list):
                raise ValueError('[Dataset] K: arithmetic on a list')
            return {ast.Add: a + b, ast.Sub: a - b, ast.Mult: a * b}[type(node.op)] if not isinstance(node.op, ast.Div) else a / b
        raise ValueError('[Dataset] K: only numbers, + - * / and lists are allowed, found %s' % type(node).__name__)
    return ev(ast.parse(text.strip(), mode='eval').body)


_BATCH_GEOMETRY_OK = None


def _batch_geometry_matches_scalar():
    """Once per process: do NumPy's array loops of arctan / cos / sin / sqrt give an element the bits the scalar calls of
    auto_pose6d (codebook.py:118-127) get?  (One ufunc inner loop serves both on the builds seen so far; a build whose SIMD
    kernels disagree with its scalar path makes poses_from_indices fall back to the per-detection loop.)"""
    global _BATCH_GEOMETRY_OK
    if _BATCH_GEOMETRY_OK is None:
        rng = np.random.default_rng(12345)
        v = np.concatenate([rng.uniform(-3.0, 3.0, 509), rng.uniform(-1e-3, 1e-3, 64), rng.uniform(-300.0, 300.0, 64), [0.0, 1.0, -1.0]])
        ok = True
        for fn in (np.arctan, np.cos, np.sin):
            ok = ok and np.array_equal(fn(v), np.array([fn(x) for x in v]))
        a = np.abs(v) + 1e-9
        ok = ok and np.array_equal(np.sqrt(a), np.array([np.sqrt(x) for x in a]))
        _BATCH_GEOMETRY_OK = bool(ok)
    return _BATCH_GEOMETRY_OK


class Codebook(object):

    def __init__(self, encoder, dataset, embed_bb):
        self._encoder = encoder
        self._dataset = dataset
        self.embed_bb = embed_bb

        J = encoder.latent_space_size
        embedding_size = self._dataset.embedding_size
        # tf.Variable(np.zeros((embedding_size, J)), float32, trainable=False) -- codebook.py:28-33
        self._embedding_host = np.zeros((embedding_size, J), dtype=np.float32)
        self._engine = None
        self._device = None

        self.normalized_embedding_query = S.Op('normalized_embedding_query', lambda feed: self._run_query(feed, True))
        self.embedding_normalized = S.Op('embedding_normalized', lambda feed: self.embedding_value())
        self.embedding = S.Placeholder((embedding_size, J), 'embedding')
        self.embedding_assign_op = S.Op('embedding_assign_op', lambda feed: self.assign_embedding(self._fed(feed, self.embedding)))
        if embed_bb:
            self._obj_bbs_host = np.zeros((embedding_size, 4), dtype=np.int32)
            self.embed_obj_bbs_var = S.Op('embed_obj_bbs_var', lambda feed: self.embed_obj_bbs_value())
            self.embed_obj_bbs = S.Placeholder((embedding_size, 4), 'embed_obj_bbs')
            self.embed_obj_bbs_assign_op = S.Op('embed_obj_bbs_assign_op',
                                                lambda feed: self.assign_obj_bbs(self._fed(feed, self.embed_obj_bbs)))
            self.embed_obj_bbs_values = None
        self.cos_similarity = S.Op('cos_similarity', lambda feed: self._run_similarity(feed))
        self.nearest_neighbor_idx = S.Op('nearest_neighbor_idx', lambda feed: self._run_argmax(feed))
        S.register(codebook=self)

    # ---- variable plumbing ---------------------------------------------------
    @staticmethod
    def _fed(feed, placeholder):
        for k, v in feed.items():
            if k is placeholder:
                return v
        raise ValueError('feed_dict has no value for %r' % (placeholder,))

    def embedding_value(self):
        return self._embedding_host.copy()

    def embed_obj_bbs_value(self):
        return self._obj_bbs_host.copy()

    def assign_embedding(self, normalized_embedding):
        """embedding_assign_op: float64/float32 [N,J] -> float32 variable (codebook.py:35-36,216)."""
        emb = np.asarray(normalized_embedding).astype(np.float32)
        if emb.shape != self._embedding_host.shape:
            raise ValueError('embedding has shape %s, variable is %s' % (emb.shape, self._embedding_host.shape))
        self._embedding_host = np.ascontiguousarray(emb)
        if self._engine is not None:
            self._engine.update(self._embedding_host)
        return self._embedding_host

    def assign_obj_bbs(self, obj_bbs):
        """embed_obj_bbs_assign_op: -> int32 variable (codebook.py:46-47,219)."""
        self._obj_bbs_host = np.asarray(obj_bbs).astype(np.int32).reshape(self._embedding_host.shape[0], 4)
        self.embed_obj_bbs_values = None
        return self._obj_bbs_host

    def close(self, close_encoder=False):
        """Free the device copy of the codebook now (and, if asked, the encoder's weights) and leave the module
        registry.  The host arrays stay: a later query rebuilds the device state."""
        if self._engine is not None:
            self._engine.close()
            self._engine = None
        S.unregister(self)
        if close_encoder:
            self._encoder.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    @property
    def engine(self):
        if self._engine is None:
            from .engine import CodebookEngine
            # extension: set codebook.codebook_dtype = 'bf16' before first use to hold the rows as
            # bfloat16 on the device (half the HBM bytes per scan; default 'f32' = reference storage)
            self._engine = CodebookEngine(self._embedding_host, device=self._encoder.engine.device,
                                          dtype=getattr(self, 'codebook_dtype', 'f32'))
        return self._engine

    # ---- fetchables ------------------------------------------------------------
    def _prep(self, x):
        """codebook.py:58-61: uint8 -> /255. (done exactly, as a 256-entry table, in the
        first kernel), 3-D -> add the batch dimension."""
        if getattr(x, 'ndim', None) == 3 or (hasattr(x, 'dim') and x.dim() == 3):
            x = x[None]
        return x

    def _encode(self, x):
        """Encoder.z on the device, range-checked: in split-precision mode the flags of the forward are read here (the
        callers below derive further device results from z), in exact fp32 -- the default -- settle() is a no-op."""
        return self._encoder.engine.encode_checked(self._prep(x))

    def _encode_nn_host(self, x, stride):
        """fused encoder + top-1 scan -> indices on the host; the range flag rides along with the result copy"""
        eng = self._encoder.engine
        _, idx, _ = eng.encode_nn(self.engine, self._prep(x), stride)
        idcs = idx[:, 0].cpu().numpy()
        if eng.settle():                      # (split precision, out of range: recomputed in exact fp32, in place)
            idcs = idx[:, 0].cpu().numpy()
        return idcs

    def _run_similarity(self, feed):
        return self.engine.similarity(self._encode(self._encoder._feed(feed))).cpu().numpy()

    def _run_argmax(self, feed):
        return self._encode_nn_host(self._encoder._feed(feed), 1)

    def _run_query(self, feed, normalized):
        z = self._encode(self._encoder._feed(feed))
        if normalized:
            z = self.engine.l2_normalize(z)
        return z.cpu().numpy()

    # ---- reference API -----------------------------------------------------------
    def nearest_rotation(self, session, x, top_n=1, upright=False, return_idcs=False):
        """R_model2cam of the nearest codebook entries (codebook.py:55-75)."""
        if top_n == 1:
            # encoder + scan in one C call (aae_encode_nn): per detection that is six launches
            stride = int(self._dataset._kw['num_cyclo']) if upright else 1
            idcs = self._encode_nn_host(x, stride)
        else:
            z = self._encode(x)
            if z.shape[0] != 1:
                # the reference squeezes the [B,N] similarity (codebook.py:70): only B == 1 is meaningful
                raise ValueError('top_n > 1 needs a single crop (got a batch of %d)' % z.shape[0])
            idx, _ = self.engine.nn(z, int(top_n), 1)
            idcs = idx[0].cpu().numpy()
        if return_idcs:
            return idcs
        return self._dataset.viewsphere_for_embedding[idcs].squeeze()

    def nearest_rotation_with_scores(self, x, top_n=1, upright=False):
        """Extension: (indices, cosine scores) without materialising the similarity."""
        z = self._encode(x)
        stride = int(self._dataset._kw['num_cyclo']) if (upright and top_n == 1) else 1
        idx, score = self.engine.nn(z, int(top_n), stride)
        return idx.cpu().numpy(), score.cpu().numpy()

    def auto_pose6d(self, session, x, predicted_bb, K_test, top_n, train_args, depth_pred=None, upright=False):
        """Rotation + translation estimate from a detector crop (codebook.py:79-129)."""
        idcs = np.atleast_1d(self.nearest_rotation(session, x, top_n=top_n, upright=upright, return_idcs=True))
        return self.pose_from_indices(idcs, predicted_bb, K_test, train_args, depth_pred=depth_pred)

    def pose_prepare(self, predicted_bb, K_test, train_args, depth_pred=None):
        """The part of pose_from_indices that does not depend on the matched rows -- the estimator computes it while the GPU is
        still busy with the query and passes it back as ``prepared``.  Same expressions, same operands: same bits."""
        K_train, render_radius = self._train_geometry_of(train_args)
        K_diag_ratio = np.sqrt(K_test[0, 0] ** 2 + K_test[1, 1] ** 2) / np.sqrt(K_train[0, 0] ** 2 + K_train[1, 1] ** 2)
        if self.embed_obj_bbs_values is None:
            self.embed_obj_bbs_values = self.embed_obj_bbs_value()
        pred_diag = np.linalg.norm(np.float32(predicted_bb[2:])) if depth_pred is None else None
        cx_test = predicted_bb[0] + predicted_bb[2] / 2 - K_test[0, 2]
        cy_test = predicted_bb[1] + predicted_bb[3] / 2 - K_test[1, 2]
        return K_train, render_radius, K_diag_ratio, pred_diag, cx_test, cy_test

    def pose_from_indices(self, idcs, predicted_bb, K_test, train_args, depth_pred=None, prepared=None):
        """The geometry half of auto_pose6d (codebook.py:82-129) for already-matched codebook
        rows -- what the batched estimator calls after one encode+scan over all detections."""
        idcs = np.atleast_1d(idcs)
        top_n = len(idcs)
        Rs_est = self._dataset.viewsphere_for_embedding[idcs]      # fancy index -> copy

        # [Dataset] K / RADIUS are parsed once per config (the per-detection estimator flow calls this for every box; the
        # parse was 40 % of the call, the two configparser lookups another 10 us)
        if prepared is None:
            prepared = self.pose_prepare(predicted_bb, K_test, train_args, depth_pred)
        K_train, render_radius, K_diag_ratio, pred_diag, cx_test, cy_test = prepared

        ts_est = np.empty((top_n, 3))
        for i, idx in enumerate(idcs):
            rendered_bb = self.embed_obj_bbs_values[idx].squeeze()
            if depth_pred is None:
                diag_ratio = np.linalg.norm(np.float32(rendered_bb[2:])) / pred_diag
                z = diag_ratio * K_diag_ratio * render_radius
            else:
                z = depth_pred
            cx_train = rendered_bb[0] + rendered_bb[2] / 2. - K_train[0, 2]
            cy_train = rendered_bb[1] + rendered_bb[3] / 2. - K_train[1, 2]
            tx = cx_test * z / K_test[0, 0] - cx_train * render_radius / K_train[0, 0]
            ty = cy_test * z / K_test[1, 1] - cy_train * render_radius / K_train[1, 1]
            ts_est[i] = (tx, ty, z)
            # the codebook holds centred views; compensate for the off-centre crop
            ay = np.arctan(tx / np.sqrt(z ** 2 + ty ** 2))
            ax = -np.arctan(ty / z)
            Rx = np.array([[1, 0, 0], [0, np.cos(ax), -np.sin(ax)], [0, np.sin(ax), np.cos(ax)]])
            Ry = np.array([[np.cos(ay), 0, np.sin(ay)], [0, 1, 0], [-np.sin(ay), 0, np.cos(ay)]])
            Rs_est[i] = np.dot(Ry, np.dot(Rx, Rs_est[i]))
        return (Rs_est, ts_est)

    def _train_geometry_of(self, train_args):
        """(K_train 3x3, render radius) of a training config: [Dataset] K / RADIUS parsed once per config object and text"""
        cache = self.__dict__.setdefault('_train_geometry_by_id', {})
        hit = cache.get(id(train_args))
        if hit is None or hit[0] is not train_args:
            key = (train_args.get('Dataset', 'K'), train_args.get('Dataset', 'RADIUS'))
            hit = (train_args, np.array(_parse_K(key[0])).reshape(3, 3), float(key[1]))
            cache[id(train_args)] = hit
        return hit[1], hit[2]

    def poses_prepare(self, predicted_bbs, K_test, train_args, depth_preds=None):
        """The part of poses_from_indices that does not depend on the matched rows (see pose_prepare)."""
        n = len(predicted_bbs)
        if n < 4 or not _batch_geometry_matches_scalar():
            return [self.pose_prepare(bb, K_test, train_args, None if depth_preds is None else depth_preds[k]) for k, bb in enumerate(predicted_bbs)]
        K_train, render_radius = self._train_geometry_of(train_args)
        K_diag_ratio = np.sqrt(K_test[0, 0] ** 2 + K_test[1, 1] ** 2) / np.sqrt(K_train[0, 0] ** 2 + K_train[1, 1] ** 2)
        if self.embed_obj_bbs_values is None:
            self.embed_obj_bbs_values = self.embed_obj_bbs_value()
        pb = np.array([[float(v) for v in bb] for bb in predicted_bbs], dtype=np.float64).reshape(n, 4)
        den = None if depth_preds is not None else np.array([np.linalg.norm(np.float32(bb[2:])) for bb in predicted_bbs], dtype=np.float32)
        cx_test = pb[:, 0] + pb[:, 2] / 2 - K_test[0, 2]
        cy_test = pb[:, 1] + pb[:, 3] / 2 - K_test[1, 2]
        return K_train, render_radius, K_diag_ratio, den, cx_test, cy_test

    def poses_from_indices(self, idcs, predicted_bbs, K_test, train_args, depth_preds=None, prepared=None):
        """pose_from_indices for n detections of one object at once: idcs [n] (top-1 rows), predicted_bbs [n][4] ->
        (Rs [n,3,3], ts [n,3]), every number bit-identical to n scalar calls (codebook.py:84-129).  Array arithmetic is used
        only where NumPy gives the same bits for an element of an array as for a scalar: + - * / sqrt and the ufuncs arctan /
        cos / sin (one inner loop serves both; checked once per process against the scalar form, which is kept as the
        fallback).  The two float32 norms, the scalar power z ** 2 (libm pow, not x * x) and the BLAS 3 x 3 products stay per
        detection -- their array forms round differently."""
        idcs = np.asarray(idcs, dtype=np.int64).reshape(-1)
        n = len(idcs)
        if n == 0:
            return np.empty((0, 3, 3)), np.empty((0, 3))
        if n < 4 or not _batch_geometry_matches_scalar():          # (a handful of detections: the array form's fixed cost, ~45 us, exceeds 28 us per scalar call)
            out = [self.pose_from_indices([int(i)], bb, K_test, train_args, depth_pred=None if depth_preds is None else depth_preds[k],
                                          prepared=None if prepared is None else prepared[k])
                   for k, (i, bb) in enumerate(zip(idcs, predicted_bbs))]
            return np.concatenate([r for r, _ in out], axis=0), np.concatenate([t for _, t in out], axis=0)
        Rs_est = self._dataset.viewsphere_for_embedding[idcs]      # fancy index -> copy
        # (predicted boxes keep their Python-float arithmetic -- x + w / 2 ...: float64 arrays of the same values give the same bits)
        if prepared is None:
            prepared = self.poses_prepare(predicted_bbs, K_test, train_args, depth_preds)
        K_train, render_radius, K_diag_ratio, den, cx_test, cy_test = prepared
        rb = self.embed_obj_bbs_values[idcs]                       # [n,4] int32
        if depth_preds is None:
            num = np.array([np.linalg.norm(np.float32(r[2:])) for r in rb], dtype=np.float32)
            # float32 quotient, then float64 products, as the scalar code does it -- the cast is explicit so that NumPy 1.x's
            # value-based casting (float32 array x float64 scalar -> float32) gives the same bits as NumPy 2's promotion
            z = (num / den).astype(np.float64) * K_diag_ratio * render_radius
        else:
            z = np.array([float(d) for d in depth_preds], dtype=np.float64)
        cx_train = rb[:, 0] + rb[:, 2] / 2. - K_train[0, 2]
        cy_train = rb[:, 1] + rb[:, 3] / 2. - K_train[1, 2]
        tx = cx_test * z / K_test[0, 0] - cx_train * render_radius / K_train[0, 0]
        ty = cy_test * z / K_test[1, 1] - cy_train * render_radius / K_train[1, 1]
        ts_est = np.stack([tx, ty, z], axis=1)
        zz = np.array([v ** 2 for v in z])                         # scalar power (pow), element by element
        tyy = np.array([v ** 2 for v in ty])
        ay = np.arctan(tx / np.sqrt(zz + tyy))
        ax = -np.arctan(ty / z)
        cax, sax, cay, say = np.cos(ax), np.sin(ax), np.cos(ay), np.sin(ay)
        for i in range(n):
            Rx = np.array([[1, 0, 0], [0, cax[i], -sax[i]], [0, sax[i], cax[i]]])
            Ry = np.array([[cay[i], 0, say[i]], [0, 1, 0], [-say[i], 0, cay[i]]])
            Rs_est[i] = np.dot(Ry, np.dot(Rx, Rs_est[i]))
        return Rs_est, ts_est

    def nearest_rotation_batch(self, session, x):
        """Batched arg-max on the device (codebook.py:131-133)."""
        idx, _ = self.engine.nn(self._encode(x), 1, 1)
        return self._dataset.viewsphere_for_embedding[idx[:, 0].cpu().numpy()]

    def test_embedding(self, sess, x, normalized=True):
        """Latent code of a crop, optionally L2-normalised (codebook.py:135-145)."""
        z = self._encode(x)
        if normalized:
            z = self.engine.l2_normalize(z)
        return z.cpu().numpy().squeeze()

    def update_embedding(self, session, batch_size):
        """Re-embed every viewsphere rotation and store the normalised codebook
        (codebook.py:190-219).  float64 accumulation buffer and float64 row
        normalisation without epsilon, as in the reference."""
        embedding_size = self._dataset.embedding_size
        J = self._encoder.latent_space_size
        embedding_z = np.empty((embedding_size, J))
        obj_bbs = np.empty((embedding_size, 4))
        for a, e in u.batch_iteration_indices(embedding_size, batch_size):
            batch, obj_bbs_batch = self._dataset.render_embedding_image_batch(a, e)
            embedding_z[a:e] = self._encode(batch).cpu().numpy()
            if self.embed_bb:
                obj_bbs[a:e] = obj_bbs_batch
        normalized_embedding = embedding_z / np.linalg.norm(embedding_z, axis=1, keepdims=True)
        self.assign_embedding(normalized_embedding)
        if self.embed_bb:
            self.assign_obj_bbs(obj_bbs)
