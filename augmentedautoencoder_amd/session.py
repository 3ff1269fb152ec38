"""Just enough of the tf.Session / tf.train.Saver surface for the reference's
call sites on this path to keep working.

In the reference every method of ``Codebook`` takes a ``session`` and evaluates
graph tensors through ``session.run(fetch, feed_dict)``; callers also fetch
variables directly (``sess.run(codebook.embedding_normalized)``,
/root/reference/auto_pose/eval/ae_eval.py:257).  Here the "tensors" are small
``Op`` objects bound to the HIP engines; ``Session.run`` evaluates them.  The
``session`` argument of the Codebook methods is accepted and ignored (it may be
``None``), exactly as SURVEY.md section 8b specifies.

``Saver`` persists what the reference keeps in its TF checkpoint (encoder
weights + the codebook variables) as ``<prefix>-<step>.npz`` next to a
``checkpoint`` index file (stand-in for tf.train.get_checkpoint_state).
"""
from __future__ import annotations

import os
import weakref

import numpy as np

from . import weights as W

# [(scope, weakref to an Encoder | Codebook | Decoder, kind)] in construction order.  Weak: the registry exists so
# that a Saver can find "everything built under scope X" (tf.train.Saver over a variable scope); it must not keep
# device weights and codebooks of objects alive that their owner has dropped -- a long-running estimator that rebuilds
# codebooks would otherwise leak device memory until reset_default_graph().
_GRAPH = []
_SCOPE = ['']


class variable_scope(object):
    """with variable_scope(experiment_name): ...  (ae_factory.py:131, ae_embed.py:53)"""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        _SCOPE.append(self.name)
        return self

    def __exit__(self, *exc):
        _SCOPE.pop()
        return False


def current_scope():
    return _SCOPE[-1]


def register(encoder=None, codebook=None, decoder=None):
    for kind, obj in (('encoder', encoder), ('codebook', codebook), ('decoder', decoder)):
        if obj is not None:
            _GRAPH.append((current_scope(), weakref.ref(obj), kind))


def unregister(obj):
    """Forget a module (its close() calls this)."""
    _GRAPH[:] = [(s, r, k) for (s, r, k) in _GRAPH if r() is not None and r() is not obj]


def graph_members(scope=None):
    """[(scope, encoder, codebook, decoder)] rows (one of the three set) of the live modules, construction order."""
    out = []
    for s, r, kind in _GRAPH:
        obj = r()
        if obj is None or (scope is not None and s != scope):
            continue
        out.append((s, obj if kind == 'encoder' else None, obj if kind == 'codebook' else None, obj if kind == 'decoder' else None))
    return out


def reset_default_graph():
    """tf.reset_default_graph(): forget every module; engines still referenced elsewhere are untouched."""
    del _GRAPH[:]


class Placeholder(object):
    """Stand-in for tf.placeholder(tf.float32, [None, H, W, C]) (ae_factory.py:133)."""

    def __init__(self, shape, name='x'):
        self.shape = tuple(shape)
        self.name = name

    def __repr__(self):
        return 'Placeholder(%s, shape=%s)' % (self.name, (None,) + self.shape)


class Op(object):
    """A fetchable: fn(feed_dict) -> np.ndarray."""

    def __init__(self, name, fn):
        self.name = name
        self._fn = fn

    def eval(self, feed_dict=None):
        return self._fn(feed_dict or {})

    def __repr__(self):
        return 'Op(%s)' % self.name


class Session(object):
    def __init__(self, config=None):
        self.config = config

    def run(self, fetches, feed_dict=None):
        if isinstance(fetches, (list, tuple)):
            return [self.run(f, feed_dict) for f in fetches]
        if not isinstance(fetches, Op):
            raise TypeError('cannot fetch %r: only Encoder/Codebook ops are evaluable' % (fetches,))
        return fetches.eval(feed_dict)

    def close(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


class CheckpointState(object):
    def __init__(self, model_checkpoint_path, all_model_checkpoint_paths):
        self.model_checkpoint_path = model_checkpoint_path
        self.all_model_checkpoint_paths = all_model_checkpoint_paths


def get_checkpoint_state(ckpt_dir):
    """tf.train.get_checkpoint_state stand-in.  Understands both the text CheckpointState
    proto TensorFlow writes (model_checkpoint_path: "chkpt-30000" ...) and this package's
    plain list of .npz files."""
    index = os.path.join(ckpt_dir, 'checkpoint')
    if not os.path.exists(index):
        return None
    with open(index) as f:
        text = f.read()

    def absolute(p):
        return p if os.path.isabs(p) else os.path.join(ckpt_dir, p)

    from .tf_checkpoint import parse_checkpoint_state
    latest, every = parse_checkpoint_state(text)
    if latest is not None:
        return CheckpointState(absolute(latest), [absolute(p) for p in (every or [latest])])
    paths = [absolute(line.strip()) for line in text.splitlines() if line.strip()]
    if not paths:
        return None
    return CheckpointState(paths[-1], paths)


class Saver(object):
    """Saver(scope=...) covers the modules built under that variable scope (the
    reference builds one scoped Saver per object, m3_interface/ae_pose_estimator.py:76-78);
    scope=None covers everything built so far."""

    def __init__(self, var_list=None, save_relative_paths=True, scope=None):
        self.scope = scope

    def _members(self):
        return graph_members(self.scope)

    def save(self, session, save_path, global_step=None):
        path = save_path if global_step is None else '%s-%d' % (save_path, int(global_step))
        encoder = codebook = decoder = None
        for _, e, c, d in self._members():
            encoder = e if e is not None else encoder
            codebook = c if c is not None else codebook
            decoder = d if d is not None else decoder
        if encoder is None or encoder.weights is None:
            raise RuntimeError('Saver.save: no encoder with weights under scope %r' % self.scope)
        emb = bbs = None
        if codebook is not None:
            emb = codebook.embedding_value()
            bbs = codebook.embed_obj_bbs_value() if codebook.embed_bb else None
        weights = dict(encoder.weights)
        if decoder is not None and decoder.weights is not None:
            weights.update(decoder.weights)
        W.save_npz(path + '.npz', weights, emb, bbs)
        # The index stays ONE format -- TensorFlow's text CheckpointState -- with the new file as model_checkpoint_path.
        # (A directory that came with a TensorFlow checkpoint keeps its entries in all_model_checkpoint_paths; were the
        # new .npz merely appended as a bare line, get_checkpoint_state would keep answering the TF bundle and the
        # codebook just built by ae_embed would never be restored.)
        ckpt_dir = os.path.dirname(path)
        index = os.path.join(ckpt_dir, 'checkpoint')
        every = []
        if os.path.exists(index):
            with open(index) as f:
                text = f.read()
            from .tf_checkpoint import parse_checkpoint_state
            latest, every = parse_checkpoint_state(text)
            if latest is None:                                   # index written by an earlier version: one path per line
                every = [l.strip() for l in text.splitlines() if l.strip()]
            elif not every:
                every = [latest]
        rel = os.path.basename(path) + '.npz'
        every = [p for p in every if p != rel] + [rel]

        def quoted(p):
            return '"%s"' % p.replace('\\', '\\\\').replace('"', '\\"')
        with open(index, 'w') as f:
            f.write('model_checkpoint_path: %s\n' % quoted(rel))
            for p in every:
                f.write('all_model_checkpoint_paths: %s\n' % quoted(p))
        return path

    def restore(self, session, ckpt_path):
        """ckpt_path: a native '<prefix>.npz' (extension optional) or the prefix of a TensorFlow
        checkpoint-v2 ('<prefix>.index' + '<prefix>.data-*'), read without TensorFlow by
        tf_checkpoint.load_aae_variables under this saver's variable scope."""
        if not ckpt_path.endswith('.npz') and os.path.exists(ckpt_path + '.index'):
            from .tf_checkpoint import load_aae_variables
            scope = self.scope
            if scope is None:
                scopes = sorted({m[0] for m in self._members() if m[0]})
                scope = scopes[0] if len(scopes) == 1 else None
            weights, emb, bbs = load_aae_variables(ckpt_path, scope)
        else:
            if not ckpt_path.endswith('.npz'):
                ckpt_path = ckpt_path + '.npz'
            weights, emb, bbs = W.load_npz(ckpt_path)
        for _, e, c, d in self._members():
            if e is not None:
                e.load_weights(weights)
            if d is not None:
                d.load_weights(weights)
            if c is not None:
                if emb is not None:
                    c.assign_embedding(emb)
                if bbs is not None and c.embed_bb:
                    c.assign_obj_bbs(bbs)
