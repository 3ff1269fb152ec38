"""The slice of ``Dataset`` (/root/reference/auto_pose/ae/dataset.py) that the
inference hot path touches: the cfg keyword bag, the crop shape, the
codebook-row -> rotation table, and the hook through which update_embedding
obtains the views to embed.

Rendering (OpenGL meshrenderer), augmentation and training-set generation are
out of scope: ``render_embedding_image_batch``
delegates to a pluggable *view source* instead of an OpenGL context.
"""
from __future__ import annotations

import numpy as np

from . import viewsphere as vs
from .utils import lazy_property

_VIEWSPHERE_CACHE = {}


class SyntheticViewSource(object):
    """Deterministic stand-in for the renderer: view i is a smooth pattern whose
    phase/orientation follow rotation i, on a black background, as uint8."""

    def __init__(self, shape, seed=0):
        self.shape = tuple(shape)
        self.seed = int(seed)

    def __call__(self, start, end, Rs):
        H, W, C = self.shape
        yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
        yy = (yy - H / 2.0) / H
        xx = (xx - W / 2.0) / W
        n = end - start
        batch = np.empty((n, H, W, C), dtype=np.uint8)
        bbs = np.empty((n, 4))
        for k in range(n):
            R = Rs[k]
            u = R[0, 0] * xx + R[0, 1] * yy
            v = R[1, 0] * xx + R[1, 1] * yy
            mask = (u * u / 0.16 + v * v / (0.04 + 0.1 * abs(R[2, 2]))) < 1.0
            img = np.zeros((H, W, C))
            for c in range(C):
                img[..., c] = 127.5 + 127.5 * np.sin(6.0 * (R[2, c % 3] + 1.5) * u + 9.0 * R[c % 3, 2] * v + self.seed)
            img *= mask[..., None]
            batch[k] = np.clip(img, 0, 255).astype(np.uint8)
            ys, xs = np.nonzero(mask)
            if len(xs):
                bbs[k] = vs_calc_2d_bbox(xs, ys, (W, H))
            else:
                bbs[k] = (0, 0, 1, 1)
        return batch, bbs

    def torch_batch(self, Rs, device, noise=None):
        """The same pattern for a whole stack of rotations at once, evaluated with framework tensor ops on ``device``
        (float64): uint8 [n,H,W,C] device tensor, no bounding boxes.  Input plumbing for full-size codebook builds in
        tests and benchmarks (92232 views in seconds instead of minutes); rounding may differ from __call__ in the
        last grey level, so a codebook and its queries must come from the same method.  noise: optional float tensor
        [n,H,W,C] added before the uint8 conversion."""
        import torch
        H, W, C = self.shape
        R = torch.as_tensor(np.asarray(Rs, dtype=np.float64), device=device).reshape(-1, 3, 3)
        yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float64, device=device), torch.arange(W, dtype=torch.float64, device=device),
                                indexing='ij')
        yy = ((yy - H / 2.0) / H)[None]
        xx = ((xx - W / 2.0) / W)[None]
        r = lambda i, j: R[:, i, j].reshape(-1, 1, 1)
        u = r(0, 0) * xx + r(0, 1) * yy
        v = r(1, 0) * xx + r(1, 1) * yy
        mask = (u * u / 0.16 + v * v / (0.04 + 0.1 * r(2, 2).abs())) < 1.0
        chans = [127.5 + 127.5 * torch.sin(6.0 * (r(2, c % 3) + 1.5) * u + 9.0 * r(c % 3, 2) * v + self.seed) for c in range(C)]
        img = torch.stack(chans, dim=-1) * mask[..., None]
        if noise is not None:
            img = img + noise
        return img.clamp(0, 255).to(torch.uint8)


def vs_calc_2d_bbox(xs, ys, im_size):
    """pysixd_stuff/view_sampler.py:10-15."""
    tl = (max(xs.min() - 1, 0), max(ys.min() - 1, 0))
    br = (min(xs.max() + 1, im_size[0] - 1), min(ys.max() + 1, im_size[1] - 1))
    return [tl[0], tl[1], br[0] - tl[0], br[1] - tl[1]]


class Dataset(object):

    def __init__(self, dataset_path, **kw):
        self.shape = (int(kw['h']), int(kw['w']), int(kw['c']))
        self.dataset_path = dataset_path
        self._kw = kw
        self._view_source = None

    def set_view_source(self, fn):
        """fn(start, end, Rs[start:end]) -> (batch [n,H,W,C] uint8 or float in [0,1], obj_bbs [n,4])."""
        self._view_source = fn

    @lazy_property
    def viewsphere_for_embedding(self):
        kw = self._kw
        key = (int(kw['min_n_views']), float(kw['radius']), int(kw['num_cyclo']))
        if key not in _VIEWSPHERE_CACHE:
            _VIEWSPHERE_CACHE[key] = vs.viewsphere_for_embedding(*key)
        return _VIEWSPHERE_CACHE[key]

    @property
    def embedding_size(self):
        return len(self.viewsphere_for_embedding)

    def render_embedding_image_batch(self, start, end):
        """(batch float64 in [0,1] or uint8, obj_bbs) for codebook rows [start,end)
        (dataset.py:308-352, with the OpenGL render behind the view source)."""
        if self._view_source is None:
            raise NotImplementedError(
                'Rendering is out of scope of this package: attach a view source with '
                'Dataset.set_view_source(fn) (e.g. pre-rendered views, or SyntheticViewSource).')
        return self._view_source(start, end, self.viewsphere_for_embedding[start:end])
