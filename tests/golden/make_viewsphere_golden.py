"""Generate golden fixtures for the codebook-row -> rotation mapping by running the
REFERENCE's own view sampler (pure NumPy, importable without TensorFlow/OpenCV):
/root/reference/auto_pose/ae/pysixd_stuff/view_sampler.py::sample_views.

Run in the build container (the reference tree does not exist on the GPU box):
    python tests/golden/make_viewsphere_golden.py
Writes tests/golden/viewsphere_ref.npz  (R matrices for 42/162/642/2562 views).
"""
import importlib.util
import os
import sys
import types

import numpy as np

REF = '/root/reference/auto_pose/ae/pysixd_stuff'


def load_reference_sampler():
    pkg = types.ModuleType('refpysixd')
    pkg.__path__ = [REF]
    sys.modules['refpysixd'] = pkg
    mods = {}
    for name in ('transform', 'view_sampler'):
        spec = importlib.util.spec_from_file_location('refpysixd.' + name, os.path.join(REF, name + '.py'))
        m = importlib.util.module_from_spec(spec)
        sys.modules['refpysixd.' + name] = m
        spec.loader.exec_module(m)
        mods[name] = m
    return mods['view_sampler']


def main():
    vs = load_reference_sampler()
    out = {}
    for n in (42, 162, 642, 2562):
        views, levels = vs.sample_views(n, 700.0, (0, 2 * np.pi), (-0.5 * np.pi, 0.5 * np.pi))
        out['R_%d' % n] = np.stack([v['R'] for v in views])
        out['t_%d' % n] = np.stack([v['t'] for v in views])
        out['level_%d' % n] = np.array(levels, dtype=np.int32)
    here = os.path.dirname(os.path.abspath(__file__))
    np.savez(os.path.join(here, 'viewsphere_ref.npz'), **out)
    for k, v in out.items():
        print(k, v.shape, v.dtype)


if __name__ == '__main__':
    main()
