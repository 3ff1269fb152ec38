"""Imported FIRST by the tools that switch between kernel variants of the experiments build (8-wave / 3-slab wave-split-K, the
persistent per-detection launch, register-staged igemm, the round-1 scans, in-kernel timelines, K-loop ablation): makes the
Python mirror load libaae_hip_experiments.so (`python __graft_entry__.py experiments` builds it) instead of the product library."""
import os
import sys

os.environ.setdefault('AAE_EXPERIMENTS', '1')
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = os.path.join(_ROOT, 'augmentedautoencoder_amd', 'libaae_hip_experiments.so')
if os.environ['AAE_EXPERIMENTS'] not in ('', '0') and not os.path.exists(_LIB):
    sys.stderr.write('%s is not built: run `python __graft_entry__.py experiments` (hipcc -DAAE_EXPERIMENTS) first\n' % _LIB)
    sys.exit(3)
