"""Seeded synthetic inputs (weights, crops, codebooks) for the oracle, the tests and the CPU baseline.
The generators live in the package (augmentedautoencoder_amd/synth.py) because bench.py feeds the
product path with them; this module only re-exports them so that test code keeps one import site."""
from augmentedautoencoder_amd.synth import (DEFAULT_KERNEL, DEFAULT_LATENT, DEFAULT_N, DEFAULT_NUM_FILTER,  # noqa: F401
                                            DEFAULT_SHAPE, DEFAULT_STRIDES, make_codebook, make_crops,
                                            make_queries_near_rows, make_weights)
