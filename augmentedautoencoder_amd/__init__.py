"""MI355X-native (gfx950) implementation of the AugmentedAutoencoder
orientation-inference hot path: convolutional encoder forward + cosine
nearest-neighbour against the rotation codebook, behind the reference's
Encoder / Codebook / ae_factory / ae_embed API.  See DESIGN.md."""

__version__ = '0.1.0'
