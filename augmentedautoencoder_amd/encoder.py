"""``Encoder`` with the constructor and properties of
/root/reference/auto_pose/ae/encoder.py:12-68 (inference part), backed by the
HIP encoder engine instead of a TF graph."""
from __future__ import annotations

import numpy as np

from . import session as S
from .weights import EncoderConfig


class Encoder(object):

    def __init__(self, input, latent_space_size, num_filters, kernel_size, strides, batch_norm, is_training=False):
        if is_training:
            raise NotImplementedError('training graphs are out of scope (SURVEY.md section 2 rows 6-7); is_training must be False')
        self._input = input
        self._latent_space_size = int(latent_space_size)
        self._num_filters = list(num_filters)
        self._kernel_size = int(kernel_size)
        self._strides = list(strides)
        self._batch_normalization = bool(batch_norm)
        self._is_training = False
        shape = tuple(input.shape) if hasattr(input, 'shape') else tuple(input)
        self.config = EncoderConfig(shape[-3:], self._num_filters, self._strides, self._kernel_size,
                                    self._latent_space_size, self._batch_normalization)
        self.weights = None
        self._engine = None
        self._device = None
        self._z_op = S.Op('encoder/z', lambda feed: self.engine.encode_checked(self._feed(feed)).cpu().numpy())
        self._z_op.owner = self                  # lets Decoder(latent_code=encoder.z) find its encoder
        S.register(encoder=self)

    # -- reference properties ------------------------------------------------
    @property
    def x(self):
        return self._input

    @property
    def latent_space_size(self):
        return self._latent_space_size

    @property
    def z(self):
        """Fetchable latent code (encoder.py:58-68): session.run(encoder.z, {encoder.x: batch})."""
        return self._z_op

    @property
    def encoder_out(self):
        """Flattened conv output (encoder.py:37-56) as a fetchable."""
        def run(feed):
            self.engine.encode_checked(self._feed(feed))
            a = self.engine.activation(self.config.num_layers - 1)
            return a.reshape(a.shape[0], -1).cpu().numpy()
        return S.Op('encoder/encoder_out', run)

    # -- plumbing --------------------------------------------------------------
    def _feed(self, feed):
        for k, v in feed.items():
            if k is self._input:
                return v
        raise ValueError('feed_dict has no value for this encoder\'s input placeholder')

    def load_weights(self, weights, device=None):
        """Stand-in for Saver.restore on the encoder variables."""
        self.weights = {k: np.asarray(v) for k, v in weights.items()}
        self._device = device
        if self._engine is not None:
            self._engine.close()
            self._engine = None

    def close(self):
        """Free the device weights now (the TF session of the reference frees them when it is closed) and leave the
        module registry; the object can be given weights again with load_weights()."""
        if self._engine is not None:
            self._engine.close()
            self._engine = None
        S.unregister(self)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    @property
    def engine(self):
        if self._engine is None:
            if self.weights is None:
                raise RuntimeError('encoder has no weights: restore a checkpoint (factory.restore_checkpoint) '
                                   'or call Encoder.load_weights first')
            from .engine import EncoderEngine
            self._engine = EncoderEngine(self.config, self.weights, device=self._device)
        return self._engine
