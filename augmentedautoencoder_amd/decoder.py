"""``Decoder`` with the constructor and the inference properties of
/root/reference/auto_pose/ae/decoder.py:12-84, backed by the HIP decoder engine
(aae_decoder_* in include/aae_hip.h) instead of a TF graph.

Reference call sites kept working (auto_pose/eval/eval_plots.py:24-34,37-72,75-80):
    sess.run(decoder.x, feed_dict={encoder.x: crops})            # encode, then reconstruct
    sess.run(decoder.x, feed_dict={decoder._latent_code: codes}) # reconstruct given codes
The training-only members (reconstr_loss, bootstrap loss, mask loss) are out of scope."""
from __future__ import annotations

import numpy as np

from . import session as S
from .weights import DecoderConfig


class Decoder(object):

    def __init__(self, reconstruction_target, latent_code, num_filters, kernel_size, strides, loss='L2',
                 bootstrap_ratio=1, auxiliary_mask=False, batch_norm=False, is_training=False, encoder=None):
        """num_filters / strides arrive REVERSED, as ae_factory.build_decoder passes them
        (ae_factory.py:59-70).  latent_code: the encoder's z fetchable; encoder: the Encoder it
        belongs to (found through latent_code when omitted)."""
        if is_training:
            raise NotImplementedError('training graphs are out of scope; is_training must be False')
        self._reconstruction_target = reconstruction_target
        self._latent_code = latent_code
        self._auxiliary_mask = bool(auxiliary_mask)
        self._num_filters = list(num_filters)
        self._kernel_size = int(kernel_size)
        self._strides = list(strides)
        self._loss = loss
        self._bootstrap_ratio = bootstrap_ratio
        self._batch_normalization = bool(batch_norm)
        self._is_training = False
        self._encoder = encoder if encoder is not None else getattr(latent_code, 'owner', None)
        shape = tuple(reconstruction_target.shape) if hasattr(reconstruction_target, 'shape') else tuple(reconstruction_target)
        latent = self._encoder.latent_space_size if self._encoder is not None else None
        if latent is None:
            raise ValueError('Decoder needs the encoder its latent code comes from (pass encoder=...)')
        self.config = DecoderConfig(shape[-3:], list(reversed(self._num_filters)), list(reversed(self._strides)),
                                    self._kernel_size, latent, self._batch_normalization, self._auxiliary_mask)
        self.weights = None
        self._engine = None
        self._device = None
        self._x_op = S.Op('decoder/x', self._run)
        S.register(decoder=self)

    @property
    def reconstruction_target(self):
        return self._reconstruction_target

    @property
    def x(self):
        """Fetchable reconstruction [B,H,W,C] float32 in [0,1] (decoder.py:36-84)."""
        return self._x_op

    # -- plumbing ------------------------------------------------------------------
    def load_weights(self, weights, device=None):
        """Stand-in for Saver.restore on the decoder variables (dense_1, conv2d_<L>.., BN)."""
        self.weights = {k: np.asarray(v) for k, v in weights.items()}
        self._device = device
        if self._engine is not None:
            self._engine.close()
            self._engine = None

    def close(self):
        """Free the device weights now and leave the module registry."""
        if self._engine is not None:
            self._engine.close()
            self._engine = None
        S.unregister(self)

    @property
    def engine(self):
        if self._engine is None:
            if self.weights is None:
                raise RuntimeError('decoder has no weights: restore a checkpoint (factory.restore_checkpoint) '
                                   'or call Decoder.load_weights first')
            from .engine import DecoderEngine
            self._engine = DecoderEngine(self.config, self.weights, device=self._device)
        return self._engine

    def _run(self, feed):
        for k, v in feed.items():
            if k is self._latent_code:
                return self.engine.decode(np.asarray(v, dtype=np.float32)).cpu().numpy()
        if self._encoder is None:
            raise ValueError('feed_dict has no value for decoder._latent_code')
        z = self._encoder.engine.encode_checked(self._encoder._feed(feed))  # stays on the device
        return self.engine.decode(z).cpu().numpy()
