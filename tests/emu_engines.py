"""Engine doubles that run the emulated kernels (tests/emu/) behind the same interface as
augmentedautoencoder_amd.engine -- injected into Encoder/Codebook by CPU tests of the
reference-shaped Python API.  Not importable from the product package."""
import numpy as np
import torch

import emu_backend as eb


class EmuEncoderEngine(object):
    device = torch.device('cpu')

    def __init__(self, cfg, weights):
        self.cfg = cfg
        self._e = eb.EmuEncoder(weights, cfg)

    def encode(self, x):
        if torch.is_tensor(x):
            x = x.numpy()
        x = np.asarray(x)
        if x.ndim == 3:
            x = x[None]
        if x.dtype != np.uint8:
            x = x.astype(np.float32)
        if len(x) == 0:                        # same empty-batch contract as engine.EncoderEngine.encode
            return torch.empty((0, self.cfg.latent_space_size), dtype=torch.float32)
        return torch.from_numpy(self._e.forward(x))

    def encode_nn(self, codebook_engine, x, col_stride=1):
        if torch.is_tensor(x):
            x = x.numpy()
        x = np.asarray(x)
        if x.ndim == 3:
            x = x[None]
        if x.dtype != np.uint8:
            x = x.astype(np.float32)
        if len(x) == 0:
            return (torch.empty((0, self.cfg.latent_space_size), dtype=torch.float32), torch.empty((0, 1), dtype=torch.int64),
                    torch.empty((0, 1), dtype=torch.float32))
        z, idx, score = eb.encode_nn(self._e, codebook_engine._c, x, col_stride)
        return torch.from_numpy(z), torch.from_numpy(idx), torch.from_numpy(score)

    def settle(self):
        """engine.EncoderEngine.settle(): the double runs exact fp32 synchronously, nothing is ever pending"""
        return 0

    def encode_checked(self, x):
        return self.encode(x)

    def activation(self, layer):
        return torch.from_numpy(self._e.activation(layer))

    def close(self):
        self._e.close()


class EmuCodebookEngine(object):
    device = torch.device('cpu')

    def __init__(self, E):
        self._c = eb.EmuCodebook(E)

    def update(self, E):
        self._c.close()
        self._c = eb.EmuCodebook(E)

    def nn(self, z, topk=1, col_stride=1):
        z = z.numpy() if torch.is_tensor(z) else np.asarray(z)
        if len(z) == 0:
            return torch.empty((0, topk), dtype=torch.int64), torch.empty((0, topk), dtype=torch.float32)
        idx, score = self._c.nn(z, topk, col_stride)
        return torch.from_numpy(idx), torch.from_numpy(score)

    def similarity(self, z):
        z = z.numpy() if torch.is_tensor(z) else np.asarray(z)
        if len(z) == 0:
            return torch.empty((0, self._c.E.shape[0]), dtype=torch.float32)
        return torch.from_numpy(self._c.similarity(z))

    def l2_normalize(self, z):
        z = z.numpy() if torch.is_tensor(z) else np.asarray(z)
        if len(z) == 0:
            return torch.from_numpy(np.asarray(z, dtype=np.float32))
        return torch.from_numpy(eb.l2_normalize(z))

    def close(self):
        self._c.close()


class EmuDecoderEngine(object):
    device = torch.device('cpu')

    def __init__(self, cfg, weights):
        self.cfg = cfg
        self._d = eb.EmuDecoder(weights, cfg)

    def decode(self, z):
        z = z.numpy() if torch.is_tensor(z) else np.asarray(z)
        if z.ndim == 1:
            z = z[None]
        return torch.from_numpy(self._d.forward(z))

    def close(self):
        self._d.close()
