"""Builder functions with the names and signatures of the reference's factory module
(/root/reference/auto_pose/ae/ae_factory.py) for everything the inference path needs: dataset,
encoder, decoder, codebook, ``build_codebook_from_name`` and ``restore_checkpoint``.

The cfg keys are parsed once, by ``EncoderConfig.from_cfg`` / ``DecoderConfig.from_cfg``
(weights.py); the builders below only hand the parsed shapes to the API classes.  Builders of
the training graph (queue, autoencoder loss, train op) raise ``NotImplementedError``: training
is outside the scope of this package.  Configuration errors raise instead of ``exit()``-ing.
"""
from __future__ import annotations

import configparser
import os

from . import session as S
from . import utils as u
from .codebook import Codebook
from .dataset import Dataset
from .decoder import Decoder
from .encoder import Encoder
from .weights import DecoderConfig, EncoderConfig

_DATASET_SECTIONS = ('Dataset', 'Paths', 'Augmentation', 'Queue', 'Embedding')


def build_dataset(dataset_path, args):
    """ae_factory.py:5-15 -- every key of the dataset-related cfg sections becomes a keyword."""
    merged = {}
    for section in _DATASET_SECTIONS:
        if args.has_section(section):
            merged.update(args.items(section))
    return Dataset(dataset_path, **merged)


def build_encoder(x, args, is_training=False):
    """ae_factory.py:33-48."""
    net = EncoderConfig.from_cfg(args)
    return Encoder(x, net.latent_space_size, net.num_filter, net.kernel_size, net.strides, net.batch_norm,
                   is_training=is_training)


def build_decoder(reconstruction_target, encoder, args, is_training=False):
    """ae_factory.py:50-70, inference form (the variational sampling branch exists only while training):
    the decoder receives NUM_FILTER / STRIDES reversed and KERNEL_SIZE_DECODER."""
    net = DecoderConfig.from_cfg(args)
    return Decoder(reconstruction_target, encoder.z, net.num_filters, net.kernel_size, net.strides,
                   loss=args.get('Network', 'LOSS', fallback='L2'),
                   bootstrap_ratio=args.getint('Network', 'BOOTSTRAP_RATIO', fallback=1),
                   auxiliary_mask=net.auxiliary_mask, batch_norm=net.batch_norm,
                   is_training=is_training, encoder=encoder)


def build_codebook(encoder, dataset, args):
    """ae_factory.py:97-100."""
    return Codebook(encoder, dataset, args.getboolean('Embedding', 'EMBED_BB'))


def _training_only(name):
    def refuse(*unused_args, **unused_kw):
        raise NotImplementedError('%s builds the training graph, which is out of scope of this package' % name)
    refuse.__name__ = name
    return refuse


build_queue = _training_only('build_queue')
build_ae = _training_only('build_ae')
build_train_op = _training_only('build_train_op')


def _experiment_args(experiment_name, experiment_group):
    """(train cfg of the experiment, dataset path) below $AE_WORKSPACE_PATH."""
    workspace = os.environ.get('AE_WORKSPACE_PATH')
    if workspace is None:
        raise RuntimeError('Please define a workspace path: export AE_WORKSPACE_PATH=/path/to/workspace')
    log_dir = u.get_log_dir(workspace, experiment_name, experiment_group)
    cfg_path = u.get_train_config_exp_file_path(log_dir, experiment_name)
    if not os.path.exists(cfg_path):
        raise FileNotFoundError('Config File not found: %s' % cfg_path)
    args = configparser.ConfigParser()
    args.read(cfg_path)
    return args, u.get_dataset_path(workspace)


def build_codebook_from_name(experiment_name, experiment_group='', return_dataset=False, return_decoder=False):
    """Dataset + encoder + codebook (+ decoder) of a trained experiment, built under the experiment's
    variable scope (ae_factory.py:102-146).  Return shape as in the reference: codebook,
    (codebook, dataset) or (codebook, dataset, decoder)."""
    args, dataset_path = _experiment_args(experiment_name, experiment_group)
    decoder = None
    with S.variable_scope(experiment_name):
        dataset = build_dataset(dataset_path, args)
        encoder = build_encoder(S.Placeholder(dataset.shape, 'x'), args)
        codebook = build_codebook(encoder, dataset, args)
        if return_decoder:
            decoder = build_decoder(S.Placeholder(dataset.shape, 'reconst_target'), encoder, args)
    if not return_dataset:
        return codebook
    return (codebook, dataset, decoder) if return_decoder else (codebook, dataset)


def restore_checkpoint(session, saver, ckpt_dir, at_step=None):
    """Weights + codebook variables from the newest checkpoint in ``ckpt_dir``, or from every
    checkpoint whose name contains ``at_step`` (ae_factory.py:149-172).  ``saver=None`` restores all
    modules built so far.  Understands TensorFlow checkpoint-v2 files and the native .npz."""
    state = S.get_checkpoint_state(ckpt_dir)
    if not state or not state.model_checkpoint_path:
        raise FileNotFoundError('No checkpoint found. Expected one in: %s' % ckpt_dir)
    saver = S.Saver() if saver is None else saver
    if at_step is None:
        saver.restore(session, state.model_checkpoint_path)
        return
    for path in state.all_model_checkpoint_paths:
        if str(at_step) in str(path):
            saver.restore(session, path)
            print('restoring', os.path.basename(path))
