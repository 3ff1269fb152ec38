"""Builders with the signatures of /root/reference/auto_pose/ae/ae_factory.py
for the inference path (dataset, encoder, codebook, decoder, build_codebook_from_name,
restore_checkpoint).  Training builders (ae, train_op, queue) are out of scope and raise
NotImplementedError."""
from __future__ import annotations

import ast
import configparser
import os

from . import session as S
from . import utils as u
from .codebook import Codebook
from .dataset import Dataset
from .decoder import Decoder
from .encoder import Encoder


def _section_items(args, name):
    return args.items(name) if args.has_section(name) else []


def build_dataset(dataset_path, args):
    dataset_args = {k: v for k, v in
                    _section_items(args, 'Dataset') + _section_items(args, 'Paths') +
                    _section_items(args, 'Augmentation') + _section_items(args, 'Queue') +
                    _section_items(args, 'Embedding')}
    return Dataset(dataset_path, **dataset_args)


def build_encoder(x, args, is_training=False):
    LATENT_SPACE_SIZE = args.getint('Network', 'LATENT_SPACE_SIZE')
    NUM_FILTER = ast.literal_eval(args.get('Network', 'NUM_FILTER'))
    KERNEL_SIZE_ENCODER = args.getint('Network', 'KERNEL_SIZE_ENCODER')
    STRIDES = ast.literal_eval(args.get('Network', 'STRIDES'))
    BATCH_NORM = args.getboolean('Network', 'BATCH_NORMALIZATION')
    return Encoder(x, LATENT_SPACE_SIZE, NUM_FILTER, KERNEL_SIZE_ENCODER, STRIDES, BATCH_NORM, is_training=is_training)


def build_codebook(encoder, dataset, args):
    embed_bb = args.getboolean('Embedding', 'EMBED_BB')
    return Codebook(encoder, dataset, embed_bb)


def _out_of_scope(name):
    def fn(*a, **k):
        raise NotImplementedError('%s builds the training graph, which is out of scope of this package' % name)
    fn.__name__ = name
    return fn


build_queue = _out_of_scope('build_queue')


def build_decoder(reconstruction_target, encoder, args, is_training=False):
    """ae_factory.py:50-70, inference form: reversed NUM_FILTER / STRIDES, KERNEL_SIZE_DECODER;
    the variational sampling branch only exists while training."""
    NUM_FILTER = ast.literal_eval(args.get('Network', 'NUM_FILTER'))
    KERNEL_SIZE_DECODER = args.getint('Network', 'KERNEL_SIZE_DECODER')
    STRIDES = ast.literal_eval(args.get('Network', 'STRIDES'))
    LOSS = args.get('Network', 'LOSS', fallback='L2')
    BOOTSTRAP_RATIO = args.getint('Network', 'BOOTSTRAP_RATIO', fallback=1)
    AUXILIARY_MASK = args.getboolean('Network', 'AUXILIARY_MASK', fallback=False)
    BATCH_NORM = args.getboolean('Network', 'BATCH_NORMALIZATION')
    return Decoder(reconstruction_target, encoder.z, list(reversed(NUM_FILTER)), KERNEL_SIZE_DECODER,
                   list(reversed(STRIDES)), LOSS, BOOTSTRAP_RATIO, AUXILIARY_MASK, BATCH_NORM,
                   is_training=is_training, encoder=encoder)


build_ae = _out_of_scope('build_ae')
build_train_op = _out_of_scope('build_train_op')


def build_codebook_from_name(experiment_name, experiment_group='', return_dataset=False, return_decoder=False):
    """Encoder + codebook of a trained experiment under $AE_WORKSPACE_PATH
    (ae_factory.py:102-146).  Raises instead of exit()ing on missing paths."""
    workspace_path = os.environ.get('AE_WORKSPACE_PATH')
    if workspace_path is None:
        raise RuntimeError('Please define a workspace path: export AE_WORKSPACE_PATH=/path/to/workspace')
    log_dir = u.get_log_dir(workspace_path, experiment_name, experiment_group)
    cfg_file_path = u.get_train_config_exp_file_path(log_dir, experiment_name)
    dataset_path = u.get_dataset_path(workspace_path)
    if not os.path.exists(cfg_file_path):
        raise FileNotFoundError('Config File not found: %s' % cfg_file_path)
    args = configparser.ConfigParser()
    args.read(cfg_file_path)

    with S.variable_scope(experiment_name):
        dataset = build_dataset(dataset_path, args)
        x = S.Placeholder(dataset.shape, 'x')
        encoder = build_encoder(x, args)
        codebook = build_codebook(encoder, dataset, args)
        decoder = build_decoder(S.Placeholder(dataset.shape, 'reconst_target'), encoder, args) if return_decoder else None

    if return_dataset:
        if return_decoder:
            return codebook, dataset, decoder
        return codebook, dataset
    return codebook


def restore_checkpoint(session, saver, ckpt_dir, at_step=None):
    """Load encoder weights + codebook from ckpt_dir (ae_factory.py:149-172).
    `saver` may be None (then every module built so far is restored)."""
    saver = saver if saver is not None else S.Saver()
    chkpt = S.get_checkpoint_state(ckpt_dir)
    if chkpt and chkpt.model_checkpoint_path:
        if at_step is None:
            saver.restore(session, chkpt.model_checkpoint_path)
        else:
            for ckpt_path in chkpt.all_model_checkpoint_paths:
                if str(at_step) in str(ckpt_path):
                    saver.restore(session, ckpt_path)
                    print('restoring', os.path.basename(ckpt_path))
    else:
        raise FileNotFoundError('No checkpoint found. Expected one in: %s' % ckpt_dir)
