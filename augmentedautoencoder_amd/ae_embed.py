"""``ae_embed <group/experiment> [--at_step N]``: rebuild the rotation codebook of
a trained experiment (/root/reference/auto_pose/ae/ae_embed.py:17-93) with the
HIP encoder.  The OpenGL renderer is out of scope, so the views to embed come
from ``--views file.npy`` ([N,H,W,C] uint8, e.g. pre-rendered with the
reference renderer; memory-mapped) or ``--synthetic`` (deterministic patterns,
for benchmarking)."""
from __future__ import annotations

import argparse
import configparser
import os

import numpy as np

from . import ae_factory as factory
from . import session as S
from . import utils as u
from .dataset import SyntheticViewSource


def main(argv=None):
    workspace_path = os.environ.get('AE_WORKSPACE_PATH')
    if workspace_path is None:
        raise SystemExit('Please define a workspace path:\nexport AE_WORKSPACE_PATH=/path/to/workspace')

    parser = argparse.ArgumentParser()
    parser.add_argument('experiment_name')
    parser.add_argument('--at_step', default=None, required=False)
    parser.add_argument('--views', default=None, help='[N,H,W,C] uint8 .npy of the views to embed (+ optional --bbs)')
    parser.add_argument('--bbs', default=None, help='[N,4] object bounding boxes of the rendered views (.npy)')
    parser.add_argument('--synthetic', action='store_true', help='embed deterministic synthetic views')
    arguments = parser.parse_args(argv)
    full_name = arguments.experiment_name.split('/')
    experiment_name = full_name.pop()
    experiment_group = full_name.pop() if len(full_name) > 0 else ''

    cfg_file_path = u.get_config_file_path(workspace_path, experiment_name, experiment_group)
    log_dir = u.get_log_dir(workspace_path, experiment_name, experiment_group)
    checkpoint_file = u.get_checkpoint_basefilename(log_dir)
    ckpt_dir = u.get_checkpoint_dir(log_dir)
    dataset_path = u.get_dataset_path(workspace_path)
    if not os.path.exists(cfg_file_path):
        raise SystemExit('Could not find config file:\n%s' % cfg_file_path)
    args = configparser.ConfigParser()
    args.read(cfg_file_path)

    with S.variable_scope(experiment_name):
        dataset = factory.build_dataset(dataset_path, args)
        encoder = factory.build_encoder(S.Placeholder(dataset.shape, 'x'), args)
        codebook = factory.build_codebook(encoder, dataset, args)
        saver = S.Saver(save_relative_paths=True, scope=experiment_name)

    if arguments.views:
        views = np.load(arguments.views, mmap_mode='r')
        bbs = np.load(arguments.bbs) if arguments.bbs else np.zeros((len(views), 4))
        if len(views) != dataset.embedding_size:
            raise SystemExit('%s holds %d views, the codebook needs %d' % (arguments.views, len(views), dataset.embedding_size))
        dataset.set_view_source(lambda a, e, Rs: (np.ascontiguousarray(views[a:e]), bbs[a:e]))
    elif arguments.synthetic:
        dataset.set_view_source(SyntheticViewSource(dataset.shape))
    else:
        raise SystemExit('rendering is out of scope: pass --views <file.npy> or --synthetic')

    batch_size = args.getint('Training', 'BATCH_SIZE')
    with S.Session() as sess:
        factory.restore_checkpoint(sess, saver, ckpt_dir, at_step=arguments.at_step)
        codebook.update_embedding(sess, batch_size)
        print('Saving new checkpoint ..')
        step = arguments.at_step if arguments.at_step is not None else 0
        chk = S.get_checkpoint_state(ckpt_dir)
        if arguments.at_step is None and chk is not None:
            tail = os.path.basename(chk.model_checkpoint_path).replace('.npz', '').split('-')[-1]
            step = int(tail) if tail.isdigit() else 0
        saver.save(sess, checkpoint_file, global_step=int(step))
        print('done')


if __name__ == '__main__':
    main()
