"""``ae_embed <group/experiment> [--at_step N]`` -- rebuild the rotation codebook of a trained
experiment with the HIP encoder (the reference's command: /root/reference/auto_pose/ae/ae_embed.py:17-93;
BASELINE config 3).

The reference renders the 92 232 views with its OpenGL renderer while embedding; rendering is out
of scope here, so the views come from ``--views file.npy`` ([N,H,W,C] uint8, memory-mapped -- e.g.
rendered once with the reference tooling) plus optional ``--bbs file.npy``, or ``--synthetic``
(deterministic patterns, for benchmarking the encoder-only throughput)."""
from __future__ import annotations

import argparse
import configparser
import os

import numpy as np

from . import ae_factory as factory
from . import session as S
from . import utils as u
from .dataset import SyntheticViewSource


def _parse(argv):
    ap = argparse.ArgumentParser(prog='ae_embed')
    ap.add_argument('experiment_name', help='<group>/<name> or <name>')
    ap.add_argument('--at_step', default=None)
    ap.add_argument('--views', default=None, help='[N,H,W,C] uint8 .npy holding the views to embed')
    ap.add_argument('--bbs', default=None, help='[N,4] object bounding boxes of those views (.npy)')
    ap.add_argument('--synthetic', action='store_true', help='embed deterministic synthetic views')
    return ap.parse_args(argv)


def _attach_views(dataset, opts):
    if opts.views:
        views = np.load(opts.views, mmap_mode='r')
        if len(views) != dataset.embedding_size:
            raise SystemExit('%s holds %d views, the codebook needs %d' % (opts.views, len(views), dataset.embedding_size))
        boxes = np.load(opts.bbs) if opts.bbs else np.zeros((len(views), 4))
        dataset.set_view_source(lambda a, e, Rs: (np.ascontiguousarray(views[a:e]), boxes[a:e]))
    elif opts.synthetic:
        dataset.set_view_source(SyntheticViewSource(dataset.shape))
    else:
        raise SystemExit('rendering is out of scope: pass --views <file.npy> or --synthetic')


def _step_of(path):
    tail = os.path.basename(path).replace('.npz', '').rsplit('-', 1)[-1]
    return int(tail) if tail.isdigit() else 0


def main(argv=None):
    workspace = os.environ.get('AE_WORKSPACE_PATH')
    if workspace is None:
        raise SystemExit('Please define a workspace path:\nexport AE_WORKSPACE_PATH=/path/to/workspace')
    opts = _parse(argv)
    group, _, name = opts.experiment_name.rpartition('/')
    group = group.rsplit('/', 1)[-1]

    cfg_path = u.get_config_file_path(workspace, name, group)
    if not os.path.exists(cfg_path):
        raise SystemExit('Could not find config file:\n%s' % cfg_path)
    args = configparser.ConfigParser()
    args.read(cfg_path)
    log_dir = u.get_log_dir(workspace, name, group)
    ckpt_dir = u.get_checkpoint_dir(log_dir)

    with S.variable_scope(name):
        dataset = factory.build_dataset(u.get_dataset_path(workspace), args)
        encoder = factory.build_encoder(S.Placeholder(dataset.shape, 'x'), args)
        codebook = factory.build_codebook(encoder, dataset, args)
        saver = S.Saver(save_relative_paths=True, scope=name)
    _attach_views(dataset, opts)

    with S.Session() as sess:
        factory.restore_checkpoint(sess, saver, ckpt_dir, at_step=opts.at_step)
        codebook.update_embedding(sess, args.getint('Training', 'BATCH_SIZE'))
        print('Saving new checkpoint ..')
        if opts.at_step is not None:
            step = int(opts.at_step)
        else:
            state = S.get_checkpoint_state(ckpt_dir)
            step = _step_of(state.model_checkpoint_path) if state is not None else 0
        saver.save(sess, u.get_checkpoint_basefilename(log_dir), global_step=step)
        print('done')


if __name__ == '__main__':
    main()
