"""Helpers the hot path needs from /root/reference/auto_pose/ae/utils.py
(batch iteration, workspace path layout, lazy_property)."""
from __future__ import annotations

import functools
import os

import numpy as np


def lazy_property(function):
    attribute = '_cache_' + function.__name__

    @property
    @functools.wraps(function)
    def decorator(self):
        if not hasattr(self, attribute):
            setattr(self, attribute, function(self))
        return getattr(self, attribute)

    return decorator


def batch_iteration_indices(N, batch_size):
    """(start, end) pairs covering range(N) in steps of batch_size; the last
    batch is short (utils.py:20-26: 92232 = 1441*64 + 8)."""
    n_batches = int(np.ceil(float(N) / float(batch_size)))
    for i in range(n_batches):
        a = i * batch_size
        yield (a, min(a + batch_size, N))


# workspace layout (utils.py:28-90)
def get_dataset_path(workspace_path):
    return os.path.join(workspace_path, 'tmp_datasets')


def get_checkpoint_dir(log_dir):
    return os.path.join(log_dir, 'checkpoints')


def get_log_dir(workspace_path, experiment_name, experiment_group=''):
    return os.path.join(workspace_path, 'experiments', experiment_group, experiment_name)


def get_train_config_exp_file_path(log_dir, experiment_name):
    return os.path.join(log_dir, '{}.cfg'.format(experiment_name))


def get_checkpoint_basefilename(log_dir):
    return os.path.join(log_dir, 'checkpoints', 'chkpt')


def get_config_file_path(workspace_path, experiment_name, experiment_group=''):
    return os.path.join(workspace_path, 'cfg', experiment_group, '{}.cfg'.format(experiment_name))
