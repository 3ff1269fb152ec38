"""The few helpers of the reference's ``auto_pose/ae/utils.py`` that the inference path uses, under the
same names: the batch index generator, the workspace directory layout and a cached property.
(Golden-tested against the reference module's outputs: tests/test_golden_codebook_logic.py.)"""
from __future__ import annotations

import os


class lazy_property(object):
    """Non-data descriptor: computes the attribute on first access and stores it in the instance
    dict, so later reads never reach the descriptor again (same observable behaviour as the
    reference's decorator of that name, utils.py:7-18)."""

    def __init__(self, fget):
        self.fget = fget
        self.__name__ = fget.__name__
        self.__doc__ = fget.__doc__

    def __get__(self, obj, owner=None):
        if obj is None:
            return self
        value = obj.__dict__[self.__name__] = self.fget(obj)
        return value


def batch_iteration_indices(N, batch_size):
    """Yields (start, end) over range(N) in strides of batch_size, the last pair cut at N
    (utils.py:20-26; 92232 views at 64 per batch = 1441 full batches + one of 8)."""
    start = 0
    while start < N:
        stop = start + batch_size
        yield (start, stop if stop <= N else N)
        start = stop


# ---- $AE_WORKSPACE_PATH layout (utils.py:28-90): <ws>/experiments/<group>/<name>/{<name>.cfg, checkpoints/chkpt-*} ----
def _under(*parts):
    return os.path.join(*parts)


def get_dataset_path(workspace_path):
    return _under(workspace_path, 'tmp_datasets')


def get_log_dir(workspace_path, experiment_name, experiment_group=''):
    return _under(workspace_path, 'experiments', experiment_group, experiment_name)


def get_checkpoint_dir(log_dir):
    return _under(log_dir, 'checkpoints')


def get_checkpoint_basefilename(log_dir):
    return _under(get_checkpoint_dir(log_dir), 'chkpt')


def get_train_config_exp_file_path(log_dir, experiment_name):
    return _under(log_dir, experiment_name + '.cfg')


def get_config_file_path(workspace_path, experiment_name, experiment_group=''):
    return _under(workspace_path, 'cfg', experiment_group, experiment_name + '.cfg')
