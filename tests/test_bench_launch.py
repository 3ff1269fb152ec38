"""bench.py launch forms: the driver starts `python bench.py --gpus N ...` WITHOUT a launcher;
for N > 1 the script must become its own launcher (one rank per GPU under torch.distributed.run) or say, in one JSON
line and without a traceback, why it cannot."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')


def _run(argv, env_drop=('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'), timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in env_drop}
    p = subprocess.run([sys.executable, BENCH] + argv, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith('{')]
    return p.returncode, [json.loads(l) for l in lines], p.stderr.decode()


def test_plain_command_with_two_ranks_launches_itself_up_to_the_process_group():
    rc, out, err = _run(['--gpus', '2', '--dry-run-dist'])
    assert rc == 0, err[-2000:]
    assert len(out) == 1, out                                  # rank 0 only
    line = out[0]
    assert line['world_size'] == 2 and line['n_gpus'] == 2 and sorted(line['ranks_seen']) == [0, 1]
    assert line['launched_by'] == 'self'


def test_external_launcher_form_still_works():
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE')}
    p = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', '29611', BENCH, '--gpus', '2', '--dry-run-dist'], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [json.loads(l) for l in p.stdout.decode().splitlines() if l.startswith('{')]
    assert len(lines) == 1 and lines[0]['world_size'] == 2 and lines[0]['launched_by'] == 'external'


def test_more_ranks_than_gpus_is_one_json_line_with_an_error_and_no_traceback():
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    rc, out, err = _run(['--gpus', str(have + 2), '--steps', '1', '--warmup', '0'])
    assert rc == 2
    assert len(out) == 1 and out[0]['value'] is None and 'error' in out[0] and out[0]['visible_gpus'] == have
    assert out[0]['metric'].startswith('crops/sec')
    assert 'Traceback' not in err


@pytest.mark.gpu
def test_rccl_code_path_of_the_bench_on_one_gpu():
    """the N > 1 code path (RCCL process group, pair packing, all_gather, config4) as a subprocess at world size 1"""
    rc, out, err = _run(['--gpus', '1', '--force-dist', '--steps', '2', '--warmup', '1', '--no-extras', '--config4', '--no-split-precision',
                         '--no-cpu-baseline', '--profile-steps', '1'])
    assert rc == 0, err[-3000:]
    line = out[-1]
    assert line['rccl_world_size'] == 1 and line['n_gpus'] == 1
    assert line['config4']['answers_complete'] is True
    assert line['config4']['objects'] == 8 and line['config4']['objects_per_gpu'] == 8        # eight objects at every N
    assert line['value'] > 1000
