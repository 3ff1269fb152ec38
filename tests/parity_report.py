"""Collects what the GPU parity tests measured -- index flips per tolerance, minimum top-2 gaps, per-layer errors in
two norms -- and writes it to gpurun_out/parity_report.json when the session ends (tests/conftest.py), so that a green
run also leaves its numbers behind (copied to profiles/ for the round).  Test infrastructure only."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_records = {}


def current_test():
    name = os.environ.get('PYTEST_CURRENT_TEST', '')
    return name.split('::')[-1].split(' ')[0] if name else 'unknown'


def record(kind, where, **values):
    _records.setdefault(kind, []).append(dict(where=where, test=current_test(), **values))


def dump(path=None):
    if not _records:
        return None
    path = path or os.path.join(ROOT, 'gpurun_out', 'parity_report.json')
    os.makedirs(os.path.dirname(path), exist_ok=True)
    summary = {}
    idx = _records.get('indices', [])
    if idx:
        summary['index_checks'] = len(idx)
        summary['queries_checked'] = sum(r['queries'] for r in idx)
        summary['flips_total'] = sum(r['flips'] for r in idx)
        summary['flips_with_gap_above_2e6'] = sum(r['flips_with_gap_above_2e6'] for r in idx)
        summary['min_gap_seen'] = min(r['min_gap'] for r in idx)
    lay = _records.get('layers', [])
    if lay:
        summary['layer_checks'] = len(lay)
        summary['worst_layer_max_rel'] = max(r['max_rel'] for r in lay)
        summary['worst_layer_l2_rel'] = max(r['l2_rel'] for r in lay)
    with open(path, 'w') as f:
        json.dump({'summary': summary, 'records': _records}, f, indent=1, sort_keys=True)
    return path
