"""block_ticket_arrive (csrc/device_intrinsics.h) restated on std::atomic and stressed with host threads standing in
for the blocks of a launch: clean words, garbage, leftovers of a dead launch; single- and two-level tickets.  The GPU
memory-model side (sc1 stores / loads around the ticket) is covered by the -m gpu race-screen tests."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def test_ticket_protocol_counts_every_arrival_once_and_never_spins_forever():
    src = os.path.join(HERE, 'native', 'ticket_stress.cpp')
    exe = os.path.join(HERE, 'native', 'ticket_stress')
    subprocess.check_call(['g++', '-O2', '-std=c++17', '-pthread', src, '-o', exe])
    out = subprocess.run([exe, '300'], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.startswith('OK'), out.stdout + out.stderr
