"""bench.py's watchdog (the N > 1 runs: spanning the devices, every exchange, every barrier run under it) on the CPU: a phase that does not
come back ends the process with status 3 and one JSON object of diagnostics on stderr; phases that do come back leave a record."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(code):
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60, cwd=ROOT)


def test_a_phase_that_hangs_ends_the_process_with_its_diagnostics():
    p = _run("import time, bench\n"
             "dog = bench.Watchdog(lambda: {'rccl_version': 22203, 'peer_access': [[True, False], [False, True]]})\n"
             "with dog.phase('warm-up', 5.0):\n    time.sleep(0.05)\n"
             "with dog.phase('exchange #1 onto GPU 0 (gather)', 0.6, bytes_per_peer=4096, peers=7):\n    time.sleep(30)\n"
             "print('not reached')\n")
    assert p.returncode == 3 and "not reached" not in p.stdout
    lines = p.stderr.strip().splitlines()
    assert "did not return within 0.6 s" in lines[-2]
    diag = json.loads(lines[-1])["multi_gpu"]["diagnostics"]
    assert diag["stuck_in"] == "exchange #1 onto GPU 0 (gather)" and diag["bytes_per_peer"] == 4096 and diag["peers"] == 7
    assert diag["rccl_version"] == 22203 and diag["peer_access"] == [[True, False], [False, True]]
    assert diag["completed_before"][0][0] == "warm-up" and 0.6 <= diag["elapsed_s"] < 5


def test_broken_diagnostics_do_not_keep_the_process_alive():
    p = _run("import time, bench\n"
             "def facts():\n    raise RuntimeError('no device')\n"
             "dog = bench.Watchdog(facts)\n"
             "with dog.phase('dist.barrier', 0.4):\n    time.sleep(30)\n")
    assert p.returncode == 3
    diag = json.loads(p.stderr.strip().splitlines()[-1])["multi_gpu"]["diagnostics"]
    assert diag["stuck_in"] == "dist.barrier" and "no device" in diag["facts_error"]


def test_phases_that_return_are_recorded_and_nothing_fires():
    p = _run("import time, bench\n"
             "dog = bench.Watchdog(lambda: {})\n"
             "for i in range(3):\n    with dog.phase('exchange #%d' % i, 0.5):\n        time.sleep(0.02)\n"
             "time.sleep(0.8)\n"
             "print([n for n, _ in dog.history])\n")
    assert p.returncode == 0 and p.stdout.strip() == "['exchange #0', 'exchange #1', 'exchange #2']" and not p.stderr.strip()
