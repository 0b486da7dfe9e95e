"""Every kernel build over every batch size (-m gpu).  Which build a launch gets - the lone-wavefront one or the
large-batch one, 64- or 256-room blocks - is normally chosen from the batch size (ge_step.hip fill_args / create_impl); the
knobs GE_LOWOCC_ROOMS / GE_BLOCK_THREADS / GE_NT_LOADS force the choice.  They are read once per process, so every case runs its scenario
(tests/knob_worker.py: event trace, host-driven seats + batched injection, mixed batch in steady state - each against the
oracle, with single-turn and fused launches) in a fresh child process.  Without these the large-batch build's trace path
and the lone-wavefront mixed kernel with restart would only be covered at the sizes that pick them by default."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
NEVER_LOW, ALWAYS_LOW = {"GE_LOWOCC_ROOMS": "0"}, {"GE_LOWOCC_ROOMS": "99999999"}


@pytest.mark.parametrize("knobs,scenario,rooms", [
    (NEVER_LOW, "trace", 3000), (NEVER_LOW, "trace", 140001), (ALWAYS_LOW, "trace", 3000), (ALWAYS_LOW, "trace", 140001),
    (NEVER_LOW, "humans", 9000), (ALWAYS_LOW, "humans", 9000), (ALWAYS_LOW, "humans", 140001),
    (NEVER_LOW, "mixed", 2500), (ALWAYS_LOW, "mixed", 2500), (ALWAYS_LOW, "mixed", 70001),
    ({"GE_LOWOCC_ROOMS": "0", "GE_BLOCK_THREADS": "256"}, "mixed", 2500),
    ({"GE_LOWOCC_ROOMS": "99999999", "GE_BLOCK_THREADS": "128"}, "trace", 3000),
    # the streaming-load form of the large-batch single-turn Werewolf x 8 kernel (normally only for a state beyond the Infinity Cache) ...
    ({"GE_NT_LOADS": "1", "GE_LOWOCC_ROOMS": "0"}, "trace", 3000), ({"GE_NT_LOADS": "1"}, "humans", 140001),
    # ... and the plain-load forms of the other layouts' (normally for states of 50 - 290 MiB)
    ({"GE_NT_LOADS": "0", "GE_LOWOCC_ROOMS": "0"}, "trace", 3000), ({"GE_NT_LOADS": "0"}, "humans", 140001),
], ids=lambda v: "-".join(f"{k[3:].lower()}={x}" for k, x in v.items()) if isinstance(v, dict) else str(v))
def test_scenario_under_forced_launch_knobs(knobs, scenario, rooms):
    env = dict(os.environ, **knobs)
    env["PYTHONPATH"] = HERE + os.pathsep + os.path.dirname(HERE) + os.pathsep + env.get("PYTHONPATH", "")
    p = subprocess.run([sys.executable, os.path.join(HERE, "knob_worker.py"), scenario, str(rooms)], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert p.returncode == 0 and b"knob scenario ok" in p.stdout, p.stdout.decode(errors="replace")[-3000:]
