"""The library's tuning switches (environment variables read by csrc/*.hip, DESIGN.md sections 4c, 7, 9b) select HOW a sync
does its work, never WHAT comes out: one process per setting runs the same syncs (tests/switch_worker.py) and every
setting must produce the same digest of everything a client can see."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SETTINGS = [
    {},
    {"CSTONE_D2H_BLIT": "1", "CSTONE_UPLOAD_BLIT": "1"},  # the runtime's copies instead of the copy kernels
    {"CSTONE_NO_GATHER_OVERLAP": "1", "CSTONE_MR_NO_PLACE_OVERLAP": "1"},  # one stream
    {"CSTONE_DEVICE_GLOBAL_STEP": "1", "CSTONE_MR_DEVICE_GLOBAL_STEP": "1", "CSTONE_MR_NO_SPECULATIVE_CUTS": "1"},
    {"CSTONE_FUSED_LEAF_PASS": "1"},  # the field-carrying leaf pass (four scratch arrays)
    {"CSTONE_NO_RESORT": "1"},  # no incremental re-sort: radix passes
]


@pytest.mark.gpu
def test_switches_change_nothing_a_client_can_see():
    seen = {}
    for env_extra in SETTINGS:
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
            env.pop(k, None)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "switch_worker.py")], capture_output=True,
                           text=True, timeout=600, cwd=ROOT, env=env)
        assert r.returncode == 0, (env_extra, r.stderr[-3000:])
        m = re.search(r"DIGEST ([0-9a-f]{64}) resorts (\d+) (\d+)", r.stdout)
        assert m, r.stdout[-2000:]
        seen[tuple(sorted(env_extra.items()))] = (m.group(1), int(m.group(2)), int(m.group(3)))
    digests = {v[0] for v in seen.values()}
    assert len(digests) == 1, seen
    # the default really took the incremental path in both domains, the last setting really did not
    assert seen[()][1] >= 1 and seen[()][2] >= 1
    assert seen[(("CSTONE_NO_RESORT", "1"),)][1:] == (0, 0)
