"""The C ABI from plain C++ hosts (no Python, no torch in the process): tests/native/cabi_ant_step allocates with
hipMalloc, launches on its own stream and checks igx_ant_step against the CPU checker; tests/native/cabi_amp_fetch
uploads motion tables, runs whole demo fetches (draws + batch in one launch) and streams every batch to pinned host
memory through a whole-row recorder."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
EXE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "native", "cabi_ant_step")


@pytest.mark.parametrize("n", [1, 63, 100003])
def test_native_host_program(n):
    if not os.path.exists(EXE):
        pytest.fail("tests/native/cabi_ant_step is not built (python -c 'import __graft_entry__ as g; g.build()')")
    r = subprocess.run([EXE, str(n)], capture_output=True, text=True, timeout=100)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "cabi ok" in r.stdout


FETCH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "native", "cabi_amp_fetch")


@pytest.mark.parametrize("n,fetches", [(1, 2), (512, 6), (4099, 3)])
def test_native_imitation_sampling_program(n, fetches):
    if not os.path.exists(FETCH):
        pytest.fail("tests/native/cabi_amp_fetch is not built (python -c 'import __graft_entry__ as g; g.build()')")
    r = subprocess.run([FETCH, str(n), str(fetches)], capture_output=True, text=True, timeout=100)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "cabi amp fetch ok" in r.stdout
