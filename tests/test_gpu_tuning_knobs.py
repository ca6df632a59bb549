"""The launch-shape knobs the library reads once per process (IGX_ANT_TILE, IGX_HUMANOID_TILE, IGX_ALLEGRO_TILE and the
LDS reservations that cap the waves per CU) select other instantiations of the step kernels.  They exist for tuning
runs, but a user can set them, so the parity files of those kernels are run once more in a child process with every
knob off its default."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

KNOBS = {"IGX_ANT_TILE": "32", "IGX_HUMANOID_TILE": "64", "IGX_ALLEGRO_TILE": "32", "IGX_ANT_LDS_RESERVE": "0",
         "IGX_HUMANOID_LDS_RESERVE": "9000", "IGX_ALLEGRO_LDS_RESERVE": "20000", "IGX_AMP_LDS_RESERVE": "9000",
         "IGX_SHADOW_LDS_RESERVE": "7000", "IGX_HIM_LDS_RESERVE": "5000"}
FILES = ["tests/test_gpu_parity_loco.py", "tests/test_gpu_parity_allegro.py", "tests/test_gpu_parity_shadow.py",
         "tests/test_gpu_parity_amp.py", "tests/test_gpu_humanoid_imitate.py", "tests/test_gpu_imitate.py"]


def test_parity_files_with_every_knob_off_its_default():
    if any(k in os.environ for k in KNOBS):
        pytest.skip("this process already runs with knobs set")
    env = dict(os.environ, **KNOBS)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", "--timeout", "150",
                        "--timeout-method", "thread"] + FILES, cwd=ROOT, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    assert " passed" in r.stdout and "failed" not in r.stdout.splitlines()[-1], r.stdout[-500:]
