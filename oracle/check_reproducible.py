"""Regenerates every golden fixture from the reference's own Python into a scratch directory and compares the
arrays, byte for byte, with the files committed under tests/golden/ (the `meta` entry, which names library
versions, aside).  Test infrastructure, like everything under oracle/; needs the reference checkout, so it runs
where the fixtures are made, not on the GPU box.

    python oracle/check_reproducible.py [scratch directory]
"""
import glob
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NEED_EAGER = ["humanoid_imitate", "extract_positions", "imitate_extreme"]  # their generators ask for PYTORCH_JIT=0


def main():
    scratch = sys.argv[1] if len(sys.argv) > 1 else tempfile.mkdtemp(prefix="igx_golden_")
    os.makedirs(scratch, exist_ok=True)
    gen = [sys.executable, os.path.join(ROOT, "oracle", "gen_golden.py")]
    env = dict(os.environ, IGX_GOLDEN_OUT=scratch)
    with open(os.path.join(scratch, "generate.log"), "w") as log:
        subprocess.check_call(gen, env=env, stdout=log, stderr=subprocess.STDOUT, cwd=ROOT)
        subprocess.check_call(gen + NEED_EAGER, env=dict(env, PYTORCH_JIT="0"), stdout=log, stderr=subprocess.STDOUT, cwd=ROOT)
    committed = sorted(os.path.basename(p) for p in glob.glob(os.path.join(ROOT, "tests", "golden", "*.npz")))
    made = set(os.path.basename(p) for p in glob.glob(os.path.join(scratch, "*.npz")))
    bad = ["%s: not regenerated" % n for n in committed if n not in made] + \
          ["%s: generated but not committed" % n for n in sorted(made - set(committed))]
    for name in committed:
        if name not in made:
            continue
        a = np.load(os.path.join(ROOT, "tests", "golden", name), allow_pickle=False)
        b = np.load(os.path.join(scratch, name), allow_pickle=False)
        if set(a.files) != set(b.files):
            bad.append("%s: entries differ: %s" % (name, sorted(set(a.files) ^ set(b.files))))
            continue
        for k in a.files:
            if k != "meta" and (a[k].shape != b[k].shape or a[k].dtype != b[k].dtype or a[k].tobytes() != b[k].tobytes()):
                bad.append("%s[%s] differs" % (name, k))
    print("%d committed fixtures, %d regenerated, %d differences" % (len(committed), len(made), len(bad)))
    for line in bad:
        print("  " + line)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
