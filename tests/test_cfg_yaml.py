"""The built-in defaults of isaacgym_amd.cfg against the reference's task files they were taken from (where the
reference tree is at hand).  The files are hydra documents; only their literal values are compared, entries that
are `${...}` interpolations are skipped (hydra is not a dependency and is not re-implemented)."""
import os

import pytest
import yaml

from isaacgym_amd import cfg as C

REF = "/root/reference/IsaacGymEnvs/isaacgymenvs/cfg/task"


def literal(v):
    return not (isinstance(v, str) and "${" in v)


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is not on this machine")
@pytest.mark.parametrize("name", ["Cartpole", "CartpoleSample", "Ant", "AntImitate", "Humanoid", "HumanoidImitate", "AllegroHand",
                                  "ShadowHand", "HumanoidAMP"])
def test_builtin_defaults_equal_the_reference_task_files(name):
    """Every `env` / `sim` value the package ships as a default is the one in the reference's own file."""
    with open(os.path.join(REF, name + ".yaml")) as f:
        ref = yaml.safe_load(f)
    mine = C.task_config(name)
    compared = 0
    for section in ("env", "sim"):
        for k, v in mine[section].items():
            if k not in ref[section] or not literal(ref[section][k]):
                continue  # a knob of this package, a value the reference hard-codes, or an interpolation
            want = ref[section][k]
            if isinstance(v, dict):
                for kk, vv in v.items():
                    assert want[kk] == vv, (name, section, k, kk, want[kk], vv)
            else:
                assert want == v or (isinstance(want, str) and os.path.basename(want) == v), (name, section, k, want, v)
            compared += 1
    # numEnvs is `${resolve_default:<literal>,${...num_envs}}` in every file: the literal is the default
    spelled = str(ref["env"]["numEnvs"])
    assert spelled.startswith("${resolve_default:%d," % mine["env"]["numEnvs"]), (spelled, mine["env"]["numEnvs"])
    # (CartpoleSample: is_train / sample / max_frame_length / seed come from the command line, not from the file)
    assert compared >= 0.5 * len(mine["env"]), compared


def test_task_config_returns_copies_and_takes_overrides():
    a = C.task_config("Ant", 64, deathCost=-3.0)
    b = C.task_config("Ant")
    assert a["env"]["numEnvs"] == 64 and a["env"]["deathCost"] == -3.0 and b["env"]["deathCost"] == -2.0
    a["sim"]["gravity"][2] = 0.0
    assert C.task_config("Ant")["sim"]["gravity"][2] == -9.81
    with pytest.raises(KeyError):
        C.task_config("NoSuchTask")
