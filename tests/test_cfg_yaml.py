"""The reader of the reference's task files (hydra-style YAML) and, where the reference tree is at hand, the built-in
defaults of isaacgym_amd.cfg against the files they were taken from."""
import os

import pytest

from isaacgym_amd import cfg as C

SAMPLE = """
name: Walker
physics_engine: ${..physics_engine}
env:
  numEnvs: ${resolve_default:4096,${...num_envs}}
  episodeLength: 1000
  clipActions: 1.0
  deathCost: -2.0
  plane: {staticFriction: 1.0}
  asset:
    assetFileName: "mjcf/nv_ant.xml"
sim:
  dt: 0.0166 # 1/60 s
  use_gpu_pipeline: ${eq:${...pipeline},"gpu"}
  gravity: [0.0, 0.0, -9.81]
  physx:
    num_threads: ${....num_threads}
    use_gpu: ${contains:"cuda",${....sim_device}} # set to False to run on CPU
    contact_collection: ${if:${...test},0,2}
task:
  randomize: False
"""


def test_reads_hydra_style_task_files(tmp_path):
    p = tmp_path / "Walker.yaml"
    p.write_text(SAMPLE)
    d = C.load_task_yaml(str(p))
    assert d["name"] == "Walker" and d["physics_engine"] == "physx"
    assert d["env"]["numEnvs"] == 4096 and isinstance(d["env"]["numEnvs"], int)
    assert d["env"]["deathCost"] == -2.0 and d["env"]["asset"]["assetFileName"] == "mjcf/nv_ant.xml"
    assert d["sim"]["use_gpu_pipeline"] is True and d["sim"]["physx"]["use_gpu"] is True
    assert d["sim"]["physx"]["num_threads"] == 4 and d["sim"]["physx"]["contact_collection"] == 2
    assert d["sim"]["gravity"] == [0.0, 0.0, -9.81] and d["task"]["randomize"] is False
    e = C.load_task_yaml(str(p), num_envs=64, pipeline="cpu", sim_device="cpu", num_threads=2, test=True)
    assert e["env"]["numEnvs"] == 64 and e["sim"]["use_gpu_pipeline"] is False and e["sim"]["physx"]["use_gpu"] is False
    assert e["sim"]["physx"]["num_threads"] == 2 and e["sim"]["physx"]["contact_collection"] == 0


def test_unknown_pieces_are_errors(tmp_path):
    p = tmp_path / "x.yaml"
    p.write_text("env:\n  a: ${..no_such_setting}\n")
    with pytest.raises(KeyError, match="no_such_setting"):
        C.load_task_yaml(str(p))
    p.write_text("env:\n  a: ${shout:1,2}\n")
    with pytest.raises(ValueError, match="shout"):
        C.load_task_yaml(str(p))


REF = "/root/reference/IsaacGymEnvs/isaacgymenvs/cfg/task"


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is not on this machine")
@pytest.mark.parametrize("name", ["Cartpole", "CartpoleSample", "Ant", "AntImitate", "Humanoid", "HumanoidImitate", "AllegroHand",
                                  "ShadowHand", "HumanoidAMP"])
def test_builtin_defaults_equal_the_reference_task_files(name):
    """Every `env` / `sim` value the package ships as a default is the one in the reference's own file."""
    ref = C.load_task_yaml(os.path.join(REF, name + ".yaml"))
    mine = C.task_config(name)
    for section in ("env", "sim"):
        for k, v in mine[section].items():
            if k not in ref[section]:
                continue  # a knob of this package (e.g. quantileCollectives) or a value the reference hard-codes
            want = ref[section][k]
            if isinstance(v, dict):
                for kk, vv in v.items():
                    assert want[kk] == vv, (name, section, k, kk, want[kk], vv)
            else:
                assert want == v or (isinstance(want, str) and os.path.basename(want) == v), (name, section, k, want, v)
    assert ref["env"]["numEnvs"] == mine["env"]["numEnvs"]
    shared = set(mine["env"]) & set(ref["env"])
    # (CartpoleSample: is_train / sample / max_frame_length / seed come from the command line, not from the file)
    assert len(shared) >= 0.6 * len(mine["env"]), sorted(set(mine["env"]) - set(ref["env"]))
