"""Host logic of the task classes that needs no GPU."""
import numpy as np
import pytest

from isaacgym_amd import cfg, synthetic
from isaacgym_amd.sim import TensorSim
from isaacgym_amd.tasks import Ant, isaacgym_task_map
from isaacgym_amd.tasks.allegro_hand import success_threshold_c


def test_success_threshold_constants():
    # the values the reference recordings were decided with
    assert success_threshold_c(0.1) == synthetic.ALLEGRO_SUCCESS_C[0.1]
    assert success_threshold_c(0.2) == synthetic.ALLEGRO_SUCCESS_C[0.2]
    assert success_threshold_c(4.0) == 1.0


def test_default_configs_cover_the_task_map():
    for name in isaacgym_task_map:
        c = cfg.task_config(name, 64)
        assert c["env"]["numEnvs"] == 64 and c["name"] == name
    assert cfg.task_config("Ant")["env"]["episodeLength"] == 1000
    assert cfg.task_config("HumanoidAMP", numAMPObsSteps=3)["env"]["numAMPObsSteps"] == 3
    c = cfg.task_config("Cartpole")
    c["env"]["resetDist"] = 9.0
    assert cfg.task_config("Cartpole")["env"]["resetDist"] == 3.0  # deep copies
    with pytest.raises(KeyError):
        cfg.task_config("Trifinger")


def test_tasks_refuse_a_cpu_pipeline():
    c = cfg.task_config("Ant", 8)
    with pytest.raises(ValueError, match="GPU only"):
        Ant(c, "cpu", "cpu", sim=TensorSim(8, "cpu"))
    with pytest.raises(ValueError, match="sim object"):
        Ant(c, "cuda:0", "cuda:0")
