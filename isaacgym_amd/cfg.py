"""Default task configurations: the `env` / `sim` values of the reference's
cfg/task/{Cartpole,Ant,Humanoid,AllegroHand,HumanoidAMP}.yaml as plain dicts
(hydra is not a dependency).  `task_config(name, num_envs, **env_overrides)`
returns a deep copy."""
import copy

_SIM = dict(dt=0.0166, substeps=2, up_axis="z", use_gpu_pipeline=True, gravity=[0.0, 0.0, -9.81])

_TASKS = {
    "Cartpole": dict(
        env=dict(numEnvs=512, envSpacing=4.0, resetDist=3.0, maxEffort=400.0, clipObservations=5.0, clipActions=1.0,
                 enableCameraSensors=False),
        sim=_SIM, task=dict(randomize=False, randomization_params={})),
    "CartpoleSample": dict(
        env=dict(numEnvs=512, envSpacing=4.0, resetDist=3.0, maxEffort=400.0, clipObservations=5.0, clipActions=1.0,
                 enableCameraSensors=False, is_train=True, sample=False, max_frame_length=900, seed=42),
        sim=_SIM, task=dict(randomize=False, randomization_params={})),
    "Ant": dict(
        env=dict(numEnvs=4096, envSpacing=5, episodeLength=1000, enableDebugVis=False, clipActions=1.0, powerScale=1.0,
                 controlFrequencyInv=1, headingWeight=0.5, upWeight=0.1, actionsCost=0.005, energyCost=0.05,
                 dofVelocityScale=0.2, contactForceScale=0.1, jointsAtLimitCost=0.1, deathCost=-2.0,
                 terminationHeight=0.31, plane=dict(staticFriction=1.0, dynamicFriction=1.0, restitution=0.0),
                 enableCameraSensors=False),
        sim=_SIM, task=dict(randomize=False, randomization_params=dict(
            frequency=600, observations=dict(range=[0, .002], operation="additive", distribution="gaussian"),
            actions=dict(range=[0., .02], operation="additive", distribution="gaussian"), actor_params={}))),
    "AntImitate": dict(
        env=dict(numEnvs=4096, envSpacing=5, episodeLength=1000, enableDebugVis=False, clipActions=1.0, powerScale=1.0,
                 controlFrequencyInv=1, headingWeight=0.5, upWeight=0.1, actionsCost=0.005, energyCost=0.05,
                 dofVelocityScale=0.2, contactForceScale=0.1, jointsAtLimitCost=0.1, deathCost=-2.0,
                 terminationHeight=0.31, plane=dict(staticFriction=1.0, dynamicFriction=1.0, restitution=0.0),
                 enableCameraSensors=False, camera_outdir=""),
        sim=_SIM, task=dict(randomize=False, randomization_params={})),
    "Humanoid": dict(
        env=dict(numEnvs=4096, envSpacing=5, episodeLength=1000, enableDebugVis=False, clipActions=1.0, powerScale=1.0,
                 headingWeight=0.5, upWeight=0.1, actionsCost=0.01, energyCost=0.05, dofVelocityScale=0.1,
                 angularVelocityScale=0.25, contactForceScale=0.01, jointsAtLimitCost=0.25, deathCost=-1.0,
                 terminationHeight=0.8, plane=dict(staticFriction=1.0, dynamicFriction=1.0, restitution=0.0),
                 enableCameraSensors=False),
        sim=_SIM, task=dict(randomize=False, randomization_params={})),
    "HumanoidImitate": dict(
        env=dict(numEnvs=4096, envSpacing=5, episodeLength=1000, enableDebugVis=False, clipActions=1.0, powerScale=1.0,
                 headingWeight=0.5, upWeight=0.1, actionsCost=0.01, energyCost=0.05, dofVelocityScale=0.1,
                 angularVelocityScale=0.25, contactForceScale=0.01, jointsAtLimitCost=0.25, deathCost=-1.0,
                 terminationHeight=0.5, terminationRadius=1.0,
                 plane=dict(staticFriction=1.0, dynamicFriction=1.0, restitution=0.0), enableCameraSensors=False),
        sim=_SIM, task=dict(randomize=False, randomization_params={})),
    "AllegroHand": dict(
        env=dict(numEnvs=16384, envSpacing=0.75, episodeLength=600, enableDebugVis=False, aggregateMode=1,
                 clipObservations=5.0, clipActions=1.0, stiffnessScale=1.0, forceLimitScale=1.0,
                 useRelativeControl=False, dofSpeedScale=20.0, actionsMovingAverage=1.0, controlFrequencyInv=2,
                 startPositionNoise=0.01, startRotationNoise=0.0, resetPositionNoise=0.01, resetRotationNoise=0.0,
                 resetDofPosRandomInterval=0.2, resetDofVelRandomInterval=0.0, startObjectPoseDY=-0.19,
                 startObjectPoseDZ=0.06, forceScale=0.0, forceProbRange=[0.001, 0.1], forceDecay=0.99,
                 forceDecayInterval=0.08, distRewardScale=-10.0, rotRewardScale=1.0, rotEps=0.1,
                 actionPenaltyScale=-0.0002, reachGoalBonus=250, fallDistance=0.24, fallPenalty=0.0,
                 objectType="block", observationType="full_state", asymmetric_observations=False,
                 successTolerance=0.1, printNumSuccesses=False, maxConsecutiveSuccesses=0),
        sim=dict(_SIM, dt=0.01667), task=dict(randomize=False, randomization_params={})),
    "ShadowHand": dict(
        env=dict(numEnvs=16384, envSpacing=0.75, episodeLength=600, enableDebugVis=False, aggregateMode=1,
                 clipObservations=5.0, clipActions=1.0, stiffnessScale=1.0, forceLimitScale=1.0,
                 useRelativeControl=False, dofSpeedScale=20.0, actionsMovingAverage=1.0, controlFrequencyInv=1,
                 startPositionNoise=0.01, startRotationNoise=0.0, resetPositionNoise=0.01, resetRotationNoise=0.0,
                 resetDofPosRandomInterval=0.2, resetDofVelRandomInterval=0.0, forceScale=0.0,
                 forceProbRange=[0.001, 0.1], forceDecay=0.99, forceDecayInterval=0.08, distRewardScale=-10.0,
                 rotRewardScale=1.0, rotEps=0.1, actionPenaltyScale=-0.0002, reachGoalBonus=250, fallDistance=0.24,
                 fallPenalty=0.0, objectType="block", observationType="full_state", asymmetric_observations=False,
                 successTolerance=0.1, printNumSuccesses=False, maxConsecutiveSuccesses=0),
        sim=dict(_SIM, dt=0.01667),
        task=dict(randomize=False, randomization_params=dict(
            frequency=720,
            observations=dict(range=[0, .002], range_correlated=[0, .001], operation="additive", distribution="gaussian"),
            actions=dict(range=[0., .05], range_correlated=[0, .015], operation="additive", distribution="gaussian"),
            actor_params={}))),
    "HumanoidAMP": dict(
        env=dict(numEnvs=4096, envSpacing=5, episodeLength=300, cameraFollow=True, enableDebugVis=False,
                 pdControl=True, powerScale=1.0, controlFrequencyInv=2, stateInit="Random", hybridInitProb=0.5,
                 numAMPObsSteps=2, localRootObs=False, contactBodies=["right_foot", "left_foot"],
                 terminationHeight=0.5, enableEarlyTermination=True, motion_file="amp_humanoid_run.npy",
                 plane=dict(staticFriction=1.0, dynamicFriction=1.0, restitution=0.0)),
        sim=_SIM, task=dict(randomize=False, randomization_params={})),
}


def task_config(name, num_envs=None, **env_overrides):
    if name not in _TASKS:
        raise KeyError("no default configuration for task %r (have %s)" % (name, sorted(_TASKS)))
    cfg = copy.deepcopy(_TASKS[name])
    cfg["name"] = name
    cfg["physics_engine"] = "physx"
    if num_envs is not None:
        cfg["env"]["numEnvs"] = int(num_envs)
    cfg["env"].update(env_overrides)
    return cfg
