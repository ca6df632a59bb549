"""A whole VecTask.step recorded once into a HIP graph and replayed.

Every step entry point of libigx only enqueues kernels on the stream it is given, so a step
can be captured (torch.cuda.graph -> hipStreamBeginCapture) and replayed with one launch -
for callers whose loop is recorded as a whole.  As a speed-up of a single step it does not
pay on this runtime: a replay costs about 5 us per graph node, more than the eager host
path of a step at the reference's default 4 096 environments.

A recorded launch repeats its argument values, rng_offset included.  The reset draws stay
fresh through a device-resident counter that the graph itself advances by one per replay and
that the kernels add to rng_offset (`rng_offset_dev` of the parameter structs).

What a replay cannot do is run host code: tasks whose step depends on it are refused
(domain-randomisation schedules, host-side printing, the trajectory recorder).  Resets that
the reference does between steps on the host (HumanoidAMP's reset_done) can ride along as a
`prologue` when the task has a device-only form of them (HumanoidAMP.reset_flagged).
"""
import torch


class GraphedStep:
    def __init__(self, task, warmup=3, prologue=None):
        """Records `prologue()` (optional, e.g. task.reset_flagged) followed by task.step.  `warmup` eager
        rounds run first (code objects load, lazy tensors appear)."""
        if getattr(task, "randomize", False):
            raise ValueError("a replayed graph cannot run the host-side domain-randomisation schedule")
        if getattr(task, "print_success_stat", False):
            raise ValueError("printNumSuccesses prints from the host every step")
        if getattr(task, "_recorder", None) is not None:
            raise ValueError("the trajectory recorder counts its captures on the host")
        if torch.device(task.rl_device) != torch.device(task.device):
            raise ValueError("rl_device must be the simulation device")
        self.task = task
        dev = task.device
        # the kernels add this counter to rng_offset modulo 2^64
        task._rng_counter = torch.zeros(1, dtype=torch.int64, device=dev)
        self.actions = torch.zeros(task.num_envs, task.num_actions, dtype=torch.float32, device=dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                if prologue is not None:
                    prologue()
                task.step(self.actions)
        torch.cuda.current_stream(dev).wait_stream(side)
        # the graph advances the counter before the step: from -1 the first replay draws with exactly
        # the recorded offsets, the next one with those plus one, ...
        task._rng_counter.fill_(-1)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            task._rng_counter.add_(1)
            if prologue is not None:
                prologue()
            self.outputs = task.step(self.actions)
        task.control_steps -= 1  # recording enqueued nothing: that step has not happened

    def step(self, actions):
        """Same contract as task.step(actions).  The returned tensors are the graph's own: each call overwrites them."""
        self.actions.copy_(actions.reshape(self.actions.shape))
        self.graph.replay()
        self.task.control_steps += 1
        return self.outputs
