import cProfile, pstats, torch
from isaacgym_amd import tasks
t = tasks.make("Ant", 4096, "cuda:0")
a = torch.zeros(4096, 8, device="cuda:0")
for _ in range(20): t.step(a)
pr = cProfile.Profile(); pr.enable()
for _ in range(500): t.step(a)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
