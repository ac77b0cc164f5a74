"""cProfile of eager C3 train steps (host side): where the Python time of the launch-bound eager
path (what N > 1 DDP runs use) goes."""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.step_trace import build_train_step

model, step = build_train_step(torch.device("cuda:0"), 2, (1025, 2049), "bf16")
for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    step()
torch.cuda.synchronize()
print("eager ms/step %.1f" % ((time.perf_counter() - t0) / 5 * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
