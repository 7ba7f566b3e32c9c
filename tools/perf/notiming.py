"""Step throughput with and without the per-kernel HIP events of bench.py's roofline measurement."""
import sys, time, torch
sys.path.insert(0, "/root/repo")
from miniworld_amd.vec_env import MiniWorldVecEnv
n = 4096
vec = MiniWorldVecEnv("MiniWorld-Hallway-v0", n)
vec.reset()
g = torch.Generator(device="cuda").manual_seed(1234)
acts = torch.randint(0, 3, (420, n), generator=g, device="cuda", dtype=torch.int32)
for t in range(20): vec.step(acts[t])
for timing in (False, True, False):
    if timing: vec.engine.kernel_time_ms()
    else: vec.engine.kernel_time_ms(reset=-1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in range(20, 420): vec.step(acts[t])
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    print("events", timing, "steps/s %.4g  ms/step %.4f" % (n * 400 / el, 1e3 * el / 400))
