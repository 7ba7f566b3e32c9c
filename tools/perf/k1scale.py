import sys, torch, time
sys.path.insert(0, "/root/repo")
from miniworld_amd.vec_env import MiniWorldVecEnv
for n in (64, 256, 1024, 4096, 16384):
    vec = MiniWorldVecEnv("MiniWorld-Hallway-v0", n, autoreset=False)
    vec.reset()
    g = torch.Generator(device="cuda").manual_seed(0)
    acts = torch.randint(0, 3, (60, n), generator=g, device="cuda", dtype=torch.int32)
    for t in range(10): vec.step(acts[t])
    vec.engine.kernel_time_ms()
    for t in range(10, 60): vec.step(acts[t])
    torch.cuda.synchronize()
    r, s, _ = vec.engine.kernel_time_ms()
    print("N", n, "raster ms %.4f setup ms %.4f" % (r, s), "us/env raster %.4f setup %.4f" % (1e3 * r / n, 1e3 * s / n))
    vec.close()
