import sys, torch, time
sys.path.insert(0, "/root/repo")
from miniworld_amd.vec_env import MiniWorldVecEnv
for env_id, n in (("MiniWorld-Maze-v0", 1024), ("MiniWorld-PickupObjects-v0", 2048)):
    for ar in (True, False):
        vec = MiniWorldVecEnv(env_id, n, autoreset=ar, domain_rand=("Pickup" in env_id))
        vec.reset()
        g = torch.Generator(device="cuda").manual_seed(0)
        acts = torch.randint(0, 3, (220, n), generator=g, device="cuda", dtype=torch.int32)
        for t in range(20): vec.step(acts[t])
        vec.engine.kernel_time_ms()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for t in range(20, 220): vec.step(acts[t])
        torch.cuda.synchronize(); el = time.perf_counter() - t0
        r, s, k = vec.engine.kernel_time_ms()
        print(env_id, "autoreset", ar, "steps/s %.3g" % (n * 200 / el), "raster ms %.4f setup ms %.4f" % (r, s))
        vec.close()
