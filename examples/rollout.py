#!/usr/bin/env python3
"""A minimal rollout loop on the batched engine: a (random-weight) CNN policy picks actions from the
observation tensor the raster kernel has just written — observations, rewards and flags never leave the GPU.

    python examples/rollout.py --env MiniWorld-Hallway-v0 --envs 4096 --steps 200
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--env", default="MiniWorld-Hallway-v0")
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()

    import torch
    from miniworld_amd.vector import MiniWorldVectorEnv

    # obs_layout="cwh": the kernel stores uint8[N, 3, W, H] (the reference's PyTorchObsWrapper layout) directly
    envs = MiniWorldVectorEnv(args.env, args.envs, seed=args.seed, obs_layout="cwh")
    n_act = envs.single_action_space.n
    policy = torch.nn.Sequential(
        torch.nn.Conv2d(3, 16, 5, stride=2), torch.nn.ReLU(),
        torch.nn.Conv2d(16, 32, 5, stride=2), torch.nn.ReLU(),
        torch.nn.AdaptiveAvgPool2d(1), torch.nn.Flatten(), torch.nn.Linear(32, n_act),
    ).cuda().half()

    obs, _ = envs.reset(seed=args.seed)
    with torch.no_grad():           # warm-up: MIOpen picks its convolution kernels on the first calls
        for _ in range(5):
            policy(obs.half() / 255.0)
    episodes, returns = 0, torch.zeros(args.envs, device="cuda")
    finished_returns = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.no_grad():
        for _ in range(args.steps):
            logits = policy(obs.half() / 255.0)
            actions = torch.distributions.Categorical(logits=logits.float()).sample().to(torch.int32)
            obs, rew, term, trunc, _ = envs.step(actions)
            returns += rew
            done = term | trunc
            if done.any():
                finished_returns.append(returns[done].clone())
                episodes += int(done.sum())
                returns[done] = 0.0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    mean_ret = torch.cat(finished_returns).mean().item() if finished_returns else float("nan")
    print(f"{args.env}: {args.envs * args.steps / dt:,.0f} env-steps/s with the policy in the loop, "
          f"{episodes} episodes finished, mean return {mean_ret:.3f}")
    envs.close()


if __name__ == "__main__":
    main()
