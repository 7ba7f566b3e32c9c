"""The real Gymnasium surface (reference: tests/test_miniworld.py:136-150, envs/__init__.py:44-157): runs wherever `gymnasium`
is installed beside the package — this image has none (gymshim.py stands in, tests/test_host_logic_cpu.py covers that), so
here and on the GPU box these tests are skipped."""
import numpy as np
import pytest

gymnasium = pytest.importorskip("gymnasium")
pytestmark = pytest.mark.gpu


def test_make_through_the_registry_and_check_env():
    """gymnasium.make over the package's own registration (the reference's 23 ids), then gymnasium's env checker."""
    import miniworld_amd  # noqa: F401  (registers the ids)
    from gymnasium.utils.env_checker import check_env
    from miniworld_amd.envs import ENV_IDS
    for env_id in ENV_IDS:
        assert env_id in gymnasium.registry
    env = gymnasium.make("MiniWorld-Hallway-v0")
    obs, info = env.reset(seed=0)
    assert obs.shape == (60, 80, 3) and obs.dtype == np.uint8 and isinstance(info, dict)
    obs, reward, terminated, truncated, info = env.step(env.action_space.sample())
    assert env.observation_space.contains(obs)
    check_env(env.unwrapped, skip_render_check=True)
    env.close()


def test_reset_seed_reproduces_the_episode():
    import miniworld_amd  # noqa: F401
    a = gymnasium.make("MiniWorld-OneRoom-v0")
    o1, _ = a.reset(seed=7)
    o2, _ = a.reset(seed=7)
    assert np.array_equal(o1, o2)
    a.close()


def test_vector_env_is_a_gymnasium_vector_env():
    """MiniWorldVectorEnv: a gymnasium.vector.VectorEnv with same-step autoreset, batched spaces, the 5-tuple step."""
    from miniworld_amd.vector import MiniWorldVectorEnv
    envs = MiniWorldVectorEnv("MiniWorld-Hallway-v0", num_envs=8, to_numpy=True)
    assert isinstance(envs, gymnasium.vector.VectorEnv)
    assert str(envs.metadata["autoreset_mode"]).lower().replace("_", "-").endswith("same-step")
    obs, infos = envs.reset(seed=0)
    assert obs.shape == (8, 60, 80, 3) and envs.observation_space.shape == obs.shape
    assert envs.single_action_space.n == envs.vec.n_actions and envs.action_space.shape == (8,)
    obs, rew, term, trunc, infos = envs.step(envs.action_space.sample())
    assert obs.shape == (8, 60, 80, 3) and rew.shape == term.shape == trunc.shape == (8,)
    envs.close()
