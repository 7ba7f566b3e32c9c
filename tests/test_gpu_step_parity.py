"""HIP step kernel vs trajectories of the reference's own dynamics (GPU box).

The golden trajectories were produced by the reference's unmodified miniworld.py /
entity.py / math.py run under GL stubs (tools/gen_golden.py).  The per-step parameters the
reference drew (forward_step, forward_drift, turn_step) are injected, so the comparison is
state-for-state: flags / rewards / carried slot identical, positions within 1e-12.
"""
import numpy as np
import pytest

import helpers
from conftest import golden_cases

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", golden_cases())
def test_step_matches_reference_trajectory(case):
    import torch
    s0, tr, meta, obs = helpers.load_case(case)
    rule = helpers.rule_of(meta)
    if rule == "api_only":
        pytest.skip("the env's Python rule moves entities (CollectHealth respawn): covered through the env API")
    task = helpers.task_of(meta)
    g0, g1 = helpers.goals_of(meta)
    eng = helpers.make_engine_for_scene(s0, 1, task=task, goal_ent=g0, goal_ent2=g1,
                                        agent_radius=float(meta.get("agent_radius", 0.4)))
    eng.set_state(helpers.scene_state_arrays([s0]))
    E = len(s0["ents_kind"])
    T = len(tr["action"])
    rgb = torch.zeros((1, 60, 80, 3), dtype=torch.uint8, device="cuda")
    act = torch.zeros(1, dtype=torch.int32, device="cuda")
    rew = torch.zeros(1, dtype=torch.float32, device="cuda")
    term = torch.zeros(1, dtype=torch.uint8, device="cuda")
    trunc = torch.zeros(1, dtype=torch.uint8, device="cuda")
    maxerr = 0.0
    poke = meta.get("poke", np.array([-1.0]))
    for t in range(T):
        if int(poke[0]) == t:
            st = eng.get_state()
            st["ent_pos"][0, int(poke[1])] = poke[2:5]
            eng.set_state({"ent_pos": st["ent_pos"]})
        eng.set_step_params(np.array([[tr["fwd_step"][t], tr["fwd_drift"][t], tr["turn_step"][t]]]))
        act[0] = int(tr["action"][t])
        eng.step(act, rgb, None, rew, term, trunc)
        st = eng.get_state()
        if rule == "engine":
            assert np.float32(tr["reward"][t]) == rew.item(), (case, t)
            assert bool(term.item()) == bool(tr["term"][t]), (case, t)
        assert bool(trunc.item()) == bool(tr["trunc"][t]), (case, t)
        assert int(st["carrying"][0]) == int(tr["carrying"][t]), (case, t)
        maxerr = max(maxerr, np.abs(st["agent_pos"][0] - tr["pos"][t]).max(), abs(st["agent_dir"][0] - tr["dir"][t]))
        alive = tr["ents_alive"][t].astype(bool)
        assert np.array_equal(st["ent_kind"][0, :E] != 0, alive), (case, t)
        if alive.any():
            maxerr = max(maxerr, np.abs(st["ent_pos"][0, :E][alive] - tr["ents_pos"][t][alive]).max())
            maxerr = max(maxerr, np.abs(st["ent_dir"][0, :E][alive] - tr["ents_dir"][t][alive]).max())
        assert int(st["step_count"][0]) == t + 1
    assert maxerr < 1e-12, f"{case}: max state error {maxerr}"
    eng.check()
    eng.close()
