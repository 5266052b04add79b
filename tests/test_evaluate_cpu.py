"""Host logic of the evaluation path (test.py): safe-rate statistics on hand-computed cases, the
config.yaml round trip (including the reference's argparse.Namespace tag, read without executing anything), checkpoint
step discovery and the test_log.csv row."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import yaml

from dgppo_amd.trainer import evaluate as EV


def test_unsafe_mask_and_episode_stats_hand_case():
    # 2 episodes, T=3, 2 agents, 2 cost components; cost >= 0 is unsafe (test.py:103-105: `cost >= 0.0`, zero included)
    costs = np.full((2, 3, 2, 2), -0.6, dtype=np.float32)
    costs[0, 1, 0, 1] = 0.0            # episode 0, agent 0 unsafe at t=1 (exactly zero counts)
    costs[1, 2, 1, 0] = 0.7            # episode 1, agent 1 unsafe at t=2
    costs[1, 0, 1, 1] = -1e-9          # negative: still safe
    rewards = np.array([[-1.0, -2.0, -3.0], [0.5, 0.25, 0.25]], dtype=np.float32)
    m = EV.unsafe_mask(costs)
    assert m.shape == (2, 3, 2) and m.sum() == 2 and m[0, 1, 0] and m[1, 2, 1]
    st = EV.episode_stats(rewards, costs)
    np.testing.assert_allclose(st["reward"], [-6.0, 1.0])
    np.testing.assert_allclose(st["cost"], [0.0, 0.7], rtol=1e-6)
    np.testing.assert_allclose(st["safe_rate"], [0.5, 0.5])          # one of two agents ever unsafe in each episode
    agg = EV.aggregate(st)
    assert agg["safe_mean"] == pytest.approx(0.5) and agg["safe_std"] == pytest.approx(0.5)
    assert agg["reward"] == pytest.approx(-2.5) and agg["reward_min"] == -6.0 and agg["reward_max"] == 1.0
    assert agg["cost"] == pytest.approx(0.35) and agg["cost_max"] == pytest.approx(0.7)


def test_all_safe_and_all_unsafe():
    safe = EV.aggregate(EV.episode_stats(np.zeros((3, 4)), np.full((3, 4, 5, 2), -1.0)))
    assert safe["safe_mean"] == 1.0 and safe["safe_std"] == 0.0
    unsafe = EV.aggregate(EV.episode_stats(np.zeros((3, 4)), np.full((3, 4, 5, 2), 1.0)))
    assert unsafe["safe_mean"] == 0.0 and unsafe["safe_std"] == 0.0


def test_config_yaml_round_trip(tmp_path):
    # what this build's train.py writes: two safe_dump'ed mappings back to back
    p = tmp_path / "config.yaml"
    with open(p, "w") as f:
        yaml.safe_dump({"env": "LidarSpread", "num_agents": 8, "obs": 3, "algo": "dgppo", "seed": 0}, f)
        yaml.safe_dump({"cost_weight": 0.0, "actor_gnn_layers": 2, "Vl_gnn_layers": 2, "Vh_gnn_layers": 1, "lr_actor": 3e-4,
                        "lr_Vl": 1e-3, "use_rnn": True, "rnn_layers": 1, "use_lstm": False}, f)
    c = EV.load_config(str(p))
    assert c.env == "LidarSpread" and c.num_agents == 8 and c.obs == 3 and c.Vh_gnn_layers == 1 and c.use_rnn is True


def test_config_yaml_reference_namespace_tag_is_read_as_mapping(tmp_path):
    # the layout the reference's train.py produces (yaml.dump(args) of an argparse.Namespace + yaml.dump(algo.config))
    p = tmp_path / "config.yaml"
    p.write_text("!!python/object:argparse.Namespace\nalgo: dgppo\nenv: LidarTarget\nnum_agents: 4\nobs: 2\nseed: 3\n"
                 "cost_weight: 0.0\nactor_gnn_layers: 2\nVl_gnn_layers: 2\nlr_actor: 0.0003\nlr_Vl: 0.001\n"
                 "use_rnn: true\nrnn_layers: 1\nuse_lstm: false\n")
    c = EV.load_config(str(p))
    assert c.algo == "dgppo" and c.env == "LidarTarget" and c.num_agents == 4 and not hasattr(c, "Vh_gnn_layers")


def test_config_yaml_refuses_other_python_tags(tmp_path):
    p = tmp_path / "config.yaml"
    p.write_text("a: !!python/object/apply:os.system ['echo pwned']\n")
    with pytest.raises(yaml.YAMLError):
        EV.load_config(str(p))


def test_latest_step_and_csv_line(tmp_path):
    for d in ("0", "100", "2000", "tmp", "30x"):
        os.makedirs(tmp_path / "models" / d)
    assert EV.latest_step(str(tmp_path / "models")) == 2000
    with pytest.raises(FileNotFoundError):
        os.makedirs(tmp_path / "empty")
        EV.latest_step(str(tmp_path / "empty"))
    env = SimpleNamespace(num_agents=8, max_episode_steps=128, area_size=1.5, params={"n_obs": 3})
    line = EV.csv_line(env, 5, {"safe_mean": 0.875, "safe_std": 0.125})
    assert line == "8,5,128,1.5,3,87.500,12.500\n"
