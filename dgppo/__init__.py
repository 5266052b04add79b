"""`dgppo` import alias: the reference's package name mapped onto the MI355X implementation, so that
`from dgppo.env import make_env`, `from dgppo.algo import make_algo`, `from dgppo.trainer.trainer import Trainer`
(train.py:9-12 of the reference) work unchanged."""
import importlib
import sys

import dgppo_amd

_ALIASES = ["env", "env.base", "env.envs", "algo", "algo.base", "algo.dgppo", "trainer", "trainer.trainer", "trainer.utils",
            "trainer.data", "utils", "utils.graph"]
for _name in _ALIASES:
    _mod = importlib.import_module(f"dgppo_amd.{_name}")
    sys.modules[f"dgppo.{_name}"] = _mod
    if "." not in _name:
        globals()[_name] = _mod
