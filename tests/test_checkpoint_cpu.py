"""Weights-only checkpoint reader (dgppo_amd/utils/checkpoint.py): our own files round-trip, files shaped like the
reference's (flax FrozenDict + jax.Array leaves, informarl_lagr.py:311-327) load WITHOUT jax, and a pickle naming any
other global is refused before anything runs."""
import io
import os
import pickle
import sys
import types

import numpy as np
import pytest

from dgppo_amd.utils import checkpoint as C


def _tree():
    rng = np.random.default_rng(0)
    return {"params": {"Dense_0": {"kernel": rng.standard_normal((7, 96)).astype(np.float32), "bias": np.zeros(96, np.float32)},
                       "LayerNorm_0": {"scale": np.ones(64, np.float32)}}}


def test_round_trip():
    buf = io.BytesIO()
    C.save_tree(_tree(), buf)
    got = C.loads_tree(buf.getvalue())
    np.testing.assert_array_equal(got["params"]["Dense_0"]["kernel"], _tree()["params"]["Dense_0"]["kernel"])
    assert got["params"]["LayerNorm_0"]["scale"].dtype == np.float32


class _Boom:
    def __reduce__(self):
        return (os.system, ("echo this must never run",))


@pytest.mark.parametrize("payload", [{"x": _Boom()}, _Boom(), {"params": {"k": [_Boom()]}}])
def test_foreign_global_is_refused(payload, capfd):
    with pytest.raises(pickle.UnpicklingError, match="refusing to resolve global"):
        C.loads_tree(pickle.dumps(payload))
    assert "must never run" not in capfd.readouterr().out


def test_builtin_eval_is_refused():
    evil = b"cbuiltins\neval\n(S'1+1'\ntR."
    with pytest.raises(pickle.UnpicklingError):
        C.loads_tree(evil)


def test_non_array_leaf_is_refused():
    with pytest.raises(pickle.UnpicklingError, match="expected an array"):
        C.loads_tree(pickle.dumps({"params": {"k": "a string"}}))
    with pytest.raises(pickle.UnpicklingError, match="object dtype"):
        C.loads_tree(pickle.dumps({"k": np.array([{"a": 1}], dtype=object)}))


def test_reference_shaped_checkpoint_loads_without_jax():
    """A file as the reference writes it: FrozenDict of jax.Array.  jax / flax are not installed, so stand-in modules with
    the same qualified names produce the same pickle opcodes (GLOBAL 'jax._src.array _reconstruct_array', REDUCE with numpy's
    own reduce tuple + two state dicts); they are removed again before loading."""
    assert "jax" not in sys.modules
    fake_jax = types.ModuleType("jax._src.array")

    def _reconstruct_array(fun, args, arr_state, aval_state):      # never called here: pickled BY REFERENCE only
        raise AssertionError
    _reconstruct_array.__module__ = "jax._src.array"
    _reconstruct_array.__qualname__ = "_reconstruct_array"
    fake_jax._reconstruct_array = _reconstruct_array

    class FakeJaxArray:
        def __init__(self, a):
            self.a = a

        def __reduce__(self):
            fun, args, arr_state = self.a.__reduce__()
            return _reconstruct_array, (fun, args, arr_state, {"weak_type": False, "named_shape": {}})

    fake_flax = types.ModuleType("flax.core.frozen_dict")

    class FrozenDict(dict):
        def __reduce__(self):
            return FrozenDict, (dict(self),)
    FrozenDict.__module__ = "flax.core.frozen_dict"
    FrozenDict.__qualname__ = "FrozenDict"
    fake_flax.FrozenDict = FrozenDict
    added = {"jax._src.array": fake_jax, "flax.core.frozen_dict": fake_flax}
    for parent in ("jax", "jax._src", "flax", "flax.core"):          # importable parents so pickle can verify the reference
        pkg = types.ModuleType(parent)
        pkg.__path__ = []
        added[parent] = pkg
    sys.modules.update(added)
    try:
        t = _tree()
        wrapped = FrozenDict({"params": FrozenDict({"Dense_0": FrozenDict({k: FakeJaxArray(v) for k, v in t["params"]["Dense_0"].items()})})})
        data = pickle.dumps(wrapped)
    finally:
        for k in added:
            sys.modules.pop(k, None)
    assert b"jax._src.array" in data and b"flax.core.frozen_dict" in data
    got = C.loads_tree(data)
    assert type(got) is dict and type(got["params"]) is dict
    np.testing.assert_array_equal(got["params"]["Dense_0"]["kernel"], t["params"]["Dense_0"]["kernel"])
