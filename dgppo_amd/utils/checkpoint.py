"""Weights-only reader / writer for the `{actor,Vl,Vh}.pkl` files (dgppo/algo/informarl_lagr.py:311-327,
informarl.py:459-470): a flax-named tree of arrays.

The reference loads them with plain `pickle.load`; `test.py --path` is user-supplied, so this build never does that.
`load_tree` runs a `pickle.Unpickler` whose `find_class` resolves ONLY the globals a tree of arrays needs and raises
`pickle.UnpicklingError` for anything else — nothing named in the file is imported or called:

  * numpy's array reconstruction (`_reconstruct`, `ndarray`, `dtype`, `scalar`, `_frombuffer`),
  * `collections.OrderedDict`,
  * flax's `FrozenDict` (read as a plain dict), and
  * `jax._src.array._reconstruct_array` (how a `jax.Array` pickles itself: numpy's own reduce tuple plus two state
    dicts) — rebuilt as the numpy array it wraps, so a checkpoint written by the reference loads without jax.
"""
from __future__ import annotations

import collections
import io
import pickle
from typing import Any, BinaryIO

import numpy as np

try:                                    # numpy >= 2 moved the C helpers
    import numpy._core as _np_core
    import numpy._core.multiarray  # noqa: F401
    import numpy._core.numeric  # noqa: F401
except ImportError:                     # pragma: no cover
    import numpy.core as _np_core
_np_multiarray = _np_core.multiarray


def _frozen_dict(*args, **kwargs) -> dict:
    """flax.core.frozen_dict.FrozenDict(mapping) -> dict"""
    return dict(*args, **kwargs)


def _jax_reconstruct_array(fun, args, arr_state, aval_state):
    """jax._src.array._reconstruct_array(fun, args, arr_state, aval_state): `fun(*args)` is numpy's reduce of the host
    copy; the rest (weak_type, named_shape) has no meaning for a weight."""
    if fun is not _np_multiarray._reconstruct:
        raise pickle.UnpicklingError("jax array payload is not a numpy reconstruction")
    arr = fun(*args)
    arr.__setstate__(arr_state)
    return arr


def _numpy_globals() -> dict:
    """numpy pickles name its helpers under numpy.core.* (numpy 1.x) or numpy._core.* (numpy 2.x): accept both spellings."""
    out = {("numpy", "ndarray"): np.ndarray, ("numpy", "dtype"): np.dtype}
    helpers = {"multiarray": ("_reconstruct", "scalar"), "numeric": ("_frombuffer",)}
    for sub, names in helpers.items():
        mod = getattr(_np_core, sub)
        for name in names:
            fn = getattr(mod, name, None)
            if fn is not None:
                out[(f"numpy.core.{sub}", name)] = fn
                out[(f"numpy._core.{sub}", name)] = fn
    return out


_ALLOWED = {
    **_numpy_globals(),
    ("collections", "OrderedDict"): collections.OrderedDict,
    ("flax.core.frozen_dict", "FrozenDict"): _frozen_dict,
    ("jax._src.array", "_reconstruct_array"): _jax_reconstruct_array,
}


class _TreeUnpickler(pickle.Unpickler):
    def find_class(self, module: str, name: str) -> Any:
        fn = _ALLOWED.get((module, name))
        if fn is None:
            raise pickle.UnpicklingError(f"refusing to resolve global '{module}.{name}': checkpoints may only contain "
                                         f"dicts of numpy (or jax) arrays")
        return fn

    def persistent_load(self, pid):     # no external object references in a weights file
        raise pickle.UnpicklingError("persistent ids are not allowed in a checkpoint")


def _check_tree(tree, path="") -> None:
    if isinstance(tree, dict):
        for k, v in tree.items():
            if not isinstance(k, str):
                raise pickle.UnpicklingError(f"checkpoint key {path}/{k!r} is not a string")
            _check_tree(v, f"{path}/{k}")
    elif isinstance(tree, np.ndarray):
        if tree.dtype.hasobject:
            raise pickle.UnpicklingError(f"checkpoint leaf {path} has an object dtype")
    elif not isinstance(tree, (np.generic, int, float)):
        raise pickle.UnpicklingError(f"checkpoint leaf {path} is a {type(tree).__name__}, expected an array")


def load_tree(f: BinaryIO):
    """-> nested dict of numpy arrays.  Raises pickle.UnpicklingError on anything that is not one."""
    tree = _TreeUnpickler(f).load()
    _check_tree(tree)
    return tree


def loads_tree(data: bytes):
    return load_tree(io.BytesIO(data))


def save_tree(tree, f: BinaryIO) -> None:
    """the reference's on-disk format: pickle of the nested dict with numpy leaves."""
    _check_tree(tree)
    pickle.dump(tree, f)
