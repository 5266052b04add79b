"""dgppo.algo equivalent (dgppo/algo/__init__.py:8-18)."""
from .base import Algorithm
from .dgppo import DGPPO
from .informarl import HCBFCRPO, InforMARL, InforMARLLagr


def make_algo(algo: str, **kwargs) -> Algorithm:
    if algo == "dgppo":
        return DGPPO(**kwargs)
    if algo == "informarl":
        return InforMARL(**kwargs)
    if algo == "hcbfcrpo":
        return HCBFCRPO(**kwargs)
    if algo == "informarl_lagr":
        return InforMARLLagr(**kwargs)
    raise ValueError(f"Unknown algorithm: {algo}")
