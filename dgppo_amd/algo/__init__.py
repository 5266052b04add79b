"""dgppo.algo equivalent (dgppo/algo/__init__.py:8-18)."""
from .base import Algorithm
from .dgppo import DGPPO
from .informarl import HCBFCRPO, InforMARL


def make_algo(algo: str, **kwargs) -> Algorithm:
    if algo == "dgppo":
        return DGPPO(**kwargs)
    if algo == "informarl":
        return InforMARL(**kwargs)
    if algo == "hcbfcrpo":
        return HCBFCRPO(**kwargs)
    if algo == "informarl_lagr":
        raise NotImplementedError(f"algo '{algo}' is a baseline of the reference outside the scope of this build "
                                  f"(SURVEY §2 rows 18-19, §8f rank 3); available: 'dgppo', 'informarl', 'hcbfcrpo'")
    raise ValueError(f"Unknown algorithm: {algo}")
