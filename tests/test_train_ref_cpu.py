"""BASELINE config 1 (MPETarget n=3, obs=0, 32 envs, dgppo) runs entirely on the CPU oracle path: plumbing check.
The unmodified reference would raise on this config; the restatement guards n_obs == 0 like MPESpread."""
import numpy as np

from oracle import train_ref


def test_config1_mpetarget_cpu_plumbing():
    # reduced horizon so the CPU suite stays fast; same code path as 32 envs x 128 steps, batch 4096
    state, info = train_ref.iteration("MPETarget", 3, 0, B=32, T=16, batch_size=256, seed=0)
    for k in ("Vl/loss", "Vh/loss_Vh", "policy/loss", "policy/entropy", "policy/clip_frac", "eval/safe_data"):
        assert np.isfinite(info[k]), k
    assert info["Vl/grad_norm"] > 0 and info["policy/grad_norm"] > 0
    p0 = state["trees"]["policy"]["params"]["OutputDenseMean"]["kernel"].clone()
    state, info2 = train_ref.iteration("MPETarget", 3, 0, B=32, T=16, batch_size=256, seed=1, state=state, step=1)
    assert not np.array_equal(p0.numpy(), state["trees"]["policy"]["params"]["OutputDenseMean"]["kernel"].numpy())
    assert state["opt"].s["policy"][2] == 2 * (32 // (256 // 16))
