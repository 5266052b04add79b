#!/usr/bin/env python
"""Training entry point of the MI355X build.  The command line is the reference's (train.py:133-182): every flag keeps
its name, type and default so that existing launch scripts work unchanged; the flag table below is the single source of
truth for them."""
import argparse
import datetime
import os

import numpy as np
import yaml

from dgppo.algo import make_algo
from dgppo.env import make_env
from dgppo.trainer.trainer import Trainer
from dgppo.trainer.utils import is_connected

# (flags, kind, default)   kind: a type -> typed option, "flag" -> store_true, "req:<type>" -> required option
_T = {"int": int, "float": float, "str": str}
FLAGS = [
    (("--env",), "req:str", None), (("-n", "--num-agents"), "req:int", None), (("--algo",), "req:str", None),
    (("--obs",), "req:int", None),
    (("--seed",), "int", 0), (("--steps",), "int", 200000), (("--name",), "str", None), (("--debug",), "flag", False),
    (("--cost-weight",), "float", 0.0), (("--n-rays",), "int", 32), (("--full-observation",), "flag", False),
    (("--clip-eps",), "float", 0.25), (("--lagr-init",), "float", 0.5), (("--lr-lagr",), "float", 1e-7),
    (("--cbf-weight",), "float", 1.0), (("--cbf-eps",), "float", 1e-2), (("--alpha",), "float", 10.0),
    (("--no-cbf-schedule",), "flag", False), (("--cost-schedule",), "flag", False), (("--no-rnn",), "flag", False),
    (("--actor-gnn-layers",), "int", 2), (("--Vl-gnn-layers",), "int", 2), (("--Vh-gnn-layers",), "int", 1),
    (("--lr-actor",), "float", 3e-4), (("--lr-Vl",), "float", 1e-3), (("--lr-Vh",), "float", 1e-3),
    (("--rnn-layers",), "int", 1), (("--use-lstm",), "flag", False), (("--coef-ent",), "float", 1e-2),
    (("--rnn-step",), "int", 16), (("--n-env-train",), "int", 128), (("--batch-size",), "int", 16384),
    (("--n-env-test",), "int", 32), (("--log-dir",), "str", "./logs"), (("--eval-interval",), "int", 50),
    (("--eval-epi",), "int", 1), (("--save-interval",), "int", 50),
    # not in the reference (it is single-device): data-parallel training, one process per GPU
    (("--gpus",), "int", 1),
]


def build_parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser(description=__doc__)
    for names, kind, default in FLAGS:
        if kind == "flag":
            ap.add_argument(*names, action="store_true", default=False)
        elif kind.startswith("req:"):
            ap.add_argument(*names, type=_T[kind[4:]], required=True)
        else:
            ap.add_argument(*names, type=_T[kind], default=default)
    return ap


def _unique_run_dir(root: str, seed: int):
    """{root}/seed{seed}_{MMDDhhmmss}_{4 random capitals}, bumped until it does not exist (train.py:81-93)."""
    tag = "".join(chr(c) for c in np.random.default_rng().integers(65, 91, size=4))
    stamp = int(datetime.datetime.now().strftime("%m%d%H%M%S"))
    while os.path.exists(f"{root}/seed{seed}_{stamp}_{tag}"):
        stamp += 1
    return f"{root}/seed{seed}_{stamp}_{tag}", stamp, tag


def _algo_kwargs(a, env, world: int = 1) -> dict:
    """what the reference hands to make_algo (train.py:44-77); --batch-size is the GLOBAL minibatch, a rank trains on its
    1/world share of it"""
    return dict(
        algo=a.algo, env=env, node_dim=env.node_dim, edge_dim=env.edge_dim, state_dim=env.state_dim,
        action_dim=env.action_dim, n_agents=env.num_agents, seed=a.seed, train_steps=a.steps, batch_size=a.batch_size // world,
        gamma=0.99, max_grad_norm=2.0, clip_eps=a.clip_eps, coef_ent=a.coef_ent,
        actor_gnn_layers=a.actor_gnn_layers, Vl_gnn_layers=a.Vl_gnn_layers, Vh_gnn_layers=a.Vh_gnn_layers,
        lr_actor=a.lr_actor, lr_Vl=a.lr_Vl, lr_Vh=a.lr_Vh,
        use_rnn=not a.no_rnn, rnn_layers=a.rnn_layers, rnn_step=a.rnn_step, use_lstm=a.use_lstm,
        alpha=a.alpha, cbf_eps=a.cbf_eps, cbf_weight=a.cbf_weight, cbf_schedule=not a.no_cbf_schedule,
        cost_weight=a.cost_weight, cost_schedule=a.cost_schedule, lagr_init=a.lagr_init, lr_lagr=a.lr_lagr)


def _setup_ranks(a):
    """-> (rank, world, allreduce, close).  `--gpus N` under `python -m torch.distributed.run --nproc-per-node N` (or any
    launcher that sets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*) uses that environment; a bare `python train.py --gpus N` has
    started its own supervised ranks in main() before getting here.  Each rank binds to GPU LOCAL_RANK, joins the gloo
    control plane and the RCCL communicator of the C ABI, and proves the all-reduce with a known-answer check."""
    import torch
    from dgppo_amd import dist as D
    rank, local_rank, world = D.env_info()
    if world != a.gpus:
        raise SystemExit(f"train.py: --gpus {a.gpus} but WORLD_SIZE={world}: launch one rank per GPU")
    if world == 1:
        return 0, 1, None, (lambda: None)
    backend = os.environ.get("DGPPO_DIST_BACKEND", "rccl")
    n_dev = torch.cuda.device_count()
    if world > n_dev and backend != "gloo":
        raise SystemExit(f"train.py: {world} ranks but {n_dev} GPU(s) visible (DGPPO_DIST_BACKEND=gloo rehearses several "
                         f"ranks on one GPU)")
    assert a.n_env_train % world == 0 and a.batch_size % world == 0, "--n-env-train and --batch-size must be multiples of --gpus"
    torch.cuda.set_device(local_rank % max(n_dev, 1))
    D.init_control_plane()
    allreduce, close = D.make_allreduce(world, backend)
    D.selfcheck_allreduce(allreduce, rank, world, torch.device("cuda", torch.cuda.current_device()))
    if rank == 0:
        print(f"> data-parallel: {world} ranks, data plane {backend} (RCCL version {D.rccl_version()}), "
              f"{a.n_env_train // world} envs and a minibatch share of {a.batch_size // world} per rank", flush=True)

    def close_all():
        close()
        D.shutdown(world)
    return rank, world, allreduce, close_all


def train(a):
    rank, world, allreduce, close = _setup_ranks(a)
    if rank == 0:
        print(f"> Running train.py {a}")
    np.random.seed(a.seed)
    if a.debug or rank != 0:
        os.environ["WANDB_MODE"] = "disabled"
    elif not is_connected():
        os.environ["WANDB_MODE"] = "offline"

    def new_env():
        return make_env(env_id=a.env, num_agents=a.num_agents, num_obs=a.obs, n_rays=a.n_rays,
                        full_observation=a.full_observation)

    env, env_test = new_env(), new_env()
    algo = make_algo(**_algo_kwargs(a, env, world), allreduce=allreduce, world=world, rank=rank)

    root = f"{a.log_dir}/{a.env}/{a.algo}"
    write = not a.debug and rank == 0                # one writer (rank 0); the other ranks never touch the log directory
    if write:
        os.makedirs(root, exist_ok=True)
    log_dir, stamp, tag = _unique_run_dir(root, a.seed)
    run_name = f"{a.algo}_seed{a.seed:03}_{stamp}_{tag}"
    if a.name is not None:
        run_name = f"{run_name}_{a.name}_seed{a.seed:03}_{stamp}_{tag}"
    schedule = {"run_name": run_name, "training_steps": a.steps, "eval_interval": a.eval_interval, "eval_epi": a.eval_epi,
                "save_interval": a.save_interval}
    trainer = Trainer(env=env, env_test=env_test, algo=algo, gamma=0.99, log_dir=log_dir, n_env_train=a.n_env_train,
                      n_env_test=a.n_env_test, seed=a.seed, params=schedule, save_log=write, rank=rank, world=world)
    if write:   # plain mappings (the reference dumps the argparse.Namespace object itself; test.py reads both)
        with open(f"{log_dir}/config.yaml", "w") as f:
            yaml.safe_dump(vars(a), f)
            yaml.safe_dump(algo.config, f)
    trainer.train()
    close()


def main():
    a = build_parser().parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher environment: start the ranks ourselves, before this process touches the GPU, and supervise them
        # (per-rank logs, everything stops on the first failure — dgppo_amd/launch.py)
        import sys
        from dgppo_amd import launch
        here = os.path.dirname(os.path.abspath(__file__))
        sys.exit(launch.spawn_ranks(os.path.abspath(__file__), sys.argv[1:], a.gpus, launch.default_log_dir(here),
                                    stall_seconds=float(os.environ.get("DGPPO_STALL_SECONDS", "900"))))
    train(a)


if __name__ == "__main__":
    main()
