"""Entry point with the semantics of the reference's main.py:8-21 (hydra compose + instantiate + run_rl)."""
import logging
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
for p in (HERE, os.path.dirname(HERE)):
    if p not in sys.path:
        sys.path.insert(0, p)

from matinvent_amd import config as C  # noqa: E402


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    config_dir, config_name, overrides = os.path.join(HERE, "configs"), "base", []
    while argv:
        a = argv.pop(0)
        if a in ("--config-dir", "-cd"):
            config_dir = os.path.abspath(argv.pop(0))
        elif a in ("--config-name", "-cn"):
            config_name = argv.pop(0)
        else:
            overrides.append(a)
    logging.basicConfig(level=logging.INFO, format="[%(asctime)s][%(levelname)s] %(message)s")
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:  # one process per GPU, RCCL (backend "nccl" on ROCm)
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl" if torch.cuda.is_available() else "gloo")
    cfg = C.compose(config_dir, config_name, overrides)
    run_dir = C.run_dir(cfg)
    os.makedirs(run_dir, exist_ok=True)
    os.chdir(run_dir)                                   # hydra.run.dir
    C.save(cfg, "hparams.yaml")                         # main.py:13
    cfg = C.resolved(cfg)
    reinl = C.instantiate(cfg.pipeline, model_suite=cfg.model, reward=cfg.reward, logger=cfg.logger)  # main.py:15-20
    reinl.run_rl()
    return reinl


if __name__ == "__main__":
    main()
