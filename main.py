"""Command-line entry point with the reference's interface (main.py:30-124 there):

    python main.py -c <config.py> -m training|evaluation|eval_time|pass [-i checkpoint | --load_last] [--steps_per_epoch N] [--eval_steps N]
    torchrun --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 main.py -c <config.py> -d        (one process per GPU, RCCL)

The config is a Python file that builds `model` (an nnet.Model, already compiled) and optionally `training_dataset`, `evaluation_dataset`,
`callback_path`, `epochs`, `precision`, `accumulated_steps`, ... exactly like configs/LRS23/AV/EffConfInterCTC.py of the reference."""
import argparse
import importlib.util
import os
import sys

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import functions  # noqa: E402


def load_config(path):
    from avec_amd.compat import ensure_torchvision
    ensure_torchvision()              # configs do `import torchvision` for three transform classes: a stand-in when the package is absent (avec_amd/compat)
    spec = importlib.util.spec_from_file_location("avec_config", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def run(args):
    if args.rank == 0:
        print("Mode: {}".format(args.mode))
    if args.distributed:
        torch.cuda.set_device(args.local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.distributed.init_process_group(backend=args.backend, init_method="env://", world_size=args.world_size, rank=args.rank)
    args.config = load_config(args.config_file)
    cfg = args.config
    model = functions.load_model(args)
    dataset_train, dataset_eval = functions.load_datasets(args)
    assert args.mode in ("training", "evaluation", "pass", "eval_time"), "modes: training, evaluation, eval_time, pass"
    if args.mode == "training":
        initial_epoch = int(args.checkpoint.split("_")[2]) if args.checkpoint is not None else 0
        model.fit(dataset_train, epochs=getattr(cfg, "epochs", 1000), dataset_eval=dataset_eval, eval_steps=getattr(cfg, "eval_steps", args.eval_steps),
                  verbose_eval=args.verbose_eval, initial_epoch=initial_epoch, callback_path=cfg.callback_path, steps_per_epoch=args.steps_per_epoch,
                  precision=getattr(cfg, "precision", torch.float32), accumulated_steps=getattr(cfg, "accumulated_steps", 1),
                  eval_period_epoch=getattr(cfg, "eval_period_epoch", args.eval_period_epoch),
                  saving_period_epoch=getattr(cfg, "saving_period_epoch", args.saving_period_epoch), step_log_period=args.step_log_period,
                  eval_training=getattr(cfg, "eval_training", not args.no_eval_training))
    elif args.mode == "evaluation":
        loaders = dataset_eval if isinstance(dataset_eval, list) else [dataset_eval]
        for loader in loaders:
            res = model.evaluate(loader, eval_steps=getattr(cfg, "eval_steps", args.eval_steps), verbose=args.verbose_eval,
                                 recompute_metrics=getattr(cfg, "recompute_metrics", False))
            if args.rank == 0:
                print("Evaluation:", {k: round(v, 4) for k, v in res.items()})
    elif args.mode == "eval_time":
        loaders = dataset_eval if isinstance(dataset_eval, list) else [dataset_eval]
        t = sum(model.eval_time(loader, eval_steps=getattr(cfg, "eval_steps", args.eval_steps)) for loader in loaders)
        if args.rank == 0:
            print("Eval time: {}".format(t))
    if args.distributed:
        torch.distributed.destroy_process_group()


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("-c", "--config_file", type=str, default="configs/LRS23/AV/EffConfInterCTC.py")
    ap.add_argument("-m", "--mode", type=str, default="training")
    ap.add_argument("-i", "--checkpoint", type=str, default=None)
    ap.add_argument("-j", "--num_workers", type=int, default=0)
    ap.add_argument("--cpu", action="store_true")
    ap.add_argument("--load_last", action="store_true")
    ap.add_argument("-d", "--distributed", action="store_true")
    ap.add_argument("--dist_log", action="store_true")
    ap.add_argument("--backend", type=str, default="nccl")
    ap.add_argument("--steps_per_epoch", type=int, default=None)
    ap.add_argument("--saving_period_epoch", type=int, default=1)
    ap.add_argument("--step_log_period", type=int, default=100)
    ap.add_argument("--no_eval_training", action="store_true")
    ap.add_argument("--eval_period_epoch", type=int, default=1)
    ap.add_argument("--verbose_eval", type=int, default=0)
    ap.add_argument("--eval_steps", type=int, default=None)
    ap.add_argument("--show_dict", action="store_true")
    args = ap.parse_args()
    # one process per GPU: ranks come from the launcher's environment (torchrun); -d without a launcher = a single-rank group
    args.world_size = int(os.environ.get("WORLD_SIZE", "1"))
    args.rank = int(os.environ.get("RANK", "0"))
    args.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.distributed = args.distributed or args.world_size > 1
    if args.distributed:
        os.environ.setdefault("MASTER_PORT", "29501")
        os.environ.setdefault("RANK", str(args.rank))
        os.environ.setdefault("WORLD_SIZE", str(args.world_size))
    run(args)


if __name__ == "__main__":
    main()
