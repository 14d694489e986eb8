"""Entry-point helpers used by main.py: move the config's model to its GPU, resume from a checkpoint, wrap the config's datasets in DataLoaders.
Mirrors the call surface of the reference's functions.py (load_model / load_datasets / find_last_checkpoint, functions.py:25-185) so its
main.py and configs keep working against this package; written for the one-process-per-GPU launch (torchrun or --distributed spawn)."""
import glob
import os
import re

import torch


def find_last_checkpoint(callback_path, return_full_path=False):
    """newest `checkpoints_epoch_<E>_step_<S>.ckpt` under callback_path (by epoch, then step), or None"""
    best, best_key = None, None
    for path in glob.glob(os.path.join(callback_path, "checkpoints_epoch_*_step_*.ckpt")):
        m = re.search(r"checkpoints_epoch_(\d+)_step_(\d+)\.ckpt$", path)
        if m and (best_key is None or (int(m.group(1)), int(m.group(2))) > best_key):
            best, best_key = path, (int(m.group(1)), int(m.group(2)))
    if best is None:
        return None
    return best if return_full_path else os.path.basename(best)


def load_model(args):
    if args.cpu or not torch.cuda.is_available():
        raise RuntimeError("avec_amd has no CPU execution path (the CPU restatement under oracle/ is test infrastructure): a MI355X is required")
    device = torch.device("cuda", args.local_rank)
    torch.cuda.set_device(device)
    if args.rank == 0 or args.dist_log:
        props = torch.cuda.get_device_properties(device)
        print("Rank {} device: {}, {}, {} MB".format(args.rank, device, props.name, props.total_memory // 10 ** 6))
    if args.distributed:
        torch.distributed.barrier()
    model = args.config.model.to(device)
    if not hasattr(args.config, "callback_path"):
        args.config.callback_path = os.path.join("callbacks", *os.path.splitext(args.config_file)[0].split(os.sep)[1:])
    if args.load_last:
        last = find_last_checkpoint(args.config.callback_path)
        if last is not None:
            args.checkpoint = last
    if args.checkpoint is not None:
        model.load(os.path.join(args.config.callback_path, args.checkpoint))
    if args.distributed:
        torch.distributed.barrier()
    if args.rank == 0:
        model.summary(show_dict=args.show_dict)
    if args.distributed:
        if args.rank == 0:
            print("Parallelize model on", args.world_size, "GPUs")
        model.distribute_strategy(args.local_rank)
    return model


def _sample_lengths(dataset):
    """per-sample durations when the dataset can tell them without loading the clips (nnet.datasets.LRS: `durations`; ConcatDataset of such), else None"""
    if hasattr(dataset, "durations"):
        return list(dataset.durations)
    if hasattr(dataset, "datasets") and all(hasattr(d, "durations") for d in dataset.datasets):
        return [x for d in dataset.datasets for x in d.durations]
    return None


def _loader(dataset, args, drop_last):
    # length-bucketed batches (avec_amd/nnet/samplers.py): opt-in per dataset (`bucket_by_length=True` / a window size) or by AVEC_BUCKET_BY_LENGTH=1 -- the batch
    # COMPOSITION differs from the reference's uniform draw (same samples per epoch), which is why it is not the default
    want = getattr(dataset, "bucket_by_length", None) or (os.environ.get("AVEC_BUCKET_BY_LENGTH", "0") == "1")
    lengths = _sample_lengths(dataset) if (want and drop_last) else None
    if lengths is not None:
        from avec_amd.nnet.samplers import LengthBucketBatchSampler
        bs = LengthBucketBatchSampler(lengths, dataset.batch_size, window=want if isinstance(want, int) and not isinstance(want, bool) else None,
                                      shuffle=getattr(dataset, "shuffle", False), drop_last=True, rank=args.rank if args.distributed else 0,
                                      world_size=args.world_size if args.distributed else 1)
        loader = torch.utils.data.DataLoader(dataset, batch_sampler=bs, num_workers=args.num_workers, collate_fn=dataset.collate_fn, pin_memory=False)
        loader.sampler_for_epoch = bs
        if args.rank == 0:
            print("Training dataset: {}, {:,} samples - {:,} length-bucketed batches of {} (padded-frame efficiency {:.2f})".format(
                type(dataset).__name__, len(dataset), len(loader), dataset.batch_size, bs.padded_frame_efficiency()))
        return loader
    sampler = None
    if args.distributed:
        sampler = torch.utils.data.distributed.DistributedSampler(dataset, num_replicas=args.world_size, rank=args.rank, shuffle=getattr(dataset, "shuffle", False))
    loader = torch.utils.data.DataLoader(dataset, batch_size=dataset.batch_size, shuffle=False if args.distributed else getattr(dataset, "shuffle", False),
                                         sampler=sampler, num_workers=args.num_workers, collate_fn=dataset.collate_fn, pin_memory=False, drop_last=drop_last)
    if args.rank == 0:
        print("{} dataset: {}, {:,} samples - {:,} batches - batch size {}{}".format(
            "Training" if drop_last else "Evaluation", type(dataset).__name__, len(dataset), len(loader), dataset.batch_size,
            " x {}".format(args.world_size) if args.distributed else ""))
    return loader


def load_datasets(args):
    cfg = args.config
    train = _loader(cfg.training_dataset, args, True) if hasattr(cfg, "training_dataset") else None
    evaluation = None
    if hasattr(cfg, "evaluation_dataset"):
        ev = cfg.evaluation_dataset
        evaluation = [_loader(d, args, False) for d in ev] if isinstance(ev, (list, tuple)) else _loader(ev, args, False)
    return train, evaluation
