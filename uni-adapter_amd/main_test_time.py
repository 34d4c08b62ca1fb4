"""Per-corruption driver with the reference's flags and summary (`main_test-time.py:25-104`,
`utils/params.py`): for every corruption build the stream, run `test_zeroshot_3d_core`, collect
top-1, log the average.  Run as  `python -m uni_adapter_amd.main_test_time ...`.

Out of scope here as in DESIGN.md section 8: the CLIP text tower (pass --precomputed-text-features)."""
import argparse
import logging
import os
import random
import sys
from datetime import datetime

import numpy as np
import torch

from .tta_data import CORRUPTIONS, load_point_encoder, load_tta_dataset
from .Uni_Adapter import get_hyperparams, test_zeroshot_3d_core


def parse_args(argv=None):
    """The subset of `utils/params.py` this path reads, same spellings and defaults."""
    p = argparse.ArgumentParser('Uni-Adapter test-time adaptation (MI355X path)')
    p.add_argument('--name', type=str, default=None)
    p.add_argument('--output-dir', type=str, default='./outputs')
    p.add_argument('--vlm3d', type=str, default='uni3d', choices=['uni3d', 'ulip', 'openshape'])
    p.add_argument('--precomputed-text-features', type=str, default=None)
    p.add_argument('--pretrained-pc-uni3d', type=str, default=None)
    p.add_argument('--pretrained-pc-ulip', type=str, default=None)
    p.add_argument('--pretrained-pc-oshape', type=str, default=None)
    p.add_argument('--num-group-uni3d', type=int, default=512)
    p.add_argument('--group_size_uni3d', type=int, default=64)
    p.add_argument('--myroot', type=str, required=True)
    p.add_argument('--dataset_name', type=str, default='modelnet')
    p.add_argument('--validate_dataset_name', type=str, default='modelnet40_openshape')
    p.add_argument('--batch-size', type=int, default=1)
    p.add_argument('--workers', type=int, default=0)
    p.add_argument('--npoints', type=int, default=1024)
    p.add_argument('--corruption', type=str, default='all')
    p.add_argument('--severity', type=int, default=5)
    p.add_argument('--seed', type=int, default=42)
    p.add_argument('--device', type=str, default='cuda:0')
    p.add_argument('--debug', action='store_true')
    p.add_argument('--dota-epsilon', type=float, default=0.0001)
    p.add_argument('--dota-sigma', type=float, default=0.0001)
    p.add_argument('--dota-eta', type=float, default=0.1)
    p.add_argument('--dota-rho', type=float, default=0.02)
    p.add_argument('--use-mode-dota', action='store_true', default=True)
    p.add_argument('--no-use-mode-dota', dest='use_mode_dota', action='store_false', help='with --use-dota: the full-covariance DOTA cache (dota.py)')
    p.add_argument('--use-dota', action='store_true', default=False)
    p.add_argument('--dota-prior_pre_steps', type=int, default=None)
    p.add_argument('--use-new-approximation', default=True, type=bool)
    p.add_argument('--mode-M', type=int, default=4)
    p.add_argument('--res-learning', action='store_true', default=True)
    p.add_argument('--no-res-learning', dest='res_learning', action='store_false')
    # additions of this build
    p.add_argument('--chunk-size', type=int, default=16, help='samples encoded per batch (the cache replay stays in stream order)')
    p.add_argument('--gemm-precision', type=str, default='split', choices=['split', 'f32'])
    p.add_argument('--mmap', action='store_true', help='memory-map the corruption .npy instead of reading it')
    p.add_argument('--replay-mode', type=str, default=None, choices=['owner', 'replicated', 'class', 'delta', 'streams'],
                   help='multi-GPU placement of the cache replay (uni_adapter_amd/stream.py); default owner, or UA_REPLAY_MODE. '
                        "'streams': whole corruption streams are dealt to the ranks (independent states, no collective on the data path)")
    args = p.parse_args(argv)
    if args.replay_mode:
        os.environ['UA_REPLAY_MODE'] = args.replay_mode
    return args


def random_seed(seed=42, rank=0):
    torch.manual_seed(seed + rank); np.random.seed(seed + rank); random.seed(seed + rank)          # main_test-time.py:18-23


def run(args, model=None):
    """Returns {corruption: top-1}.  `model` may be passed in (tests); otherwise it is loaded from the checkpoint flags."""
    if args.name is None:
        args.name = datetime.now().strftime('%Y_%m_%d-%H_%M_%S')
    log_dir = os.path.join(args.output_dir, args.name)
    os.makedirs(log_dir, exist_ok=True)
    logging.basicConfig(level=logging.INFO, format='%(asctime)s | %(levelname)s | %(message)s',
                        handlers=[logging.FileHandler(os.path.join(log_dir, 'out.log')), logging.StreamHandler(sys.stdout)], force=True)
    random_seed(args.seed, 0)
    logging.info(f'Running Experiment: {args.name}')
    logging.info(f'Args: {args}')
    if model is None:
        model = load_point_encoder(args, device=args.device)
    hp = get_hyperparams(args.dataset_name)                                                       # main_test-time.py:42-43
    logging.info(f'Hyperparameters: {hp}')
    corruptions = CORRUPTIONS if args.corruption == 'all' else [c for c in args.corruption.split(',') if c]
    dist_on = torch.distributed.is_available() and torch.distributed.is_initialized()
    world = torch.distributed.get_world_size() if dist_on else 1
    rank = torch.distributed.get_rank() if dist_on else 0
    by_stream = world > 1 and os.environ.get('UA_REPLAY_MODE', 'owner') == 'streams'
    if by_stream:
        # the reference rebuilds all adaptation state per corruption (Uni_Adapter.py:336-352): the 15 streams are independent
        # units, so rank r takes streams r, r+W, ... whole - exact, and no collective until the results are merged
        corruptions = corruptions[rank::world]
        logging.info(f'rank {rank} of {world}: corruption streams {corruptions}')
    results = {}
    for corr in corruptions:
        args.corruption = corr
        logging.info(f"\n{'=' * 20} Processing Corruption: {corr} {'=' * 20}")
        dataset = load_tta_dataset(args)
        loader = torch.utils.data.DataLoader(dataset, batch_size=args.batch_size, shuffle=False, num_workers=args.workers,
                                             pin_memory=True, drop_last=False)
        result = test_zeroshot_3d_core(test_loader=loader, validate_dataset_name=args.validate_dataset_name, model=model,
                                       clip_model=None, tokenizer=None, args=args, hp=hp)
        results[corr] = result['acc1']
    if by_stream:
        parts = [None] * world
        torch.distributed.all_gather_object(parts, results)
        results = {k: v for part in parts for k, v in part.items()}
    logging.info(f'Summary of Results: {results}')
    logging.info(f'Average Top-1: {np.mean(list(results.values())):.3f}')
    return results


def main(argv=None):
    """Single process, or one rank of `python -m torch.distributed.run --nproc-per-node N -m uni_adapter_amd.main_test_time ...`
    (one process per GPU, RCCL): every rank reads the same stream, `TtaEngine` shards it (stream.py)."""
    args = parse_args(argv)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world > 1:
        rank, local = int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0'))
        args.device = f'cuda:{local % max(1, torch.cuda.device_count())}'
        torch.cuda.set_device(args.device)
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        backend = os.environ.get('UA_DIST_BACKEND', 'nccl')
        if backend == 'nccl':
            torch.distributed.init_process_group('nccl', device_id=torch.device(args.device))
        else:
            torch.distributed.init_process_group(backend)
        args.name = f"{args.name or datetime.now().strftime('%Y_%m_%d-%H_%M')}_rank{rank}"      # one log directory per rank
    try:
        run(args)
    finally:
        if world > 1 and torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
