"""RCCL smoke test of the collectives the N>1 paths use, runnable with ANY world size (also 1, the only size a one-GPU box
allows: RCCL refuses two ranks on one device):
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port 29511 tools/nccl_sanity.py
Checks process-group creation as bench.py does it, all_gather (list and flat forms), broadcast, all_reduce(MAX/SUM), barrier."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    world, rank, local = (int(os.environ.get(k, d)) for k, d in (('WORLD_SIZE', '1'), ('RANK', '0'), ('LOCAL_RANK', '0')))
    dev = torch.device('cuda', local % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29511')
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    import uni_adapter_amd  # noqa: F401
    from uni_adapter_amd import stream
    assert dist.get_backend() == 'nccl'
    x = torch.full((5, 2, 8), float(rank + 1), device=dev)
    parts = stream._all_gather(x, world, None)
    assert len(parts) == world and all(float(p[0, 0, 0]) == r + 1 for r, p in enumerate(parts))
    flat = torch.empty(world, 7, device=dev)
    dist.all_gather_into_tensor(flat, torch.full((7,), float(rank), device=dev))
    assert flat[:, 0].tolist() == [float(r) for r in range(world)]
    n = 4 * world + 1
    counts = stream.shard_counts(n, world)
    local_block = torch.full((counts[rank], 2, 8), float(rank), device=dev)
    full = stream.gather_feature_blocks(local_block, n, rank, world)
    assert full.shape == (n, 2, 8)
    d = stream.gather_deltas(torch.arange(6, device=dev, dtype=torch.float32) + rank, world)
    assert d.shape == (world, 6) and float(d[world - 1, 0]) == world - 1
    t = torch.tensor([3, 1, 4, 1 + rank], device=dev, dtype=torch.int64)
    stream.broadcast_from_owner(t)
    assert t.tolist() == [3, 1, 4, 1]
    m = torch.tensor([float(rank)], device=dev, dtype=torch.float64)
    dist.all_reduce(m, op=dist.ReduceOp.MAX)
    assert float(m) == world - 1
    torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
    if rank == 0:
        print(f'nccl sanity ok: world {world}, backend {dist.get_backend()}, device {torch.cuda.get_device_name(dev)}')
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
