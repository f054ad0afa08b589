#!/usr/bin/env python3
"""Config 5 of BASELINE.json: SSB star joins Q2.1 / Q4.1 (hyrise_amd/ssb.py) at a scale factor on one GPU, or -- under
torch.distributed.run -- with lineorder chunk-sharded over the ranks, the dimensions replicated and the groups combined by
the sharded AggregateHash (fixed-slot all-reduce); `--repartition` sends lineorder x part through the hash repartition.
    python tools/ssb_bench.py --sf 30
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/ssb_bench.py --sf 30"""
import argparse
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sf", type=float, default=30.0)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--verify", action="store_true", help="compare with SQLite (small scale factors only)")
    args = ap.parse_args()
    rank, world, local_rank = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))
    share_gpu = bool(os.environ.get("HY_BENCH_SHARE_GPU"))
    if share_gpu:
        local_rank = 0
    import torch
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo" if share_gpu else "nccl")
    from hyrise_amd import ssb
    out = ssb.bench(args.sf, args.steps, world, rank, dist, share_gpu, local_rank, args.verify)
    if rank == 0:
        for entry in out.values():
            if isinstance(entry, dict):
                entry.pop("_rows", None)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
