"""Stub rank program for tests/test_bench_launcher_cpu.py: what bench.py's ranks do around the timed region -- rendezvous,
barrier, max-over-ranks reduction, the ranks_seen count, ONE JSON line from rank 0 -- on gloo, with no kernels.
`--fail-rank R` makes rank R exit non-zero before the rendezvous completes for the others' line."""
import argparse
import json
import os
import sys

import torch
import torch.distributed as dist

ap = argparse.ArgumentParser()
ap.add_argument("--gpus", type=int, default=1)
ap.add_argument("--steps", type=int, default=1)
ap.add_argument("--warmup", type=int, default=0)
ap.add_argument("--fail-rank", type=int, default=-1)
ap.add_argument("--lie", action="store_true", help="print n_gpus: 1 whatever the world size (the launcher must reject it)")
args = ap.parse_args()
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ.get("BENCH_LAUNCHED_RANKS") == str(args.gpus)
if rank == args.fail_rank:
    sys.exit(7)
dist.init_process_group("gloo")
ones = torch.ones(1, dtype=torch.int32)
dist.all_reduce(ones)
t = torch.tensor([1.0 + rank], dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
if rank == 0:
    print("noise before the line")
    print(json.dumps({"metric": "stub", "value": float(world), "n_gpus": 1 if args.lie else world, "ranks_seen": int(ones.item()),
                      "steps": args.steps, "warmup": args.warmup, "ms_per_step": t.item()}), flush=True)
dist.destroy_process_group()
