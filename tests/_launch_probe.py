"""A stand-in for bench.py in tests/test_launch.py: one rank of a torch.distributed.run job on the CPU (gloo)."""
import argparse
import json
import os
import sys

import torch
import torch.distributed as dist

ap = argparse.ArgumentParser()
ap.add_argument('--fail-rank', type=int, default=-1)
ap.add_argument('--lie', action='store_true')
ap.add_argument('--silent', action='store_true')
args = ap.parse_args()
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
dist.init_process_group('gloo')
t = torch.ones(1)
dist.all_reduce(t)
dist.barrier()
if rank == args.fail_rank:
    sys.exit(7)
if rank == 0 and not args.silent:
    print('some log line')
    print(json.dumps({'n_gpus': 1 if args.lie else world, 'ranks_in_collective': int(t.item())}), flush=True)
dist.destroy_process_group()
