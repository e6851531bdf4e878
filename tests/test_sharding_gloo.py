"""world_size-2 run on CPU (gloo): shards are independent (a rank's forcing and its
oracle results for its global point ids equal the corresponding slice of the
single-process run) and the bench's only collectives (barrier, MAX of the elapsed
time) behave."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_helpers as oh
from roadsurf_amd import abi, sharding

N_TOTAL, L = 64, 241


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    off, cnt = sharding.strong_shard(N_TOTAL, world, rank)
    f = oh.synth_forcing(cnt, L, seed=5, point_offset=off)
    s = abi.default_settings(L); p = abi.default_parameters(); l = abi.default_local(); l.InitLenI = 1
    out, _, _ = oh.run_oracle("port", f, s, p, l, nthreads=1)
    dist.barrier()
    t = sharding.max_over_ranks(1.0 + rank, dist)
    q.put((rank, off, cnt, out["tsurf"], f["tair"], t))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shards_match_single_process():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    f = oh.synth_forcing(N_TOTAL, L, seed=5)
    s = abi.default_settings(L); p = abi.default_parameters(); l = abi.default_local(); l.InitLenI = 1
    whole, _, _ = oh.run_oracle("port", f, s, p, l)
    covered = np.zeros(N_TOTAL, bool)
    for rank, off, cnt, tsurf, tair, t in got:
        assert np.array_equal(tair, f["tair"][off:off + cnt])
        assert np.array_equal(tsurf, whole["tsurf"][off:off + cnt])
        assert t == float(world)  # MAX over ranks of (1 + rank)
        covered[off:off + cnt] = True
    assert covered.all()
