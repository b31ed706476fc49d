"""CPU, world_size 2 over gloo: the N > 1 plumbing of the sampling path (contiguous slice shards, one
flattened parameter broadcast at load, MAX-over-ranks timing, metric reduction).  No HIP compute here."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

from conftest import PKG, REPO


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    for p in (REPO, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from oracle import mudiff_oracle as O
    from mudiff_hip import distributed as D
    from backbones.ncsnpp_generator_adagn_feat import NCSNpp, NCSNpp_adaptive
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        cfg = O.default_config(image_size=32, num_channels_dae=16, ch_mult=[1, 2], attn_resolutions=(), num_res_blocks=1)
        torch.manual_seed(100 + rank)                      # different random init on every rank
        g1, g2 = NCSNpp(cfg), NCSNpp_adaptive(cfg)
        if rank == 0:                                      # only rank 0 "reads the checkpoint"
            g1.load_state_dict(O.make_state_dict(cfg, 'g1', 3))
            g2.load_state_dict(O.make_state_dict(cfg, 'g2', 3))
        nbytes = D.broadcast_parameters(g1) + D.broadcast_parameters(g2)
        ok = all(torch.equal(v, O.make_state_dict(cfg, 'g1', 3)[k]) for k, v in g1.state_dict().items())
        ok = ok and all(torch.equal(v, O.make_state_dict(cfg, 'g2', 3)[k]) for k, v in g2.state_dict().items())
        lo, hi = D.shard_range(37, rank, world)
        tmax = D.max_over_ranks(1.0 + rank, 'cpu')
        sums = D.sum_over_ranks([float(hi - lo), 1.0], 'cpu')
        q.put((rank, ok, nbytes, (lo, hi), tmax, sums))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_broadcast_and_sharding():
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), 'parameters differ after the broadcast'
    assert res[0][2] == res[1][2] > 0
    assert res[0][3] == (0, 19) and res[1][3] == (19, 37)            # contiguous, covers everything once
    assert all(r[4] == 2.0 for r in res)                              # MAX over ranks
    assert all(r[5] == [37.0, 2.0] for r in res)


@pytest.mark.parametrize('n,world', [(0, 4), (1, 8), (7, 8), (8, 8), (4650, 8), (512, 3)])
def test_shard_ranges_partition(n, world):
    from mudiff_hip.distributed import shard_range
    edges = [shard_range(n, r, world) for r in range(world)]
    assert edges[0][0] == 0 and edges[-1][1] == n
    assert all(a[1] == b[0] for a, b in zip(edges, edges[1:]))
    sizes = [hi - lo for lo, hi in edges]
    assert max(sizes) - min(sizes) <= 1
