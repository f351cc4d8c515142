"""CPU, world_size 2, gloo: the multi-GPU path of bench.py.  Rays are independent, ranks own
disjoint batches and the only exchange is the SUM all-reduce of the parameter gradient, plus the
MAX-over-ranks of the timed region.  The per-rank "step" here is the CPU oracle (test
infrastructure) on a small ray batch; the check is that the all-reduced gradient equals the
gradient of the union batch computed in one process."""
import os
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

from conftest import ROOT

R_PER_RANK = 256


def _local_grad(rank):
    """d/d(sigma scale) of sum of opacities over this rank's rays (analytic, via the oracle)."""
    sys.path.insert(0, ROOT)
    import bench
    from oracle import oracle as O
    o, d = bench.make_rays(R_PER_RANK, "random", rank=rank)
    b = bench.make_grid(32, "iid10")
    aabb = np.array([[-1, -1, -1, 1, 1, 1]], np.float32)
    ri, ts, te = O.occgrid_sampling(o, d, b, aabb, render_step_size=0.02)
    pi = O.pack_info(ri, R_PER_RANK)
    sig = (4.0 * (0.5 + 0.5 * np.sin(20.0 * (ts + te)))).astype(np.float32)
    # loss = sum_k w_k ; dL/ds at s = 1 for sigma = s * sig
    g_sigma = O.render_weight_from_density_backward(ts, te, sig, pi, np.ones_like(sig))
    return float((g_sigma * sig).sum()), int(ri.size)


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import bench
    w, r, lr = bench.init_distributed("gloo")
    assert (w, r) == (world, rank)
    # BASELINE cfg 4: the occupancy grid is built on rank 0 only and broadcast bit-packed; every rank must end up with the
    # same torch.bool grid (here 32^3; the bench uses 256^3)
    shared = bench.shared_grid(torch.device("cpu"), 32, "shell10", rank, w)
    grid_sha = __import__("hashlib").sha256(shared.numpy().tobytes()).hexdigest()
    # grid maintenance across ranks: every rank updates its occupancies from its own samples (here: its own field); with
    # OccGridEstimator.sync_group set (opt-in: it makes _update a collective) the evaluated occupancies are MAX-all-reduced
    # and the decay is applied to the union of the sampled cells -> identical occs / binaries everywhere
    import nerfacc_amd as na
    torch.manual_seed(100 + rank)
    est = na.OccGridEstimator(roi_aabb=[-1.0, -1.0, -1.0, 1.0, 1.0, 1.0], resolution=12, levels=1)
    assert est.sync_group is None                       # the default communicates nothing (the reference is single-device)
    est._update(step=0, occ_eval_fn=lambda x: (x[:, rank:rank + 1] > 0).float() * 0.5, occ_thre=0.01, ema_decay=0.95)
    local_frac = float(est.binaries.float().mean())     # one half-space: this rank's own field only
    est = na.OccGridEstimator(roi_aabb=[-1.0, -1.0, -1.0, 1.0, 1.0, 1.0], resolution=12, levels=1)
    est.sync_group = torch.distributed.group.WORLD
    est._update(step=0, occ_eval_fn=lambda x: (x[:, rank:rank + 1] > 0).float() * 0.5, occ_thre=0.01, ema_decay=0.95)
    # a second update after the warm-up: every rank samples DIFFERENT cells (its own RNG) and evaluates an empty field; the
    # union of the sampled cells must decay on every rank alike
    est._update(step=1000, occ_eval_fn=lambda x: torch.zeros_like(x[:, :1]), occ_thre=0.01, ema_decay=0.5, warmup_steps=256)
    upd_sha = __import__("hashlib").sha256(est.binaries.numpy().tobytes() + est.occs.numpy().tobytes()).hexdigest()
    upd_frac = float((est.occs > 0).float().mean())
    decayed = float(((est.occs > 0) & (est.occs < 0.5)).float().mean())
    g, n = _local_grad(rank)
    p = torch.nn.Parameter(torch.zeros(2, dtype=torch.float64))
    p.grad = torch.tensor([g, float(n)], dtype=torch.float64)
    bench.allreduce_grads([p], w)
    torch.distributed.barrier()
    dt = bench.max_over_ranks(0.1 * (rank + 1), w)
    q.put((rank, p.grad.tolist(), dt, grid_sha, tuple(shared.shape), upd_sha, upd_frac, local_frac, decayed))
    torch.distributed.destroy_process_group()


def test_two_rank_gradient_allreduce_matches_union_batch():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g0, n0 = _local_grad(0)
    g1, n1 = _local_grad(1)
    sys.path.insert(0, ROOT)
    import bench
    want = __import__("hashlib").sha256(bench.make_grid(32, "shell10").tobytes()).hexdigest()
    assert res[0][5] == res[1][5]                               # same occs and binaries after the synchronised updates ...
    assert 0.70 < res[0][6] < 0.80                              # ... = the union of the two half-spaces (x > 0) | (y > 0)
    assert all(0.45 < r[7] < 0.55 for r in res)                # without sync_group a rank sees its own half-space only
    assert res[0][8] == res[1][8] and res[0][8] > 0.2           # the cells either rank sampled in the second update decayed on both
    for rank, grad, dt, grid_sha, shape, _, _, _, _ in res:
        assert grid_sha == want and shape == (1, 32, 32, 32)   # identical binaries on every rank (= rank 0's grid)
        assert abs(grad[0] - (g0 + g1)) < 1e-9 * max(1.0, abs(g0 + g1))
        assert grad[1] == n0 + n1
        assert abs(dt - 0.2) < 1e-12          # MAX over ranks of the per-rank times
    assert n0 > 0 and n1 > 0 and abs(g0 - g1) > 0  # the two shards really differ
