#!/usr/bin/env python3
"""Headline benchmark: rays/sec (fwd+bwd) through a 128^3 occupancy grid, 1M-ray batch.

One "step" = one pass of the hot path over one synthetic batch (cfg 2):
    OccGridEstimator.sampling (traversal + sigma callback + visibility + compaction)
 -> rendering (rgb/sigma callback, fused weights, fused accumulation)
 -> loss.backward()  (+ all-reduce of the 4-float parameter gradient when N > 1)

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  Inputs are resident in HBM before the timed region.  Rays shard
across ranks (independent batches, weak scaling); the only collective is the all-reduce of the
2-float parameter gradient.
"""
from __future__ import annotations

import argparse
import gc
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


# ----------------------------------------------------------------------------- synthetic radiance field (harness)
FIELD_SRC = os.path.join(ROOT, "bench_csrc", "field.hip")
FIELD_LIB = os.path.join(ROOT, "bench_csrc", "libbench_field.so")


def build_field(force: bool = False) -> str:
    """Compile bench_csrc/field.hip (the synthetic field's three streaming kernels; harness code, see its
    header) in-tree.  Rebuilt when the source is newer than the library or its hash changed."""
    import hashlib
    import subprocess
    from nerfacc_amd import _build
    want = hashlib.sha256(open(FIELD_SRC, "rb").read()).hexdigest()
    hp = FIELD_LIB + ".hash"
    if not force and os.path.exists(FIELD_LIB) and os.path.exists(hp) and open(hp).read().strip() == want:
        return FIELD_LIB
    cc = _build.hipcc()
    if cc is None:
        raise RuntimeError("hipcc not found: cannot build the bench's synthetic field")
    tmp = FIELD_LIB + f".{os.getpid()}.tmp"
    subprocess.run([cc, "-O3", "-std=c++17", "-fPIC", "-shared", f"--offload-arch={_build.ARCH}", "-o", tmp, FIELD_SRC],
                   check=True)
    os.replace(tmp, FIELD_LIB)
    with open(hp, "w") as f:
        f.write(want)
    return FIELD_LIB


class NativeField:
    """sigma_base(t) = 4 (1/2 + 1/2 sin(20 (ts + te))); sigma = p0 sigma_base; rgb = p1 ts (grey ramp).
    Same function as `torch_field`, evaluated by bench_csrc/field.hip: one kernel per callback and one
    for the backward to the two parameters, so the step is dominated by the hot path it measures."""

    def __init__(self, params: torch.Tensor, sigma_scale: float = 1.0):
        import ctypes
        self.params = params
        self.sigma_scale = float(sigma_scale)   # density scale of the sampler's no-grad callback (= the initial p0)
        self.lib = ctypes.CDLL(build_field())
        vp, i64 = ctypes.c_void_p, ctypes.c_int64
        self.lib.bf_field_sigma.argtypes = [vp, vp, i64, ctypes.c_float, vp, vp]
        self.lib.bf_field_fwd.argtypes = [vp, vp, i64, vp, vp, vp, vp]
        self.lib.bf_field_bwd.argtypes = [vp, vp, vp, vp, i64, vp, vp]
        self.blocks = int(self.lib.bf_grid_blocks())
        lib, blocks = self.lib, self.blocks
        st = lambda: torch.cuda.current_stream().cuda_stream
        ptr = lambda t: None if t is None else t.data_ptr()

        class Fn(torch.autograd.Function):
            @staticmethod
            def forward(ctx, params, ts, te):
                ts, te = ts.contiguous(), te.contiguous()
                n = ts.numel()
                sigma, rgb = torch.empty_like(ts), torch.empty((n, 3), dtype=ts.dtype, device=ts.device)
                assert lib.bf_field_fwd(ptr(ts), ptr(te), n, ptr(params), ptr(sigma), ptr(rgb), st()) == 0
                ctx.save_for_backward(ts, te)
                return rgb, sigma

            @staticmethod
            def backward(ctx, g_rgb, g_sigma):
                ts, te = ctx.saved_tensors
                g_rgb = None if g_rgb is None else g_rgb.contiguous()
                g_sigma = None if g_sigma is None else g_sigma.contiguous()
                partial = torch.empty((blocks, 2), dtype=torch.float32, device=ts.device)
                assert lib.bf_field_bwd(ptr(ts), ptr(te), ptr(g_sigma), ptr(g_rgb), ts.numel(), ptr(partial), st()) == 0
                return partial.sum(0), None, None

        self.Fn = Fn

    def sigma_fn(self, ts, te, ri):                          # used inside sampling (no grad)
        ts, te = ts.contiguous(), te.contiguous()
        out = torch.empty_like(ts)
        assert self.lib.bf_field_sigma(ts.data_ptr(), te.data_ptr(), ts.numel(), self.sigma_scale, out.data_ptr(),
                                       torch.cuda.current_stream().cuda_stream) == 0
        return out

    def rgb_sigma_fn(self, ts, te, ri):                      # used inside rendering (with grad)
        return self.Fn.apply(self.params, ts, te)


class NativePropField:
    """cfg 3's networks as harness kernels (bench_csrc/field.hip): proposal density p0 exp(-(mid - p1)^2) with a
    gradient to (p0, p1), fine density 5 exp(-2 (mid - 4)^2).  The torch-lambda version of the same functions is ~40
    elementwise launches over (R, 64) tensors per step, 4 ms -- more than the whole native path it feeds."""

    def __init__(self, params: torch.Tensor):
        import ctypes
        self.params = params
        lib = ctypes.CDLL(build_field())
        vp, i64, f32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_float
        lib.bf_prop_fwd.argtypes = [vp, vp, i64, vp, f32, f32, f32, vp, vp]
        lib.bf_prop_bwd.argtypes = [vp, vp, vp, i64, vp, vp, vp]
        self.lib = lib
        blocks = int(lib.bf_grid_blocks())
        st = lambda: torch.cuda.current_stream().cuda_stream

        class Fn(torch.autograd.Function):
            @staticmethod
            def forward(ctx, params, ts, te):
                ts, te = ts.contiguous(), te.contiguous()
                out = torch.empty_like(ts)
                assert lib.bf_prop_fwd(ts.data_ptr(), te.data_ptr(), ts.numel(), params.data_ptr(), 0.0, 0.0, 0.0, out.data_ptr(), st()) == 0
                ctx.save_for_backward(params, ts, te)
                return out

            @staticmethod
            def backward(ctx, g):
                params, ts, te = ctx.saved_tensors
                partial = torch.empty((blocks, 2), dtype=torch.float32, device=ts.device)
                assert lib.bf_prop_bwd(ts.data_ptr(), te.data_ptr(), g.contiguous().data_ptr(), ts.numel(), params.data_ptr(),
                                       partial.data_ptr(), st()) == 0
                return partial.sum(0), None, None

        self.Fn = Fn

    def prop(self, ts, te):
        return self.Fn.apply(self.params, ts, te)

    def fine(self, ts, te):
        ts, te = ts.contiguous(), te.contiguous()
        out = torch.empty_like(ts)
        assert self.lib.bf_prop_fwd(ts.data_ptr(), te.data_ptr(), ts.numel(), None, 5.0, 2.0, 4.0, out.data_ptr(),
                                    torch.cuda.current_stream().cuda_stream) == 0
        return out


class TorchField:
    """The same field with torch elementwise ops (~25 launches, 1.3 ms per step on 32 M samples)."""

    def __init__(self, params: torch.Tensor, sigma_scale: float = 1.0):
        self.params = params
        self.sigma_scale = float(sigma_scale)

    @staticmethod
    def base_sigma(ts, te):
        return 4.0 * (0.5 + 0.5 * torch.sin(20.0 * (ts + te)))

    def sigma_fn(self, ts, te, ri):
        return self.base_sigma(ts, te) * self.sigma_scale

    def rgb_sigma_fn(self, ts, te, ri):
        rgbs = (ts * self.params[1])[:, None].expand(-1, 3)  # grey ramp; made contiguous by rendering()
        return rgbs, self.base_sigma(ts, te) * self.params[0]


# ----------------------------------------------------------------------------- synthetic workload
def make_grid(res: int, variant: str, seed: int = 42) -> np.ndarray:
    rng = np.random.default_rng(seed)
    if variant == "iid10":
        return rng.random((1, res, res, res)) < 0.10
    c = (np.arange(res) + 0.5) / res * 2 - 1
    x, y, z = np.meshgrid(c, c, c, indexing="ij")
    r = np.sqrt(x * x + y * y + z * z)
    shell = (r >= 0.50) & (r <= 0.66)                       # ~8.5 % of the cube
    speckle = rng.random((res, res, res)) < 0.02
    return (shell | speckle)[None]


def make_rays(n_rays: int, variant: str, rank: int = 0, seed: int = 42):
    if variant == "random":
        rng = np.random.default_rng(seed + rank)
        o = rng.standard_normal((n_rays, 3)).astype(np.float32)
        d = rng.standard_normal((n_rays, 3)).astype(np.float32)
        d /= np.linalg.norm(d, axis=-1, keepdims=True)
        return o, d
    # pinhole camera on a ring around the scene (rank-seeded azimuth), looking at the origin,
    # fov chosen so the frustum just covers the [-1,1]^3 box; row-major pixel order.
    side = int(round(math.sqrt(n_rays)))
    assert side * side == n_rays, "image variant needs a square ray count"
    t = 1.0 / 2.2
    u = (np.arange(side, dtype=np.float32) + 0.5) / side * 2 - 1
    px, py = np.meshgrid(u * t, u * t, indexing="xy")
    d_cam = np.stack([px, py, np.ones_like(px)], -1).reshape(-1, 3)
    d_cam /= np.linalg.norm(d_cam, axis=-1, keepdims=True)
    az = 2 * math.pi * rank / 8.0
    rot = np.array([[math.cos(az), 0, math.sin(az)], [0, 1, 0], [-math.sin(az), 0, math.cos(az)]], np.float32)
    d = (d_cam @ rot.T).astype(np.float32)
    o = np.broadcast_to((np.array([0, 0, -3.2], np.float32) @ rot.T), d.shape).copy()
    return o, d


def make_workload(dev, n_rays: int = 1024 * 1024, res: int = 128, grid: str = "shell10", rays: str = "image",
                  rank: int = 0, field: str = "native", sigma_scale: float = 1.0, binaries: torch.Tensor = None):
    """``binaries``: a (1, res, res, res) bool tensor already on the device (the shared grid of a multi-GPU run, broadcast
    from rank 0); otherwise the grid is built here from ``grid`` / seed 42."""
    import nerfacc_amd as na
    o, d = make_rays(n_rays, rays, rank)
    est = na.OccGridEstimator(roi_aabb=[-1.0, -1.0, -1.0, 1.0, 1.0, 1.0], resolution=res, levels=1).to(dev)
    if binaries is None:
        b = make_grid(res, grid)
        binaries = torch.from_numpy(b).to(dev)
    else:
        b = None
    est.binaries = binaries
    est.occs = binaries.reshape(-1).float()
    step = 2 * math.sqrt(3) / 1024                          # <= 1024 samples per ray
    # two scalar parameters (density scale, colour scale): the "network" of this synthetic step
    params = torch.nn.Parameter(torch.tensor([float(sigma_scale), 1.0], device=dev))
    fld = NativeField(params, sigma_scale) if field == "native" else TorchField(params, sigma_scale)
    sigma_fn, rgb_sigma_fn = fld.sigma_fn, fld.rgb_sigma_fn

    return dict(estimator=est, rays_o=torch.from_numpy(o).to(dev), rays_d=torch.from_numpy(d).to(dev),
                binaries_np=b, rays_np=(o, d), step=step, params=params, sigma_fn=sigma_fn, rgb_sigma_fn=rgb_sigma_fn,
                n_rays=n_rays, res=res, sigma_scale=float(sigma_scale))


_DIST_INFO: dict = {}   # what the distributed glue did in this process (reported on the JSON line under "distributed")


def shared_grid(dev, res: int, grid: str, rank: int, world: int) -> torch.Tensor:
    """BASELINE cfg 4: ONE occupancy grid for all ranks.  Rank 0 builds it and broadcasts it once, bit-packed (res^3 / 8
    bytes over RCCL), every rank unpacks it into the torch.bool layout the estimator keeps."""
    n = res ** 3
    packed = torch.empty((n + 7) // 8, dtype=torch.uint8, device=dev)
    if rank == 0:
        packed.copy_(torch.from_numpy(np.packbits(make_grid(res, grid).reshape(-1))).to(dev))
    if world > 1 or _group_live():
        torch.distributed.broadcast(packed, src=0)
    import hashlib
    _DIST_INFO["shared_grid_sha256"] = hashlib.sha256(packed.cpu().numpy().tobytes()).hexdigest()   # what this rank received
    _DIST_INFO["shared_grid_bytes"] = int(packed.numel())
    shifts = torch.arange(7, -1, -1, device=dev, dtype=torch.uint8)
    bits = ((packed[:, None] >> shifts) & 1).reshape(-1)[:n]
    return bits.bool().reshape(1, res, res, res)


def run_step(w, world_size: int = 1, handle=None, prefetch: bool = False):
    """One step.  With ``prefetch`` the traversal of the NEXT batch is issued on the estimator's side stream right
    after this step's rendering / backward were queued (it overlaps with them) and its handle is returned."""
    import nerfacc_amd as na
    est, n = w["estimator"], w["n_rays"]
    ri, ts, te = est.sampling(w["rays_o"], w["rays_d"], sigma_fn=w["sigma_fn"], render_step_size=w["step"],
                              early_stop_eps=1e-4, alpha_thre=0.0, traversal=handle)
    colors, opac, depth, _ = na.rendering(ts, te, ri, n_rays=n, rgb_sigma_fn=w["rgb_sigma_fn"])
    loss = colors.sum()
    w["params"].grad = None
    loss.backward()
    allreduce_grads([w["params"]], world_size)              # RCCL over xGMI: 8 bytes
    nxt = None
    if prefetch:
        nxt = est.prefetch_traversal(w["rays_o"], w["rays_d"], render_step_size=w["step"], wait_for_inputs=False)
    w["last"] = (ri, ts, te, colors)                        # kept for the parity check that follows the timed loops
    return ri.numel(), loss, nxt


# ----------------------------------------------------------------------------- distributed glue
def init_distributed(backend: str, device=None):
    """(world, rank, local_rank) from the torchrun environment; initialises the process group when
    WORLD_SIZE > 1.  backend "nccl" is RCCL on ROCm; the CPU tests use "gloo"."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # under torchrun the process group is created also for a single rank, so that `--nproc-per-node 1` exercises the
    # same RCCL calls (init, all_reduce, barrier) as the multi-GPU launch
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ and "MASTER_PORT" in os.environ
    if (world > 1 or launched) and not torch.distributed.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        # librccl prints a "Librccl path : ..." banner to STDOUT when its communicator is created; stdout carries the
        # one JSON line, so file descriptor 1 points at stderr while the group and its first collective are set up
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            torch.distributed.init_process_group(backend=backend, **kw)
            if backend == "nccl":
                torch.distributed.barrier()
                torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)
    return world, rank, local_rank


def allreduce_grads(params, world: int):
    """Rays are independent, so ranks own disjoint ray batches and the only exchange is the SUM of
    the (tiny) parameter gradient: d/dp sum_over_all_rays = sum over ranks of the local gradient."""
    if world > 1 or _group_live():
        for p in params:
            torch.distributed.all_reduce(p.grad, op=torch.distributed.ReduceOp.SUM)
        _DIST_INFO["grad_allreduces"] = _DIST_INFO.get("grad_allreduces", 0) + len(params)


def _group_live() -> bool:
    return torch.distributed.is_available() and torch.distributed.is_initialized()


def max_over_ranks(seconds: float, world: int, device="cpu") -> float:
    if world == 1 and not _group_live():
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    return float(t.item())


# ----------------------------------------------------------------------------- per-kernel timing (HIP events)
class KernelTimer:
    """Brackets every native call with HIP events on the stream it is launched on."""

    def __init__(self):
        self.records = []

    def install(self):
        import ctypes
        from nerfacc_amd import _backend as B
        self._orig = B.call
        timer = self
        # a ~40 us busy-wait kernel in front of every timed call (bench_csrc/field.hip: bf_spin): the host issues the
        # first event, the call and the second event while the GPU spins, so the event pair brackets the kernel alone and
        # not the launch latency of a call the GPU was already waiting for
        spin_lib = ctypes.CDLL(build_field())
        spin_lib.bf_spin.argtypes = [ctypes.c_longlong, ctypes.c_void_p]

        def timed_call(name, *args):
            key = name
            if name == "nfa_traverse_grids":
                key = "nfa_traverse_grids[mode=%d]" % args[0]._obj.mode
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            spin_lib.bf_spin(4000, torch.cuda.current_stream().cuda_stream)
            e0.record()
            timer._orig(name, *args)
            e1.record()
            timer.records.append((key, e0, e1))

        B.call = timed_call
        self._orig_group = B.call_group

        def timed_group(name, fn):
            # one logical op issued as several launches on two streams: bracket the group on the calling stream (its last
            # instruction there is the wait for the other stream) and keep the per-call events out of it
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            spin_lib.bf_spin(4000, torch.cuda.current_stream().cuda_stream)
            e0.record()
            B.call = timer._orig
            try:
                fn()
            finally:
                B.call = timed_call
            e1.record()
            timer.records.append((name, e0, e1))

        B.call_group = timed_group
        for mod in list(sys.modules.values()):
            if getattr(mod, "__name__", "").startswith("nerfacc_amd") and getattr(mod, "B", None) is B:
                pass  # modules reach call() through the B namespace, nothing else to patch

    def uninstall(self):
        from nerfacc_amd import _backend as B
        B.call = self._orig
        B.call_group = self._orig_group

    def summary(self, steps):
        torch.cuda.synchronize()
        agg = {}
        for key, e0, e1 in self.records:
            ms = e0.elapsed_time(e1)
            a = agg.setdefault(key, [0.0, 0])
            a[0] += ms; a[1] += 1
        return {k: dict(ms_per_step=v[0] / steps, launches_per_step=v[1] / steps, ms_per_launch=v[0] / v[1])
                for k, v in agg.items()}


def algorithmic_bytes(R, M, Mv, res, G=1):
    """compulsory traffic with the API's dtypes, each array touched once."""
    grid = G * res ** 3
    return {
        "nfa_traverse_grids[mode=0]": R * (24 + 8) + grid + R * 8,                 # rays + planes, grid, counts
        # (nfa_traverse_grids[mode=1] is, in the flows timed here, the fill of the FEW rays whose records did not fit: it is
        #  credited with no bytes -- its samples are a fraction of M that the host does not know -- and rated by time alone)
        "nfa_traverse_runs": R * (24 + 8) + grid + R * 8,                            # one DDA walk: rays, grid, counts
        "nfa_traverse_cone_runs": R * (24 + 8) + grid + R * 8,
        "nfa_traverse_cone_walk": R * (24 + 8) + grid + R * 8,
        "nfa_expand_runs": M * (4 + 4 + 8) + R * 16,                                 # the sampler's output, once
        "nfa_expand_cone_runs": M * (4 + 4 + 8) + R * 16,
        "nfa_render_visibility": M * (4 + 4 + 4) + R * 16 + M * 1 + R * 8,
        "nfa_compact_samples": M * (1 + 4 + 4) + R * 24 + Mv * 16,
        "nfa_render_from_density_fwd": Mv * (12 + 12) + R * 16,
        "nfa_render_from_density_bwd": Mv * (4 + 12 + 4) + R * 16,
        "nfa_render_accumulate_fwd": Mv * (4 + 12 + 8) + R * (16 + 20),
        "nfa_render_accumulate_bwd": Mv * (4 + 12 + 8 + 4 + 12) + R * 36,
        # rendering() as one pass each way: (ts, te, sigma, rgb) -> (w, T, alpha) + per-ray (colour, opacity, depth);
        # backward (ts, te, rgb, T, alpha) + per-ray gradients -> (g_sigma, g_rgb)
        "nfa_render_fused_fwd": Mv * (12 + 12 + 12) + R * (16 + 20),
        "nfa_render_fused_bwd": Mv * (8 + 12 + 8 + 4 + 12) + R * (16 + 20),
    }


#: what bounds each native call: the walk is bound by instruction issue, everything else streams
KERNEL_BOUND = {"nfa_traverse_runs": "issue", "nfa_traverse_cone_runs": "issue", "nfa_traverse_cone_walk": "issue",
                "nfa_traverse_grids[mode=0]": "issue"}


def kernel_table(ksum, ab):
    kernels = {}
    for k, v in ksum.items():
        entry = dict(v)
        if k in ab:
            per_launch = ab[k] / max(v["launches_per_step"], 1e-9)
            entry["algorithmic_bytes_per_launch"] = per_launch
            entry["achieved_GBps"] = per_launch / (v["ms_per_launch"] * 1e-3) / 1e9
            entry["frac_of_hbm_peak"] = entry["achieved_GBps"] / HBM_PEAK_GBPS
            entry["bound"] = KERNEL_BOUND.get(k, "hbm")
        kernels[k] = entry
    return kernels


def timed_steps(fn, steps, warmup, with_kernels=True, per_step=None):
    """(seconds per step, per-native-call HIP-event table) of `fn` -- used for the secondary configurations; the headline
    loop in main() is timed without the per-call events.  `per_step` (a list): every step is timed by itself (a
    synchronisation after each: for steps of many milliseconds), the list receives the times and the MEDIAN is returned -- a
    configuration that is given three steps must not report one allocator stall as half of its figure."""
    for _ in range(warmup):
        out = fn()
    gc.collect(); gc.disable()
    torch.cuda.synchronize()
    if per_step is not None:
        for _ in range(steps):
            t0 = time.perf_counter()
            out = fn()
            torch.cuda.synchronize()
            per_step.append(time.perf_counter() - t0)
        dt = float(np.median(per_step))
    else:
        t0 = time.perf_counter()
        for _ in range(steps):
            out = fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
    gc.enable()
    ks = {}
    if with_kernels:
        timer = KernelTimer(); timer.install()
        try:
            for _ in range(steps):
                out = fn()
            ks = timer.summary(steps)
        finally:
            timer.uninstall()
    return dt, ks, out


# ----------------------------------------------------------------------------- secondary configurations (extra keys)
def extra_cfg2_variant(dev, args, **kw):
    """cfg 2 with one thing changed: `sigma_scale=16` makes early termination bite (T < 1e-4 inside the shell), so the
    visibility mask drops samples and nfa_compact_samples runs in the step; `rays="random"` is a training-style batch of
    unrelated rays."""
    w = make_workload(dev, args.rays, args.res, args.grid, kw.get("rays", args.ray_variant), 0, args.field,
                      sigma_scale=kw.get("sigma_scale", 1.0))
    est = w["estimator"]
    ri, ts, te, pi = __import__("nerfacc_amd").grid._traverse_samples(
        w["rays_o"], w["rays_d"], est.binaries, est.aabbs, torch.zeros(args.rays, device=dev),
        torch.full((args.rays,), 1e10, device=dev), w["step"], 0.0)
    M = int(ri.numel())
    del ri, ts, te, pi
    dt, ks, _ = timed_steps(lambda: run_step(w, 1)[0], max(3, args.steps // 2), 3)
    Mv = int(w["last"][0].numel())
    kernels = kernel_table(ks, algorithmic_bytes(args.rays, M, Mv, args.res))
    out = {"ms_per_step": dt * 1e3, "rays_per_s": args.rays / dt, "samples_before_compaction": M,
           "samples_after_compaction": Mv, "kernels": {k: {kk: vv for kk, vv in v.items() if kk != "launches_per_step"}
                                                       for k, v in kernels.items()}}
    if "nfa_compact_samples" in kernels:
        c = kernels["nfa_compact_samples"]
        out["compaction"] = {"us_per_launch": c["ms_per_launch"] * 1e3, "achieved_GBps": c.get("achieved_GBps"),
                             "kept_fraction": Mv / max(M, 1)}
    return out


def extra_cfg4_per_rank(dev, args, res=256):
    """What ONE rank of `bench.py --gpus N` (N > 1: BASELINE cfg 4) does per step, on this GPU: the same 1024 x 1024 rays through
    cfg 4's shared 256^3 grid.  The N = 1 line itself is cfg 2 (128^3, the metric's configuration), whose walk crosses half as
    many cells: the scaling efficiency of cfg 4 is value(N) / (N x this rate), not value(N) / (N x value(1))."""
    binaries = shared_grid(dev, res, args.grid, 0, 1)
    w = make_workload(dev, args.rays, res, args.grid, "image", 0, args.field, binaries=binaries)
    dt, ks, _ = timed_steps(lambda: run_step(w, 1)[0], max(3, args.steps // 2), 3)
    return {"workload": f"cfg 4's per-rank step on one GPU: {args.rays} image rays, shared {res}^3 {args.grid} grid, sampling + rendering fwd + bwd",
            "ms_per_step": dt * 1e3, "rays_per_s": args.rays / dt, "samples": int(w["last"][0].numel()),
            "native_ms_per_step": sum(v["ms_per_step"] for v in ks.values())}


def _cfg3_fields(p0=3.0, p1=4.0):
    """numpy twins of the cfg-3 density callbacks (NativePropField / the torch lambdas of extra_cfg3) and d sigma / d p."""
    f = np.float32
    prop = lambda ts, te: (np.exp(-((ts + te) * f(0.5) - f(p1)) ** 2) * f(p0)).astype(np.float32)
    fine = lambda ts, te: (np.exp(-((ts + te) * f(0.5) - f(4.0)) ** 2 * f(2.0)) * f(5.0)).astype(np.float32)
    dprop = lambda ts, te, sig: (sig / f(p0), sig * f(2.0) * ((ts + te) * f(0.5) - f(p1)))
    return prop, fine, dprop


def _cfg3_oracle_step(O, R, p0=3.0, p1=4.0, backward=True):
    """One cfg-3 step restated by the oracle: PropNetEstimator.sampling's level loop (ref estimators/prop_net.py:38-129), the fine
    transmittance, compute_loss (:131-154) and its backward to the proposal density's two parameters.
    Returns (t_starts, t_ends, loss, grad_p)."""
    prop, fine, dprop = _cfg3_fields(p0, p1)
    ts, te, levels, fvals = O.propnet_sampling([prop, prop], [64, 64], 16, R, 2.0, 6.0, sampling_type="uniform", stratified=False,
                                               return_final_vals=True)
    trans = O.batched_transmittance_from_density(ts, te, fine(ts, te))
    if not backward:
        return ts, te, None, None
    loss, g_cdfs = O.propnet_loss(levels, fvals, trans)
    gp = np.zeros(2, np.float64)
    for (vals, _), g in zip(levels, g_cdfs):
        t = O.transform_stot("uniform", vals, 2.0, 6.0)
        a, b = t[..., :-1], t[..., 1:]
        sig = prop(a, b)
        g_sig = O.density_cdf_backward(a, b, sig, g)
        d0, d1 = dprop(a, b, sig)
        gp += [float((g_sig * d0).astype(np.float64).sum()), float((g_sig * d1).astype(np.float64).sum())]
    return ts, te, loss, gp


def _cfg3_cpu_baseline(R=1 << 17, min_seconds=4.0):
    """cfg 3 on the host, forward AND backward like the GPU figure: the oracle's restatement of PropNetEstimator.sampling's level
    loop (2 -> 64 -> 64 -> 16, uniform; ref estimators/prop_net.py:38-129, resampling = pdf.cu's kernels in C/OpenMP, the batched
    transmittance and the s -> t map in numpy as the reference's own CPU path is torch elementwise), the fine transmittance,
    compute_loss and its backward to the proposal parameters (oracle.propnet_loss / density_cdf_backward), on a bounded
    sample of rays."""
    from oracle import oracle as O
    O.build()
    total, reps = 0.0, 0
    while total < min_seconds and reps < 50:
        t0 = time.perf_counter()
        _cfg3_oracle_step(O, R)
        total += time.perf_counter() - t0
        reps += 1
    return dict(value=R * reps / total, unit="rays/s", cores=O.max_threads(), kind="port", cpu_model=cpu_model(),
                sample=f"{reps} passes over {R} rays (of the GPU's {1 << 20}), forward + proposal-loss backward, {total:.1f} s; C/OpenMP "
                       f"resampling and searchsorted, numpy (one thread) for the elementwise s -> t map, densities, transmittance, loss")


def extra_cfg2_testmode(dev, args, n_img=3, parity=True):
    """The a12 test-mode loop (render_rays_test_mode; ref examples/utils.py:252-425) on the headline scene: one 1024 x 1024
    image through the 128^3 grid, constant step (cone_angle 0: one sample per alive ray and iteration to begin with), max_samples
    1024, early_stop_eps 1e-4 -- the synthetic-scene regime; cfg 5 carries the unbounded-scene one.  Parity: a 2048-ray
    subset (its own image: the schedule depends on the ray count) against oracle.test_mode_loop."""
    from nerfacc_amd.marching import render_rays_test_mode
    w = make_workload(dev, args.rays, args.res, args.grid, "image", 0, args.field)
    est, step = w["estimator"], w["step"]
    kw = dict(render_step_size=step, early_stop_eps=1e-4)
    with torch.no_grad():
        render_rays_test_mode(1024, w["rgb_sigma_fn"], est, w["rays_o"], w["rays_d"], **kw)            # warm-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_img):
            rgb, opa, dep, total = render_rays_test_mode(1024, w["rgb_sigma_fn"], est, w["rays_o"], w["rays_d"], **kw)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n_img
    R = w["n_rays"]
    out = {"workload": f"render_rays_test_mode, cfg 2's scene: {R} image rays, {args.res}^3 {args.grid}, step {step:.6f}, cone 0, "
                       f"max_samples 1024, early_stop_eps 1e-4",
           "ms_per_image": dt * 1e3, "rays_per_s": R / dt, "total_samples": int(total),
           "note": "ms_per_image: the exact-shape loop (the callback sees exactly the iteration's samples; one host read per iteration since round 4, two before); padded: the same loop with fixed "
                   "shapes, the schedule on the device and one iteration replayed as a hipGraph (no host read inside)"}
    try:   # the loop without the host in it (nerfacc_amd/marching.py: PaddedTestModeLoop): wall time and GPU-busy time
        from nerfacc_amd.marching import PaddedTestModeLoop
        with torch.no_grad():
            loop = PaddedTestModeLoop(1024, w["rgb_sigma_fn"], est, w["rays_o"], w["rays_d"], 0.0, 1e10, step, 0.0, 1e-4)
            p_rgb, p_opa, p_dep, p_total = loop.render(None)                                             # warm-up + capture
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record()
            for _ in range(n_img):
                p_rgb, p_opa, p_dep, p_total = loop.render(None)
            e1.record()
            torch.cuda.synchronize()
            dtp = (time.perf_counter() - t0) / n_img
        out["padded"] = {"ms_per_image": dtp * 1e3, "rays_per_s": R / dtp, "gpu_busy_ms_per_image": e0.elapsed_time(e1) / n_img,
                         "iterations_run": loop.iterations_run, "iterations_queued": loop.iterations_queued,
                         "total_samples": int(p_total),
                         "identical_to_exact_loop": bool(int(p_total) == int(total) and torch.equal(p_rgb, rgb) and torch.equal(p_opa, opa)
                                                         and torch.equal(p_dep, dep)),
                         "gpu_busy_note": "HIP events around the images' replays on the stream they run on; the host queues "
                                          "iterations ahead of the GPU, so the stream has no idle gaps between them"}
        del loop
    except Exception as e:
        out["padded"] = {"error": repr(e)}
    if parity:
        try:
            from oracle import oracle as O
            O.build()
            o, d = w["rays_np"]
            sub = np.arange(0, R, R // 2048)[:2048]
            o2, d2 = np.ascontiguousarray(o[sub]), np.ascontiguousarray(d[sub])
            with torch.no_grad():
                g_rgb, g_opa, g_dep, g_total = render_rays_test_mode(1024, w["rgb_sigma_fn"], est, torch.from_numpy(o2).to(dev),
                                                                     torch.from_numpy(d2).to(dev), **kw)
            ss = np.float32(w["sigma_scale"])
            field_np = lambda ts, te, ri: (np.repeat(ts[:, None], 3, 1).astype(np.float32),
                                           (ss * (4.0 * (0.5 + 0.5 * np.sin(20.0 * (ts + te))))).astype(np.float32))
            orgb, oopa, odep, ototal, tinfo = O.test_mode_loop(1024, field_np, o2, d2, w["binaries_np"], est.aabbs.cpu().numpy(), **kw)
            e_rgb = float(np.abs(g_rgb.cpu().numpy() - orgb).max()); e_opa = float(np.abs(g_opa.cpu().numpy() - oopa).max())
            guarded = int(tinfo["guard_rays"].sum())
            ok = (g_total == ototal or guarded > 0) and e_rgb <= 2e-5 * max(1.0, float(np.abs(orgb).max())) and e_opa <= 2e-5
            out["parity"] = {"checked": bool(ok), "rays": int(sub.size), "total_samples": int(g_total), "oracle_total_samples": int(ototal),
                             "rays_on_the_termination_threshold": guarded, "rgb_max_abs_err": e_rgb, "opacity_max_abs_err": e_opa,
                             "iterations": int(tinfo["iterations"])}
        except Exception as e:  # the check must never cost the timing
            out["parity_error"] = repr(e)
    return out


def extra_cfg3(dev, R, steps, field="native", cpu_base=True):
    """BASELINE cfg 3: PropNetEstimator, 2 proposal levels 64 -> 64 -> 16, uniform, fwd + proposal-loss backward."""
    import nerfacc_amd as na
    p = torch.nn.Parameter(torch.tensor([3.0, 4.0], device=dev))
    est = na.PropNetEstimator(optimizer=torch.optim.SGD([p], lr=1e-3)).to(dev)
    if field == "native":
        fld = NativePropField(p)
        prop, fine = fld.prop, fld.fine
    else:
        prop = lambda ts, te: torch.exp(-((ts + te) * 0.5 - p[1]) ** 2) * p[0]          # proposal density, 2 parameters
        fine = lambda ts, te: torch.exp(-((ts + te) * 0.5 - 4.0) ** 2 * 2.0) * 5.0

    def step():
        ts, te = est.sampling([prop, prop], [64, 64], 16, R, 2.0, 6.0, sampling_type="uniform", stratified=False,
                              requires_grad=True)
        trans, _ = na.render_transmittance_from_density(ts, te, fine(ts, te))
        return est.update_every_n_steps(trans, requires_grad=True)

    dt, ks, loss = timed_steps(step, steps, 2)
    # algorithmic bytes: importance_sampling R (8 E_in + 4 (S + 1)); the fused s -> t variant also writes 2 S floats
    ab = {"nfa_importance_sampling_t": R * ((8 * 2 + 4 * 65 + 8 * 64) + (8 * 65 + 4 * 65 + 8 * 64) + (8 * 65 + 4 * 17 + 8 * 16)),
          "nfa_pdf_loss_fwd": R * (4 * (17 + 17 + 65 + 65) + 4 * 16 + 4 * 16) * 2,
          "nfa_pdf_loss_bwd": R * (4 * (17 + 65) + 4 * 16 + 4 * 16 + 4 * 65) * 2}
    kernels = {}
    for k, v in ks.items():
        e = {"ms_per_step": v["ms_per_step"], "ms_per_launch": v["ms_per_launch"], "launches_per_step": v["launches_per_step"]}
        if k in ab:
            e["algorithmic_bytes_per_step"] = ab[k]
            e["achieved_GBps"] = ab[k] / (v["ms_per_step"] * 1e-3) / 1e9
            e["frac_of_hbm_peak"] = e["achieved_GBps"] / HBM_PEAK_GBPS
        kernels[k] = e
    cpu = None
    if cpu_base:
        try:
            cpu = _cfg3_cpu_baseline()
        except Exception as e:  # must never cost the timing
            cpu = {"error": repr(e)}
    # parity of the timed configuration: one more step at the parameters training has reached, every n-th ray restated by
    # the oracle (final samples, loss, gradient of the loss to the two proposal parameters)
    parity = None
    if cpu_base:
        try:
            from oracle import oracle as O
            O.build()
            p0, p1 = (float(v) for v in p.detach().cpu().tolist())
            ts, te = est.sampling([prop, prop], [64, 64], 16, R, 2.0, 6.0, sampling_type="uniform", stratified=False, requires_grad=True)
            trans, _ = na.render_transmittance_from_density(ts, te, fine(ts, te))
            p.grad = None
            g_loss = est.compute_loss(trans)
            g_loss.backward()
            g_grad = p.grad.detach().double().cpu().numpy()
            n_sub = 4096
            sub = torch.arange(0, R, max(1, R // n_sub), device=dev)[:n_sub]
            o_ts, o_te, o_loss, o_grad = _cfg3_oracle_step(O, int(sub.numel()), p0, p1)
            e_t = max(float(np.abs(ts[sub].detach().cpu().numpy() - o_ts).max()), float(np.abs(te[sub].detach().cpu().numpy() - o_te).max()))
            # (every ray of this synthetic batch sees the same 1-D problem, so the subset's mean loss is the batch's; the
            #  gradient is a sum over rays divided by the ray count inside the mean: the same)
            e_l = abs(float(g_loss) - o_loss) / max(abs(o_loss), 1e-30)
            e_g = float(np.abs(g_grad - o_grad).max() / max(np.abs(o_grad).max(), 1e-30))
            parity = {"checked": True, "rays_restated": int(sub.numel()), "params": [p0, p1], "max_abs_err_t": e_t, "t_range": 4.0,
                      "tol_t": 1e-5 * 4.0, "loss_gpu": float(g_loss), "loss_oracle": o_loss, "rel_err_loss": e_l, "tol_loss": 1e-5,
                      "grad_gpu": g_grad.tolist(), "grad_oracle": o_grad.tolist(), "rel_err_grad": e_g, "tol_grad": 1e-4,
                      "ok": bool(e_t <= 4e-5 and e_l <= 1e-5 and e_g <= 1e-4)}
        except Exception as e:
            parity = {"checked": False, "error": repr(e)}
    return {"cpu_baseline": cpu, "parity": parity,
            "workload": f"cfg3: PropNetEstimator 2 -> 64 -> 64 -> 16, R={R}, uniform, fwd + proposal-loss bwd, {field} proposal / fine "
                        f"density callbacks",
            "ms_per_step": dt * 1e3, "rays_per_s": R / dt, "loss": float(loss),
            "native_ms_per_step": sum(v["ms_per_step"] for v in ks.values()), "kernels": kernels}


def extra_cfg5(dev, R, steps, res=512, G=4, parity=True, train=True):
    """BASELINE cfg 5: G nested res^3 levels, R rays from inside the level-0 box, step 1e-3, cone 0.004, near 0.2,
    alpha_thre 1e-2, early_stop_eps 1e-4: train-mode sampling + rendering fwd + bwd."""
    import nerfacc_amd as na
    rng = np.random.default_rng(5)
    est = na.OccGridEstimator(roi_aabb=[-1.0, -1.0, -1.0, 1.0, 1.0, 1.0], resolution=res, levels=G).to(dev)
    ax = (torch.arange(res, device=dev, dtype=torch.float32) + 0.5) / res * 2 - 1
    r = torch.sqrt(ax[:, None, None] ** 2 + ax[None, :, None] ** 2 + ax[None, None, :] ** 2)
    shell = (r > 0.5) & (r < 0.66)
    g = torch.Generator(device=dev); g.manual_seed(5)
    b = torch.stack([shell | (torch.rand((res, res, res), device=dev, generator=g) < 0.02) for _ in range(G)])
    est.binaries = b
    est.occs = b.reshape(-1).float()
    if os.environ.get("NFA_BENCH_CFG5_BIN", "") in ("0", "1"):                      # (A/B of the ray binning; default: the library decides)
        est.bin_rays = os.environ["NFA_BENCH_CFG5_BIN"] == "1"
    del r, shell
    o = (rng.random((R, 3)).astype(np.float32) - 0.5)                               # inside the level-0 box
    d = rng.standard_normal((R, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=-1, keepdims=True)
    rays_o, rays_d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    params = torch.nn.Parameter(torch.tensor([1.0, 1.0], device=dev))
    fld = NativeField(params)
    last = {}

    def step():
        ri, ts, te = est.sampling(rays_o, rays_d, sigma_fn=fld.sigma_fn, near_plane=0.2, render_step_size=1e-3,
                                  cone_angle=0.004, alpha_thre=1e-2, early_stop_eps=1e-4)
        colors, opac, depth, _ = na.rendering(ts, te, ri, n_rays=R, rgb_sigma_fn=fld.rgb_sigma_fn)
        params.grad = None
        colors.sum().backward()
        last["m"] = ri.numel()
        last["out"] = (ri, ts, te, colors)
        return ri.numel()

    out = {"workload": f"cfg5: {G} nested {res}^3 levels, R={R} rays from inside, step 1e-3, cone 0.004, near 0.2, "
                       f"alpha_thre 1e-2, sampling + rendering fwd + bwd"}
    if train:
        step_s: list = []
        dt, ks, m = timed_steps(step, max(steps, 5), 3, per_step=step_s)
        ri, ts, te, pi = na.grid._traverse_samples(rays_o, rays_d, est.binaries, est.aabbs, torch.full((R,), 0.2, device=dev),
                                                   torch.full((R,), 1e10, device=dev), 1e-3, 0.004)
        M = int(ri.numel())
        del ri, ts, te, pi
        kernels = kernel_table(ks, algorithmic_bytes(R, M, int(m), res, G))
        out.update({"ms_per_step": dt * 1e3, "ms_per_step_is": "median of the steps timed one by one", "ms_each_step": [round(x * 1e3, 2) for x in step_s],
                    "rays_per_s": R / dt, "samples_before_compaction": M, "samples_after_compaction": int(m),
                    "native_ms_per_step": sum(v["ms_per_step"] for v in ks.values()),
                    "kernels": {k: {kk: vv for kk, vv in v.items() if kk != "launches_per_step"} for k, v in kernels.items()}})
    # ---- the a12 test-mode loop on the same scene (SURVEY 8d cfg 5: "run both sampling (train) and the test-mode loop")
    from nerfacc_amd.marching import render_rays_test_mode
    tm_kw = dict(near_plane=0.2, render_step_size=1e-3, cone_angle=0.004, alpha_thre=1e-2, early_stop_eps=1e-4)
    with torch.no_grad():
        render_rays_test_mode(1024, fld.rgb_sigma_fn, est, rays_o, rays_d, **tm_kw)            # warm-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n_img = max(1, steps // 2)
        for _ in range(n_img):
            rgb_t, opa_t, dep_t, total_t = render_rays_test_mode(1024, fld.rgb_sigma_fn, est, rays_o, rays_d, **tm_kw)
        torch.cuda.synchronize()
        dt_img = (time.perf_counter() - t0) / n_img
    out["test_mode_loop"] = {"workload": f"render_rays_test_mode (examples/utils.py:252-425), max_samples 1024, the same {R} rays",
                             "ms_per_image": dt_img * 1e3, "rays_per_s": R / dt_img, "total_samples": int(total_t)}
    # ---- parity at full size: every `stride`-th ray of the timed batch restated by the oracle
    if parity and train:
        try:
            out.update(_cfg5_parity(dev, est, fld, o, d, last, R, tm_kw))
        except Exception as e:  # the check must never cost the timing
            out["parity_error"] = repr(e)
    return out


def _cfg5_parity(dev, est, fld, o, d, last, R, tm_kw, stride=256):
    """cfg 5 after its timed loop: every 256th ray of the timed batch restated by the oracle -- (ray_indices, t_starts, t_ends)
    bit for bit (guard band on the visibility thresholds, oracle/check.py), colours within 1e-5; the test-mode loop on a
    2048-ray subset against oracle.test_mode_loop; and the oracle's time for that sample as cfg 5's CPU baseline."""
    from oracle import check as OC
    from oracle import oracle as O
    from nerfacc_amd.marching import render_rays_test_mode
    O.build()
    b = est.binaries.cpu().numpy()
    ab = est.aabbs.cpu().numpy()
    sel = np.arange(0, R, stride)
    os_, ds_ = np.ascontiguousarray(o[sel]), np.ascontiguousarray(d[sel])
    sig = lambda ts, te, ri: (np.float32(fld.sigma_scale) * (4.0 * (0.5 + 0.5 * np.sin(20.0 * (ts + te))))).astype(np.float32)
    occs_mean = float(b.mean())
    t0 = time.perf_counter()
    (ori, ots, ote), (fri, fts, fte, fpi) = O.occgrid_sampling(os_, ds_, b, ab, sigma_fn=sig, near_plane=0.2, render_step_size=1e-3,
                                                               cone_angle=0.004, alpha_thre=1e-2, early_stop_eps=1e-4,
                                                               occs_mean=occs_mean, return_all=True)
    s_o = sig(ots, ote, ori)
    oc, oo, od, _ = O.rendering(ots, ote, ori, sel.size, np.repeat(ots[:, None], 3, 1), sigmas=s_o)
    g_w = np.repeat(ots[:, None], 3, 1).sum(-1)
    O.render_weight_from_density_backward(ots, ote, s_o, O.pack_info(ori, sel.size), g_w)
    t_cpu = time.perf_counter() - t0
    tr, al = O.render_transmittance_from_density(fts, fte, sig(fts, fte, fri), fpi)
    ri, ts, te, colors = last["out"]
    keep = (ri % stride) == 0
    got = ((ri[keep] // stride).cpu().numpy(), ts[keep].cpu().numpy(), te[keep].cpu().numpy())
    ok, info = OC.compare_sampling(got, (ori, ots, ote), (fri, fts, fte), tr, al, early_stop_eps=1e-4, alpha_thre=min(1e-2, occs_mean))
    cg = colors.detach()[torch.from_numpy(sel).to(dev)].cpu().numpy()
    scale = max(1.0, float(np.abs(oc).max()))
    # colours of the rays none of whose samples lies on a visibility threshold's guard band (such a sample may be kept or
    # dropped, which changes its ray's colour; oracle/check.py)
    near = (np.abs(tr - np.float32(1e-4)) < 1e-6) | (np.abs(al - np.float32(min(1e-2, occs_mean))) < 1e-6)
    clean = np.ones(sel.size, bool); clean[np.unique(fri[near])] = False
    cerr = float(np.abs(cg - oc)[clean].max())
    info.update(rays=int(sel.size), rays_compared_for_colours=int(clean.sum()), colors_max_abs_err=cerr, colors_tolerance=1e-5 * scale)
    ok = ok and cerr <= 1e-5 * scale
    res = {"parity_checked": bool(ok), "parity": info,
           "cpu_baseline": {"value": sel.size / t_cpu, "unit": "rays/s", "cores": O.max_threads(), "kind": "port", "cpu_model": cpu_model(),
                            "sample": f"every {stride}th ray of the batch ({sel.size} rays, {int(fri.size)} samples before / {int(ori.size)} after "
                                      f"compaction), one pass of sampling + rendering forward + backward, {t_cpu:.1f} s"}}
    # test-mode loop, 2048-ray subset (the schedule depends on the number of rays, so the subset is its own image)
    sub = np.arange(0, R, R // 2048)[:2048]
    o2, d2 = np.ascontiguousarray(o[sub]), np.ascontiguousarray(d[sub])
    with torch.no_grad():
        rgb, opa, dep, total = render_rays_test_mode(1024, fld.rgb_sigma_fn, est, torch.from_numpy(o2).to(dev), torch.from_numpy(d2).to(dev), **tm_kw)
    field_np = lambda ts, te, ri: (np.repeat((ts * np.float32(1.0))[:, None], 3, 1).astype(np.float32), sig(ts, te, ri))
    orgb, oopa, odep, ototal, tinfo = O.test_mode_loop(1024, field_np, o2, d2, b, ab, **tm_kw)
    e_rgb = float(np.abs(rgb.cpu().numpy() - orgb).max()); e_opa = float(np.abs(opa.cpu().numpy() - oopa).max())
    guarded = int(tinfo["guard_rays"].sum())
    tm_ok = (total == ototal or guarded > 0) and e_rgb <= 2e-5 * max(1.0, float(np.abs(orgb).max())) and e_opa <= 2e-5
    res["test_mode_parity"] = {"checked": bool(tm_ok), "rays": int(sub.size), "total_samples": int(total), "oracle_total_samples": int(ototal),
                               "rays_on_the_termination_threshold": guarded, "rgb_max_abs_err": e_rgb, "opacity_max_abs_err": e_opa,
                               "iterations": int(tinfo["iterations"])}
    return res


# ----------------------------------------------------------------------------- CPU baseline (oracle, bounded sample) + parity
def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _oracle_step(O, o, d, b, aabb, step, sigma_scale):
    """The oracle's restatement of one step on a batch: sampling + rendering forward + analytic backward."""
    def sig(ts, te, ri):
        return (np.float32(sigma_scale) * (4.0 * (0.5 + 0.5 * np.sin(20.0 * (ts + te))))).astype(np.float32)

    n = o.shape[0]
    (ri, ts, te), full = O.occgrid_sampling(o, d, b, aabb, sigma_fn=sig, render_step_size=step, early_stop_eps=1e-4,
                                            alpha_thre=0.0, occs_mean=float(b.mean()), return_all=True)
    pi = O.pack_info(ri, n)
    s = sig(ts, te, ri)
    wts, tr, al = O.render_weight_from_density(ts, te, s, pi)
    rgb = np.repeat(ts[:, None], 3, 1)
    colors = O.accumulate_along_rays(wts, rgb, ri, n)
    # backward of colors.sum(): g_w = sum_c rgb, then the reverse scan (vectorised restatement)
    gw = rgb.sum(-1)
    suffix = O.packed_scan("exclusive_sum", gw * wts, pi, backward=True)
    gsig = (te - ts) * (gw * tr * (1 - al) - suffix)
    return (ri, ts, te), full, colors, gsig, sig


def cpu_baseline(w, min_seconds: float = 10.0, max_reps: int = 200):
    """The CPU restatement (oracle/) of the step on the SAME batch: sampling + rendering forward + analytic backward,
    repeated until about `min_seconds` of CPU work have been timed, on all host cores -- ``oracle.bench_step``: every stage an
    OpenMP loop over rays in C (the numpy composition ``_oracle_step`` spends most of its time in single-threaded
    elementwise passes whatever the core count; it is run ONCE, untimed, as the reference the GPU results of the very same
    batch are checked against -- parity_check -- and the two restatements are checked against each other in
    tests/test_oracle_golden.py); then the same on one thread, on a 1/4 sample of the rays."""
    from oracle import oracle as O
    O.build()
    o, d = w["rays_np"]
    b = w["estimator"].binaries.cpu().numpy()
    aabb = np.array([[-1, -1, -1, 1, 1, 1]], np.float32)
    n = o.shape[0]
    kept, full, colors, gsig, sig = _oracle_step(O, o, d, b, aabb, w["step"], w["sigma_scale"])     # pinned path: the checker
    O.bench_step(o, d, b, aabb, w["step"], w["sigma_scale"])                                        # warm-up (page faults, thread pool)
    total, reps = 0.0, 0
    while total < min_seconds and reps < max_reps:
        t0 = time.perf_counter()
        fk, M, fcol, fg = O.bench_step(o, d, b, aabb, w["step"], w["sigma_scale"])
        total += time.perf_counter() - t0
        reps += 1
    samples = int(fk[0].size)
    agree = bool(fk[0].size == kept[0].size and np.allclose(fcol, colors, atol=1e-5 * max(1.0, float(np.abs(colors).max()))))
    out = dict(value=n * reps / total, unit="rays/s", cores=O.max_threads(), kind="port", cpu_model=cpu_model(),
               sample=f"{reps} passes over the full {n}-ray batch ({M} samples before / {samples} after visibility each), "
                      f"{total:.1f} s of CPU work, C/OpenMP over rays for every stage (oracle.bench_step)",
               agrees_with_pinned_composition=agree)
    # 1 thread, every 4th ray (a 1-thread figure alongside)
    try:
        import ctypes
        omp = ctypes.CDLL("libgomp.so.1")
        n_thr = O.max_threads()
        omp.omp_set_num_threads(1)
        o1, d1 = np.ascontiguousarray(o[::4]), np.ascontiguousarray(d[::4])
        t0 = time.perf_counter()
        O.bench_step(o1, d1, b, aabb, w["step"], w["sigma_scale"])
        t1 = time.perf_counter() - t0
        omp.omp_set_num_threads(n_thr)
        out["single_thread"] = dict(value=o1.shape[0] / t1, unit="rays/s", cores=1,
                                    sample=f"1 pass over every 4th ray ({o1.shape[0]} rays), {t1:.1f} s")
        out["threads_speedup"] = out["value"] / out["single_thread"]["value"]
    except Exception as e:  # pragma: no cover
        out["single_thread"] = dict(error=repr(e))
    return out, (kept, full, colors, sig)


def parity_check(w, oracle_out):
    """The product's results on the full batch of the timed loops against the oracle's: (ray_indices, t_starts, t_ends)
    bit for bit (samples on the visibility threshold's guard band excepted, oracle/check.py), colours within 1e-5."""
    from oracle import check as OC
    from oracle import oracle as O
    kept, full, colors, sig = oracle_out
    ri, ts, te, col = w["last"]
    got = (ri.cpu().numpy(), ts.cpu().numpy(), te.cpu().numpy())
    fri, fts, fte, fpi = full
    tr = al = None
    if not (got[0].shape == kept[0].shape):
        tr, al = O.render_transmittance_from_density(fts, fte, sig(fts, fte, fri), fpi)
    ok, info = OC.compare_sampling(got, kept, (fri, fts, fte), tr, al, early_stop_eps=1e-4)
    cg = col.detach().cpu().numpy()
    scale = max(1.0, float(np.abs(colors).max()))
    cerr = float(np.abs(cg - colors).max())
    info["colors_max_abs_err"] = cerr
    info["colors_tolerance"] = 1e-5 * scale
    info["rays"] = int(cg.shape[0])
    # colours are compared when the sample sets are identical (a sample on the guard band changes its ray's colour)
    ok = ok and (cerr <= 1e-5 * scale or not info.get("identical", False))
    return bool(ok), info


# ----------------------------------------------------------------------------- main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # (defaults: 100 steps = a 150 ms timed region.  The loop is driven by the host -- two size reads per step -- and the hosts of
    #  the pool occasionally stall a process for 3-30 ms: in a 30 ms region one such stall doubled `ms_per_step` of a run whose
    #  kernels all had their usual durations, profiles/README.md)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--rays", type=int, default=1024 * 1024)
    ap.add_argument("--res", type=int, default=0,
                    help="grid resolution; default: 128 on one GPU (BASELINE cfg 2), 256 (the shared grid of cfg 4) on several")
    ap.add_argument("--grid", default="shell10", choices=["shell10", "iid10"])
    ap.add_argument("--ray-variant", default="image", choices=["image", "random"])
    ap.add_argument("--field", default="native", choices=["native", "torch"],
                    help="synthetic radiance field: three streaming HIP kernels (bench_csrc/field.hip) or torch elementwise ops")
    ap.add_argument("--bin-rays", default="auto", choices=["auto", "on", "off"],
                    help="estimator.bin_rays: auto = decided from the previous batch's coherence (the default of the library)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--pipelined", action="store_true",
                    help="also time the K steps with the next batch's traversal prefetched on a second stream (OccGridEstimator."
                         "prefetch_traversal); off by default since round 4: with that round's walk it measures like the sequential "
                         "step (1.39-1.45 ms against 1.40-1.44) -- DESIGN.md 6")
    ap.add_argument("--no-pipelined", action="store_true", help="(accepted for older scripts: the pipelined loop is off unless --pipelined)")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary configurations (cfg2 variants, cfg3, cfg5)")
    ap.add_argument("--only", default="", choices=["", "cfg2_compacting", "cfg2_random", "cfg2_testmode", "cfg4_per_rank", "cfg3", "cfg5", "cfg5_testmode"],
                    help="run ONE secondary configuration alone and print its object (for rocprofv3 passes: profiles/<round>_<cfg>_*)")
    args = ap.parse_args()

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the product path has no CPU fallback)")
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and env_world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} must be launched with one rank per GPU: python -m torch.distributed.run "
                         f"--nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 bench.py --gpus {args.gpus} "
                         f"(WORLD_SIZE is {env_world})")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    world, rank, local_rank = init_distributed("nccl", dev)
    if world > 1:  # one process per GPU: the ranks share the host's cores (the step itself is single-threaded on the host)
        torch.set_num_threads(max(1, (os.cpu_count() or world) // (2 * world)))
    if args.res <= 0:
        args.res = 128 if world == 1 else 256
    if args.only:
        assert world == 1, "--only runs on one GPU"
        fn = {"cfg2_compacting": lambda: extra_cfg2_variant(dev, args, sigma_scale=16.0),
              "cfg2_random": lambda: extra_cfg2_variant(dev, args, rays="random"),
              "cfg3": lambda: extra_cfg3(dev, 1 << 20, max(3, args.steps // 4), args.field),
              "cfg2_testmode": lambda: extra_cfg2_testmode(dev, args, n_img=max(2, args.steps // 4)),
              "cfg4_per_rank": lambda: extra_cfg4_per_rank(dev, args),
              "cfg5": lambda: extra_cfg5(dev, 1 << 21, max(2, args.steps // 6)),
              "cfg5_testmode": lambda: extra_cfg5(dev, 1 << 21, max(2, args.steps // 6), train=False)}[args.only]
        print(json.dumps({args.only: fn()}))
        return

    if world > 1 or _group_live():   # cfg 4: one grid for all ranks, broadcast once from rank 0 (also under torchrun with one rank)
        binaries = shared_grid(dev, args.res, args.grid, rank, world)
        w = make_workload(dev, args.rays, args.res, args.grid, args.ray_variant, rank, args.field, binaries=binaries)
    else:
        w = make_workload(dev, args.rays, args.res, args.grid, args.ray_variant, rank, args.field)
    w["estimator"].bin_rays = {"auto": None, "on": True, "off": False}[args.bin_rays]

    def sync():
        torch.cuda.synchronize()
        if world > 1 or _group_live():
            torch.distributed.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        run_step(w, world)
    # As timeit does: no cyclic garbage collection of the host interpreter inside the timed region.  A generation-2
    # pass (about one per 60 steps here, ~35 ms with torch loaded) would otherwise land in one of the 40 ms loops
    # at random and double its time; everything the step allocates is reference-counted and freed as usual.
    gc.collect(); gc.disable()
    sync()
    t0 = time.perf_counter()
    m_last = 0
    for _ in range(args.steps):
        m_last, _, _ = run_step(w, world)
    sync()
    dt = time.perf_counter() - t0
    gc.enable()
    dt = max_over_ranks(dt, world, dev)

    # Per-kernel HIP-event times: a separate loop of the same K steps (two event records per native call would otherwise
    # sit inside `value`).
    ksum = None
    if not args.no_kernel_timing and rank == 0:
        timer = KernelTimer(); timer.install()
        try:
            for _ in range(args.steps):
                run_step(w, world)
            ksum = timer.summary(args.steps)
        finally:
            timer.uninstall()
    elif not args.no_kernel_timing:
        for _ in range(args.steps):      # the other ranks take part in the collectives of rank 0's extra loop
            run_step(w, world)

    # Third, separately timed loop: the same K steps software-pipelined (the geometry-only traversal of batch i+1
    # runs on a side stream under the HBM-bound rendering / backward of batch i).  Reported beside `value`,
    # which stays the strictly sequential step the per-kernel numbers and the rocprof summaries refer to.
    dt_pipe, pipe_error = None, None
    if args.pipelined and not args.no_pipelined:
        try:
            handle = w["estimator"].prefetch_traversal(w["rays_o"], w["rays_d"], render_step_size=w["step"],
                                                       wait_for_inputs=False)
            for _ in range(max(2, args.warmup // 2)):
                _, _, handle = run_step(w, world, handle, prefetch=True)
            gc.collect(); gc.disable()
            sync()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                _, _, handle = run_step(w, world, handle, prefetch=True)
            sync()
            dt_pipe = time.perf_counter() - t0
            gc.enable()
        except Exception as e:  # the extra loop must never cost the headline line
            pipe_error = repr(e)
            gc.enable()
        if world > 1 or _group_live():  # every rank takes part in the reduction, also after a local failure
            dt_pipe = max_over_ranks(dt_pipe if dt_pipe is not None else float("inf"), world, dev)
            if dt_pipe == float("inf"):
                dt_pipe = None

    rc = 0
    if rank == 0:
        ms = dt / args.steps * 1e3
        value = world * args.rays / (dt / args.steps)
        cfg = "cfg2" if world == 1 else "cfg4"
        grid_note = f"{args.res}^3 {args.grid} occ grid (G=1" + (", shared: built on rank 0, broadcast bit-packed)" if world > 1 else ")")
        out = {
            "metric": "rays/sec (fwd+bwd) through 128^3 occ-grid, 1M-ray batch",
            "value": value, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{cfg}: {args.rays} {args.ray_variant} rays/GPU, {grid_note}, step 2*sqrt(3)/1024, "
                                   f"sampling+rendering fwd+bwd" + (", all-reduce of the parameter gradient over RCCL" if world > 1 else ""),
                       "rays_per_gpu": args.rays, "resolution": args.res, "grid": args.grid,
                       "samples_after_compaction": int(m_last), "parallelism": f"ray-sharded x{world}",
                       "field": f"synthetic analytic field, {args.field} callbacks (see bench.py NativeField/TorchField)"},
        }
        if pipe_error is not None:
            out["pipelined_error"] = pipe_error
        if dt_pipe is not None:
            out["pipelined"] = {
                "value": world * args.rays / (dt_pipe / args.steps), "unit": "rays/s", "ms_per_step": dt_pipe / args.steps * 1e3,
                "note": "same K steps, traversal of batch i+1 prefetched on a second stream "
                        "(OccGridEstimator.prefetch_traversal) under the rendering/backward of batch i; every step "
                        "still does one traversal and one rendering pass, results identical"}
        if ksum is not None:
            # total samples before compaction: size of the traversal output, from the estimator
            import nerfacc_amd as na
            ri, ts, te, pi = na.grid._traverse_samples(
                w["rays_o"], w["rays_d"], w["estimator"].binaries, w["estimator"].aabbs,
                torch.zeros(args.rays, device=dev), torch.full((args.rays,), 1e10, device=dev), w["step"], 0.0)
            M = int(ri.numel())
            del ri, ts, te, pi
            ab = algorithmic_bytes(args.rays, M, int(m_last), args.res)
            kernels = kernel_table(ksum, ab)
            # Op-level view: one logical op of the reference API may be several launches here.
            groups = {
                "traverse_grids (nfa_traverse_runs + cumsum + nfa_expand_runs [+ nfa_traverse_grids fill of overflow rays])":
                    ["nfa_traverse_runs", "nfa_exclusive_cumsum_pairs_stats_i64", "nfa_expand_runs",
                     "nfa_traverse_grids[mode=0]", "nfa_traverse_grids[mode=1]"],
                "rendering fwd (render_weight_from_density + 3 accumulations, one pass)": ["nfa_render_fused_fwd"],
                "rendering bwd (3 accumulations + render_weight_from_density, one pass)": ["nfa_render_fused_bwd"],
                "render_weight_from_density fwd": ["nfa_render_from_density_fwd"],
                "render_weight_from_density bwd": ["nfa_render_from_density_bwd"],
                "accumulate_along_rays x3 fwd": ["nfa_render_accumulate_fwd"],
                "accumulate_along_rays x3 bwd": ["nfa_render_accumulate_bwd"],
                "render_visibility (+count)": ["nfa_render_visibility"],
                "sample compaction": ["nfa_compact_samples"],
            }
            trav_bytes = args.rays * (24 + 8) + args.res ** 3 + M * 16 + args.rays * 16   # B_trav, SURVEY 8(d)
            ops = {}
            for name, ks in groups.items():
                ks = [k for k in ks if k in kernels]
                if not ks:
                    continue
                t_ms = sum(kernels[k]["ms_per_step"] for k in ks)
                nbytes = trav_bytes if name.startswith("traverse_grids") else sum(ab[k] for k in ks)
                ops[name] = {"ms_per_step": t_ms, "launches": [k for k in ks], "algorithmic_bytes": nbytes,
                             "achieved_GBps": nbytes / (t_ms * 1e-3) / 1e9, "frac_of_hbm_peak": nbytes / (t_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS}
            # HBM traffic from the PMC passes committed under profiles/ (collected with
            # scripts/collect_profiles.sh in separate rocprofv3 --pmc runs; 2 x FETCH_SIZE + WRITE_SIZE)
            pmc, prof, stale = {}, None, None
            try:
                from nerfacc_amd import _build as _nb
                prof = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_hbm_traffic.json"))[-1]
                pmc = json.load(open(os.path.join(ROOT, "profiles", prof)))
                if pmc.pop("_library_source_hash", None) != _nb._source_hash():
                    # counters of another build of the library say nothing about this one: not quoted
                    stale, pmc = "profiles/" + prof + " was collected with another build of the library (scripts/collect_profiles.sh)", {}
            except Exception:
                pass
            sym = {"nfa_traverse_runs": "walk_", "nfa_expand_runs": "expand_runs_kernel",
                   "nfa_render_from_density_fwd": "DensityFwdOp", "nfa_render_from_density_bwd": "DensityBwdOp",
                   "nfa_render_accumulate_fwd": "RenderAccumOp", "nfa_render_accumulate_bwd": "RenderAccumBwdOp",
                   "nfa_render_fused_fwd": "RenderFusedFwdOp", "nfa_render_fused_bwd": "RenderFusedBwdOp",
                   "nfa_render_visibility": "VisibilityOp", "nfa_compact_samples": "CompactOp"}

            def traffic_of(k):
                pat = sym.get(k)
                if pat is None:
                    return None
                hit = [v for kk, v in pmc.items() if pat in kk]
                return hit[0]["hbm_bytes_per_launch"] if hit else None

            for k, e in kernels.items():
                t = traffic_of(k)
                if t is not None:
                    e["hbm_traffic_bytes"] = t
                    e["hbm_traffic_source"] = "profiles/" + prof
            for name, o in ops.items():
                ts_ = [traffic_of(k) for k in o["launches"] if k in sym]
                o["hbm_traffic_bytes"] = sum(ts_) if ts_ and all(t is not None for t in ts_) else None
            # roofline: the dominant KERNEL of the step (longest launch); the op-level table and the kernel furthest below
            # its roofline are given beside it
            rated = {k: v for k, v in kernels.items() if "achieved_GBps" in v}
            dom = max(rated, key=lambda k: rated[k]["ms_per_launch"])
            weakest = min(rated, key=lambda k: rated[k]["frac_of_hbm_peak"])
            a = rated[dom]["achieved_GBps"]
            out["roofline"] = {"bound": rated[dom]["bound"] if rated[dom]["bound"] != "issue" else "hbm", "kernel": dom,
                               "achieved": a, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": a / HBM_PEAK_GBPS,
                               "traffic": rated[dom].get("hbm_traffic_bytes"),
                               "traffic_source": rated[dom].get("hbm_traffic_source") or stale,
                               "ms_per_launch": rated[dom]["ms_per_launch"],
                               "algorithmic_bytes_per_launch": rated[dom]["algorithmic_bytes_per_launch"],
                               "limiter": rated[dom]["bound"]}
            out["weakest_kernel"] = {"kernel": weakest, "frac_of_hbm_peak": rated[weakest]["frac_of_hbm_peak"],
                                     "ms_per_launch": rated[weakest]["ms_per_launch"], "limiter": rated[weakest]["bound"],
                                     "note": "bound by instruction issue, not by HBM (DESIGN.md 4)" if rated[weakest]["bound"] == "issue" else ""}
            out["ops"] = ops
            # SURVEY 8(d) headline: (B_trav + B_rw_f + B_rw_b) / (t_trav + t_rw_f + t_rw_b)
            hk = [k for k in ops if k.startswith(("traverse_grids", "render_weight_from_density", "rendering "))]
            hb = sum(ops[k]["algorithmic_bytes"] for k in hk); ht = sum(ops[k]["ms_per_step"] for k in hk)
            out["headline_roofline"] = {"ops": hk, "achieved": hb / (ht * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                        "frac": hb / (ht * 1e-3) / 1e9 / HBM_PEAK_GBPS, "target_frac": 0.60}
            out["kernels"] = kernels
            out["config"]["samples_before_compaction"] = M
            native_ms = sum(v["ms_per_step"] for v in ksum.values())
            out["native_ms_per_step"] = native_ms
        if not args.no_cpu_baseline and world == 1:
            run_step(w, 1)                       # the batch the oracle is about to restate (w["last"])
            base, oracle_out = cpu_baseline(w)
            out["cpu_baseline"] = base
            ok, info = parity_check(w, oracle_out)
            out["parity_checked"] = ok
            out["parity"] = info
            if not ok:
                rc = 3
            del oracle_out
        if _group_live():
            g = w["params"].grad.detach().float().cpu().tolist()
            out["distributed"] = dict(_DIST_INFO, backend=torch.distributed.get_backend(), world_size=torch.distributed.get_world_size(),
                                      grad_after_allreduce=g, grad_finite=bool(all(math.isfinite(x) for x in g)))
        if not args.no_extras and world == 1:
            w.pop("last", None)
            for key, fn in (("cfg2_compacting", lambda: extra_cfg2_variant(dev, args, sigma_scale=16.0)),
                            ("cfg2_random", lambda: extra_cfg2_variant(dev, args, rays="random")),
                            ("cfg2_testmode", lambda: extra_cfg2_testmode(dev, args)),
                            ("cfg4_per_rank", lambda: extra_cfg4_per_rank(dev, args)),
                            ("cfg3", lambda: extra_cfg3(dev, 1 << 20, 5, args.field)),
                            ("cfg5", lambda: extra_cfg5(dev, 1 << 21, 3))):
                try:
                    out[key] = fn()
                except Exception as e:  # a secondary configuration must never cost the headline line
                    out[key] = {"error": repr(e)}
                torch.cuda.empty_cache()
        print(json.dumps(out))
    if _group_live():
        torch.distributed.destroy_process_group()
    if rc:
        sys.exit(rc)


if __name__ == "__main__":
    main()
