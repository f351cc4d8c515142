#!/usr/bin/env python3
"""Which native calls survive hipGraph capture?  Each piece runs in its own process (a failing capture can take the
process down)."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PIECES = ["sampling_nograd", "sampling_grad", "sampling_trans", "sampling_loss", "sampling_loss_grad", "torch_only", "importance_sampling", "importance_sampling_t", "transmittance_fwd", "transmittance_fwd_bwd", "pdf_loss", "pdf_loss_bwd"]

def child(name):
    sys.path.insert(0, ROOT)
    import torch
    import nerfacc_amd as na
    from nerfacc_amd.estimators.prop_net import _pdf_loss
    dev = torch.device("cuda:0")
    R = int(os.environ.get("GB_R", "4096"))
    v = torch.sort(torch.rand((R, 65), device=dev))[0]
    c = torch.sort(torch.rand((R, 65), device=dev))[0]
    sig = torch.rand((R, 64), device=dev, requires_grad=True)
    qv = torch.sort(torch.rand((R, 17), device=dev))[0]; qc = torch.sort(torch.rand((R, 17), device=dev))[0]
    kc = c.clone().requires_grad_(True)

    pp = torch.nn.Parameter(torch.tensor([3.0, 4.0], device=dev))
    est = na.PropNetEstimator().to(dev)
    prop = lambda ts, te: torch.exp(-((ts + te) * 0.5 - pp[1]) ** 2) * pp[0]
    fine = lambda ts, te: torch.exp(-((ts + te) * 0.5 - 4.0) ** 2 * 2.0) * 5.0

    def fn():
        if name.startswith("sampling"):
            rg = name != "sampling_nograd"
            est.prop_cache.clear()
            ts, te = est.sampling([prop, prop], [64, 64], 16, R, 2.0, 6.0, sampling_type="uniform", stratified=False, requires_grad=rg)
            if name in ("sampling_nograd", "sampling_grad"):
                return ts
            trans, _ = na.render_transmittance_from_density(ts, te, fine(ts, te))
            if name == "sampling_trans":
                return trans
            loss = est.compute_loss(trans)
            if name == "sampling_loss":
                return loss
            return torch.autograd.grad(loss, [pp])[0]
        if name == "torch_only":
            return (v * 2 + c).sum()
        if name == "importance_sampling":
            iv, sm = na.importance_sampling(na.RayIntervals(vals=v), c, 16)
            return iv.vals
        if name == "importance_sampling_t":
            out = na.importance_sampling(na.RayIntervals(vals=v), c, 16, transform=("uniform", 2.0, 6.0), need_samples=False)
            return out[2]
        if name == "transmittance_fwd":
            with torch.no_grad():
                return na.render_transmittance_from_density(v[:, :-1], v[:, 1:], sig)[0]
        if name == "transmittance_fwd_bwd":
            t = na.render_transmittance_from_density(v[:, :-1], v[:, 1:], sig)[0]
            return torch.autograd.grad(t.sum(), [sig])[0]
        if name == "pdf_loss":
            with torch.no_grad():
                return _pdf_loss(na.RayIntervals(vals=qv), qc, na.RayIntervals(vals=v), c)
        if name == "pdf_loss_bwd":
            l = _pdf_loss(na.RayIntervals(vals=qv), qc, na.RayIntervals(vals=v), kc)
            return torch.autograd.grad(l.sum(), [kc])[0]
    if os.environ.get("GB_EAGER_FIRST"):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            ref = fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    g.replay(); torch.cuda.synchronize()
    print("OK", name, bool(torch.equal(out, ref)), flush=True)

if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
    else:
        for p in PIECES:
            r = subprocess.run([sys.executable, __file__, p], capture_output=True, text=True, timeout=120)
            tail = [l for l in (r.stdout + r.stderr).splitlines() if l.startswith("OK") or "Error" in l or "error" in l][-2:]
            print(p, "rc=%d" % r.returncode, tail, flush=True)
