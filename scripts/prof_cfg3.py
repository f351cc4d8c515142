import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, nerfacc_amd as na
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0"); R = 1 << 20
p = torch.nn.Parameter(torch.tensor([3.0, 4.0], device=dev))
est = na.PropNetEstimator().to(dev)
import bench
fld = bench.NativePropField(p)
prop, fine = fld.prop, fld.fine
def step():
    ts, te = est.sampling([prop, prop], [64, 64], 16, R, 2.0, 6.0, sampling_type="uniform", stratified=False, requires_grad=True)
    trans, _ = na.render_transmittance_from_density(ts, te, fine(ts, te))
    loss = est.compute_loss(trans)
    return torch.autograd.grad(loss, [p])[0]
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU], with_stack=False) as prof:
    for _ in range(3): step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=70))
