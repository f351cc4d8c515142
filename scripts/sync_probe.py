"""Host wake-up latency of the ways to read a device value: event.synchronize, stream.synchronize, .item(), polling pinned memory."""
import time

import torch

dev = torch.device("cuda:0")
x = torch.zeros(64 << 20, device=dev)            # 256 MB: a ~100 us kernel
val = torch.zeros(8, dtype=torch.int64, device=dev)
host = torch.zeros(8, dtype=torch.int64).pin_memory()
hn = host.numpy()


def gpu_work():
    x.add_(1.0)
    val.add_(1)


def timed(fn, reps=50):
    lat = []
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        gpu_work()
        e1.record()
        fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        lat.append((t1 - t0) * 1e6 - e0.elapsed_time(e1) * 1e3)
    lat.sort()
    return lat[len(lat) // 2], lat[0], lat[-1]


def ev_sync():
    host.copy_(val, non_blocking=True)
    e = torch.cuda.Event(); e.record(); e.synchronize()
    return int(host[0])


def ev_sync_blocking():
    host.copy_(val, non_blocking=True)
    e = torch.cuda.Event(blocking=True); e.record(); e.synchronize()
    return int(host[0])


def stream_sync():
    host.copy_(val, non_blocking=True)
    torch.cuda.current_stream().synchronize()
    return int(host[0])


def item():
    return int(val[0].item())


def poll():
    hn[7] = -1
    val[7] = 5   # (a fill kernel; keeps the flag slot distinct)
    host.copy_(val, non_blocking=True)
    while hn[7] == -1:
        pass
    return int(hn[0])


def ev_query_spin():
    host.copy_(val, non_blocking=True)
    e = torch.cuda.Event(); e.record()
    while not e.query():
        pass
    return int(host[0])


for name, fn in (("event.synchronize", ev_sync), ("event(blocking).synchronize", ev_sync_blocking), ("stream.synchronize", stream_sync),
                 (".item()", item), ("poll pinned memory", poll), ("spin on event.query", ev_query_spin)):
    med, lo, hi = timed(fn)
    print(f"{name:30s} host latency beyond the GPU work: median {med:7.1f} us  min {lo:7.1f}  max {hi:7.1f}")
