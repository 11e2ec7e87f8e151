"""GPU box: the chip's sustained HBM rates for the three simplest access mixes (torch kernels, 4 GiB tensors, best of 5):
write-only (fill), read + write (copy), read-only (sum) -- the yardsticks for kernels that are bound by one direction."""
import json, torch
n = 1 << 30
x = torch.empty(n, device="cuda", dtype=torch.float32); y = torch.empty_like(x)
x.fill_(1.0); y.fill_(2.0)
def best(fn, nbytes):
    t = 1e9
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        t = min(t, a.elapsed_time(b) * 1e-3)
    return round(nbytes / t / 1e12, 3)
print(json.dumps({"write_only_fill_TBps": best(lambda: x.fill_(3.0), 4 * n),
                  "copy_read_plus_write_TBps": best(lambda: y.copy_(x), 8 * n),
                  "read_only_sum_TBps": best(lambda: x.sum(), 4 * n)}))
