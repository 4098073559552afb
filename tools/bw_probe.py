"""HBM bandwidth probes with large (> Infinity Cache) buffers: fill (write only), sum (read only), copy."""
import torch
def t_ms(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for mb in (367, 1024, 4096):
    n = mb * 1024 * 1024 // 4
    a = torch.empty(n, device="cuda"); b = torch.empty(n, device="cuda")
    w = t_ms(lambda: a.fill_(1.0)); r = t_ms(lambda: a.sum()); c = t_ms(lambda: b.copy_(a))
    print(f"{mb:5d} MB: fill {mb/1024/w*1e3/1e3*1.0737:6.2f} TB/s ({w*1e3:7.1f} us)  sum(read) {mb/1024/r*1e3/1e3*1.0737:6.2f} TB/s  copy {2*mb/1024/c*1e3/1e3*1.0737:6.2f} TB/s")
