"""what a HIP-event bracket around ONE library GEMM call measures, against the time per call of 50 calls back to back"""
import torch
dev = torch.device('cuda:0')
def ev(): return torch.cuda.Event(enable_timing=True)
for (m, n, k, dt) in [(65536, 576, 192, torch.bfloat16), (4096, 2304, 768, torch.bfloat16), (400, 256, 256, torch.float32), (8, 8, 8, torch.bfloat16)]:
    a = torch.randn(m, k, device=dev, dtype=dt); b = torch.randn(n, k, device=dev, dtype=dt); bias = torch.randn(n, device=dev, dtype=dt)
    for name, fn in (('mm', lambda: a @ b.t()), ('linear', lambda: torch.nn.functional.linear(a, b, bias))):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        s, e = ev(), ev()
        s.record()
        for _ in range(50): fn()
        e.record(); torch.cuda.synchronize()
        per = s.elapsed_time(e) / 50 * 1e3
        res = {}
        for spin in (False, True):
            pairs = []
            if spin: torch.cuda._sleep(200_000_000)
            for _ in range(50):
                x, y = ev(), ev(); x.record(); fn(); y.record(); pairs.append((x, y))
            torch.cuda.synchronize()
            ts = sorted(x.elapsed_time(y) * 1e3 for x, y in pairs)
            res[spin] = ts[len(ts) // 2]
        print(f'{name:7s} {m}x{n}x{k} {str(dt)[6:]:9s} back-to-back {per:7.1f} us  bracket {res[False]:7.1f}  bracket behind a spin {res[True]:7.1f}')
x, y = ev(), ev(); torch.cuda._sleep(100_000_000); x.record(); y.record(); torch.cuda.synchronize(); print('empty bracket behind spin', x.elapsed_time(y) * 1e3)
