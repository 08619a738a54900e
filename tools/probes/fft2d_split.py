"""Probe: 2-D C2C (rocFFT's own plan) vs the two 1-D passes issued separately (rows, then strided columns)."""
import torch, time
dev = torch.device('cuda:0')
def t(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e3
for (h, w) in [(1080, 1920), (764, 1358), (540, 960), (382, 679), (270, 480)]:
    x = torch.randn(24, h, w, dtype=torch.complex64, device=dev)
    a = t(lambda: torch.fft.ifft2(x))
    r = t(lambda: torch.fft.ifft(x, dim=-1))
    c = t(lambda: torch.fft.ifft(x, dim=-2))
    byt = 24 * h * w * 8 * 2
    print(f"{h}x{w}: 2-D {a:.3f} ms | rows {r:.3f} ms ({byt / r / 1e9:.0f} GB/s) | cols {c:.3f} ms ({byt / c / 1e9:.0f} GB/s) | rows+cols {r + c:.3f}")
