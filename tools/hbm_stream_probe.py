"""What plain streaming kernels reach on this GPU (PyTorch copy / sum / fill / scale on 2 GiB tensors): the yardstick for the
operand-writer (modulate) kernels, which move 23 GB per BAIR step at 5.1 TB/s.  python tools/hbm_stream_probe.py"""
import torch, time
x = torch.empty(512 * 1024 * 1024, dtype=torch.float32, device="cuda").normal_()
y = torch.empty_like(x)
def t(f, n=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
ms = t(lambda: y.copy_(x)); print(f"copy 2 GiB -> 2 GiB: {ms:.3f} ms = {2 * x.numel() * 4 / ms / 1e9:.2f} TB/s (read + write)")
ms = t(lambda: x.sum()); print(f"read 2 GiB (sum): {ms:.3f} ms = {x.numel() * 4 / ms / 1e9:.2f} TB/s")
ms = t(lambda: y.fill_(1.0)); print(f"write 2 GiB (fill): {ms:.3f} ms = {x.numel() * 4 / ms / 1e9:.2f} TB/s")
z = torch.empty(3 * x.numel() // 2, dtype=torch.float32, device="cuda")
ms = t(lambda: torch.mul(x[: x.numel() // 2], 2.0, out=z[: x.numel() // 2])); print(f"scale 1 GiB -> 1 GiB: {ms:.3f} ms = {x.numel() * 4 / ms / 1e9:.2f} TB/s")
