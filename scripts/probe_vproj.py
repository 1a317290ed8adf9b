"""GPU probe: value_proj as one GEMM (dense [rows, M*D]) vs a per-head batched GEMM writing head-major [M, rows, D]."""
import torch, sys
dev = torch.device("cuda:0")
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 16 * 6 * 4820
C, M, D = 256, 8, 32
x = torch.randn(rows, C, device=dev); W = torch.randn(M * D, C, device=dev) * 0.05; b = torch.randn(M * D, device=dev)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n
Wh = W.view(M, D, C).transpose(1, 2).contiguous()          # [M, C, D]
bh = b.view(M, 1, D)
out_hm = torch.empty(M, rows, D, device=dev)
dense = lambda: torch.nn.functional.linear(x, W, b)
batched = lambda: torch.baddbmm(bh, x.unsqueeze(0).expand(M, rows, C), Wh, out=out_hm)
mm = lambda: torch.matmul(x, Wh)
print("dense linear", round(timeit(dense), 4), "batched baddbmm(out=)", round(timeit(batched), 4), "matmul bcast", round(timeit(mm), 4))
ref = dense().view(rows, M, D).permute(1, 0, 2)
print("max diff", (batched() - ref).abs().max().item())
