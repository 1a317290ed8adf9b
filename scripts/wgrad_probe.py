"""GPU probe: the weight / bias gradient of value_proj (rows = T*S pixels of a clip, 256 -> 256): the library's skinny GEMM against
a split-K batched product."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench


def main():
    dev = torch.device("cuda:0")
    for dtype in (torch.float32, torch.bfloat16):
        for R in (28920, 6 * 22223, 16 * 28920):
            g = torch.randn(R, 256, device=dev, dtype=dtype)
            x = torch.randn(R, 256, device=dev, dtype=dtype)
            ref = g.t() @ x
            out = ["%-8s R=%6d  g.t()@x %.4f  sum(0) %.4f" % (str(dtype)[6:], R, bench._event_ms(lambda: g.t() @ x, 30, 5),
                                                                bench._event_ms(lambda: g.sum(0), 30, 5))]
            for rows in (128, 256, 512, 1024):
                k = max(1, R // rows)
                r = R // k
                main_rows = r * k

                def split():
                    w = torch.bmm(g[:main_rows].view(k, r, 256).transpose(1, 2), x[:main_rows].view(k, r, 256)).sum(0)
                    if main_rows < R:
                        w = w.addmm_(g[main_rows:].t(), x[main_rows:])
                    return w
                err = (split().float() - ref.float()).abs().max().item() / ref.float().abs().max().item()
                out.append("k=%d %.4f (rel %.1e)" % (k, bench._event_ms(split, 30, 5), err))
            ones = torch.ones(R, 1, device=dev, dtype=dtype)
            out.append("bias as g.t()@1 %.4f" % bench._event_ms(lambda: g.t() @ ones, 30, 5))
            k = R // 256
            r = R // k
            o3 = torch.ones(k, 1, r, device=dev, dtype=dtype)
            out.append("bias as bmm %.4f" % bench._event_ms(lambda: torch.bmm(o3, g[:r * k].view(k, r, 256)).sum(0), 30, 5))
            out.append("bias view-sum %.4f" % bench._event_ms(lambda: g[:r * k].view(k, r, 256).sum(1).sum(0), 30, 5))
            print(" | ".join(out), flush=True)


if __name__ == "__main__":
    main()
