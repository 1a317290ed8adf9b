"""GPU: the three fused calls ONE DeVIS clip makes per transformer layer -- decoder (300 queries per frame), encoder at 360x640 and
encoder at 800x1333 (every pixel a query) -- forward + backward, N launches each, for `rocprofv3 --kernel-trace --stats`:

    rocprofv3 --kernel-trace --stats --output-format csv -d out -- python3 scripts/devis_calls.py [which] [n]     which: dec | encA | encB
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import scatter_ab


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "dec"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    case = {"dec": lambda: scatter_ab.temporal_case(1, "A", "uniform", 300, torch.float32, n),
            "encA": lambda: scatter_ab.temporal_case(1, "A", "local", 4820, torch.float32, n),
            "encB": lambda: scatter_ab.temporal_case(1, "B", "local", 22223, torch.float32, n)}[which]
    fwd, bwd, gv, _ = case()
    for _ in range(n):
        fwd()
        bwd()
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
