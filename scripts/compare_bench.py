#!/usr/bin/env python3
"""Compare two bench.py JSON lines time by time -- the headline step, every kernel and EVERY `other_configs` entry -- and fail
loudly when one got slower.  Round 3 lost 45 % on BASELINE configs[1] to a change tuned on another shape because only the
headline tail was looked at; scripts/profile_round.sh now ends with this comparison against the previous round's file.

    python scripts/compare_bench.py profiles/r03_e_bench.json gpurun_out/r04_a/bench.json [--tol 0.05]

Exit status 1 when any time grew by more than --tol (boxes differ by a few per cent: confirm on one box with
scripts/ab_all.sh before believing a small difference).
"""
import json
import sys


def load(path):
    with open(path) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def times(node, prefix=""):
    """{dotted key: milliseconds} of every time-like leaf."""
    out = {}
    if isinstance(node, dict):
        for k, v in node.items():
            key = "%s.%s" % (prefix, k) if prefix else k
            if isinstance(v, (dict, list)):
                out.update(times(v, key))
            elif isinstance(v, (int, float)) and (k.endswith("_ms") or k in ("ms_per_step", "avg_ms")):
                out[key] = float(v)
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    tol = 0.05
    for i, a in enumerate(sys.argv):
        if a == "--tol":
            tol = float(sys.argv[i + 1])
            args = [x for x in args if x != sys.argv[i + 1]]
    old, new = times(load(args[0])), times(load(args[1]))
    worse = []
    # host-side times: they follow the box's CPU, not the kernels (the small plain calls of configs[4] run 0.03-0.07 ms of kernels
    # under 0.09-0.25 ms of Python + autograd: the same build measured 0.085 and 0.18 ms on one box within a minute)
    host_bound = ("eager", "single_clip_latency", "single_clip_60_queries.fwd", "cpu_baseline", "cfg4_swinl_fp16_decoder_like", "cfg4_mask_head_like")
    print("%-110s %10s %10s %8s" % ("time (ms)", "before", "after", "ratio"))
    for k in sorted(set(old) & set(new)):
        r = new[k] / old[k] if old[k] > 0 else float("nan")
        soft = any(h in k for h in host_bound)
        flag = ("  (slower; host-bound, not counted)" if soft else "  <-- SLOWER") if r > 1 + tol else ("  faster" if r < 1 - tol else "")
        print("%-110s %10.4f %10.4f %8.3f%s" % (k[-110:], old[k], new[k], r, flag))
        if r > 1 + tol and not soft:
            worse.append((k, r))
    for k in sorted(set(new) - set(old)):
        print("%-110s %10s %10.4f" % (k[-110:], "-", new[k]))
    if worse:
        print("\n%d time(s) grew by more than %.0f %%:" % (len(worse), 100 * tol))
        for k, r in worse:
            print("   %s  x%.3f" % (k, r))
        sys.exit(1)
    print("\nno time grew by more than %.0f %%" % (100 * tol))


if __name__ == "__main__":
    main()
