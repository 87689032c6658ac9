"""Where an image-phase of the drop-in path spends its time on the device: reads a rocprofv3 --kernel-trace CSV of
`python bench.py --staged` and, for the chains of the LOCKED one-thread sweep (one image-phase in flight: copy in -> k_expect_local ->
k_expect_reduce -> k_expect_final -> copy out), prints the median duration of every kernel of the chain, the median gap between
consecutive kernels of a chain and the median gap between chains (the host's part).  Under the profiler every dispatch costs more
than without it: the numbers say where the time goes, not how much there is of it.

usage: python tools/probes/chain_trace.py <kernel_trace.csv> [first_chain last_chain]
"""
import csv
import statistics
import sys


def main():
    rows = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    chains, cur, last_copy = [], None, None
    for s, e, name in rows:
        if "k_expect_local" in name:
            if cur is not None and "final" in cur:
                chains.append(cur)
            cur = {"local": (s, e), "in": last_copy}   # the copy in is the last copyBuffer before it
        elif "k_expect_reduce" in name and cur is not None:
            cur["reduce"] = (s, e)
        elif "k_expect_final" in name and cur is not None:
            cur["final"] = (s, e)
        elif "copyBuffer" in name:
            if cur is not None and "final" in cur and "out" not in cur and s - cur["final"][1] < 3000:
                cur["out"] = (s, e)     # a copy right behind the finalise kernel: the copy back (absent in the zero-copy form)
            last_copy = (s, e)
    if cur is not None and "final" in cur:
        chains.append(cur)
    lo = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    hi = int(sys.argv[3]) if len(sys.argv) > 3 else min(len(chains), 1500)
    sel = [c for c in chains[lo:hi] if c.get("in") and "reduce" in c and "final" in c]
    has_out = sum("out" in c for c in sel) > 0.8 * len(sel)
    if has_out:
        sel = [c for c in sel if "out" in c]
    med = lambda v: statistics.median(v) / 1e3
    print("%d chains in the trace, %d .. %d used (%s)" % (len(chains), lo, hi, "copy out" if has_out else "zero-copy outputs: no copy out"))
    order = ("in", "local", "reduce", "final") + (("out",) if has_out else ())
    last = order[-1]
    for k in order:
        print("  %-7s duration %6.1f us" % (k, med([c[k][1] - c[k][0] for c in sel])))
    for a, b in zip(order[:-1], order[1:]):
        print("  gap %-7s -> %-7s %6.1f us" % (a, b, med([c[b][0] - c[a][1] for c in sel])))
    print("  chain, copy in's start -> %s's end %6.1f us" % (last, med([c[last][1] - c["in"][0] for c in sel])))
    print("  between chains (%s's end -> next copy in's start) %6.1f us" % (last, med([b["in"][0] - a[last][1] for a, b in zip(sel[:-1], sel[1:])])))
    print("  chain period %6.1f us" % med([b["in"][0] - a["in"][0] for a, b in zip(sel[:-1], sel[1:])]))


if __name__ == "__main__":
    main()
