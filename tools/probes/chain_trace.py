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
            cur = {"local": (s, e)}
            # the copy in is the last copyBuffer before it
            cur["in"] = last_copy
        elif "k_expect_reduce" in name and cur is not None:
            cur["reduce"] = (s, e)
        elif "k_expect_final" in name and cur is not None:
            cur["final"] = (s, e)
        elif "copyBuffer" in name:
            last_copy = (s, e)
            if cur is not None and "final" in cur and "out" not in cur:
                cur["out"] = (s, e)
                chains.append(cur)
                cur = None
    lo = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    hi = int(sys.argv[3]) if len(sys.argv) > 3 else min(len(chains), 1500)
    sel = [c for c in chains[lo:hi] if c.get("in")]
    med = lambda v: statistics.median(v) / 1e3
    print("%d chains in the trace, %d .. %d used" % (len(chains), lo, hi))
    order = ("in", "local", "reduce", "final", "out")
    for k in order:
        print("  %-7s duration %6.1f us" % (k, med([c[k][1] - c[k][0] for c in sel])))
    for a, b in zip(order[:-1], order[1:]):
        print("  gap %-7s -> %-7s %6.1f us" % (a, b, med([c[b][0] - c[a][1] for c in sel])))
    print("  chain, copy in's start -> copy out's end %6.1f us" % med([c["out"][1] - c["in"][0] for c in sel]))
    print("  between chains (copy out's end -> next copy in's start) %6.1f us" % med([b["in"][0] - a["out"][1] for a, b in zip(sel[:-1], sel[1:])]))
    print("  chain period %6.1f us" % med([b["in"][0] - a["in"][0] for a, b in zip(sel[:-1], sel[1:])]))


if __name__ == "__main__":
    main()
