"""parse the two counter_collection.csv files of tools/pmc_traffic.sh into profiles-ready JSON"""
import csv, glob, json, sys, collections
out = sys.argv[1]
res = {}
for which in ("fetch", "write"):
    fs = glob.glob("%s/%s/**/*counter_collection.csv" % (out, which), recursive=True)
    rows = list(csv.DictReader(open(fs[0])))
    per = collections.defaultdict(list)
    for r in rows:
        per[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    res[which] = {k: v for k, v in per.items()}
GiB = 1024.0 ** 3
def find(d, key):
    return [(k, v) for k, v in d.items() if key in k]
summary = {}
# calibration: the 1 GiB device-to-device copy issued by tools/traffic_probe.py runs as __amd_rocclr_copyBuffer
# (float4 streaming: 1 GiB read + 1 GiB written); it is the largest dispatch of that name.
cal = {}
for which in ("fetch", "write"):
    best = None
    for k, v in res[which].items():
        if "copyBuffer" in k and max(v) > 0:
            if best is None or max(v) > best[1]:
                best = (k, max(v))
    cal[which] = best
summary["calibration"] = {w: {"kernel": cal[w][0][:80], "raw_for_1GiB": cal[w][1], "bytes_per_unit": GiB / cal[w][1]} for w in cal if cal[w]}
summary["note"] = ("FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies a wide coalesced read stream at half its "
                   "bytes (MI355X_MICROARCH.md, HBM section): the 1 GiB copy reads 524288 KiB by the counter. bytes_per_unit "
                   "below therefore applies the same x2 to the kernels' fetch counts (an upper bound for their mixed-width "
                   "gathers); WRITE_SIZE is exact (1 GiB -> 1048576 KiB).")
for name in ("k_insert_win", "k_insert_tiles", "k_expect_local<"):
    ent = {}
    for which in ("fetch", "write"):
        f = find(res[which], name)
        if f:
            vals = f[0][1]
            ent[which + "_raw_per_launch"] = sum(vals) / len(vals)
            ent[which + "_bytes_per_launch"] = ent[which + "_raw_per_launch"] * summary["calibration"][which]["bytes_per_unit"]
            ent["launches"] = len(vals)
    if ent:
        ent["hbm_bytes_per_launch"] = ent.get("fetch_bytes_per_launch", 0) + ent.get("write_bytes_per_launch", 0)
        summary[name] = ent
print(json.dumps(summary, indent=1))
json.dump(summary, open(out + "/summary.json", "w"), indent=1)
# per-image figures bench.py scales into roofline.traffic (tools/traffic_probe.py launches 1024 images per kernel)
import os
n_per_launch = int(os.environ.get("THX_PROBE_PARTICLES", "2048")) // 2
pm = {"box": 256, "images_per_launch": n_per_launch,
      "source": "tools/pmc_traffic.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over "
                "tools/traffic_probe.py (particle-filter support points, one phase); KiB units; FETCH_SIZE x2 (gfx950 "
                "wide-read correction, calibrated on a 1 GiB device copy in the same passes); WRITE_SIZE exact"}
if "k_expect_local<" in summary:
    pm["hbm_bytes_per_image_phase"] = summary["k_expect_local<"]["hbm_bytes_per_launch"] / n_per_launch
ins = summary.get("k_insert_win") or summary.get("k_insert_tiles")
if ins:
    pm["insert_hbm_bytes_per_image"] = ins["hbm_bytes_per_launch"] / n_per_launch
json.dump(pm, open(out + "/pmc_traffic.json", "w"), indent=1)
