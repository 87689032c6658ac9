"""parse the counter_collection.csv files of tools/pmc_traffic.sh into profiles-ready JSON.

FETCH_SIZE / WRITE_SIZE are reported in KiB.  Their bytes-per-unit are NOT assumed: each is calibrated on the 1 GiB
dispatches of tools/pmc_calib profiled in the same pass -- the streaming copy for WRITE_SIZE (and as a reference point
for FETCH_SIZE), the scattered 64-byte-cell gather for the E-step kernel's FETCH_SIZE (same request shape)."""
import collections
import csv
import glob
import json
import os
import re
import sys

out = sys.argv[1]
GiB = 1024.0 ** 3


def load(dirname):
    fs = glob.glob("%s/%s/**/*counter_collection.csv" % (out, dirname), recursive=True)
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    if not fs:
        return per
    for r in csv.DictReader(open(fs[0])):
        per[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return per


def pick(per, key, counter, how=max):
    vals = [v for k, d in per.items() if key in k for v in d.get(counter, [])]
    return how(vals) if vals else None


summary = {"calibration": {}, "note": "raw = counter value per dispatch (KiB for FETCH_SIZE / WRITE_SIZE); bytes_per_unit = "
           "known bytes of the calibration dispatch / its raw value, measured in the same rocprofv3 pass"}
cal_f, cal_w, cal_l2 = load("cal_fetch"), load("cal_write"), load("cal_l2")
for kern, nbytes in (("k_cal_stream", GiB), ("k_cal_gather64", GiB), ("k_cal_gather16", GiB)):
    raw = pick(cal_f, kern, "FETCH_SIZE")
    ent = {"known_read_bytes": nbytes, "FETCH_SIZE_raw": raw, "fetch_bytes_per_unit": (nbytes / raw) if raw else None}
    if kern == "k_cal_stream":
        raww = pick(cal_w, kern, "WRITE_SIZE")
        ent.update({"known_written_bytes": nbytes, "WRITE_SIZE_raw": raww, "write_bytes_per_unit": (nbytes / raww) if raww else None})
    for c in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_HIT_sum", "TCC_MISS_sum"):
        v = pick(cal_l2, kern, c)
        if v is not None:
            ent[c] = v
    summary["calibration"][kern] = ent
fetch, write, l2 = load("fetch"), load("write"), load("l2")
u_gather = summary["calibration"]["k_cal_gather64"]["fetch_bytes_per_unit"]
u_stream = summary["calibration"]["k_cal_stream"]["fetch_bytes_per_unit"]
u_write = summary["calibration"]["k_cal_stream"].get("write_bytes_per_unit")
n_per_launch = int(os.environ.get("THX_PROBE_PARTICLES", "2048")) // 2
for name, unit_f, shape in (("k_expect_local", u_gather, "scattered 64-byte cells"), ("k_insert_win", u_stream, "streamed rows")):
    ent = {"fetch_calibration": shape}
    fr = pick(fetch, name, "FETCH_SIZE", how=lambda v: sum(v) / len(v))
    wr = pick(write, name, "WRITE_SIZE", how=lambda v: sum(v) / len(v))
    if fr is not None and unit_f:
        ent["fetch_raw_per_launch"] = fr
        ent["fetch_bytes_per_launch"] = fr * unit_f
        ent["fetch_bytes_per_launch_if_stream_unit"] = fr * u_stream if u_stream else None
    if wr is not None and u_write:
        ent["write_raw_per_launch"] = wr
        ent["write_bytes_per_launch"] = wr * u_write
    for c in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_HIT_sum", "TCC_MISS_sum"):
        v = pick(l2, name, c, how=lambda v: sum(v) / len(v))
        if v is not None:
            ent[c + "_per_launch"] = v
    if "fetch_bytes_per_launch" in ent:
        ent["hbm_bytes_per_launch"] = ent["fetch_bytes_per_launch"] + ent.get("write_bytes_per_launch", 0.0)
        ent["images_per_launch"] = n_per_launch
        summary[name] = ent
# the brick-sorted insertion runs k_bin and k_acc once per CHUNK of images: totals over every launch of the run, per image
# (images of the run = images per E-step launch x its launches / 3 phases)
n_exp = pick(fetch, "k_expect_local", "FETCH_SIZE", how=len)
n_img_total = n_per_launch * (n_exp or 0) / 3.0
for name in ("k_bin", "k_acc"):
    fr = pick(fetch, name, "FETCH_SIZE", how=sum)
    wr = pick(write, name, "WRITE_SIZE", how=sum)
    if fr is not None and wr is not None and u_stream and u_write and n_img_total:
        summary[name] = {"fetch_calibration": "streamed rows", "launches": pick(fetch, name, "FETCH_SIZE", how=len),
                         "fetch_bytes_per_image": fr * u_stream / n_img_total, "write_bytes_per_image": wr * u_write / n_img_total,
                         "images_total": n_img_total}
# measured LDS integer-add rates of the chip (tools/lds_atomic_bench): the insertion kernels' own bound
lds_rate, lds_rate64 = None, None
try:
    for line in open(out + "/lds_atomic_bench.txt"):
        m = re.match(r"ds_add_u32 random\s+[\d.]+ ms\s+([\d.]+) G lane-ops/s", line)
        if m:
            lds_rate = float(m.group(1)) * 1e9
        m = re.match(r"ds_add_u64 random\s+[\d.]+ ms\s+([\d.]+) G lane-ops/s", line)
        if m:
            lds_rate64 = float(m.group(1)) * 1e9
except OSError:
    pass
summary["lds_add_u32_random_per_s"] = lds_rate
summary["lds_add_u64_random_per_s"] = lds_rate64
print(json.dumps(summary, indent=1))
json.dump(summary, open(out + "/summary.json", "w"), indent=1)
pm = {"box": int(os.environ.get("THX_PROBE_BOX", "256")), "images_per_launch": n_per_launch, "lds_add_u32_per_s": lds_rate, "lds_add_u64_per_s": lds_rate64,
      "source": "tools/pmc_traffic.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over "
                + (("`%s` (average over every launch of the run)" % os.environ["PMC_CMD"]) if os.environ.get("PMC_CMD") else
                   "tools/traffic_probe.py (particle-filter support points, one phase)") + "; KiB units; FETCH_SIZE of the E-step kernel "
                "scaled by the bytes-per-unit measured on tools/pmc_calib's 1 GiB scattered 64-byte-cell gather in the same "
                "pass (x%.3f; the wide streaming copy gives x%.3f), WRITE_SIZE by the 1 GiB streaming copy" % (
                    (u_gather or 0) / 1024.0, (u_stream or 0) / 1024.0)}
if "k_expect_local" in summary:
    pm["hbm_bytes_per_image_phase"] = summary["k_expect_local"]["hbm_bytes_per_launch"] / n_per_launch
if "k_insert_win" in summary:
    pm["insert_hbm_bytes_per_image"] = summary["k_insert_win"]["hbm_bytes_per_launch"] / n_per_launch
if "k_bin" in summary and "k_acc" in summary:
    pm["insert_hbm_bytes_per_image"] = sum(summary[k]["fetch_bytes_per_image"] + summary[k]["write_bytes_per_image"] for k in ("k_bin", "k_acc"))
json.dump(pm, open(out + "/pmc_traffic.json", "w"), indent=1)
