#!/usr/bin/env python
"""Calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE on path_stream_kernel's access pattern (tools/probes/fetch_calib.hip): known bytes moved /
bytes the counter reports, per access width.  Run on the GPU box:  python tools/fetch_calib.py [out.json]
bench.py reads the committed record (profiles/fetch_calib.json) and applies the factors to its measured traffic."""
import csv
import glob
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tools", "probes", "fetch_calib.hip")


def counters(exe, pmc):
    with tempfile.TemporaryDirectory(dir="/tmp") as d:
        cmd = ["rocprofv3", "--kernel-trace", "--pmc", *pmc, "-f", "csv", "-d", d, "--", exe]
        subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True, timeout=600)
        acc, cnt = {}, {}
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                key = (row["Kernel_Name"].split("(")[0].split("<")[0], row["Counter_Name"])
                acc[key] = acc.get(key, 0.0) + float(row["Counter_Value"])
                cnt[key] = cnt.get(key, 0) + 1
        return {k: acc[k] / cnt[k] for k in acc}


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "fetch_calib.json")
    exe = os.path.join(tempfile.gettempdir(), "fetch_calib")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-o", exe, SRC], check=True)
    known = json.loads(subprocess.run([exe], check=True, capture_output=True, text=True).stdout)
    fetch = counters(exe, ["FETCH_SIZE"])
    write = counters(exe, ["WRITE_SIZE"])
    rec = {"what": "known bytes / bytes reported by rocprofv3 (FETCH_SIZE, WRITE_SIZE in KiB x 1024), per launch, 4 GiB footprint, tools/probes/fetch_calib.hip", "kernels": {}}
    for name, kb in known.items():
        f = next((v for (k, c), v in fetch.items() if k.endswith(name) and c == "FETCH_SIZE"), None)
        w = next((v for (k, c), v in write.items() if k.endswith(name) and c == "WRITE_SIZE"), None)
        e = dict(kb, fetch_size_kib=f, write_size_kib=w)
        if f and kb["read"]:
            e["read_factor"] = kb["read"] / (f * 1024.0)
        if w and kb["written"]:
            e["write_factor"] = kb["written"] / (w * 1024.0)
        rec["kernels"][name] = e
    k = rec["kernels"]
    rec["factors"] = {"read_8B_per_lane": k["read_b64"].get("read_factor"), "read_4B_per_lane": k["read_b32"].get("read_factor"),
                      "read_16B_per_lane": k["read_b128"].get("read_factor"), "write_8B_per_lane": k["write_b64"].get("write_factor"),
                      "write_4B_per_lane": k["write_b32"].get("write_factor"), "sweep_like_read": k["sweep_like"].get("read_factor"),
                      "sweep_like_write": k["sweep_like"].get("write_factor")}
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps(rec["factors"]))


if __name__ == "__main__":
    main()
