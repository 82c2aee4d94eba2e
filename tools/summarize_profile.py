"""Turn rocprofv3 outputs under gpurun_out/ into the compact summaries committed under profiles/.

    python tools/summarize_profile.py stats  gpurun_out/profX  profiles/r01_kernel_stats.csv  [steps_in_run]
    python tools/summarize_profile.py pmc    gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmc_SQ profiles/r01_pmc_summary.json
    python tools/summarize_profile.py mix    gpurun_out/pmc_mix profiles/r02_diag_sq_counters.json
(the gpurun_out directories are written by tools/profile_round.sh on the GPU box)
"""
import collections
import csv
import glob
import json
import sys


def one(pat):
    f = glob.glob(pat) or glob.glob(pat.replace("/*/", "/"))   # rocprofv3 -o NAME writes into the directory itself
    if not f:
        raise SystemExit(f"no file matches {pat}")
    return f[0]


def stats(src, dst, steps):
    rows = list(csv.DictReader(open(one(src + "/*/*_kernel_stats.csv"))))
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "calls_per_step", "total_ms", "ms_per_step", "avg_us", "min_us", "max_us", "percent"])
        for r in rows:
            w.writerow([r["Name"], r["Calls"], f"{int(r['Calls']) / steps:.2f}", f"{float(r['TotalDurationNs']) / 1e6:.4f}",
                        f"{float(r['TotalDurationNs']) / 1e6 / steps:.4f}", f"{float(r['AverageNs']) / 1e3:.3f}",
                        f"{float(r['MinNs']) / 1e3:.3f}", f"{float(r['MaxNs']) / 1e3:.3f}", r["Percentage"]])
    print("wrote", dst, len(rows), "kernels")


def pmc(fetch_dir, write_dir, sq_dir, dst):
    def agg(d):
        out = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(one(d + "/*/*_counter_collection.csv"))):
            out[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        return out
    f, w, s = agg(fetch_dir), agg(write_dir), agg(sq_dir)
    res = {}
    for k in f:
        n = len(f[k]["FETCH_SIZE"])
        fe = sum(f[k]["FETCH_SIZE"]) / n * 1024          # rocprofv3 reports KiB
        wr = sum(w.get(k, {}).get("WRITE_SIZE", [0])) / max(1, len(w.get(k, {}).get("WRITE_SIZE", [0]))) * 1024
        mf = s.get(k, {}).get("SQ_VALU_MFMA_BUSY_CYCLES", [0])
        gui = s.get(k, {}).get("GRBM_GUI_ACTIVE", [0])
        res[k] = {"launches": n, "fetch_bytes_raw": fe, "write_bytes": wr,
                  # MI355X_MICROARCH.md: FETCH_SIZE reports half the bytes of wide (16 B/lane) coalesced reads
                  "hbm_bytes_corrected": 2 * fe + wr,
                  "mfma_busy_cycles": sum(mf) / max(1, len(mf)), "grbm_gui_active": sum(gui) / max(1, len(gui))}
    json.dump(res, open(dst, "w"), indent=1, sort_keys=True)
    print("wrote", dst, len(res), "kernels")


def mix(src, dst):
    """per kernel, per launch: the SQ instruction-mix counters of one --pmc pass (all counters of the pass)"""
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(one(src + "/*/*_counter_collection.csv"))):
        out[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {}
    for k, c in out.items():
        if "ark::" not in k and "ark" not in k:
            continue
        e = {n: sum(v) / len(v) for n, v in c.items()}
        e["launches"] = max(len(v) for v in c.values())
        if e.get("SQ_WAVES"):
            for n in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS"):
                if n in e:
                    e[n + "_per_wave"] = e[n] / e["SQ_WAVES"]
        res[k] = e
    json.dump(res, open(dst, "w"), indent=1, sort_keys=True)
    print("wrote", dst, len(res), "kernels")


if __name__ == "__main__":
    if sys.argv[1] == "mix":
        mix(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3], float(sys.argv[4]) if len(sys.argv) > 4 else 1.0)
    else:
        pmc(*sys.argv[2:6])
