"""Condense rocprofv3 CSV output (gpurun_out/...) into the small summaries kept under profiles/.

  python tools/summarize_rocprof.py stats  <kernel_stats.csv>            > profiles/rNN_<what>_kernel_stats.csv
  python tools/summarize_rocprof.py pmc    <fetch_counter_collection.csv> <write_counter_collection.csv> > profiles/rNN_pmc.json
  python tools/summarize_rocprof.py db-stats <results.db>   /   db-pmc <fetch_results.db> <write_results.db>   (rocpd SQLite output)

PMC units and the gfx950 correction follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and
WRITE_SIZE are in KiB; FETCH_SIZE reports one half of the bytes of a coalesced streaming read on gfx950, so it is
doubled; WRITE_SIZE is taken as is.  Collected in separate --pmc passes (TCC slots).
"""
import collections
import csv
import json
import sys


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


def stats(path):
    rows = list(csv.DictReader(open(path)))
    w = csv.writer(sys.stdout)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"])
    for r in rows[:25]:
        w.writerow([short(r["Name"])[:90], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]])


def pmc(fetch_path, write_path):
    out = collections.defaultdict(dict)
    for path, counter in ((fetch_path, "FETCH_SIZE"), (write_path, "WRITE_SIZE")):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter:
                agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            out[k][counter + "_KiB_mean_per_launch"] = sum(v) / len(v)
            out[k][counter + "_launches"] = len(v)
    res = {}
    for k, d in out.items():
        f = d.get("FETCH_SIZE_KiB_mean_per_launch", 0.0)
        w = d.get("WRITE_SIZE_KiB_mean_per_launch", 0.0)
        d["hbm_bytes_per_launch_corrected"] = (2.0 * f + w) * 1024.0
        res[k] = d
    keep = {k: v for k, v in res.items() if v["hbm_bytes_per_launch_corrected"] > 1e8}
    print(json.dumps({"correction": "FETCH_SIZE x2 (gfx950, coalesced streaming reads), KiB -> bytes x1024", "kernels": keep}, indent=1))


def db_stats(path):
    """rocprofv3's default output is a rocpd SQLite file; `top_kernels` is its --stats summary.  The view reports MICROSECONDS
    (k_strip_spmv at C3: 3536 = 3.5 ms, matching the HIP-event timing in bench.py)."""
    import sqlite3

    w = csv.writer(sys.stdout)
    w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
    cur = sqlite3.connect(path).cursor()
    for name, calls, total, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels limit 25"):
        w.writerow([short(name)[:90], calls, int(total), int(avg), round(pct, 3)])


def db_pmc(fetch_path, write_path):
    import sqlite3

    out = collections.defaultdict(dict)
    for path, counter in ((fetch_path, "FETCH_SIZE"), (write_path, "WRITE_SIZE")):
        cur = sqlite3.connect(path).cursor()
        agg = collections.defaultdict(list)
        for name, value in cur.execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)):
            agg[short(name)].append(float(value))
        for k, v in agg.items():
            out[k][counter + "_KiB_mean_per_launch"] = sum(v) / len(v)
            out[k][counter + "_launches"] = len(v)
    res = {}
    for k, d in out.items():
        f = d.get("FETCH_SIZE_KiB_mean_per_launch", 0.0)
        w = d.get("WRITE_SIZE_KiB_mean_per_launch", 0.0)
        d["hbm_bytes_per_launch_corrected"] = (2.0 * f + w) * 1024.0
        res[k] = d
    keep = {k: v for k, v in res.items() if v["hbm_bytes_per_launch_corrected"] > 1e8}
    print(json.dumps({"correction": "FETCH_SIZE x2 (gfx950, coalesced streaming reads), KiB -> bytes x1024", "kernels": keep}, indent=1))


SQ_KERNELS = ("strip_spmv", "tall_spmv", "strip_fill", "strip_count", "random_rows", "value_set", "gs_sweep")


def db_sq(path):
    """Per-kernel launch means of whatever SQ_* counters one --pmc pass collected, plus the ratios quoted in DESIGN.md."""
    import sqlite3

    cur = sqlite3.connect(path).cursor()
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for name, counter, value in cur.execute("select kernel_name, counter_name, value from counters_collection"):
        agg[short(name)][counter].append(float(value))
    out = {}
    for k, d in agg.items():
        if not k.startswith("slp::k_") or not any(w in k for w in SQ_KERNELS):
            continue
        r = {c: sum(v) / len(v) for c, v in d.items()}
        r["launches"] = len(next(iter(d.values())))
        for a, b in (("SQ_WAIT_ANY", "SQ_WAVE_CYCLES"), ("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"), ("SQ_LDS_IDX_ACTIVE", "SQ_ACTIVE_INST_ANY"),
                     ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY")):
            if a in r and b in r and r[b]:
                r[a + "/" + b] = r[a] / r[b]
        out[k] = r
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    {"stats": stats, "pmc": pmc, "db-stats": db_stats, "db-pmc": db_pmc, "db-sq": db_sq}[sys.argv[1]](*sys.argv[2:])
