"""Top kernels of a rocprofv3 results database (the default output format of rocprofv3 7.x): python tools/lab/db_top.py <dir> [n]"""
import glob
import sqlite3
import sys

f = glob.glob(sys.argv[1].rstrip("/") + "/*.db")[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
db = sqlite3.connect(f)
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
q = (f"select s.kernel_name, count(*), sum(d.end-d.start)/1e6, avg(d.end-d.start)/1e6 from {kd} d join {ks} s on d.kernel_id=s.id "
     f"group by s.kernel_name order by 3 desc limit {n}")
for r in db.execute(q):
    print(f"{r[0][:80]:80s} calls {r[1]:5d}  total {r[2]:10.2f} ms  avg {r[3]:9.3f} ms")
