#!/usr/bin/env python3
"""Summarise a rocprofv3 results .db (rocpd SQLite) into the text table kept under
profiles/: per-kernel calls / total / avg / min / max, plus PMC counters when the run
collected them.   python tools/rocprof_summary.py <results.db> [<out.txt>]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    out = []
    rows = cur.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, "
                       "max(end-start)/1e3 from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1.0
    out.append(f"{'kernel':96s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'%':>6s}")
    for r in rows:
        out.append(f"{r[0][:96]:96s} {r[1]:6d} {r[2]:10.3f} {r[3]:10.1f} {r[4]:10.1f} {r[5]:10.1f} {100*r[2]/tot:6.2f}")
    try:
        cols = [d[0] for d in cur.execute("select * from counters_collection limit 1").description]
        if "counter_name" in cols and "value" in cols:
            kcol = "kernel_name" if "kernel_name" in cols else ("name" if "name" in cols else None)
            if kcol:
                pm = cur.execute(f"select {kcol}, counter_name, count(*), sum(value), avg(value) from counters_collection "
                                 f"group by {kcol}, counter_name order by 4 desc").fetchall()
                if pm:
                    out.append("")
                    out.append(f"{'kernel':80s} {'counter':>16s} {'dispatches':>10s} {'sum':>18s} {'avg/dispatch':>18s}")
                    for r in pm:
                        out.append(f"{str(r[0])[:80]:80s} {r[1]:>16s} {r[2]:10d} {r[3]:18.1f} {r[4]:18.1f}")
    except sqlite3.Error as e:
        out.append(f"(no counters: {e})")
    text = "\n".join(out) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main()
