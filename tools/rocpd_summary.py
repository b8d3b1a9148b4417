"""Summarise a rocprofv3 rocpd sqlite database: per-kernel time stats and per-kernel mean counter values."""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*", "", name)
    return name[-70:]


def main(path):
    c = sqlite3.connect(path)
    cols = [d[1] for d in c.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [x for x in cols if "name" in x][0]
    rows = c.execute(f"select {name_col}, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) from kernels group by {name_col} order by 6 desc").fetchall()
    tot = sum(r[5] for r in rows) or 1
    print(f"{'kernel':70s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'total_ms':>10s} {'%':>6s}")
    for r in rows[:12]:
        print(f"{short(r[0]):70s} {r[1]:6d} {r[2] / 1e3:10.1f} {r[3] / 1e3:10.1f} {r[4] / 1e3:10.1f} {r[5] / 1e6:10.3f} {100 * r[5] / tot:6.1f}")
    try:
        ccols = [d[1] for d in c.execute("pragma table_info(counters_collection)")]
        kn = "kernel_name" if "kernel_name" in ccols else name_col
        rows = c.execute(f"select {kn}, counter_name, avg(value), count(*) from counters_collection group by {kn}, counter_name order by 1,2").fetchall()
        if rows:
            print("\nmean counter value per dispatch")
            for r in rows:
                print(f"{short(r[0]):70s} {r[1]:32s} {r[2]:18.1f}  (n={r[3]})")
    except sqlite3.Error as e:
        print("no counters:", e)


if __name__ == "__main__":
    main(sys.argv[1])
