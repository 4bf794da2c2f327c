#!/usr/bin/env python3
"""Kernel durations out of a rocprofv3 result database (rocpd .db, what `rocprofv3 --kernel-trace -d DIR -- prog` writes on
ROCm 7): per kernel name calls / total / mean / min / max microseconds and share of the device time (the table
`--stats` prints), or -- with --dispatches PATTERN -- every dispatch of the kernels whose name contains PATTERN in launch
order (start offset, duration, grid, workgroup).
    python tools/rocpd_kernels.py DIR_OR_DB [--dispatches PATTERN] [--csv] [--top N]"""
import glob
import os
import sqlite3
import sys


def open_db(path):
    if os.path.isdir(path):
        found = sorted(glob.glob(os.path.join(path, "**", "*.db"), recursive=True))
        if not found:
            sys.exit(f"no .db under {path}")
        path = found[-1]
    con = sqlite3.connect(path)
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    return con, sym, disp


def short(name, width=110):
    name = name.replace(".kd", "")
    return name if len(name) <= width else name[:width - 3] + "..."


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    con, sym, disp = open_db(args[0])
    csv = "--csv" in sys.argv
    if "--dispatches" in sys.argv:
        pat = sys.argv[sys.argv.index("--dispatches") + 1]
        rows = list(con.execute(
            f"select s.kernel_name, d.start, d.end, d.grid_size_x, d.grid_size_y, d.grid_size_z, d.workgroup_size_x "
            f"from {disp} d join {sym} s on d.kernel_id = s.id where s.kernel_name like ? order by d.start", (f"%{pat}%",)))
        t0 = rows[0][1] if rows else 0
        for name, st, en, gx, gy, gz, wx in rows:
            print(f"{(st - t0) / 1e3:12.1f} us  {(en - st) / 1e3:9.2f} us  grid {gx}x{gy}x{gz} wg {wx}  {short(name, 80)}")
        return
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 40
    rows = list(con.execute(
        f"select s.kernel_name, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start) "
        f"from {disp} d join {sym} s on d.kernel_id = s.id group by s.kernel_name order by 3 desc"))
    total = sum(r[2] for r in rows) or 1
    if csv:
        print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"')
        for name, n, tot, mn, mx in rows:
            print(f'"{name}",{n},{tot},{tot / n:.1f},{100.0 * tot / total:.3f},{mn},{mx}')
        return
    print(f"{'calls':>7} {'total ms':>10} {'mean us':>9} {'min us':>9} {'max us':>9} {'share':>6}  kernel")
    for name, n, tot, mn, mx in rows[:top]:
        print(f"{n:7d} {tot / 1e6:10.3f} {tot / n / 1e3:9.2f} {mn / 1e3:9.2f} {mx / 1e3:9.2f} {100.0 * tot / total:5.1f}%  {short(name)}")


if __name__ == "__main__":
    main()
