#!/usr/bin/env python3
"""Summarise rocprofv3 (rocpd SQLite) outputs into the text files kept under profiles/.
usage: python tools/rocpd_summary.py gpurun_out/prof_<tag> > profiles/<name>.txt"""
import glob
import os
import sqlite3
import sys


def main(root):
    for db in sorted(glob.glob(os.path.join(root, "*", "*_results.db"))):
        con = sqlite3.connect(db)
        print(f"== {os.path.relpath(db, root)}")
        rows = con.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
        if rows:
            print(f"{'kernel':100s} {'calls':>6s} {'avg_us':>12s} {'total_us':>12s} {'pct':>6s}")
            for n, c, tot, avg, pct in rows[:8]:
                print(f"{n[:100]:100s} {c:6d} {avg:12.1f} {tot:12.1f} {pct:6.2f}")
        k = con.execute("select name, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, scratch_size, grid_x, workgroup_x "
                        "from kernels where name like '%ros23%' order by id desc limit 1").fetchone()
        if k:
            print(f"   last ros23 dispatch: vgpr={k[1]} agpr={k[2]} sgpr={k[3]} lds={k[4]} scratch={k[5]} grid={k[6]} wg={k[7]}")
        try:
            cur = con.execute("select * from counters_collection limit 1")
            cols = [d[0] for d in cur.description]
            if cur.fetchone() is not None:
                nm = "kernel_name" if "kernel_name" in cols else "name"
                q = (f"select {nm}, counter_name, count(*), sum(value), avg(value) from counters_collection "
                     f"where ({nm} like '%ros23%' or {nm} like '%hychem%' or {nm} like '%cathode_%' or {nm} like '%auto_adj%' "
                     f"or {nm} like '%tsit5%' or {nm} like '%sort_steps%' or {nm} like 'k\_%' escape '\\') group by {nm}, counter_name")
                print(f"   {'counter':24s} {'dispatches':>10s} {'sum':>18s} {'per_dispatch':>18s}")
                for n, cn, cnt, s, a in con.execute(q):
                    print(f"   {cn:24s} {cnt:10d} {s:18.1f} {a:18.1f}   [{n[:60]}]")
        except sqlite3.OperationalError as e:
            print("   (no counters:", e, ")")
        print()


if __name__ == "__main__":
    main(sys.argv[1])
