#!/usr/bin/env python3
"""The `rocprofv3 --kernel-trace --stats` summary from the database this rocprofv3 writes (rocpd .db: no CSV is emitted):
per kernel calls / average / total / share, and for one kernel (default: the headline's residual kernel, whose name is shared
by launches of several sizes -- the 2^21-row headline launch, h(x)'s 2^20-row launch, single 2^16-row systems of the
host-buffer calls, the cache-resident variant) the same split by duration class, so that the headline launch's own average
can be read beside bench.py's `roofline.kernel_us`.   python tools/prof_stats.py DIR_OR_DB [--kernel k_r1cs_sell_split<acx::Bn254Fr, 0>]"""
import argparse, glob, os, sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("path")
    ap.add_argument("--kernel", default="k_r1cs_sell_split<acx::Bn254Fr, 0>")
    ap.add_argument("--top", type=int, default=24)
    a = ap.parse_args()
    dbs = [a.path] if a.path.endswith(".db") else glob.glob(os.path.join(a.path, "**", "*.db"), recursive=True)
    for db in dbs:
        con = sqlite3.connect(db)
        rows = con.execute("select name, count(*), avg(end - start), sum(end - start), min(end - start), max(end - start) from kernels group by name "
                           "order by sum(end - start) desc").fetchall()
        total = sum(r[3] for r in rows) or 1
        print(f"# {db}: {sum(r[1] for r in rows)} kernel dispatches, {total / 1e6:.1f} ms of kernel time")
        print(f"{'calls':>7} {'avg us':>10} {'min us':>9} {'max us':>10} {'total ms':>10} {'%':>6}  kernel")
        for name, calls, avg, tot, mn, mx in rows[: a.top]:
            print(f"{calls:>7} {avg / 1e3:>10.2f} {mn / 1e3:>9.2f} {mx / 1e3:>10.2f} {tot / 1e6:>10.2f} {100 * tot / total:>6.1f}  {name.split('(')[0][:90]}")
        d = sorted(r[0] / 1e3 for r in con.execute("select end - start from kernels where name like ?", (f"%{a.kernel}%",)))
        if d:
            print(f"\n{a.kernel}: {len(d)} launches by duration class (the kernel name is shared by launches of several sizes)")
            for lo, hi, what in ((0, 20, "single 2^16-row systems (host-buffer calls, the literal configs[1] launch)"),
                                 (20, 90, "2^20-row launches (h(x)'s residual step)"),
                                 (90, 108, "2^21 rows below 108 us (the cache-resident variant -- same system 32 times, labelled, not the metric -- and the fastest headline launches)"),
                                 (108, 135, "2^21 rows from HBM: THE HEADLINE LAUNCH"), (135, 1e12, "2^21 rows, outliers (clock ramp, first launches)")):
                c = [x for x in d if lo <= x < hi]
                if c:
                    print(f"  {len(c):>6} launches  avg {sum(c) / len(c):8.2f} us  median {c[len(c) // 2]:8.2f} us   {what}")


if __name__ == "__main__":
    main()
