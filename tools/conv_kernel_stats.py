#!/usr/bin/env python3
"""Per-kernel summary (calls, average and total microseconds) out of a rocprofv3 rocpd database -- the form in which
`rocprofv3 --kernel-trace --stats` leaves its results on this image.
    python tools/conv_kernel_stats.py gpurun_out/conv_prof/conv_results.db [substring]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    like = f"%{sys.argv[2]}%" if len(sys.argv) > 2 else "%"
    q = ("select s.kernel_name, count(*), avg(d.end - d.start) / 1e3, sum(d.end - d.start) / 1e3, max(s.arch_vgpr_count), "
         "max(d.grid_size_x), max(d.grid_size_y) from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
         "where s.kernel_name like ? group by 1 order by 4 desc")
    print("kernel,calls,avg_us,total_us,vgpr,grid_x,grid_y")
    for r in db.execute(q, (like,)):
        print(f"{r[0]},{r[1]},{r[2]:.1f},{r[3]:.1f},{r[4]},{r[5]},{r[6]}")


if __name__ == "__main__":
    main()
