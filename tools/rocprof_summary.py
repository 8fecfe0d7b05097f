#!/usr/bin/env python3
"""Condense a rocprofv3 rocpd database (the `-d DIR -o NAME` output of `rocprofv3 --kernel-trace --stats`)
into a small per-kernel CSV: calls, total/avg/min/max duration (us), launch geometry, VGPR/LDS/scratch.
Usage: tools/rocprof_summary.py gpurun_out/prof/r01_results.db profiles/r01_kernel_stats.csv [name-filter]"""
import csv
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'HIP_vector_type<float, 2u>', 'c32', name)
    name = re.sub(r'HIP_vector_type<float, 4u>', 'float4', name)
    name = re.sub(r'\(.*$', '', name)
    return name[:100]


def main():
    db, out = sys.argv[1], sys.argv[2]
    filt = sys.argv[3] if len(sys.argv) > 3 else None
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(grid_x), max(workgroup_x), "
        "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size) from kernels group by name "
        "order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows)
    with open(out, 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['kernel', 'calls', 'total_us', 'avg_us', 'min_us', 'max_us', 'pct_of_gpu_time', 'grid_x', 'wg_x', 'vgpr', 'agpr',
                    'sgpr', 'lds_bytes', 'scratch_bytes'])
        for r in rows:
            if filt and filt not in r[0]:
                continue
            w.writerow([short(r[0]), r[1], round(r[2] / 1e3, 1), round(r[3] / 1e3, 2), round(r[4] / 1e3, 2), round(r[5] / 1e3, 2),
                        round(100.0 * r[2] / total, 2)] + list(r[6:]))


if __name__ == '__main__':
    main()
