#!/usr/bin/env python3
"""Per-kernel PMC sums from rocprofv3 rocpd databases -> JSON.
Usage: tools/pmc_extract.py out.json COUNTER=path/to/results.db [COUNTER=...]
For every kernel name: number of dispatches and the per-dispatch mean of the counter."""
import json
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'HIP_vector_type<float, 2u>', 'c32', name)
    name = re.sub(r'\(.*$', '', name)
    return name[:120]


def main():
    out = {}
    for spec in sys.argv[2:]:
        counter, db = spec.split('=', 1)
        cur = sqlite3.connect(db).cursor()
        cols = [d[0] for d in cur.execute('select * from counters_collection limit 1').description]
        # columns of interest: kernel name, counter name, value, dispatch id
        name_col = 'kernel_name' if 'kernel_name' in cols else 'name'
        rows = cur.execute(f'select {name_col}, counter_name, sum(value), count(distinct dispatch_id) from counters_collection '
                           f'group by {name_col}, counter_name').fetchall()
        for kname, cname, total, ndisp in rows:
            if cname != counter:
                continue
            out.setdefault(short(kname), {})[counter] = {'sum': total, 'dispatches': ndisp, 'per_dispatch': total / max(ndisp, 1)}
        out.setdefault('_columns', cols)
    json.dump(out, open(sys.argv[1], 'w'), indent=1)


if __name__ == '__main__':
    main()
