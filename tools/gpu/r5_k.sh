#!/bin/bash
# Round 5, last pass: final_pass.sh on the final sources, then -- with the counter files of THIS pass in place, as the next plain run of the command will
# find them -- smoke() and the plain bench line once more (so that the committed line carries its `traffic` fields)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
bash tools/gpu/final_pass.sh r05_zz 2>&1 | grep -v "^\"void\|^void\|^disco::\|SQ_WAVES" | tail -40
for c in C3 C5 C2 C2x4000; do cp gpurun_out/r05_zz_${c}_pmc_traffic.json profiles/pmc_traffic_${c}.json; done
cp gpurun_out/pmc_alu_online1.json profiles/pmc_alu_online1.json
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
mv gpurun_out/r05_zz_bench_default.json gpurun_out/r05_zz_bench_default_before_counters.json
timeout 900 python bench.py > gpurun_out/r05_zz_bench_default.json 2> gpurun_out/r05_zz_bench_default.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r05_zz_bench_default.json').read().strip().splitlines()[-1])
print(json.dumps(d['summary'])[:1600])
PY
