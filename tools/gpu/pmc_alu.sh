# ALU-side PMC counters of the pipeline kernels (two passes: kernel-trace + pmc only, as the guide prescribes)
TAG=${1:-r01_f}
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc
B="python bench.py ${BARGS} --extras none --no-parity --steps 1 --warmup 0 --no-cpu-baseline --no-stage-timing"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM -d gpurun_out/pmc -o alu1 -- $B > gpurun_out/pmc_alu1.log 2>&1; echo "alu1 rc $?"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d gpurun_out/pmc -o alu2 -- $B > gpurun_out/pmc_alu2.log 2>&1; echo "alu2 rc $?"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY -d gpurun_out/pmc -o alu3 -- $B > gpurun_out/pmc_alu3.log 2>&1; echo "alu3 rc $?"
ls gpurun_out/pmc | head
ARGS=""
for c in SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM; do ARGS="$ARGS $c=gpurun_out/pmc/alu1_results.db"; done
for c in SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE; do ARGS="$ARGS $c=gpurun_out/pmc/alu2_results.db"; done
for c in SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY; do ARGS="$ARGS $c=gpurun_out/pmc/alu3_results.db"; done
python tools/pmc_extract.py gpurun_out/${TAG}_pmc_alu.json $ARGS 2>&1 | tail -3
rm -f gpurun_out/pmc/*.db
python - <<PY
import json
d=json.load(open('gpurun_out/${TAG}_pmc_alu.json'))
for k,v in d.items():
    if 'disco::' in k or 'k_' in k: print(k[:60], {c: round(x['per_dispatch']) for c,x in v.items()})
PY
tail -3 gpurun_out/pmc_alu1.log gpurun_out/pmc_alu2.log gpurun_out/pmc_alu3.log | cut -c1-200
