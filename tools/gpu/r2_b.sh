cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for pid in 164 165; do echo "=== pid $pid"; DISCO_DBG_PID=$pid DISCO_DBG_T=47 DISCO_HIP_LIB=$PWD/exp_libs/libdisco_dbg.so timeout 300 python tools/gpu/dbg_online3.py 2>&1 | grep -v amdgpu.ids | grep "DBG\|n bad" | cut -c1-300; done > gpurun_out/r2b_dbg4.log 2>&1
cat gpurun_out/r2b_dbg4.log | head -120
