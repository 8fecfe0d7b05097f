cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python tools/end_to_end_demo.py 16 /tmp/zds 2>&1 | tail -3
ls /tmp/zds/raw/0-6/zs_hat | head -3; ls /tmp/zds/raw/0-6/zs_hat | wc -l
