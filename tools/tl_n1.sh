cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/rp_tl; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/rp_tl -o r -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --cpu-sample 0 --single-shots 0 --e2e-reads 0 > /dev/null 2>&1; python3 $GRAFT_REPO_ROOT/tools/timeline.py /tmp/rp_tl 0 | grep "k_dm_\|total"
