# round 6, late: k_lm_prepare's Y records through LDS; per-kernel times of a call
set -x
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
python -m pytest tests/test_gpu_ba.py -x -q -m gpu 2>&1 | tail -3
OVS_BA_TRACE=1 python tools/time_lba.py device 5 2>&1 | grep -E "total" | tail -2
run() {   # tag, command...
    local tag=$1; shift
    rm -rf /tmp/rp_$tag
    ( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/rp_$tag -o $tag -- "$@" > /tmp/rp_$tag.log 2>&1 )
    local db=$(find /tmp/rp_$tag -name "*.db" | head -1)
    [ -n "$db" ] && python tools/rocpd_summary.py $db "$tag" | tee $OUT/r06al_${tag}_kernel_stats.txt
}
run lba python $GRAFT_REPO_ROOT/tools/time_lba.py device 3
