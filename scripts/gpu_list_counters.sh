mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/counters_list.txt 2>&1
grep -c "" $GRAFT_REPO_ROOT/gpurun_out/counters_list.txt
