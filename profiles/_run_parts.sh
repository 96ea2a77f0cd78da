cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_pipe -o pipe -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --algo 4 --steps 20 --warmup 5 --profile-steps 0 > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/profiles/pipe_roles.py $GRAFT_REPO_ROOT/gpurun_out/prof_pipe/pipe_kernel_trace.csv
