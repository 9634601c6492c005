set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
scripts/ubench/store_width > gpurun_out/r06_store_width2.txt 2>&1; cat gpurun_out/r06_store_width2.txt
