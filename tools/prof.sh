#!/bin/bash
# rocprofv3 kernel stats of one command on the GPU box: tools/prof.sh <name> <program> [args...]   (env passes through)
# summary -> gpurun_out/<name>_stats.md, raw csv -> gpurun_out/<name>_kernel_stats.csv
name=$1; shift
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
rm -rf gpurun_out/prof_$name
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$name -o $name -- "$@" > gpurun_out/${name}_run.log 2>&1 || { tail -20 gpurun_out/${name}_run.log; exit 1; }
f=$(find gpurun_out/prof_$name -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/${name}_kernel_stats.csv
python3 tools/prof_summary.py gpurun_out/${name}_kernel_stats.csv "$name" > gpurun_out/${name}_stats.md
rm -rf gpurun_out/prof_$name
cat gpurun_out/${name}_stats.md
