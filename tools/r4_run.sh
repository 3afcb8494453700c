#!/bin/bash
# tools/r4_run.sh "<ENV1=.. ENV2=..>" ... : one tools/r4_acc.py run per argument (its words are environment assignments)
cd "$GRAFT_REPO_ROOT" || exit 1
for spec in "$@"; do
  env $spec NAME="$spec" python3 tools/r4_acc.py 2>&1 | tail -2
done
