#!/bin/bash
# Which of the library's streams share a hardware queue decides how the assembly's chains interleave: shift the mapping with
# k throwaway streams in front (FI_DUMMY_STREAMS: a timing build, tools/build_variant.sh) and watch the assembly time.
# usage (GPU box): bash tools/r6_dummy_streams.sh
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6
for k in 0 1 2 3 4 5 6; do
  FI_DUMMY_STREAMS=$k python bench.py --steps 6 --warmup 2 --cpu-side 0 --no-accuracy --no-cold --no-host-io --no-roofline-512 > gpurun_out/r6/b_dummy$k.json 2>/dev/null || exit 1
  echo -n "dummy streams $k: "; python3 tools/bench_brief.py gpurun_out/r6/b_dummy$k.json
done
