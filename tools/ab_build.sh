#!/bin/bash
# Build the libraries of another commit next to the working tree's for same-box A/B runs:
#   tools/ab_build.sh <git-rev> <tag>   ->  gpurun_ab/<tag>/libssg_hip.so, libssg_hip_prof.so   (travels with gpurun)
# then on the GPU box: tools/ab_run.sh <tagA> <tagB> [bench args]   (alternates the two builds in one call)
set -e
rev="$1"; tag="$2"; root="$(cd "$(dirname "$0")/.." && pwd)"
tmp=$(mktemp -d); git -C "$root" worktree add -f --detach "$tmp/wt" "$rev" > /dev/null 2>&1
make -C "$tmp/wt/ssl_amd/csrc" -j8 > /dev/null 2>&1
mkdir -p "$root/gpurun_ab/$tag"; cp "$tmp/wt/ssl_amd/csrc/"libssg_hip*.so "$root/gpurun_ab/$tag/"
git -C "$root" worktree remove --force "$tmp/wt"; rm -rf "$tmp"; ls -la "$root/gpurun_ab/$tag"
