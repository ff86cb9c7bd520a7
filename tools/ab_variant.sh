#!/bin/bash
# Build a VARIANT of the working tree's library for a same-box A/B (tools/r6_ab.sh / ab_run.sh): the tree (with its
# up-to-date objects, so that only what the variant touches is recompiled) is copied to a temporary directory, the
# operations are applied there in order, the libraries are built and left in gpurun_ab/<tag>/ (travels with gpurun).
# The working tree itself is not touched.
#   tools/ab_variant.sh <tag> [<file.patch> | -e '<sed script>' <path relative to the root>] ...
set -e
tag="$1"; shift; root="$(cd "$(dirname "$0")/.." && pwd)"
make -C "$root/ssl_amd/csrc" -j8 > /dev/null 2>&1 || true
tmp=$(mktemp -d); mkdir -p "$tmp/wt"; cp -rp "$root/ssl_amd" "$root/include" "$tmp/wt/"
while [ $# -gt 0 ]; do
  if [ "$1" = "-e" ]; then sed -i -e "$2" "$tmp/wt/$3"; shift 3
  else pf="$(cd "$(dirname "$1")" && pwd)/$(basename "$1")"; (cd "$tmp/wt" && patch -s -p1 < "$pf"); shift; fi
done
make -C "$tmp/wt/ssl_amd/csrc" -j8 > "$tmp/build.log" 2>&1 || { tail -30 "$tmp/build.log"; exit 1; }
mkdir -p "$root/gpurun_ab/$tag"; cp "$tmp/wt/ssl_amd/csrc/"libssg_hip*.so "$root/gpurun_ab/$tag/"
rm -rf "$tmp"; echo "built gpurun_ab/$tag"
