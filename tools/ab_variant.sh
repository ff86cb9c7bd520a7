#!/bin/bash
# Build a VARIANT of the working tree's library for a same-box A/B (tools/ab_run.sh): the tree is copied to a temporary
# directory, `patch -p1 < <patch>` (or a sed script with -e) is applied there, the libraries are built and left in
# gpurun_ab/<tag>/ (travels with gpurun).  The working tree itself is not touched.
#   tools/ab_variant.sh <tag> <file.patch>          tools/ab_variant.sh <tag> -e 's/.../.../' <path relative to the root>
set -e
tag="$1"; shift; root="$(cd "$(dirname "$0")/.." && pwd)"
tmp=$(mktemp -d); mkdir -p "$tmp/wt"; cp -r "$root/ssl_amd" "$root/include" "$tmp/wt/"
rm -f "$tmp/wt/ssl_amd/csrc/"*.o "$tmp/wt/ssl_amd/csrc/"*.so; rm -rf "$tmp/wt/ssl_amd/csrc/prof"
if [ "$1" = "-e" ]; then sed -i -e "$2" "$tmp/wt/$3"; else pf="$(cd "$(dirname "$1")" && pwd)/$(basename "$1")"; (cd "$tmp/wt" && patch -p1 < "$pf"); fi
make -C "$tmp/wt/ssl_amd/csrc" -j8 > "$tmp/build.log" 2>&1 || { tail -30 "$tmp/build.log"; exit 1; }
mkdir -p "$root/gpurun_ab/$tag"; cp "$tmp/wt/ssl_amd/csrc/"libssg_hip*.so "$root/gpurun_ab/$tag/"
rm -rf "$tmp"; ls -la "$root/gpurun_ab/$tag"
