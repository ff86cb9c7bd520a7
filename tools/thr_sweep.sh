#!/bin/bash
# dense-tile threshold sweep (edge pixels per 8x32 tile from which the shared-term kernel takes over)
cd "${GRAFT_REPO_ROOT:-.}"
for t in 0 1 16 24 28 32 40 48 64; do
  echo "== SSG_DENSE_THR=$t"; SSG_DENSE_THR=$t python tools/ablate.py 2>&1 | head -1
done
