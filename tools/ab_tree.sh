#!/bin/bash
# Same-box A/B of two code states of the HOST side (Python) with the SAME built library:
#   tools/ab_tree.sh <commit>      (in the authoring container) extracts <commit>'s package + bench.py into ab_prev/ (git-ignored, travels
#                                  with the gpurun snapshot) and links the current build of the library into it
# on the GPU box:  (cd ab_prev && python bench.py ...)  against  python bench.py ...
set -e
cd "$(dirname "$0")/.."
rm -rf ab_prev; mkdir -p ab_prev
git archive "$1" one-peace_amd one_peace_amd.py bench.py oracle tests/golden 2>/dev/null | tar -x -C ab_prev || git archive "$1" one-peace_amd one_peace_amd.py bench.py oracle | tar -x -C ab_prev
mkdir -p ab_prev/one-peace_amd/lib ab_prev/profiles
cp one-peace_amd/lib/*.so one-peace_amd/lib/build.sha256 ab_prev/one-peace_amd/lib/
cp profiles/r5_gemm_hbm_traffic.json ab_prev/profiles/ 2>/dev/null || true
echo "ab_prev/ = $(git rev-parse --short "$1")"
