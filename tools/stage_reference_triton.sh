#!/bin/bash
# Stages the reference's OWN Triton attention (three files, unmodified) from /root/reference into _ref_scratch/ -- untracked,
# git-ignored, shipped to the GPU box by gpurun -- for the tools-only comparator tools/ref_triton_compare.py.  Nothing under
# _ref_scratch/ is imported by the package, the tests or bench.py.
set -e
cd "$(dirname "$0")/.."
R=/root/reference/generative_recommenders
D=_ref_scratch/generative_recommenders
rm -rf _ref_scratch
mkdir -p $D/ops/triton
cp $R/common.py $D/common.py
cp $R/ops/triton/triton_hstu_attention.py $R/ops/triton/triton_attention_utils.py $D/ops/triton/
grep -qx "_ref_scratch/" .gitignore || echo "_ref_scratch/" >> .gitignore
echo staged: $(find _ref_scratch -type f | wc -l) files
