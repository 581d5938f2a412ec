#!/bin/bash
# time the attention backward with each variant library (see dq_skip_build.sh)
cd "$(dirname "$0")/.."
echo "baseline"; B=20 T=340 python tools/attn_bench.py 2>&1 | grep "materialise=True"
for f in emoasr_amd/build/variants/lib_skip*.so; do
  echo "$f"; EMOASR_HIP_LIB=$PWD/$f B=20 T=340 python tools/attn_bench.py 2>&1 | grep "materialise=True"
done
