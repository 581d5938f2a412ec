#!/bin/bash
# build libemoasr_hip variants with pieces of attn_bwd_dq2 left out (timing experiments only; results wrong)
set -e
cd "$(dirname "$0")/.."
mkdir -p emoasr_amd/build/variants
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -Wno-unused-value -Wno-comment -ffp-contract=off"
OBJS=$(ls emoasr_amd/build/*.o | grep -v attention.o)
for n in "$@"; do
  ( /opt/rocm/bin/hipcc $FLAGS -DEMO_DQ_SKIP=$n -c emoasr_amd/csrc/attention.hip -o emoasr_amd/build/variants/attention_$n.o 2>/dev/null &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o emoasr_amd/build/variants/lib_skip$n.so $OBJS emoasr_amd/build/variants/attention_$n.o ) &
done
wait
ls -la emoasr_amd/build/variants/*.so
