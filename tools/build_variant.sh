#!/bin/bash
# A/B aid: build a variant of the library with extra compile flags for the kernel file(s) into llama-nuts-and-bolts_amd/variants/<name>.so
#   tools/build_variant.sh <name> <flags...>     then:  LNB_SO=$PWD/llama-nuts-and-bolts_amd/variants/<name>.so python tools/kernel_ab.py
set -e
cd "$(dirname "$0")/../llama-nuts-and-bolts_amd/csrc"
name=$1; shift
mkdir -p ../variants /tmp/lnb_variant_$name
make -s                                                      # the default objects (lnb_api.o, ...) are shared
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-function -Wno-unused-value -Wno-unused-result"
/opt/rocm/bin/hipcc $FLAGS "$@" -c lnb_kernels.hip -o /tmp/lnb_variant_$name/lnb_kernels.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/$name.so /tmp/lnb_variant_$name/lnb_kernels.o lnb_fast.o lnb_api.o lnb_pipeline.o lnb_checkpoint.o lnb_tokenizer.o -ldl
echo "built ../variants/$name.so"
