#!/bin/bash
# development aid: the library with extra compiler flags, as trust4_amd/variants/NAME/libt4hip.so (for T4_LIB=... A/B runs)
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p trust4_amd/variants/$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared "$@" -o trust4_amd/variants/$name/libt4hip.so trust4_amd/csrc/t4_api.hip trust4_amd/csrc/t4_assembler.cpp -lz -lpthread -ldl 2>&1 | grep -i " error" | head
echo "built $name"
