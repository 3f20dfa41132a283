#!/bin/bash
# ASan + UBSan over the CPU-side code (oracle C, C++ host mirror); GPU sanitizers are not available.
set -e
cd "$(dirname "$0")/.."
make -C oracle -s asan
g++ -O1 -g -std=c++17 -ffp-contract=off -fsanitize=address,undefined -fPIC -shared -Iinclude -Iray-marching_amd/csrc \
    -o /tmp/librm_host_asan.so ray-marching_amd/csrc/rm_host.cpp
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0 \
    python tools/sanitize_cpu.py
