#!/bin/bash
# Dev-only: ASan + UBSan build of the host simulator (the device step machine compiled for the CPU) and the oracle.
#   tools/host_sim/sanitize.sh && /tmp/host_sim_san 64 200 && /tmp/host_sim_san fuzz 90 32 200
set -e
cd "$(dirname "$0")/../.."
gcc -O1 -g -std=gnu11 -ffp-contract=off -fsanitize=address,undefined -c oracle/pokerl_oracle.c -o /tmp/hs_oracle_san.o
g++ -std=c++20 -O1 -g -ffp-contract=off -fsanitize=address,undefined -DPK_HOST_SIM -include tools/host_sim/hip_shim.h -I. \
    tools/host_sim/host_sim.cpp /tmp/hs_oracle_san.o -o /tmp/host_sim_san
