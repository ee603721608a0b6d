#!/bin/bash
# Dev-only: CPU build of the device step machine diffed against the oracle.  usage: tools/host_sim/build.sh && /tmp/host_sim [T] [K]
set -e
cd "$(dirname "$0")/../.."
gcc -O2 -std=gnu11 -ffp-contract=off -c oracle/pokerl_oracle.c -o /tmp/host_sim_oracle.o
g++ -std=c++20 -O2 -ffp-contract=off ${PK_SIM_DEFS:-} -DPK_HOST_SIM -include tools/host_sim/hip_shim.h -I. tools/host_sim/host_sim.cpp /tmp/host_sim_oracle.o -o /tmp/host_sim
