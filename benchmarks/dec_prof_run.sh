#!/bin/bash
# Builds a -DDEC_PROF copy of the library in a scratch directory on the GPU box and runs the decoder phase probe with it.
set -e
cd "$(dirname "$0")/.."
ROOT=$PWD
rm -rf /tmp/scp_prof && mkdir -p /tmp/scp_prof && cp -r scpair_amd include benchmarks /tmp/scp_prof/
cd /tmp/scp_prof && rm -rf scpair_amd/csrc/build scpair_amd/libscpair_hip.so
EXTRA_FLAGS="-DDEC_PROF" bash scpair_amd/csrc/build.sh > /dev/null 2>&1
timeout -k 10 120 python3 benchmarks/dec_phase_probe.py
