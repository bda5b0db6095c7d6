#!/bin/bash
# k4_variants_build.sh NAME [-DK4X_...]: the K4 translation units ALONE, with the experiment macros of k4_variants.patch applied
# to a fresh copy of csrc/window_attn.hip, into scratch/_k4var/k4_NAME.so (git-ignored) for scratch/bench_k4_variants.py.
# Macros: K4X_NO_TBL_ATOMIC K4X_NO_P1 K4X_NO_P2 K4X_NO_STAGE K4X_NO_ELEM K4X_NO_TR K4X_NO_STORE K4X_NO_EPI_TABLE K4X_NO_EPI_PAD K4X_ROLL
cd "$(dirname "$0")"; mkdir -p _k4var
cp ../mask_bev_amd/csrc/window_attn.hip ../mask_bev_amd/csrc/window_attn_f16.hip _k4var/ && patch -s _k4var/window_attn.hip k4_variants.patch || exit 1
n=$1; shift
cd _k4var && hipcc -shared -fPIC -O3 -std=c++17 --offload-arch=gfx950 -fno-fast-math -Wno-unused-function -I../../mask_bev_amd/csrc -I../../include "$@" window_attn.hip window_attn_f16.hip -o k4_$n.so
