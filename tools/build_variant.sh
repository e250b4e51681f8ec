#!/bin/bash
# Builds a variant of the HIP library into build/variants/<name>.so for tools/ab_variants.py.
#   tools/build_variant.sh <name> [git-rev|WORK] [extra hipcc flags...]
# git-rev: build csrc/ as of that commit (default WORK = the working tree).
set -eo pipefail
name=$1; rev=${2:-WORK}; shift; shift || true
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$root/build/variants"
src="$root/smoothed_particle_hydrodynamics_amd/csrc"
if [ "$rev" != "WORK" ]; then
   tmp=$(mktemp -d)
   mkdir -p "$tmp/smoothed_particle_hydrodynamics_amd" "$tmp/include"
   git -C "$root" archive "$rev" smoothed_particle_hydrodynamics_amd/csrc include | tar -x -C "$tmp"
   src="$tmp/smoothed_particle_hydrodynamics_amd/csrc"
fi
# (SPH_ABLATE hooks only compile in a declared diagnostic build, which the Python binding loads
# only with SPH_HIP_ALLOW_DIAGNOSTIC=1: csrc/full_tiled.h)
diag=""
case " $* " in *SPH_ABLATE*) diag="-DSPH_DIAGNOSTIC_BUILD";; esac
(cd "$src" && hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wall -ldl $diag "$@" \
   -o "$root/build/variants/$name.so" sph_hip.hip)
echo "built build/variants/$name.so from $rev $*"
