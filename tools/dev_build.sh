#!/bin/bash
# Quick experimental build of libatmrt.so: a scratch copy of csrc/ compiled with DEV=1 (spherical-earth kernel variants only,
# ~1 min instead of 4) into atm-raytracer_amd/csrc/dev/<name>/libatmrt.so, which travels to the GPU box and is selected with
#   ATMRT_LIB=atm-raytracer_amd/csrc/dev/<name>/libatmrt.so python bench.py ...
# usage: tools/dev_build.sh NAME [EXTRA compiler flags, e.g. -DATMRT_MARCH_WAVES=3]
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
NAME=${1:?name}; shift || true
W=/tmp/dev/$NAME
mkdir -p $W/atm-raytracer_amd/csrc $W/include
cp $REPO/include/*.h $W/include/
cp $REPO/atm-raytracer_amd/csrc/*.h $REPO/atm-raytracer_amd/csrc/*.hip $REPO/atm-raytracer_amd/csrc/Makefile $W/atm-raytracer_amd/csrc/
make -s -j8 -C $W/atm-raytracer_amd/csrc DEV=1 EXTRA="$*"
mkdir -p $REPO/atm-raytracer_amd/csrc/dev/$NAME
cp $W/atm-raytracer_amd/csrc/libatmrt.so $REPO/atm-raytracer_amd/csrc/dev/$NAME/libatmrt.so
echo "built atm-raytracer_amd/csrc/dev/$NAME/libatmrt.so"
