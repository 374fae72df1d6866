#!/bin/bash
# round 3: the shipped build (no experimental flavours) and the experimental build, whole GPU suite each
set -x
O=gpurun_out/r3p
mkdir -p $O
python -c "from volumerendering_amd import capi; print('experimental', capi.experimental_flavours())" > $O/which.txt 2>&1
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/pytest_shipped.txt 2>&1
rc=$?
tail -4 $O/pytest_shipped.txt
[ $rc -eq 0 ] || exit $rc
VR_EXPERIMENTAL_FLAVOURS=1 python -c "from volumerendering_amd import build as b; b.build_all()" > $O/build_exp.txt 2>&1 || { tail -5 $O/build_exp.txt; exit 1; }
python -c "from volumerendering_amd import capi; print('experimental', capi.experimental_flavours())" >> $O/which.txt 2>&1
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/pytest_experimental.txt 2>&1
rc=$?
tail -4 $O/pytest_experimental.txt
cat $O/which.txt
exit $rc
