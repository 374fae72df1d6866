#!/bin/bash
O=gpurun_out/r4r
mkdir -p $O
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; rc=$?; tail -6 $O/pytest.txt; exit $rc
