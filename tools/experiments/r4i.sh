#!/bin/bash
# whole GPU suite after the queue / mgpu / upload changes
O=gpurun_out/r4i
mkdir -p $O
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; rc=$?; tail -8 $O/pytest.txt; exit $rc
