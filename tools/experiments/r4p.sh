#!/bin/bash
O=gpurun_out/r4p
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_p2_gpu.py -x -q -m gpu -k "rank_shares" > $O/pytest.txt 2>&1; tail -30 $O/pytest.txt
