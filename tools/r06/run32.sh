#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r06_run32; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x > $O/suite.txt 2>&1
echo "rc=$?" >> $O/suite.txt; tail -6 $O/suite.txt
