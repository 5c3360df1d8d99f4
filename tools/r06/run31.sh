#!/bin/bash
cd "$(dirname "$0")/../.."
V2X_CENSUS_SHOW=v2v_message timeout 600 python tools/train_op_census.py v2v 2 2>&1 | grep "^# " | cut -c1-120
V2X_CENSUS_SHOW=v2v_message timeout 600 python tools/train_op_census.py v2v 8 2>&1 | grep "^# " | cut -c1-120
