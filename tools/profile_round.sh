#!/bin/bash
# Runs on the GPU box (gpurun): the round's rocprofv3 passes over the default bench workload.
#   tools/profile_round.sh <tag>      -> gpurun_out/prof_<tag>/{bench.json, kt/, pf/, pw/, sq/}
# Passes are separate on purpose (MI355X_MICROARCH.md: never combine --pmc with the trace domains; FETCH_SIZE and
# WRITE_SIZE in their own runs).  Summaries are made by tools/rocpd_kernel_stats.py / pmc_traffic.py / pmc_sq_summary.py.
#   tools/profile_round.sh --config <lowerbound|upperbound|v2vnet|when2com|who2com|seg> <tag>
#       the same four passes over tools/config_run.py (one BASELINE.json config, 64 frames, eager launches)
#       -> gpurun_out/prof_<tag>_<config>/{layers.json, kernel_stats.csv, pmc_traffic.json, pmc_sq.csv, table.txt}
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
if [ "${1:-}" = "--config" ]; then
    CFG=$2
    TAG=${3:-vX}
    OUT=gpurun_out/prof_${TAG}_$CFG
    mkdir -p $OUT
    RUN="python3 tools/config_run.py --config $CFG"
    $RUN --reps 5 > $OUT/run.json 2> $OUT/run.err
    $RUN --reps 3 --layers $OUT/layers.json > /dev/null 2>> $OUT/run.err
    rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt --output-format csv -- $RUN --reps 5 > $OUT/kt.log 2>&1
    rocprofv3 --pmc FETCH_SIZE -d $OUT/pf -o pf --output-format csv -- $RUN --reps 1 > $OUT/pf.log 2>&1
    rocprofv3 --pmc WRITE_SIZE -d $OUT/pw -o pw --output-format csv -- $RUN --reps 1 > $OUT/pw.log 2>&1
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/sq -o sq --output-format csv -- $RUN --reps 1 > $OUT/sq.log 2>&1
    python3 tools/pmc_traffic.py $(find $OUT/pf -name "*counter_collection.csv") $(find $OUT/pw -name "*counter_collection.csv") $OUT/pmc_traffic.json > $OUT/pmc_traffic.txt 2>&1
    python3 tools/pmc_sq_summary.py $(find $OUT/sq -name "*counter_collection.csv") $OUT/pmc_sq.csv > /dev/null 2>&1
    cp $(find $OUT/kt -name "*kernel_stats.csv") $OUT/kernel_stats.csv 2>/dev/null
    python3 tools/profile_table.py $OUT/layers.json $OUT/kernel_stats.csv $OUT/pmc_traffic.json $OUT/pmc_sq.csv > $OUT/table.txt 2>> $OUT/run.err
    rm -rf $OUT/kt $OUT/pf $OUT/pw $OUT/sq 2>/dev/null
    cat $OUT/run.json $OUT/table.txt
    exit 0
fi
TAG=${1:-vX}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt --output-format csv -- python3 bench.py --steps 20 --warmup 3 --graph 0 --no-cpu-baseline --no-gpu-baseline --no-extras --no-calibration --no-shard-check > $OUT/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pf -o pf --output-format csv -- python3 bench.py --steps 2 --warmup 1 --graph 0 --no-cpu-baseline --no-gpu-baseline --no-extras --no-calibration --no-shard-check --no-roofline > $OUT/pf.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pw -o pw --output-format csv -- python3 bench.py --steps 2 --warmup 1 --graph 0 --no-cpu-baseline --no-gpu-baseline --no-extras --no-calibration --no-shard-check --no-roofline > $OUT/pw.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/sq -o sq --output-format csv -- python3 bench.py --steps 2 --warmup 1 --graph 0 --no-cpu-baseline --no-gpu-baseline --no-extras --no-calibration --no-shard-check --no-roofline > $OUT/sq.log 2>&1
find $OUT -name "*.csv" | head -20
python3 tools/pmc_traffic.py $(find $OUT/pf -name "*counter_collection.csv") $(find $OUT/pw -name "*counter_collection.csv") $OUT/pmc_traffic.json > $OUT/pmc_traffic.txt 2>&1
python3 tools/pmc_sq_summary.py $(find $OUT/sq -name "*counter_collection.csv") $OUT/pmc_sq.csv > /dev/null 2>&1
cp $(find $OUT/kt -name "*kernel_stats.csv") $OUT/kernel_stats.csv 2>/dev/null
# keep the pull small: drop the kernel trace, keep the raw counter csv files next to their summaries
cp $(find $OUT/pf -name "*counter_collection.csv") $OUT/pmc_fetch_size.csv 2>/dev/null
cp $(find $OUT/pw -name "*counter_collection.csv") $OUT/pmc_write_size.csv 2>/dev/null
rm -rf $OUT/kt/*kernel_trace.csv $OUT/kt/*/*kernel_trace.csv $OUT/pf $OUT/pw $OUT/sq 2>/dev/null
ls -la $OUT
