#!/bin/bash
# Runs on the GPU box: time AND counters of the pixel-fragment read patterns (tools/lds_conflict_probe.hip `counters` mode).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
OUT=gpurun_out/lds_probe_counters.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/lds_conflict_probe.hip -o /tmp/lds_probe || exit 1
/tmp/lds_probe counters > /tmp/lds_times.txt
rm -rf /tmp/lds_pmc
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS -d /tmp/lds_pmc -o p --output-format csv -- /tmp/lds_probe counters > /tmp/lds_pmc.log 2>&1
python3 - <<'PY' > $OUT
import csv, glob, collections
times = [l.strip() for l in open('/tmp/lds_times.txt') if l.startswith('dispatch')]
f = glob.glob('/tmp/lds_pmc/**/*counter_collection.csv', recursive=True)[0]
rows = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if 'read_loop' not in r['Kernel_Name']: continue
    rows.setdefault(int(r['Dispatch_Id']), {})[r['Counter_Name']] = float(r['Counter_Value'])
print("pattern (4 x 16-byte slots per pixel; lane = (fj, fq); 8 reads in flight, 4 waves per CU) | ns per read | SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE | LDS cycles per ds_read_b128")
for t, (d, c) in zip(times, sorted(rows.items())):
    print("%-62s | conflicts %5.1f %% of LDS cycles | %.2f cycles per read" % (t[9:], 100 * c['SQ_LDS_BANK_CONFLICT'] / max(c['SQ_LDS_IDX_ACTIVE'], 1), c['SQ_LDS_IDX_ACTIVE'] / max(c['SQ_INSTS_LDS'], 1)))
PY
cat $OUT
