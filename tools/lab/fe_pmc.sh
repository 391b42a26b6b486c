# DEV TOOL: MFMA / LDS counters of the frontend's kernels (separate --pmc pass, kernel trace only)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/fp && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d /tmp/fp -o f -- python $GRAFT_REPO_ROOT/tools/lab/att_bench.py > /tmp/fp.log 2>&1
ls /tmp/fp /tmp/fp/* | head
python - <<'PY'
import csv, glob, collections, os
f = glob.glob('/tmp/fp/**/*counter_collection.csv', recursive=True)
print(f)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r['Kernel_Name'][:60]
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'SQ_WAVE_CYCLES': cnt[k] += 1
out = open(os.path.join(os.environ['GRAFT_REPO_ROOT'], 'gpurun_out', 'fe_pmc.txt'), 'w')
for k, v in acc.items():
    n = max(cnt[k], 1)
    line = f"{k:62s} launches {n:5d} " + " ".join(f"{c}={x / n / 1e6:.3f}M" for c, x in sorted(v.items()))
    print(line); out.write(line + "\n")
PY
