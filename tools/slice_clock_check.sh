cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/clk && mkdir -p gpurun_out/clk
PA_SLICE=50 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/clk -o clk -- python tools/quick_bench.py 100000x8192 > gpurun_out/clk/log.txt 2>&1
tail -1 gpurun_out/clk/log.txt
python - <<'PY'
import csv,glob,collections
agg=collections.defaultdict(list)
for f in glob.glob('gpurun_out/clk/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'slice_kernel' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
dur=[]
for f in glob.glob('gpurun_out/clk/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        if 'slice_kernel' in r['Kernel_Name']: dur.append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6)
print({k:sum(v)/len(v) for k,v in agg.items()}, dur)
if 'GRBM_GUI_ACTIVE' in agg and dur: print('clock GHz', sum(agg['GRBM_GUI_ACTIVE'])/len(agg['GRBM_GUI_ACTIVE'])/8/(sum(dur)/len(dur)*1e-3)/1e9)
PY
rocm-smi --showpower --showtemp 2>/dev/null | grep -i "power\|temp" | head
