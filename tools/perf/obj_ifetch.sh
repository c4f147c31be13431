cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
hipcc -O3 --offload-arch=gfx950 -std=c++17 -Wno-unused-value -Wno-unused-result -DOBJ_ONLY_P=10 -o $R/tools/perf/_bin/obj_bench_c $R/tools/perf/obj_bench.hip -L$R/rvspecfit_amd -l:librvsgpu.so -Wl,-rpath,$R/rvspecfit_amd 2>/dev/null
export OBJ_BENCH_SKIP_REF=1
rm -rf /tmp/if1
rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/if1 -o p -- $R/tools/perf/_bin/obj_bench_c 9000 2 7,7,7,7 1 > /tmp/if1.log 2>&1
python3 - <<PY
import csv, glob, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/if1/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0]
        if 'objective_kernel' in k: acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k,d in acc.items():
    print(k, {c: sum(v)/len(v) for c,v in d.items()})
PY
