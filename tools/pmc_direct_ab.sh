# A/B of the two 8 -> 8 3x3 kernels (needs an -DMMSEG_AB build at csrc/libmmseg_hip_ab.so):  gpurun -- "bash tools/pmc_direct_ab.sh [BATCH]"
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pq; mkdir -p $O
export MMSEG_HIP_LIB=$R/multimodal_segmentation_amd/csrc/libmmseg_hip_ab.so
export BATCH=${1:-8}
cd /tmp; export TMPDIR=/tmp
for m in 0 1; do
  export MMSEG_DIRECT_MFMA=$m
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f$m -o p -- python3 $R/tools/conv_one.py 256 8 8 fwd > /dev/null 2> $O/f$m.err
  rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/a$m -o p -- python3 $R/tools/conv_one.py 256 8 8 fwd > /dev/null 2> $O/a$m.err
  rocprofv3 --pmc SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS --output-format csv -d $O/b$m -o p -- python3 $R/tools/conv_one.py 256 8 8 fwd > /dev/null 2> $O/b$m.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/t$m -o p -- python3 $R/tools/conv_one.py 256 8 8 fwd > /dev/null 2> $O/t$m.err
done
cd $R
python3 - <<'PY'
import csv,glob,os
O=os.path.join(os.environ['GRAFT_REPO_ROOT'],'gpurun_out/pq')
print('BATCH', os.environ['BATCH'])
for m in '01':
    for k in 'fab':
        for f in glob.glob(O+'/%s%s/**/*counter_collection.csv'%(k,m), recursive=True):
            tot={}
            for r in csv.DictReader(open(f)):
                if 'conv_direct' in r['Kernel_Name']:
                    tot.setdefault(r['Counter_Name'],[]).append(float(r['Counter_Value']))
            for c,v in tot.items(): print('mfma=%s %-28s per dispatch %.4g  (n %d)'%(m, c, sum(v)/len(v), len(v)))
    for f in glob.glob(O+'/t%s/**/*kernel_stats.csv'%m, recursive=True):
        for r in csv.DictReader(open(f)):
            if 'conv_direct' in r['Name']: print('mfma=%s'%m, r['Name'][:40], r['Calls'], r['AverageNs'])
PY
for d in f0 f1 a0 a1 b0 b1 t0 t1; do rm -rf $O/$d; done
