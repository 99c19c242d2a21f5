export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for v in rot0 rot1; do
  BIN=./tools/conv_bench; [ $v = rot0 ] && BIN=./tools/conv_bench_rot0
  rm -rf gpurun_out/ldsc_$v
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS -d gpurun_out/ldsc_$v -o p --output-format csv -- $BIN 32 2 > /dev/null 2> gpurun_out/ldsc_$v.err || exit 1
done
python3 - <<'PY'
import csv,glob,collections
for v in ("rot0","rot1"):
    f=glob.glob(f"gpurun_out/ldsc_{v}/**/*counter_collection.csv",recursive=True)[0]
    agg=collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"][:70]
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    print(v)
    for k,c in agg.items():
        if c.get("SQ_LDS_IDX_ACTIVE",0)>0 and ("rbpair" in k or "conv_mfma" in k):
            print(f"  {k:70s} conflict/active {c['SQ_LDS_BANK_CONFLICT']/c['SQ_LDS_IDX_ACTIVE']:.3f}  idx_active {c['SQ_LDS_IDX_ACTIVE']:.3e} conflict {c['SQ_LDS_BANK_CONFLICT']:.3e}")
PY
