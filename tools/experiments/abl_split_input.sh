out=gpurun_out/ablsplit; mkdir -p $out
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
for k in 0 1; do
  if [ $k = 1 ]; then export SF_ABL_SPLIT=1; fi
  rocprofv3 --kernel-trace --output-format csv -d $root/$out/prof$k -o p -- python3 $root/tools/bench_cnn.py --tiles 2048 --width 512 --batch 512 --lanes 1 --route split > $root/$out/prof$k.log 2>&1
done
python3 - <<'PY'
import csv,collections,os
root=os.environ.get('GRAFT_REPO_ROOT','.')
res={}
for k in (0,1):
    d=collections.defaultdict(list)
    import glob
    f=glob.glob(root+'/gpurun_out/ablsplit/prof%d/**/*kernel_trace.csv'%k, recursive=True)[0]
    for r in csv.DictReader(open(f)):
        n=r['Kernel_Name']
        if 'k_conv_split' not in n: continue
        d[(n[:52],int(r['Grid_Size_X']))].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
    res[k]=d
for key in sorted(set(res[0])|set(res[1]), key=lambda x:x[1]):
    if key[1] < 131072: continue
    a=res[0].get(key); b=res[1].get(key)
    med=lambda v: sorted(v)[len(v)//2] if v else -1
    print("%-54s %8d  fp32-in %7.1f  split-in %7.1f"%(key[0],key[1],med(a),med(b)))
PY
