"""The kernel launches of ONE call, in order, from a rocprofv3 --kernel-trace CSV: start offset, duration, grid, name.
usage: python profiles/launch_sequence.py <trace dir> <name fragment of the call's first kernel>
(e.g. WHICH=lidar rocprofv3 --kernel-trace -f csv -d out -o t -- python profiles/merged_only.py;
      python profiles/launch_sequence.py out merged_bundle)"""
import csv,glob,sys
f=glob.glob(sys.argv[1]+"/**/*kernel_trace.csv",recursive=True)[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if sys.argv[2] in r["Kernel_Name"]]
a=idx[-2]; b=idx[-1]
t0=int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    n=r["Kernel_Name"].replace("void rocprim::ROCPRIM_400200_NS::detail::","rp::").replace("vgx::(anonymous namespace)::","").replace("rocprim::ROCPRIM_400200_NS::","")
    print("%8.1f %6.1f  grid=%-8s %s"%((int(r["Start_Timestamp"])-t0)/1e3,(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3,r["Grid_Size_X"],n[:100]))
print(len(rows[a:b]),"launches, period %.1f us"%((int(rows[b]["Start_Timestamp"])-t0)/1e3))
