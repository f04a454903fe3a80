import csv,sys,glob
for tag in ("split","nosplit"):
    f=glob.glob(f"gpurun_out/r3/prof_{tag}/**/*kernel_stats.csv", recursive=True)[0]
    rows=list(csv.DictReader(open(f)))
    gemm=sum(float(r["TotalDurationNs"]) for r in rows if r["Name"].startswith(("Cijk","Custom_Cijk")))
    red=sum(float(r["TotalDurationNs"]) for r in rows if "reduce_kernel" in r["Name"] or "elementwise" in r["Name"] or "copy" in r["Name"].lower())
    tot=sum(float(r["TotalDurationNs"]) for r in rows)
    print(tag,"gemm_ms",round(gemm/1e6,1),"reduce/elementwise/copy_ms",round(red/1e6,1),"total_ms",round(tot/1e6,1))
    for r in rows[:12]:
        if r["Name"].startswith(("Cijk","Custom_Cijk")): print("   ",r["Calls"],round(float(r["AverageNs"])/1e3,1),"us",round(float(r["TotalDurationNs"])/1e6,1),"ms",r["Name"][:90])
