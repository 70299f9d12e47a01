import csv, glob, sys
rows=[]
for f in glob.glob(sys.argv[1]+"/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])-int(r["Start_Timestamp"]), r["Kernel_Name"]))
rows.sort()
per=320
out=[]
for g in range(len(rows)//per):
    grp=rows[g*per+20:(g+1)*per]; d=sorted(x[1] for x in grp); n=grp[0][2]
    out.append((d[len(d)//2]/1e3, n[n.index("gemm"):][:36]))
for i in range(0,len(out),3):
    print("%s  same %6.2f  160MB %6.2f  640MB %6.2f us" % (out[i][1], out[i][0], out[i+1][0], out[i+2][0]))
