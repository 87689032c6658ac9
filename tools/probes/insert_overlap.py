import csv,sys
rows=[]
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Kernel_Name"]
    if "k_bin" in n or "k_acc" in n:
        rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),"bin" if "k_bin" in n else "acc", r.get("Queue_Id","?")))
rows.sort()
t0=rows[0][0]
for s,e,k,q in rows[:24]:
    print("%s q%s start %9.3f ms  dur %7.3f ms"%(k,q,(s-t0)/1e6,(e-s)/1e6))
# overlap total
bins=[(s,e) for s,e,k,q in rows if k=="bin"]; accs=[(s,e) for s,e,k,q in rows if k=="acc"]
ov=0
for s,e in bins:
    for s2,e2 in accs:
        ov+=max(0,min(e,e2)-max(s,s2))
print("bin total %.1f ms, acc total %.1f ms, overlapped %.1f ms, span %.1f ms"%(sum(e-s for s,e in bins)/1e6,sum(e-s for s,e in accs)/1e6,ov/1e6,(max(e for s,e,_,_ in rows)-t0)/1e6))
