import csv, sys, collections
rows=[r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# steady state: last 40 % of dispatches; frames are sequences cover,raygen,closest,shade,tail,shadow,resolve on one queue
n=len(rows); rows=rows[int(n*0.5):]
byq=collections.defaultdict(list)
for r in rows: byq[r["Queue_Id"]].append(r)
gaps=collections.defaultdict(list); durs=collections.defaultdict(list)
for q,rs in byq.items():
    for a,b in zip(rs,rs[1:]):
        ka=a["Kernel_Name"].split("(")[0][-22:]; kb=b["Kernel_Name"].split("(")[0][-22:]
        gaps[ka+" -> "+kb].append((int(b["Start_Timestamp"])-int(a["End_Timestamp"]))/1e3)
    for a in rs: durs[a["Kernel_Name"].split("(")[0][-30:]].append((int(a["End_Timestamp"])-int(a["Start_Timestamp"]))/1e3)
for k,v in sorted(gaps.items(), key=lambda kv:-len(kv[1]))[:12]:
    v.sort(); print("%-50s n=%3d median gap %.1f us" % (k, len(v), v[len(v)//2]))
for k,v in durs.items():
    v.sort(); print("dur %-32s n=%3d median %.1f us" % (k,len(v),v[len(v)//2]))
