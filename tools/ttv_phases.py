import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
segs=[];cur={}
for r in rows:
    n=r['Kernel_Name']
    if 'transit' not in n: continue
    name=n.split('::')[1].split('(')[0]
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    if 'window' in name and cur and len(cur.get(name,[]))>=13:
        segs.append(cur); cur={}
    cur.setdefault(name,[]).append(d)
segs.append(cur)
for sgm in segs:
    print({k.replace('transit_','')[:40]:round(sum(v[3:])/max(len(v[3:]),1),1) for k,v in sgm.items() if 'window' not in k and 'reduce' not in k})
