import re,sys
from collections import Counter
s=open(sys.argv[1]).read()
pat=sys.argv[2]
minlen=int(sys.argv[3]) if len(sys.argv)>3 else 300
for m in re.finditer(r'^(_Z\S*%s\S*):[^\n]*\n(.*?)s_endpgm'%pat,s,re.S|re.M):
    lines=m.group(2).split('\n')
    ins=[];labels={}
    for l in lines:
        t=l.strip()
        if not t or t.startswith(';'): continue
        if re.match(r'^\.LBB\d+_\d+:',t):
            labels[t.split(':')[0]]=len(ins); continue
        if t.startswith('.'): continue
        ins.append(t)
    print(m.group(1)[:100], 'total',len(ins))
    for i,t in enumerate(ins):
        mm=re.match(r'(s_cbranch\w+|s_branch)\s+(\.LBB\d+_\d+)',t)
        if mm and mm.group(2) in labels and labels[mm.group(2)]<=i:
            a=labels[mm.group(2)]
            body=ins[a:i+1]
            if len(body)<minlen: continue
            v=[x.split()[0] for x in body if x.startswith('v_')]
            c=Counter(v)
            lit=sum(1 for x in body if x.startswith('v_mov_b') and x.split(',')[1].strip().startswith('0x'))
            print(f"  loop {mm.group(2)} [{a},{i}] n={len(body)} valu={len(v)} f64={sum(1 for x in v if 'f64' in x)} mov={c['v_mov_b32_e32']+c['v_mov_b64_e32']} litmov={lit} rdlane={c['v_readlane_b32']} wrlane={c['v_writelane_b32']} cnd={c['v_cndmask_b32_e32']+c['v_cndmask_b32_e64']} salu={sum(1 for x in body if x.startswith('s_'))} scratch={sum(1 for x in body if x.startswith('scratch'))}")
