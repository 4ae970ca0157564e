import re,sys
from collections import Counter
s=open(sys.argv[1]).read()
for name in sys.argv[2:]:
    for m in re.finditer(r'^(_Z\S*%s\S*):[^\n]*\n(.*?)s_endpgm'%name,s,re.S|re.M):
        body=m.group(2)
        ins=[l.strip().split()[0] for l in body.split('\n') if l.strip() and l.startswith("\t") and not l.strip().startswith('.') and not l.strip().startswith(';')]
        print(m.group(1)[:90],len(ins),'valu',sum(1 for i in ins if i.startswith('v_')),'f64',sum(1 for i in ins if 'f64' in i), 'mem',Counter(i for i in ins if i.startswith(('global','scratch','ds_','buffer','flat'))))
        print(Counter(i for i in ins if i.startswith('v_')).most_common(24))
