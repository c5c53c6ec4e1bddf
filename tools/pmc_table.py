import re,sys
txt=open(sys.argv[1]).read()
blocks=re.split(r'\n(?=\S)', txt)
rows=[]
for b in blocks:
    lines=b.strip().split('\n')
    if len(lines)<5: continue
    name=lines[0].strip(); d={}
    for l in lines[1:]:
        m=re.match(r'\s+(\S+)\s+(\d+)',l)
        if m: d[m.group(1)]=float(m.group(2))
    if 'SQ_WAVES' in d and d.get('SQ_BUSY_CYCLES',0)>0 and 'vamp' in name: rows.append((name,d))
print(f"{'kernel':46s} {'waves':>7s} {'valu/w':>7s} {'vmrd/w':>7s} {'vmwr/w':>7s} {'lds/w':>6s} {'life_us':>8s} {'active%':>7s} {'istall%':>7s} {'parked%':>7s} {'L2hit%':>6s} {'L2req':>9s}")
for name,d in sorted(rows,key=lambda r:-r[1].get('SQ_WAVE_CYCLES',0)):
    w=d['SQ_WAVES']; wc=d['SQ_WAVE_CYCLES']
    a=100*d['SQ_ACTIVE_INST_ANY']/wc; st=100*d['SQ_WAIT_INST_ANY']/wc
    print(f"{name[:46]:46s} {w:7.0f} {d['SQ_INSTS_VALU']/w:7.0f} {d['SQ_INSTS_VMEM_RD']/w:7.1f} {d['SQ_INSTS_VMEM_WR']/w:7.1f} {d['SQ_INSTS_LDS']/w:6.1f} {wc/w*4/2400:8.1f} {a:7.1f} {st:7.1f} {100-a-st:7.1f} {100*d['TCC_HIT_sum']/max(1,d['TCC_REQ_sum']):6.1f} {d['TCC_REQ_sum']:9.0f}")
